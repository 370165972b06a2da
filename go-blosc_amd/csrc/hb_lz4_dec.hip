// hb_lz4_dec.hip — LZ4 block decoder for gfx950: replaces lz4Codec.Decompress (codec.go:77-84, i.e.
// lz4.UncompressBlock of pierrec/lz4 v4.1.23) on the device.
//
// An LZ4 block is a serial chain, so there are two decoders:
//
//   k_dec_indexed : for blocks that come with a restart index (HBIX, see hb_lz4.h) — every block this
//       library encodes.  One wavefront per index unit (= one 4 KiB chunk of output), the unit's slice of the
//       stream staged in LDS through a moving 2.5 KiB window (7.4 KiB of LDS per wave, 20 waves per CU).  FILL: the 64 lanes parse 64 stream bytes "as if a token started at my byte"; the
//       real token chain is followed with one s_bitset1_b64 + one v_readlane per token and the real tokens are
//       compacted into an LDS queue.  DRAIN: one queued token per lane — a wave scan gives the output positions,
//       every lane copies its own literals and its own match into an LDS image of the chunk (dependency rounds: a
//       match is ready when its source ends before the first pending match or lies in the lane's own literals;
//       overlapping matches by pattern replication), long ones are copied by the whole wave; the image is flushed
//       with coalesced 16-byte stores -- or, for byte-shuffled frames with typesize 2 / 4, straight to the
//       un-shuffled positions with byte-strided stores (filter fused; bitshuffle with typesize 4 likewise, as an
//       in-place transform of the image).  Literal-only units go HBM -> HBM.  Tokens with multi-byte length
//       extensions, or at the edges of the unit, take a one-sequence-at-a-time slow path.  The index is NOT
//       trusted: each unit checks that it ends exactly in the state the next entry claims (stream offset, output
//       offset, literals left in the current run, token position) and that no match reaches before its
//       own output.  By induction over the units the result is then byte-identical to a serial decode.
//       Any violation only raises a flag ...
//   k_dec_serial  : ... in which case (or when there is no index: streams produced by the reference) a
//       single wavefront decodes the whole block front to back through HBM.  Correct for every input the
//       reference decoder accepts, rejects what it rejects (offset 0, offset before the start of the
//       block, truncated input, output overflow); slow.
//
// Algorithmic HBM bytes: C (read) + n (write).
#include "hb_lz4.h"
#include "hb_dec_common.h"

size_t hb_lz4_dec_workspace(size_t n_out) { return 256 + hb_lz4_region_workspace(n_out); }

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return ld4u(p); }

// 1 thread: is there a usable index?
__device__ __forceinline__ void dec_plan_check(const uint8_t *__restrict__ index, uint64_t index_bytes, uint64_t n_src, uint64_t cap,
                                               DecPlan *plan, hb_result *result) {
    plan->mode = DEC_SERIAL; plan->fail = 0; plan->nunits = 0; plan->nbytes = 0; plan->post = 0; plan->stride = 1;
    result->status = HB_OK; result->flags = 0; result->bytes = 0; result->total_bytes = 0; result->reserved = 0;
    if (!index || index_bytes < HB_IDX_HDR_BYTES + 2 * HB_IDX_ENTRY) return;
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = ld32(index + 4 * i);
    if (h[0] != HB_IDX_MAGIC || h[1] != (HB_IDX_VERSION | (HB_IDX_ENTRY << 16))) return;
    if (h[7] != (h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5])) return;
    const uint64_t nunits = h[2];
    if (nunits == 0 || HB_IDX_HDR_BYTES + (nunits + 1) * HB_IDX_ENTRY > index_bytes) return;
    if (h[4] != n_src || h[5] > cap) return;
    // one unit per HB_CHUNK bytes of output, exactly: the launch shape and the fused un-shuffle's unit order are derived from
    // the header's nbytes, so an index with any other unit geometry (even a self-consistent one) is not used
    if (h[3] != HB_CHUNK || nunits != ((uint64_t)h[5] + HB_CHUNK - 1) / HB_CHUNK) return;
    plan->nunits = (uint32_t)nunits;
    plan->nbytes = h[5];
    // unit order of the un-fused launch (see k_dec_indexed): a quarter turn per step through the groups of 8 units
    const uint32_t m = ((uint32_t)nunits + 7u) / 8u;
    uint32_t P = m / 4u + 1u;
    for (;;) {
        uint32_t x = P, y = m;
        while (y) { const uint32_t t = x % y; x = y; y = t; }
        if (x == 1u) break;
        P++;
    }
    plan->stride = P;
    plan->mode = DEC_INDEXED;
}
__global__ void k_dec_plan(const uint8_t *__restrict__ index, uint64_t index_bytes, uint64_t n_src, uint64_t cap,
                           DecPlan *plan, hb_result *result) {
    dec_plan_check(index, index_bytes, n_src, cap, plan, result);
}

#ifndef DEC_IN_WIN
#define DEC_IN_WIN  1536u                // bytes of a unit's stream slice that are staged in LDS at a time (the window moves)
#endif
#define DEC_IN_MARGIN 320u               // a token closer than this to the end of the window is parsed after re-staging
#define DEC_OUT_MAX HB_CHUNK             // largest output a unit may have
#ifndef DEC_LEAN
#define DEC_LEAN 1                       // the window parser only finds the token chain; the drain parses the tokens it decodes (hb_dec_common.h)
#endif
#ifndef DEC_WAVES
#define DEC_WAVES 6                      // with the 1.5 KiB window and the u16 token queue: 6 KiB of LDS per wave, 79 VGPRs -- 1.42 -> 1.32 ms against 5 waves and a 2.5 KiB window
#endif

// One index unit (4 KiB of output), one wavefront: everything the indexed decoder does for unit `u` of a block.  The block comes as a
// DecCtx so that the same code serves one frame (k_dec_indexed: the context is the kernel's arguments) and batches of frames
// (k_dec_indexed_batch: the context of the unit's frame).
struct DecCtx {
    const uint8_t *src; uint64_t n_src;      // the LZ4 block
    uint8_t *dst;                            // decoded (and, with a fused un-filter, un-filtered) bytes
    const uint8_t *ent;                      // index entries
    DecPlan *plan;
    uint32_t nbytes, nunits;
    int bun4, ush;
};
// lab ablations of the fused un-shuffle flush (tools/lab/ab.py; the frames decode to garbage under 1 / 2: timing only)
#if defined(LAB_DEC_FLUSH) && LAB_DEC_FLUSH == 1
#define LAB_FLUSH_USH() do { } while (0)
#elif defined(LAB_DEC_FLUSH) && LAB_DEC_FLUSH == 2
#define LAB_FLUSH_USH() do { for (uint32_t i = lane * 16u; i < outlen; i += 1024u) st16u(dst + d0 + i, *(const u32x4 *)(s_out + i)); } while (0)
#else
// (unrolled by eight: the loop control of 64 single-byte rounds is 200 scalar instructions per unit otherwise -- the decoder issues more scalar than vector instructions)
#define LAB_FLUSH_USH() do { \
        uint32_t i_ = lane; \
        for (; i_ + 448u < outlen; i_ += 512u) { \
            _Pragma("unroll") for (uint32_t k_ = 0; k_ < 8u; k_++) udst[(size_t)(i_ + 64u * k_) * (uint32_t)ush] = s_out[i_ + 64u * k_]; \
        } \
        for (; i_ < outlen; i_ += 64) udst[(size_t)i_ * (uint32_t)ush] = s_out[i_]; \
    } while (0)
#endif
__device__ __forceinline__ void dec_unit(const DecCtx &c, const uint32_t u, uint8_t *s_in, uint8_t *s_out, uint2 *s_tq, const int lane) {
    const uint8_t *const src = c.src; const uint64_t n_src = c.n_src; uint8_t *const dst = c.dst; const uint8_t *const ent = c.ent;
    DecPlan *const plan = c.plan; const uint32_t nbytes = c.nbytes, nunits = c.nunits; const int bun4 = c.bun4, ush = c.ush;
    const u32x4 e0 = ld16u(ent + 16 * (size_t)u), e1 = ld16u(ent + 16 * (size_t)(u + 1));
    // wave-uniform values that come out of vector loads are moved to scalar registers: the compiler cannot know
    // they are uniform, and would otherwise run the whole state machine on the vector side under exec masks
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
    const uint32_t s0 = RFL(e0.x), d0 = RFL(e0.y), s1 = RFL(e1.x), d1 = RFL(e1.y), rem1 = RFL(e1.z), tok1 = RFL(e1.w);
    uint32_t rem = RFL(e0.z), tokpos = RFL(e0.w);
    const bool last = (u + 1 == nunits);
    bool ok = s0 <= s1 && s1 <= n_src && d0 <= d1 && d1 <= nbytes && (d1 - d0) <= DEC_OUT_MAX;
    if (u == 0) ok = ok && s0 == 0 && d0 == 0 && rem == HB_IDX_AT_TOKEN;
    if (last) ok = ok && s1 == n_src && d1 == nbytes;
    if (rem != HB_IDX_AT_TOKEN && tokpos >= n_src) ok = false;
    if (bun4 && ((d0 | d1) & 31u)) ok = false;              // fused un-filter works on whole 32-byte windows
    const uint32_t ne = ush ? nbytes / (uint32_t)ush : 1u;  // bytes per plane
    const uint32_t pj = ush ? d0 / ne : 0u;                 // my plane
    if (ush && (pj >= (uint32_t)ush || d1 > (pj + 1u) * ne)) ok = false;   // a unit never straddles two planes
    uint8_t *const udst = ush ? dst + (size_t)(d0 - pj * ne) * (uint32_t)ush + pj : dst + d0;
    if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); return; }
    const uint32_t slen = s1 - s0, outlen = d1 - d0;
    const uint8_t *g = src + s0;

    // the whole unit lies inside one literal run (incompressible chunk): HBM -> HBM, no LDS
    if (rem != HB_IDX_AT_TOKEN && rem >= outlen) {
        const uint32_t left = rem - outlen;
        bool fine = slen == outlen;
        // a block that ends inside/after a literal run is accepted only if that token announces no match
        // (UncompressBlock: si == len(src) && matchNibble == 0; oracle/blosc_oracle.c ob_lz4_decompress) -- else the serial decoder decides
        if (last) fine = fine && left == 0 && (RFL((uint32_t)src[tokpos]) & 15u) == 0u; else fine = fine && rem1 == left && tok1 == tokpos;
        if (!fine) { if (lane == 0) atomicExch(&plan->fail, 1u); return; }
        if (ush) {                                          // wide loads (any alignment) into the image, then the strided stores
            for (uint32_t i = lane * 16u; i < outlen; i += 1024u) {
                if (i + 16u <= outlen) *(u32x4 *)(s_out + i) = ld16u(g + i);
                else for (uint32_t r = i; r < outlen; r++) s_out[r] = g[r];
            }
            wave_sync();
            LAB_FLUSH_USH();
            wave_sync();
        }
        else if (!bun4) wave_copy_g2g(dst + d0, g, outlen, lane);
        else {
            for (uint32_t w = lane; w < outlen / 32u; w += 64) {
                u32x4 oa, ob;
                bitshuffle4_window<true>(ld16u(g + 32u * w), ld16u(g + 32u * w + 16u), oa, ob);
                st16u(dst + d0 + 32u * w, oa);
                st16u(dst + d0 + 32u * w + 16u, ob);
            }
        }
        return;
    }

    // stage a window of the slice (all of it, for a dense unit); `at` = slice position the window has to start at.
    // LDS byte k of s_in is global byte g - sh + a16 + k, a16 a multiple of 16: 16-byte aligned vector loads.
    const uint32_t sh = (uint32_t)((uintptr_t)g & 15u);
    uint32_t wlo = 0, staged = 0;                           // the window holds slice positions [wlo, staged)
    int shw = 0;                                            // LDS index of slice position p = p + shw
    auto stage = [&](const uint32_t at) __attribute__((always_inline)) {
        wave_sync();
        const uint32_t a16 = (sh + at) & ~15u;
        const uint32_t avail = sh + slen - a16;
        const uint32_t cnt = avail < DEC_IN_WIN ? avail : DEC_IN_WIN;
        const u32x4 *ga = (const u32x4 *)(g - sh + a16);
        const uint32_t nv = (cnt + 15u) >> 4;
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_in)[i] = ga[i];
        wlo = a16 > sh ? a16 - sh : 0u;
        staged = a16 + cnt - sh;
        shw = (int)sh - (int)a16;
        wave_sync();
    };
    stage(0u);
    uint32_t tok = 0;
    if (rem != HB_IDX_AT_TOKEN) tok = RFL((uint32_t)src[tokpos]);
#define INB(i) s_in[(uint32_t)((int)(i) + shw)]              /* stream byte at slice position i (inside the window) */
    uint32_t si = 0, di = 0;
    bool at_token = false;       // state when the unit stops
    bool done = false;

    // State machine.  slow != 0: handle ONE sequence of any shape (length extensions of any size, bytes
    // outside the staged window, end of the unit inside a literal run); slow == 2 starts at a token,
    // slow == 1 inside the literal run (rem, tok) the unit begins in.  slow == 0: the window parser below.
    int slow = 1;
    uint32_t nq = 0;                                        // tokens queued in s_tq
    uint32_t last_ntok = DEC_BPERM_MIN; (void)last_ntok;    // tokens the last parsed window held (dec_fill_lean picks its chain walk by it)
    if (rem == HB_IDX_AT_TOKEN) { rem = 0; slow = 0; }
    while (ok && !done) {
        if (slow) {
            if (slow == 2) {
                if (si >= slen) { ok = false; break; }
                tokpos = s0 + si;
                tok = RFL((uint32_t)((si >= wlo && si < staged) ? INB(si) : g[si]));
                si++;
                rem = tok >> 4;
                if (rem == 15u && !dec_read_ext(s_in, shw, wlo, staged, g, slen, si, rem, lane)) { ok = false; break; }
            }
            slow = 0;
            {   // literal phase
                const uint32_t take = min(rem, outlen - di);
                if (take > slen - si) { ok = false; break; }
                if (si >= wlo && si + take <= staged) { for (uint32_t i = lane; i < take; i += 64) s_out[di + i] = INB(si + i); }
                else { for (uint32_t i = lane; i < take; i += 64) s_out[di + i] = g[si + i]; }
                si += take; di += take; rem -= take;
            }
            if (rem > 0 || di == outlen) { at_token = false; done = true; break; }
            // match phase
            if (slen - si < 2) { ok = false; break; }             // also: block ends after literals -> serial decides
            const uint32_t b0 = RFL((uint32_t)((si >= wlo && si < staged) ? INB(si) : g[si]));
            const uint32_t b1 = RFL((uint32_t)((si + 1 >= wlo && si + 1 < staged) ? INB(si + 1) : g[si + 1]));
            const uint32_t offset = b0 | (b1 << 8);
            si += 2;
            uint32_t mlen = (tok & 15u) + 4u;
            if ((tok & 15u) == 15u && !dec_read_ext(s_in, shw, wlo, staged, g, slen, si, mlen, lane)) { ok = false; break; }
            if (offset == 0 || offset > di || mlen > outlen - di) { ok = false; break; }
            dec_match_copy(s_out, di, offset, mlen, lane);
            di += mlen;
            // peek: a next token with a multi-byte match extension would come straight back from the window parser
            // (highly compressible units are a handful of such tokens): stay on this path
            if (si >= wlo && si + 4u <= staged && di < outlen) {
                const uint32_t t2 = RFL((uint32_t)INB(si)), l2 = t2 >> 4;
                if (l2 < 15u && (t2 & 15u) == 15u) {
                    const uint32_t op = si + 1u + l2;
                    if (op + 3u <= staged && RFL((uint32_t)INB(op + 2u)) == 255u) slow = 2;
                }
            }
            continue;
        }
        if (nq == 0u && staged < slen && si + DEC_IN_MARGIN > staged) {     // move the window (queued tokens point into it)
            stage(si);
        }
#if DEC_LEAN
        const bool stop = dec_fill_lean(s_in, (uint32_t)shw, staged, slen, si, nq, s_tq, lane, last_ntok);
        bool rewound = false;
        ok = dec_drain<true>(s_in, shw, s_out, outlen, 0u, di, si, nq, s_tq, stop, rewound, lane);
#else
        const bool stop = dec_fill(s_in, (uint32_t)shw, staged, slen, si, nq, s_tq, lane);
        bool rewound = false;
        ok = dec_drain(s_in, shw, s_out, outlen, 0u, di, si, nq, s_tq, stop, rewound, lane);
#endif
        if (!ok) break;
        if (rewound) { slow = 2; continue; }
        if (stop) {
            if (si == slen || di == outlen) { at_token = true; done = true; }
            else if (staged < slen && si + DEC_IN_MARGIN > staged) continue;   // stopped at the end of the window, not at a complex token
            else slow = 2;
        }
    }
    // end-state check against the next entry
    if (ok) {
        ok = (si == slen) && (di == outlen);
        if (last) ok = ok && (at_token || (rem == 0 && (tok & 15u) == 0u));   // ends after literals: the token must announce no match
        else if (at_token) ok = ok && rem1 == HB_IDX_AT_TOKEN;
        else ok = ok && rem1 == rem && tok1 == tokpos;
    }
    if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); wave_sync(); return; }
    wave_sync();
    if (bun4) {                                             // fused bit-unshuffle: every window in place
        for (uint32_t w = lane; w < outlen / 32u; w += 64) {
            u32x4 oa, ob;
            bitshuffle4_window<true>(((const u32x4 *)s_out)[2 * w], ((const u32x4 *)s_out)[2 * w + 1], oa, ob);
            ((u32x4 *)s_out)[2 * w] = oa;
            ((u32x4 *)s_out)[2 * w + 1] = ob;
        }
        wave_sync();
    }
    // flush the chunk image
    if (ush) {
        LAB_FLUSH_USH();
    } else {
        uint8_t *o = dst + d0;
        uint32_t head = (uint32_t)((16u - ((uintptr_t)o & 15u)) & 15u);
        if (head > outlen) head = outlen;
        if ((uint32_t)lane < head) o[lane] = s_out[lane];
        const uint32_t body = (outlen - head) >> 4;
        if (head == 0) { for (uint32_t i = lane; i < body; i += 64) *(u32x4 *)(o + i * 16u) = *(const u32x4 *)(s_out + i * 16u); }
        else {
            for (uint32_t i = lane; i < body; i += 64) {
                const uint8_t *q = s_out + head + i * 16u;
                u32x4 v;
                v.x = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                v.y = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
                v.z = (uint32_t)q[8] | ((uint32_t)q[9] << 8) | ((uint32_t)q[10] << 16) | ((uint32_t)q[11] << 24);
                v.w = (uint32_t)q[12] | ((uint32_t)q[13] << 8) | ((uint32_t)q[14] << 16) | ((uint32_t)q[15] << 24);
                *(u32x4 *)(o + head + i * 16u) = v;
            }
        }
        const uint32_t done_b = head + body * 16u;
        if (done_b + lane < outlen) o[done_b + lane] = s_out[done_b + lane];
    }
    wave_sync();
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DEC_WAVES))) void k_dec_indexed(const uint8_t *__restrict__ src, uint64_t n_src,
                                                    uint8_t *__restrict__ dst, const uint8_t *__restrict__ index,
                                                    DecPlan *plan, int bun4, int ush, uint32_t plane_mask) {
    // ush != 0: the frame was byte-shuffled with typesize `ush` and has only whole planes of whole chunks; the un-shuffle is
    // fused: a unit is a piece of ONE byte plane j, and its byte i goes straight to dst[(e0 + i) * ush + j] with byte
    // stores (64 lanes cover 64 * ush bytes; the other planes' waves fill in the rest of those lines, and the
    // memory-side cache merges the partial lines before they reach HBM) -- no filtered buffer, no un-shuffle pass.
    // bun4 != 0: the frame was bitshuffled with typesize 4 -- an in-place transform of every 32-byte window -- and the
    // un-filter is fused: every unit un-shuffles its own windows before they leave the chip (dst is the final output).
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN_WIN + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DEC_OUT_MAX + 64];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DEC_LEAN ? DTQ / 4 : DTQ];     // tokens waiting for their lane (lean: their positions, u16)
    if (plan->mode != DEC_INDEXED) return;
    const int lane = threadIdx.x;
    const uint32_t nunits = plan->nunits;
    const uint8_t *ent = index + HB_IDX_HDR_BYTES;
    const uint32_t nbytes = plan->nbytes;

    // Unit order.  Workgroups are placed on the 8 XCDs round-robin (workgroup i -> XCD i % 8), and the byte planes of a shuffled
    // frame differ 3x in cost and are contiguous quarters of the unit range: an order in which workgroup i takes plane i % 4
    // (the obvious "next plane per workgroup" scramble) gives every plane to TWO of the eight XCDs for the whole launch, and the
    // launch lasts as long as the two with the token-dense plane need (measured: 2.28 ms against 1.41 for the same work).  So:
    // workgroup `it` = (step k = it / 8, XCD x = it % 8) takes unit 8 * (k * P mod m) + (x + k) % 8, m = groups of 8 units,
    // P coprime to m and about m / 4: every XCD walks the whole buffer a quarter turn per step (its resident waves are an even
    // mix of all planes: issue-bound token-dense units next to bandwidth-bound literal ones), and the rotation by k takes
    // it through all residues mod 8 (streams that interleave planes with period 2 / 4 / 8 -- hb_cblosc.hip -- stay balanced
    // too).  The grid is a multiple of 8, so the XCD of a workgroup does not change from pass to pass.  Any order is correct:
    // units are independent.
    const uint32_t P = plan->stride;                        // computed once by k_dec_plan
    const uint32_t mgrp = (nunits + 7u) / 8u;
    for (uint32_t it = blockIdx.x; it < mgrp * 8u; it += gridDim.x) {
        uint32_t u = (uint32_t)(((uint64_t)(it >> 3) * P) % mgrp) * 8u + ((it + (it >> 3)) & 7u);
        const bool fused_order = ush && nunits % (uint32_t)ush == 0u && (gridDim.x % (8u * (uint32_t)ush) == 0u || gridDim.x >= nunits);
        if (fused_order ? it >= nunits : u >= nunits) continue;
        if (fused_order) {
            // fused un-shuffle: the `ush` units that make up one 4096-element block write interleaved bytes of the same
            // lines, so they get workgroup ids that are equal mod 8 (same XCD under round-robin placement: the
            // partial lines meet in one L2) and run close in time; the plane rotates with the pass (planes differ in
            // cost).  Within a group of 8 blocks, work item (plane j, block b % 8) has index j * 8 + b % 8.
            const uint32_t T = (uint32_t)ush, nblk = nunits / T;
            const uint32_t grp = it / (8u * T), r = it % (8u * T);
            uint32_t b = grp * 8u + (r & 7u), j = ((r >> 3) + it / gridDim.x) % T;
            if (grp * 8u + 8u > nblk) {                         // ragged last group: plain (b, j) order
                const uint32_t k = it - grp * 8u * T, nb = nblk - grp * 8u;
                b = grp * 8u + k % nb; j = k / nb;
            }
            u = j * nblk + b;
            if (!((plane_mask >> j) & 1u)) continue;            // hb_debug_plane_mask: per-plane timing
        }
        DecCtx c; c.src = src; c.n_src = n_src; c.dst = dst; c.ent = ent; c.plan = plan; c.nbytes = nbytes; c.nunits = nunits; c.bun4 = bun4; c.ush = ush;
        dec_unit(c, u, s_in, s_out, s_tq, lane);
    }
}

// ---- batches of frames (hb_decompress_frames_batch_dev): the indexed decoder over the units of ALL frames of a batch ----
// frame of every global unit (gaps between frames: no frame), and every frame's plan (one workgroup per frame)
__global__ __launch_bounds__(64) void k_bt_dec_plan(const DecBatchFrame *__restrict__ bf, uint32_t nframes, uint32_t *__restrict__ unit_frame, uint32_t total_units) {
    const uint32_t f = blockIdx.x;
    const DecBatchFrame b = bf[f];
    const uint32_t next = f + 1 < nframes ? bf[f + 1].unit0 : total_units;
    __shared__ uint32_t s_mode;
    if (threadIdx.x == 0) {
        dec_plan_check(b.preset == 1 ? b.index : nullptr, b.index_bytes, b.n_src, (uint64_t)b.nbytes, b.plan, b.result);
        b.plan->pad[0] = 0; b.plan->pad[1] = 0;                      // (stream decoder's verdict and byte count, k_bt_dec_streams)
        // an index that speaks of another geometry than the frame's header (fewer bytes, other units) is no index: the stream decides
        if (b.plan->mode == DEC_INDEXED && (b.plan->nunits != b.nunits || b.plan->nbytes != b.nbytes)) b.plan->mode = DEC_SERIAL;
        s_mode = b.plan->mode;
    }
    __syncthreads();
    // the units of a frame whose index did not check out belong to nobody: k_dec_indexed_batch then needs no look at the plan (one dependent memory
    // round trip less in front of every unit)
    const bool indexed = s_mode == DEC_INDEXED;
    for (uint32_t u = threadIdx.x; u < next - b.unit0; u += 64) unit_frame[b.unit0 + u] = (u < b.nunits && indexed) ? f : 0xFFFFFFFFu;
}

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(DEC_WAVES))) void k_dec_indexed_batch(const DecBatchFrame *__restrict__ bf,
                                                    const uint32_t *__restrict__ unit_frame, uint32_t total_units) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN_WIN + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DEC_OUT_MAX + 64];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DEC_LEAN ? DTQ / 4 : DTQ];
    const int lane = threadIdx.x;
    for (uint32_t it = blockIdx.x; it < total_units; it += gridDim.x) {
        const uint32_t fid = unit_frame[it];
        if (fid == 0xFFFFFFFFu) continue;
        const DecBatchFrame &f = bf[fid];
        DecCtx c; c.src = f.src; c.n_src = f.n_src; c.dst = f.dst; c.ent = f.index + HB_IDX_HDR_BYTES; c.plan = f.plan;
        c.nbytes = f.nbytes; c.nunits = f.nunits; c.bun4 = f.bun4; c.ush = f.ush;       // (= the plan's: k_bt_dec_plan gave the units to this frame only then)
        uint32_t u = it - f.unit0;
        if (c.ush) {
            // fused un-shuffle: as in k_dec_indexed -- the `ush` units of an element block get workgroup ids equal mod 8 (a frame's
            // first unit is a multiple of 8 * ush in the flat space) and the plane rotates with the pass
            const uint32_t T = (uint32_t)c.ush, nblk = c.nunits / T;
            const uint32_t grp = u / (8u * T), r = u % (8u * T);
            uint32_t b = grp * 8u + (r & 7u), j = ((r >> 3) + it / gridDim.x) % T;
            if (grp * 8u + 8u > nblk) { const uint32_t k = u - grp * 8u * T, nb = nblk - grp * 8u; b = grp * 8u + k % nb; j = k / nb; }
            u = j * nblk + b;
        }
        if (u >= c.nunits) continue;
        dec_unit(c, u, s_in, s_out, s_tq, lane);
    }
}

// unit0 of every frame (host): frames with a fused byte un-shuffle start at a multiple of 8 * typesize; returns the size of the flat space
size_t hb_lz4_dec_batch_units(int nframes, const DecBatchFrame *h, uint32_t *unit0_out) {
    uint64_t units = 0;
    for (int k = 0; k < nframes; k++) {
        const uint32_t g = h[k].ush ? 8u * (uint32_t)h[k].ush : 8u;
        units = (units + g - 1) / g * g;
        unit0_out[k] = (uint32_t)units;
        units += h[k].nunits;
    }
    return (size_t)((units + 7) / 8 * 8);
}

int hb_launch_lz4_decode_batch_indexed(int nframes, const DecBatchFrame *d_bf, uint32_t *d_unit_frame, uint32_t total_units, int any_ush, hipStream_t s) {
    hb_prof_begin("k_dec_plan", s);
    hipLaunchKernelGGL(k_bt_dec_plan, dim3((unsigned)nframes), dim3(64), 0, s, d_bf, (uint32_t)nframes, d_unit_frame, total_units);
    hb_prof_end(s);
    if (total_units) {
        unsigned grid = total_units < 256u * 256u ? total_units : 256u * 256u;
        if (any_ush && total_units > 256u * 16u) {                    // as for one frame: `typesize` passes per workgroup in large batches
            const unsigned gran = 32u;                               // a multiple of 8 * typesize for every fused typesize (2, 4): the groups of
            unsigned g2 = total_units / (unsigned)any_ush / gran * gran;   // 8 * typesize work items of a frame must share their pass number
            if (total_units > 256u * 64u && g2 < 256u * 64u) g2 = 256u * 64u;
            if (g2 > 256u * 256u) g2 = 256u * 256u;
            if (g2) grid = g2;
        }
        hb_prof_begin("k_dec_indexed", s);
        hipLaunchKernelGGL(k_dec_indexed_batch, dim3(grid), dim3(64), 0, s, d_bf, (const uint32_t *)d_unit_frame, total_units);
        hb_prof_end(s);
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}

// ------------------------------------------------------------------------------------------------------------
// k_dec_serial: streams without (or with a rejected) index, e.g. frames written by the reference.  One wavefront,
// whole block, front to back -- an LZ4 block is one serial chain and nothing says where its tokens are.  It still
// uses the window-parallel token parser and the lane-parallel copies of the indexed decoder: the stream is staged
// through an 8 KiB LDS window, the output is built in an LDS image that keeps the last 64 KiB as match history
// (offsets are 16 bits) in front of a 32 KiB page; the page is flushed to HBM with 16-byte stores and the image is
// shifted down.  Semantics of lz4.UncompressBlock (restated in oracle/blosc_oracle.c ob_lz4_decompress): empty
// input -> 0 bytes; the stream may end right after a match; offset 0, offset before the start of the block,
// truncated input and output overflow are errors.
// ------------------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_dec_serial(const uint8_t *__restrict__ src, uint64_t n_src,
                                                   uint8_t *__restrict__ dst, uint64_t cap, DecPlan *plan,
                                                   hb_result *result, int frame, uint32_t expect, int mark_post) {
    // mark_post: dst is a staging buffer (the un-filter was fused into the indexed decoder); when this kernel
    // really decodes it raises plan->post so that the gated un-filter pass behind it runs
    __shared__ __attribute__((aligned(16))) uint8_t s_win[SER_WIN + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_img[SER_HIST + SER_PAGE + 1024];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    if (plan->mode == DEC_INDEXED && !plan->fail) {
        if (lane == 0) {
            const uint64_t got = plan->nbytes;
            result->flags = 1; result->bytes = got; result->total_bytes = got;
            result->status = (frame && got != expect) ? HB_ERR_SIZE_MISMATCH : HB_OK;     // blosc.go:429-431
        }
        return;
    }
    if (mark_post && lane == 0) plan->post = 1;
    int err;
    const uint64_t got = dec_serial_core(src, n_src, dst, cap, s_win, s_img, s_tq, lane, err);
    if (lane == 0) {
        result->flags = 0; result->total_bytes = 0;
        if (err) { result->status = HB_ERR_DECOMPRESSION_FAILED; result->bytes = 0; }                    // blosc.go:411-413
        else if (frame && got != expect) { result->status = HB_ERR_SIZE_MISMATCH; result->bytes = got; }   // blosc.go:429-431
        else { result->status = HB_OK; result->bytes = got; }
    }
}

__global__ void k_set_result(hb_result *result, int status, uint64_t bytes) {
    result->status = status; result->flags = 0; result->bytes = bytes; result->total_bytes = bytes; result->reserved = 0;
}

int hb_launch_lz4_decode(const hb_dec_args &a, hipStream_t s) {
    if (a.memcpy_payload) {                                           // blosc.go:398-400
        if (a.n != a.expect) {                                        // -> blosc.go:429-431
            hipLaunchKernelGGL(k_set_result, dim3(1), dim3(1), 0, s, a.result, HB_ERR_SIZE_MISMATCH, (uint64_t)a.n);
        } else {
            if (a.n) HB_HIP_TRY(hipMemcpyAsync(a.dst, a.src, a.n, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_set_result, dim3(1), dim3(1), 0, s, a.result, HB_OK, (uint64_t)a.n);
        }
        HB_HIP_TRY(hipGetLastError());
        return HB_OK;
    }
    DecPlan *plan = (DecPlan *)a.work;
    const uint8_t *index = a.index;
    size_t index_bytes = a.index_bytes;
    // no index at all (a frame written without the trailer): rebuild it from the stream (hb_lz4_region.hip); what comes out is
    // checked like a stored index, and a block that was not written chunk-locally simply ends up with the single wavefront
    if (hb_lz4_region_wanted(a)) {
        const int rc = hb_launch_lz4_region_index(a, &index, &index_bytes, s);
        if (rc) return rc;
    }
    hb_prof_begin("k_dec_plan", s);
    hipLaunchKernelGGL(k_dec_plan, dim3(1), dim3(1), 0, s, index, (uint64_t)index_bytes, (uint64_t)a.n,
                       (uint64_t)a.cap, plan, a.result);
    hb_prof_end(s);
    if (index) {
        const uint64_t units = (a.cap + HB_CHUNK - 1) / HB_CHUNK;
        // a few units per workgroup (measured on 1 GiB: 16384 workgroups 1.84 ms, 65536 1.70 ms, 131072 2.09 ms, one unit per
        // workgroup 3.56 ms); a multiple of 8: a workgroup stays on its XCD from pass to pass (unit order in the kernel)
        const uint64_t units8 = (units + 7) / 8 * 8;
        unsigned grid = (unsigned)(units8 < 8 ? 8 : (units8 < 256u * 256u ? units8 : 256u * 256u));
        if (a.fused_unshuffle_ts && units > 256u * 16u) {
            // fused un-shuffle, beyond 16 MiB: `typesize` passes per workgroup where possible (it then meets every plane
            // once), and a multiple of 8 * typesize (see the unit order in the kernel)
            const unsigned gran = 8u * (unsigned)a.fused_unshuffle_ts;
            unsigned g2 = (unsigned)(units / (uint64_t)a.fused_unshuffle_ts) / gran * gran;
            if (units > 256u * 64u && g2 < 256u * 64u) g2 = 256u * 64u;
            if (g2 > 256u * 256u) g2 = 256u * 256u;
            if (g2) grid = g2;
        }
        hb_prof_begin("k_dec_indexed", s);
        hipLaunchKernelGGL(k_dec_indexed, dim3(grid), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, index, plan,
                           a.fused_bitunshuffle4, a.fused_unshuffle_ts, hb_dbg_plane_mask());
        hb_prof_end(s);
    }
    // with a fused un-filter the indexed decoder wrote FINAL bytes to a.dst; the serial decoder (if it has to run)
    // produces filtered bytes, so it goes to the staging buffer and the gated un-filter pass finishes the job
    const int fused_any = a.fused_bitunshuffle4 || a.fused_unshuffle_ts;
    uint8_t *serial_dst = fused_any ? a.staged : a.dst;
    // a block without an index whose rebuilt index did not hold was written by someone else (the reference: one block, 64 KiB
    // window): decoded in parallel from the verified token chain, symbolically (hb_lz4_sym.hip); no-op when the index held
    if (hb_lz4_region_wanted(a) && a.sym_work) {
        const int rc = hb_launch_lz4_sym_decode(a, serial_dst, a.sym_work, fused_any, s);
        if (rc) return rc;
    }
    hb_prof_begin("k_dec_serial", s);
    hipLaunchKernelGGL(k_dec_serial, dim3(1), dim3(64), 0, s, a.src, (uint64_t)a.n, serial_dst, (uint64_t)a.cap, plan,
                       a.result, a.frame, a.expect, fused_any);
    hb_prof_end(s);
    if (fused_any) {
        const int rc = a.fused_bitunshuffle4
                           ? hb_launch_filter_gated(HB_OP_BITUNSHUFFLE, a.dst, a.staged, a.expect, 4, &plan->post, s)
                           : hb_launch_filter_gated(HB_OP_UNSHUFFLE, a.dst, a.staged, a.expect, a.fused_unshuffle_ts, &plan->post, s);
        if (rc) return rc;
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
