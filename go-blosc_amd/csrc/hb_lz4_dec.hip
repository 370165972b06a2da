// hb_lz4_dec.hip — LZ4 block decoder for gfx950: replaces lz4Codec.Decompress (codec.go:77-84, i.e.
// lz4.UncompressBlock of pierrec/lz4 v4.1.23) on the device.
//
// An LZ4 block is a serial chain, so there are two decoders:
//
//   k_dec_indexed : for blocks that come with a restart index (HBIX, see hb_lz4.h) — every block this
//       library encodes.  One wavefront per index unit (= one 4 KiB chunk of output), the unit's slice of the
//       stream staged in LDS.  FILL: the 64 lanes parse 64 stream bytes "as if a token started at my byte"; the
//       real token chain is followed with one s_bitset1_b64 + one v_readlane per token and the real tokens are
//       compacted into an LDS queue.  DRAIN: one queued token per lane — a wave scan gives the output positions,
//       every lane copies its own literals and its own match into an LDS image of the chunk (dependency rounds: a
//       match is ready when its source ends before the first pending match or lies in the lane's own literals;
//       overlapping matches by pattern replication), long ones are copied by the whole wave; the image is flushed
//       with coalesced 16-byte stores.  Literal-only units go HBM -> HBM.  Tokens with multi-byte length
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

struct DecPlan {
    uint32_t mode;        // 0 = serial, 1 = indexed
    uint32_t fail;        // set by any indexed unit that cannot vouch for its slice
    uint32_t nunits;
    uint32_t nbytes;      // decoded size the index declares
};
enum { DEC_SERIAL = 0, DEC_INDEXED = 1 };

size_t hb_lz4_dec_workspace(size_t) { return 256; }

__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return ld4u(p); }

// 1 thread: is there a usable index?
__global__ void k_dec_plan(const uint8_t *__restrict__ index, uint64_t index_bytes, uint64_t n_src, uint64_t cap,
                           DecPlan *plan, hb_result *result) {
    plan->mode = DEC_SERIAL; plan->fail = 0; plan->nunits = 0; plan->nbytes = 0;
    result->status = HB_OK; result->flags = 0; result->bytes = 0; result->total_bytes = 0; result->reserved = 0;
    if (!index || index_bytes < HB_IDX_HDR_BYTES + 2 * HB_IDX_ENTRY) return;
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = ld32(index + 4 * i);
    if (h[0] != HB_IDX_MAGIC || h[1] != (HB_IDX_VERSION | (HB_IDX_ENTRY << 16))) return;
    if (h[7] != (h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5])) return;
    const uint64_t nunits = h[2];
    if (nunits == 0 || HB_IDX_HDR_BYTES + (nunits + 1) * HB_IDX_ENTRY > index_bytes) return;
    if (h[4] != n_src || h[5] > cap) return;
    plan->nunits = (uint32_t)nunits;
    plan->nbytes = h[5];
    plan->mode = DEC_INDEXED;
}

#define DEC_IN_MAX  (HB_CHUNK + 512u)    // bytes of a unit's stream slice that are staged in LDS
#define DEC_OUT_MAX HB_CHUNK             // largest output a unit may have

// Sum of an LZ4 length extension (bytes 255 ... 255 r) starting at slice offset si, read cooperatively 64 bytes
// at a time; bytes beyond the staged window come straight from HBM (a literal run of many MiB has an
// extension of tens of KiB).  Returns false when the extension runs off the slice or is absurdly long.
__device__ __forceinline__ bool dec_read_ext(const uint8_t *in, uint32_t staged, const uint8_t *g, uint32_t slen,
                                             uint32_t &si, uint32_t &acc, int lane) {
    uint64_t sum = acc;
    for (uint32_t round = 0;; round++) {
        if (round == 1) {
            // 64 bytes of 255 and counting: a literal run of MiB.  Scan 4 KiB per round trip with four 16-byte
            // loads per lane in flight, straight from HBM/L2; the byte-granular loop below finishes the tail.
            for (;;) {
                bool allff = true;
                u32x4 v[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const uint32_t i = si + (uint32_t)k * 1024u + (uint32_t)lane * 16u;
                    if (i + 16u <= slen) v[k] = ld16u(g + i); else { v[k].x = 0; v[k].y = 0; v[k].z = 0; v[k].w = 0; }
                }
                uint32_t adv = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const bool ff = (v[k].x & v[k].y & v[k].z & v[k].w) == 0xFFFFFFFFu;
                    const unsigned long long bad = __ballot(!ff);
                    if (allff) {
                        if (bad) { adv += 16u * (uint32_t)__builtin_ctzll(bad); allff = false; }
                        else adv += 1024u;
                    }
                }
                sum += 255ull * adv; si += adv;
                if (sum > 0xFFFFFFF0ull) return false;
                if (!allff) break;
            }
        }
        const uint32_t i = si + lane;
        uint32_t b = 0;                                   // out of range reads as a terminator
        if (i < slen) b = (i < staged) ? in[i] : g[i];
        const unsigned long long stop = __ballot(b != 255u);
        if (stop == 0) { sum += 255u * 64u; si += 64; if (sum > 0xFFFFFFF0ull) return false; continue; }
        const int f = __builtin_ctzll(stop);
        if (si + (uint32_t)f >= slen) return false;
        sum += 255u * (uint32_t)f + (uint32_t)__builtin_amdgcn_readlane(b, f);
        si += (uint32_t)f + 1;
        if (sum > 0xFFFFFFF0ull) return false;
        acc = (uint32_t)sum;
        return true;
    }
}

#define DLITCAP 16u
#define DMCAP 32u             // matches up to this long are copied by their own lane

// 4 bytes at any byte address of a 4-byte aligned LDS array: two aligned dwords + v_alignbyte.
// gfx950 LDS also takes misaligned ds_read/ds_write_b16/b32/b64 (hipcc emits them for align-1 types) but
// replays them; they still beat byte loops for the short exact-length copies below.
__device__ __forceinline__ uint32_t dec_read4(const uint8_t *base, uint32_t a) {
    const uint32_t *w = (const uint32_t *)base;
    const uint32_t w0 = w[a >> 2], w1 = w[(a >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
}

// exact-length copy inside LDS, no overlap between [d, d+len) and [s, s+len): 8/4/2/1-byte pieces
__device__ __forceinline__ void lds_copy_exact(uint8_t *d, const uint8_t *s, uint32_t len) {
    uint32_t k = 0;
    for (; k + 8u <= len; k += 8u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(s + k))->v;
    if (len & 4u) { ((hb_u32u *)(d + k))->v = ((const hb_u32u *)(s + k))->v; k += 4u; }
    if (len & 2u) { ((hb_u16u *)(d + k))->v = ((const hb_u16u *)(s + k))->v; k += 2u; }
    if (len & 1u) d[k] = s[k];
}
// a match copied by one lane: out[md + k] = out[md - off + k]; pieces never read bytes they have not written yet
__device__ __forceinline__ void lds_match_lane(uint8_t *out, uint32_t md, uint32_t off, uint32_t len) {
    if (off >= 8u) { lds_copy_exact(out + md, out + md - off, len); return; }
    if (off == 1u) {                                            // run of one byte
        const uint64_t pat = 0x0101010101010101ull * out[md - 1u];
        uint32_t k = 0;
        for (; k + 8u <= len; k += 8u) ((hb_u64u *)(out + md + k))->v = pat;
        if (len & 4u) { ((hb_u32u *)(out + md + k))->v = (uint32_t)pat; k += 4u; }
        if (len & 2u) { ((hb_u16u *)(out + md + k))->v = (uint16_t)pat; k += 2u; }
        if (len & 1u) out[md + k] = (uint8_t)pat;
        return;
    }
    for (uint32_t k = 0; k < len; k++) out[md + k] = out[md - off + k];   // offsets 2..7: byte by byte
}
__device__ __forceinline__ uint32_t dec_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// copy a match inside the LDS image of the chunk; all arguments wave-uniform, source fully produced
__device__ __forceinline__ void dec_match_copy(uint8_t *s_out, uint32_t md, uint32_t off, uint32_t ml, int lane) {
    if (off >= 64u || off >= ml) {
        for (uint32_t c0 = 0; c0 < ml; c0 += 64) {
            const uint32_t i = c0 + lane;
            uint8_t v = 0;
            if (i < ml) v = s_out[md + i - off];
            if (i < ml) s_out[md + i] = v;
        }
    } else {                                                   // overlapping: the source is [md-off, md), repeated
        uint32_t m = (uint32_t)lane % off;
        const uint32_t step = 64u % off;
        for (uint32_t c0 = 0; c0 < ml; c0 += 64) {
            const uint32_t i = c0 + lane;
            if (i < ml) s_out[md + i] = s_out[md - off + m];
            m += step; if (m >= off) m -= off;
        }
    }
}

__global__ __launch_bounds__(64) void k_dec_indexed(const uint8_t *__restrict__ src, uint64_t n_src,
                                                    uint8_t *__restrict__ dst, const uint8_t *__restrict__ index,
                                                    DecPlan *plan) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_IN_MAX + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[DEC_OUT_MAX + 64];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[128];     // parsed tokens waiting for their lane
    if (plan->mode != DEC_INDEXED) return;
    const int lane = threadIdx.x;
    const uint32_t nunits = plan->nunits;
    const uint8_t *ent = index + HB_IDX_HDR_BYTES;
    const uint32_t nbytes = plan->nbytes;

    // Units are visited in a scrambled order (u = i * P mod nunits, P odd-ish and coprime to nunits, about
    // nunits/4): neighbouring workgroups then work on far-apart parts of the buffer (different byte planes of a
    // shuffled frame: issue-bound token-dense units next to bandwidth-bound literal-only ones), and the odd grid
    // size rotates the mix from pass to pass.  Any order is correct: units are independent.
    uint32_t P = nunits / 4u + 1u;
    for (;;) {                                              // gcd(P, nunits) == 1
        uint32_t x = P, y = nunits;
        while (y) { const uint32_t t = x % y; x = y; y = t; }
        if (x == 1u) break;
        P++;
    }
    for (uint32_t it = blockIdx.x; it < nunits; it += gridDim.x) {
        const uint32_t u = (uint32_t)(((uint64_t)it * P) % nunits);
        const u32x4 e0 = ld16u(ent + 16 * (size_t)u), e1 = ld16u(ent + 16 * (size_t)(u + 1));
        const uint32_t s0 = e0.x, d0 = e0.y, s1 = e1.x, d1 = e1.y, rem1 = e1.z, tok1 = e1.w;
        uint32_t rem = e0.z, tokpos = e0.w;
        const bool last = (u + 1 == nunits);
        bool ok = s0 <= s1 && s1 <= n_src && d0 <= d1 && d1 <= nbytes && (d1 - d0) <= DEC_OUT_MAX;
        if (u == 0) ok = ok && s0 == 0 && d0 == 0 && rem == HB_IDX_AT_TOKEN;
        if (last) ok = ok && s1 == n_src && d1 == nbytes;
        if (rem != HB_IDX_AT_TOKEN && tokpos >= n_src) ok = false;
        if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); continue; }
        const uint32_t slen = s1 - s0, outlen = d1 - d0;
        const uint8_t *g = src + s0;

        // the whole unit lies inside one literal run (incompressible chunk): HBM -> HBM, no LDS
        if (rem != HB_IDX_AT_TOKEN && rem >= outlen) {
            const uint32_t left = rem - outlen;
            bool fine = slen == outlen;
            if (last) fine = fine && left == 0; else fine = fine && rem1 == left && tok1 == tokpos;
            if (!fine) { if (lane == 0) atomicExch(&plan->fail, 1u); continue; }
            wave_copy_g2g(dst + d0, g, outlen, lane);
            continue;
        }

        // stage the head of the slice (all of it, normally)
        const uint32_t sh = (uint32_t)((uintptr_t)g & 15u);
        const uint32_t staged = slen < DEC_IN_MAX - 16u ? slen : DEC_IN_MAX - 16u;
        {
            const u32x4 *ga = (const u32x4 *)(g - sh);
            const uint32_t nv = (sh + staged + 15u) >> 4;
            for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_in)[i] = ga[i];
        }
        uint32_t tok = 0;
        if (rem != HB_IDX_AT_TOKEN) tok = src[tokpos];
        wave_sync();
        const uint8_t *in = s_in + sh;
        uint32_t si = 0, di = 0;
        bool at_token = false;       // state when the unit stops
        bool done = false;

        // State machine.  slow != 0: handle ONE sequence of any shape (length extensions of any size, bytes
        // outside the staged window, end of the unit inside a literal run); slow == 2 starts at a token,
        // slow == 1 inside the literal run (rem, tok) the unit begins in.  slow == 0: the window parser below.
        int slow = 1;
        uint32_t nq = 0;                                        // tokens queued in s_tq
        if (rem == HB_IDX_AT_TOKEN) { rem = 0; slow = 0; }
        const uint32_t lim = staged;                            // the window parser only looks at staged bytes
        while (ok && !done) {
            if (slow) {
                if (slow == 2) {
                    if (si >= slen) { ok = false; break; }
                    tokpos = s0 + si;
                    tok = (si < staged) ? in[si] : g[si];
                    si++;
                    rem = tok >> 4;
                    if (rem == 15u && !dec_read_ext(in, staged, g, slen, si, rem, lane)) { ok = false; break; }
                }
                slow = 0;
                {   // literal phase
                    const uint32_t take = min(rem, outlen - di);
                    if (take > slen - si) { ok = false; break; }
                    if (si + take <= staged) { for (uint32_t i = lane; i < take; i += 64) s_out[di + i] = in[si + i]; }
                    else { for (uint32_t i = lane; i < take; i += 64) s_out[di + i] = g[si + i]; }
                    si += take; di += take; rem -= take;
                }
                if (rem > 0 || di == outlen) { at_token = false; done = true; break; }
                // match phase
                if (slen - si < 2) { ok = false; break; }             // also: block ends after literals -> serial decides
                const uint32_t b0 = (si < staged) ? in[si] : g[si], b1 = (si + 1 < staged) ? in[si + 1] : g[si + 1];
                const uint32_t offset = b0 | (b1 << 8);
                si += 2;
                uint32_t mlen = (tok & 15u) + 4u;
                if ((tok & 15u) == 15u && !dec_read_ext(in, staged, g, slen, si, mlen, lane)) { ok = false; break; }
                if (offset == 0 || offset > di || mlen > outlen - di) { ok = false; break; }
                dec_match_copy(s_out, di, offset, mlen, lane);
                di += mlen;
                continue;
            }
            // ---- fill: parse windows of 64 stream bytes until 64 tokens are queued, the slice ends, or a token needs the
            //      slow path (length extension > 1 byte, too close to the edge of the staged bytes) ----
            bool stop = false;
            while (nq < 64u && !stop) {
                if (si == slen) { stop = true; break; }
                // every lane parses "as if a token started at my byte"
                const uint32_t base = si, p = base + (uint32_t)lane;
                const uint32_t w = dec_read4(s_in, sh + p);
                const uint32_t t = w & 255u;
                uint32_t lit = t >> 4, nbl = 0;
                bool cplx = p >= lim;
                if (lit == 15u) { const uint32_t b1 = (w >> 8) & 255u; if (b1 == 255u) cplx = true; else { lit = 15u + b1; nbl = 1; } }
                const uint32_t lsrc = p + 1u + nbl, offpos = lsrc + lit;
                if (offpos + 3u > lim) cplx = true;              // literal-only tail, or too close to the edge
                const uint32_t x = dec_read4(s_in, sh + (cplx ? 0u : offpos));
                const uint32_t offv = x & 0xFFFFu, mb = (x >> 16) & 255u, mn = t & 15u;
                uint32_t mlen = 4u + mn, nbm = 0;
                if (mn == 15u) { if (mb == 255u) cplx = true; else { mlen = 19u + mb; nbm = 1; } }
                const uint32_t nxt = offpos + 2u + nbm;
                // follow the real token chain through the window: one bit-set + one readlane per token; a
                // "complex" lane ends the walk (its successor is >= 64)
                const unsigned long long cmask = __ballot(cplx);
                unsigned long long tmask = 0;
                uint32_t cur;
                {
                    const uint32_t nrel = cplx ? 64u : nxt - base;
                    uint32_t j = 0;
                    do {
                        asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j));
                        j = __builtin_amdgcn_readlane(nrel, (int)j);
                    } while (j < 64u);
                    cur = base + j;
                    const unsigned long long cm = tmask & cmask;  // at most the last visited lane
                    if (cm) { tmask &= ~cm; cur = base + (uint32_t)__builtin_ctzll(cm); stop = true; }
                }
                // queue the real tokens, compacted in stream order: {lsrc | lit << 13 | mlen << 22, offset | tokpos << 16}
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tmask, 0u));
                if ((tmask >> lane) & 1ull) {
                    uint2 e; e.x = lsrc | (lit << 13) | (mlen << 22); e.y = offv | (p << 16);
                    s_tq[nq + rank] = e;
                }
                nq += (uint32_t)__builtin_popcountll(tmask);
                si = cur;
            }
            // ---- drain: one queued token per lane ----
            bool rewound = false;
            while (nq >= 64u || (stop && nq > 0u)) {
                const uint32_t cntb = nq < 64u ? nq : 64u;
                const uint2 e = s_tq[lane];
                const uint32_t lsrc = e.x & 0x1FFFu, lit = (e.x >> 13) & 0x1FFu, mlen = e.x >> 22;
                const uint32_t offv = e.y & 0xFFFFu, tp = e.y >> 16;
                const uint32_t olen = (uint32_t)lane < cntb ? lit + mlen : 0u;
                const uint32_t incl = dec_incl_scan(olen, lane);
                const uint32_t dpos = di + incl - olen;
                uint32_t total = __builtin_amdgcn_readlane(incl, 63);
                unsigned long long amask = cntb >= 64u ? ~0ull : ((1ull << cntb) - 1ull);
                const unsigned long long om = __ballot(olen != 0u && dpos + olen > outlen);
                if (om) {                                        // this sequence passes the end of the unit: slow path from its token
                    const int jx = __builtin_ctzll(om);
                    amask &= (1ull << jx) - 1ull;
                    total = __builtin_amdgcn_readlane(dpos, jx) - di;
                    si = __builtin_amdgcn_readlane(tp, jx);
                    rewound = true;
                }
                const bool istok = (amask >> lane) & 1ull;
                // a match may only read what this unit has produced (else: not ours to decide -> serial decoder)
                if (__ballot(istok && (offv == 0u || offv > dpos + lit))) { ok = false; break; }
                // literals: short runs by their own lane, long runs by the whole wave
                if (istok && lit <= DLITCAP) lds_copy_exact(s_out + dpos, in + lsrc, lit);
                unsigned long long lm = __ballot(istok && lit > DLITCAP);
                while (lm) {
                    const int l = __builtin_ctzll(lm);
                    const uint32_t sp = __builtin_amdgcn_readlane(lsrc, l), dp = __builtin_amdgcn_readlane(dpos, l);
                    const uint32_t ln = __builtin_amdgcn_readlane(lit, l);
                    for (uint32_t k = lane; k < ln; k += 64) s_out[dp + k] = in[sp + k];
                    lm &= lm - 1;
                }
                // matches: every lane copies its own short match as soon as its source is final.  Everything before the
                // first pending match is final, so each round retires at least that one; matches longer than DMCAP
                // bytes are copied by the whole wave when they come first.
                const uint32_t mdv = dpos + lit;                               // where my match goes
                const uint32_t srcend = mdv - offv + (mlen < offv ? mlen : offv); // end of the source that is not my own output
                const uint32_t mend = mdv + mlen;                              // end of my match
                unsigned long long pend = amask;
                while (pend) {
                    const int f = __builtin_ctzll(pend);
                    const uint32_t X = __builtin_amdgcn_readlane(mdv, f);
                    const uint32_t mlf = __builtin_amdgcn_readlane(mlen, f);
                    if (mlf > DMCAP) {
                        dec_match_copy(s_out, X, __builtin_amdgcn_readlane(offv, f), mlf, lane);
                        pend &= pend - 1;
                        continue;
                    }
                    // ready: the source ends before the first pending match, or starts at / after the end of the
                    // nearest pending match before me (everything between that and my own match is final: literals
                    // and retired matches; with no pending predecessor that is simply "anything before me")
                    const unsigned long long below = pend & ((1ull << lane) - 1ull);
                    const uint32_t pj = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe = (uint32_t)__shfl((int)mend, (int)pj);
                    const bool ready = ((pend >> lane) & 1ull) && mlen <= DMCAP &&
                                       (srcend <= X || below == 0ull || mdv - offv >= pe);
                    if (ready) lds_match_lane(s_out, mdv, offv, mlen);
                    pend &= ~__ballot(ready);
                }
                di += total;
                if (rewound) { nq = 0; break; }
                const uint2 rest = s_tq[64 + lane];              // keep what is queued beyond the 64 just decoded
                nq -= cntb;
                if ((uint32_t)lane < nq) s_tq[lane] = rest;
            }
            if (!ok) break;
            if (rewound) { slow = 2; continue; }
            if (stop) {
                if (si == slen || di == outlen) { at_token = true; done = true; }
                else slow = 2;
            }
        }
        // end-state check against the next entry
        if (ok) {
            ok = (si == slen) && (di == outlen);
            if (last) ok = ok && (at_token || rem == 0);
            else if (at_token) ok = ok && rem1 == HB_IDX_AT_TOKEN;
            else ok = ok && rem1 == rem && tok1 == tokpos;
        }
        if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); wave_sync(); continue; }
        wave_sync();
        // flush the chunk image
        {
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
}

// agent-scope relaxed byte load: served by L2, never by this CU's L1 (the wave re-reads bytes it stored)
__device__ __forceinline__ uint8_t ld_l2(const uint8_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One wavefront, whole block, front to back.  Semantics of lz4.UncompressBlock (see oracle/blosc_oracle.c
// ob_lz4_decompress for the restated rules).
__global__ __launch_bounds__(64) void k_dec_serial(const uint8_t *__restrict__ src, uint64_t n_src,
                                                   uint8_t *__restrict__ dst, uint64_t cap, DecPlan *plan,
                                                   hb_result *result, int frame, uint32_t expect) {
    const int lane = threadIdx.x;
    if (plan->mode == DEC_INDEXED && !plan->fail) {
        if (lane == 0) {
            const uint64_t got = plan->nbytes;
            result->flags = 1; result->bytes = got; result->total_bytes = got;
            result->status = (frame && got != expect) ? HB_ERR_SIZE_MISMATCH : HB_OK;     // blosc.go:429-431
        }
        return;
    }
    uint64_t si = 0, di = 0;
    int err = 0;
    while (si < n_src) {
        const uint32_t b = src[si++];
        uint64_t ll = b >> 4;
        if (ll == 15) {
            for (;;) {
                if (si >= n_src) { err = 1; break; }
                const uint32_t x = src[si++];
                ll += x;
                if (x != 255u) break;
            }
            if (err) break;
        }
        if (ll) {
            if (ll > n_src - si || ll > cap - di) { err = 1; break; }
            for (uint64_t off = 0; off < ll; off += 1u << 30) {
                const uint32_t part = (uint32_t)min((uint64_t)1u << 30, ll - off);
                wave_copy_g2g(dst + di + off, src + si + off, part, lane);
            }
            si += ll; di += ll;
        }
        uint64_t ml = b & 15u;
        if (si == n_src && ml == 0) break;
        if (si >= n_src || n_src - si < 2) { err = 1; break; }
        const uint32_t offset = (uint32_t)src[si] | ((uint32_t)src[si + 1] << 8);
        if (offset == 0) { err = 1; break; }
        si += 2;
        ml += 4;
        if (ml == 19) {
            for (;;) {
                if (si >= n_src) { err = 1; break; }
                const uint32_t x = src[si++];
                ml += x;
                if (x != 255u) break;
            }
            if (err) break;
        }
        if (di < offset || ml > cap - di) { err = 1; break; }
        // the source bytes were stored by this wave: wait for the stores, then read them from L2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint8_t *m = dst + di - offset;
        if (offset >= 64u) {
            for (uint64_t b0 = 0; b0 < ml; b0 += 64) {
                const uint64_t i = b0 + lane;
                uint8_t v = 0;
                if (i < ml) v = ld_l2(m + i);
                if (i < ml) dst[di + i] = v;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        } else {
            uint32_t mm = (uint32_t)lane % offset;
            const uint32_t step = 64u % offset;
            uint8_t pat = ld_l2(m + mm);
            for (uint64_t b0 = 0; b0 < ml; b0 += 64) {
                const uint64_t i = b0 + lane;
                if (step) pat = ld_l2(m + mm);
                if (i < ml) dst[di + i] = pat;
                mm += step; if (mm >= offset) mm -= offset;
            }
        }
        di += ml;
    }
    if (lane == 0) {
        result->flags = 0;
        if (err) { result->status = HB_ERR_DECOMPRESSION_FAILED; result->bytes = 0; }            // blosc.go:411-413
        else if (frame && di != expect) { result->status = HB_ERR_SIZE_MISMATCH; result->bytes = di; }   // blosc.go:429-431
        else { result->status = HB_OK; result->bytes = di; }
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
    hb_prof_begin("k_dec_plan", s);
    hipLaunchKernelGGL(k_dec_plan, dim3(1), dim3(1), 0, s, a.index, (uint64_t)a.index_bytes, (uint64_t)a.n,
                       (uint64_t)a.cap, plan, a.result);
    hb_prof_end(s);
    if (a.index) {
        const uint64_t units = (a.cap + HB_CHUNK - 1) / HB_CHUNK;
        const unsigned grid = (unsigned)(units < 1 ? 1 : (units < 256u * 64u + 1u ? units : 256u * 64u + 1u));   // odd when capped
        hb_prof_begin("k_dec_indexed", s);
        hipLaunchKernelGGL(k_dec_indexed, dim3(grid), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, a.index, plan);
        hb_prof_end(s);
    }
    hb_prof_begin("k_dec_serial", s);
    hipLaunchKernelGGL(k_dec_serial, dim3(1), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, (uint64_t)a.cap, plan,
                       a.result, a.frame, a.expect);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
