// hb_dec_common.h — lane-parallel copy machinery shared by the LZ4 and Snappy block decoders (hb_lz4_dec.hip,
// hb_snappy.hip): a queue of parsed tokens {literal run, match} in LDS, one token per lane, output positions from a wave
// scan, every lane copies its own literals and its own match into an LDS image of the output, matches in dependency rounds.
#pragma once
#include "hb_lz4.h"

struct DecPlan {
    uint32_t mode;        // 0 = serial, 1 = indexed
    uint32_t fail;        // set by any indexed unit that cannot vouch for its slice
    uint32_t nunits;
    uint32_t nbytes;      // decoded size the index declares
    uint32_t post;        // set by k_dec_serial when it decoded into the staging buffer: the gated un-filter must run
    uint32_t stride;      // unit order of the indexed decoder (k_dec_indexed): coprime to the number of groups of 8 units
    uint32_t pad[2];
};
enum { DEC_SERIAL = 0, DEC_INDEXED = 1 };

// hb_lz4_region.hip: rebuilds the restart index of an LZ4 block that came without one (guess-and-verify token discovery)
size_t hb_lz4_region_workspace(size_t n_out);
bool hb_lz4_region_wanted(const hb_dec_args &a);
int hb_launch_lz4_region_index(const hb_dec_args &a, const uint8_t **index, size_t *index_bytes, hipStream_t s);
// hb_lz4_sym.hip: decodes a block whose rebuilt index did not hold (a foreign block) from the verified token chain
int hb_launch_lz4_sym_decode(const hb_dec_args &a, uint8_t *dst, uint8_t *sym_work, int mark_post, hipStream_t s, int codec = 0 /* RG_LZ4; 1 = RG_SNAPPY: elements */);

// ---- batches of frames in one set of launches (hb_decompress_frames_batch_dev): frame f of the batch as the kernels see it ----
struct DecBatchFrame {
    const uint8_t *src; uint64_t n_src;      // the payload: an LZ4 block (or the stored bytes of a memcpy frame)
    uint8_t *dst;                            // target of the indexed decoder (final bytes when the un-filter is fused into it)
    uint8_t *serial_dst;                     // target of the stream / serial decoders (bytes still filtered)
    const uint8_t *index; uint64_t index_bytes;   // restart index behind NBytesComp, if the frame brought one
    DecPlan *plan; hb_result *result;
    uint32_t nbytes;                         // NBytesOrig: capacity and expected length (blosc.go:429-431)
    uint32_t unit0, nunits;                  // place in the flat unit space of the batch (nunits = 0: no index, one wavefront decodes the stream)
    int32_t preset;                          // 1: decode; else the status the host already knows (memcpy frames, blosc.go:398-400)
    int32_t ush, bun4;                       // un-filter fused into the indexed decoder (as hb_dec_args)
    int32_t post_needed, pad;                // the stream / serial decoders' output still needs the (gated) un-filter pass
};
size_t hb_lz4_dec_batch_units(int nframes, const DecBatchFrame *h, uint32_t *unit0_out);
int hb_launch_lz4_decode_batch_indexed(int nframes, const DecBatchFrame *d_bf, uint32_t *d_unit_frame, uint32_t total_units, int any_ush, hipStream_t s);

#ifndef DEC_BPERM_WALK
#define DEC_BPERM_WALK 1
#endif
#ifndef DEC_BPERM_MIN
#define DEC_BPERM_MIN 8u                 // tokens in the previous window from which the chain is followed by pointer doubling (dec_fill_lean)
#endif
#ifndef DTQ
#define DTQ 96                           // token queue slots: < 64 queued before a window is parsed; a 64-byte window adds <= 22 LZ4 tokens or <= 32 Snappy elements
#endif

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
    // offsets 2..7: the first eo bytes one by one, eo = the smallest multiple of the period that is >= 8 (8, 9, 8, 10, 12, 14) -- from there
    // on out[md + k] = out[md + k - eo] reads 8 bytes that are all written already (64 bytes: 8-14 + 7 steps instead of 64)
    const uint32_t eo = (0xECA898u >> (4u * (off - 2u))) & 15u;
    uint32_t k = 0;
    const uint32_t head = len < eo ? len : eo;
    for (; k < head; k++) out[md + k] = out[md - off + k];
    for (; k + 8u <= len; k += 8u) ((hb_u64u *)(out + md + k))->v = ((const hb_u64u *)(out + md + k - eo))->v;
    if (k + 4u <= len) { ((hb_u32u *)(out + md + k))->v = ((const hb_u32u *)(out + md + k - eo))->v; k += 4u; }
    for (; k < len; k++) out[md + k] = out[md + k - eo];
}
__device__ __forceinline__ uint32_t dec_incl_scan(uint32_t v, int lane) { (void)lane; return wave_incl_scan_dpp(v); }

// copy a match inside the LDS image of the chunk; all arguments wave-uniform, source fully produced
__device__ __forceinline__ void dec_match_copy(uint8_t *s_out, uint32_t md, uint32_t off, uint32_t ml, int lane) {
    if (off >= 64u || off >= ml) {
        for (uint32_t c0 = 0; c0 < ml; c0 += 64) {
            const uint32_t i = c0 + lane;
            uint8_t v = 0;
            if (i < ml) v = s_out[md + i - off];
            if (i < ml) s_out[md + i] = v;
        }
    } else if (off <= 4u && off != 3u && ml >= 128u) {
        // a long run with period 1, 2 or 4 (what a shuffled plane of slowly varying values is made of): the period
        // divides 4, so every aligned dword of the destination holds the same rotated pattern -- 16 bytes per lane
        uint8_t *d = s_out + md;
        uint32_t w;                                             // the period, repeated to 4 bytes (md >= off: no read before the image)
        if (off == 1u) w = (uint32_t)d[-1] * 0x01010101u;
        else if (off == 2u) w = ((uint32_t)d[-2] | ((uint32_t)d[-1] << 8)) * 0x00010001u;
        else w = (uint32_t)d[-4] | ((uint32_t)d[-3] << 8) | ((uint32_t)d[-2] << 16) | ((uint32_t)d[-1] << 24);
        // w = bytes for positions j = 0..3 (mod 4) relative to md; destination dwords start at j = head (mod 4)
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
        const uint32_t wr = __builtin_amdgcn_alignbyte(w, w, head & 3u);
        if ((uint32_t)lane < head) d[lane] = (uint8_t)(w >> (8u * ((uint32_t)lane & 3u)));
        const uint32_t nblk = (ml - head) >> 4;
        u32x4 pat; pat.x = wr; pat.y = wr; pat.z = wr; pat.w = wr;
        for (uint32_t b = lane; b < nblk; b += 64) *(u32x4 *)(d + head + 16u * b) = pat;
        const uint32_t done = head + 16u * nblk;
        if (done + (uint32_t)lane < ml) d[done + lane] = (uint8_t)(wr >> (8u * ((uint32_t)lane & 3u)));
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

// DRAIN: one queued token per lane, while 64 are queued (or `stop` and any are).  out[0] is the first byte of the image,
// `hist` bytes before it are valid match sources, `outlen` is the room.  A token that does not fit is not decoded:
// `rewound` is set and si goes back to that token.  Returns false on a match that reaches before out[-hist] or has offset 0.
// LEAN (the indexed LZ4 decoder, round 4): the queue holds only the POSITIONS of the tokens (u16, slice-relative; dec_fill_lean) and the
// fields are parsed here, by 64 lanes that all hold a real token -- the window parser then spends its instructions on finding the chain,
// not on parsing 64 "as if" tokens of which a dozen are real.  A token whose length extension is longer than one byte is not decoded:
// like a sequence that passes the end of the unit it rewinds to the one-sequence-at-a-time path.
// MERGE (the Snappy decoders): queued copies that follow each other at the same distance, no literals in between, are decoded as ONE copy -- a Snappy copy
// ends at 64 bytes, a run of KiB is dozens of them, each waiting for the one before (out[p + k] = out[p + k - off] over all of them is the same bytes).
template <bool LEAN = false, bool MERGE = false>
__device__ __forceinline__ bool dec_drain(const uint8_t *in, const int inoff, uint8_t *s_out, const uint32_t outlen, const uint32_t hist,
                                          uint32_t &di, uint32_t &si, uint32_t &nq, uint2 *s_tq, const bool stop,
                                          bool &rewound, const int lane) {
    bool ok = true;
    while (nq >= 64u || (stop && nq > 0u)) {
        const uint32_t cntb = nq < 64u ? nq : 64u;
        uint32_t lsrc, lit, mlen, offv, tp;
        unsigned long long xm = 0;                       // LEAN: tokens this path does not decode (multi-byte length extensions)
        if constexpr (LEAN) {
            tp = ((const uint16_t *)s_tq)[lane];
            const uint32_t w = dec_read4(in, (uint32_t)((int)tp + inoff));
            const uint32_t t = w & 255u, b1 = (w >> 8) & 255u;
            const bool big = t >= 0xF0u;
            lit = big ? 15u + b1 : (t >> 4);
            lsrc = (uint32_t)((int)tp + inoff) + (big ? 2u : 1u);
            const uint32_t x = dec_read4(in, lsrc + lit);
            offv = x & 0xFFFFu;
            const uint32_t mb = (x >> 16) & 255u, mn = t & 15u;
            mlen = mn == 15u ? 19u + mb : 4u + mn;
            xm = hb_ballot((big && b1 == 255u) || (mn == 15u && mb == 255u));
        } else {
            const uint2 e = s_tq[lane];
            lsrc = (uint32_t)((int)(e.x & 0x1FFFu) + inoff); lit = (e.x >> 13) & 0x1FFu; mlen = e.x >> 22;
            offv = e.y & 0xFFFFu; tp = e.y >> 16;
            if constexpr (MERGE) {
                const bool cp = (uint32_t)lane < cntb && mlen != 0u && lit == 0u;
                const uint32_t poff = wave_shr1(offv, 0u);
                const bool pcp = wave_shr1((uint32_t)((uint32_t)lane < cntb && mlen != 0u), 0u) != 0u;      // (the lane before ends in a copy, literals in front of it or not)
                const bool cont = cp && pcp && lane != 0 && offv == poff;
                const unsigned long long heads = ~hb_ballot(cont);
                const uint32_t incl = dec_incl_scan(((uint32_t)lane < cntb && mlen != 0u) ? mlen : 0u, lane);
                const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;
                const uint32_t endl = above ? (uint32_t)lane + (uint32_t)__builtin_ctzll(above) : 63u;
                const uint32_t incl_end = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(endl << 2), (int)incl);
                if (cont) mlen = 0u;
                else if ((uint32_t)lane < cntb && mlen != 0u) mlen = incl_end - (incl - mlen);
            }
        }
        const uint32_t olen = (uint32_t)lane < cntb ? lit + mlen : 0u;
        const uint32_t incl = dec_incl_scan(olen, lane);
        const uint32_t dpos = di + incl - olen;
        uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        unsigned long long amask = cntb >= 64u ? ~0ull : ((1ull << cntb) - 1ull);
        const unsigned long long om = (hb_ballot(olen != 0u && dpos + olen > outlen) | xm) & amask;
        if (om) {                                        // this sequence passes the end of the unit: slow path from its token
            const int jx = __builtin_ctzll(om);
            amask &= (1ull << jx) - 1ull;
            total = __builtin_amdgcn_readlane(dpos, jx) - di;
            si = __builtin_amdgcn_readlane(tp, jx);
            rewound = true;
        }
        const bool istok = hb_lane_in(amask);
        // a match may only read what this unit has produced (else: not ours to decide -> serial decoder)
        // (a token with mlen == 0 carries no match: the Snappy decoder queues literal elements that way)
        if (hb_ballot(istok && mlen != 0u && (offv == 0u || offv > dpos + lit + hist))) { ok = false; break; }
        // literals: short runs by their own lane, long runs by the whole wave
#if !(defined(LAB_DEC) && (LAB_DEC & 1))     /* lab ablation bit 0: no literal copies (garbage output, timing only) */
        if (istok && lit <= DLITCAP) lds_copy_exact(s_out + dpos, in + lsrc, lit);
#endif
        unsigned long long lm = hb_ballot(istok && lit > DLITCAP);
#if defined(LAB_DEC) && (LAB_DEC & 1)
        lm = 0;
#endif
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
        const int src0 = (int)mdv - (int)offv;                             // < 0: the source starts in the history
        const int srcend = src0 + (int)(mlen < offv ? mlen : offv);        // end of the source that is not my own output
        const uint32_t mend = mdv + mlen;                              // end of my match
        unsigned long long pend = amask & hb_ballot(mlen != 0u);
#if defined(LAB_DEC) && (LAB_DEC & 2)        /* lab ablation bit 1: no match copies */
        pend = 0;
#endif
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
            const bool ready = hb_lane_in(pend) && mlen <= DMCAP &&
                               (srcend <= (int)X || below == 0ull || src0 >= (int)pe);
            if (ready) lds_match_lane(s_out, mdv, offv, mlen);
            pend &= ~hb_ballot(ready);
        }
        di += total;
        if (rewound) { nq = 0; break; }
        if constexpr (LEAN) {
            const uint16_t rest = ((const uint16_t *)s_tq)[64 + lane < DTQ ? 64 + lane : 0];  // keep what is queued beyond the 64 just decoded
            nq -= cntb;
            ((uint16_t *)s_tq)[lane] = rest;                    // (every lane: the slots behind nq are free)
        } else {
            const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];
            nq -= cntb;
            if ((uint32_t)lane < nq) s_tq[lane] = rest;
        }
    }
    return ok;
}

// ---- LZ4 token parsing shared by the indexed, serial and region decoders ----
// Sum of an LZ4 length extension (bytes 255 ... 255 r) starting at slice offset si, read cooperatively 64 bytes
// at a time; bytes beyond the staged window come straight from HBM (a literal run of many MiB has an
// extension of tens of KiB).  Returns false when the extension runs off the slice or is absurdly long.
__device__ __forceinline__ bool dec_read_ext(const uint8_t *in, int inoff, uint32_t wlo, uint32_t staged, const uint8_t *g, uint32_t slen,
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
                    const unsigned long long bad = hb_ballot(!ff);
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
        if (i < slen) b = (i >= wlo && i < staged) ? in[(uint32_t)((int)i + inoff)] : g[i];
        const unsigned long long stop = hb_ballot(b != 255u);
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

// FILL: parse windows of 64 stream bytes (in[k] = byte k of the slice, lim = bytes that may be looked at) until 64 tokens
// are queued, the slice ends, or a token needs the one-at-a-time path (multi-byte length extension, too close to lim).
// Returns true when it stopped for one of the latter reasons.  Tokens go to s_tq as {lsrc | lit << 13 | mlen << 22,
// offset | tokpos << 16}, all slice-relative.
__device__ __forceinline__ bool dec_fill(const uint8_t *s_in, const uint32_t sh, const uint32_t lim, const uint32_t slen,
                                         uint32_t &si, uint32_t &nq, uint2 *s_tq, const int lane) {
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
        const uint32_t x = dec_read4(s_in, sh + (cplx ? p : offpos));
        const uint32_t offv = x & 0xFFFFu, mb = (x >> 16) & 255u, mn = t & 15u;
        uint32_t mlen = 4u + mn, nbm = 0;
        if (mn == 15u) { if (mb == 255u) cplx = true; else { mlen = 19u + mb; nbm = 1; } }
        const uint32_t nxt = offpos + 2u + nbm;
        // follow the real token chain through the window: one bit-set + one readlane per token; a
        // "complex" lane ends the walk (its successor is >= 64)
        const unsigned long long cmask = hb_ballot(cplx);
        unsigned long long tmask = 0;
        uint32_t cur;
        {
            const uint32_t nrel = cplx ? 64u : nxt - base;
            const uint32_t succ = nrel < 64u ? nrel : (uint32_t)lane;   // the last token of the window points at itself
            uint32_t j = 0, lastj;
            for (;;) {                                       // unrolled by 4: setting the last bit again is harmless
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j));
                const uint32_t j1 = __builtin_amdgcn_readlane(succ, (int)j);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j1));
                const uint32_t j2 = __builtin_amdgcn_readlane(succ, (int)j1);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j2));
                const uint32_t j3 = __builtin_amdgcn_readlane(succ, (int)j2);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j3));
                j = __builtin_amdgcn_readlane(succ, (int)j3);
                lastj = j3;
                if (j == j3) break;
            }
            cur = base + __builtin_amdgcn_readlane(nrel, (int)lastj);
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
    return stop;
}

// FILL, lean: like dec_fill, but a lane only works out how LONG the sequence would be that starts at its byte (token, at most one
// extension byte per length, literals, offset: two byte reads and a handful of VALU), the chain is followed as before, and the queue gets
// the positions of the real tokens (u16); dec_drain<true> parses the fields of the 64 tokens it decodes.  An extension byte is taken to
// be the only one here; dec_drain<true> looks at it and rewinds at a token where that does not hold.
__device__ __forceinline__ bool dec_fill_lean(const uint8_t *s_in, const uint32_t sh, const uint32_t lim, const uint32_t slen,
                                              uint32_t &si, uint32_t &nq, uint2 *s_tq, const int lane, uint32_t &last_ntok) {
    bool stop = false;
    // which of the two ways to follow the chain a window takes is decided by the window before it: the doubling costs nine ds_bpermute whatever the
    // window holds, the scalar walk three instructions per token (bit-shuffled integers: 3 tokens per window -- 1.07 ms against 1.32 with the doubling
    // everywhere; the headline's dense planes: 14 per window, 1.32 against 1.28)
    // (last_ntok: the caller's, carried from call to call; it starts a unit at DEC_BPERM_MIN)
    while (nq < 64u && !stop) {
        if (si == slen) { stop = true; break; }
        const uint32_t base = si, p = base + (uint32_t)lane;
        const uint32_t t = s_in[sh + p], b1 = s_in[sh + p + 1u];
        const uint32_t lit = t >= 0xF0u ? 16u + b1 : (t >> 4);                     // literals (+ their extension byte)
        const uint32_t len = lit + ((t & 15u) == 15u ? 4u : 3u);                    // token + literals + offset (+ the match extension byte)
        const uint32_t nrel = (uint32_t)lane + len;
        // a token whose fields do not lie inside the staged bytes (with room for the dword reads of the drain) ends the walk
        const unsigned long long cmask = hb_ballot(p + len + 1u > lim);
        unsigned long long tmask = 0;
        uint32_t cur;
        const unsigned long long selfm = cmask | hb_ballot(nrel >= 64u);
        const uint32_t succ = hb_lane_in(selfm) ? (uint32_t)lane : nrel;              // the last token of the window points at itself
#if DEC_BPERM_WALK
        if (last_ntok >= DEC_BPERM_MIN) {
            // The chain WITHOUT a scalar walk (one s_bitset + s_nop + v_readlane per token: the decoder issues more scalar than vector
            // instructions, and half of them were this walk): successor tables by pointer doubling (S2 = S1 o S1, ... S16: four
            // ds_bpermute), then lane k composes the powers its bits name -- c_k = S1^k(lane 0), five more -- and holds the k-th token of
            // the window: the queue is written compacted, no rank computation either.  A window holds at most 22 tokens (3 bytes each).
            const uint32_t s1 = succ;
            const uint32_t s2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s1 << 2), (int)s1);
            const uint32_t s4 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s2 << 2), (int)s2);
            const uint32_t s8 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s4 << 2), (int)s4);
            const uint32_t s16 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(s8 << 2), (int)s8);
            uint32_t c = 0;
            { const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)s1); c = (lane & 1) ? y : c; }
            { const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)s2); c = (lane & 2) ? y : c; }
            { const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)s4); c = (lane & 4) ? y : c; }
            { const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)s8); c = (lane & 8) ? y : c; }
            { const uint32_t y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)s16); c = (lane & 16) ? y : c; }
            // c_k repeats once the chain has reached its last lane: the tokens are the lanes whose c differs from their left neighbour's
            const uint32_t left = wave_shr1(c, 0xFFFFFFFFu);
            const unsigned long long distinct = hb_ballot(c != left) | 0xFFFFFFFF00000000ull;   // (lanes 32.. take no part)
            uint32_t ntok = (uint32_t)__builtin_ctzll(~distinct | (1ull << 32));                 // leading run of distinct lanes (lane 0 always is)
            const uint32_t lastj = __builtin_amdgcn_readlane(c, (int)ntok - 1);
            cur = base + __builtin_amdgcn_readlane(nrel, (int)lastj);
            if ((cmask >> lastj) & 1ull) { ntok--; cur = base + lastj; stop = true; }           // a token this parser does not take ends the walk
            ((uint16_t *)s_tq)[(uint32_t)lane < ntok ? nq + (uint32_t)lane : (uint32_t)(DTQ - 1)] = (uint16_t)(base + c);
            nq += ntok;
            si = cur;
            last_ntok = ntok;
            continue;
        }
#endif
        {
            uint32_t j = 0, lastj;
            for (;;) {                                       // unrolled by 4: setting the last bit again is harmless
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j));
                const uint32_t j1 = __builtin_amdgcn_readlane(succ, (int)j);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j1));
                const uint32_t j2 = __builtin_amdgcn_readlane(succ, (int)j1);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j2));
                const uint32_t j3 = __builtin_amdgcn_readlane(succ, (int)j2);
                asm volatile("s_bitset1_b64 %0, %1" : "+s"(tmask) : "s"(j3));
                j = __builtin_amdgcn_readlane(succ, (int)j3);
                lastj = j3;
                if (j == j3) break;
            }
            cur = base + __builtin_amdgcn_readlane(nrel, (int)lastj);
            const unsigned long long cm = tmask & cmask;  // at most the last visited lane
            if (cm) { tmask &= ~cm; cur = base + (uint32_t)__builtin_ctzll(cm); stop = true; }
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tmask, 0u));
        ((uint16_t *)s_tq)[hb_select_lane(tmask, nq + rank, (uint32_t)(DTQ - 1))] = (uint16_t)p;
        last_ntok = (uint32_t)__builtin_popcountll(tmask);
        nq += last_ntok;
        si = cur;
    }
    return stop;
}

// ---- the serial block decoder (hb_lz4_dec.hip k_dec_serial; hb_cblosc.hip: one wavefront per stream of a C-Blosc-1 frame) ----
#define SER_WIN   8192u
#define SER_HIST  65536u
#define SER_PAGE  32768u      // room for output between flushes
#define SER_SOFT  16384u      // flush when the page holds this much

// One LZ4 block, front to back, one wavefront (the body of k_dec_serial; also what decodes the streams of a C-Blosc-1 frame,
// hb_cblosc.hip).  s_win: SER_WIN + 128 bytes, s_img: SER_HIST + SER_PAGE + 1024 bytes, s_tq: DTQ entries.  Returns the bytes
// produced; err != 0: malformed (what lz4.UncompressBlock rejects) or more than cap bytes.
__device__ __forceinline__ uint64_t dec_serial_core(const uint8_t *__restrict__ src, const uint64_t n_src, uint8_t *__restrict__ dst, const uint64_t cap,
                                                    uint8_t *s_win, uint8_t *s_img, uint2 *s_tq, const int lane, int &err) {
    bool fin = false;
    err = 0;
    uint8_t *out = s_img + SER_HIST;         // out[i], i in [-SER_HIST, SER_PAGE): output byte gbase + i
    uint64_t gbase = 0;                      // output bytes already flushed to dst
    uint32_t di = 0;                         // bytes in the page
    uint64_t wpos = 0;                       // stream position of the window
    uint32_t wlen = 0, wsh = 0;              // s_win[wsh + k] = src[wpos + k], k < wlen
    uint64_t si = 0;                         // stream position

    auto refill = [&](uint64_t at) __attribute__((always_inline)) {
        const uint8_t *g = src + at;
        wsh = (uint32_t)((uintptr_t)g & 15u);
        const uint64_t left = n_src - at;
        wlen = (uint32_t)(left < (uint64_t)(SER_WIN - 16u) ? left : (uint64_t)(SER_WIN - 16u));
        const u32x4 *ga = (const u32x4 *)(g - wsh);
        const uint32_t nv = (wsh + wlen + 15u) >> 4;
        wave_sync();
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
        wpos = at;
        wave_sync();
    };
    // write the first fl (multiple of 16) page bytes to dst and slide the image down by fl
    auto flush = [&](bool all) __attribute__((always_inline)) {
        const uint32_t fl = all ? di : (di & ~15u);
        if (fl == 0) return;
        wave_sync();
        uint8_t *o = dst + gbase;
        for (uint32_t i = lane * 16u; i + 16u <= fl; i += 1024u) st16u(o + i, *(const u32x4 *)(out + i));
        const uint32_t tail0 = fl & ~15u;
        if (tail0 + lane < fl) o[tail0 + lane] = out[tail0 + lane];
        if (!all) {
            const uint32_t mv = SER_HIST + (di - fl);                 // bytes that stay: history + the unflushed tail
            for (uint32_t k = lane * 16u; k < mv; k += 1024u) {       // ascending, 1 KiB per step: reads run ahead of writes
                const u32x4 v = *(const u32x4 *)(s_img + fl + k);
                *(u32x4 *)(s_img + k) = v;
            }
            wave_sync();
        }
        gbase += fl; di -= fl;
    };

    uint64_t lrem = 0, mrem = 0;             // literals / match bytes of the current sequence still to copy
    uint32_t moff = 0, tok = 0;
    int phase = 0;                           // 0 token, 1 literals, 2 offset + match length, 3 match
    uint32_t nq = 0;
    if (n_src) refill(0); else fin = true;
    while (!err && !fin) {
        if (gbase + di > cap) { err = 1; break; }              // before anything reaches dst
        if (di >= SER_SOFT) flush(false);
        const uint32_t room = SER_PAGE - di;
        if (phase == 0) {
            if (si == n_src) { fin = true; break; }                   // ran out of input at a token boundary: done
            // keep a comfortable look-ahead in the window
            if (si < wpos || si - wpos + 1024u > wlen) { if (si != wpos || wlen == 0) refill(si); }
            uint32_t rel = (uint32_t)(si - wpos);
            // ---- fast path: window-parallel parse, lane-parallel copies ----
            const uint32_t hist = (uint32_t)(gbase < (uint64_t)SER_HIST ? gbase : (uint64_t)SER_HIST);
            const bool stop = dec_fill(s_win, wsh, wlen, wlen, rel, nq, s_tq, lane);
            bool rewound = false;
            if (!dec_drain(s_win + wsh, 0, out, SER_PAGE, hist, di, rel, nq, s_tq, true, rewound, lane)) { err = 1; break; }
            const bool moved = (wpos + rel) != si;
            si = wpos + rel;
            if (rewound) continue;                                    // page full: flush, then go on from that token
            if (moved && !stop) continue;
            if (si == n_src) { fin = true; break; }
            if (moved && si - wpos + 1024u > wlen && wpos + wlen < n_src) continue;    // stopped at the window edge: refill first
            // ---- one token the slow way (length extensions of any size, literal runs of any size) ----
            if (si < wpos || si >= wpos + wlen) refill(si);
            rel = (uint32_t)(si - wpos);
            tok = __builtin_amdgcn_readfirstlane((uint32_t)s_win[wsh + rel]);   // uniform, but from a vector load
            rel++;
            uint32_t ll = tok >> 4;
            if (ll == 15u) {
                const uint64_t span = n_src - wpos;
                if (!dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel, ll, lane)) { err = 1; break; }
            }
            si = wpos + rel;
            lrem = ll;
            phase = 1;
            continue;
        }
        if (phase == 1) {
            if (lrem) {
                const uint32_t take = (uint32_t)(lrem < (uint64_t)room ? lrem : (uint64_t)room);
                if (take == 0) { flush(false); continue; }
                if ((uint64_t)take > n_src - si || gbase + di + take > cap) { err = 1; break; }
                const uint8_t *g = src + si;
                uint32_t k0 = 0;
                for (; k0 + 4096u <= take; k0 += 4096u) {                // 4 x 16 B per lane in flight
                    u32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = ld16u(g + k0 + (uint32_t)q * 1024u + (uint32_t)lane * 16u);
#pragma unroll
                    for (int q = 0; q < 4; q++) ((hb_u128u *)(out + di + k0 + (uint32_t)q * 1024u + (uint32_t)lane * 16u))->v = v[q];
                }
                for (uint32_t k = k0 + lane; k < take; k += 64) out[di + k] = g[k];
                si += take; di += take; lrem -= take;
                continue;
            }
            if (si == n_src) {
                if ((tok & 15u) != 0u) err = 1;                       // input ends after literals but a match was announced
                fin = true;
                break;
            }
            phase = 2;
            continue;
        }
        if (phase == 2) {
            if (n_src - si < 2) { err = 1; break; }
            moff = __builtin_amdgcn_readfirstlane((uint32_t)src[si] | ((uint32_t)src[si + 1] << 8));
            si += 2;
            if (moff == 0) { err = 1; break; }
            uint32_t ml = (tok & 15u) + 4u;
            if ((tok & 15u) == 15u) {
                if (si < wpos || si - wpos + 64u > wlen) refill(si);
                uint32_t rel = (uint32_t)(si - wpos);
                const uint64_t span = n_src - wpos;
                if (!dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel, ml, lane)) { err = 1; break; }
                si = wpos + rel;
            }
            if ((uint64_t)moff > gbase + di) { err = 1; break; }      // before the start of the block (no dictionary)
            mrem = ml;
            phase = 3;
            continue;
        }
        {   // phase 3: match, in pieces that fit the page
            const uint32_t take = (uint32_t)(mrem < (uint64_t)room ? mrem : (uint64_t)room);
            if (take == 0) { flush(false); continue; }
            if (gbase + di + take > cap) { err = 1; break; }
            wave_sync();
            dec_match_copy(out, di, moff, take, lane);
            di += take; mrem -= take;
            if (mrem == 0) phase = 0;
        }
    }
    if (!err && gbase + di > cap) err = 1;
    if (!err) flush(true);
    return gbase + di;
}

