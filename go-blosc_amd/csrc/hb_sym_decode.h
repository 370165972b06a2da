// hb_sym_decode.h — one wavefront decodes a run of LZ4 sequences whose first token, end and output position are known, straight into HBM
// through a small LDS image of its recent output (pass A of the symbolic decoder, hb_lz4_sym.hip; with SYM = false the same machine
// decodes the streams of a C-Blosc-1 frame, hb_cblosc.hip: no references, a source in front of the unit's first byte is an error).
#pragma once
#include "hb_lz4_region.h"

#ifndef SY_IMG
#define SY_IMG      4096u         // bytes of output a wave holds in LDS: the last SY_HIST bytes it wrote to HBM + what it is building
#define SY_HIST     2048u
#endif
#define SY_NCAP     64u           // image-to-image copies up to this long are done by their own lane
#ifndef SY_BIG
#define SY_BIG      (256u << 10)  // literal runs / matches from this size on are copied by the whole chip (k_sy_big)
#endif

struct SyPlan { uint32_t go, fail, groups, per, nbig, nunits, nact, live, merge, usetok; uint32_t pad[6]; };   // per: units WITH OUTPUT per group of pass B; live: regions with output; merge: light neighbours share a unit
// what one wavefront of pass A decodes: a region of the token discovery, or one of SY_SUB parts of a region whose output is large
// state: bit 0 done, bit 1: the literal run of the token at rtp is being / has been copied by k_sy_big -- the rest of that sequence is on record
// (plit / pmlen / poff, pnext = the token behind it): a resumed unit never parses a parked token again (the length extension of a 256 MiB run is a
// MiB of FF bytes).  A parked MATCH leaves no token behind: rtp = the next token, rout = the output position behind the match, state 0.
struct SyUnit { uint32_t entry, exit, opos, outlen, rtp, rout, state, plit, pmlen, poff, pnext, pad; };
struct SyBig { uint32_t kind, dst, src, len, O, pad[3]; };          // kind 0: literals from stream position src; 1: match, src = offset

#ifdef SY_DEBUG_TIMES
// phase clocks of pass A (build variant only: scratch/mkvariant.sh sytimes hb_lz4_sym.hip -DSY_DEBUG_TIMES; read by hb_debug_sy_times)
extern __device__ unsigned long long sy_dbg[48];
#define SYT_NOW() __builtin_readcyclecounter()
#define SYT_ADD(slot, t0) do { dbg_t[slot] += SYT_NOW() - (t0); } while (0)
#define SYT_CNT(slot) do { dbg_t[slot] += 1ull; } while (0)
#else
#define SYT_NOW() 0ull
#define SYT_ADD(slot, t0) do { (void)(t0); } while (0)
#define SYT_CNT(slot) do { } while (0)
#endif

// every store of this wave so far has reached the cache all lanes of the CU read through, and later loads are not started early
__device__ __forceinline__ void sy_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ---- copies of pass A.  D = output values, S = references (u16 per output byte; Sb = the same as bytes), O = first output byte of
// the region.  A load through HBM/L2 costs a microsecond, so every copy issues ALL its loads before its first store (the compiler
// cannot do that: for all it knows they alias), and pieces are placed so that no store is partial: the last piece of a copy is
// moved back to end exactly at the end (it rewrites a few bytes with the same values). ----

// ---- the output image.  A wave keeps the last SY_IMG bytes of its output in LDS (values s_d, references s_s): what it is building,
// behind SY_HIST bytes of what it has written to HBM already.  Matches that copy what the image holds -- the short offsets, and the
// chains of matches that each copy the one before -- run LDS to LDS in cheap dependency rounds; only sources in front of the image
// ("far") are fetched from HBM, all lanes at once, before the rounds start (nothing in the batch can change them). ----

// What one lane brings into the image from HBM: the `far` first bytes of its match, whose source out[sp...) lies in front of the image,
// to image[tf...).  Lengths come in three classes (8-byte pieces, two 4-byte pieces, single bytes) and the lanes of a wave are in
// all of them at once: every class issues its loads before any class stores, so the wave pays ONE round trip, not three.  More
// than 32 bytes: 32 per further round trip.
template <bool SYM>
__device__ __forceinline__ void sy_fetch_lane(const uint8_t *D, const uint16_t *S, uint8_t *s_d, uint16_t *s_s, uint32_t tf, uint32_t sp, uint32_t far, const uint32_t O) {
    const uint8_t *Sb = (const uint8_t *)S;
    uint8_t *s_sb = (uint8_t *)s_s;
    {   // bytes from in front of the region are references: nothing to load
        const uint32_t nctx = sp < O ? (O - sp < far ? O - sp : far) : 0u;
        if (SYM) for (uint32_t k = 0; k < nctx; k++) s_s[tf + k] = (uint16_t)(O - sp - k);
        tf += nctx; sp += nctx; far -= nctx;
    }
    const bool f8 = far >= 8u, f4 = far >= 4u && far < 8u, f1 = far != 0u && far < 4u;
    uint64_t fv[4]; u32x4 fa[4]; uint32_t fk[4];
    uint32_t fw0 = 0, fw1 = 0; uint64_t fb0 = 0, fb1 = 0;
    uint8_t fx[3]; uint16_t fy[3];
    // ---- loads ----
    if (f8) {
#pragma unroll
        for (int c = 0; c < 4; c++) { fk[c] = 8u * c; if (fk[c] < far) { if (fk[c] + 8u > far) fk[c] = far - 8u; fv[c] = ld8u(D + sp + fk[c]); if (SYM) fa[c] = ld16u(Sb + 2u * (size_t)(sp + fk[c])); } }
    }
    if (f4) { fw0 = ld4u(D + sp); fw1 = ld4u(D + sp + far - 4u); if (SYM) { fb0 = ld8u(Sb + 2u * (size_t)sp); fb1 = ld8u(Sb + 2u * (size_t)(sp + far - 4u)); } }
    if (f1) {
#pragma unroll
        for (int k = 0; k < 3; k++) if ((uint32_t)k < far) { fx[k] = D[sp + k]; if (SYM) fy[k] = S[sp + k]; }
    }
    // ---- stores ----
    if (f8) {
#pragma unroll
        for (int c = 0; c < 4; c++) if (8u * c < far) { ((hb_u64u *)(s_d + tf + fk[c]))->v = fv[c]; if (SYM) ((hb_u128u *)(s_sb + 2u * (tf + fk[c])))->v = fa[c]; }
    }
    if (f4) {
        ((hb_u32u *)(s_d + tf))->v = fw0; ((hb_u32u *)(s_d + tf + far - 4u))->v = fw1;
        if (SYM) { ((hb_u64u *)(s_sb + 2u * tf))->v = fb0; ((hb_u64u *)(s_sb + 2u * (tf + far - 4u)))->v = fb1; }
    }
    if (f1) {
#pragma unroll
        for (int k = 0; k < 3; k++) if ((uint32_t)k < far) { s_d[tf + k] = fx[k]; if (SYM) s_s[tf + k] = fy[k]; }
    }
    // ---- beyond 32 bytes ----
    for (uint32_t blk = 32u; blk < far; blk += 32u) {
        uint64_t v[4]; u32x4 a[4]; uint32_t k[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            k[c] = blk + 8u * c;
            if (k[c] < far) { if (k[c] + 8u > far) k[c] = far - 8u; v[c] = ld8u(D + sp + k[c]); if (SYM) a[c] = ld16u(Sb + 2u * (size_t)(sp + k[c])); }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) if (blk + 8u * c < far) { ((hb_u64u *)(s_d + tf + k[c]))->v = v[c]; if (SYM) ((hb_u128u *)(s_sb + 2u * (tf + k[c])))->v = a[c]; }
    }
}
// the same by the whole wave (len > 64)
template <bool SYM>
__device__ __forceinline__ void sy_far_wave(const uint8_t *D, const uint16_t *S, uint8_t *s_d, uint16_t *s_s, uint32_t t, uint32_t sp, uint32_t len, const uint32_t O, const int lane) {
    const uint32_t nctx = sp < O ? (O - sp < len ? O - sp : len) : 0u;
    if (SYM) for (uint32_t i = lane; i < nctx; i += 64) s_s[t + i] = (uint16_t)(O - sp - i);
    t += nctx; sp += nctx; len -= nctx;
    const uint8_t *Sb = (const uint8_t *)S;
    uint8_t *s_sb = (uint8_t *)s_s;
    if (len < 16u) { if ((uint32_t)lane < len) { s_d[t + lane] = D[sp + lane]; if (SYM) s_s[t + lane] = S[sp + lane]; } return; }
    for (uint32_t blk = 0; blk < len; blk += 4096u) {
        u32x4 v[4], a[4], b[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < len) {
                if (j[q] + 16u > len) j[q] = len - 16u;
                v[q] = ld16u(D + sp + j[q]); if (SYM) { a[q] = ld16u(Sb + 2u * (size_t)(sp + j[q])); b[q] = ld16u(Sb + 2u * (size_t)(sp + j[q]) + 16u); }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < len) {
                ((hb_u128u *)(s_d + t + j[q]))->v = v[q]; if (SYM) { ((hb_u128u *)(s_sb + 2u * (t + j[q])))->v = a[q]; ((hb_u128u *)(s_sb + 2u * (t + j[q]) + 16u))->v = b[q]; }
            }
    }
}
// literals: staged stream window (LDS) -> image; values, never references
template <bool SYM>
__device__ __forceinline__ void sy_lits_img_lane(uint8_t *s_d, uint16_t *s_s, const uint32_t t, const uint8_t *lp, const uint32_t lit) {
    lds_copy_exact(s_d + t, lp, lit);
    if (!SYM) return;
    uint8_t *z = (uint8_t *)(s_s + t);
    uint32_t k = 0;
    for (; k + 8u <= 2u * lit; k += 8u) ((hb_u64u *)(z + k))->v = 0ull;
    if ((2u * lit) & 4u) { ((hb_u32u *)(z + k))->v = 0u; k += 4u; }
    if ((2u * lit) & 2u) ((hb_u16u *)(z + k))->v = 0;
}
template <bool SYM>
__device__ __forceinline__ void sy_lits_img_wave(uint8_t *s_d, uint16_t *s_s, const uint32_t t, const uint8_t *lp, const uint32_t lit, const int lane) {
    for (uint32_t k = lane; k < lit; k += 64) { s_d[t + k] = lp[k]; if (SYM) s_s[t + k] = 0; }
}
// near copies: tile[md + k] = tile[md - off + k], values and references alike.  One lane (what lds_match_lane does for bytes):
template <bool SYM>
__device__ __forceinline__ void sy_near_lane(uint8_t *s_d, uint16_t *s_s, const uint32_t md, const uint32_t off, const uint32_t len) {
    lds_match_lane(s_d, md, off, len);
    if (!SYM) return;
    uint16_t *d = s_s + md;
    const uint16_t *s = d - off;
    uint32_t k = 0;
    if (off >= 4u) {
        for (; k + 4u <= len; k += 4u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(s + k))->v;
        for (; k < len; k++) d[k] = s[k];
        return;
    }
    // periods 1..3: the first eo entries one by one (eo = 4, 4, 6: the smallest multiple of the period that is >= 4), then four at a time
    // from eo entries back -- a run of 64 zero bytes that are references is 4 + 15 steps, not 64
    const uint32_t eo = off == 3u ? 6u : 4u;
    const uint32_t head = len < eo ? len : eo;
    for (; k < head; k++) d[k] = s[k];
    for (; k + 4u <= len; k += 4u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(d + k - eo))->v;
    for (; k < len; k++) d[k] = d[k - eo];
}
// ... and the whole wave
template <bool SYM>
__device__ __forceinline__ void sy_near_wave(uint8_t *s_d, uint16_t *s_s, const uint32_t md, const uint32_t off, const uint32_t len, const int lane) {
    dec_match_copy(s_d, md, off, len, lane);
    if (!SYM) return;
    if (off >= 64u || off >= len) {
        for (uint32_t i = lane; i < len; i += 64) s_s[md + i] = s_s[md + i - off];
    } else {
        uint32_t m = (uint32_t)lane % off;
        const uint32_t step = 64u % off;
        for (uint32_t i = lane; i < len; i += 64) {
            s_s[md + i] = s_s[md - off + m];
            m += step; if (m >= off) m -= off;
        }
    }
}
template <bool SYM>
__device__ __forceinline__ void sy_lits_wave(uint8_t *D, uint16_t *S, const uint32_t d0, const uint8_t *g, const uint32_t lit, const int lane) {
    uint8_t *Sb = (uint8_t *)S;
    u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
    if (lit < 16u) { if ((uint32_t)lane < lit) { D[d0 + lane] = g[lane]; if (SYM) S[d0 + lane] = 0; } return; }
    for (uint32_t blk = 0; blk < lit; blk += 4096u) {
        u32x4 v[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < lit) { if (j[q] + 16u > lit) j[q] = lit - 16u; v[q] = ld16u(g + j[q]); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < lit) { st16u(D + d0 + j[q], v[q]); if (SYM) { st16u(Sb + 2u * (size_t)(d0 + j[q]), z); st16u(Sb + 2u * (size_t)(d0 + j[q]) + 16u, z); } }
    }
}
// out[md + i] = out[s0 + i], i < len, the whole wave, source and destination do not overlap (len <= md - s0)
template <bool SYM>
__device__ __forceinline__ void sy_copy_wave(uint8_t *D, uint16_t *S, const uint32_t md, const uint32_t s0, const uint32_t len, const uint32_t O, const int lane) {
    const uint32_t nctx = s0 < O ? (O - s0 < len ? O - s0 : len) : 0u;          // leading bytes that come from in front of the region
    if (SYM) for (uint32_t i = lane; i < nctx; i += 64) S[md + i] = (uint16_t)(O - s0 - i);
    uint8_t *Sb = (uint8_t *)S;
    const uint32_t n = len - nctx, d = md + nctx, s = s0 + nctx;
    if (n < 16u) {
        if ((uint32_t)lane < n) { const uint8_t v = D[s + lane]; D[d + lane] = v; if (SYM) { const uint16_t a = S[s + lane]; S[d + lane] = a; } }
        return;
    }
    for (uint32_t blk = 0; blk < n; blk += 4096u) {
        u32x4 v[4], a[4], b[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < n) {
                if (j[q] + 16u > n) j[q] = n - 16u;
                v[q] = ld16u(D + s + j[q]); if (SYM) { a[q] = ld16u(Sb + 2u * (size_t)(s + j[q])); b[q] = ld16u(Sb + 2u * (size_t)(s + j[q]) + 16u); }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < n) {
                st16u(D + d + j[q], v[q]); if (SYM) { st16u(Sb + 2u * (size_t)(d + j[q]), a[q]); st16u(Sb + 2u * (size_t)(d + j[q]) + 16u, b[q]); }
            }
    }
}
// a whole match by the whole wave.  Overlapping (off < mlen: the `off` bytes in front of it, repeated): the first 128..256 bytes are
// gathered byte by byte from that period, then pieces that double -- what is copied already is source for the next piece -- so a run
// of any length and period costs 1 + log2(length / 256) round trips.
template <bool SYM>
__device__ __forceinline__ void sy_match_wave(uint8_t *D, uint16_t *S, const uint32_t md, const uint32_t off, const uint32_t mlen, const uint32_t O, const int lane) {
    const uint32_t s0 = md - off;
    uint32_t done = 0;
    if (off < mlen && off < 256u) {
        const uint32_t reps = 256u / off;
        const uint32_t P = off * reps < mlen ? off * reps : mlen;      // (a multiple of the period unless it is the whole match)
        uint8_t v[4]; uint16_t a[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t k = (uint32_t)lane + 64u * q;
            if (k < P) { const uint32_t sp = s0 + k % off; if (sp >= O) { v[q] = D[sp]; if (SYM) a[q] = S[sp]; } else { v[q] = 0; a[q] = (uint16_t)(O - sp); } }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t k = (uint32_t)lane + 64u * q; if (k < P) { D[md + k] = v[q]; if (SYM) S[md + k] = a[q]; } }
        done = P;
        if (done < mlen) sy_sync();
    }
    while (done < mlen) {
        const uint32_t room = off + done, left = mlen - done;
        const uint32_t piece = left < room ? left : room;               // `done` stays a multiple of the period until the last piece
        sy_copy_wave<SYM>(D, S, md + done, s0, piece, O, lane);
        done += piece;
        if (done < mlen) sy_sync();
    }
}


// The unit: tokens [start, exitp) of the stream `src` (n_src bytes), output from position `out` on; `base` = the first position a match
// may copy from (the start of the block); O = the unit's first output byte (sources in front of it become references when SYM).
// R / sy / big: where a unit of the symbolic decoder parks in front of a big copy (NULL: never parks).  Returns false when the
// stream is not what the chain promised (offset 0, offset in front of `base`, a final sequence that announces a match, output past
// `limit`: nothing is written there).
// TOK: the stretch is walked region by region of the token discovery and fed from its token store `ts` where that is usable (see below).
template <bool SYM, uint32_t PWIN = RG_PWIN, bool TOK = false, int CODEC = RG_LZ4>
__device__ __forceinline__ bool sy_decode_unit(const uint8_t *__restrict__ src, const uint64_t n_src, const uint32_t start, const uint32_t exitp, const uint32_t base,
                                               const uint32_t O, uint32_t &out, uint8_t *D, uint16_t *S, uint8_t *s_win, uint2 *s_tq, uint8_t *s_d, uint16_t *s_s,
                                               const int lane, const int last, const uint32_t rtp, const uint32_t st, SyUnit *R, SyPlan *sy, SyBig *big, bool &parked,
                                               const uint32_t limit, const RgTokStore *ts = nullptr) {
        uint8_t *Sb = (uint8_t *)S;
        parked = false;
        const bool resumed = SYM && R != nullptr && (st & 2u) != 0u;             // behind a literal run k_sy_big copied (see SyUnit)
#ifdef SY_DEBUG_TIMES
        unsigned long long dbg_t[16] = {0};
        const unsigned long long dbg_t0 = SYT_NOW();
#endif
        // the image: s_d[i] / s_s[i] = output byte ib + i for i < out - ib; bytes below fl are in HBM already (history kept for near copies)
        uint32_t ib = out, fl = out;
        bool bad = false, unsynced = false;
        // the new part of the image goes to HBM
        auto flush = [&]() __attribute__((always_inline)) {
            const uint32_t n = out - fl;
            if (n == 0u) return;
            const unsigned long long tq0 = SYT_NOW();
            wave_sync();
            const uint32_t o = fl - ib;
            const uint8_t *s_sb = (const uint8_t *)s_s;
            if (n < 16u) {
                if ((uint32_t)lane < n) { D[fl + lane] = s_d[o + lane]; if (SYM) S[fl + lane] = s_s[o + lane]; }
            } else {
                for (uint32_t i = (uint32_t)lane * 16u; i < n; i += 1024u) {
                    const uint32_t j = i + 16u <= n ? i : n - 16u;
                    st16u(D + fl + j, ((const hb_u128u *)(s_d + o + j))->v);
                    if (SYM) {
                        st16u(Sb + 2u * (size_t)(fl + j), ((const hb_u128u *)(s_sb + 2u * (o + j)))->v);
                        st16u(Sb + 2u * (size_t)(fl + j) + 16u, ((const hb_u128u *)(s_sb + 2u * (o + j) + 16u))->v);
                    }
                }
            }
            fl = out;
            unsynced = true;                                             // far loads must wait for these stores (sy_sync), not the flush itself
            SYT_ADD(5, tq0);
        };
        // make room: flush, keep the last SY_HIST bytes as history at the bottom of the image
        auto slide = [&]() __attribute__((always_inline)) {
            SYT_CNT(11);
            flush();
            const unsigned long long tq1 = SYT_NOW();
            const uint32_t have = out - ib, keep = have < SY_HIST ? have : SY_HIST, delta = have - keep;
            if (delta) {
                uint8_t *s_sb = (uint8_t *)s_s;
                wave_sync();
                for (uint32_t k = (uint32_t)lane * 16u; k < keep; k += 1024u) {          // ascending: a step's reads are done before its writes
                    const u32x4 v = ((const hb_u128u *)(s_d + delta + k))->v;
                    ((hb_u128u *)(s_d + k))->v = v;
                }
                if (SYM) for (uint32_t k = (uint32_t)lane * 16u; k < 2u * keep; k += 1024u) {
                    const u32x4 v = ((const hb_u128u *)(s_sb + 2u * delta + k))->v;
                    ((hb_u128u *)(s_sb + k))->v = v;
                }
                wave_sync();
            }
            ib = out - keep;
            SYT_ADD(6, tq1);
        };
        auto batch = [&](uint32_t cnt, uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t lp) __attribute__((always_inline)) -> bool {
            (void)tp; (void)ls;
            SYT_CNT(8);
            const unsigned long long tb0 = SYT_NOW();
            const bool tok = (uint32_t)lane < cnt;
            const uint32_t olen = tok ? lit + mlen : 0u;
            const uint32_t incl = wave_incl_scan_dpp(olen);
            const uint32_t d0 = out + incl - olen, md = d0 + lit;
            if (hb_ballot(tok && (off == 0u || off > md - base))) { bad = true; return false; }     // offset 0 / before the start of the block
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
            if (total > limit - out) { bad = true; return false; }                              // (more than the unit may produce)
            // what a lane fetches from HBM itself (32 bytes per round trip): up to four times the batch's average, 64 at least
            const uint32_t thr = total < 64u * 16u ? 64u : (total > 1024u * 16u ? 1024u : total >> 4);
            const uint32_t s0 = md - off;
            const uint32_t end_all = out + total;
            uint32_t lo = 0;
            while (lo < cnt) {
                const unsigned long long over = hb_ballot(tok && (uint32_t)lane >= lo && d0 + olen - ib > SY_IMG);
                const uint32_t hi = over ? (uint32_t)__builtin_ctzll(over) : cnt;
                if (hi == lo) {
                    // the next sequence does not fit into what is left of the image
                    const uint32_t ol = __builtin_amdgcn_readlane(olen, (int)lo);
                    if (ol > SY_IMG - SY_HIST) {                          // nor behind the history alone: straight in HBM, the whole wave
                        const unsigned long long th0 = SYT_NOW();
                        SYT_CNT(12);
                        flush();
                        sy_sync(); unsynced = false;
                        const uint32_t dl = __builtin_amdgcn_readlane(d0, (int)lo), ll = __builtin_amdgcn_readlane(lit, (int)lo);
                        if (ll) { sy_lits_wave<SYM>(D, S, dl, src + __builtin_amdgcn_readlane(ls, (int)lo), ll, lane); sy_sync(); }
                        sy_match_wave<SYM>(D, S, dl + ll, __builtin_amdgcn_readlane(off, (int)lo), __builtin_amdgcn_readlane(mlen, (int)lo), O, lane);
                        sy_sync();
                        out = dl + ol; ib = out; fl = out;
                        lo++;
                        SYT_ADD(7, th0);
                    } else slide();
                    continue;
                }
                SYT_CNT(9);
                const unsigned long long tl0 = SYT_NOW();
                const bool act = tok && (uint32_t)lane >= lo && (uint32_t)lane < hi;
                const uint32_t t0 = d0 - ib, tm = md - ib;
                // literals: from the staged stream window
                if (act && lit <= 32u) sy_lits_img_lane<SYM>(s_d, s_s, t0, s_win + lp, lit);
                unsigned long long lm = hb_ballot(act && lit > 32u);
                while (lm) {
                    const int l = __builtin_ctzll(lm);
                    sy_lits_img_wave<SYM>(s_d, s_s, __builtin_amdgcn_readlane(t0, l), s_win + __builtin_amdgcn_readlane(lp, l), __builtin_amdgcn_readlane(lit, l), lane);
                    lm &= lm - 1;
                }
                SYT_ADD(2, tl0);
                // the part of every match whose source lies in front of the image: from HBM, all lanes at once (nothing in the batch can
                // change those bytes)
                const uint32_t farlen = (act && s0 < ib) ? (ib - s0 < mlen ? ib - s0 : mlen) : 0u;
                if (hb_ballot(farlen != 0u)) {
                    SYT_CNT(10);
                    const unsigned long long ts0 = SYT_NOW();
                    if (unsynced) { sy_sync(); unsynced = false; SYT_CNT(13); }
                    SYT_ADD(14, ts0);
                    const unsigned long long tf0 = SYT_NOW();
                    sy_fetch_lane<SYM>(D, S, s_d, s_s, tm, s0, farlen <= thr ? farlen : 0u, O);
                    lm = hb_ballot(farlen > thr);
                    while (lm) {
                        const int l = __builtin_ctzll(lm);
                        sy_far_wave<SYM>(D, S, s_d, s_s, __builtin_amdgcn_readlane(tm, l), __builtin_amdgcn_readlane(s0, l), __builtin_amdgcn_readlane(farlen, l), O, lane);
                        lm &= lm - 1;
                        SYT_CNT(15);
                    }
                    wave_sync();
                    SYT_ADD(3, tf0);
                }
                const unsigned long long tn0 = SYT_NOW();
                // the rest of every match copies what the image holds: dependency rounds in LDS (the rule of dec_drain, hb_dec_common.h: a
                // match is ready when its source ends before the first pending match, or starts at / after the end of the nearest
                // pending match in front of it)
                const uint32_t nlen = act ? mlen - farlen : 0u, nmd = tm + farlen;
                const uint32_t srcs = nmd - off, srcend = srcs + (nlen < off ? nlen : off), mend = nmd + nlen;
                unsigned long long pend = hb_ballot(nlen != 0u);
                while (pend) {
                    const int f = __builtin_ctzll(pend);
                    const uint32_t X = __builtin_amdgcn_readlane(nmd, f);
                    const uint32_t nlf = __builtin_amdgcn_readlane(nlen, f);
                    if (nlf > SY_NCAP) {
                        sy_near_wave<SYM>(s_d, s_s, X, __builtin_amdgcn_readlane(off, f), nlf, lane);
                        pend &= pend - 1;
                        continue;
                    }
                    const unsigned long long below = pend & ((1ull << lane) - 1ull);
                    const uint32_t pj = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe = (uint32_t)__shfl((int)mend, (int)pj);
                    const bool ready = ((pend >> lane) & 1ull) && nlen <= SY_NCAP && (srcend <= X || below == 0ull || srcs >= pe);
                    if (ready) sy_near_lane<SYM>(s_d, s_s, nmd, off, nlen);
                    pend &= ~hb_ballot(ready);
                }
                out = hi == cnt ? end_all : __builtin_amdgcn_readlane(d0, (int)hi);
                lo = hi;
                SYT_ADD(4, tn0);
            }
            SYT_ADD(1, tb0);
            return true;
        };
        // kind 0: the literal run [ls, ls + lit) of the sequence at tp to out; the rest of the sequence (mlen at distance off, then the token at
        // pnext) goes on record.  kind 1: the match of mlen bytes at distance off to md; the unit resumes at the token pnext with output md + mlen.
        auto park = [&](uint32_t tp, uint32_t kind, uint32_t dst, uint32_t a, uint32_t len, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t pnext) __attribute__((always_inline)) {
#ifdef SY_DEBUG_TIMES
            if (lane == 0) { atomicAdd(&sy_dbg[32 + kind], 1ull); atomicAdd(&sy_dbg[34 + kind], (unsigned long long)(len >> 10)); }
#endif
            if (lane == 0) {
                const uint32_t slot = atomicAdd(&sy->nbig, 1u);             // (at most one per region and launch: nreg slots)
                SyBig b; b.kind = kind; b.dst = dst; b.src = a; b.len = len; b.O = O; b.pad[0] = b.pad[1] = b.pad[2] = 0;
                big[slot] = b;
                if (kind == 0u) { R->rtp = tp; R->rout = out; R->state = 2u; R->plit = lit; R->pmlen = mlen; R->poff = off; R->pnext = pnext; }
                else { R->rtp = pnext; R->rout = dst + len; R->state = 0u; R->plit = 0u; }
            }
            parked = true;
        };
        // sequences the window parser leaves alone (literal runs that leave the staged window, lengths of KiB and more): the literals go
        // from the stream straight to HBM -- nothing has to be waited for: they read the stream and write bytes nobody has written -- and
        // a match that fits into the image is then decoded through it like any other, with the end of those literals as its history
        // (read from the stream again, all values).  Incompressible data is made of such sequences (a literal run of KiB, a 4-byte match):
        // with a fence before and after each of them pass A took 6 ms per GiB of it.
        auto single = [&](uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t tok) __attribute__((always_inline)) -> bool {
            if (mlen == 0u && (tok & 15u) != 0u) { bad = true; return false; }        // the input ends after literals but a match was announced
            if ((uint64_t)lit + mlen > limit - out) { bad = true; return false; }
            const unsigned long long tg0 = SYT_NOW();
            flush();
            const uint32_t have = (resumed && tp == rtp) ? 1u : 0u;             // the literal run of this token is in place (k_sy_big)
            const uint32_t md = out + lit;
            // the token behind this sequence (LZ4's length code is unique: the number of extension bytes follows from the length)
            const uint32_t pnext = ls + lit + (mlen ? 2u + (mlen >= 19u ? (mlen - 19u) / 255u + 1u : 0u) : 0u);
            if (lit && !(have & 1u)) {
                if (lit >= SY_BIG && !last) { park(tp, 0u, out, ls, lit, lit, mlen, off, pnext); return false; }
                sy_lits_wave<SYM>(D, S, out, src + ls, lit, lane);
                unsynced = true;
            }
            if (mlen) {
                if (off == 0u || off > md - base) { bad = true; return false; }
                if (mlen >= SY_BIG && !last) { park(tp, 1u, md, off, mlen, lit, mlen, off, pnext); return false; }
                if (mlen <= SY_IMG - SY_HIST) {
                    const uint32_t K = lit < SY_HIST ? lit : SY_HIST;
                    const uint8_t *g = src + ls + (lit - K);
                    wave_sync();
                    if (K < 16u) { if ((uint32_t)lane < K) { s_d[lane] = g[lane]; if (SYM) s_s[lane] = 0; } }
                    else for (uint32_t i = (uint32_t)lane * 16u; i < K; i += 1024u) {
                        const uint32_t j = i + 16u <= K ? i : K - 16u;
                        u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
                        ((hb_u128u *)(s_d + j))->v = ld16u(g + j);
                        if (SYM) { ((hb_u128u *)((uint8_t *)s_s + 2u * j))->v = z; ((hb_u128u *)((uint8_t *)s_s + 2u * j + 16u))->v = z; }
                    }
                    wave_sync();
                    ib = md - K; fl = md; out = md;
                    SYT_ADD(7, tg0);
                    return batch(1u, tp, md, 0u, mlen, off, 0u);
                }
                sy_sync(); unsynced = false;
                sy_match_wave<SYM>(D, S, md, off, mlen, O, lane);
            }
            out = md + mlen; ib = out; fl = out;
            unsynced = true;
            SYT_ADD(7, tg0);
            return true;
        };
        // Resuming behind a literal run k_sy_big has copied: the token is NOT parsed again -- the run in front of an incompressible byte plane is
        // 256 MiB and more, its length extension a MiB of FF bytes that one wavefront reads at 4 KiB per memory round trip (0.3 ms for the headline
        // frame, 1 ms for 850 MiB of random floats: that was the whole second launch of pass A, everybody else idle).  The whole sequence is on record
        // (SyUnit), and the number of extension bytes follows from a length (LZ4's length code is unique).
        uint32_t walk_from = start;
        bool ok = true;
        if (resumed) {
            const uint32_t lit = RFL(R->plit), mlen = RFL(R->pmlen), off = RFL(R->poff);
            const uint32_t ls = rtp + 1u + (lit >= 15u ? (lit - 15u) / 255u + 1u : 0u);
            walk_from = RFL(R->pnext);
            ok = single(rtp, ls, lit, mlen, off, mlen ? 0u : (uint32_t)src[rtp] & 0xF0u);
        }
        if constexpr (CODEC == RG_SNAPPY) {                            // elements instead of sequences (sn_walk, hb_lz4_region.h); nothing ever parks
            if (ok && !parked) ok = sn_walk<PWIN>(src, n_src, walk_from, exitp, s_win, s_tq, lane, batch, single);
        } else if constexpr (!TOK) {
            if (ok && !parked) ok = rg_walk<PWIN>(src, n_src, walk_from, exitp, s_win, s_tq, lane, batch, single);
        } else {
            // Region by region of the token discovery: where a region's first parse left its tokens (from RgRegion.pad0 on they are the chain's, if that
            // parse ends where the chain does), they are read, not parsed again (rg_walk_tok); the stretch in front of pad0 -- what the first parse
            // needed to fall onto the chain -- regions without a usable store and tokens the store cannot hand over are parsed as before.  One call site
            // per walker: every inlined copy carries the whole batch decoder.  (Round 3: pass A 2.95 -> 2.30 ms on the headline reference frame, 6.2 -> 4.65
            // on bit-shuffled data -- as a kernel of its own; sharing one kernel with the parsing walk behind a run-time switch it spilled 57 registers
            // and lost.)
            uint32_t pos = walk_from;
            const uint2 *tk = nullptr;
            uint32_t k0 = 0, k1 = 0, tokend = 0;                            // stored tokens still to read: tk[k0 .. k1), then the position is tokend
            while (ok && !parked && (pos < exitp || k0 < k1)) {
                uint32_t wfrom = pos, seg_end = exitp;                      // walk [wfrom, seg_end) unless the store takes over
                bool one = false;                                           // walking ONE token the store could not hand over
                if (k0 < k1) {
                    const int rc = rg_walk_tok<PWIN>(src, n_src, tk, k0, k1, s_win, lane, batch);
                    if (rc == 0) { ok = false; break; }
                    if (rc == 1) { pos = tokend; k0 = k1 = 0; continue; }
                    wfrom = RFL(tk[k0].x); seg_end = wfrom + 1u; one = true; k0 += 1u;   // rg_walk parses exactly that token
                } else if (ts != nullptr) {
                    const uint32_t r = pos / ts->rs;
                    if (r < ts->nreg) {
                        const RgRegion *G = ts->reg + r;
                        const uint32_t gexit = RFL(G->exit), pad0 = RFL(G->pad0), nt = RFL(G->pad1[0]);
                        const uint32_t rend = gexit < exitp ? gexit : exitp;             // where this unit's share of region r ends
                        if (rend > pos) {
                            seg_end = rend;
                            const bool usable = nt != RG_INVALID && nt != 0u && pad0 != RG_INVALID && RFL(G->exit0) == gexit && pad0 < rend;
                            if (usable && pos < pad0) seg_end = pad0;                    // the head first
                            else if (usable) {
                                tk = ts->tok + (size_t)r * ts->tokcap;
                                k0 = rg_tok_lower(tk, nt, pos, lane);
                                k1 = rend >= gexit ? nt : rg_tok_lower(tk, nt, rend, lane);
                                if (k0 < k1 && RFL(tk[k0].x) == pos) { tokend = rend; continue; }
                                k0 = k1 = 0;                                             // (pos is not a stored token: walk)
                            }
                        }
                    }
                }
                ok = rg_walk<PWIN>(src, n_src, wfrom, seg_end, s_win, s_tq, lane, batch, single);
                if (!one) pos = seg_end;
                else if (k0 >= k1) { pos = tokend; k0 = k1 = 0; }
            }
        }
        if (!parked) flush();
#ifdef SY_DEBUG_TIMES
        dbg_t[0] = SYT_NOW() - dbg_t0;
        if (lane == 0) {
            for (int k = 0; k < 16; k++) atomicAdd(&sy_dbg[k], dbg_t[k]);
            int hb = 63 - __builtin_clzll(dbg_t[0] | 1ull) - 12;           // histogram of unit times: bin b = [2^(b+12), 2^(b+13)) clocks
            hb = hb < 0 ? 0 : (hb > 14 ? 14 : hb);
            atomicAdd(&sy_dbg[16 + hb], 1ull);
            atomicMax(&sy_dbg[31], dbg_t[0]);
            if (rtp) { atomicAdd(&sy_dbg[36], 1ull); atomicAdd(&sy_dbg[37], dbg_t[0]); atomicMax(&sy_dbg[38], dbg_t[0]); atomicAdd(&sy_dbg[39], dbg_t[8]); }
        }
#endif
        return ok && !bad;
}
