// hb_lz4_sym.hip — parallel decode of FOREIGN LZ4 blocks: frames the reference writes (lz4.CompressBlock, codec.go:63-75: one block,
// matches reach 65535 bytes back and cross every boundary one could cut at), for which no restart index can exist.
//
// hb_lz4_region.hip finds and verifies the token chain of such a block and gives every region of the stream its place in the output,
// but a region still cannot be DECODED on its own: its matches copy bytes that other regions produce.  They can be decoded
// SYMBOLICALLY though (the idea of pugz, Kerbiriou & Chikhi 2019, for gzip): a byte a region cannot know is not a value but a
// reference "the byte d positions before my first output byte", d in 1..65535 (LZ4 offsets are 16 bits, so nothing further back is
// ever named).  Copies copy references like values, so after one pass every output byte is either a value or a reference into the
// 65535 bytes in front of its region -- chains of matches inside the region are flattened by the decode itself.
//
//   pass A  k_sy_decode    one wavefront per region: tokens 64 at a time, one per lane (window-parallel parser), literals and matches
//                          copied lane-parallel in dependency rounds straight in HBM; values go to the output buffer, references to a
//                          u16 side array (0 = "is a value").
//   pass B  the 65535 bytes in front of a region are the tail of its predecessors, themselves symbolic: the tail of region r as a
//           function of the tail in front of it is a MAP of 65535 entries {value | reference}, maps compose associatively, so the
//           tails are resolved by a scan over the regions: k_sy_compose (per group of regions: the composed map, sequential inside
//           the group, groups in parallel), k_sy_chain (one workgroup: the resolved tail in front of every group, one map per step,
//           the tail in LDS), k_sy_resolve (per group again: every region's bytes resolved against the now known tail in front of
//           it -- 64 KiB of LDS, one byte gather per reference -- and the tail rolled forward).
//
// Whatever is wrong with the stream (offset 0, offset before the start of the block) only raises SyPlan.fail: the single wavefront
// then decodes and reports what lz4.UncompressBlock reports.  A block that passes is decoded to exactly the bytes the serial
// decoder produces -- every output byte is written from the same source by the same rule, only in a different order.
#include "hb_lz4_region.h"

#define SY_W        65536u        // entries of a tail map / bytes of a tail image (index = distance 1..65535; entry 0 unused)
#define SY_GROUPS   128u          // groups of regions in pass B
#define SY_IMG      4096u         // bytes of output a wave holds in LDS: the last SY_HIST bytes it wrote to HBM + what it is building
#define SY_HIST     2048u
#define SY_SUB      8u            // a region with more than SY_HEAVY bytes of output is decoded in this many parts
#define SY_HEAVY    (1u << 20)
#define SY_MAXUNITS (RG_MAXREG * SY_SUB)
#define SY_PIECE    (256u << 10)  // bytes of a region one workgroup resolves at a time (k_sy_resolve)
#define SY_NCAP     64u           // tile-to-tile copies up to this long are done by their own lane
#define SY_BIG      (256u << 10)  // literal runs / matches from this size on are copied by the whole chip (k_sy_big)
#define SY_ROUNDS   3             // launches of pass A; the last one copies everything inline

struct SyPlan { uint32_t go, fail, groups, per, nbig, nunits, nact; uint32_t pad[9]; };   // per: units WITH OUTPUT per group of pass B
// what one wavefront of pass A decodes: a region of the token discovery, or one of SY_SUB parts of a region whose output is large
struct SyUnit { uint32_t entry, exit, opos, outlen, rtp, rout, state, pad; };   // state: bit 0 done, bit 1 / 2: literals / match of token rtp copied
struct SyBig { uint32_t kind, dst, src, len, O, pad[3]; };          // kind 0: literals from stream position src; 1: match, src = offset
struct SyLayout { size_t plan, par, units, list, big, items, sym, maps, tails, total; };
static inline uint32_t sy_max_groups(size_t n_out) {
    const size_t units_max = (hb_lz4_bound(n_out) / RG_MINREG + 1) * SY_SUB;       // a block of n_out bytes is at most this long
    return (uint32_t)(units_max < SY_GROUPS ? units_max : SY_GROUPS);
}
static inline SyLayout sy_layout(size_t n_out) {
    SyLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
    const uint32_t g = sy_max_groups(n_out);
    L.plan = take(sizeof(SyPlan));
    L.par = take((size_t)SY_GROUPS * 4);
    L.units = take((size_t)SY_MAXUNITS * sizeof(SyUnit));
    L.list = take((size_t)SY_MAXUNITS * 4);
    L.big = take((size_t)SY_MAXUNITS * sizeof(SyBig));
    L.items = take((size_t)(SY_MAXUNITS + 1) * 4);
    L.sym = take(2 * (n_out + 64));
    L.maps = take((size_t)g * 2 * SY_W * 4);
    L.tails = take((size_t)(g + 1) * SY_W);
    L.total = o;
    return L;
}
size_t hb_lz4_sym_workspace(size_t n_out) { return sy_layout(n_out).total; }

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
__device__ __forceinline__ void sy_fetch_lane(const uint8_t *D, const uint16_t *S, uint8_t *s_d, uint16_t *s_s, uint32_t tf, uint32_t sp, uint32_t far, const uint32_t O) {
    const uint8_t *Sb = (const uint8_t *)S;
    uint8_t *s_sb = (uint8_t *)s_s;
    {   // bytes from in front of the region are references: nothing to load
        const uint32_t nctx = sp < O ? (O - sp < far ? O - sp : far) : 0u;
        for (uint32_t k = 0; k < nctx; k++) s_s[tf + k] = (uint16_t)(O - sp - k);
        tf += nctx; sp += nctx; far -= nctx;
    }
    const bool f8 = far >= 8u, f4 = far >= 4u && far < 8u, f1 = far != 0u && far < 4u;
    uint64_t fv[4]; u32x4 fa[4]; uint32_t fk[4];
    uint32_t fw0 = 0, fw1 = 0; uint64_t fb0 = 0, fb1 = 0;
    uint8_t fx[3]; uint16_t fy[3];
    // ---- loads ----
    if (f8) {
#pragma unroll
        for (int c = 0; c < 4; c++) { fk[c] = 8u * c; if (fk[c] < far) { if (fk[c] + 8u > far) fk[c] = far - 8u; fv[c] = ld8u(D + sp + fk[c]); fa[c] = ld16u(Sb + 2u * (size_t)(sp + fk[c])); } }
    }
    if (f4) { fw0 = ld4u(D + sp); fw1 = ld4u(D + sp + far - 4u); fb0 = ld8u(Sb + 2u * (size_t)sp); fb1 = ld8u(Sb + 2u * (size_t)(sp + far - 4u)); }
    if (f1) {
#pragma unroll
        for (int k = 0; k < 3; k++) if ((uint32_t)k < far) { fx[k] = D[sp + k]; fy[k] = S[sp + k]; }
    }
    // ---- stores ----
    if (f8) {
#pragma unroll
        for (int c = 0; c < 4; c++) if (8u * c < far) { ((hb_u64u *)(s_d + tf + fk[c]))->v = fv[c]; ((hb_u128u *)(s_sb + 2u * (tf + fk[c])))->v = fa[c]; }
    }
    if (f4) {
        ((hb_u32u *)(s_d + tf))->v = fw0; ((hb_u32u *)(s_d + tf + far - 4u))->v = fw1;
        ((hb_u64u *)(s_sb + 2u * tf))->v = fb0; ((hb_u64u *)(s_sb + 2u * (tf + far - 4u)))->v = fb1;
    }
    if (f1) {
#pragma unroll
        for (int k = 0; k < 3; k++) if ((uint32_t)k < far) { s_d[tf + k] = fx[k]; s_s[tf + k] = fy[k]; }
    }
    // ---- beyond 32 bytes ----
    for (uint32_t blk = 32u; blk < far; blk += 32u) {
        uint64_t v[4]; u32x4 a[4]; uint32_t k[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            k[c] = blk + 8u * c;
            if (k[c] < far) { if (k[c] + 8u > far) k[c] = far - 8u; v[c] = ld8u(D + sp + k[c]); a[c] = ld16u(Sb + 2u * (size_t)(sp + k[c])); }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) if (blk + 8u * c < far) { ((hb_u64u *)(s_d + tf + k[c]))->v = v[c]; ((hb_u128u *)(s_sb + 2u * (tf + k[c])))->v = a[c]; }
    }
}
// the same by the whole wave (len > 64)
__device__ __forceinline__ void sy_far_wave(const uint8_t *D, const uint16_t *S, uint8_t *s_d, uint16_t *s_s, uint32_t t, uint32_t sp, uint32_t len, const uint32_t O, const int lane) {
    const uint32_t nctx = sp < O ? (O - sp < len ? O - sp : len) : 0u;
    for (uint32_t i = lane; i < nctx; i += 64) s_s[t + i] = (uint16_t)(O - sp - i);
    t += nctx; sp += nctx; len -= nctx;
    const uint8_t *Sb = (const uint8_t *)S;
    uint8_t *s_sb = (uint8_t *)s_s;
    if (len < 16u) { if ((uint32_t)lane < len) { s_d[t + lane] = D[sp + lane]; s_s[t + lane] = S[sp + lane]; } return; }
    for (uint32_t blk = 0; blk < len; blk += 4096u) {
        u32x4 v[4], a[4], b[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < len) {
                if (j[q] + 16u > len) j[q] = len - 16u;
                v[q] = ld16u(D + sp + j[q]); a[q] = ld16u(Sb + 2u * (size_t)(sp + j[q])); b[q] = ld16u(Sb + 2u * (size_t)(sp + j[q]) + 16u);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < len) {
                ((hb_u128u *)(s_d + t + j[q]))->v = v[q]; ((hb_u128u *)(s_sb + 2u * (t + j[q])))->v = a[q]; ((hb_u128u *)(s_sb + 2u * (t + j[q]) + 16u))->v = b[q];
            }
    }
}
// literals: staged stream window (LDS) -> image; values, never references
__device__ __forceinline__ void sy_lits_img_lane(uint8_t *s_d, uint16_t *s_s, const uint32_t t, const uint8_t *lp, const uint32_t lit) {
    lds_copy_exact(s_d + t, lp, lit);
    uint8_t *z = (uint8_t *)(s_s + t);
    uint32_t k = 0;
    for (; k + 8u <= 2u * lit; k += 8u) ((hb_u64u *)(z + k))->v = 0ull;
    if ((2u * lit) & 4u) { ((hb_u32u *)(z + k))->v = 0u; k += 4u; }
    if ((2u * lit) & 2u) ((hb_u16u *)(z + k))->v = 0;
}
__device__ __forceinline__ void sy_lits_img_wave(uint8_t *s_d, uint16_t *s_s, const uint32_t t, const uint8_t *lp, const uint32_t lit, const int lane) {
    for (uint32_t k = lane; k < lit; k += 64) { s_d[t + k] = lp[k]; s_s[t + k] = 0; }
}
// near copies: tile[md + k] = tile[md - off + k], values and references alike.  One lane (what lds_match_lane does for bytes):
__device__ __forceinline__ void sy_near_lane(uint8_t *s_d, uint16_t *s_s, const uint32_t md, const uint32_t off, const uint32_t len) {
    lds_match_lane(s_d, md, off, len);
    uint16_t *d = s_s + md;
    const uint16_t *s = d - off;
    uint32_t k = 0;
    if (off >= 4u) for (; k + 4u <= len; k += 4u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(s + k))->v;
    for (; k < len; k++) d[k] = s[k];
}
// ... and the whole wave
__device__ __forceinline__ void sy_near_wave(uint8_t *s_d, uint16_t *s_s, const uint32_t md, const uint32_t off, const uint32_t len, const int lane) {
    dec_match_copy(s_d, md, off, len, lane);
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
__device__ __forceinline__ void sy_lits_wave(uint8_t *D, uint16_t *S, const uint32_t d0, const uint8_t *g, const uint32_t lit, const int lane) {
    uint8_t *Sb = (uint8_t *)S;
    u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
    if (lit < 16u) { if ((uint32_t)lane < lit) { D[d0 + lane] = g[lane]; S[d0 + lane] = 0; } return; }
    for (uint32_t blk = 0; blk < lit; blk += 4096u) {
        u32x4 v[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < lit) { if (j[q] + 16u > lit) j[q] = lit - 16u; v[q] = ld16u(g + j[q]); }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < lit) { st16u(D + d0 + j[q], v[q]); st16u(Sb + 2u * (size_t)(d0 + j[q]), z); st16u(Sb + 2u * (size_t)(d0 + j[q]) + 16u, z); }
    }
}
// out[md + i] = out[s0 + i], i < len, the whole wave, source and destination do not overlap (len <= md - s0)
__device__ __forceinline__ void sy_copy_wave(uint8_t *D, uint16_t *S, const uint32_t md, const uint32_t s0, const uint32_t len, const uint32_t O, const int lane) {
    const uint32_t nctx = s0 < O ? (O - s0 < len ? O - s0 : len) : 0u;          // leading bytes that come from in front of the region
    for (uint32_t i = lane; i < nctx; i += 64) S[md + i] = (uint16_t)(O - s0 - i);
    uint8_t *Sb = (uint8_t *)S;
    const uint32_t n = len - nctx, d = md + nctx, s = s0 + nctx;
    if (n < 16u) {
        if ((uint32_t)lane < n) { const uint8_t v = D[s + lane]; const uint16_t a = S[s + lane]; D[d + lane] = v; S[d + lane] = a; }
        return;
    }
    for (uint32_t blk = 0; blk < n; blk += 4096u) {
        u32x4 v[4], a[4], b[4]; uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            j[q] = blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u;
            if (j[q] < n) {
                if (j[q] + 16u > n) j[q] = n - 16u;
                v[q] = ld16u(D + s + j[q]); a[q] = ld16u(Sb + 2u * (size_t)(s + j[q])); b[q] = ld16u(Sb + 2u * (size_t)(s + j[q]) + 16u);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++)
            if (blk + (uint32_t)q * 1024u + (uint32_t)lane * 16u < n) {
                st16u(D + d + j[q], v[q]); st16u(Sb + 2u * (size_t)(d + j[q]), a[q]); st16u(Sb + 2u * (size_t)(d + j[q]) + 16u, b[q]);
            }
    }
}
// a whole match by the whole wave.  Overlapping (off < mlen: the `off` bytes in front of it, repeated): the first 128..256 bytes are
// gathered byte by byte from that period, then pieces that double -- what is copied already is source for the next piece -- so a run
// of any length and period costs 1 + log2(length / 256) round trips.
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
            if (k < P) { const uint32_t sp = s0 + k % off; if (sp >= O) { v[q] = D[sp]; a[q] = S[sp]; } else { v[q] = 0; a[q] = (uint16_t)(O - sp); } }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { const uint32_t k = (uint32_t)lane + 64u * q; if (k < P) { D[md + k] = v[q]; S[md + k] = a[q]; } }
        done = P;
        if (done < mlen) sy_sync();
    }
    while (done < mlen) {
        const uint32_t room = off + done, left = mlen - done;
        const uint32_t piece = left < room ? left : room;               // `done` stays a multiple of the period until the last piece
        sy_copy_wave(D, S, md + done, s0, piece, O, lane);
        done += piece;
        if (done < mlen) sy_sync();
    }
}

// go: the token chain is verified and nothing has decoded the block yet
__global__ void k_sy_gate(const RgPlan *rg, const DecPlan *dp, SyPlan *sy, uint32_t groups, uint32_t per) {
    sy->go = (rg->ok && !(dp->mode == DEC_INDEXED && !dp->fail)) ? 1u : 0u;
    sy->fail = 0; sy->groups = groups; sy->per = per; sy->nbig = 0; sy->nunits = rg->nreg * SY_SUB;
}

// ---- the units of pass A.  A region of the discovery is a fixed span of the STREAM; where the data compresses 100:1 that is tens of
// MiB of output made of KiB-long matches, and one wavefront copying them is what everybody else would wait for.  Such a region is
// cut into up to SY_SUB parts of about equal output, at tokens the discovery has on record anyway: its trace holds the first
// token of each of RG_BUCKETS slices of the region's stream with the output produced up to there (counted from the start of the
// recorded parse; behind the point where that parse met the final chain, RgRegion.pad0, the difference of the two output
// lengths converts it). ----
__global__ __launch_bounds__(64) void k_sy_units(const RgPlan *rg, const RgRegion *__restrict__ reg, const uint2 *__restrict__ traces, const SyPlan *sy, SyUnit *un) {
    __shared__ uint32_t s_e[SY_SUB + 1], s_o[SY_SUB + 1];
    if (!sy->go) return;
    const int lane = threadIdx.x;
    const uint32_t nreg = rg->nreg;
    for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x) {
        const RgRegion R = reg[r];
        const uint32_t entry = R.entry, exitp = R.exit, L = R.outlen, O = (uint32_t)R.opos;
        SyUnit *u = un + (size_t)r * SY_SUB;
        wave_sync();
        if ((uint32_t)lane <= SY_SUB) { s_e[lane] = exitp; s_o[lane] = O + L; }
        if (lane == 0) { s_e[0] = entry; s_o[0] = O; }
        wave_sync();
        if (L > SY_HEAVY && entry < exitp && R.exit0 == exitp) {
            const uint2 *tr = traces + (size_t)r * RG_TRACE + RG_DENSE;
            const uint32_t conv = L - R.outlen0;                         // recorded output count -> output since `entry` (mod 2^32)
            const uint32_t step = (L + SY_SUB - 1u) / SY_SUB;
            uint32_t k = 1;
            for (uint32_t b0 = 0; b0 < RG_BUCKETS && k < SY_SUB; b0 += 64) {
                const uint2 t = tr[b0 + lane];
                const bool ok = t.x != RG_INVALID && t.x >= R.pad0 && t.x > entry && t.x < exitp;
                const uint32_t at = t.y + conv;                          // output bytes in front of this token
                while (k < SY_SUB) {
                    const unsigned long long m = hb_ballot(ok && at >= k * step && at < L);
                    if (!m) break;
                    const int j = __builtin_ctzll(m);
                    if (lane == j) { s_e[k] = t.x; s_o[k] = O + at; }
                    k++;
                }
            }
            wave_sync();
        }
        if ((uint32_t)lane < SY_SUB) {
            SyUnit x;
            x.entry = s_e[lane]; x.exit = s_e[lane + 1]; x.opos = s_o[lane]; x.outlen = s_o[lane + 1] - s_o[lane];
            x.rtp = 0; x.rout = 0; x.state = 0; x.pad = 0;
            u[lane] = x;
        }
    }
}

// the units that have output, in order: all later kernels walk this list (most slots of `un` are empty)
__global__ __launch_bounds__(1024) void k_sy_compact(SyPlan *sy, const SyUnit *__restrict__ un, uint32_t *__restrict__ list) {
    __shared__ uint32_t s[1024];
    if (!sy->go) return;
    const int t = threadIdx.x;
    const uint32_t nu = sy->nunits;
    constexpr uint32_t PER = SY_MAXUNITS / 1024;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < PER; k++) { const uint32_t r = (uint32_t)t * PER + k; if (r < nu && un[r].outlen != 0u) cnt++; }
    s[t] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t y = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += y;
        __syncthreads();
    }
    uint32_t o = s[t] - cnt;
    for (uint32_t k = 0; k < PER; k++) { const uint32_t r = (uint32_t)t * PER + k; if (r < nu && un[r].outlen != 0u) list[o++] = r; }
    if (t == 1023) {
        const uint32_t n = s[1023], g = sy->groups;
        sy->nact = n;
        sy->per = n ? (n + g - 1u) / g : 1u;
    }
}

// ---- pass A ----
// A literal run or a match of SY_BIG bytes and more is not for one wavefront (2-3 GB/s): the region posts it, stops in front of it
// and is resumed by the next launch, after k_sy_big has done all posted copies with the whole chip.  The LAST launch copies inline.
// Unit state between launches (SyUnit): rtp = stream position of the token to resume at (0: not started), rout = output position
// there, state bit 0 = unit done, bit 1 = that token's literals are copied, bit 2 = its match too.
__global__ __launch_bounds__(64) void k_sy_decode(const uint8_t *__restrict__ src, uint64_t n_src, SyUnit *un, const uint32_t *__restrict__ list,
                                                   SyPlan *sy, SyBig *big, uint8_t *D, uint16_t *S, int last) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    __shared__ __attribute__((aligned(16))) uint8_t s_d[SY_IMG + 64];
    __shared__ __attribute__((aligned(16))) uint16_t s_s[SY_IMG + 64];
    if (!sy->go || sy->fail) return;
    const int lane = threadIdx.x;
    const uint32_t nact = sy->nact;
    uint8_t *Sb = (uint8_t *)S;
    for (uint32_t i = blockIdx.x; i < nact; i += gridDim.x) {
        SyUnit *R = un + list[i];
        const uint32_t entry = RFL(R->entry), exitp = RFL(R->exit);
        if (RFL(R->outlen) == 0u || entry >= exitp) continue;
        const uint32_t st = RFL(R->state);
        if (st & 1u) continue;
        const uint32_t O = RFL(R->opos);
        const uint32_t rtp = RFL(R->rtp);
        const uint32_t start = rtp ? rtp : entry;
        uint32_t out = rtp ? RFL(R->rout) : O;         // next output byte
        // the image: s_d[i] / s_s[i] = output byte ib + i for i < out - ib; bytes below fl are in HBM already (history kept for near copies)
        uint32_t ib = out, fl = out;
        bool bad = false, parked = false, unsynced = false;
        // the new part of the image goes to HBM
        auto flush = [&]() __attribute__((always_inline)) {
            const uint32_t n = out - fl;
            if (n == 0u) return;
            wave_sync();
            const uint32_t o = fl - ib;
            const uint8_t *s_sb = (const uint8_t *)s_s;
            if (n < 16u) {
                if ((uint32_t)lane < n) { D[fl + lane] = s_d[o + lane]; S[fl + lane] = s_s[o + lane]; }
            } else {
                for (uint32_t i = (uint32_t)lane * 16u; i < n; i += 1024u) {
                    const uint32_t j = i + 16u <= n ? i : n - 16u;
                    st16u(D + fl + j, ((const hb_u128u *)(s_d + o + j))->v);
                    st16u(Sb + 2u * (size_t)(fl + j), ((const hb_u128u *)(s_sb + 2u * (o + j)))->v);
                    st16u(Sb + 2u * (size_t)(fl + j) + 16u, ((const hb_u128u *)(s_sb + 2u * (o + j) + 16u))->v);
                }
            }
            fl = out;
            unsynced = true;                                             // far loads must wait for these stores (sy_sync), not the flush itself
        };
        // make room: flush, keep the last SY_HIST bytes as history at the bottom of the image
        auto slide = [&]() __attribute__((always_inline)) {
            flush();
            const uint32_t have = out - ib, keep = have < SY_HIST ? have : SY_HIST, delta = have - keep;
            if (delta) {
                uint8_t *s_sb = (uint8_t *)s_s;
                wave_sync();
                for (uint32_t k = (uint32_t)lane * 16u; k < keep; k += 1024u) {          // ascending: a step's reads are done before its writes
                    const u32x4 v = ((const hb_u128u *)(s_d + delta + k))->v;
                    ((hb_u128u *)(s_d + k))->v = v;
                }
                for (uint32_t k = (uint32_t)lane * 16u; k < 2u * keep; k += 1024u) {
                    const u32x4 v = ((const hb_u128u *)(s_sb + 2u * delta + k))->v;
                    ((hb_u128u *)(s_sb + k))->v = v;
                }
                wave_sync();
            }
            ib = out - keep;
        };
        auto batch = [&](uint32_t cnt, uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t lp) __attribute__((always_inline)) -> bool {
            (void)tp; (void)ls;
            const bool tok = (uint32_t)lane < cnt;
            const uint32_t olen = tok ? lit + mlen : 0u;
            const uint32_t incl = wave_incl_scan_dpp(olen);
            const uint32_t d0 = out + incl - olen, md = d0 + lit;
            if (hb_ballot(tok && (off == 0u || off > md))) { bad = true; return false; }     // offset 0 / before the start of the block
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
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
                        flush();
                        sy_sync(); unsynced = false;
                        const uint32_t dl = __builtin_amdgcn_readlane(d0, (int)lo), ll = __builtin_amdgcn_readlane(lit, (int)lo);
                        if (ll) { sy_lits_wave(D, S, dl, src + __builtin_amdgcn_readlane(ls, (int)lo), ll, lane); sy_sync(); }
                        sy_match_wave(D, S, dl + ll, __builtin_amdgcn_readlane(off, (int)lo), __builtin_amdgcn_readlane(mlen, (int)lo), O, lane);
                        sy_sync();
                        out = dl + ol; ib = out; fl = out;
                        lo++;
                    } else slide();
                    continue;
                }
                const bool act = tok && (uint32_t)lane >= lo && (uint32_t)lane < hi;
                const uint32_t t0 = d0 - ib, tm = md - ib;
                // literals: from the staged stream window
                if (act && lit <= 32u) sy_lits_img_lane(s_d, s_s, t0, s_win + lp, lit);
                unsigned long long lm = hb_ballot(act && lit > 32u);
                while (lm) {
                    const int l = __builtin_ctzll(lm);
                    sy_lits_img_wave(s_d, s_s, __builtin_amdgcn_readlane(t0, l), s_win + __builtin_amdgcn_readlane(lp, l), __builtin_amdgcn_readlane(lit, l), lane);
                    lm &= lm - 1;
                }
                // the part of every match whose source lies in front of the image: from HBM, all lanes at once (nothing in the batch can
                // change those bytes)
                const uint32_t farlen = (act && s0 < ib) ? (ib - s0 < mlen ? ib - s0 : mlen) : 0u;
                if (hb_ballot(farlen != 0u)) {
                    if (unsynced) { sy_sync(); unsynced = false; }
                    sy_fetch_lane(D, S, s_d, s_s, tm, s0, farlen <= thr ? farlen : 0u, O);
                    lm = hb_ballot(farlen > thr);
                    while (lm) {
                        const int l = __builtin_ctzll(lm);
                        sy_far_wave(D, S, s_d, s_s, __builtin_amdgcn_readlane(tm, l), __builtin_amdgcn_readlane(s0, l), __builtin_amdgcn_readlane(farlen, l), O, lane);
                        lm &= lm - 1;
                    }
                }
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
                        sy_near_wave(s_d, s_s, X, __builtin_amdgcn_readlane(off, f), nlf, lane);
                        pend &= pend - 1;
                        continue;
                    }
                    const unsigned long long below = pend & ((1ull << lane) - 1ull);
                    const uint32_t pj = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
                    const uint32_t pe = (uint32_t)__shfl((int)mend, (int)pj);
                    const bool ready = ((pend >> lane) & 1ull) && nlen <= SY_NCAP && (srcend <= X || below == 0ull || srcs >= pe);
                    if (ready) sy_near_lane(s_d, s_s, nmd, off, nlen);
                    pend &= ~hb_ballot(ready);
                }
                out = hi == cnt ? end_all : __builtin_amdgcn_readlane(d0, (int)hi);
                lo = hi;
            }
            return true;
        };
        auto park = [&](uint32_t tp, uint32_t flags, uint32_t kind, uint32_t dst, uint32_t a, uint32_t len) __attribute__((always_inline)) {
            if (lane == 0) {
                const uint32_t slot = atomicAdd(&sy->nbig, 1u);             // (at most one per region and launch: nreg slots)
                SyBig b; b.kind = kind; b.dst = dst; b.src = a; b.len = len; b.O = O; b.pad[0] = b.pad[1] = b.pad[2] = 0;
                big[slot] = b;
                R->rtp = tp; R->rout = out; R->state = flags;
            }
            parked = true;
        };
        // sequences the window parser leaves alone (lengths of KiB and more, the edges of the staged window): straight in HBM
        auto single = [&](uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t tok) __attribute__((always_inline)) -> bool {
            if (mlen == 0u && (tok & 15u) != 0u) { bad = true; return false; }        // the input ends after literals but a match was announced
            flush();
            sy_sync(); unsynced = false;
            const uint32_t have = (tp == rtp) ? (st >> 1) : 0u;                 // what earlier launches did of this token
            if (lit && !(have & 1u)) {
                if (lit >= SY_BIG && !last) { park(tp, 2u, 0u, out, ls, lit); return false; }
                sy_lits_wave(D, S, out, src + ls, lit, lane);
            }
            const uint32_t md = out + lit;
            if (mlen && !(have & 2u)) {
                if (off == 0u || off > md) { bad = true; return false; }
                if (mlen >= SY_BIG && !last) { park(tp, 2u | 4u, 1u, md, off, mlen); return false; }
                sy_sync();
                sy_match_wave(D, S, md, off, mlen, O, lane);
            }
            out = md + mlen; ib = out; fl = out;
            sy_sync();
            return true;
        };
        const bool ok = rg_walk(src, n_src, start, exitp, s_win, s_tq, lane, batch, single);
        if (!parked) {
            flush();
            if ((!ok || bad || out != O + RFL(R->outlen)) && lane == 0) atomicExch(&sy->fail, 1u);
            if (lane == 0) R->state = 1u;
        }
        wave_sync();
    }
}

// the copies pass A posted, with the whole chip: 16 KiB per workgroup and step
__global__ __launch_bounds__(256) void k_sy_big(const uint8_t *__restrict__ src, SyPlan *sy, const SyBig *__restrict__ big, uint8_t *D, uint16_t *S, int reset) {
    if (!sy->go || sy->fail) return;
    const uint32_t nb = sy->nbig;
    const int t = threadIdx.x;
    uint8_t *Sb = (uint8_t *)S;
    for (uint32_t i = 0; i < nb; i++) {
        const SyBig b = big[i];
        const uint32_t nch = (b.len + 16383u) >> 14;
        for (uint32_t c = blockIdx.x; c < nch; c += gridDim.x) {
            const uint32_t x0 = c << 14, x1 = x0 + 16384u < b.len ? x0 + 16384u : b.len;
            if (b.kind == 0u) {                                          // literals from the stream (b.len >= SY_BIG: the last piece moves back)
                u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
                for (uint32_t x = x0 + (uint32_t)t * 16u; x < x1; x += 4096u) {
                    const uint32_t j = x + 16u <= b.len ? x : b.len - 16u;
                    st16u(D + b.dst + j, ld16u(src + b.src + j));
                    st16u(Sb + 2u * (size_t)(b.dst + j), z); st16u(Sb + 2u * (size_t)(b.dst + j) + 16u, z);
                }
            } else {                                                     // a match: the b.src bytes in front of b.dst, repeated
                const uint32_t off = b.src, s0 = b.dst - off;
                for (uint32_t x = x0 + (uint32_t)t; x < x1; x += 256u) {
                    const uint32_t sp = s0 + x % off;
                    if (sp >= b.O) { D[b.dst + x] = D[sp]; S[b.dst + x] = S[sp]; } else S[b.dst + x] = (uint16_t)(b.O - sp);
                }
            }
        }
    }
    (void)reset;
}
// between two launches of pass A
__global__ void k_sy_big_reset(SyPlan *sy) { sy->nbig = 0; }

// ---- pass B ----
// Tail map entry: reference << 16 | value; reference 0 = "is the value".  Index = distance from the END of the span the map covers.
__global__ __launch_bounds__(1024) void k_sy_compose(const SyUnit *__restrict__ un, const uint32_t *__restrict__ list, const SyPlan *sy, const uint8_t *__restrict__ D,
                                                     const uint16_t *__restrict__ S, uint32_t *maps, uint32_t *par) {
    if (!sy->go || sy->fail) return;
    const uint32_t g = blockIdx.x, per = sy->per, nreg = sy->nact;
    const uint32_t r0 = g * per, r1 = r0 + per < nreg ? r0 + per : nreg;
    const int t = threadIdx.x;
    uint32_t *cur = maps + (size_t)g * 2 * SY_W, *nxt = cur + SY_W;
    for (uint32_t d = t; d < SY_W; d += 1024) cur[d] = d << 16;        // the empty span: every byte is the byte in front of it
    __threadfence_block();
    __syncthreads();
    for (uint32_t r = r0; r < r1; r++) {
        const SyUnit u = un[list[r]];
        const uint32_t L = u.outlen;
        if (L == 0u) continue;
        const uint32_t E = u.opos + L;
        const uint32_t *__restrict__ c = cur;
        uint32_t *__restrict__ n = nxt;
        for (uint32_t d0 = 0; d0 < SY_W; d0 += 8192u) {
            uint32_t idx[8], val[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const uint32_t d = d0 + (uint32_t)q * 1024u + (uint32_t)t;
                idx[q] = 0; val[q] = 0;
                if (d == 0u) continue;
                if (d > L) idx[q] = d - L;                              // in front of this region: the same byte, L further from the end
                else if (d <= E) { const uint32_t p = E - d, s = S[p]; if (s) idx[q] = s; else val[q] = D[p]; }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) if (idx[q]) val[q] = c[idx[q]];
#pragma unroll
            for (int q = 0; q < 8; q++) n[d0 + (uint32_t)q * 1024u + (uint32_t)t] = val[q];
        }
        __threadfence_block();
        __syncthreads();
        uint32_t *x = cur; cur = nxt; nxt = x;
    }
    if (t == 0) par[g] = (cur == maps + (size_t)g * 2 * SY_W) ? 0u : 1u;
}

// tails[g][d] = the final byte d positions in front of group g's first output byte (d = 1..65535)
__global__ __launch_bounds__(1024) void k_sy_chain(const SyPlan *sy, const uint32_t *__restrict__ maps, const uint32_t *__restrict__ par, uint8_t *__restrict__ tails) {
    __shared__ __attribute__((aligned(16))) uint8_t s_f[SY_W];
    if (!sy->go || sy->fail) return;
    const int t = threadIdx.x;
    const uint32_t G = sy->groups;
    u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
    for (int q = 0; q < 4; q++) ((u32x4 *)s_f)[t * 4 + q] = z;         // nothing in front of the block (never referenced: pass A checked)
    __syncthreads();
    // the map of group g is fetched while the tail in front of group g is still being written (the maps are all there: k_sy_compose is done)
    u32x4 e[16];
    if (G > 1u) {
        const u32x4 *m = (const u32x4 *)(maps + ((size_t)0 * 2 + par[0]) * SY_W) + (size_t)t * 16;
#pragma unroll
        for (int q = 0; q < 16; q++) e[q] = m[q];
    }
    for (uint32_t g = 0; g < G; g++) {
        for (int q = 0; q < 4; q++) ((u32x4 *)(tails + (size_t)g * SY_W))[t * 4 + q] = ((const u32x4 *)s_f)[t * 4 + q];
        if (g + 1 == G) break;
        uint32_t w[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const u32x4 x = e[q];
            const uint32_t b0 = (x.x >> 16) ? s_f[x.x >> 16] : (x.x & 255u), b1 = (x.y >> 16) ? s_f[x.y >> 16] : (x.y & 255u);
            const uint32_t b2 = (x.z >> 16) ? s_f[x.z >> 16] : (x.z & 255u), b3 = (x.w >> 16) ? s_f[x.w >> 16] : (x.w & 255u);
            w[q] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
        }
        if (g + 2 < G) {
            const u32x4 *m = (const u32x4 *)(maps + ((size_t)(g + 1) * 2 + par[g + 1]) * SY_W) + (size_t)t * 16;
#pragma unroll
            for (int q = 0; q < 16; q++) e[q] = m[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; q++) { u32x4 v; v.x = w[4 * q]; v.y = w[4 * q + 1]; v.z = w[4 * q + 2]; v.w = w[4 * q + 3]; ((u32x4 *)s_f)[t * 4 + q] = v; }
        __syncthreads();
    }
}

// out[p] for p in [p0, p1): a reference becomes the byte it names; s_ring[q & 0xFFFF] = the final byte at position q for the 65535
// positions in front of O.  NT threads; 16 bytes per thread and step between the 16-byte boundaries.
template <int NT>
__device__ __forceinline__ void sy_resolve_range(uint8_t *D, const uint16_t *__restrict__ S, const uint8_t *s_ring, const uint32_t O, const uint32_t p0, const uint32_t p1, const int t) {
    const uint32_t a0 = (p0 + 15u) & ~15u, a1 = p1 & ~15u;
    if (a0 >= a1) {
        for (uint32_t p = p0 + t; p < p1; p += NT) { const uint32_t s = S[p]; if (s) D[p] = s_ring[(O - s) & 0xFFFFu]; }
        return;
    }
    if (p0 + t < a0) { const uint32_t p = p0 + t, s = S[p]; if (s) D[p] = s_ring[(O - s) & 0xFFFFu]; }
    if (a1 + t < p1) { const uint32_t p = a1 + t, s = S[p]; if (s) D[p] = s_ring[(O - s) & 0xFFFFu]; }
    for (uint32_t p = a0 + (uint32_t)t * 16u; p < a1; p += (uint32_t)NT * 16u) {
        const u32x4 sa = ld16u((const uint8_t *)(S + p)), sb = ld16u((const uint8_t *)(S + p + 8));
        if ((sa.x | sa.y | sa.z | sa.w | sb.x | sb.y | sb.z | sb.w) == 0u) continue;
        u32x4 v = ld16u(D + p);
        const uint32_t ss[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t lo = ss[k] & 0xFFFFu, hi = ss[k] >> 16;
            const int sh = (k & 1) * 16;
            if (lo) w[k >> 1] = (w[k >> 1] & ~(0xFFu << sh)) | ((uint32_t)s_ring[(O - lo) & 0xFFFFu] << sh);
            if (hi) w[k >> 1] = (w[k >> 1] & ~(0xFF00u << sh)) | ((uint32_t)s_ring[(O - hi) & 0xFFFFu] << (sh + 8));
        }
        v.x = w[0]; v.y = w[1]; v.z = w[2]; v.w = w[3];
        st16u(D + p, v);
    }
}

// the last 64 KiB of every region of the group, front to back (a region's tail is all a later region can name): resolved against the
// 64 KiB in front of the region, kept as a ring by output position, which then rolls forward over them
__global__ __launch_bounds__(1024) void k_sy_tails(const SyUnit *__restrict__ un, const uint32_t *__restrict__ list, const SyPlan *sy, uint8_t *D, const uint16_t *__restrict__ S,
                                                   const uint8_t *__restrict__ tails) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[SY_W];
    if (!sy->go || sy->fail) return;
    const uint32_t g = blockIdx.x, per = sy->per, nreg = sy->nact;
    const uint32_t r0 = g * per, r1 = r0 + per < nreg ? r0 + per : nreg;
    if (r0 >= nreg) return;
    const int t = threadIdx.x;
    {
        const uint32_t O = un[list[r0]].opos;
        const uint8_t *f = tails + (size_t)g * SY_W;
        for (uint32_t d = t; d < SY_W; d += 1024) s_ring[(O - d) & 0xFFFFu] = f[d];    // (d = 0 lands on O's own slot: rewritten before use)
    }
    __syncthreads();
    for (uint32_t r = r0; r < r1; r++) {
        const SyUnit u = un[list[r]];
        const uint32_t L = u.outlen;
        if (L == 0u) continue;
        const uint32_t O = u.opos, E = O + L;
        const uint32_t from = L > SY_W ? E - SY_W : O;
        sy_resolve_range<1024>(D, S, s_ring, O, from, E, t);
        __threadfence_block();
        __syncthreads();
        for (uint32_t p = from + t; p < E; p += 1024) s_ring[p & 0xFFFFu] = D[p];
        __syncthreads();
    }
}

// what is left: the part of every region in front of its last 64 KiB, in pieces of SY_PIECE bytes over the whole chip.  Everything a
// piece can name is final by now (tails of earlier regions), so the 64 KiB in front of its region are simply read back.
__global__ __launch_bounds__(1024) void k_sy_items(const SyUnit *__restrict__ un, const uint32_t *__restrict__ list, const SyPlan *sy, uint32_t *itembase) {
    __shared__ uint32_t s[1024];
    if (!sy->go || sy->fail) return;
    const int t = threadIdx.x;
    const uint32_t nreg = sy->nact;
    constexpr uint32_t PER = SY_MAXUNITS / 1024;
    uint32_t mine[PER], sum = 0;
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        const uint32_t L = r < nreg ? un[list[r]].outlen : 0u;
        mine[k] = L > SY_W ? (L - SY_W + SY_PIECE - 1u) / SY_PIECE : 0u;
        sum += mine[k];
    }
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t y = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] += y;
        __syncthreads();
    }
    uint32_t o = s[t] - sum;
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        if (r < nreg) itembase[r] = o;
        o += mine[k];
    }
    if (t == 1023) itembase[SY_MAXUNITS] = s[1023];
}
__global__ __launch_bounds__(512) void k_sy_resolve(const SyUnit *__restrict__ un, const uint32_t *__restrict__ list, const SyPlan *sy, uint8_t *D, const uint16_t *__restrict__ S,
                                                    const uint32_t *__restrict__ itembase) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[SY_W];
    __shared__ uint32_t s_r;
    if (!sy->go || sy->fail) return;
    const int t = threadIdx.x;
    const uint32_t nreg = sy->nact, nitems = itembase[SY_MAXUNITS];
    uint32_t have = RG_INVALID;                                        // region whose front is in s_ring
    for (uint32_t i = blockIdx.x; i < nitems; i += gridDim.x) {
        if (t == 0) {                                                   // last region whose first item is <= i
            uint32_t lo = 0, hi = nreg;
            while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (itembase[mid] <= i) lo = mid; else hi = mid; }
            s_r = lo;
        }
        __syncthreads();
        const uint32_t r = s_r;
        const uint32_t O = un[list[r]].opos, L = un[list[r]].outlen, E = O + L;
        if (r != have) {
            const uint32_t nf = O < SY_W - 1u ? O : SY_W - 1u;           // bytes that exist in front of the region
            for (uint32_t d = 16u * (uint32_t)t; d < nf; d += 16u * 512u) {
                const uint32_t q = O - nf + d;                           // 16 consecutive positions (the last piece may pass O: never named)
                if (d + 16u <= nf) {
                    const u32x4 v = ld16u(D + q);
                    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int k = 0; k < 16; k++) s_ring[(q + (uint32_t)k) & 0xFFFFu] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
                } else {
                    for (uint32_t k = d; k < nf; k++) s_ring[(O - nf + k) & 0xFFFFu] = D[O - nf + k];
                }
            }
            have = r;
        }
        __syncthreads();
        const uint32_t c = i - itembase[r];
        const uint32_t p0 = O + c * SY_PIECE, lim = E - SY_W;
        const uint32_t p1 = p0 + SY_PIECE < lim ? p0 + SY_PIECE : lim;
        sy_resolve_range<512>(D, S, s_ring, O, p0, p1, t);
        __syncthreads();
    }
}

// the block is decoded: k_dec_serial only reports (hb_lz4_dec.hip)
__global__ void k_sy_finish(const RgPlan *rg, const SyPlan *sy, DecPlan *dp, int mark_post) {
    if (!sy->go || sy->fail) return;
    dp->mode = DEC_INDEXED; dp->fail = 0; dp->nbytes = (uint32_t)rg->total;
    if (mark_post) dp->post = 1;
}

// Runs behind k_dec_plan / k_dec_indexed on a block whose index was rebuilt (hb_launch_lz4_region_index): does nothing when that
// index held; decodes into `dst` otherwise.  `work` = the region workspace, `sym_work` = hb_lz4_sym_workspace(cap) bytes.
int hb_launch_lz4_sym_decode(const hb_dec_args &a, uint8_t *dst, uint8_t *sym_work, int mark_post, hipStream_t s) {
    const RgLayout RL = rg_layout(a.cap);
    uint8_t *w = a.work + 256;
    RgPlan *rg = (RgPlan *)(w + RL.plan);
    RgRegion *reg = (RgRegion *)(w + RL.reg);
    DecPlan *dp = (DecPlan *)a.work;
    const SyLayout L = sy_layout(a.cap);
    SyPlan *sy = (SyPlan *)(sym_work + L.plan);
    uint32_t *par = (uint32_t *)(sym_work + L.par);
    uint16_t *S = (uint16_t *)(sym_work + L.sym);
    uint32_t *maps = (uint32_t *)(sym_work + L.maps);
    uint8_t *tails = sym_work + L.tails;
    uint64_t rs; uint32_t nreg;
    rg_regions(a.n, &rs, &nreg);
    const uint32_t nunits = nreg * SY_SUB;
    const uint32_t gmax = sy_max_groups(a.cap);
    uint32_t per = (nunits + gmax - 1) / gmax;
    if (per == 0) per = 1;
    const uint32_t groups = (nunits + per - 1) / per;
    SyUnit *un = (SyUnit *)(sym_work + L.units);
    SyBig *big = (SyBig *)(sym_work + L.big);
    uint32_t *list = (uint32_t *)(sym_work + L.list);
    hb_prof_begin("k_sy_units", s);
    hipLaunchKernelGGL(k_sy_gate, dim3(1), dim3(1), 0, s, rg, dp, sy, groups, per);
    hipLaunchKernelGGL(k_sy_units, dim3((nreg + 3) / 4), dim3(64), 0, s, rg, reg, (const uint2 *)(w + RL.trace), sy, un);
    hipLaunchKernelGGL(k_sy_compact, dim3(1), dim3(1024), 0, s, sy, un, list);
    hb_prof_end(s);
    for (int k = 0; k < SY_ROUNDS; k++) {
        const int last = k + 1 == SY_ROUNDS;
        hb_prof_begin("k_sy_decode", s);
        hipLaunchKernelGGL(k_sy_decode, dim3(nunits < 4096u ? nunits : 4096u), dim3(64), 0, s, a.src, (uint64_t)a.n, un, list, sy, big, dst, S, last);
        hb_prof_end(s);
        if (!last) {
            hb_prof_begin("k_sy_big", s);
            hipLaunchKernelGGL(k_sy_big, dim3(1024), dim3(256), 0, s, a.src, sy, big, dst, S, 0);
            hipLaunchKernelGGL(k_sy_big_reset, dim3(1), dim3(1), 0, s, sy);
            hb_prof_end(s);
        }
    }
    hb_prof_begin("k_sy_compose", s);
    hipLaunchKernelGGL(k_sy_compose, dim3(groups), dim3(1024), 0, s, un, list, sy, dst, S, maps, par);
    hb_prof_end(s);
    hb_prof_begin("k_sy_chain", s);
    hipLaunchKernelGGL(k_sy_chain, dim3(1), dim3(1024), 0, s, sy, maps, par, tails);
    hb_prof_end(s);
    hb_prof_begin("k_sy_tails", s);
    hipLaunchKernelGGL(k_sy_tails, dim3(groups), dim3(1024), 0, s, un, list, sy, dst, S, tails);
    hb_prof_end(s);
    hb_prof_begin("k_sy_resolve", s);
    uint32_t *itembase = (uint32_t *)(sym_work + L.items);
    hipLaunchKernelGGL(k_sy_items, dim3(1), dim3(1024), 0, s, un, list, sy, itembase);
    hipLaunchKernelGGL(k_sy_resolve, dim3(2048), dim3(512), 0, s, un, list, sy, dst, S, itembase);
    hipLaunchKernelGGL(k_sy_finish, dim3(1), dim3(1), 0, s, rg, sy, dp, mark_post);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
