// hb_lz4_region.h — state of the guess-and-verify token discovery (hb_lz4_region.hip) shared with the symbolic decoder of
// foreign blocks (hb_lz4_sym.hip): the regions of the stream, their verified token chain, the window-parallel parser.
#pragma once
#include "hb_lz4.h"
#include "hb_dec_common.h"

#define RG_TRACE    256u          // trace entries per region: RG_DENSE first tokens + one per bucket of the region's stream range
#define RG_DENSE    128u
#define RG_BUCKETS  128u
#define RG_INVALID  0xFFFFFFFFu
#define RG_MAXREG   16384u
#define RG_MINREG   8192u         // smallest region, stream bytes (a region is one wavefront's serial walk: ~30 us per KiB -- small frames want small regions)
#define RG_MINREG_SHORT 4096u     // ... of a stream of at most RG_SHORT bytes (1 - 16 MiB frames: 10-30 % less time; 2048: worse again, and 64 MiB frames lose with 4096)
#define RG_SHORT    (8u << 20)
#define RG_PWIN     4096u         // parse window: every kernel that walks tokens is bound by one wavefront's latency, so its throughput is the number
                                  // of resident waves -- 4 KiB gives k_rg_parse / k_rg_index 32 per CU (8 KiB: 17; index-less 1 GiB decode 5.1 -> 4.6 ms)
#ifndef RG_FIXROUNDS
#define RG_FIXROUNDS 6             // k_rg_settle launches (each iterates to a standstill; idle once settled), re-parses in between (16 until the end of round 3:
                                  // the chains of every frame measured settle in the first or second; an idle pair of launches is ~9 us; what is still moving after six goes to the last launch)
#endif
#define RG_FPARSERS 2             // wavefronts of the last settle launch that parse (LDS: the regions' state takes 128 KiB)
#define RG_FLIST    64            // regions it hands them per hop
#define RG_MAXHOPS  1024          // hops it makes at most
#define RG_WALKCAP   8             // tokens a lane walks serially (3 us each) before it asks for a wave-parallel re-parse (70 ns each)

struct __attribute__((aligned(16))) RgRegion {
    uint32_t b;          // nominal start (stream position)
    uint32_t entry;      // belief: first token of this region (>= b of the next region: the region is empty)
    uint32_t exit;       // first token at / after the next region's b when parsing from `entry`; RG_INVALID: parse failed
    uint32_t outlen;     // output bytes of the tokens in [entry, exit)
    uint32_t entry0, exit0, outlen0, ntrace;   // the parse the trace belongs to
    uint32_t needfull;   // full parse from `entry` pending
    uint32_t pad0;       // from this stream position on the recorded parse (the trace) runs on the region's final chain
    uint64_t opos;       // output position of `entry`
    uint32_t pad1[4];
};
struct RgPlan { uint32_t ok, fail, nreg, rs; uint64_t total; uint32_t pad[10]; };

// the most regions a block that decodes to at most n_out bytes can have (rg_regions below)
static inline size_t rg_max_regions(size_t n_out) {
    const size_t bound = n_out + n_out / 255 + 16;
    size_t nr = bound / RG_MINREG + 2;
    const size_t ns = (bound < RG_SHORT ? bound : RG_SHORT) / RG_MINREG_SHORT + 2;
    if (ns > nr) nr = ns;
    return nr > RG_MAXREG ? RG_MAXREG : nr;
}
// bytes of token store for any block that decodes to at most n_out bytes
static inline size_t rg_tok_bytes(size_t n_out) {
    const size_t bound = n_out + n_out / 255 + 16;
    return (bound / 3 + 80 * rg_max_regions(n_out) + 4096) * 8;
}
// ---- the token store (round 3).  Every stage behind the first parse used to walk the stream again; with room for it (the larger, "foreign"
// workspace) k_rg_parse keeps what it parsed: {position, literal length | match length << 16} per token, RG region r's at tok[r * tokcap ...], their
// number in RgRegion.pad1[0] (RG_INVALID: none / more than fit).  Like the bucket records they lie on the final chain from RgRegion.pad0 on, if the
// recorded parse ends where the chain does (exit0 == exit).  A token whose lengths do not fit 16 bits is stored as {position, RG_INVALID}.
static inline uint32_t rg_tokcap(uint64_t rs) { return (uint32_t)(rs / 3u + 64u); }     // (a sequence is at least 3 bytes)
struct RgLayout { size_t plan, reg, pmax, trace, total; };
// sized for the regions a block that decodes to at most n_out bytes can have (a stream is never much longer than its output)
static inline RgLayout rg_layout(size_t n_out) {
    RgLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
    const size_t nr = rg_max_regions(n_out);
    L.plan = take(sizeof(RgPlan));
    L.reg = take(nr * sizeof(RgRegion));
    L.pmax = take(nr * 4);
    L.trace = take(nr * RG_TRACE * sizeof(uint2));
    L.total = o;
    return L;
}

static inline void rg_regions(size_t n, uint64_t *rs_out, uint32_t *nreg_out);
// ---- batches (hb_decompress_frames_batch_dev): the same kernels over MANY blocks in one set of launches.  Every kernel body takes its block
// index and grid size as arguments; the `_b` launch reads its block's arguments from a job record (blockIdx.y = job) -- nothing else differs,
// so a frame's index comes out exactly as the one-frame path builds it.  Blocks of a batch have at most RGB_MAXR regions (k_rg_settle keeps
// entry + exit of every region in LDS: 128 KiB for the one-frame path's 16384 regions, 8 KiB here).
#define RGB_MAXR 1024u
struct RgJob {
    const uint8_t *src; uint64_t n_src, cap;
    RgPlan *plan; RgRegion *reg; uint2 *traces; uint32_t *pmax; uint2 *tok; uint8_t *idx;
    uint32_t tokcap, nreg, rs, pad;
};

struct RgBatchLayout { size_t plan, reg, pmax, trace, tok, total; };
// scratch of one job: sized by the block's own stream length (its regions), not by what a block of its decoded size could have
// regions of a block inside a batch: at least rs_min bytes each (a batch of a thousand frames has parallelism to spare: fewer, longer regions cost
// less start-up -- every region's parse begins with a guess and a stretch of tokens until it falls onto the chain)
static inline void rg_regions_min(size_t n, uint64_t rs_min, uint64_t *rs_out, uint32_t *nreg_out) {
    rg_regions(n, rs_out, nreg_out);
    if (*rs_out < rs_min) { *rs_out = (rs_min + 15) & ~(uint64_t)15; *nreg_out = (uint32_t)((n + *rs_out - 1) / *rs_out); }
}
static inline RgBatchLayout rg_batch_layout(size_t n_src, uint64_t rs_min = 0) {
    RgBatchLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
    uint64_t rs; uint32_t nreg;
    rg_regions_min(n_src, rs_min, &rs, &nreg);
    L.plan = take(sizeof(RgPlan));
    L.reg = take((size_t)(nreg + 1) * sizeof(RgRegion));
    L.pmax = take((size_t)RGB_MAXR * 4);                     // (k_rg_index_tok clears done[r] for every workgroup of the launch: the batch's largest region count)
    L.trace = take((size_t)nreg * RG_TRACE * sizeof(uint2));
    L.tok = take((size_t)nreg * rg_tokcap(rs) * sizeof(uint2));
    L.total = o;
    return L;
}
bool hb_lz4_region_batch_wanted(size_t n_src, size_t cap);
size_t hb_lz4_region_batch_bytes(size_t n_src, size_t cap);
void hb_lz4_region_batch_job(uint8_t *w, uint8_t *idx, const uint8_t *src, size_t n_src, size_t cap, RgJob *j, uint64_t rs_min = 0);
int hb_launch_lz4_region_index_batch(const RgJob *d_jobs, int njobs, uint32_t max_nreg, hipStream_t s);

#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))

// ---- the same discovery for Snappy blocks (hb_snappy.hip: blocks that come without a unit index, i.e. written by any other encoder) ----
// The kernels that parse are templated on the codec: elements instead of sequences.  CODEC of rg_parse_region / k_rg_settle_body / k_rg_fix_body.
enum { RG_LZ4 = 0, RG_SNAPPY = 1 };
// one element, all values wave-uniform: `p[k]` = stream byte k of a buffer holding `avail` bytes from the element's tag.  false: the header runs off the buffer.
struct SnElem { uint32_t kind, hdr; uint64_t lit; uint32_t mlen; uint64_t off; };
__device__ __forceinline__ bool sn_parse_uniform(const uint8_t *p, uint64_t avail, SnElem &e) {
    if (avail < 1) return false;
    const uint32_t t = __builtin_amdgcn_readfirstlane((uint32_t)p[0]);
    e.kind = t & 3u; e.lit = 0; e.mlen = 0; e.off = 0; e.hdr = 1;
    const uint32_t x = t >> 2;
    auto byte = [&](uint32_t i) { return (uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)p[i]); };
    if (e.kind == 0u) {
        if (x < 60u) e.lit = x + 1u;
        else {
            const uint32_t nb = x - 59u;                          // 1..4 length bytes
            if (avail < 1u + nb) return false;
            uint64_t v = 0;
            for (uint32_t i = 0; i < nb; i++) v |= byte(1u + i) << (8u * i);
            e.lit = v + 1u; e.hdr = 1u + nb;
        }
    } else if (e.kind == 1u) { if (avail < 2) return false; e.mlen = 4u + (x & 7u); e.off = ((uint64_t)(t >> 5) << 8) | byte(1); e.hdr = 2; }
    else if (e.kind == 2u) { if (avail < 3) return false; e.mlen = 1u + x; e.off = byte(1) | (byte(2) << 8); e.hdr = 3; }
    else { if (avail < 5) return false; e.mlen = 1u + x; e.off = byte(1) | (byte(2) << 8) | (byte(3) << 16) | (byte(4) << 24); e.hdr = 5; }
    return true;
}
// one token (LZ4 sequence / Snappy element) at stream position p, by one lane: p moves behind it, cum grows by its output bytes; false: it runs off
// the stream or is malformed (the caller then asks for a wave-parallel parse, which decides)
template <int CODEC>
__device__ __forceinline__ bool rg_step_serial(const uint8_t *__restrict__ src, const uint64_t n_src, uint64_t &p, uint64_t &cum) {
    if constexpr (CODEC == RG_SNAPPY) {
        if (p >= n_src) return false;
        // (the tag and the next three bytes in one read where the stream has them: one memory round trip per element for the walks that go lane by lane)
        const bool wide = n_src - p >= 4u;
        const uint32_t w = wide ? ld4u(src + p) : (uint32_t)src[p];
        const uint32_t t = w & 255u, kind = t & 3u, x = t >> 2;
        uint64_t hdr = 1, lit = 0, ml = 0;
        if (kind == 0u) {
            if (x < 60u) lit = x + 1u;
            else {
                const uint32_t nb = x - 59u;
                if (n_src - p < 1u + nb) return false;
                uint64_t v = 0;
                if (wide && nb <= 3u) v = (w >> 8) & (nb == 1u ? 0xFFu : nb == 2u ? 0xFFFFu : 0xFFFFFFu);
                else for (uint32_t i = 0; i < nb; i++) v |= (uint64_t)src[p + 1u + i] << (8u * i);
                lit = v + 1u; hdr = 1u + nb;
                if (lit > 65536u) return false;                          // (as rg_parse_region: no chain through literals of more than 64 KiB)
            }
        } else if (kind == 1u) { hdr = 2; ml = 4u + (x & 7u); }
        else if (kind == 2u) { hdr = 3; ml = 1u + x; }
        else return false;                                           // (copy-4: as rg_parse_region, no chain through one)
        if (n_src - p < hdr || lit > n_src - p - hdr) return false;
        p += hdr + lit; cum += lit + ml;
        return true;
    } else {
        const uint32_t tok = src[p];
        uint64_t q = p + 1, ll = tok >> 4;
        bool bad = false;
        if (ll == 15u) { int kk = 0; for (;; kk++) { if (q >= n_src || kk > 2048) { bad = true; break; } const uint32_t x = src[q++]; ll += x; if (x != 255u) break; } }
        if (bad || ll > n_src - q) return false;
        q += ll;
        uint64_t ml = 0;
        if (q != n_src) {
            if (n_src - q < 2) return false;
            q += 2; ml = (tok & 15u) + 4u;
            if ((tok & 15u) == 15u) { int kk = 0; for (;; kk++) { if (q >= n_src || kk > 2048) { bad = true; break; } const uint32_t x = src[q++]; ml += x; if (x != 255u) break; } }
            if (bad) return false;
        }
        cum += ll + ml;
        p = q;
        return true;
    }
}

// Window-parallel element parser for the passes that copy nothing (rg_fill's counterpart): 64 lanes parse 64 stream bytes "as if an element started
// at my byte", the chain is followed on the scalar side (Snappy windows hold up to 32 elements, but a parse is not the decoder's hot loop).
// Queue entry: {position (window-relative), output bytes of the element}.  An element whose header does not lie inside the window, or a literal whose
// length takes two to four bytes (more than 256 bytes), ends the walk (-> sn_parse_uniform).  si may come back beyond lim: a literal's bytes need not be staged.
// guess: the caller's parse is only a guess (a region's first parse): an element that cannot be on a chain -- a copy-4, a literal whose length takes three
// or four bytes -- is then a hop of ONE byte, queued with bit 31 of its position set and no output (the caller restarts its record behind it); else such an
// element ends the walk like a long literal does.
__device__ __forceinline__ bool sn_rg_fill(const uint8_t *s_in, const uint32_t sh, const uint32_t lim, uint32_t &si, uint32_t &nq, uint2 *s_tq, const int lane, const bool guess = false) {
    bool stop = false;
    while (nq < 64u && !stop) {
        if (si >= lim) { stop = true; break; }
        const uint32_t base = si, p = base + (uint32_t)lane;
        const uint32_t w = dec_read4(s_in, sh + p);
        const uint32_t t = w & 255u, b1 = (w >> 8) & 255u, kind = t & 3u, x = t >> 2;
        bool cplx = p + 5u > lim, hop = false;
        uint32_t olen, hdr, lit = 0;
        if (kind == 0u) {
            hdr = 1u; lit = x + 1u;
            if (x == 60u) { lit = b1 + 1u; hdr = 2u; }
            else if (x == 61u) cplx = true;                      // (a literal of more than 256 bytes: one at a time -- the caller has rules about those)
            else if (x > 61u) { if (guess) hop = true; else cplx = true; }
            olen = lit;
        } else if (kind == 1u) { olen = 4u + (x & 7u); hdr = 2u; }
        else { olen = 1u + x; hdr = 3u; if (kind == 3u) { if (guess) hop = true; else cplx = true; } }     // (copy-4: the caller has rules about those too)
        if (hop) { hdr = 1u; lit = 0u; olen = 0u; }
        const uint32_t nrel = cplx ? 64u : (uint32_t)lane + hdr + lit;
        const unsigned long long cmask = hb_ballot(cplx);
        unsigned long long tmask = 0;
        uint32_t cur;
        {
            const uint32_t succ = nrel < 64u ? nrel : (uint32_t)lane;
            uint32_t j = 0, lastj;
            for (;;) {
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
            const unsigned long long cm = tmask & cmask;
            if (cm) { tmask &= ~cm; cur = base + (uint32_t)__builtin_ctzll(cm); stop = true; }
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tmask, 0u));
        if ((tmask >> lane) & 1ull) { uint2 e; e.x = p | (hop ? 0x80000000u : 0u); e.y = olen; s_tq[nq + rank] = e; }
        nq += (uint32_t)__builtin_popcountll(tmask);
        si = cur;
        if (si > lim) break;                                     // (the rest of a literal lies behind the window: the caller moves it)
    }
    return stop;
}
// FILL: 64 lanes parse 64 stream bytes "as if an element started at my byte"; the real chain is walked on the scalar side.
// Elements go to s_tq as {lsrc | lit << 13 | mlen << 22, offset | pos << 16}.  Returns true when it stopped at the end of the
// slice or at an element the lane-parallel path does not take (literal longer than 511 bytes, 4-byte offset, too close to lim).
__device__ __forceinline__ bool sn_fill(const uint8_t *s_in, const uint32_t sh, const uint32_t lim, uint32_t &si, uint32_t &nq,
                                        uint2 *s_tq, const int lane) {
    bool stop = false;
    while (nq < 64u && !stop) {
        if (si == lim) { stop = true; break; }
        const uint32_t base = si, p = base + (uint32_t)lane;
        const uint32_t w = dec_read4(s_in, sh + p);
        const uint32_t t = w & 255u, b1 = (w >> 8) & 255u, b2 = (w >> 16) & 255u, kind = t & 3u, x = t >> 2;
        bool cplx = p >= lim;
        uint32_t lit = 0, mlen = 0, offv = 0, hdr = 1;
        if (kind == 0u) {
            if (x < 60u) lit = x + 1u;
            else if (x == 60u) { lit = b1 + 1u; hdr = 2u; }
            else if (x == 61u) { lit = (b1 | (b2 << 8)) + 1u; hdr = 3u; }
            else cplx = true;
            if (lit > 511u) cplx = true;
        } else if (kind == 1u) { mlen = 4u + (x & 7u); offv = ((t >> 5) << 8) | b1; hdr = 2u; }
        else if (kind == 2u) { mlen = 1u + x; offv = b1 | (b2 << 8); hdr = 3u; }
        else cplx = true;
        const uint32_t lsrc = p + hdr, nxt = lsrc + lit;
        if (nxt > lim) cplx = true;
        const unsigned long long cmask = hb_ballot(cplx);
        unsigned long long tmask = 0;
        uint32_t cur;
        {
            const uint32_t nrel = cplx ? 64u : nxt - base;
            const uint32_t succ = nrel < 64u ? nrel : (uint32_t)lane;
            uint32_t j = 0, lastj;
            for (;;) {
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
            const unsigned long long cm = tmask & cmask;
            if (cm) { tmask &= ~cm; cur = base + (uint32_t)__builtin_ctzll(cm); stop = true; }
        }
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


// The verified element chain [start, exitp) of a Snappy block for a decoder, with rg_walk's callbacks (hb_sym_decode.h: the unit decoder of the
// symbolic pass takes its tokens from either): an element is a token with literals and no match (mlen == 0; `off` is 1 then, a value every check
// passes) or with a match and no literals.  What sn_fill does not take -- literals of more than 511 bytes or not staged in full -- comes through
// `single`; a copy from more than 65535 bytes back (4-byte offset: no 64 KiB-block encoder writes one) ends the walk with false.
template <uint32_t PWIN = RG_PWIN, class Batch, class Single>
__device__ __forceinline__ bool sn_walk(const uint8_t *__restrict__ src, const uint64_t n_src, const uint32_t start, const uint32_t exitp,
                                        uint8_t *s_win /* PWIN + 128 */, uint2 *s_tq /* DTQ */, const int lane, Batch &&batch, Single &&single) {
    uint64_t si = start, wpos = 0;
    uint32_t wlen = 0, wsh = 0, nq = 0;
    auto refill = [&](uint64_t at) __attribute__((always_inline)) {
        const uint8_t *g = src + at;
        wsh = (uint32_t)((uintptr_t)g & 15u);
        const uint64_t left = n_src - at;
        wlen = (uint32_t)(left < (uint64_t)(PWIN - 16u) ? left : (uint64_t)(PWIN - 16u));
        const u32x4 *ga = (const u32x4 *)(g - wsh);
        const uint32_t nv = (wsh + wlen + 15u) >> 4;
        wave_sync();
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
        wpos = at;
        wave_sync();
    };
    auto drain = [&](const bool all) __attribute__((always_inline)) -> bool {
        while (nq >= 64u || (all && nq > 0u)) {
            const uint32_t cntb = nq < 64u ? nq : 64u;
            const uint2 e = s_tq[lane];                                    // {lsrc | lit << 13 | mlen << 22, offset | pos << 16}, window-relative (sn_fill)
            const uint32_t lw = e.x & 0x1FFFu, lit = (e.x >> 13) & 0x1FFu, tw = e.y >> 16;
            uint32_t mlen = e.x >> 22;
            const uint32_t off = mlen ? (e.y & 0xFFFFu) : 1u;
            // A Snappy copy ends at 64 bytes, so a run of KiB is dozens of copies at the same distance, each reading what the one before wrote: 64
            // dependency rounds per batch for the decoder behind this walk (measured, a float ramp at ratio 0.05: pass A 9.1 ms).  Copies that follow
            // each other at the same distance ARE one copy (out[p + k] = out[p + k - off] for k over both): the first of such a stretch takes the
            // whole length, the others become empty tokens.
            {
                const bool cp = (uint32_t)lane < cntb && mlen != 0u;                    // (a copy element: no literals)
                const uint32_t poff = wave_shr1(off, 0u);
                const bool pcp = wave_shr1((uint32_t)cp, 0u) != 0u;
                const bool cont = cp && pcp && lane != 0 && off == poff;            // continues the copy in the lane before
                const unsigned long long heads = ~hb_ballot(cont);                  // lanes that start a token of their own
                const uint32_t incl = wave_incl_scan_dpp(cp ? mlen : 0u);
                const unsigned long long above = lane < 63 ? heads >> (lane + 1) : 0ull;
                const uint32_t endl = above ? (uint32_t)lane + (uint32_t)__builtin_ctzll(above) : 63u;      // last lane of my stretch
                const uint32_t incl_end = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(endl << 2), (int)incl);
                if (cont) mlen = 0u;
                else if (cp) mlen = incl_end - (incl - mlen);
            }
            const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];
            if (!batch(cntb, (uint32_t)wpos + tw, (uint32_t)wpos + lw, lit, mlen, off, wsh + lw)) return false;
            nq -= cntb;
            if ((uint32_t)lane < nq) s_tq[lane] = rest;
        }
        return true;
    };
    wave_sync();
    while (si < exitp) {
        if (si < wpos || si - wpos + PWIN / 4u > wlen) { if (si != wpos || wlen == 0) { if (!drain(true)) return false; refill(si); } }
        uint32_t rel = (uint32_t)(si - wpos);
        const uint64_t tolim = (uint64_t)exitp - wpos;
        const uint32_t lim = (uint32_t)(tolim < (uint64_t)wlen ? tolim : (uint64_t)wlen);
        const bool stop = sn_fill(s_win, wsh, lim, rel, nq, s_tq, lane);
        if (!drain(false)) return false;
        const bool moved = (wpos + rel) != si;
        si = wpos + rel;
        if (moved && !stop) continue;
        if (!drain(true)) return false;
        if (si >= exitp) break;
        if (moved && si - wpos + PWIN / 4u > wlen && wpos + wlen < n_src && wpos + wlen < exitp) continue;
        // ---- one element the slow way ----
        if (si < wpos || si + 8u > wpos + wlen) refill(si);
        rel = (uint32_t)(si - wpos);
        SnElem e;
        if (!sn_parse_uniform(s_win + wsh + rel, n_src - si, e)) return false;
        const uint64_t ls = si + e.hdr;
        if (e.lit > n_src - ls || e.lit > 0xFFFFFFF0ull || e.off > 65535u) return false;
        if (!single((uint32_t)si, (uint32_t)ls, (uint32_t)e.lit, e.mlen, e.kind == 0u ? 1u : (uint32_t)e.off, 0u)) return false;
        si = ls + e.lit;
    }
    return drain(true);
}
// the chain of a Snappy block (hb_lz4_region.hip): regions, parses, beliefs, verification -- RgPlan.ok / RgRegion.{entry, exit, opos} as for LZ4.
// w: rg_layout(cap).total bytes; entry0: device pointer to the stream position of the first element (SnPlan.hdr)
int hb_launch_snappy_region_chain(const uint8_t *src, size_t n, size_t cap, uint8_t *w, const uint32_t *entry0, hipStream_t s);

// Window-parallel token parser for the passes that copy nothing.  Like dec_fill (hb_dec_common.h), but every lane also sums
// multi-byte length extensions itself (up to 24 bytes each, i.e. lengths up to 6 KiB): a stream made of long runs (20-byte
// sequences, each with a 16-byte match extension) would otherwise go through the one-token path sequence by sequence.
// Queue entry: { tokpos | nbl << 16, lit | mlen << 16 }, positions relative to the window; literal bytes start at tokpos + 1 + nbl.
__device__ __forceinline__ bool rg_fill(const uint8_t *s_in, const uint32_t sh, const uint32_t lim, uint32_t &si, uint32_t &nq, uint2 *s_tq, const int lane) {
    bool stop = false;
    uint32_t last_ntok = DEC_BPERM_MIN;                   // (tokens of the previous window: picks the way the chain is followed; a call starts with the doubling)
    while (nq < 64u && !stop) {
        if (si == lim) { stop = true; break; }
        const uint32_t base = si, p = base + (uint32_t)lane;
        // token + the first three bytes of a literal-length extension in one read; offset + the first two bytes of a match-length
        // extension in a second one: lengths up to 15 + 3 * 255 / 19 + 2 * 255 take no further LDS round trip (the staged window has
        // slack behind `lim`; what is read there is only used when the position checks below pass)
        const uint32_t w = dec_read4(s_in, sh + p);
        const uint32_t t = w & 255u;
        bool cplx = p + 4u > lim;
        uint32_t lit = t >> 4, q = p + 1u;
        // a longer extension, four bytes per step (a lane that sits inside a run of 0xFF -- every byte of a long extension looks like
        // the start of another one -- gives up after 24 bytes instead of crawling through it)
        auto ext = [&](uint32_t &len, uint32_t &at) __attribute__((always_inline)) {
            bool open = true;
            for (int k = 0; k < 6 && open; k++) {
                if (at + 4u > lim) break;
                const uint32_t w = dec_read4(s_in, sh + at);
                if (w == 0xFFFFFFFFu) { len += 1020u; at += 4u; }
                else {
                    const uint32_t nff = (uint32_t)__builtin_ctz(~w) >> 3;
                    len += 255u * nff + ((w >> (8u * nff)) & 255u);
                    at += nff + 1u;
                    open = false;
                }
            }
            return !open;
        };
        bool longl = false;
        if (lit == 15u) {
            const uint32_t e = ~(w >> 8) & 0xFFFFFFu;                     // a zero byte where the extension byte is 255
            if (e) {
                const uint32_t nff = (uint32_t)__builtin_ctz(e) >> 3;     // extension bytes that are 255
                lit = 15u + 255u * nff + ((w >> (8u * (nff + 1u))) & 255u);
                q = p + 2u + nff;
            } else { lit = 15u + 765u; q = p + 4u; longl = true; }
        }
        if (longl && !cplx && !ext(lit, q)) cplx = true;
        const uint32_t nbl = q - p - 1u, offpos = q + lit;
        uint32_t mlen = 4u + (t & 15u), q2 = offpos + 2u;
        if (cplx || offpos + 4u > lim) cplx = true;              // literal-only tail, or too close to the edge
        else if ((t & 15u) == 15u) {
            const uint32_t x = dec_read4(s_in, sh + offpos);
            const uint32_t e1 = (x >> 16) & 255u, e2 = x >> 24;
            if (e1 != 255u) { mlen = 19u + e1; q2 = offpos + 3u; }
            else if (e2 != 255u) { mlen = 19u + 255u + e2; q2 = offpos + 4u; }
            else { mlen = 19u + 510u; q2 = offpos + 4u; if (!ext(mlen, q2)) cplx = true; }
        }
        if (lit > 0xFFFFu || mlen > 0xFFFFu) cplx = true;
        const uint32_t nxt = q2;
        const unsigned long long cmask = hb_ballot(cplx);
        unsigned long long tmask = 0;
        uint32_t cur;
        const uint32_t nrel = cplx ? 64u : nxt - base;
        const uint32_t succ = nrel < 64u ? nrel : (uint32_t)lane;
#if DEC_BPERM_WALK
        if (last_ntok >= DEC_BPERM_MIN) {
            // the chain of a token-dense window by pointer doubling, as dec_fill_lean follows it (hb_dec_common.h): lane k ends up holding the k-th
            // token's lane and fetches that lane's entry -- the queue is written compacted, no scalar walk, no rank computation
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
            const uint32_t left = wave_shr1(c, 0xFFFFFFFFu);
            const unsigned long long distinct = hb_ballot(c != left) | 0xFFFFFFFF00000000ull;
            uint32_t ntok = (uint32_t)__builtin_ctzll(~distinct | (1ull << 32));
            const uint32_t lastj = __builtin_amdgcn_readlane(c, (int)ntok - 1);
            cur = base + __builtin_amdgcn_readlane(nrel, (int)lastj);
            if ((cmask >> lastj) & 1ull) { ntok--; cur = base + lastj; stop = true; }
            uint2 e;
            e.x = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)(p | (nbl << 16)));
            e.y = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(c << 2), (int)(lit | (mlen << 16)));
            s_tq[(uint32_t)lane < ntok ? nq + (uint32_t)lane : (uint32_t)(DTQ - 1)] = e;
            nq += ntok;
            si = cur;
            last_ntok = ntok;
            continue;
        }
#endif
        {
            uint32_t j = 0, lastj;
            for (;;) {
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
            const unsigned long long cm = tmask & cmask;
            if (cm) { tmask &= ~cm; cur = base + (uint32_t)__builtin_ctzll(cm); stop = true; }
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(tmask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)tmask, 0u));
        if ((tmask >> lane) & 1ull) { uint2 e; e.x = p | (nbl << 16); e.y = lit | (mlen << 16); s_tq[nq + rank] = e; }
        last_ntok = (uint32_t)__builtin_popcountll(tmask);
        nq += last_ntok;
        si = cur;
    }
    return stop;
}


// regions of an n-byte stream: at most RG_MAXREG, at least RG_MINREG bytes each.  A region is one wavefront's serial walk, so a launch
// lasts about as long as a region is long: as many regions as there may be (a 64 MiB frame then decodes in 1.0 ms instead of 2.0,
// 256 MiB in 1.7 instead of 2.4).
static inline void rg_regions(size_t n, uint64_t *rs_out, uint32_t *nreg_out) {
    uint64_t rs = (n + RG_MAXREG - 1) / RG_MAXREG;
    const uint64_t lo = n <= RG_SHORT ? RG_MINREG_SHORT : RG_MINREG;
    if (rs < lo) rs = lo;
    rs = (rs + 15) & ~(uint64_t)15;
    *rs_out = rs; *nreg_out = (uint32_t)((n + rs - 1) / rs);
}

// Walks the tokens of [start, exitp) of a VERIFIED chain (k_rg_scan passed: every length extension ends inside the stream, exitp is
// a token start or the end of the block), one wavefront.  Tokens the window-parallel parser takes come 64 at a time, one per lane:
//   batch(cnt, tp, ls, lit, mlen, off, lp)  lane < cnt holds a token at stream position tp with `lit` literals at ls (staged at
//                                        s_win[lp]), then a match of mlen >= 4 bytes at distance off; returns false (wave-uniform) to stop the walk
// the rest one at a time with all arguments wave-uniform:
//   single(tp, ls, lit, mlen, off, tok)  mlen == 0: the block's final, literal-only sequence
// Returns false when a callback stopped the walk or the stream turned out malformed after all.
// PWIN: bytes of the staged stream window (s_win holds PWIN + 128); it moves once less than a quarter of it is left.
template <uint32_t PWIN = RG_PWIN, class Batch, class Single>
__device__ __forceinline__ bool rg_walk(const uint8_t *__restrict__ src, const uint64_t n_src, const uint32_t start, const uint32_t exitp,
                                        uint8_t *s_win /* PWIN + 128 */, uint2 *s_tq /* DTQ */, const int lane, Batch &&batch, Single &&single) {
    uint64_t si = start, wpos = 0;
    uint32_t wlen = 0, wsh = 0, nq = 0;
    auto refill = [&](uint64_t at) __attribute__((always_inline)) {
        const uint8_t *g = src + at;
        wsh = (uint32_t)((uintptr_t)g & 15u);
        const uint64_t left = n_src - at;
        wlen = (uint32_t)(left < (uint64_t)(PWIN - 16u) ? left : (uint64_t)(PWIN - 16u));
        const u32x4 *ga = (const u32x4 *)(g - wsh);
        const uint32_t nv = (wsh + wlen + 15u) >> 4;
        wave_sync();
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
        wpos = at;
        wave_sync();
    };
    // queued tokens go to `batch` 64 at a time; fewer only when the order demands it (before the window moves: queue entries are
    // window-relative; before a token takes the slow path; at the end)
    auto drain = [&](const bool all) __attribute__((always_inline)) -> bool {
        while (nq >= 64u || (all && nq > 0u)) {
            const uint32_t cntb = nq < 64u ? nq : 64u;
            const uint2 e = s_tq[lane];
            const uint32_t lit = e.y & 0xFFFFu, mlen = e.y >> 16;
            const uint32_t tw = e.x & 0xFFFFu, lw = tw + 1u + (e.x >> 16);       // token / first literal, window-relative
            uint32_t off = 0;
            if ((uint32_t)lane < cntb) off = (uint32_t)s_win[wsh + lw + lit] | ((uint32_t)s_win[wsh + lw + lit + 1u] << 8);
            const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];
            if (!batch(cntb, (uint32_t)wpos + tw, (uint32_t)wpos + lw, lit, mlen, off, wsh + lw)) return false;
            nq -= cntb;
            if ((uint32_t)lane < nq) s_tq[lane] = rest;
        }
        return true;
    };
    wave_sync();
    while (si < exitp) {
        if (si < wpos || si - wpos + PWIN / 4u > wlen) { if (si != wpos || wlen == 0) { if (!drain(true)) return false; refill(si); } }
        uint32_t rel = (uint32_t)(si - wpos);
        const uint64_t tolim = (uint64_t)exitp - wpos;
        const uint32_t lim = (uint32_t)(tolim < (uint64_t)wlen ? tolim : (uint64_t)wlen);
        const bool stop = rg_fill(s_win, wsh, lim, rel, nq, s_tq, lane);
        if (!drain(false)) return false;
        const bool moved = (wpos + rel) != si;
        si = wpos + rel;
        if (moved && !stop) continue;
        if (!drain(true)) return false;
        if (si >= exitp) break;
        if (moved && si - wpos + PWIN / 4u > wlen && wpos + wlen < n_src && wpos + wlen < exitp) continue;
        // ---- one token the slow way: runs of any length ----
        if (si < wpos || si >= wpos + wlen) refill(si);
        rel = (uint32_t)(si - wpos);
        const uint32_t tp = (uint32_t)si;
        const uint32_t tok = RFL((uint32_t)s_win[wsh + rel]);
        rel++;
        uint32_t ll = tok >> 4;
        {
            const uint64_t span = n_src - wpos;
            if (ll == 15u && !dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel, ll, lane)) return false;
        }
        uint64_t p = wpos + rel;
        const uint32_t ls = (uint32_t)p;
        if ((uint64_t)ll > n_src - p) return false;
        p += ll;
        uint32_t ml = 0, off = 0;
        if (p != n_src) {
            if (n_src - p < 2) return false;
            off = RFL((uint32_t)src[p] | ((uint32_t)src[p + 1] << 8));
            p += 2;
            ml = (tok & 15u) + 4u;
            if ((tok & 15u) == 15u) {
                if (p < wpos || p - wpos + 64u > wlen) refill(p);
                uint32_t rel2 = (uint32_t)(p - wpos);
                const uint64_t span = n_src - wpos;
                if (!dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel2, ml, lane)) return false;
                p = wpos + rel2;
            }
        }
        if (!single(tp, ls, ll, ml, off, tok)) return false;
        si = p;
    }
    return drain(true);
}

// ---- the walk fed from the token store (rg_tokcap above): the tokens tk[k0 .. k1) of ONE region, all on the verified chain.  Nothing is parsed:
// the stream window is staged for the literals and the offsets, 64 stored tokens are read per step (the next 64 while this batch is decoded) and
// handed to `batch` as rg_walk hands them over.  At a token that does not fit the window, carries 32-bit lengths or is the block's literal-only end
// it stops and says so (k0 = that token: the caller has rg_walk parse exactly that one and comes back).  Returns 0: a callback stopped the walk,
// 1: through, 2: stopped at token k0.  Used by pass A of the symbolic decoder where nearly every region is token-dense (k_sy_gate).
struct RgTokStore { const RgRegion *reg; const uint2 *tok; uint32_t tokcap, rs, nreg; };
// first index in tk[0 .. nt) whose position is >= pos (positions ascend); wave-uniform
__device__ __forceinline__ uint32_t rg_tok_lower(const uint2 *__restrict__ tk, const uint32_t nt, const uint32_t pos, const int lane) {
    uint32_t lo = 0, hi = nt;                                           // the answer is in [lo, hi]
    while (hi - lo > 64u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t k = lo + (uint32_t)lane * step;
        const bool ge = k >= hi || tk[k].x >= pos;
        const unsigned long long m = hb_ballot(ge);
        const uint32_t j = m ? (uint32_t)__builtin_ctzll(m) : 64u;      // first probe that is not below pos (lane 0 probes lo)
        if (j == 0u) return lo;
        const uint32_t nhi = lo + j * step < hi ? lo + j * step : hi;
        lo = lo + (j - 1u) * step;
        hi = nhi;
    }
    const uint32_t k = lo + (uint32_t)lane;
    const unsigned long long m = hb_ballot(k >= hi || tk[k].x >= pos);
    return m ? lo + (uint32_t)__builtin_ctzll(m) : hi;
}
template <uint32_t PWIN = RG_PWIN, class Batch>
__device__ __forceinline__ int rg_walk_tok(const uint8_t *__restrict__ src, const uint64_t n_src, const uint2 *__restrict__ tk, uint32_t &k0, const uint32_t k1,
                                           uint8_t *s_win /* PWIN + 128 */, const int lane, Batch &&batch) {
    uint64_t wpos = 0;
    uint32_t wlen = 0, wsh = 0;
    wave_sync();
    uint2 nx; nx.x = 0; nx.y = 0;
    uint32_t nxk = RG_INVALID;                                          // nx holds tk[nxk + lane]
    while (k0 < k1) {
        const uint32_t k = k0 + (uint32_t)lane;
        uint2 t; t.x = 0; t.y = 0;
        const bool have = k < k1;
        if (nxk == k0) t = nx; else if (have) t = tk[k];
        nxk = k0 + 64u;
        nx.x = 0; nx.y = 0;
        if (nxk + (uint32_t)lane < k1) nx = tk[nxk + (uint32_t)lane];
        const uint32_t tp = t.x, lit = t.y & 0xFFFFu, mlen = t.y >> 16;
        const uint32_t ls = tp + 1u + (lit >= 15u ? (((lit - 15u) * 0x8081u) >> 23) + 1u : 0u);     // (x / 255 for x < 65536)
        const uint32_t tp0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)tp);
        if (tp0 < wpos || tp0 - wpos + PWIN / 4u > wlen) {              // move the window (as rg_walk does: once less than a quarter is left)
            const uint8_t *g = src + tp0;
            wsh = (uint32_t)((uintptr_t)g & 15u);
            const uint64_t left = n_src - tp0;
            wlen = (uint32_t)(left < (uint64_t)(PWIN - 16u) ? left : (uint64_t)(PWIN - 16u));
            const u32x4 *ga = (const u32x4 *)(g - wsh);
            const uint32_t nv = (wsh + wlen + 15u) >> 4;
            wave_sync();
            for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
            wpos = tp0;
            wave_sync();
        }
        // leading tokens that lie in the window whole (token, literals, offset) and are ordinary
        const bool fits = have && t.y != RG_INVALID && mlen != 0u && (uint64_t)ls + lit + 2u <= wpos + wlen;
        const unsigned long long nm = hb_ballot(!fits);
        const uint32_t cnt = nm ? (uint32_t)__builtin_ctzll(nm) : 64u;
        if (cnt == 0u) return 2;                                        // this token the slow way (any size; the block's final sequence)
        const uint32_t lw = (uint32_t)(ls - wpos);                      // first literal, window-relative
        uint32_t off = 0;
        if ((uint32_t)lane < cnt) off = (uint32_t)s_win[wsh + lw + lit] | ((uint32_t)s_win[wsh + lw + lit + 1u] << 8);
        if (!batch(cnt, tp, ls, lit, mlen, off, wsh + lw)) return 0;
        k0 += cnt;
    }
    return 1;
}
