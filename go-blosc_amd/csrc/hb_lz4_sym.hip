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
//   units   k_sy_units / k_sy_compact: one unit per region (two light neighbours share one); a region with more than 256 KiB of output is cut into 8 at tokens the
//           discovery has on record (one wavefront copying 14 MiB of KiB-long matches is what everybody else would wait for).
//   pass A  k_sy_decode    one wavefront per unit (the unit decoder of hb_sym_decode.h): tokens 64 at a time, one per lane; the last
//                          4 KiB of output live in LDS, matches into it run LDS to LDS in dependency rounds, older sources are
//                          fetched from HBM all lanes at once; values go to the output buffer, references to a u16 side array
//                          (0 = "is a value").  Literal runs / matches of 256 KiB and more are posted and copied by the whole chip
//                          (k_sy_big) between two launches.
//   pass B  the 65535 bytes in front of a unit are the tail of its predecessors, themselves symbolic: the tail of unit r as a
//           function of the tail in front of it is a MAP of 65535 entries {value | reference}, maps compose associatively, so the
//           tails are resolved by a scan over the units: k_sy_compose (per group of units: the composed map, sequential inside
//           the group, 256 groups in parallel), k_sy_scan + k_sy_front (an inclusive scan over the group maps, one launch per
//           doubling step: the resolved tail in front of every group), k_sy_tails (per group again: every unit's last 64 KiB resolved against a 64 KiB ring in LDS that
//           rolls forward), k_sy_resolve (all the other bytes, in pieces over the whole chip: the 64 KiB in front of a unit are
//           final by then and are read back into LDS; one byte gather per reference).
//
// Whatever is wrong with the stream (offset 0, offset before the start of the block) only raises SyPlan.fail: the single wavefront
// then decodes and reports what lz4.UncompressBlock reports.  A block that passes is decoded to exactly the bytes the serial
// decoder produces -- every output byte is written from the same source by the same rule, only in a different order.
// Pass A is bound by one wavefront's latency (round 3, phase clocks: a third of a unit's time is the token walk, a fifth the dependency rounds in
// LDS, an eighth the fetches from HBM; VALU busy 17 %), so its rate is the number of resident wavefronts: a 2 KiB image (+ 4 KiB of references)
// and a 2 KiB stream window are 9.3 KiB per wavefront -- 17 per CU instead of 9 -- and 1 GiB of shuffled float32 as the reference writes it
// takes 3.2 ms instead of 4.7 (3 KiB / 1 KiB history: 3.9; 2 KiB / 1 KiB: 3.25; long sequences suffer below 1.5 KiB of room: a ramp 1.9 -> 2.4).
#ifndef SY_IMG
#define SY_IMG   2048u
#define SY_HIST  512u
#endif
#ifndef SY_PWIN
#define SY_PWIN  2048u
#endif
#ifndef SY_WAVES
#define SY_WAVES 4
#endif
#include "hb_sym_decode.h"
#ifdef SY_DEBUG_TIMES
__device__ unsigned long long sy_dbg[48];
extern "C" void hb_debug_sy_times(unsigned long long *out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sy_dbg), sizeof(unsigned long long) * 48);
    if (reset) { unsigned long long z[48] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(sy_dbg), z, sizeof z); }
}
#endif

#define SY_W        65536u        // entries of a tail map / bytes of a tail image (index = distance 1..65535; entry 0 unused)
#ifndef SY_GROUPS
#define SY_GROUPS   256u          // groups of units in pass B (128: k_sy_compose 1.9 ms, 256: 1.05, 512: 1.0 but the scan over the groups 0.4)
#endif
#define SY_SUB      8u            // a region with more than SY_HEAVY bytes of output is decoded in this many parts
#ifndef SY_HEAVY
#define SY_HEAVY    (256u << 10)  // (1 MiB until regions went down to 8 KiB: a short frame's 100:1 plane then sat in a few 800 KiB units -- 4 MiB frame 2.9 -> 1.8 ms)
#endif
#define SY_MAXUNITS (RG_MAXREG * SY_SUB)
#define SY_PIECE    (256u << 10)  // bytes of a region one workgroup resolves at a time (k_sy_resolve)
#define SY_ROUNDS   3             // launches of pass A; the last one copies everything inline

struct SyLayout { size_t tok, plan, par, units, list, big, items, sym, maps, tails, total; };
static inline uint32_t sy_max_groups(size_t n_out) {
    // a group costs 512 KiB of map buffers: at most one per sixteen regions of the longest block n_out bytes can come from (a 1 MiB frame
    // then has 8 groups and 4 MiB of maps, not 256 and 128 MiB)
    const size_t g = (rg_max_regions(n_out) + 15) / 16;
    return (uint32_t)(g < 1 ? 1 : (g < SY_GROUPS ? g : SY_GROUPS));
}
static inline SyLayout sy_layout(size_t n_out) {
    SyLayout L; size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
    const uint32_t g = sy_max_groups(n_out);
    L.tok = take(rg_tok_bytes(n_out));                           // FIRST: the token discovery's store (hb_lz4_region.h), written by k_rg_parse
    L.plan = take(sizeof(SyPlan));
    L.par = take((size_t)SY_GROUPS * 4 * 2);                  // which of a group's two map buffers is current: two arrays, k_sy_scan flips between them
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

// wavefronts of pass A the current device holds at a time: CUs x (LDS per CU / LDS per wavefront), at most SY_WAVES per SIMD
static uint32_t hb_sy_slots() {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint32_t lds = (uint32_t)(SY_PWIN + 128 + DTQ * 8 + (SY_IMG + 64) * 3);
    uint32_t per_cu = (160u << 10) / ((lds + 511u) & ~511u);
    if (per_cu > 4u * SY_WAVES) per_cu = 4u * SY_WAVES;
    return (uint32_t)cus * per_cu;
}

// go: the token chain is verified and nothing has decoded the block yet
// ... and how many regions have output at all.  Pass A is a set of independent latency chains, one per unit, and `slots` wavefronts of it are
// resident at a time; pass B pays ~0.2 us per unit (k_sy_compose).  So the units should at least fill the chip once: two light neighbours
// share a unit only while that leaves at least 3/4 of the slots with a unit each (the headline frame as the reference writes it: 6876 regions -> 4025
// units on 4352 slots; random floats have 2860 token-dense regions behind three incompressible planes: merged into 879 units they kept a fifth
// of the chip busy for 5.3 ms, unmerged 2.0 ms).
__global__ __launch_bounds__(1024) void k_sy_gate(const RgPlan *rg, const RgRegion *__restrict__ reg, const DecPlan *dp, SyPlan *sy, uint32_t groups, uint32_t per, uint32_t slots,
                                                  int have_tok) {
    __shared__ uint32_t s_n[16];
    const int t = threadIdx.x;
    const uint32_t go = (rg->ok && !(dp->mode != DEC_SERIAL && !dp->fail)) ? 1u : 0u;      // (a Snappy block's plan: mode 2 = its 64 KiB units held, hb_snappy.hip)
    uint32_t n = 0;
    if (go) for (uint32_t r = t; r < rg->nreg; r += 1024u) n += reg[r].outlen != 0u ? 1u : 0u;
    for (int d = 32; d; d >>= 1) n += (uint32_t)__shfl_down((int)n, d);
    if ((t & 63) == 0) s_n[t >> 6] = n;
    __syncthreads();
    if (t == 0) {
        uint32_t live = 0;
        for (int k = 0; k < 16; k++) live += s_n[k];
        sy->go = go;
        sy->fail = 0; sy->groups = groups; sy->per = per; sy->nbig = 0; sy->nunits = rg->nreg * SY_SUB;
        // merge level: groups of 2^merge neighbours share a unit; pairs while live / 2 >= 3/4 slots.  (Groups of four when that still fills the
        // slots -- bit-shuffled data, 16378 token-dense regions -- were measured: k_sy_compose 1.35 -> 0.98 ms, but pass A 6.2 -> 9.2: one
        // generation of long units ends with its slowest one, two generations of shorter ones are dealt out as slots come free.)
        // pass A from the discovery's token store (k_sy_decode<true>) whenever there is one.  (As ONE kernel with a run-time switch the longer code
        // spilled 57 registers and lost 3.5 % on the headline frame; as a kernel of its own -- 11 spills -- it wins everywhere: the headline
        // reference frame 2.95 -> 2.30 ms, bit-shuffled ones 6.2 -> 4.65, random floats 2.94 -> 2.11.)
        sy->usetok = have_tok ? 1u : 0u;
        sy->live = live; sy->merge = (uint64_t)live * 2u >= (uint64_t)slots * 3u ? 1u : 0u;
    }
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
        uint32_t entry = R.entry, exitp = R.exit, L = R.outlen;
        const uint32_t O = (uint32_t)R.opos;
        // two neighbouring light regions make ONE unit (the even one gets it, the odd one stays empty): the discovery likes its regions
        // small -- a region is one wavefront's serial parse -- but every unit costs pass B a 64 Ki-entry map (k_sy_compose)
        for (uint32_t m = sy->merge; m > 0u; m--) {                       // aligned groups of 4, else of 2 (k_sy_gate says how far to go)
            const uint32_t w = 1u << m, g0 = r & ~(w - 1u);
            if (g0 + w > nreg) continue;
            uint64_t tot = 0;
            for (uint32_t k = 0; k < w; k++) tot += reg[g0 + k].outlen;
            if (tot <= SY_HEAVY) {
                if (r != g0) { entry = exitp; L = 0u; }
                else { exitp = reg[g0 + w - 1u].exit; L = (uint32_t)tot; }
                break;
            }
        }
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
            x.rtp = 0; x.rout = 0; x.state = 0; x.plit = 0; x.pmlen = 0; x.poff = 0; x.pnext = 0; x.pad = 0;
            u[lane] = x;
        }
    }
}

// the units that have output, in order: all later kernels walk this list (most slots of `un` are empty)
__global__ __launch_bounds__(1024) void k_sy_compact(SyPlan *sy, const SyUnit *__restrict__ un, uint32_t *__restrict__ list) {
    // sixteen wavefronts, each over a contiguous 1/16 of the slots, 64 slots (2 KiB, coalesced) per step: the occupancy masks stay in LDS and
    // the list is written from them (one read pass; a thread per 128 slots read them twice at a stride of 4 KiB: 0.17 ms)
    constexpr uint32_t SPAN = SY_MAXUNITS / 16, STEPS = SPAN / 64;
    __shared__ unsigned long long s_m[16][STEPS];
    __shared__ uint32_t s_c[16];
    if (!sy->go) return;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const uint32_t nu = sy->nunits;
    const uint32_t r0 = (uint32_t)w * SPAN;
    const uint32_t steps = r0 >= nu ? 0u : ((nu - r0 < SPAN ? nu - r0 : SPAN) + 63u) / 64u;
    uint32_t cnt = 0;
    for (uint32_t i = 0; i < steps; i += 8u) {                        // eight loads in flight (STEPS is a multiple of 8)
        uint32_t v[8];
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) { const uint32_t r = r0 + (i + q) * 64u + (uint32_t)lane; v[q] = (i + q < steps && r < nu) ? un[r].outlen : 0u; }
#pragma unroll
        for (uint32_t q = 0; q < 8u; q++) {
            const unsigned long long m = hb_ballot(v[q] != 0u);
            if (lane == 0) s_m[w][i + q] = m;
            cnt += (uint32_t)__builtin_popcountll(m);
        }
    }
    if (lane == 0) s_c[w] = cnt;
    __syncthreads();
    uint32_t o = 0, total = 0;
    for (int k = 0; k < 16; k++) { const uint32_t c = s_c[k]; if (k < w) o += c; total += c; }
    for (uint32_t i = 0; i < steps; i++) {
        const unsigned long long m = s_m[w][i];
        if ((m >> lane) & 1ull) list[o + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = r0 + i * 64u + (uint32_t)lane;
        o += (uint32_t)__builtin_popcountll(m);
    }
    if (t == 0) {
        const uint32_t g = sy->groups;
        sy->nact = total;
        sy->per = total ? (total + g - 1u) / g : 1u;
    }
}

// ---- pass A ----
// A literal run or a match of SY_BIG bytes and more is not for one wavefront (2-3 GB/s): the region posts it, stops in front of it
// and is resumed by the next launch, after k_sy_big has done all posted copies with the whole chip.  The LAST launch copies inline.
// Unit state between launches (SyUnit): rtp = stream position of the token to resume at, rout = output position there, state bit 0 = unit
// done, bit 1 = the literal run of the token at rtp was posted (the rest of that sequence is on record and is not parsed again).
template <bool TOK, int CODEC = RG_LZ4>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(SY_WAVES))) void k_sy_decode(const uint8_t *__restrict__ src, uint64_t n_src, SyUnit *un, const uint32_t *__restrict__ list,
                                                   SyPlan *sy, SyBig *big, uint8_t *D, uint16_t *S, int last, const RgPlan *rg, const RgRegion *reg,
                                                   const uint2 *tok, uint32_t tokcap) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[SY_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    __shared__ __attribute__((aligned(16))) uint8_t s_d[SY_IMG + 64];
    __shared__ __attribute__((aligned(16))) uint16_t s_s[SY_IMG + 64];
    if (!sy->go || sy->fail || (sy->usetok != 0u) != TOK) return;      // (both kernels are launched; k_sy_gate picked one)
    const int lane = threadIdx.x;
    const uint32_t nact = sy->nact;
    RgTokStore ts; ts.reg = reg; ts.tok = tok; ts.tokcap = tokcap; ts.rs = TOK ? rg->rs : 1u; ts.nreg = TOK ? rg->nreg : 0u;
    for (uint32_t i = blockIdx.x; i < nact; i += gridDim.x) {
        SyUnit *R = un + list[i];
        const uint32_t entry = RFL(R->entry), exitp = RFL(R->exit);
        if (RFL(R->outlen) == 0u || entry >= exitp) continue;
        const uint32_t st = RFL(R->state);
        if (st & 1u) continue;
        const uint32_t O = RFL(R->opos);
        const uint32_t rtp = RFL(R->rtp);
        const bool cont = rtp != 0u || (st & 2u) != 0u;                  // parked before (the block's first token is at position 0)
        const uint32_t start = cont ? rtp : entry;
        uint32_t out = cont ? RFL(R->rout) : O;        // next output byte
        bool parked;
        const bool ok = sy_decode_unit<true, SY_PWIN, TOK, CODEC>(src, n_src, start, exitp, 0u, O, out, D, S, s_win, s_tq, s_d, s_s, lane, last, rtp, st, R, sy, big, parked, O + RFL(R->outlen),
                                                           TOK ? &ts : nullptr);
        if (!parked) {
            if ((!ok || out != O + RFL(R->outlen)) && lane == 0) atomicExch(&sy->fail, 1u);
            if (lane == 0) R->state = 1u;
        }
        wave_sync();
    }
}

// the copies pass A posted, with the whole chip: 16 KiB per workgroup and step
__global__ __launch_bounds__(256) void k_sy_big(const uint8_t *__restrict__ src, SyPlan *sy, const SyBig *__restrict__ big, uint8_t *D, uint16_t *S) {
    if (!sy->go || sy->fail) return;
    const uint32_t nb = sy->nbig;
    const int t = threadIdx.x;
    uint8_t *Sb = (uint8_t *)S;
    // the 16 KiB pieces of ALL posted copies are dealt round-robin over the workgroups (piece c of copy i has the running number
    // base_i + c): a frame of many 256 KiB matches -- 16 pieces each -- keeps the whole grid busy instead of 16 workgroups per copy
    uint32_t base = 0;
    for (uint32_t i = 0; i < nb; i++) {
        const SyBig b = big[i];
        const uint32_t nch = (b.len + 16383u) >> 14;
        const uint32_t c0 = (blockIdx.x + gridDim.x - base % gridDim.x) % gridDim.x;
        base += nch;
        for (uint32_t c = c0; c < nch; c += gridDim.x) {
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
                uint32_t m = (x0 + (uint32_t)t) % off;                  // x mod off, kept up by addition
                const uint32_t step = 256u % off;
                for (uint32_t x = x0 + (uint32_t)t; x < x1; x += 256u) {
                    const uint32_t sp = s0 + m;
                    if (sp >= b.O) { D[b.dst + x] = D[sp]; S[b.dst + x] = S[sp]; } else S[b.dst + x] = (uint16_t)(b.O - sp);
                    m += step; if (m >= off) m -= off;
                }
            }
        }
    }
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
        // eight consecutive entries per thread and step: d0 .. d0 + 7 are the bytes E - d0 - 7 .. E - d0 of the output in reverse -- one 16-byte
        // read of the references, one 8-byte read of the values, two 16-byte stores (an entry at a time this kernel issued 3000 memory
        // instructions per unit and workgroup: 1.18 ms per GiB, 0.90 now; unrolling further changes nothing: what is left is the dependent
        // read - gather - write chain of each of the ~16 units of a group, and the gathers of one CU)
#pragma unroll 2
        for (uint32_t d0 = 8u * (uint32_t)t; d0 < SY_W; d0 += 8192u) {
            uint32_t val[8];
            if (d0 != 0u && d0 + 7u <= L && d0 + 7u <= E) {
                const uint32_t p = E - d0 - 7u;
                const u32x4 sv = ld16u((const uint8_t *)(S + p));
                const uint64_t dv = ld8u(D + p);
                const uint32_t sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
                for (int i = 0; i < 8; i++) {                           // entry d0 + i <- position p + 7 - i
                    const int j = 7 - i;
                    const uint32_t sref = (sw[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
                    val[i] = sref ? c[sref] : (uint32_t)((dv >> (8 * j)) & 255ull);
                }
            } else if (d0 > L) {                                        // in front of this unit: the same bytes, L further from the end
                const u32x4 a = ld16u((const uint8_t *)(c + (d0 - L))), b = ld16u((const uint8_t *)(c + (d0 - L) + 4));
                val[0] = a.x; val[1] = a.y; val[2] = a.z; val[3] = a.w; val[4] = b.x; val[5] = b.y; val[6] = b.z; val[7] = b.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const uint32_t d = d0 + (uint32_t)i;
                    uint32_t v = 0;
                    if (d != 0u) {
                        if (d > L) v = c[d - L];
                        else if (d <= E) { const uint32_t p = E - d, sref = S[p]; v = sref ? c[sref] : (uint32_t)D[p]; }
                    }
                    val[i] = v;
                }
            }
            u32x4 o0, o1;
            o0.x = val[0]; o0.y = val[1]; o0.z = val[2]; o0.w = val[3]; o1.x = val[4]; o1.y = val[5]; o1.z = val[6]; o1.w = val[7];
            *(u32x4 *)(n + d0) = o0; *(u32x4 *)(n + d0 + 4) = o1;
        }
        __threadfence_block();
        __syncthreads();
        uint32_t *x = cur; cur = nxt; nxt = x;
    }
    if (t == 0) par[g] = (cur == maps + (size_t)g * 2 * SY_W) ? 0u : 1u;
}

// tails[g][d] = the final byte d positions in front of group g's first output byte (d = 1..65535), for every group: an inclusive scan over the group maps (Hillis-Steele, one launch per round: after the round
// with `step` the map of group g covers the groups (g - 2 step, g]), then every group reads the bytes in front of it off its
// predecessor's scanned map.  Composition of an earlier span A with a later one B: the byte d from the end of B is B[d] if that is a
// value, else the byte (reference) in front of B = that far from the end of A.  A group's two map buffers take turns (par_in / par_out
// say which is current); the workgroup of g + step reads group g's current buffer while g's own workgroup writes the other.
// (First version: one workgroup that walked the groups front to back, the resolved tail in LDS -- 127 dependent steps, 1.6 ms; the scan: 0.1-0.2.)
__global__ __launch_bounds__(1024) void k_sy_scan(const SyPlan *sy, uint32_t *maps, const uint32_t *__restrict__ par_in, uint32_t *__restrict__ par_out, uint32_t step) {
    if (!sy->go || sy->fail) return;
    const uint32_t g = blockIdx.x;
    if (g >= sy->groups) return;
    const int t = threadIdx.x;
    const uint32_t pg = par_in[g];
    if (g < step) { if (t == 0) par_out[g] = pg; return; }             // covers everything in front of it already
    const uint32_t *__restrict__ B = maps + ((size_t)g * 2 + pg) * SY_W;
    const uint32_t *__restrict__ A = maps + ((size_t)(g - step) * 2 + par_in[g - step]) * SY_W;
    uint32_t *__restrict__ N = maps + ((size_t)g * 2 + (pg ^ 1u)) * SY_W;
    for (uint32_t d0 = 0; d0 < SY_W; d0 += 8192u) {
        uint32_t e[8];
#pragma unroll
        for (int q = 0; q < 8; q++) e[q] = B[d0 + (uint32_t)q * 1024u + (uint32_t)t];
#pragma unroll
        for (int q = 0; q < 8; q++) if (e[q] >> 16) e[q] = A[e[q] >> 16];
#pragma unroll
        for (int q = 0; q < 8; q++) N[d0 + (uint32_t)q * 1024u + (uint32_t)t] = e[q];
    }
    if (t == 0) par_out[g] = pg ^ 1u;
}
// tails[g][d] = the final byte d positions in front of group g's first output byte: the scanned map of group g - 1 (what still is a
// reference there points in front of the block: never used, pass A checked)
__global__ __launch_bounds__(1024) void k_sy_front(const SyPlan *sy, const uint32_t *__restrict__ maps, const uint32_t *__restrict__ par, uint8_t *__restrict__ tails) {
    if (!sy->go || sy->fail) return;
    const uint32_t g = blockIdx.x;
    if (g >= sy->groups) return;
    const int t = threadIdx.x;
    uint32_t *o = (uint32_t *)(tails + (size_t)g * SY_W);
    if (g == 0u) { for (uint32_t i = t; i < SY_W / 4u; i += 1024u) o[i] = 0u; return; }
    const u32x4 *m = (const u32x4 *)(maps + ((size_t)(g - 1u) * 2 + par[g - 1u]) * SY_W);
    for (uint32_t i = t; i < SY_W / 4u; i += 1024u) {
        const u32x4 x = m[i];
        const uint32_t b0 = (x.x >> 16) ? 0u : (x.x & 255u), b1 = (x.y >> 16) ? 0u : (x.y & 255u), b2 = (x.z >> 16) ? 0u : (x.z & 255u), b3 = (x.w >> 16) ? 0u : (x.w & 255u);
        o[i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
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
#ifdef SY_DEBUG_TIMES
        atomicAdd(&sy_dbg[40], 1ull);                                   // 16-byte groups looked at / with references: how sparse the references are
        if ((sa.x | sa.y | sa.z | sa.w | sb.x | sb.y | sb.z | sb.w) != 0u) atomicAdd(&sy_dbg[41], 1ull);
#endif
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

__global__ __launch_bounds__(1024) void k_sy_items(const SyUnit *__restrict__ un, const uint32_t *__restrict__ list, const SyPlan *sy, uint32_t *itembase) {
    // itembase[r] = pieces of the units in front of list[r]; sixteen wavefronts, each a running sum over a contiguous 1/16 of the list
    // (64 entries per step), then everybody adds what the wavefronts in front of it counted
    constexpr uint32_t SPAN = SY_MAXUNITS / 16;
    __shared__ uint32_t s_c[16];
    if (!sy->go || sy->fail) return;
    const int t = threadIdx.x, w = t >> 6, lane = t & 63;
    const uint32_t nreg = sy->nact;
    const uint32_t r0 = (uint32_t)w * SPAN;
    const uint32_t steps = r0 >= nreg ? 0u : ((nreg - r0 < SPAN ? nreg - r0 : SPAN) + 63u) / 64u;
    uint32_t carry = 0;
    for (uint32_t i = 0; i < steps; i += 4u) {                        // four gathers in flight
        uint32_t Lq[4];
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) { const uint32_t r = r0 + (i + q) * 64u + (uint32_t)lane; Lq[q] = (i + q < steps && r < nreg) ? un[list[r]].outlen : 0u; }
#pragma unroll
        for (uint32_t q = 0; q < 4u; q++) {
            const uint32_t r = r0 + (i + q) * 64u + (uint32_t)lane;
            const uint32_t mine = Lq[q] > SY_W ? (Lq[q] - SY_W + SY_PIECE - 1u) / SY_PIECE : 0u;
            const uint32_t incl = wave_incl_scan_dpp(mine);
            if (i + q < steps && r < nreg) itembase[r] = carry + incl - mine;
            carry += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
        }
    }
    if (lane == 0) s_c[w] = carry;
    __syncthreads();
    uint32_t o = 0, total = 0;
    for (int k = 0; k < 16; k++) { const uint32_t c = s_c[k]; if (k < w) o += c; total += c; }
    if (o) for (uint32_t i = 0; i < steps; i++) { const uint32_t r = r0 + i * 64u + (uint32_t)lane; if (r < nreg) itembase[r] += o; }
    if (t == 0) itembase[SY_MAXUNITS] = total;
}

// what is left: the part of every region in front of its last 64 KiB, in pieces of SY_PIECE bytes over the whole chip.  Everything a
// piece can name is final by now (tails of earlier regions), so the 64 KiB in front of its region are simply read back.
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
// a Snappy block (hb_snappy.hip): the plan at a.work is an SnPlan -- same first words -- and the block also has to produce the length it declares
__global__ void k_sy_finish_sn(const RgPlan *rg, const SyPlan *sy, DecPlan *dp) {
    if (!sy->go || sy->fail) return;
    if (rg->total != (uint64_t)dp->nbytes) return;                      // (SnPlan.nbytes: the uvarint; the single wavefront reports the mismatch)
    dp->mode = DEC_INDEXED; dp->fail = 0;
}

// codec: RG_LZ4, or RG_SNAPPY (elements; no token store; the chain is hb_launch_snappy_region_chain's)
int hb_launch_lz4_sym_decode(const hb_dec_args &a, uint8_t *dst, uint8_t *sym_work, int mark_post, hipStream_t s, int codec) {
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
    static const bool no_tok = [] { const char *e = getenv("HIPBLOSC_DEBUG_NO_TOKEN_STORE"); return e && *e && *e != '0'; }();   // A/B (k_rg_parse then wrote none either)
    const uint2 *tok = (no_tok || codec == RG_SNAPPY) ? nullptr : (const uint2 *)(sym_work + L.tok);
    hipLaunchKernelGGL(k_sy_gate, dim3(1), dim3(1024), 0, s, rg, (const RgRegion *)reg, dp, sy, groups, per, hb_sy_slots(), tok ? 1 : 0);
    hipLaunchKernelGGL(k_sy_units, dim3((nreg + 3) / 4), dim3(64), 0, s, rg, reg, (const uint2 *)(w + RL.trace), sy, un);
    hipLaunchKernelGGL(k_sy_compact, dim3(1), dim3(1024), 0, s, sy, un, list);
    hb_prof_end(s);
    for (int k = codec == RG_SNAPPY ? SY_ROUNDS - 1 : 0; k < SY_ROUNDS; k++) {       // (Snappy: nothing parks -- no element reaches SY_BIG -- so one launch, the last)
        const int last = k + 1 == SY_ROUNDS;
        hb_prof_begin("k_sy_decode", s);
        if (codec == RG_SNAPPY) hipLaunchKernelGGL((k_sy_decode<false, RG_SNAPPY>), dim3(nunits < 16384u ? nunits : 16384u), dim3(64), 0, s, a.src, (uint64_t)a.n, un, list, sy, big, dst, S, last,
                                                   (const RgPlan *)rg, (const RgRegion *)reg, tok, rg_tokcap(rs));
        else if (tok) hipLaunchKernelGGL(k_sy_decode<true>, dim3(nunits < 16384u ? nunits : 16384u), dim3(64), 0, s, a.src, (uint64_t)a.n, un, list, sy, big, dst, S, last,
                                    (const RgPlan *)rg, (const RgRegion *)reg, tok, rg_tokcap(rs));
        else hipLaunchKernelGGL(k_sy_decode<false>, dim3(nunits < 16384u ? nunits : 16384u), dim3(64), 0, s, a.src, (uint64_t)a.n, un, list, sy, big, dst, S, last,
                                (const RgPlan *)rg, (const RgRegion *)reg, tok, rg_tokcap(rs));      // (HIPBLOSC_DEBUG_NO_TOKEN_STORE: A/B)
        hb_prof_end(s);
        if (!last) {
            hb_prof_begin("k_sy_big", s);
            hipLaunchKernelGGL(k_sy_big, dim3(1024), dim3(256), 0, s, a.src, sy, big, dst, S);
            hipLaunchKernelGGL(k_sy_big_reset, dim3(1), dim3(1), 0, s, sy);
            hb_prof_end(s);
        }
    }
    hb_prof_begin("k_sy_compose", s);
    hipLaunchKernelGGL(k_sy_compose, dim3(groups), dim3(1024), 0, s, un, list, sy, dst, S, maps, par);
    hb_prof_end(s);
    hb_prof_begin("k_sy_scan", s);
    {
        uint32_t *pin = par, *pout = par + SY_GROUPS;
        for (uint32_t step = 1; step < groups; step <<= 1) {
            hipLaunchKernelGGL(k_sy_scan, dim3(groups), dim3(1024), 0, s, sy, maps, pin, pout, step);
            uint32_t *x = pin; pin = pout; pout = x;
        }
        hipLaunchKernelGGL(k_sy_front, dim3(groups), dim3(1024), 0, s, sy, maps, pin, tails);
    }
    hb_prof_end(s);
    hb_prof_begin("k_sy_tails", s);
    hipLaunchKernelGGL(k_sy_tails, dim3(groups), dim3(1024), 0, s, un, list, sy, dst, S, tails);
    hb_prof_end(s);
    hb_prof_begin("k_sy_resolve", s);
    uint32_t *itembase = (uint32_t *)(sym_work + L.items);
    hipLaunchKernelGGL(k_sy_items, dim3(1), dim3(1024), 0, s, un, list, sy, itembase);
    hipLaunchKernelGGL(k_sy_resolve, dim3(2048), dim3(512), 0, s, un, list, sy, dst, S, itembase);
    if (codec == RG_SNAPPY) hipLaunchKernelGGL(k_sy_finish_sn, dim3(1), dim3(1), 0, s, (const RgPlan *)rg, (const SyPlan *)sy, dp);
    else hipLaunchKernelGGL(k_sy_finish, dim3(1), dim3(1), 0, s, rg, sy, dp, mark_post);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
