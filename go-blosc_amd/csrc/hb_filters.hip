// hb_filters.hip — byte-shuffle / unshuffle / bitshuffle / bitunshuffle for gfx950.
//
// Semantics: shuffle.go:16-73, :76-133, :145-219, :222-295 of the reference (scalar branches;
// its AVX2/NEON paths produce identical bytes).  These are byte/bit PERMUTATIONS: HBM-bound,
// 2*n algorithmic bytes, no MFMA.  Layout in HBM: src is the caller's AoS buffer, dst the
// `typesize` byte planes of ne = n/typesize bytes each (tail n%typesize verbatim).
//
// Vector kernels (typesize 2/4/8/16): one wavefront owns a tile of 1024 elements.  Each lane
// reads 16-byte vectors (coalesced), transposes 4 elements x 4 bytes in registers with v_perm_b32,
// and parks one dword per plane in a wave-private LDS slab; the slab is then read back as one
// ds_read_b128 per plane and stored with 16 B per lane = 1 KiB contiguous per plane per
// instruction.  No workgroup barrier: a wave only ever reads its own slab.
// Launch shape (round 3, tools/lab/filter_lab.hip): ONE tile per single-wave workgroup and no loop -- 262 144 workgroups per GiB
// at typesize 4.  A streaming kernel on this chip wants the dispatcher, not a persistent grid, to deal out the work: the same
// tile code as a grid-stride loop over 2048 workgroups of 4 waves moved 5.2 TB/s, as one tile per 64-thread workgroup 6.0
// (a plain 16-byte copy: 5.6 against 6.4-6.5).
// Generic kernels (any typesize, ragged ends) are byte-granular and finish what the tiles leave.
#include "hb_common.h"

#define TILE_ELEMS 1024

// ----------------------------------------------------------------------------------------------
// byte shuffle / unshuffle, vector path
// ----------------------------------------------------------------------------------------------
template <int TS>
__device__ __forceinline__ void shuffle_tile(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, const uint64_t ne, const uint64_t tile,
                                             uint32_t (*my)[256], const int lane) {
    {
        const uint64_t e0 = tile * TILE_ELEMS;
        if constexpr (TS == 2) {
#pragma unroll
            for (int it = 0; it < 2; it++) {       // 8 elements (16 B) per lane per step
                const u32x4 v = ld16u_nt(src + (e0 + (uint64_t)it * 512 + lane * 8) * 2);
                u32x2 p0, p1;
                p0.x = __builtin_amdgcn_perm(v.y, v.x, 0x06040200u); p1.x = __builtin_amdgcn_perm(v.y, v.x, 0x07050301u);
                p0.y = __builtin_amdgcn_perm(v.w, v.z, 0x06040200u); p1.y = __builtin_amdgcn_perm(v.w, v.z, 0x07050301u);
                *(u32x2 *)&my[0][it * 128 + lane * 2] = p0;
                *(u32x2 *)&my[1][it * 128 + lane * 2] = p1;
            }
        } else {
            u32x4 vin[4][TS / 4];                  // all loads of the tile in flight before the first use
#pragma unroll
            for (int it = 0; it < 4; it++)
#pragma unroll
                for (int q = 0; q < TS / 4; q++) vin[it][q] = ld16u_nt(src + (e0 + (uint64_t)it * 256 + lane * 4) * TS + q * 16);
#pragma unroll
            for (int it = 0; it < 4; it++) {       // 4 elements (4*TS bytes) per lane per step
                uint32_t w[TS];                    // w[e*TS/4 + q] = dword q of element e
#pragma unroll
                for (int q = 0; q < TS / 4; q++) {
                    const u32x4 v = vin[it][q];
                    w[q * 4 + 0] = v.x; w[q * 4 + 1] = v.y; w[q * 4 + 2] = v.z; w[q * 4 + 3] = v.w;
                }
#pragma unroll
                for (int q = 0; q < TS / 4; q++) { // byte planes 4q .. 4q+3
                    uint32_t p0, p1, p2, p3;
                    transpose4x4(w[0 * (TS / 4) + q], w[1 * (TS / 4) + q], w[2 * (TS / 4) + q], w[3 * (TS / 4) + q],
                                 p0, p1, p2, p3);
                    my[4 * q + 0][it * 64 + lane] = p0;
                    my[4 * q + 1][it * 64 + lane] = p1;
                    my[4 * q + 2][it * 64 + lane] = p2;
                    my[4 * q + 3][it * 64 + lane] = p3;
                }
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < TS; j++) {
            const u32x4 v = *(const u32x4 *)&my[j][lane * 4];
            st16u_nt(dst + (uint64_t)j * ne + e0 + lane * 16, v);
        }
    }
}
template <int TS>
__global__ __launch_bounds__(64) void k_shuffle_vec(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                    uint64_t ne, uint64_t ntiles, const uint32_t *gate) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[TS][256];
    if (gate && *gate == 0) return;
    shuffle_tile<TS>(dst, src, ne, blockIdx.x, slab, threadIdx.x);
}

template <int TS>
__device__ __forceinline__ void unshuffle_tile(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, const uint64_t ne, const uint64_t tile,
                                               uint32_t (*my)[256], const int lane) {
    {
        const uint64_t e0 = tile * TILE_ELEMS;
        u32x4 vin[TS];                             // all loads of the tile in flight before the first LDS write
#pragma unroll
        for (int j = 0; j < TS; j++) vin[j] = ld16u_nt(src + (uint64_t)j * ne + e0 + lane * 16);
#pragma unroll
        for (int j = 0; j < TS; j++) *(u32x4 *)&my[j][lane * 4] = vin[j];
        wave_sync();
        if constexpr (TS == 2) {
#pragma unroll
            for (int it = 0; it < 2; it++) {
                const u32x2 p0 = *(const u32x2 *)&my[0][it * 128 + lane * 2];
                const u32x2 p1 = *(const u32x2 *)&my[1][it * 128 + lane * 2];
                u32x4 v;                            // element e = {p0.byte e, p1.byte e}
                v.x = __builtin_amdgcn_perm(p1.x, p0.x, 0x05010400u); v.y = __builtin_amdgcn_perm(p1.x, p0.x, 0x07030602u);
                v.z = __builtin_amdgcn_perm(p1.y, p0.y, 0x05010400u); v.w = __builtin_amdgcn_perm(p1.y, p0.y, 0x07030602u);
                st16u_nt(dst + (e0 + (uint64_t)it * 512 + lane * 8) * 2, v);
            }
        } else {
#pragma unroll
            for (int it = 0; it < 4; it++) {
                uint32_t w[TS];
#pragma unroll
                for (int q = 0; q < TS / 4; q++) {
                    uint32_t e0_, e1_, e2_, e3_;
                    transpose4x4(my[4 * q + 0][it * 64 + lane], my[4 * q + 1][it * 64 + lane],
                                 my[4 * q + 2][it * 64 + lane], my[4 * q + 3][it * 64 + lane], e0_, e1_, e2_, e3_);
                    w[0 * (TS / 4) + q] = e0_; w[1 * (TS / 4) + q] = e1_; w[2 * (TS / 4) + q] = e2_; w[3 * (TS / 4) + q] = e3_;
                }
                uint8_t *p = dst + (e0 + (uint64_t)it * 256 + lane * 4) * TS;
#pragma unroll
                for (int q = 0; q < TS / 4; q++) {
                    u32x4 v; v.x = w[q * 4 + 0]; v.y = w[q * 4 + 1]; v.z = w[q * 4 + 2]; v.w = w[q * 4 + 3];
                    st16u_nt(p + q * 16, v);
                }
            }
        }
    }
}
template <int TS>
__global__ __launch_bounds__(64) void k_unshuffle_vec(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                      uint64_t ne, uint64_t ntiles, const uint32_t *gate) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[TS][256];
    if (gate && *gate == 0) return;
    unshuffle_tile<TS>(dst, src, ne, blockIdx.x, slab, threadIdx.x);
}

// ---- the same tiles for a batch of buffers (hb_*_frames_batch_dev): job blockIdx.y, tile blockIdx.x of that job ----
template <int TS, bool INVERSE>
__global__ __launch_bounds__(64) void k_shuffle_vec_batch(const hb_filter_job *__restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[TS][256];
    const hb_filter_job j = jobs[blockIdx.y];
    if (j.gate && *j.gate == 0) return;
    const uint64_t ne = j.n / TS;
    for (uint64_t t = blockIdx.x; t < ne / TILE_ELEMS; t += gridDim.x) {      // (one round unless the launch is a gated one: a few workgroups per job then)
        if (INVERSE) unshuffle_tile<TS>(j.dst, j.src, ne, t, slab, threadIdx.x);
        else shuffle_tile<TS>(j.dst, j.src, ne, t, slab, threadIdx.x);
        wave_sync();
    }
}

// ----------------------------------------------------------------------------------------------
// byte shuffle / unshuffle, generic path: elements [e_begin, e_end) of every plane + the tail bytes
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void shuffle_generic_body(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                     uint64_t n, uint64_t ne, uint32_t ts, uint64_t e_begin, int inverse) {
    const uint64_t cnt = ne - e_begin, total = cnt * ts, tail0 = ne * ts;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        if (!inverse) {                 // idx walks the planes: coalesced stores   (shuffle.go:62)
            const uint64_t j = idx / cnt, i = e_begin + idx % cnt;
            dst[j * ne + i] = src[i * ts + j];
        } else {                        // idx walks the elements: coalesced stores (shuffle.go:122)
            const uint64_t i = e_begin + idx / ts, j = idx % ts;
            dst[i * ts + j] = src[j * ne + i];
        }
    }
    for (uint64_t t = tail0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        dst[t] = src[t];                // shuffle.go:67-70 / :127-130
}
__global__ __launch_bounds__(256) void k_shuffle_generic(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                         uint64_t n, uint64_t ne, uint32_t ts,
                                                         uint64_t e_begin, int inverse, const uint32_t *gate) {
    if (gate && *gate == 0) return;
    shuffle_generic_body(dst, src, n, ne, ts, e_begin, inverse);
}
// batch: job blockIdx.y; vec != 0: the vector kernel has done the whole tiles of every job
__global__ __launch_bounds__(256) void k_shuffle_generic_batch(const hb_filter_job *__restrict__ jobs, uint32_t ts, int inverse, int vec) {
    const hb_filter_job j = jobs[blockIdx.y];
    if (j.gate && *j.gate == 0) return;
    const uint64_t ne = j.n / ts;
    const uint64_t e_begin = vec ? ne / TILE_ELEMS * TILE_ELEMS : 0;
    if (j.n == 0 || (e_begin == ne && ne * ts == j.n)) return;
    shuffle_generic_body(j.dst, j.src, j.n, ne, ts, e_begin, inverse);
}

// ----------------------------------------------------------------------------------------------
// bitshuffle / bitunshuffle
// ----------------------------------------------------------------------------------------------
// typesize 4: one lane owns one group of 8 elements = one 32-byte window (in and out).  One single-wave workgroup per 64 windows
// (2 KiB), no loop (launch shape: see the top of the file).  The window's two 16-byte halves reach their lane through LDS, so that
// every global access of the wave is one contiguous KiB (a lane reading its own 32 bytes makes every instruction touch half of each
// 128-byte line: 5.6 TB/s in the lab against 6.4 for this exchange); the last, partial workgroup takes the direct path.
// The GATED launches (the memcpy fallback of a fused frame, the un-filter behind the serial decoder) nearly always find their gate shut:
// a quarter of a million empty workgroups cost 55 us per launch (round 4: that was the 0.1 ms between the step's kernels), so they get a small
// grid that strides over the tiles -- slower by a tenth when the gate is open, 2 us when it is shut.
template <int TS, bool INVERSE>
__global__ __launch_bounds__(64) void k_shuffle_vec_gated(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                          uint64_t ne, uint64_t ntiles, const uint32_t *gate) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[TS][256];
    if (*gate == 0) return;
    for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (INVERSE) unshuffle_tile<TS>(dst, src, ne, t, slab, threadIdx.x);
        else shuffle_tile<TS>(dst, src, ne, t, slab, threadIdx.x);
        wave_sync();
    }
}

template <bool INVERSE>
__device__ __forceinline__ void bitshuffle4_tile(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, const uint64_t ngroups, u32x4 *slab, const uint64_t tile) {
    const int lane = threadIdx.x;
    const uint64_t g0 = tile * 64;
    if (g0 + 64 <= ngroups) {
        const uint8_t *s = src + g0 * 32;
        const u32x4 v0 = ld16u_nt(s + lane * 16), v1 = ld16u_nt(s + 1024 + lane * 16);
        slab[lane] = v0; slab[64 + lane] = v1;
        wave_sync();
        const u32x4 a = slab[2 * lane], b = slab[2 * lane + 1];
        u32x4 oa, ob;
        bitshuffle4_window<INVERSE>(a, b, oa, ob);
        wave_sync();
        slab[2 * lane] = oa; slab[2 * lane + 1] = ob;
        wave_sync();
        uint8_t *d = dst + g0 * 32;
        st16u_nt(d + lane * 16, slab[lane]);
        st16u_nt(d + 1024 + lane * 16, slab[64 + lane]);
    } else {
        const uint64_t g = g0 + lane;
        if (g < ngroups) {
            const u32x4 a = ld16u(src + g * 32), b = ld16u(src + g * 32 + 16);
            u32x4 oa, ob;
            bitshuffle4_window<INVERSE>(a, b, oa, ob);
            st16u(dst + g * 32, oa);
            st16u(dst + g * 32 + 16, ob);
        }
    }
}
template <bool INVERSE>
__global__ __launch_bounds__(64) void k_bitshuffle4(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                    uint64_t ngroups, const uint32_t *gate) {
    __shared__ __attribute__((aligned(16))) u32x4 slab[128];
    if (gate) {                                          // gated: a small grid strides over the tiles (see k_shuffle_vec_gated)
        if (*gate == 0) return;
        for (uint64_t t = blockIdx.x; t * 64 < ngroups; t += gridDim.x) { bitshuffle4_tile<INVERSE>(dst, src, ngroups, slab, t); wave_sync(); }
        return;
    }
    bitshuffle4_tile<INVERSE>(dst, src, ngroups, slab, blockIdx.x);
}
template <bool INVERSE>
__global__ __launch_bounds__(64) void k_bitshuffle4_batch(const hb_filter_job *__restrict__ jobs) {
    __shared__ __attribute__((aligned(16))) u32x4 slab[128];
    const hb_filter_job j = jobs[blockIdx.y];
    if (j.gate && *j.gate == 0) return;
    const uint64_t ng = j.n / 32;
    for (uint64_t t = blockIdx.x; t * 64 < ng; t += gridDim.x) { bitshuffle4_tile<INVERSE>(j.dst, j.src, ng, slab, t); wave_sync(); }
}

// any typesize: one thread per (group, byte position).  shuffle.go:184-200 / :261-277
__device__ __forceinline__ void bitshuffle_generic_body(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                        uint64_t n, uint64_t ngroups, uint32_t ts, int inverse, uint64_t g_begin) {
    const uint64_t total = ngroups * ts, done = ngroups * 8 * ts;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t idx = g_begin * ts + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const uint64_t g = idx / ts, bp = idx % ts, base = g * 8 * ts;
        uint64_t x = 0;
        if (!inverse) {
#pragma unroll
            for (int e = 0; e < 8; e++) x |= (uint64_t)src[base + (uint64_t)e * ts + bp] << (8 * e);
            const uint64_t y = bit_transpose8x8_msb(x);
#pragma unroll
            for (int k = 0; k < 8; k++) dst[base + bp * 8 + k] = (uint8_t)(y >> (8 * k));
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) x |= (uint64_t)src[base + bp * 8 + i] << (8 * i);
            const uint64_t y = bit_transpose8x8_msb(x);
#pragma unroll
            for (int e = 0; e < 8; e++) dst[base + (uint64_t)e * ts + bp] = (uint8_t)(y >> (8 * e));
        }
    }
    for (uint64_t t = done + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride)
        dst[t] = src[t];                // leftover elements + tail bytes, shuffle.go:206-216 / :282-292
}
__global__ __launch_bounds__(256) void k_bitshuffle_generic(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src,
                                                            uint64_t n, uint64_t ngroups, uint32_t ts, int inverse,
                                                            uint64_t g_begin, const uint32_t *gate) {
    if (gate && *gate == 0) return;
    bitshuffle_generic_body(dst, src, n, ngroups, ts, inverse, g_begin);
}
// batch: job blockIdx.y; vec != 0 (typesize 4): k_bitshuffle4_batch has done every whole window, only leftovers and tails remain
__global__ __launch_bounds__(256) void k_bitshuffle_generic_batch(const hb_filter_job *__restrict__ jobs, uint32_t ts, int inverse, int vec) {
    const hb_filter_job j = jobs[blockIdx.y];
    if (j.gate && *j.gate == 0) return;
    const uint64_t ng = j.n / ts / 8;
    if (j.n == 0 || (vec && ng * 8 * ts == j.n)) return;
    bitshuffle_generic_body(j.dst, j.src, j.n, ng, ts, inverse, vec ? ng : 0);
}

// ----------------------------------------------------------------------------------------------
// launch
// ----------------------------------------------------------------------------------------------
static inline unsigned grid_for(uint64_t work_items, unsigned per_block, unsigned cap_blocks) {
    uint64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap_blocks) b = cap_blocks;
    return (unsigned)b;
}

template <int TS>
static void launch_shuffle_vec(bool inverse, uint8_t *dst, const uint8_t *src, uint64_t ne, uint64_t ntiles, const uint32_t *gate, hipStream_t s) {
    if (gate) {
        const unsigned g = (unsigned)(ntiles < 4096 ? ntiles : 4096);
        if (!inverse) hipLaunchKernelGGL((k_shuffle_vec_gated<TS, false>), dim3(g), dim3(64), 0, s, dst, src, ne, ntiles, gate);
        else hipLaunchKernelGGL((k_shuffle_vec_gated<TS, true>), dim3(g), dim3(64), 0, s, dst, src, ne, ntiles, gate);
        return;
    }
    const unsigned grid = (unsigned)ntiles;             // one tile per single-wave workgroup (n < 4 GiB: at most 2^21 tiles)
    if (!inverse) hipLaunchKernelGGL(k_shuffle_vec<TS>, dim3(grid), dim3(64), 0, s, dst, src, ne, ntiles, gate);
    else hipLaunchKernelGGL(k_unshuffle_vec<TS>, dim3(grid), dim3(64), 0, s, dst, src, ne, ntiles, gate);
}

static int launch_filter_impl(int op, uint8_t *dst, const uint8_t *src, size_t n, int typesize, const uint32_t *gate, hipStream_t s);
int hb_launch_filter(int op, uint8_t *dst, const uint8_t *src, size_t n, int typesize, hipStream_t s) {
    static const char *names[4] = {"filter_shuffle", "filter_unshuffle", "filter_bitshuffle", "filter_bitunshuffle"};
    if (op < 0 || op > 3) return HB_ERR_BAD_ARG;
    hb_prof_begin(names[op], s);
    const int rc = launch_filter_impl(op, dst, src, n, typesize, nullptr, s);
    hb_prof_end(s);
    return rc;
}
int hb_launch_filter_gated(int op, uint8_t *dst, const uint8_t *src, size_t n, int typesize, const uint32_t *gate, hipStream_t s) {
    if (op < 0 || op > 3) return HB_ERR_BAD_ARG;
    return launch_filter_impl(op, dst, src, n, typesize, gate, s);
}
static int launch_filter_impl(int op, uint8_t *dst, const uint8_t *src, size_t n, int typesize, const uint32_t *gate, hipStream_t s) {
    if (n == 0) return HB_OK;
    if (typesize <= 1 || n < (size_t)typesize) {        // shuffle.go:17-19 etc.: identity
        HB_HIP_TRY(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s));
        return HB_OK;
    }
    const uint64_t ts = (uint64_t)typesize, ne = n / ts;
    if (op == HB_OP_SHUFFLE || op == HB_OP_UNSHUFFLE) {
        const bool inv = (op == HB_OP_UNSHUFFLE);
        uint64_t ntiles = 0;
        if (ts == 2 || ts == 4 || ts == 8 || ts == 16) {
            ntiles = ne / TILE_ELEMS;
            if (ntiles) {
                switch (ts) {
                case 2: launch_shuffle_vec<2>(inv, dst, src, ne, ntiles, gate, s); break;
                case 4: launch_shuffle_vec<4>(inv, dst, src, ne, ntiles, gate, s); break;
                case 8: launch_shuffle_vec<8>(inv, dst, src, ne, ntiles, gate, s); break;
                default: launch_shuffle_vec<16>(inv, dst, src, ne, ntiles, gate, s); break;
                }
            }
        }
        const uint64_t e_begin = ntiles * TILE_ELEMS;
        if (e_begin < ne || ne * ts < n) {
            const uint64_t items = (ne - e_begin) * ts + (n - ne * ts);
            hipLaunchKernelGGL(k_shuffle_generic, dim3(grid_for(items, 256, 256 * 16)), dim3(256), 0, s,
                               dst, src, (uint64_t)n, ne, (uint32_t)ts, e_begin, inv ? 1 : 0, gate);
        }
    } else {
        const bool inv = (op == HB_OP_BITUNSHUFFLE);
        const uint64_t ng = ne / 8;
        if (ts == 4 && ng > 0) {
            unsigned grid = (unsigned)((ng + 63) / 64);
            if (gate && grid > 4096u) grid = 4096u;
            if (!inv) hipLaunchKernelGGL(k_bitshuffle4<false>, dim3(grid), dim3(64), 0, s, dst, src, ng, gate);
            else hipLaunchKernelGGL(k_bitshuffle4<true>, dim3(grid), dim3(64), 0, s, dst, src, ng, gate);
            if (ng * 32 < n)   // leftover elements + tail bytes only (g_begin = ng: no groups)
                hipLaunchKernelGGL(k_bitshuffle_generic, dim3(1), dim3(256), 0, s, dst, src, (uint64_t)n, ng, 4u,
                                   inv ? 1 : 0, ng, gate);
        } else {
            const uint64_t items = ng * ts + (n - ng * 8 * ts);
            hipLaunchKernelGGL(k_bitshuffle_generic, dim3(grid_for(items, 256, 256 * 16)), dim3(256), 0, s,
                               dst, src, (uint64_t)n, ng, (uint32_t)ts, inv ? 1 : 0, (uint64_t)0, gate);
        }
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}

// ---- the byte shuffle applied to every `blocksize`-byte block of a buffer on its own (C-Blosc-1 frames filter per block, hb_cblosc.hip):
// tile blockIdx.x = (block, tile inside the block).  Whole blocks of whole tiles only; the caller does a last, shorter block otherwise. ----
template <int TS, bool INVERSE>
__global__ __launch_bounds__(64) void k_shuffle_vec_blocks(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t blocksize, uint32_t tiles_per_block) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[TS][256];
    const uint32_t b = blockIdx.x / tiles_per_block, t = blockIdx.x % tiles_per_block;
    const size_t base = (size_t)b * blocksize;
    if (INVERSE) unshuffle_tile<TS>(dst + base, src + base, blocksize / TS, t, slab, threadIdx.x);
    else shuffle_tile<TS>(dst + base, src + base, blocksize / TS, t, slab, threadIdx.x);
}
// returns true when it took the job: typesize 2 / 4 / 8 / 16, blocks of whole 1024-element tiles, nfull whole blocks
bool hb_launch_shuffle_blocks(bool inverse, uint8_t *dst, const uint8_t *src, uint32_t nfull, uint32_t blocksize, int typesize, hipStream_t s) {
    if (!(typesize == 2 || typesize == 4 || typesize == 8 || typesize == 16) || nfull == 0) return false;
    if (blocksize % ((uint32_t)typesize * TILE_ELEMS) != 0) return false;
    const uint32_t tpb = blocksize / (uint32_t)typesize / TILE_ELEMS;
    if ((uint64_t)nfull * tpb > 0x7FFFFFFFull) return false;
    const dim3 g(nfull * tpb);
    switch (typesize) {
    case 2: if (inverse) hipLaunchKernelGGL((k_shuffle_vec_blocks<2, true>), g, dim3(64), 0, s, dst, src, blocksize, tpb); else hipLaunchKernelGGL((k_shuffle_vec_blocks<2, false>), g, dim3(64), 0, s, dst, src, blocksize, tpb); break;
    case 4: if (inverse) hipLaunchKernelGGL((k_shuffle_vec_blocks<4, true>), g, dim3(64), 0, s, dst, src, blocksize, tpb); else hipLaunchKernelGGL((k_shuffle_vec_blocks<4, false>), g, dim3(64), 0, s, dst, src, blocksize, tpb); break;
    case 8: if (inverse) hipLaunchKernelGGL((k_shuffle_vec_blocks<8, true>), g, dim3(64), 0, s, dst, src, blocksize, tpb); else hipLaunchKernelGGL((k_shuffle_vec_blocks<8, false>), g, dim3(64), 0, s, dst, src, blocksize, tpb); break;
    default: if (inverse) hipLaunchKernelGGL((k_shuffle_vec_blocks<16, true>), g, dim3(64), 0, s, dst, src, blocksize, tpb); else hipLaunchKernelGGL((k_shuffle_vec_blocks<16, false>), g, dim3(64), 0, s, dst, src, blocksize, tpb); break;
    }
    return true;
}

// ---- batches (hb_*_frames_batch_dev): the same filter on `njobs` independent buffers in one or two launches.  d_jobs: device array;
// max_n: the largest job (sizes the grid; jobs smaller than that leave their surplus workgroups at once).  Identity cases
// (typesize <= 1, n < typesize: shuffle.go:17-19) are the caller's: it points the consumer at the source instead. ----
// gated != 0: the jobs carry gates that are nearly always shut (the memcpy fallback of fused frames): a few workgroups per job stride over its tiles
int hb_launch_filter_batch(int op, const hb_filter_job *d_jobs, int njobs, size_t max_n, int typesize, hipStream_t s, int gated) {
    if (op < 0 || op > 3) return HB_ERR_BAD_ARG;
    if (njobs <= 0 || max_n == 0 || typesize <= 1) return HB_OK;
    const uint64_t ts = (uint64_t)typesize;
    const bool inv = op == HB_OP_UNSHUFFLE || op == HB_OP_BITUNSHUFFLE;
    for (int j0 = 0; j0 < njobs; j0 += 65535) {                       // gridDim.y <= 65535
        const unsigned ny = (unsigned)(njobs - j0 < 65535 ? njobs - j0 : 65535);
        const hb_filter_job *jb = d_jobs + j0;
        if (op == HB_OP_SHUFFLE || op == HB_OP_UNSHUFFLE) {
            const bool vec = ts == 2 || ts == 4 || ts == 8 || ts == 16;
            const uint64_t mt = max_n / ts / TILE_ELEMS;
            if (vec && mt) {
                const dim3 g((unsigned)(gated && mt > 4 ? 4 : mt), ny);
                switch (ts) {
                case 2: if (inv) hipLaunchKernelGGL((k_shuffle_vec_batch<2, true>), g, dim3(64), 0, s, jb); else hipLaunchKernelGGL((k_shuffle_vec_batch<2, false>), g, dim3(64), 0, s, jb); break;
                case 4: if (inv) hipLaunchKernelGGL((k_shuffle_vec_batch<4, true>), g, dim3(64), 0, s, jb); else hipLaunchKernelGGL((k_shuffle_vec_batch<4, false>), g, dim3(64), 0, s, jb); break;
                case 8: if (inv) hipLaunchKernelGGL((k_shuffle_vec_batch<8, true>), g, dim3(64), 0, s, jb); else hipLaunchKernelGGL((k_shuffle_vec_batch<8, false>), g, dim3(64), 0, s, jb); break;
                default: if (inv) hipLaunchKernelGGL((k_shuffle_vec_batch<16, true>), g, dim3(64), 0, s, jb); else hipLaunchKernelGGL((k_shuffle_vec_batch<16, false>), g, dim3(64), 0, s, jb); break;
                }
            }
            // what the tiles leave (or everything, for the other typesizes): byte-granular, a few workgroups per job
            const uint64_t items = vec ? (uint64_t)TILE_ELEMS * ts + ts : max_n;
            hipLaunchKernelGGL(k_shuffle_generic_batch, dim3(grid_for(items, 256 * 16, 64), ny), dim3(256), 0, s, jb, (uint32_t)ts, inv ? 1 : 0, vec ? 1 : 0);
        } else {
            const bool vec = ts == 4;
            if (vec && max_n / 32) {
                const uint64_t gx = (max_n / 32 + 63) / 64;
                const dim3 g((unsigned)(gated && gx > 4 ? 4 : gx), ny);
                if (inv) hipLaunchKernelGGL(k_bitshuffle4_batch<true>, g, dim3(64), 0, s, jb); else hipLaunchKernelGGL(k_bitshuffle4_batch<false>, g, dim3(64), 0, s, jb);
            }
            const uint64_t items = vec ? 64 : max_n;
            hipLaunchKernelGGL(k_bitshuffle_generic_batch, dim3(grid_for(items, 256 * 16, 64), ny), dim3(256), 0, s, jb, (uint32_t)ts, inv ? 1 : 0, vec ? 1 : 0);
        }
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
