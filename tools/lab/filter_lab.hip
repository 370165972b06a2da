// filter_lab.hip — micro-benchmarks behind the filter kernels' launch shapes (tools only; not part of the library).
// What does THIS box give a same-size copy in the filters' access pattern, and which shape of the shuffle kernel gets closest?
//   hipcc -O3 --offload-arch=gfx950 -o filter_lab filter_lab.hip && ./filter_lab [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <algorithm>
#include <functional>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const uint8_t *p) { if (NT) return __builtin_nontemporal_load((const u32x4_a1 *)p); return *(const u32x4_a1 *)p; }
template <bool NT> __device__ __forceinline__ void st(uint8_t *p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, (u32x4_a1 *)p); else *(u32x4_a1 *)p = v; }

// grid-stride copy, U vectors of 16 B per lane in flight
template <int U, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_copy(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t nvec) {
    const uint64_t stride = (uint64_t)gridDim.x * 256 * U;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x; i < nvec; i += stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; k++) if (i + (uint64_t)k * 256 < nvec) v[k] = ld<NTL>(src + (i + (uint64_t)k * 256) * 16);
#pragma unroll
        for (int k = 0; k < U; k++) if (i + (uint64_t)k * 256 < nvec) st<NTS>(dst + (i + (uint64_t)k * 256) * 16, v[k]);
    }
}

__device__ __forceinline__ void transpose4x4(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3, uint32_t &p0, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    const uint32_t t0 = __builtin_amdgcn_perm(e1, e0, 0x05010400u), t1 = __builtin_amdgcn_perm(e1, e0, 0x07030602u);
    const uint32_t t2 = __builtin_amdgcn_perm(e3, e2, 0x05010400u), t3 = __builtin_amdgcn_perm(e3, e2, 0x07030602u);
    p0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u); p1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u); p3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// shuffle ts=4: T tiles of 1024 elements per wave iteration (all loads of the T tiles issued before the first use), W waves per workgroup
template <int T, int W, bool NTL, bool NTS>
__global__ __launch_bounds__(W * 64) void k_shuf4(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ne, uint64_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[W][T][4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t t0 = ((uint64_t)blockIdx.x * W + wave) * T; t0 < ntiles; t0 += (uint64_t)gridDim.x * W * T) {
        u32x4 v[T][4];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int it = 0; it < 4; it++) v[t][it] = ld<NTL>(src + ((t0 + t) * 1024 + (uint64_t)it * 256 + lane * 4) * 4);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int it = 0; it < 4; it++) {
                uint32_t p0, p1, p2, p3;
                transpose4x4(v[t][it].x, v[t][it].y, v[t][it].z, v[t][it].w, p0, p1, p2, p3);
                slab[wave][t][0][it * 64 + lane] = p0; slab[wave][t][1][it * 64 + lane] = p1;
                slab[wave][t][2][it * 64 + lane] = p2; slab[wave][t][3][it * 64 + lane] = p3;
            }
        wave_sync();
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) st<NTS>(dst + (uint64_t)j * ne + (t0 + t) * 1024 + lane * 16, *(const u32x4 *)&slab[wave][t][j][lane * 4]);
        wave_sync();
    }
}
// the same without LDS: a lane owns 16 consecutive elements (64 B): 4 loads at 16-B stride, register transposes, one 16-B store per plane
template <bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_shuf4_reg(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ne, uint64_t ntiles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t t0 = (uint64_t)blockIdx.x * 4 + wave; t0 < ntiles; t0 += (uint64_t)gridDim.x * 4) {
        const uint8_t *p = src + (t0 * 1024 + (uint64_t)lane * 16) * 4;
        u32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = ld<NTL>(p + k * 16);
        u32x4 o[4];
        uint32_t a, b, c, d;
        transpose4x4(v[0].x, v[0].y, v[0].z, v[0].w, a, b, c, d); o[0].x = a; o[1].x = b; o[2].x = c; o[3].x = d;
        transpose4x4(v[1].x, v[1].y, v[1].z, v[1].w, a, b, c, d); o[0].y = a; o[1].y = b; o[2].y = c; o[3].y = d;
        transpose4x4(v[2].x, v[2].y, v[2].z, v[2].w, a, b, c, d); o[0].z = a; o[1].z = b; o[2].z = c; o[3].z = d;
        transpose4x4(v[3].x, v[3].y, v[3].z, v[3].w, a, b, c, d); o[0].w = a; o[1].w = b; o[2].w = c; o[3].w = d;
#pragma unroll
        for (int j = 0; j < 4; j++) st<NTS>(dst + (uint64_t)j * ne + t0 * 1024 + lane * 16, o[j]);
    }
}
// unshuffle ts=4, T tiles per iteration
template <int T, int W, bool NTL, bool NTS>
__global__ __launch_bounds__(W * 64) void k_unshuf4(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ne, uint64_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[W][T][4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint64_t t0 = ((uint64_t)blockIdx.x * W + wave) * T; t0 < ntiles; t0 += (uint64_t)gridDim.x * W * T) {
        u32x4 v[T][4];
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) v[t][j] = ld<NTL>(src + (uint64_t)j * ne + (t0 + t) * 1024 + lane * 16);
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) *(u32x4 *)&slab[wave][t][j][lane * 4] = v[t][j];
        wave_sync();
#pragma unroll
        for (int t = 0; t < T; t++)
#pragma unroll
            for (int it = 0; it < 4; it++) {
                uint32_t e0, e1, e2, e3;
                transpose4x4(slab[wave][t][0][it * 64 + lane], slab[wave][t][1][it * 64 + lane], slab[wave][t][2][it * 64 + lane], slab[wave][t][3][it * 64 + lane], e0, e1, e2, e3);
                u32x4 o; o.x = e0; o.y = e1; o.z = e2; o.w = e3;
                st<NTS>(dst + ((t0 + t) * 1024 + (uint64_t)it * 256 + lane * 4) * 4, o);
            }
        wave_sync();
    }
}


// ---- one tile of K*256 elements per wave, no loop: grid = ntiles / W workgroups of W waves ----
template <int K, int W>
__global__ __launch_bounds__(W * 64) void k_shuf4_np(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ne, uint64_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[W][4][K * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t t0 = (uint64_t)blockIdx.x * W + wave;
    if (t0 >= ntiles) return;
    u32x4 v[K];
#pragma unroll
    for (int it = 0; it < K; it++) v[it] = ld<true>(src + (t0 * (K * 256) + (uint64_t)it * 256 + lane * 4) * 4);
#pragma unroll
    for (int it = 0; it < K; it++) {
        uint32_t p0, p1, p2, p3;
        transpose4x4(v[it].x, v[it].y, v[it].z, v[it].w, p0, p1, p2, p3);
        slab[wave][0][it * 64 + lane] = p0; slab[wave][1][it * 64 + lane] = p1; slab[wave][2][it * 64 + lane] = p2; slab[wave][3][it * 64 + lane] = p3;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int h = 0; h < K / 4; h++)
            st<true>(dst + (uint64_t)j * ne + t0 * (K * 256) + h * 1024 + lane * 16, *(const u32x4 *)&slab[wave][j][h * 256 + lane * 4]);
}
template <int K, int W>
__global__ __launch_bounds__(W * 64) void k_unshuf4_np(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ne, uint64_t ntiles) {
    __shared__ __attribute__((aligned(16))) uint32_t slab[W][4][K * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t t0 = (uint64_t)blockIdx.x * W + wave;
    if (t0 >= ntiles) return;
    u32x4 v[4][K / 4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int h = 0; h < K / 4; h++) v[j][h] = ld<true>(src + (uint64_t)j * ne + t0 * (K * 256) + h * 1024 + lane * 16);
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int h = 0; h < K / 4; h++) *(u32x4 *)&slab[wave][j][h * 256 + lane * 4] = v[j][h];
    wave_sync();
#pragma unroll
    for (int it = 0; it < K; it++) {
        uint32_t e0, e1, e2, e3;
        transpose4x4(slab[wave][0][it * 64 + lane], slab[wave][1][it * 64 + lane], slab[wave][2][it * 64 + lane], slab[wave][3][it * 64 + lane], e0, e1, e2, e3);
        u32x4 o; o.x = e0; o.y = e1; o.z = e2; o.w = e3;
        st<true>(dst + (t0 * (K * 256) + (uint64_t)it * 256 + lane * 4) * 4, o);
    }
}
// a trivial "transform" with the bitshuffle kernel's access shapes: (a) one 32-byte window per lane, two loads 16 B apart (lane stride 32 B);
// (b) the same bytes through coalesced 16-B vectors and an LDS exchange
template <int G>   // G windows per lane, no loop
__global__ __launch_bounds__(256) void k_win_direct(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ngroups) {
    const uint64_t g0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x);
    u32x4 a[G], b[G];
#pragma unroll
    for (int k = 0; k < G; k++) { const uint64_t g = g0 + (uint64_t)k * gridDim.x * 256; if (g < ngroups) { a[k] = ld<true>(src + g * 32); b[k] = ld<true>(src + g * 32 + 16); } }
#pragma unroll
    for (int k = 0; k < G; k++) { const uint64_t g = g0 + (uint64_t)k * gridDim.x * 256; if (g < ngroups) { u32x4 x = a[k] ^ b[k]; st<true>(dst + g * 32, x); st<true>(dst + g * 32 + 16, b[k]); } }
}
template <int W>
__global__ __launch_bounds__(W * 64) void k_win_lds(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t ngroups) {
    __shared__ __attribute__((aligned(16))) u32x4 slab[W][128];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint64_t base = ((uint64_t)blockIdx.x * W + wave) * 2048;      // 64 windows = 2 KiB per wave
    if (base >= ngroups * 32) return;
    const u32x4 v0 = ld<true>(src + base + lane * 16), v1 = ld<true>(src + base + 1024 + lane * 16);
    slab[wave][lane] = v0; slab[wave][64 + lane] = v1;
    wave_sync();
    u32x4 a = slab[wave][2 * lane], b = slab[wave][2 * lane + 1];
    a = a ^ b;
    wave_sync();
    slab[wave][2 * lane] = a; slab[wave][2 * lane + 1] = b;
    wave_sync();
    st<true>(dst + base + lane * 16, slab[wave][lane]); st<true>(dst + base + 1024 + lane * 16, slab[wave][64 + lane]);
}

struct Timer { hipEvent_t a, b; Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); } };
static double best_ms(std::function<void()> f, int reps = 7) {
    Timer t; f(); CK(hipDeviceSynchronize());
    std::vector<float> v;
    for (int i = 0; i < reps; i++) { CK(hipEventRecord(t.a)); f(); CK(hipEventRecord(t.b)); CK(hipEventSynchronize(t.b)); float ms; CK(hipEventElapsedTime(&ms, t.a, t.b)); v.push_back(ms); }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char **argv) {
    const size_t mib = argc > 1 ? atol(argv[1]) : 1024;
    const size_t n = mib << 20;
    uint8_t *src, *dst, *ref;
    CK(hipMalloc(&src, n + 256)); CK(hipMalloc(&dst, n + 256)); CK(hipMalloc(&ref, n + 256));
    std::vector<uint32_t> h(n / 4);
    uint64_t z = 88172645463325252ull;
    for (size_t i = 0; i < n / 4; i++) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; h[i] = (uint32_t)z; }
    CK(hipMemcpy(src, h.data(), n, hipMemcpyHostToDevice));
    const uint64_t ne = n / 4, ntiles = ne / 1024, nvec = n / 16;
    auto report = [&](const char *name, double ms) { printf("%-44s %8.4f ms  %7.1f GB/s moved  frac %.3f\n", name, ms, 2.0 * n / ms / 1e6, 2.0 * n / ms / 1e6 / 8000.0); fflush(stdout); };
    report("hipMemcpyAsync D2D", best_ms([&] { CK(hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, 0)); }));
#define COPY(U, NTL, NTS, G) report("copy U=" #U " ntl=" #NTL " nts=" #NTS " grid=" #G, best_ms([&] { hipLaunchKernelGGL((k_copy<U, NTL, NTS>), dim3(G), dim3(256), 0, 0, dst, src, nvec); }))
    COPY(1, false, false, 2048); COPY(1, true, true, 2048); COPY(4, false, false, 2048); COPY(4, true, true, 2048); COPY(4, true, false, 2048); COPY(4, false, true, 2048);
    COPY(4, true, true, 1024); COPY(4, true, true, 4096); COPY(4, true, true, 8192); COPY(4, true, true, 65536); COPY(8, true, true, 2048); COPY(2, true, true, 2048); COPY(1, true, true, 262144);
    // reference result for the shuffle variants
    hipLaunchKernelGGL((k_shuf4<1, 4, true, true>), dim3(2048), dim3(256), 0, 0, ref, src, ne, ntiles); CK(hipDeviceSynchronize());
    std::vector<uint8_t> hr(n), ho(n);
    CK(hipMemcpy(hr.data(), ref, n, hipMemcpyDeviceToHost));
    {   // check the reference itself on a sample against the definition
        for (size_t i = 0; i < 1000000; i += 7) for (int j = 0; j < 4; j++) if (hr[(size_t)j * ne + i] != ((const uint8_t *)h.data())[i * 4 + j]) { printf("reference shuffle wrong\n"); return 1; }
    }
    auto check = [&](const char *name) { CK(hipMemcpy(ho.data(), dst, n, hipMemcpyDeviceToHost)); if (ho != hr) { printf("%s: WRONG OUTPUT\n", name); } };
#define SHUF(T, W, NTL, NTS, G) do { CK(hipMemset(dst, 0, n)); const double ms = best_ms([&] { hipLaunchKernelGGL((k_shuf4<T, W, NTL, NTS>), dim3(G), dim3(W * 64), 0, 0, dst, src, ne, ntiles); }); \
        report("shuffle4 T=" #T " W=" #W " ntl=" #NTL " nts=" #NTS " grid=" #G, ms); check("shuffle4 T=" #T " W=" #W " grid=" #G); } while (0)
    SHUF(1, 4, true, true, 2048); SHUF(1, 4, false, false, 2048); SHUF(1, 4, true, false, 2048); SHUF(1, 4, false, true, 2048);
    SHUF(2, 4, true, true, 2048); SHUF(2, 4, true, true, 1024); SHUF(4, 4, true, true, 1024); SHUF(4, 4, true, true, 512);
    SHUF(1, 4, true, true, 1024); SHUF(1, 4, true, true, 4096); SHUF(1, 4, true, true, 65536); SHUF(1, 1, true, true, 262144); SHUF(1, 1, true, true, 8192); SHUF(2, 1, true, true, 8192);
    SHUF(1, 8, true, true, 1024); SHUF(2, 2, true, true, 4096); SHUF(2, 4, false, true, 2048); SHUF(2, 4, true, false, 2048);
    {
        CK(hipMemset(dst, 0, n));
        report("shuffle4 registers only (no LDS) nt", best_ms([&] { hipLaunchKernelGGL((k_shuf4_reg<true, true>), dim3(2048), dim3(256), 0, 0, dst, src, ne, ntiles); }));
        check("shuffle4 reg");
        report("shuffle4 registers only (no LDS) plain loads", best_ms([&] { hipLaunchKernelGGL((k_shuf4_reg<false, true>), dim3(2048), dim3(256), 0, 0, dst, src, ne, ntiles); }));
    }
    // unshuffle: ref -> dst must give src
    std::vector<uint8_t> hs(n); memcpy(hs.data(), h.data(), n);
    auto check_u = [&](const char *name) { CK(hipMemcpy(ho.data(), dst, n, hipMemcpyDeviceToHost)); if (ho != hs) { printf("%s: WRONG OUTPUT\n", name); } };
#define UNSHUF(T, W, NTL, NTS, G) do { CK(hipMemset(dst, 0, n)); const double ms = best_ms([&] { hipLaunchKernelGGL((k_unshuf4<T, W, NTL, NTS>), dim3(G), dim3(W * 64), 0, 0, dst, ref, ne, ntiles); }); \
        report("unshuffle4 T=" #T " W=" #W " ntl=" #NTL " nts=" #NTS " grid=" #G, ms); check_u("unshuffle4 T=" #T " W=" #W " grid=" #G); } while (0)
    UNSHUF(1, 4, true, true, 2048); UNSHUF(1, 4, false, false, 2048); UNSHUF(2, 4, true, true, 2048); UNSHUF(2, 4, true, true, 1024); UNSHUF(4, 4, true, true, 512);
    UNSHUF(1, 4, true, true, 4096); UNSHUF(1, 1, true, true, 8192); UNSHUF(1, 4, true, false, 2048); UNSHUF(1, 4, false, true, 2048); UNSHUF(1, 4, true, true, 65536);

#define SHUFNP(K, W) do { CK(hipMemset(dst, 0, n)); const uint64_t nt = ne / (K * 256); const double ms = best_ms([&] { hipLaunchKernelGGL((k_shuf4_np<K, W>), dim3((unsigned)((nt + W - 1) / W)), dim3(W * 64), 0, 0, dst, src, ne, nt); }); \
        report("shuffle4 one tile per wave K=" #K " W=" #W, ms); check("shuffle4 np K=" #K " W=" #W); } while (0)
    SHUFNP(4, 1); SHUFNP(4, 2); SHUFNP(4, 4); SHUFNP(8, 1); SHUFNP(8, 2); SHUFNP(8, 4); SHUFNP(16, 1); SHUFNP(4, 8);
#define UNSHUFNP(K, W) do { CK(hipMemset(dst, 0, n)); const uint64_t nt = ne / (K * 256); const double ms = best_ms([&] { hipLaunchKernelGGL((k_unshuf4_np<K, W>), dim3((unsigned)((nt + W - 1) / W)), dim3(W * 64), 0, 0, dst, ref, ne, nt); }); \
        report("unshuffle4 one tile per wave K=" #K " W=" #W, ms); check_u("unshuffle4 np K=" #K " W=" #W); } while (0)
    UNSHUFNP(4, 1); UNSHUFNP(4, 2); UNSHUFNP(4, 4); UNSHUFNP(8, 1); UNSHUFNP(8, 2); UNSHUFNP(8, 4); UNSHUFNP(16, 1); UNSHUFNP(4, 8);
    {
        const uint64_t ng = n / 32;
        report("window access, 1 per lane, grid = all", best_ms([&] { hipLaunchKernelGGL((k_win_direct<1>), dim3((unsigned)(ng / 256)), dim3(256), 0, 0, dst, src, ng); }));
        report("window access, 2 per lane", best_ms([&] { hipLaunchKernelGGL((k_win_direct<2>), dim3((unsigned)(ng / 512)), dim3(256), 0, 0, dst, src, ng); }));
        report("window access, 4 per lane", best_ms([&] { hipLaunchKernelGGL((k_win_direct<4>), dim3((unsigned)(ng / 1024)), dim3(256), 0, 0, dst, src, ng); }));
        report("window via LDS exchange W=1", best_ms([&] { hipLaunchKernelGGL((k_win_lds<1>), dim3((unsigned)(ng / 64)), dim3(64), 0, 0, dst, src, ng); }));
        report("window via LDS exchange W=4", best_ms([&] { hipLaunchKernelGGL((k_win_lds<4>), dim3((unsigned)(ng / 256)), dim3(256), 0, 0, dst, src, ng); }));
    }
    // copies with one vector per thread and small workgroups
    report("copy 1 vec/thread, 64-thread WGs", best_ms([&] { hipLaunchKernelGGL((k_copy<1, true, true>), dim3((unsigned)(nvec / 256)), dim3(256), 0, 0, dst, src, nvec); }));
    return 0;
}
