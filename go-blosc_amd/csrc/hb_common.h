// hb_common.h — shared device/host helpers for the gfx950 kernels (internal; not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/hipblosc.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// 16-/8-/4-byte accesses with no alignment promise.  gfx950 under amdhsa runs global memory in
// unaligned access mode, so these lower to single global_load/store_dwordx4 / dwordx2 / dword.
struct __attribute__((packed, aligned(1))) hb_u128u { u32x4 v; };
struct __attribute__((packed, aligned(1))) hb_u64u { uint64_t v; };
struct __attribute__((packed, aligned(1))) hb_u32u { uint32_t v; };
struct __attribute__((packed, aligned(1))) hb_u16u { uint16_t v; };

__device__ __forceinline__ u32x4 ld16u(const uint8_t *p) { return ((const hb_u128u *)p)->v; }
__device__ __forceinline__ void st16u(uint8_t *p, u32x4 v) { ((hb_u128u *)p)->v = v; }
// streaming variants ("nt": no reuse expected; bypass / do not retain in the caches)
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));
__device__ __forceinline__ u32x4 ld16u_nt(const uint8_t *p) { return __builtin_nontemporal_load((const u32x4_a1 *)p); }
__device__ __forceinline__ void st16u_nt(uint8_t *p, u32x4 v) { __builtin_nontemporal_store(v, (u32x4_a1 *)p); }
__device__ __forceinline__ uint64_t ld8u(const uint8_t *p) { return ((const hb_u64u *)p)->v; }
__device__ __forceinline__ void st8u(uint8_t *p, uint64_t v) { ((hb_u64u *)p)->v = v; }
__device__ __forceinline__ uint32_t ld4u(const uint8_t *p) { return ((const hb_u32u *)p)->v; }
__device__ __forceinline__ void st4u(uint8_t *p, uint32_t v) { ((hb_u32u *)p)->v = v; }

// wave-wide predicate mask straight from the compare (HIP's __ballot() first materialises the predicate as 0/1:
// v_cndmask + v_cmp per call)
#define hb_ballot(pred) __builtin_amdgcn_ballot_w64(pred)
// "is my lane's bit set in this WAVE-UNIFORM mask": the mask itself becomes the exec mask of the branch (s_and_saveexec on the SGPR pair),
// where (mask >> lane) & 1 costs a 64-bit vector shift, an and and a compare per use
#define hb_lane_in(mask) __builtin_amdgcn_inverse_ballot_w64(mask)
// mask[lane] ? a : b as ONE v_cndmask on the scalar mask (left to itself the compiler sinks the computation of `a` into an exec-masked
// block: two scalar instructions more, and the LZ4 step loops issue as many scalar as vector instructions)
__device__ __forceinline__ uint32_t hb_select_lane(unsigned long long mask, uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}

// all lanes of the wave have finished their LDS traffic up to here, and the compiler may not
// move LDS accesses across this point (single-wave producer/consumer through LDS).
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value of lane-1 (lane 0 gets `first`): one DPP move (wave_shr:1), no LDS round trip like __shfl_up
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v, uint32_t first) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}

// inclusive prefix sum over the 64 lanes with DPP adds: row_shr 1/2/4/8 inside the rows of 16, then row_bcast:15 into
// rows 1 and 3 and row_bcast:31 into rows 2 and 3 (gfx9 DPP controls)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142 /* row_bcast:15 */, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143 /* row_bcast:31 */, 0xc, 0xf, false);
    return v;
}

// 4x4 byte transpose: in e[i] = bytes of row i; out p[j] = {e0.bj, e1.bj, e2.bj, e3.bj}.  Involution.
__device__ __forceinline__ void transpose4x4(uint32_t e0, uint32_t e1, uint32_t e2, uint32_t e3,
                                             uint32_t &p0, uint32_t &p1, uint32_t &p2, uint32_t &p3) {
    // v_perm_b32 D = bytes of {S0:S1}; selector 0-3 -> S1 (2nd arg), 4-7 -> S0 (1st arg)
    const uint32_t t0 = __builtin_amdgcn_perm(e1, e0, 0x05010400u);  // e0.b0 e1.b0 e0.b1 e1.b1
    const uint32_t t1 = __builtin_amdgcn_perm(e1, e0, 0x07030602u);  // e0.b2 e1.b2 e0.b3 e1.b3
    const uint32_t t2 = __builtin_amdgcn_perm(e3, e2, 0x05010400u);
    const uint32_t t3 = __builtin_amdgcn_perm(e3, e2, 0x07030602u);
    p0 = __builtin_amdgcn_perm(t2, t0, 0x05040100u);
    p1 = __builtin_amdgcn_perm(t2, t0, 0x07060302u);
    p2 = __builtin_amdgcn_perm(t3, t1, 0x05040100u);
    p3 = __builtin_amdgcn_perm(t3, t1, 0x07060302u);
}

// 8x8 bit transpose in go-blosc's MSB-first convention (shuffle.go:192-200):
// in byte e = B[e]; out byte k has bit (7-e) = bit (7-k) of B[e].  Involution.
__device__ __forceinline__ uint64_t bit_transpose8x8_msb(uint64_t x) {
    // MSB-first on both axes = the LSB-first transpose rotated by 180 degrees = a flip about the
    // ANTI-diagonal of the 8x8 bit matrix (byte r, bit c) <-> (byte 7-c, bit 7-r): three masked
    // swap steps with strides 9/18/36 (the main-diagonal transpose uses 7/14/28).
    uint64_t t;
    t = (x ^ (x >> 9)) & 0x0055005500550055ull;  x = x ^ t ^ (t << 9);
    t = (x ^ (x >> 18)) & 0x0000333300003333ull; x = x ^ t ^ (t << 18);
    t = (x ^ (x >> 36)) & 0x000000000F0F0F0Full; x = x ^ t ^ (t << 36);
    return x;
}

// go-blosc's bitshuffle for typesize 4 on ONE window of 8 elements = 32 bytes (a = bytes 0..15, b = 16..31),
// shuffle.go:184-200 (forward) and :261-277 (inverse).  Output layout: 8 bytes per byte position.
template <bool INVERSE>
__device__ __forceinline__ void bitshuffle4_window(const u32x4 a, const u32x4 b, u32x4 &oa, u32x4 &ob) {
    if (!INVERSE) {
        // gather byte position bp of the 8 elements -> 8 bytes, bit-transpose, store at window + 8*bp
        uint32_t l0, l1, l2, l3, h0, h1, h2, h3;
        transpose4x4(a.x, a.y, a.z, a.w, l0, l1, l2, l3);   // l[bp] = byte bp of elements 0..3
        transpose4x4(b.x, b.y, b.z, b.w, h0, h1, h2, h3);   // h[bp] = byte bp of elements 4..7
        const uint64_t y0 = bit_transpose8x8_msb(((uint64_t)h0 << 32) | l0);
        const uint64_t y1 = bit_transpose8x8_msb(((uint64_t)h1 << 32) | l1);
        const uint64_t y2 = bit_transpose8x8_msb(((uint64_t)h2 << 32) | l2);
        const uint64_t y3 = bit_transpose8x8_msb(((uint64_t)h3 << 32) | l3);
        oa.x = (uint32_t)y0; oa.y = (uint32_t)(y0 >> 32); oa.z = (uint32_t)y1; oa.w = (uint32_t)(y1 >> 32);
        ob.x = (uint32_t)y2; ob.y = (uint32_t)(y2 >> 32); ob.z = (uint32_t)y3; ob.w = (uint32_t)(y3 >> 32);
    } else {
        // 8 bytes at window + 8*bp -> transpose -> byte e goes to element e, byte position bp
        const uint64_t y0 = bit_transpose8x8_msb(((uint64_t)a.y << 32) | a.x);
        const uint64_t y1 = bit_transpose8x8_msb(((uint64_t)a.w << 32) | a.z);
        const uint64_t y2 = bit_transpose8x8_msb(((uint64_t)b.y << 32) | b.x);
        const uint64_t y3 = bit_transpose8x8_msb(((uint64_t)b.w << 32) | b.z);
        uint32_t e0, e1, e2, e3, e4, e5, e6, e7;
        transpose4x4((uint32_t)y0, (uint32_t)y1, (uint32_t)y2, (uint32_t)y3, e0, e1, e2, e3);
        transpose4x4((uint32_t)(y0 >> 32), (uint32_t)(y1 >> 32), (uint32_t)(y2 >> 32), (uint32_t)(y3 >> 32), e4, e5, e6, e7);
        oa.x = e0; oa.y = e1; oa.z = e2; oa.w = e3; ob.x = e4; ob.y = e5; ob.z = e6; ob.w = e7;
    }
}

#define HB_HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return HB_ERR_HIP; } while (0)

// ---- stage timing for bench.py (hb_profile_*): HIP events on the launch stream around each kernel ----
void hb_prof_begin(const char *stage, hipStream_t s);   // no-ops unless hb_profile_enable(1)
void hb_prof_end(hipStream_t s);

void *hb_pool_take(int dev, size_t bytes, size_t *got);  // hb_api.hip: cached device scratch of the host-pointer entry points
void hb_pool_give(int dev, void *p, size_t bytes);
int hb_select_device(int device);                        // hipSetDevice with the ABI's error codes
unsigned hb_dbg_plane_mask();                            // hb_debug_plane_mask(): byte planes the fused LZ4 kernels work on (timing only)

// ---- internal launch API shared between translation units ----
int hb_launch_filter(int op, uint8_t *d_dst, const uint8_t *d_src, size_t n, int typesize, hipStream_t s);
// same, but every kernel returns immediately unless *gate != 0 (gate is read on the device)
int hb_launch_filter_gated(int op, uint8_t *d_dst, const uint8_t *d_src, size_t n, int typesize, const uint32_t *gate, hipStream_t s);
// the byte (un)shuffle of every whole `blocksize`-byte block of a buffer on its own, tile kernels only (false: not this shape, nothing launched)
bool hb_launch_shuffle_blocks(bool inverse, uint8_t *dst, const uint8_t *src, uint32_t nfull, uint32_t blocksize, int typesize, hipStream_t s);
// the same filter on a batch of independent buffers (device array of jobs; gate as above, NULL = always)
struct hb_filter_job { uint8_t *dst; const uint8_t *src; uint64_t n; const uint32_t *gate; };
int hb_launch_filter_batch(int op, const hb_filter_job *d_jobs, int njobs, size_t max_n, int typesize, hipStream_t s, int gated = 0);
