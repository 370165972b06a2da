// copy_ref.hip — the attainable-bandwidth yardstick of bench.py: a same-size device-to-device copy with 16 bytes per lane,
// nontemporal loads and stores, ONE vector per thread and no loop (the shape that measured fastest on MI355X in
// tools/lab/filter_lab.hip: 6.4-6.5 TB/s moved against 5.4-5.6 for grid-stride variants and for hipMemcpyAsync).
// Bench tooling only (libcopy_ref.so, loaded by bench.py with ctypes): not part of libhipblosc.so or of its ABI.
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libcopy_ref.so copy_ref.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 u32x4_a1 __attribute__((aligned(1)));

__global__ __launch_bounds__(256) void k_copy_ref(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint64_t nvec) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) __builtin_nontemporal_store(__builtin_nontemporal_load((const u32x4_a1 *)(src + i * 16)), (u32x4_a1 *)(dst + i * 16));
}

// n: multiple of 16; asynchronous on `stream` (a hipStream_t)
extern "C" int copy_ref_dev(void *d_dst, const void *d_src, uint64_t n, void *stream) {
    const uint64_t nvec = n / 16;
    if (nvec == 0 || nvec > 0x7FFFFFFFull * 256) return -1;
    hipLaunchKernelGGL(k_copy_ref, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (uint8_t *)d_dst, (const uint8_t *)d_src, nvec);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
