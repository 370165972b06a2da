// hb_cblosc.hip — SURVEY §8 row f4: frames in the C-Blosc-1 wire format (what go-blosc's README.md:20 claims to read and write and
// blosc.go does not: SURVEY §0.2), decoded on the device.  The format as c-blosc 1.21 writes it (blosc.c blosc_d / blosc_c; checked
// against /opt/conda/lib/libblosc.so.1.21.0, which is also what the tests produce their frames with):
//
//   header  16 bytes { version = 2, versionlz, flags, typesize, nbytes u32, blocksize u32, cbytes u32 }
//           flags: 0x01 byte shuffle, 0x02 memcpyed, 0x04 bit shuffle, 0x10 blocks are not split, bits 5-7 codec format
//           (0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd)
//   memcpyed: nbytes raw bytes follow.  Else bstarts: int32[nblocks], the offset of every block from the start of the frame;
//   a block is `typesize` streams (or ONE: flag 0x10, and always the last block when it is shorter than blocksize), a stream
//   is { int32 cbytes, data }: an LZ4 block that decodes to blocksize / nstreams bytes -- or those bytes themselves when
//   cbytes says exactly that size.  The filter works per BLOCK: byte shuffle = the block's elements transposed to `typesize`
//   byte planes (the streams ARE the planes; the bytes behind the last whole element stay where they are); bit shuffle =
//   element bit (byte j, bit k) of all elements of the block gathered in row 8 j + k, eight elements per byte, lowest first --
//   only when the block holds a multiple of 8 elements, else the block is stored unfiltered (shuffle.c blosc_internal_bitshuffle).
//
// Blocks and streams are independent, so there is no discovery problem here: k_cb_plan lists the streams (one lane per block
// walks its cbytes fields), k_cb_decode puts one wavefront on every stream (the serial block decoder of hb_dec_common.h: streams
// are 8-256 KiB, hundreds to thousands per frame), k_cb_unfilter undoes the per-block filter.  LZ4 / LZ4HC and memcpyed frames
// only: the other codec formats are entropy coders or blosclz.
#include "hb_sym_decode.h"

#define CB_FLAG_SHUFFLE    0x01u
#define CB_FLAG_MEMCPY     0x02u
#define CB_FLAG_BITSHUFFLE 0x04u
#define CB_FLAG_DONTSPLIT  0x10u

// blosc_d of c-blosc 1.x splits a block into `typesize` streams only when ALL of these hold (the rule from before the 0x10 flag existed stays
// in force next to it: frames of c-blosc < 1.15 have the bit clear and large typesizes / small blocks unsplit; checked against libblosc 1.21):
// not-split bit clear, typesize <= MAX_SPLITS (16), blocksize / typesize >= MIN_BUFFERSIZE (128), not the last, shorter block.
__host__ __device__ static inline uint32_t cb_nsplit(uint32_t flags, uint32_t typesize, uint32_t blocksize) {
    return (!(flags & 0x10u) && typesize >= 1u && typesize <= 16u && blocksize / typesize >= 128u) ? typesize : 1u;
}

struct CbStream { uint32_t src, csize, dst, usize; };
struct CbPlan { uint32_t fail, nblocks, nsplit, pad; };

static inline size_t cb_align(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- the streams of every block ----
__global__ void k_cb_plan(const uint8_t *__restrict__ frame, uint64_t n, uint32_t nbytes, uint32_t blocksize, uint32_t cbytes, uint32_t typesize, uint32_t flags,
                          CbPlan *plan, CbStream *streams) {
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nblocks) return;
    const uint32_t leftover = nbytes % blocksize;
    const bool lastshort = b + 1 == nblocks && leftover != 0u;
    const uint32_t bsize = lastshort ? leftover : blocksize;
    const uint32_t nsplit_frame = cb_nsplit(flags, typesize, blocksize);
    const uint32_t nsplit = lastshort ? 1u : nsplit_frame;
    const uint32_t neblock = bsize / nsplit;
    CbStream *out = streams + (size_t)b * nsplit_frame;
    for (uint32_t s = 0; s < nsplit_frame; s++) { CbStream z; z.src = 0; z.csize = 0; z.dst = 0; z.usize = 0; out[s] = z; }
    uint32_t p = (uint32_t)frame[16 + 4 * b] | ((uint32_t)frame[17 + 4 * b] << 8) | ((uint32_t)frame[18 + 4 * b] << 16) | ((uint32_t)frame[19 + 4 * b] << 24);
    bool bad = neblock == 0u || (uint64_t)neblock * nsplit != bsize;                  // (blosc_c only splits what divides)
    for (uint32_t s = 0; s < nsplit && !bad; s++) {
        if ((uint64_t)p + 4u > cbytes) { bad = true; break; }
        const uint32_t cb = (uint32_t)frame[p] | ((uint32_t)frame[p + 1] << 8) | ((uint32_t)frame[p + 2] << 16) | ((uint32_t)frame[p + 3] << 24);
        p += 4u;
        if (cb == 0u || cb > 0x7FFFFFFFu || (uint64_t)p + cb > cbytes || cb > neblock + neblock / 255u + 16u) { bad = true; break; }
        CbStream st; st.src = p; st.csize = cb; st.dst = b * blocksize + s * neblock; st.usize = neblock;
        out[s] = st;
        p += cb;
    }
    if (bad) atomicExch(&plan->fail, 1u);
}

#define CB_SMALL_IN 3072u        // longest stream k_cb_decode_small stages (see there)
// ---- one wavefront per stream ----
// The decoder is pass A of the symbolic decoder without the symbols (hb_sym_decode.h): 13 KiB of LDS per wavefront -- an image of the last
// 4 KiB of output, older sources read back from HBM -- so that a dozen streams per CU are in flight.  (First version: the serial
// block decoder with its 64 KiB history in LDS, one wavefront per CU: 21 GB/s on 256 MiB of float32.)
__global__ __launch_bounds__(64) void k_cb_decode(const uint8_t *__restrict__ frame, CbPlan *plan, const CbStream *__restrict__ streams, uint32_t nstreams,
                                                   uint8_t *dst, int small_elsewhere, uint32_t P) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    __shared__ __attribute__((aligned(16))) uint8_t s_d[SY_IMG + 64];
    const int lane = threadIdx.x;
    if (plan->fail) return;
    // stream order: see k_cb_decode_small (stream i is byte plane i % typesize of its block, workgroup i runs on XCD i % 8, and the
    // planes differ several times in cost); one stream per workgroup up to 65536 streams, so that a finished cheap stream makes
    // room for the next one
    const uint32_t mgrp = (nstreams + 7u) / 8u;
    for (uint32_t it = blockIdx.x; it < mgrp * 8u; it += gridDim.x) {
        const uint32_t i = (uint32_t)(((uint64_t)(it >> 3) * P) % mgrp) * 8u + ((it + (it >> 3) + it / gridDim.x) & 7u);
        if (i >= nstreams) continue;
        CbStream st = streams[i];                                              // wave-uniform, but it comes out of a vector load: to scalar registers
        st.src = RFL(st.src); st.csize = RFL(st.csize); st.dst = RFL(st.dst); st.usize = RFL(st.usize);
        if (st.usize == 0u) continue;
        if (st.csize == st.usize) { wave_copy_g2g(dst + st.dst, frame + st.src, st.usize, lane); continue; }      // stored
        if (small_elsewhere && st.usize <= HB_CHUNK && st.csize <= CB_SMALL_IN) continue;   // k_cb_decode_small has it
        uint32_t out = st.dst;
        bool parked;
        const bool ok = sy_decode_unit<false>(frame + st.src, (uint64_t)st.csize, 0u, st.csize, st.dst, st.dst, out, dst, nullptr, s_win, s_tq, s_d, nullptr,
                                              lane, 1, 0u, 0u, nullptr, nullptr, nullptr, parked, st.dst + st.usize);
        if ((!ok || out != st.dst + st.usize) && lane == 0) atomicExch(&plan->fail, 1u);     // blosc_d: "nbytes != neblock -> -2"
        wave_sync();
    }
}

// Streams of at most one chunk (what hb_cblosc_compress writes: every stream is a 4 KiB chunk of the matcher): stream and output both
// fit into LDS, so the chunk decoder's machinery applies as it is -- window-parallel token parser, one token per lane, copies inside
// the LDS image in dependency rounds (hb_dec_common.h) -- at 17 wavefronts per CU.  k_cb_decode leaves these streams alone.
#define CB_SMALL HB_CHUNK
__global__ __launch_bounds__(64) void k_cb_decode_small(const uint8_t *__restrict__ frame, CbPlan *plan, const CbStream *__restrict__ streams, uint32_t nstreams,
                                                         uint8_t *__restrict__ dst, uint32_t P) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[CB_SMALL_IN + 64 + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[CB_SMALL + 64];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    if (plan->fail) return;
    // stream order as in k_dec_indexed (hb_lz4_dec.hip): the streams of a split block are its byte planes, stream i = plane i % typesize,
    // and workgroup i runs on XCD i % 8 -- in stream order two XCDs would get all the streams of the token-dense plane (measured: 3.1 ms
    // against 1.5).  Workgroup (step k = it / 8, XCD x = it % 8) takes stream 8 * (k * P mod m) + (x + k) % 8; grid a multiple of 8.
    const uint32_t mgrp = (nstreams + 7u) / 8u;
    for (uint32_t it = blockIdx.x; it < mgrp * 8u; it += gridDim.x) {
        const uint32_t i = (uint32_t)(((uint64_t)(it >> 3) * P) % mgrp) * 8u + ((it + (it >> 3) + it / gridDim.x) & 7u);
        if (i >= nstreams) continue;
        CbStream st = streams[i];                                              // wave-uniform, but it comes out of a vector load: to scalar registers
        st.src = RFL(st.src); st.csize = RFL(st.csize); st.dst = RFL(st.dst); st.usize = RFL(st.usize);
        if (st.usize == 0u || st.usize > CB_SMALL || st.csize == st.usize || st.csize > CB_SMALL_IN) continue;      // (stored streams: k_cb_decode copies them)
        const uint8_t *g = frame + st.src;
        const uint32_t sh = (uint32_t)((uintptr_t)g & 15u), slen = st.csize;           // slen < usize <= 4096
        wave_sync();
        {
            const u32x4 *ga = (const u32x4 *)(g - sh);
            const uint32_t nv = (sh + slen + 15u) >> 4;
            for (uint32_t k = lane; k < nv; k += 64) ((u32x4 *)s_in)[k] = ga[k];
        }
        wave_sync();
        uint32_t si = 0, di = 0, nq = 0;
        bool ok = true, done = false;
        while (ok && !done) {
            const bool stop = dec_fill(s_in, sh, slen, slen, si, nq, s_tq, lane);
            bool rewound = false;
            if (!dec_drain(s_in, (int)sh, s_out, st.usize, 0u, di, si, nq, s_tq, stop, rewound, lane) || rewound) { ok = false; break; }      // (64 at a time; all of them when the parser stopped)
            if (si == slen) { done = true; break; }
            if (!stop) continue;
            // one sequence the slow way: a length extension of several bytes, or the end of the stream
            const uint32_t tok = RFL((uint32_t)s_in[sh + si]);
            si++;
            uint32_t ll = tok >> 4;
            if (ll == 15u && !dec_read_ext(s_in, (int)sh, 0u, slen, g, slen, si, ll, lane)) { ok = false; break; }
            if (ll > slen - si || ll > st.usize - di) { ok = false; break; }
            for (uint32_t k = lane; k < ll; k += 64) s_out[di + k] = s_in[sh + si + k];
            si += ll; di += ll;
            if (si == slen) { if (tok & 15u) ok = false; done = true; break; }            // the final, literal-only sequence
            if (slen - si < 2u) { ok = false; break; }
            const uint32_t off = RFL((uint32_t)s_in[sh + si] | ((uint32_t)s_in[sh + si + 1u] << 8));
            si += 2u;
            uint32_t ml = (tok & 15u) + 4u;
            if ((tok & 15u) == 15u && !dec_read_ext(s_in, (int)sh, 0u, slen, g, slen, si, ml, lane)) { ok = false; break; }
            if (off == 0u || off > di || ml > st.usize - di) { ok = false; break; }
            wave_sync();
            dec_match_copy(s_out, di, off, ml, lane);
            di += ml;
            wave_sync();
        }
        if (!ok || di != st.usize) { if (lane == 0) atomicExch(&plan->fail, 1u); continue; }
        wave_sync();
        uint8_t *o = dst + st.dst;
        for (uint32_t k = (uint32_t)lane * 16u; k + 16u <= st.usize; k += 1024u) st16u(o + k, *(const u32x4 *)(s_out + k));
        const uint32_t t0 = st.usize & ~15u;
        if (t0 + (uint32_t)lane < st.usize) o[t0 + lane] = s_out[t0 + lane];
    }
}

// ---- the per-block filters, undone ----
// byte shuffle: block bytes [plane 0 | plane 1 | ...] -> elements; one thread per element (typesize 4 / 8: one store)
__global__ void k_cb_unshuffle(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t nbytes, uint32_t blocksize, uint32_t ts) {
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const uint32_t per = blocksize / ts + 1u;                                          // threads per block: elements + one for the tail
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)nblocks * per; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per), e = (uint32_t)(i % per);
        const uint32_t base = b * blocksize, bsize = nbytes - base < blocksize ? nbytes - base : blocksize, nel = bsize / ts;
        if (e < nel) {
            const uint8_t *s = src + base + e;
            uint8_t *d = dst + base + (size_t)e * ts;
            if (ts == 4u) { st4u(d, (uint32_t)s[0] | ((uint32_t)s[nel] << 8) | ((uint32_t)s[2u * nel] << 16) | ((uint32_t)s[3u * nel] << 24)); }
            else for (uint32_t j = 0; j < ts; j++) d[j] = s[(size_t)j * nel];
        } else if (e == nel) {
            for (uint32_t k = nel * ts; k < bsize; k++) dst[base + k] = src[base + k];   // the bytes behind the last whole element
        }
    }
}
// bit shuffle: one thread per group of 8 elements
__global__ void k_cb_bitunshuffle(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t nbytes, uint32_t blocksize, uint32_t ts) {
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const uint32_t per = blocksize / (8u * ts) + 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)nblocks * per; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per), g = (uint32_t)(i % per);
        const uint32_t base = b * blocksize, bsize = nbytes - base < blocksize ? nbytes - base : blocksize, nel = bsize / ts;
        if (nel % 8u != 0u) {                                                           // not filtered at all: the threads of the block copy it
            for (uint32_t k = g; k < bsize; k += per) dst[base + k] = src[base + k];
            continue;
        }
        const uint32_t ng = nel / 8u;                                                   // bytes per bit row
        if (g < ng) {
            for (uint32_t j = 0; j < ts; j++) {
                uint64_t x = 0;                                                         // byte k = row 8 j + k, elements 8 g .. 8 g + 7
                for (uint32_t k = 0; k < 8u; k++) x |= (uint64_t)src[base + (size_t)(8u * j + k) * ng + g] << (8u * k);
                // 8 x 8 bit transpose: out byte i, bit k = in byte k, bit i
                uint64_t t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x ^= t ^ (t << 7);
                t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
                t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
                for (uint32_t e = 0; e < 8u; e++) dst[base + (size_t)(8u * g + e) * ts + j] = (uint8_t)(x >> (8u * e));
            }
        } else if (g == ng) {
            for (uint32_t k = nel * ts; k < bsize; k++) dst[base + k] = src[base + k];
        }
    }
}

// 8 x 8 bit transpose: out byte i, bit k = in byte k, bit i
__device__ __forceinline__ uint64_t cb_transpose8(uint64_t x) {
    uint64_t t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x ^= t ^ (t << 7);
    t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
    t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
    return x;
}
// typesize 4, whole blocks of a multiple of 32 elements (what every writer's block size is): one thread per 32 elements -- 32 dword
// loads (4 groups of every bit row at once), 8 x 16-byte stores -- instead of one byte per access
template <bool FWD>
__global__ void k_cb_bitshuffle4_fast(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t nfull, uint32_t blocksize) {
    const uint32_t ng = blocksize / 32u, per = ng / 4u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)nfull * per; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per), q = (uint32_t)(i % per);
        const size_t base = (size_t)b * blocksize;
        if (!FWD) {
            uint64_t x[4][4];                                                   // [byte j][group]: byte e = byte j of element 8 (4 q + group) + e
#pragma unroll
            for (int j = 0; j < 4; j++) {
                uint32_t r[8];
#pragma unroll
                for (int k = 0; k < 8; k++) r[k] = ld4u(src + base + (size_t)(8 * j + k) * ng + 4u * q);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    uint64_t v = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) v |= (uint64_t)((r[k] >> (8 * g)) & 255u) << (8 * k);
                    x[j][g] = cb_transpose8(v);
                }
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                uint32_t w[8];
#pragma unroll
                for (int e = 0; e < 8; e++)
                    w[e] = (uint32_t)((x[0][g] >> (8 * e)) & 255u) | ((uint32_t)((x[1][g] >> (8 * e)) & 255u) << 8) |
                           ((uint32_t)((x[2][g] >> (8 * e)) & 255u) << 16) | ((uint32_t)((x[3][g] >> (8 * e)) & 255u) << 24);
                u32x4 a, bb; a.x = w[0]; a.y = w[1]; a.z = w[2]; a.w = w[3]; bb.x = w[4]; bb.y = w[5]; bb.z = w[6]; bb.w = w[7];
                st16u(dst + base + (size_t)(32u * q + 8u * g) * 4u, a);
                st16u(dst + base + (size_t)(32u * q + 8u * g) * 4u + 16u, bb);
            }
        } else {
            uint32_t r[4][8];                                                   // [byte j][bit k]: 4 bytes = groups 4 q .. 4 q + 3 of row 8 j + k
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 8; k++) r[j][k] = 0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const u32x4 a = ld16u(src + base + (size_t)(32u * q + 8u * g) * 4u), bb = ld16u(src + base + (size_t)(32u * q + 8u * g) * 4u + 16u);
                const uint32_t w[8] = {a.x, a.y, a.z, a.w, bb.x, bb.y, bb.z, bb.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint64_t v = 0;
#pragma unroll
                    for (int e = 0; e < 8; e++) v |= (uint64_t)((w[e] >> (8 * j)) & 255u) << (8 * e);
                    v = cb_transpose8(v);                                       // byte k = bit k of byte j of the 8 elements
#pragma unroll
                    for (int k = 0; k < 8; k++) r[j][k] |= (uint32_t)((v >> (8 * k)) & 255u) << (8 * g);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int k = 0; k < 8; k++) st4u(dst + base + (size_t)(8 * j + k) * ng + 4u * q, r[j][k]);
        }
    }
}

__global__ void k_cb_result(const CbPlan *plan, hb_result *result, uint64_t nbytes) {
    result->flags = 1; result->total_bytes = nbytes; result->reserved = 0;
    if (plan->fail) { result->status = HB_ERR_DECOMPRESSION_FAILED; result->bytes = 0; }
    else { result->status = HB_OK; result->bytes = nbytes; }
}
__global__ void k_cb_init(CbPlan *plan) { plan->fail = 0; }

// =====================================================================================================================
// Writing the format.  A stream has to be ONE LZ4 block, and this library's encoder makes blocks of 4 KiB chunks that are
// stitched afterwards -- so the frame is laid out such that nothing needs stitching: blocksize = 4096 x typesize when the
// block is split (typesize <= 16 and a filter is on), else 4096 with the not-split flag; then every stream is exactly one
// matcher chunk, encoded as a block of its own (match_chunk MODE 2).  The filters run per block first (k_cb_shuffle /
// k_cb_bitshuffle, the mirrors of the kernels above), a stream that does not shrink is stored (cbytes == its size, like
// blosc_c), the last, shorter block is one stored stream.  Ratio = the chunk-local encoder's + 4 bytes per stream.
// =====================================================================================================================
struct CbEncPlan { uint32_t total, pad[3]; };

__global__ void k_cb_shuffle(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t nbytes, uint32_t blocksize, uint32_t ts) {
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const uint32_t per = blocksize / ts + 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)nblocks * per; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per), e = (uint32_t)(i % per);
        const uint32_t base = b * blocksize, bsize = nbytes - base < blocksize ? nbytes - base : blocksize, nel = bsize / ts;
        if (e < nel) {
            const uint8_t *s = src + base + (size_t)e * ts;
            uint8_t *d = dst + base + e;
            for (uint32_t j = 0; j < ts; j++) d[(size_t)j * nel] = s[j];
        } else if (e == nel) {
            for (uint32_t k = nel * ts; k < bsize; k++) dst[base + k] = src[base + k];
        }
    }
}
__global__ void k_cb_bitshuffle(uint8_t *__restrict__ dst, const uint8_t *__restrict__ src, uint32_t nbytes, uint32_t blocksize, uint32_t ts) {
    const uint32_t nblocks = (nbytes + blocksize - 1) / blocksize;
    const uint32_t per = blocksize / (8u * ts) + 1u;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (uint64_t)nblocks * per; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t b = (uint32_t)(i / per), g = (uint32_t)(i % per);
        const uint32_t base = b * blocksize, bsize = nbytes - base < blocksize ? nbytes - base : blocksize, nel = bsize / ts;
        if (nel % 8u != 0u) { for (uint32_t k = g; k < bsize; k += per) dst[base + k] = src[base + k]; continue; }
        const uint32_t ng = nel / 8u;
        if (g < ng) {
            for (uint32_t j = 0; j < ts; j++) {
                uint64_t x = 0;                                                         // byte e = byte j of element 8 g + e
                for (uint32_t e = 0; e < 8u; e++) x |= (uint64_t)src[base + (size_t)(8u * g + e) * ts + j] << (8u * e);
                uint64_t t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x ^= t ^ (t << 7);
                t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x ^= t ^ (t << 14);
                t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x ^= t ^ (t << 28);
                for (uint32_t k = 0; k < 8u; k++) dst[base + (size_t)(8u * j + k) * ng + g] = (uint8_t)(x >> (8u * k));
            }
        } else if (g == ng) {
            for (uint32_t k = nel * ts; k < bsize; k++) dst[base + k] = src[base + k];
        }
    }
}

struct CbChunkDesc { uint32_t lead, enc_len, last_end, mcode0; };       // = ChunkDesc of hb_lz4_enc.hip
// bytes of stream c in the frame: 4 + the block, or 4 + the chunk itself when the block is no smaller
__device__ __forceinline__ uint32_t cb_stream_bytes(const CbChunkDesc &d) { return 4u + (d.enc_len < HB_CHUNK ? d.enc_len : HB_CHUNK); }
// exclusive prefix sum of the stream sizes, 1024 chunks per tile: tile sums, one workgroup over the tiles, offsets
// (fused_nblk != 0: the descriptors / records come from the fused shuffle + match kernel, stream (block b, plane j) at index j * nblk + b)
__device__ __forceinline__ uint32_t cb_desc_index(uint32_t c, uint32_t nsplit, uint32_t fused_nblk) { return fused_nblk ? (c % nsplit) * fused_nblk + c / nsplit : c; }
__global__ __launch_bounds__(256) void k_cbe_tiles(const CbChunkDesc *__restrict__ desc, uint32_t nchunks, uint32_t nsplit, uint32_t fused_nblk, uint32_t *__restrict__ tile_sum) {
    __shared__ uint32_t s[256];
    const uint32_t t0 = blockIdx.x * 1024u;
    uint32_t sum = 0;
    for (uint32_t k = 0; k < 4u; k++) { const uint32_t c = t0 + k * 256u + threadIdx.x; if (c < nchunks) sum += cb_stream_bytes(desc[cb_desc_index(c, nsplit, fused_nblk)]); }
    s[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if ((int)threadIdx.x < d) s[threadIdx.x] += s[threadIdx.x + d]; __syncthreads(); }
    if (threadIdx.x == 0) tile_sum[blockIdx.x] = s[0];
}
__global__ __launch_bounds__(1024) void k_cbe_scan(uint32_t *tile_sum, uint32_t ntiles, CbEncPlan *plan) {
    __shared__ uint32_t s[1024];
    const int t = threadIdx.x;
    // (ntiles <= 1024: inputs below 4 GiB)
    const uint32_t v = (uint32_t)t < ntiles ? tile_sum[t] : 0u;
    s[t] = v;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const uint32_t y = t >= d ? s[t - d] : 0u; __syncthreads(); s[t] += y; __syncthreads(); }
    if ((uint32_t)t < ntiles) tile_sum[t] = s[t] - v;
    if (t == 1023) plan->total = s[1023];
}
// one wavefront per chunk writes { cbytes, block or chunk }; the first chunk of a block also writes the block's bstarts entry.
// 1024 chunks per workgroup of 16 wavefronts: latency-bound small copies, so as many of them in flight as the chip takes.
__global__ __launch_bounds__(1024) void k_cbe_pack(const CbChunkDesc *__restrict__ desc, const uint8_t *__restrict__ records, const uint8_t *__restrict__ filtered,
                                                   const uint32_t *__restrict__ tile_off, uint32_t nchunks, uint32_t nsplit, uint32_t fused_nblk, uint32_t data0, uint8_t *__restrict__ frame) {
    __shared__ uint32_t s[1024];
    __shared__ uint32_t s_off[1024];
    const uint32_t t0 = blockIdx.x * 1024u;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const uint32_t c_mine = t0 + (uint32_t)t;
    const uint32_t mine = c_mine < nchunks ? cb_stream_bytes(desc[cb_desc_index(c_mine, nsplit, fused_nblk)]) : 0u;
    s[t] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const uint32_t y = t >= d ? s[t - d] : 0u; __syncthreads(); s[t] += y; __syncthreads(); }
    s_off[t] = data0 + tile_off[blockIdx.x] + s[t] - mine;
    __syncthreads();
    for (uint32_t k = w; k < 1024u; k += 16u) {
        const uint32_t c = t0 + k;
        if (c >= nchunks) break;
        const uint32_t ci = cb_desc_index(c, nsplit, fused_nblk);
        const CbChunkDesc d = desc[ci];
        const uint32_t at = s_off[k];
        const bool stored = d.enc_len >= HB_CHUNK;
        const uint32_t cb = stored ? HB_CHUNK : d.enc_len;
        if (lane < 4) frame[at + lane] = (uint8_t)(cb >> (8 * lane));
        if (lane == 0 && c % nsplit == 0u) { const uint32_t b = c / nsplit; for (int q = 0; q < 4; q++) frame[16u + 4u * b + q] = (uint8_t)(at >> (8 * q)); }
        if (stored && fused_nblk) {                                    // no filtered buffer: plane j of element block b, gathered from the input
            const uint8_t *e = filtered + (size_t)(c / nsplit) * HB_CHUNK * nsplit;           // (16-byte aligned: the fused path asks for it)
            const uint32_t j = c % nsplit, per = 16u / nsplit;                                // nsplit = typesize = 2 / 4 / 8: elements per 16 bytes
            const uint32_t nv = HB_CHUNK * nsplit / 16u;                                       // 512 / 1024 / 2048 vectors: 8 / 16 / 32 per lane
            for (uint32_t i0 = lane; i0 < nv; i0 += 64u * 8u) {                                // 8 loads in flight per lane
                u32x4 v[8];
#pragma unroll
                for (int q = 0; q < 8; q++) v[q] = *(const u32x4 *)(e + 16u * (size_t)(i0 + 64u * q));
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const uint32_t i = i0 + 64u * q;
                    const uint32_t wv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    uint64_t o = 0;
                    for (uint32_t kk = 0; kk < per; kk++) { const uint32_t byte = kk * nsplit + j; o |= (uint64_t)((wv[byte >> 2] >> (8u * (byte & 3u))) & 255u) << (8u * kk); }
                    uint8_t *dp = frame + at + 4u + i * per;
                    if (per == 8u) st8u(dp, o); else if (per == 4u) st4u(dp, (uint32_t)o); else { dp[0] = (uint8_t)o; dp[1] = (uint8_t)(o >> 8); }
                }
            }
        } else wave_copy_g2g(frame + at + 4u, stored ? filtered + (size_t)c * HB_CHUNK : records + (size_t)ci * HB_RSTRIDE, cb, lane);
    }
}
// header, the last (shorter) block as one stored stream, the result
__global__ __launch_bounds__(64) void k_cbe_finish(const CbEncPlan *plan, const uint8_t *__restrict__ filtered, uint32_t nbytes, uint32_t blocksize, uint32_t ts, uint32_t flags,
                                                   uint32_t nfull_blocks, uint32_t data0, uint8_t *__restrict__ frame, uint64_t cap, hb_result *result) {
    const int lane = threadIdx.x;
    const uint32_t leftover = nbytes - nfull_blocks * blocksize;
    const uint32_t at = data0 + plan->total;
    const uint64_t cbytes = (uint64_t)at + (leftover ? 4u + leftover : 0u);
    if (cbytes > cap || cbytes > 0xFFFFFFFFull) { if (lane == 0) { result->status = HB_ERR_SHORT_BUFFER; result->bytes = 0; result->total_bytes = 0; result->flags = 0; result->reserved = 0; } return; }
    if (leftover) {
        if (lane < 4) { frame[at + lane] = (uint8_t)(leftover >> (8 * lane)); frame[16u + 4u * nfull_blocks + lane] = (uint8_t)(at >> (8 * lane)); }
        wave_copy_g2g(frame + at + 4u, filtered + (size_t)nfull_blocks * blocksize, leftover, lane);
    }
    if (lane == 0) {
        frame[0] = 2; frame[1] = 1; frame[2] = (uint8_t)flags; frame[3] = (uint8_t)ts;                  // BLOSC_VERSION_FORMAT, LZ4 version format
        for (int q = 0; q < 4; q++) { frame[4 + q] = (uint8_t)(nbytes >> (8 * q)); frame[8 + q] = (uint8_t)(blocksize >> (8 * q)); frame[12 + q] = (uint8_t)((uint32_t)cbytes >> (8 * q)); }
        result->status = HB_OK; result->bytes = cbytes; result->total_bytes = cbytes; result->flags = 0; result->reserved = 0;
    }
}

// inputs below one chunk: a memcpyed frame (what c-blosc itself writes for buffers it cannot shrink)
__global__ void k_cbe_memcpy_header(uint8_t *frame, uint32_t nbytes, uint32_t ts, uint32_t flags, hb_result *result) {
    const uint32_t cbytes = 16u + nbytes;
    uint32_t bs = nbytes - nbytes % ts; if (bs == 0u) bs = nbytes ? 1u : 0u;
    frame[0] = 2; frame[1] = 1; frame[2] = (uint8_t)flags; frame[3] = (uint8_t)ts;
    for (int q = 0; q < 4; q++) { frame[4 + q] = (uint8_t)(nbytes >> (8 * q)); frame[8 + q] = (uint8_t)(bs >> (8 * q)); frame[12 + q] = (uint8_t)(cbytes >> (8 * q)); }
    result->status = HB_OK; result->bytes = cbytes; result->total_bytes = cbytes; result->flags = 0; result->reserved = 0;
}

void hb_launch_match_selfcontained(const uint8_t *src, size_t n, void *desc, uint8_t *records, uint32_t nchunks, int accel, hipStream_t s);   // hb_lz4_enc.hip
bool hb_launch_match_fused_selfcontained(const uint8_t *src, int typesize, void *desc, uint8_t *records, uint32_t nblk, int accel, hipStream_t s);

struct CbEncLayout { size_t plan, tiles, desc, records, filtered, total; uint32_t blocksize, nsplit, nblocks, nfull, nchunks, ntiles; };
static CbEncLayout cbe_layout(size_t n, int shuffle, int typesize) {
    CbEncLayout L;
    const bool filt = (shuffle == 1 && typesize > 1) || shuffle == 2;
    L.nsplit = (filt && typesize <= 16) ? (uint32_t)typesize : 1u;
    if (n < (size_t)HB_CHUNK * L.nsplit) L.nsplit = 1u;                   // (c-blosc refuses a blocksize above nbytes)
    L.blocksize = HB_CHUNK * L.nsplit;
    L.nblocks = (uint32_t)((n + L.blocksize - 1) / L.blocksize);
    L.nfull = (uint32_t)(n / L.blocksize);
    L.nchunks = L.nfull * L.nsplit;
    L.ntiles = (L.nchunks + 1023u) / 1024u;
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += cb_align(b); return at; };
    L.plan = take(sizeof(CbEncPlan));
    L.tiles = take((size_t)(L.ntiles + 1) * 4);
    L.desc = take((size_t)L.nchunks * sizeof(CbChunkDesc));
    L.records = take((size_t)L.nchunks * HB_RSTRIDE + 256);
    L.filtered = take(n + 64);
    L.total = o;
    return L;
}

// launch shape of the 16-byte-per-lane filter kernel (k_cb_bitshuffle4_fast): every thread gets ONE work item, no grid-stride passes (round 3,
// tools/lab/filter_lab.hip: streaming kernels on this chip want the dispatcher to deal out the work: 0.76 -> 0.55 ms).  The byte-gathering
// kernels (k_cb_unshuffle & co.) are the opposite case: the same change took them from 0.61 to 0.88 ms -- they keep their 2048 workgroups.
static inline unsigned cb_grid(uint64_t items) { const uint64_t g = (items + 255) / 256; return (unsigned)(g < 1 ? 1 : (g > (1u << 24) ? (1u << 24) : g)); }

extern "C" {

// header fields of a C-Blosc-1 frame (host side); HB_OK or the error a malformed header gets
int hb_cblosc_parse_header(const void *frame, size_t n, hb_cblosc_header *out) {
    if (!frame || !out) return HB_ERR_BAD_ARG;
    if (n < 16) return HB_ERR_INVALID_HEADER;
    const uint8_t *f = (const uint8_t *)frame;
    auto rd = [&](int at) { return (uint32_t)f[at] | ((uint32_t)f[at + 1] << 8) | ((uint32_t)f[at + 2] << 16) | ((uint32_t)f[at + 3] << 24); };
    out->version = f[0]; out->versionlz = f[1]; out->flags = f[2]; out->typesize = f[3];
    out->nbytes = rd(4); out->blocksize = rd(8); out->cbytes = rd(12);
    out->codec_format = f[2] >> 5;
    if (out->version != 2) return HB_ERR_INVALID_VERSION;                              // BLOSC_VERSION_FORMAT
    if (out->typesize == 0) return HB_ERR_INVALID_HEADER;
    if (out->cbytes < 16 || out->cbytes > n) return HB_ERR_INVALID_DATA;
    if (out->nbytes && out->blocksize == 0) return HB_ERR_INVALID_HEADER;
    return HB_OK;
}

size_t hb_cblosc_decompress_workspace(size_t nbytes, size_t blocksize, size_t typesize) {
    const size_t nblocks = blocksize ? (nbytes + blocksize - 1) / blocksize : 0;
    const size_t nsplit = (typesize >= 1 && typesize <= 16 && blocksize / typesize >= 128) ? typesize : 1;      // cb_nsplit() without the flag: the upper bound
    return 256 + cb_align(nblocks * nsplit * sizeof(CbStream)) + cb_align(nbytes + 64);
}

// d_frame: the frame in device memory (n bytes available), d_dst: hdr.nbytes bytes.  Asynchronous on `stream`; *d_result says how it went.
int hb_cblosc_decompress_dev(const hb_cblosc_header *hdr, const void *d_frame, size_t n, void *d_dst, size_t cap, void *d_work, size_t work_bytes,
                             hb_result *d_result, void *stream) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (!hdr || !d_frame || (!d_dst && cap) || !d_work || ((uintptr_t)d_work & 255u) || !d_result) return HB_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nbytes = hdr->nbytes, blocksize = hdr->blocksize, ts = hdr->typesize, flags = hdr->flags;
    // the record may come from a caller that ignored hb_cblosc_parse_header's return value, or built it itself: repeat its checks
    // before anything divides by a field
    if (hdr->version != 2) return HB_ERR_INVALID_VERSION;
    if (ts == 0u || (nbytes && blocksize == 0u)) return HB_ERR_INVALID_HEADER;
    if (hdr->cbytes > n || hdr->cbytes < 16) return HB_ERR_INVALID_DATA;
    if (nbytes > cap) return HB_ERR_SHORT_BUFFER;
    if (work_bytes < hb_cblosc_decompress_workspace(nbytes, blocksize, ts)) return HB_ERR_SHORT_BUFFER;
    uint8_t *w = (uint8_t *)d_work;
    CbPlan *plan = (CbPlan *)w;
    hipLaunchKernelGGL(k_cb_init, dim3(1), dim3(1), 0, s, plan);
    if (nbytes == 0) { hipLaunchKernelGGL(k_cb_result, dim3(1), dim3(1), 0, s, plan, d_result, (uint64_t)0); return HB_OK; }
    if (flags & CB_FLAG_MEMCPY) {                                       // blosc.c: "memcpyed": the raw bytes follow the header
        if ((uint64_t)hdr->cbytes < 16ull + nbytes) return HB_ERR_INVALID_DATA;
        HB_HIP_TRY(hipMemcpyAsync(d_dst, (const uint8_t *)d_frame + 16, nbytes, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_cb_result, dim3(1), dim3(1), 0, s, plan, d_result, (uint64_t)nbytes);
        return HB_OK;
    }
    if (hdr->codec_format != 1) return HB_ERR_INVALID_CODEC;            // lz4 / lz4hc only
    const uint32_t nblocks = (uint32_t)(((uint64_t)nbytes + blocksize - 1) / blocksize);
    if (16ull + 4ull * nblocks > hdr->cbytes) return HB_ERR_INVALID_DATA;
    if (blocksize < ts) return HB_ERR_INVALID_DATA;                     // (c-blosc never writes it; a stream would hold less than one element)
    const uint32_t nsplit = cb_nsplit(flags, ts, blocksize);
    CbStream *streams = (CbStream *)(w + 256);
    uint8_t *staged = w + 256 + cb_align((size_t)nblocks * nsplit * sizeof(CbStream));
    // blosc_d: the byte shuffle counts for typesize > 1 only (and comes first), the bit shuffle for any typesize
    const bool unshuf = (flags & CB_FLAG_SHUFFLE) && ts > 1, unbit = !unshuf && (flags & CB_FLAG_BITSHUFFLE);
    const bool filtered = unshuf || unbit;
    uint8_t *target = filtered ? staged : (uint8_t *)d_dst;
    const uint32_t nstreams = nblocks * nsplit;
    hb_prof_begin("k_cb_plan", s);
    hipLaunchKernelGGL(k_cb_plan, dim3((nblocks + 63) / 64), dim3(64), 0, s, (const uint8_t *)d_frame, (uint64_t)n, nbytes, blocksize, hdr->cbytes, ts, flags, plan, streams);
    const bool small = blocksize / nsplit <= HB_CHUNK;                   // streams of at most one chunk: the LDS-resident decoder takes them
    hb_prof_end(s);
    const uint32_t mgrp = (nstreams + 7u) / 8u;
    uint32_t P = mgrp / 4u + 1u;                                        // coprime to the groups of 8 streams, about a quarter turn
    for (;; P++) { uint32_t x = P, y = mgrp; while (y) { const uint32_t t = x % y; x = y; y = t; } if (x == 1u) break; }
    const unsigned grid = mgrp * 8u < 65536u ? mgrp * 8u : 65536u;
    hb_prof_begin("k_cb_decode_small", s);
    if (small) hipLaunchKernelGGL(k_cb_decode_small, dim3(grid), dim3(64), 0, s, (const uint8_t *)d_frame, plan, streams, nstreams, target, P);
    hb_prof_end(s);
    hb_prof_begin("k_cb_decode", s);
    // (long streams: as many passes per workgroup as a block has streams, fewer while that leaves under 2048 workgroups -- with the
    // rotation by the pass number a workgroup decodes one stream of each plane, and all workgroups live about equally long; see
    // k_cb_decode_small)
    unsigned gbig = grid;
    if (!small && nsplit > 1u) {
        unsigned p = nsplit;
        while (p > 1u && mgrp * 8u / p < 2048u) p >>= 1;
        gbig = (mgrp * 8u / p + 7u) / 8u * 8u;
    }
    hipLaunchKernelGGL(k_cb_decode, dim3(gbig), dim3(64), 0, s, (const uint8_t *)d_frame, plan, streams, nstreams, target, small ? 1 : 0, P);
    hb_prof_end(s);
    if (filtered) {
        hb_prof_begin("k_cb_unfilter", s);
        if (unshuf)
            {
            // whole blocks of whole 1024-element tiles: the tile kernels of hb_filters.hip, block by block (6 TB/s instead of the 3.5 of the
            // byte-gathering kernel); a last, shorter block -- or any other shape -- the plain way
            const uint32_t nfull = nbytes / blocksize, tail = nbytes - nfull * blocksize;
            if (hb_launch_shuffle_blocks(true, (uint8_t *)d_dst, staged, nfull, blocksize, (int)ts, s)) {
                if (tail) hipLaunchKernelGGL(k_cb_unshuffle, dim3(64), dim3(256), 0, s, (uint8_t *)d_dst + (size_t)nfull * blocksize, staged + (size_t)nfull * blocksize, tail, blocksize, ts);
            } else hipLaunchKernelGGL(k_cb_unshuffle, dim3(2048), dim3(256), 0, s, (uint8_t *)d_dst, staged, nbytes, blocksize, ts);
        }
        else if (ts == 4u && blocksize % 512u == 0u && nbytes >= blocksize) {       // whole blocks the fast way, a last shorter one the plain way
            const uint32_t nfull = nbytes / blocksize, tail = nbytes - nfull * blocksize;
            hipLaunchKernelGGL(k_cb_bitshuffle4_fast<false>, dim3(cb_grid((uint64_t)nfull * (blocksize / 128u))), dim3(256), 0, s, (uint8_t *)d_dst, staged, nfull, blocksize);
            if (tail) hipLaunchKernelGGL(k_cb_bitunshuffle, dim3(64), dim3(256), 0, s, (uint8_t *)d_dst + (size_t)nfull * blocksize, staged + (size_t)nfull * blocksize, tail, blocksize, ts);
        } else
            hipLaunchKernelGGL(k_cb_bitunshuffle, dim3(2048), dim3(256), 0, s, (uint8_t *)d_dst, staged, nbytes, blocksize, ts);
        hb_prof_end(s);
    }
    hipLaunchKernelGGL(k_cb_result, dim3(1), dim3(1), 0, s, plan, d_result, (uint64_t)nbytes);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}


size_t hb_cblosc_bound(size_t n, int typesize) {
    const size_t ts = typesize > 0 ? (size_t)typesize : 1;
    return 16 + 4 * (n / HB_CHUNK + 2) + n + 4 * (n / HB_CHUNK + ts + 2) + 64;
}
size_t hb_cblosc_compress_workspace(size_t n, int shuffle, int typesize) { return cbe_layout(n, shuffle, typesize > 0 ? typesize : 1).total; }

// shuffle: 0 none, 1 byte shuffle, 2 bit shuffle (c-blosc's BLOSC_NOSHUFFLE / BLOSC_SHUFFLE / BLOSC_BITSHUFFLE).  Asynchronous on `stream`.
int hb_cblosc_compress_dev(const void *d_src, size_t n, void *d_frame, size_t cap, int shuffle, int typesize, void *d_work, size_t work_bytes,
                           hb_result *d_result, void *stream) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if ((!d_src && n) || !d_frame || !d_work || ((uintptr_t)d_work & 255u) || !d_result) return HB_ERR_BAD_ARG;
    if (typesize < 1 || typesize > 255 || shuffle < 0 || shuffle > 2) return HB_ERR_BAD_ARG;
    if (n > 0x7FFFFFFFull - 64u * 1024u * 1024u) return HB_ERR_DATA_TOO_LARGE;           // (c-blosc: BLOSC_MAX_BUFFERSIZE = INT_MAX - 16)
    if (cap < hb_cblosc_bound(n, typesize)) return HB_ERR_SHORT_BUFFER;
    const CbEncLayout L = cbe_layout(n, shuffle, typesize);
    if (work_bytes < L.total) return HB_ERR_SHORT_BUFFER;
    hipStream_t s = (hipStream_t)stream;
    uint8_t *w = (uint8_t *)d_work;
    CbEncPlan *plan = (CbEncPlan *)(w + L.plan);
    uint32_t *tiles = (uint32_t *)(w + L.tiles);
    CbChunkDesc *desc = (CbChunkDesc *)(w + L.desc);
    uint8_t *records = w + L.records, *filtered = w + L.filtered;
    const bool unshuf = shuffle == 1 && typesize > 1, bits = shuffle == 2;
    const uint32_t flags = (unshuf ? CB_FLAG_SHUFFLE : 0u) | (bits ? CB_FLAG_BITSHUFFLE : 0u) | (L.nsplit == 1u ? CB_FLAG_DONTSPLIT : 0u) | (1u << 5);
    if (n < HB_CHUNK) {
        if (n) HB_HIP_TRY(hipMemcpyAsync((uint8_t *)d_frame + 16, d_src, n, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_cbe_memcpy_header, dim3(1), dim3(1), 0, s, (uint8_t *)d_frame, (uint32_t)n, (uint32_t)typesize, flags | CB_FLAG_MEMCPY | CB_FLAG_DONTSPLIT, d_result);
        HB_HIP_TRY(hipGetLastError());
        return HB_OK;
    }
    const uint8_t *fsrc = (const uint8_t *)d_src;
    // byte shuffle with typesize 2 / 4 / 8 and split blocks: a C-Blosc block of 4096 elements IS the unit of the fused shuffle + match
    // kernel (hb_lz4_enc.hip), only the order of the chunks differs -- no filtered buffer, except for the last, shorter block
    const bool fuse = unshuf && L.nsplit == (uint32_t)typesize && (typesize == 2 || typesize == 4 || typesize == 8) && L.nfull != 0u && ((uintptr_t)d_src & 15u) == 0;
    const uint32_t tail = (uint32_t)(n - (size_t)L.nfull * L.blocksize);
    hb_prof_begin("k_cb_filter", s);
    if (fuse) {
        if (tail) hipLaunchKernelGGL(k_cb_shuffle, dim3(64), dim3(256), 0, s, filtered + (size_t)L.nfull * L.blocksize, (const uint8_t *)d_src + (size_t)L.nfull * L.blocksize, tail, L.blocksize, (uint32_t)typesize);
    } else if (unshuf || bits) {
        if (unshuf) {
            if (hb_launch_shuffle_blocks(false, filtered, (const uint8_t *)d_src, L.nfull, L.blocksize, typesize, s)) {
                if (tail) hipLaunchKernelGGL(k_cb_shuffle, dim3(64), dim3(256), 0, s, filtered + (size_t)L.nfull * L.blocksize, (const uint8_t *)d_src + (size_t)L.nfull * L.blocksize, tail, L.blocksize, (uint32_t)typesize);
            } else hipLaunchKernelGGL(k_cb_shuffle, dim3(2048), dim3(256), 0, s, filtered, (const uint8_t *)d_src, (uint32_t)n, L.blocksize, (uint32_t)typesize);
        }
        else if (typesize == 4 && L.blocksize % 512u == 0u && L.nfull) {
            hipLaunchKernelGGL(k_cb_bitshuffle4_fast<true>, dim3(cb_grid((uint64_t)L.nfull * (L.blocksize / 128u))), dim3(256), 0, s, filtered, (const uint8_t *)d_src, L.nfull, L.blocksize);
            if (tail) hipLaunchKernelGGL(k_cb_bitshuffle, dim3(64), dim3(256), 0, s, filtered + (size_t)L.nfull * L.blocksize, (const uint8_t *)d_src + (size_t)L.nfull * L.blocksize, tail, L.blocksize, (uint32_t)typesize);
        } else hipLaunchKernelGGL(k_cb_bitshuffle, dim3(2048), dim3(256), 0, s, filtered, (const uint8_t *)d_src, (uint32_t)n, L.blocksize, (uint32_t)typesize);
        fsrc = filtered;
    }
    hb_prof_end(s);
    HB_HIP_TRY(hipMemsetAsync(plan, 0, sizeof(CbEncPlan), s));
    const uint32_t fused_nblk = fuse ? L.nfull : 0u;
    if (L.nchunks) {
        hb_prof_begin("k_match", s);
        if (fuse) hb_launch_match_fused_selfcontained((const uint8_t *)d_src, typesize, desc, records, L.nfull, 64, s);
        else hb_launch_match_selfcontained(fsrc, (size_t)L.nchunks * HB_CHUNK, desc, records, L.nchunks, 64, s);
        hb_prof_end(s);
        hb_prof_begin("k_cbe_pack", s);
        hipLaunchKernelGGL(k_cbe_tiles, dim3(L.ntiles), dim3(256), 0, s, desc, L.nchunks, L.nsplit, fused_nblk, tiles);
        hipLaunchKernelGGL(k_cbe_scan, dim3(1), dim3(1024), 0, s, tiles, L.ntiles, plan);
        hipLaunchKernelGGL(k_cbe_pack, dim3(L.ntiles), dim3(1024), 0, s, desc, records, fsrc, tiles, L.nchunks, L.nsplit, fused_nblk, 16u + 4u * L.nblocks, (uint8_t *)d_frame);
        hb_prof_end(s);
    }
    hipLaunchKernelGGL(k_cbe_finish, dim3(1), dim3(64), 0, s, plan, fuse ? (const uint8_t *)filtered : fsrc, (uint32_t)n, L.blocksize, (uint32_t)typesize, flags, L.nfull, 16u + 4u * L.nblocks, (uint8_t *)d_frame,
                       (uint64_t)cap, d_result);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
}  // extern "C"
