// hb_lz4.h — internal interface of the device LZ4 block codec (hb_lz4_enc.hip / hb_lz4_dec.hip).
#pragma once
#include "hb_common.h"

#include "hb_format.h"

#define HB_OPT_INTERNAL_BLOCK 0x80000000u   // hb_codec_compress: no memcpy rule -- the payload is always the codec's block

struct hb_enc_args {
    const uint8_t *src; size_t n;        // bytes to encode (already filtered)
    uint8_t *dst; size_t cap;            // frame != 0: frame start (payload at +16); else the block itself
    uint8_t *index;                      // frame == 0: optional external index buffer
    uint8_t *work; hb_result *result;
    int frame, codec, shuffle, typesize;
    int level;                           // Options.Level after clamping (blosc.go:277-282): search depth / skip policy, see enc_policy()
    unsigned opts;
    const uint8_t *memcpy_src;           // what a memcpy frame stores (filtered bytes, or raw with HB_OPT_REFERENCE_MEMCPY);
                                         // NULL with fused_ts: the payload is shuffled in place by a gated filter launch
    int fused_ts;                        // != 0: src is the UN-filtered input, byte shuffle with this typesize is fused into the matcher
    int fused_bits;                      // 4: src is the UN-filtered input, bitshuffle (typesize 4) is fused into the matcher
};

struct hb_dec_args {
    const uint8_t *src; size_t n;        // LZ4 block (frame payload)
    uint8_t *dst; size_t cap;
    const uint8_t *index; size_t index_bytes;
    uint8_t *work; hb_result *result;
    int frame; uint32_t expect;          // frame != 0: decoded length must equal expect (blosc.go:429-431)
    int memcpy_payload;                  // blosc.go:398-400
    int fused_bitunshuffle4;             // the frame is bitshuffled with typesize 4 and the un-filter runs inside the decoder
    int fused_unshuffle_ts;              // != 0: the frame is byte-shuffled with this typesize and the un-shuffle runs inside the decoder
    uint8_t *staged;                     // fused un-filter: where the serial fallback puts the still-filtered bytes
    uint8_t *sym_work;                   // hb_lz4_sym_workspace(cap) bytes, or NULL: foreign blocks then stay with the single wavefront
};

// hb_zstd.hip: host ZSTD behind the device filter (BASELINE.json config 5)
bool hb_zstd_available();
int64_t hb_zstd_compress_frame(const void *src, size_t n, void *dst, size_t cap, int level, int shuffle, int typesize,
                               unsigned opts, int device);
int64_t hb_zstd_decompress_frame(const void *frame, const hb_header &h, void *dst, size_t cap, int typesize_override, int device);

// batches of frames in one set of launches (hb_lz4_enc.hip / hb_batch.hip)
struct hb_batch_frame { const uint8_t *src; size_t n; uint8_t *dst; size_t cap; hb_result *result; };
size_t hb_lz4_enc_batch_workspace(int nframes, const size_t *n, int typesize);
int hb_launch_lz4_encode_batch(int nframes, const hb_batch_frame *fr, int codec, int level, int shuffle, int typesize, unsigned opts,
                               uint8_t *work, size_t work_bytes, hipStream_t s);
size_t hb_lz4_enc_workspace(size_t n);
size_t hb_lz4_dec_workspace(size_t n_out);
size_t hb_lz4_index_bound(size_t n);
size_t hb_lz4_sym_workspace(size_t n_out);      // hb_lz4_sym.hip: scratch of the symbolic decoder of foreign blocks (~2 bytes per output byte)
int hb_launch_lz4_encode(const hb_enc_args &a, hipStream_t s);
int hb_launch_lz4_decode(const hb_dec_args &a, hipStream_t s);
// hb_snappy.hip: Snappy block decoder (codec.go:237-244); same argument record, the fused un-filter fields are ignored
int hb_launch_snappy_decode(const hb_dec_args &a, hipStream_t s);
static inline bool hb_device_codec(int codec) { return codec == HB_LZ4 || codec == HB_LZ4HC || codec == HB_SNAPPY; }
// An LZ4 block without a restart index is worth the token discovery (hb_lz4_region.hip: ~1-2 ms of fixed cost) instead of the single
// wavefront (0.15-1.2 GB/s of output) when it is long, or short but highly compressed: 16 MiB of zeros are 64 KB of stream and 14 ms
// on one wavefront.  Every place that sizes a workspace for such a block asks this (payload = stream bytes, nbytes = decoded size).
static inline bool hb_indexless_parallel(size_t payload, size_t nbytes) {
    return payload >= (256u << 10) || (payload >= (16u << 10) && nbytes >= (2u << 20));
}

// ---- small device helpers shared by encoder and decoder ----
__device__ __forceinline__ uint32_t lz4_ext_bytes(uint32_t x) { return x < 15u ? 0u : 1u + (x - 15u) / 255u; }
// the same for x < 65536 + 15: n / 255 == (n * 0x8081) >> 23 for n < 65536 -- a 24-bit multiply (full rate) instead of v_mul_hi_u32 (quarter rate)
__device__ __forceinline__ uint32_t lz4_ext_bytes16(uint32_t x) { return x < 15u ? 0u : 1u + ((uint32_t)__umul24(x - 15u, 0x8081u) >> 23); }

// wave-cooperative byte copy, any alignment, global -> global; dst-aligned 16-byte stores in the body
__device__ __forceinline__ void wave_copy_g2g(uint8_t *dst, const uint8_t *src, uint32_t len, int lane) {
    if (len == 0) return;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > len) head = len;
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t body = (len - head) >> 4;
    for (uint32_t i = lane; i < body; i += 64) st16u(dst + head + i * 16u, ld16u(src + head + i * 16u));
    const uint32_t done = head + body * 16u;
    if (done + lane < len) dst[done + lane] = src[done + lane];
}

// the same with streaming accesses (no reuse on either side: k_stitch reads a record once and writes the frame once)
__device__ __forceinline__ void wave_copy_g2g_nt(uint8_t *dst, const uint8_t *src, uint32_t len, int lane) {
    if (len == 0) return;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > len) head = len;
    if ((uint32_t)lane < head) dst[lane] = src[lane];
    const uint32_t body = (len - head) >> 4;
    for (uint32_t i = lane; i < body; i += 64) st16u_nt(dst + head + i * 16u, ld16u_nt(src + head + i * 16u));
    const uint32_t done = head + body * 16u;
    if (done + lane < len) dst[done + lane] = src[done + lane];
}

// wave-cooperative fill
__device__ __forceinline__ void wave_fill_g(uint8_t *dst, uint8_t val, uint32_t len, int lane) {
    if (len == 0) return;
    uint32_t head = (uint32_t)((16u - ((uintptr_t)dst & 15u)) & 15u);
    if (head > len) head = len;
    if ((uint32_t)lane < head) dst[lane] = val;
    const uint32_t body = (len - head) >> 4;
    const uint32_t w = val * 0x01010101u;
    u32x4 v; v.x = w; v.y = w; v.z = w; v.w = w;
    for (uint32_t i = lane; i < body; i += 64) st16u(dst + head + i * 16u, v);
    const uint32_t done = head + body * 16u;
    if (done + lane < len) dst[done + lane] = val;
}

// token + length-extension bytes of one LZ4 sequence header: [token][255 ...][rest]
__device__ __forceinline__ void wave_write_lit_header(uint8_t *dst, uint32_t lit, uint32_t mcode, int lane) {
    const uint32_t nb = lz4_ext_bytes(lit);
    if (lane == 0) dst[0] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | mcode);
    if (nb) {
        wave_fill_g(dst + 1, 0xFF, nb - 1, lane);
        if (lane == 0) dst[nb] = (uint8_t)((lit - 15u) - 255u * (nb - 1));
    }
}
