// hb_host.cpp — the entry points of include/hipblosc.h that never touch the device: header (de)serialisation, size bounds,
// error strings.  Plain C++ (compiled by g++ into libhipblosc.so, and by tests/tools/host_asan_check.cpp with
// -fsanitize=address,undefined: sanitizers run on the CPU build only).
#include "../../include/hipblosc.h"
#include "hb_format.h"

size_t hb_lz4_index_bound(size_t n) {
    return HB_IDX_HDR_BYTES + (size_t)HB_IDX_ENTRY * ((n + HB_CHUNK - 1) / HB_CHUNK + 1);
}

extern "C" {

const char *hb_version(void) { return HB_VERSION_STRING; }

const char *hb_strerror(int code) {
    switch (code) {
    case HB_OK: return "ok";
    case HB_ERR_INVALID_DATA: return "blosc: invalid compressed data";            // blosc.go:127
    case HB_ERR_INVALID_HEADER: return "blosc: invalid header";                   // blosc.go:130
    case HB_ERR_INVALID_VERSION: return "blosc: unsupported format version";      // blosc.go:133
    case HB_ERR_INVALID_CODEC: return "blosc: unsupported codec";                 // blosc.go:136
    case HB_ERR_SIZE_MISMATCH: return "blosc: decompressed size mismatch";        // blosc.go:139
    case HB_ERR_DATA_TOO_LARGE: return "blosc: data too large";                   // blosc.go:142
    case HB_ERR_COMPRESSION_FAILED: return "blosc: compression failed";           // blosc.go:145
    case HB_ERR_DECOMPRESSION_FAILED: return "blosc: decompression failed";       // blosc.go:148
    case HB_ERR_NO_DEVICE: return "hipblosc: no HIP device";
    case HB_ERR_HIP: return "hipblosc: HIP runtime error";
    case HB_ERR_BAD_ARG: return "hipblosc: bad argument";
    case HB_ERR_SHORT_BUFFER: return "hipblosc: destination or workspace too small";
    default: return "hipblosc: unknown error";
    }
}

size_t hb_lz4_bound(size_t n) { return n + n / 255 + 16; }           // codec.go:65
size_t hb_index_bound(size_t n) { return hb_lz4_index_bound(n); }

static inline uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

int hb_parse_header(const void *frame, size_t n, hb_header *h) {      // blosc.go:165-185
    if (!h || (!frame && n)) return HB_ERR_BAD_ARG;
    if (n < HB_HEADER_SIZE) return HB_ERR_INVALID_HEADER;
    const uint8_t *f = (const uint8_t *)frame;
    h->version = f[0]; h->codec = f[1]; h->flags = f[2]; h->typesize = f[3];
    h->nbytes = le32(f + 4); h->blocksize = le32(f + 8); h->cbytes = le32(f + 12);
    if (h->version != HB_FORMAT_VERSION) return HB_ERR_INVALID_VERSION;
    return HB_OK;
}

void hb_header_bytes(const hb_header *h, void *out16) {               // blosc.go:188-198
    uint8_t *o = (uint8_t *)out16;
    o[0] = h->version; o[1] = h->codec; o[2] = h->flags; o[3] = h->typesize;
    put32(o + 4, h->nbytes); put32(o + 8, h->blocksize); put32(o + 12, h->cbytes);
}

size_t hb_frame_bound(size_t n) { return HB_HEADER_SIZE + hb_lz4_bound(n) + 8 + hb_lz4_index_bound(n); }
}  // extern "C"
