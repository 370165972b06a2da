// hb_zstd.hip — BASELINE.json config 5: Shuffle1 + ZSTD, "device shuffle overlapped with host ZSTD".
//
// In the reference ZSTD is a host library behind the codec seam (zstdCodec, codec.go:173-222: klauspost
// EncodeAll / DecodeAll); it stays a host codec here too.  What moves to the MI355X is the
// filter: the frame's shuffled image is produced on the device, copied back in 16 MiB slices on a HIP stream, and
// every slice is compressed by a host thread as soon as its copy has landed -- copies and compression overlap.
// Each slice becomes one zstd frame; the payload is their concatenation, which zstd.Decoder.DecodeAll (and
// ZSTD_decompress) decode as one stream, so the go-blosc frame stays decodable by the reference.
// libzstd is loaded at run time (dlopen); without it the codec reports HB_ERR_INVALID_CODEC, like an
// unregistered codec in the reference (blosc.go:322-325).
#include "hb_common.h"
#include "hb_lz4.h"

#include <dlfcn.h>
#include <atomic>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

struct ZstdApi {
    size_t (*compress)(void *, size_t, const void *, size_t, int);
    size_t (*bound)(size_t);
    size_t (*decompress)(void *, size_t, const void *, size_t);
    unsigned (*is_error)(size_t);
    size_t (*frame_csize)(const void *, size_t);
    unsigned long long (*frame_content)(const void *, size_t);
    int (*error_code)(size_t);           // ZSTD_getErrorCode (optional)
};

const ZstdApi *zstd_api() {
    static ZstdApi api;
    static bool ok = false;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = nullptr;
        for (const char *name : {"libzstd.so.1", "libzstd.so", "/opt/conda/lib/libzstd.so.1"}) {
            h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (h) break;
        }
        if (!h) return;
        api.compress = (decltype(api.compress))dlsym(h, "ZSTD_compress");
        api.bound = (decltype(api.bound))dlsym(h, "ZSTD_compressBound");
        api.decompress = (decltype(api.decompress))dlsym(h, "ZSTD_decompress");
        api.is_error = (decltype(api.is_error))dlsym(h, "ZSTD_isError");
        api.frame_csize = (decltype(api.frame_csize))dlsym(h, "ZSTD_findFrameCompressedSize");
        api.frame_content = (decltype(api.frame_content))dlsym(h, "ZSTD_getFrameContentSize");
        api.error_code = (decltype(api.error_code))dlsym(h, "ZSTD_getErrorCode");
        ok = api.compress && api.bound && api.decompress && api.is_error && api.frame_csize && api.frame_content;
    });
    return ok ? &api : nullptr;
}

constexpr size_t SLICE = 16u << 20;

// grow-only caches of the staging buffers (pinning 1 GiB of host memory or hipMalloc-ing it costs tens of milliseconds;
// a host-bound codec should not pay that per frame).  hb_shutdown() does not free these; they live with the process.
struct Cached { void *p; size_t bytes; int dev; bool host; };
std::mutex g_cache_mu;
std::vector<Cached> g_cache;
void *cache_get(bool host, int dev, size_t bytes, size_t *got) {
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (!bytes) bytes = 4096;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        int best = -1;
        for (int i = 0; i < (int)g_cache.size(); i++)
            if (g_cache[i].host == host && (host || g_cache[i].dev == dev) && g_cache[i].bytes >= bytes &&
                (best < 0 || g_cache[i].bytes < g_cache[best].bytes)) best = i;
        if (best >= 0) { Cached c = g_cache[best]; g_cache.erase(g_cache.begin() + best); *got = c.bytes; return c.p; }
    }
    void *p = nullptr;
    const hipError_t e = host ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
    if (e != hipSuccess) return nullptr;
    *got = bytes;
    return p;
}
void cache_put(bool host, int dev, void *p, size_t bytes) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.push_back(Cached{p, bytes, dev, host});
}

int host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 4;
    return (int)(n > 16 ? 16 : n);                       // the CPU share that goes with one GPU
}

inline void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

}  // namespace

// zstdCodec.Compress's level map (codec.go:200-213): <=2 SpeedFastest, <=4 SpeedDefault, <=6 SpeedBetterCompression,
// else SpeedBestCompression; klauspost documents these as roughly zstd levels 1, 3, 7, 11.
static int zstd_level_for(int level) { return level <= 2 ? 1 : level <= 4 ? 3 : level <= 6 ? 7 : 11; }

bool hb_zstd_available() { return zstd_api() != nullptr; }

// compressBackend (blosc.go:320-374) with codec == ZSTD: filter on the device, codec on the host, overlapped.
int64_t hb_zstd_compress_frame(const void *src, size_t n, void *dst, size_t cap, int level, int shuffle, int typesize,
                               unsigned opts, int device) {
    const ZstdApi *z = zstd_api();
    if (!z) return HB_ERR_INVALID_CODEC;
    if (typesize <= 0) typesize = 1;                                  // blosc.go:274-276
    if (level < 1) level = 1;                                         // blosc.go:277-282
    if (level > 9) level = 9;
    const int zl = zstd_level_for(level);
    const bool filt = (shuffle == HB_SHUFFLE || shuffle == HB_BITSHUFFLE) && typesize > 1;   // blosc.go:329-333
    const size_t nsl = (n + SLICE - 1) / SLICE;
    if (cap < HB_HEADER_SIZE + n) return HB_ERR_SHORT_BUFFER;         // a frame never exceeds 16 + n (memcpy rule, blosc.go:342)

    // ---- filtered image: device filter, copied back slice by slice ----
    uint8_t *h_f = nullptr;            // pinned host copy of the filtered buffer
    uint8_t *d_src = nullptr, *d_f = nullptr;
    size_t b_src = 0, b_f = 0, b_h = 0;
    hipStream_t st = nullptr;
    std::vector<hipEvent_t> ev(nsl, nullptr);
    const uint8_t *payload_src = (const uint8_t *)src;
    int rc = HB_OK;
    if (filt) {
        if (hipSetDevice(device) != hipSuccess) return HB_ERR_HIP;
        d_src = (uint8_t *)cache_get(false, device, n, &b_src);
        d_f = (uint8_t *)cache_get(false, device, n, &b_f);
        h_f = (uint8_t *)cache_get(true, device, n, &b_h);
        if (!d_src || !d_f || !h_f || hipStreamCreate(&st) != hipSuccess) rc = HB_ERR_HIP;
        if (!rc && hipMemcpyAsync(d_src, src, n, hipMemcpyHostToDevice, st) != hipSuccess) rc = HB_ERR_HIP;
        if (!rc) rc = hb_launch_filter(shuffle == HB_SHUFFLE ? HB_OP_SHUFFLE : HB_OP_BITSHUFFLE, d_f, d_src, n, typesize, st);
        for (size_t i = 0; i < nsl && !rc; i++) {
            const size_t off = i * SLICE, len = n - off < SLICE ? n - off : SLICE;
            if (hipMemcpyAsync(h_f + off, d_f + off, len, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) != hipSuccess || hipEventRecord(ev[i], st) != hipSuccess) rc = HB_ERR_HIP;
        }
        payload_src = h_f;
    }
    // ---- one zstd frame per slice, host threads, each starts when its slice has landed ----
    std::vector<std::vector<uint8_t>> outs(nsl);
    std::vector<size_t> osz(nsl, 0);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    if (!rc) {
        auto worker = [&]() {
            if (filt) (void)hipSetDevice(device);
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= nsl || failed.load()) return;
                if (filt && hipEventSynchronize(ev[i]) != hipSuccess) { failed = HB_ERR_HIP; return; }
                const size_t off = i * SLICE, len = n - off < SLICE ? n - off : SLICE;
                outs[i].resize(z->bound(len));
                const size_t c = z->compress(outs[i].data(), outs[i].size(), payload_src + off, len, zl);
                if (z->is_error(c)) { failed = HB_ERR_COMPRESSION_FAILED; return; }        // blosc.go:337-339
                osz[i] = c;
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)host_threads(), nsl);
        for (int t = 0; t < nt; t++) th.emplace_back(worker);
        for (auto &t : th) t.join();
        if (failed.load()) rc = failed.load();
    }
    int64_t ret = rc;
    if (!rc) {
        size_t total = 0;
        for (size_t i = 0; i < nsl; i++) total += osz[i];
        uint8_t *o = (uint8_t *)dst;
        const bool use_memcpy = total >= n;                           // blosc.go:342
        uint8_t flags = 0;                                            // blosc.go:348-356
        if (shuffle == HB_SHUFFLE) flags |= HB_FLAG_SHUFFLE;
        else if (shuffle == HB_BITSHUFFLE) flags |= HB_FLAG_BITSHUFFLE;
        if (use_memcpy) {
            flags |= HB_FLAG_MEMCPY;
            // the reference stores the UN-filtered input here (blosc.go:343-345) and then corrupts it on decode;
            // default: the filtered bytes, which its Decompress turns back into the input
            std::memcpy(o + HB_HEADER_SIZE, (opts & HB_OPT_REFERENCE_MEMCPY) ? (const uint8_t *)src : payload_src, n);
            total = n;
        } else {
            size_t at = HB_HEADER_SIZE;
            for (size_t i = 0; i < nsl; i++) { std::memcpy(o + at, outs[i].data(), osz[i]); at += osz[i]; }
        }
        o[0] = HB_FORMAT_VERSION; o[1] = HB_ZSTD; o[2] = flags; o[3] = (uint8_t)typesize;   // blosc.go:358-366
        put32(o + 4, (uint32_t)n); put32(o + 8, (uint32_t)n); put32(o + 12, (uint32_t)(HB_HEADER_SIZE + total));
        ret = (int64_t)(HB_HEADER_SIZE + total);
    }
    for (auto e : ev) if (e) (void)hipEventDestroy(e);
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }   // nothing in flight when the buffers go back
    cache_put(false, device, d_src, b_src);
    cache_put(false, device, d_f, b_f);
    cache_put(true, device, h_f, b_h);
    return ret;
}

// decompressBackend (blosc.go:377-434) for a non-memcpy ZSTD frame: host threads decode the concatenated zstd
// frames (zstd.Decoder.DecodeAll semantics), the device un-filters.  `h` is the parsed, validated header.
int64_t hb_zstd_decompress_frame(const void *frame, const hb_header &h, void *dst, size_t cap, int typesize_override, int device) {
    const ZstdApi *z = zstd_api();
    if (!z) return HB_ERR_INVALID_CODEC;
    const uint8_t *p = (const uint8_t *)frame + HB_HEADER_SIZE;
    const size_t plen = h.cbytes - HB_HEADER_SIZE, n = h.nbytes;
    if (n > cap) return HB_ERR_SHORT_BUFFER;
    const int ts = typesize_override > 0 ? typesize_override : (int)h.typesize;            // blosc.go:417-419
    int unf = -1;
    if ((h.flags & HB_FLAG_BITSHUFFLE) && ts > 1) unf = HB_OP_BITUNSHUFFLE;                // blosc.go:422-426
    else if ((h.flags & HB_FLAG_SHUFFLE) && ts > 1) unf = HB_OP_UNSHUFFLE;
    uint8_t *h_f = nullptr, *d_a = nullptr, *d_b = nullptr;
    size_t b_h = 0, b_a = 0, b_b = 0;
    uint8_t *target = (uint8_t *)dst;
    int rc = HB_OK;
    if (unf >= 0) {
        if (hipSetDevice(device) != hipSuccess) return HB_ERR_HIP;
        h_f = (uint8_t *)cache_get(true, device, n, &b_h);
        if (!h_f) return HB_ERR_HIP;
        target = h_f;
    }
    // frame table: every zstd frame with a known content size becomes one task
    struct Task { size_t src, clen, dst, dlen; };
    std::vector<Task> tasks;
    size_t at = 0, out = 0;
    bool table_ok = true;
    while (at < plen) {
        const size_t c = z->frame_csize(p + at, plen - at);
        const unsigned long long d = z->is_error(c) ? ~0ull : z->frame_content(p + at, plen - at);
        if (z->is_error(c) || d >= 0xFFFFFFFFFFFFFFFEull) { table_ok = false; break; }     // unknown size / not a frame
        // more content than NBytesOrig: the reference's DecodeAll grows its buffer and the frame layer then reports
        // ErrSizeMismatch (blosc.go:429-431), not a codec error
        if (d > n - out) { rc = HB_ERR_SIZE_MISMATCH; break; }
        tasks.push_back(Task{at, c, out, (size_t)d});
        at += c; out += (size_t)d;
    }
    size_t got = 0;
    // with an un-filter to run, every slice goes up to the device the moment its zstd frame is decoded: the uploads of the first
    // slices overlap the decoding of the later ones (one stream; the pinned staging buffer is the source)
    hipStream_t up = nullptr;
    bool uploaded = false;
    if (!rc && table_ok && unf >= 0 && n) {
        d_a = (uint8_t *)cache_get(false, device, n, &b_a);
        d_b = (uint8_t *)cache_get(false, device, n, &b_b);
        if (!d_a || !d_b || hipStreamCreateWithFlags(&up, hipStreamNonBlocking) != hipSuccess) rc = HB_ERR_HIP;
    }
    if (!rc && table_ok) {
        std::atomic<size_t> next{0};
        std::atomic<int> failed{0};
        auto worker = [&]() {
            if (up && hipSetDevice(device) != hipSuccess) { failed = HB_ERR_HIP; return; }
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= tasks.size() || failed.load()) return;
                const Task &t = tasks[i];
                const size_t r = z->decompress(target + t.dst, t.dlen, p + t.src, t.clen);
                if (z->is_error(r) || r != t.dlen) { failed = HB_ERR_DECOMPRESSION_FAILED; return; }   // blosc.go:411-413
                if (up && t.dlen && hipMemcpyAsync(d_a + t.dst, target + t.dst, t.dlen, hipMemcpyHostToDevice, up) != hipSuccess) { failed = HB_ERR_HIP; return; }
            }
        };
        std::vector<std::thread> th;
        const int nt = (int)std::min<size_t>((size_t)host_threads(), tasks.size());
        for (int t = 0; t < nt; t++) th.emplace_back(worker);
        for (auto &t : th) t.join();
        if (up && hipStreamSynchronize(up) != hipSuccess && !failed.load()) failed = HB_ERR_HIP;
        if (failed.load()) rc = failed.load();
        got = out;
        uploaded = up != nullptr && !rc;
    } else if (!rc) {                                                 // no usable frame table: one call, as DecodeAll
        const size_t r = z->decompress(target, n, p, plen);
        if (z->is_error(r))                                           // 70 = ZSTD_error_dstSize_tooSmall: decodes, but to more than NBytesOrig
            rc = (z->error_code && z->error_code(r) == 70) ? HB_ERR_SIZE_MISMATCH : HB_ERR_DECOMPRESSION_FAILED;
        else got = r;
    }
    if (!rc && got != n) rc = HB_ERR_SIZE_MISMATCH;                   // blosc.go:429-431
    if (!rc && unf >= 0 && n) {
        if (!d_a) d_a = (uint8_t *)cache_get(false, device, n, &b_a);
        if (!d_b) d_b = (uint8_t *)cache_get(false, device, n, &b_b);
        if (!d_a || !d_b) rc = HB_ERR_HIP;
        if (!rc && !uploaded && hipMemcpy(d_a, h_f, n, hipMemcpyHostToDevice) != hipSuccess) rc = HB_ERR_HIP;
        if (!rc) rc = hb_launch_filter(unf, d_b, d_a, n, ts, nullptr);
        if (!rc && hipMemcpy(dst, d_b, n, hipMemcpyDeviceToHost) != hipSuccess) rc = HB_ERR_HIP;
    }
    if (up) (void)hipStreamDestroy(up);
    cache_put(false, device, d_a, b_a);                               // the copies above were synchronous
    cache_put(false, device, d_b, b_b);
    cache_put(true, device, h_f, b_h);
    return rc ? (int64_t)rc : (int64_t)n;
}
