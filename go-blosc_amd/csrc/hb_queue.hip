// hb_queue.hip — pipelined host API (include/hipblosc.h "hb_queue_*").
//
// One slot = one frame in flight: its own stream, device buffers (input, frame / output, workspace) and a pinned
// hb_result.  Submitting enqueues  H2D -> kernels -> result D2H  on the slot's stream and returns; waiting synchronises
// with that stream, reads how many bytes came out, and downloads exactly those.  With `depth` slots the upload of frame
// k+1 overlaps the kernels of frame k and the download of frame k-1 (the two PCIe directions and the compute queue are
// independent engines), which is what the one-call API of hb_api.hip cannot do.
#include "hb_common.h"
#include "hb_lz4.h"
#include "hb_ticket_ring.h"

#include <vector>
#include <utility>
#include <new>
#include <thread>
#include <algorithm>

extern "C" {
size_t hb_frame_bound(size_t n);
size_t hb_compress_frame_workspace(size_t n);
size_t hb_decompress_frame_workspace(size_t n_out);
size_t hb_decompress_frame_workspace_foreign(size_t n_out);
int64_t hb_compress_frame(const void *src, size_t n, void *dst, size_t cap, int codec, int level, int shuffle, int typesize,
                          unsigned opts, int device);
int64_t hb_decompress_frame(const void *frame, size_t n, void *dst, size_t cap, int typesize_override, int device);
int hb_compress_frame_dev(const void *d_src, size_t n, void *d_frame, size_t cap, int codec, int level, int shuffle,
                          int typesize, unsigned opts, void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
}

namespace {
enum SlotState { SLOT_FREE = 0, SLOT_BUSY = 1, SLOT_DONE = 2 };
struct Slot {
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr, *d_out = nullptr, *d_work = nullptr;
    hb_result *d_res = nullptr;          // device
    hb_result *h_res = nullptr;          // pinned
    int state = SLOT_FREE;
    int64_t ticket = -1;
    int64_t rc = 0;                      // SLOT_DONE: what wait() returns
    bool compress = false;
    void *dst = nullptr; size_t cap = 0; unsigned opts = 0;
};
}  // namespace

struct hb_queue {
    int device = 0;
    size_t max_n = 0, in_bytes = 0, out_bytes = 0, work_bytes = 0;
    std::vector<Slot> slots;
    int64_t next_ticket = 0;
    hb_ticket_ring kept;                 // results of tickets whose slot was re-used before they were waited for
};

namespace {

void free_slot(Slot &s) {
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.d_in) (void)hipFree(s.d_in);
    if (s.d_out) (void)hipFree(s.d_out);
    if (s.d_work) (void)hipFree(s.d_work);
    if (s.d_res) (void)hipFree(s.d_res);
    if (s.h_res) (void)hipHostFree(s.h_res);
    s = Slot();
}

// bring a busy slot to SLOT_DONE: wait for its kernels, download what they produced
void finish(hb_queue *q, Slot &s) {
    if (s.state != SLOT_BUSY) return;
    s.state = SLOT_DONE;
    if (hipSetDevice(q->device) != hipSuccess || hipStreamSynchronize(s.stream) != hipSuccess) { s.rc = HB_ERR_HIP; return; }
    const hb_result r = *s.h_res;
    if (r.status) { s.rc = r.status; return; }
    const size_t out = s.compress ? ((s.opts & HB_OPT_INDEX_TRAILER) ? r.total_bytes : r.bytes) : r.bytes;
    if (out > s.cap) { s.rc = HB_ERR_SHORT_BUFFER; return; }
    if (out) {
        if (hipMemcpyAsync(s.dst, s.d_out, out, hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
            hipStreamSynchronize(s.stream) != hipSuccess) { s.rc = HB_ERR_HIP; return; }
    }
    s.rc = (int64_t)out;
}

Slot *take_slot(hb_queue *q, int64_t *ticket) {
    *ticket = q->next_ticket++;
    Slot &s = q->slots[(size_t)(*ticket % (int64_t)q->slots.size())];
    finish(q, s);                        // all slots in flight: the oldest one is completed first ...
    if (s.state == SLOT_DONE && s.ticket >= 0) {   // ... and its result is kept for its hb_queue_wait (the data is already in its dst)
        q->kept.put(s.ticket, s.rc);
    }
    return &s;
}

}  // namespace

extern "C" {

hb_queue *hb_queue_create(int device, int depth, size_t max_nbytes) { return hb_queue_create_ex(device, depth, max_nbytes, 0u); }

hb_queue *hb_queue_create_ex(int device, int depth, size_t max_nbytes, unsigned flags) {
    if (hb_init() != HB_OK || device < 0 || device >= hb_device_count() || depth < 1 || depth > 64 || max_nbytes == 0) return nullptr;
    if (max_nbytes > 0xFFFFFFFFull - HB_HEADER_SIZE - max_nbytes / 255 - 64) return nullptr;
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    hb_queue *q = new (std::nothrow) hb_queue();
    if (!q) return nullptr;
    q->device = device;
    q->max_n = max_nbytes;
    const size_t fb = hb_frame_bound(max_nbytes) + 64;
    q->in_bytes = fb;                                       // compress: the input (<= max_n); decompress: a frame (<= bound);
                                                            // allocated with 64 bytes of slack: the kernels' 16-byte loads may read past the end
    q->out_bytes = fb;                                      // compress: the frame; decompress: the output (<= max_n)
    q->work_bytes = hb_compress_frame_workspace(max_nbytes);
    // (HB_QUEUE_FOREIGN_FRAMES: room for the symbolic decoder, so that frames of other writers decode in parallel too)
    const size_t dw = (flags & HB_QUEUE_FOREIGN_FRAMES) ? hb_decompress_frame_workspace_foreign(max_nbytes) : hb_decompress_frame_workspace(max_nbytes);
    if (dw > q->work_bytes) q->work_bytes = dw;
    q->slots.resize((size_t)depth);
    q->kept = hb_ticket_ring(4 * (size_t)depth);
    bool ok = true;
    for (auto &s : q->slots) {
        ok = ok && hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) == hipSuccess;
        ok = ok && hipMalloc((void **)&s.d_in, q->in_bytes + 64) == hipSuccess;
        ok = ok && hipMalloc((void **)&s.d_out, q->out_bytes) == hipSuccess;
        ok = ok && hipMalloc((void **)&s.d_work, q->work_bytes) == hipSuccess;
        ok = ok && hipMalloc((void **)&s.d_res, sizeof(hb_result)) == hipSuccess;
        ok = ok && hipHostMalloc((void **)&s.h_res, sizeof(hb_result), hipHostMallocDefault) == hipSuccess;
        if (!ok) break;
    }
    if (!ok) { for (auto &s : q->slots) free_slot(s); delete q; return nullptr; }
    return q;
}

void hb_queue_destroy(hb_queue *q) {
    if (!q) return;
    (void)hipSetDevice(q->device);
    for (auto &s : q->slots) { finish(q, s); free_slot(s); }
    delete q;
}

int64_t hb_queue_compress(hb_queue *q, const void *src, size_t n, void *dst, size_t cap, int codec, int level, int shuffle,
                          int typesize, unsigned opts) {
    if (!q) return HB_ERR_BAD_ARG;
    if (n == 0) return HB_ERR_INVALID_DATA;                               // blosc.go:269-271
    if (!src || !dst) return HB_ERR_BAD_ARG;
    if (!hb_device_codec(codec)) return HB_ERR_INVALID_CODEC;             // blosc.go:322-325; the queue carries the device codecs only
    if (n > q->max_n) return HB_ERR_DATA_TOO_LARGE;
    if (hipSetDevice(q->device) != hipSuccess) return HB_ERR_HIP;
    int64_t ticket;
    Slot *s = take_slot(q, &ticket);
    s->state = SLOT_DONE; s->ticket = ticket; s->compress = true; s->dst = dst; s->cap = cap; s->opts = opts;
    s->rc = HB_ERR_HIP;
    if (hipMemcpyAsync(s->d_in, src, n, hipMemcpyHostToDevice, s->stream) != hipSuccess) return ticket;   // wait() reports it
    const int rc = hb_compress_frame_dev(s->d_in, n, s->d_out, q->out_bytes, codec, level, shuffle, typesize, opts,
                                         s->d_work, q->work_bytes, s->d_res, s->stream);
    if (rc) { s->rc = rc; return ticket; }
    if (hipMemcpyAsync(s->h_res, s->d_res, sizeof(hb_result), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return ticket;
    s->state = SLOT_BUSY;
    return ticket;
}

int64_t hb_queue_decompress(hb_queue *q, const void *frame, size_t n, void *dst, size_t cap, int typesize_override) {
    if (!q) return HB_ERR_BAD_ARG;
    if (n < HB_HEADER_SIZE) return HB_ERR_INVALID_HEADER;                 // blosc.go:297-299
    if (!frame) return HB_ERR_BAD_ARG;
    hb_header h;
    int rc = hb_parse_header(frame, n, &h);                               // the frame is in host memory: no read-back
    if (rc) return rc;
    if ((size_t)h.cbytes > n || h.cbytes < HB_HEADER_SIZE) return HB_ERR_INVALID_DATA;           // blosc.go:385-390
    if (!(h.flags & HB_FLAG_MEMCPY) && !hb_device_codec(h.codec)) return HB_ERR_INVALID_CODEC;
    if ((size_t)h.nbytes > q->max_n || n > q->in_bytes) return HB_ERR_DATA_TOO_LARGE;
    if ((size_t)h.nbytes > cap) return HB_ERR_SHORT_BUFFER;
    if (hipSetDevice(q->device) != hipSuccess) return HB_ERR_HIP;
    int64_t ticket;
    Slot *s = take_slot(q, &ticket);
    s->state = SLOT_DONE; s->ticket = ticket; s->compress = false; s->dst = dst; s->cap = cap; s->opts = 0;
    s->rc = HB_ERR_HIP;
    if (hipMemcpyAsync(s->d_in, frame, n, hipMemcpyHostToDevice, s->stream) != hipSuccess) return ticket;
    rc = hb_decompress_frame_dev_hdr(&h, s->d_in, n, s->d_out, h.nbytes, typesize_override, s->d_work, q->work_bytes,
                                     s->d_res, s->stream);
    if (rc) { s->rc = rc; return ticket; }
    if (hipMemcpyAsync(s->h_res, s->d_res, sizeof(hb_result), hipMemcpyDeviceToHost, s->stream) != hipSuccess) return ticket;
    s->state = SLOT_BUSY;
    return ticket;
}

int64_t hb_queue_wait(hb_queue *q, int64_t ticket) {
    if (!q || ticket < 0 || ticket >= q->next_ticket) return HB_ERR_BAD_ARG;
    Slot &s = q->slots[(size_t)(ticket % (int64_t)q->slots.size())];
    if (s.ticket != ticket) {                                                 // slot re-used since: the result may have been kept
        int64_t rc;
        return q->kept.take(ticket, &rc) ? rc : (int64_t)HB_ERR_BAD_ARG;      // else: already waited for, or too old
    }
    if (s.state == SLOT_FREE) return HB_ERR_BAD_ARG;                          // already waited for
    finish(q, s);
    s.state = SLOT_FREE;
    return s.rc;
}


}  // extern "C"

// ---- batches of independent frames over all devices: frame k -> device k mod hb_device_count() ----
// One host thread per device, each with its own hb_queue (3 frames in flight: the upload of frame k+1 overlaps the kernels
// of frame k and the download of frame k-1).  Frames the queue does not carry (host codecs, malformed headers) go through
// the one-call entry points, which also produce the reference's error for them.  No device talks to another one.
namespace {
constexpr int MULTI_DEPTH = 3;
// Small frames of a device's share do not go through the queue one by one (a 1 MiB frame is 200 us of launches and synchronisation
// for 2 us of kernel): they are collected and sent through the batch entry points (hb_batch.hip: one set of launches for all of them),
// MULTI_BATCH_BYTES of input at a time.  What counts as small: up to MULTI_SMALL bytes (uncompressed).
constexpr size_t MULTI_SMALL = (size_t)4 << 20, MULTI_BATCH_BYTES = (size_t)256 << 20;

template <class Submit, class OneCall>
void run_device(int dev, int nd, int nframes, size_t max_n, unsigned qflags, int64_t *rc, const std::vector<char> &done, Submit submit, OneCall one_call) {
    hb_queue *q = max_n ? hb_queue_create_ex(dev, MULTI_DEPTH, max_n, qflags) : nullptr;
    std::vector<std::pair<int, int64_t>> inflight;        // {frame, ticket}
    auto retire = [&](size_t keep) {
        while (inflight.size() > keep) {
            rc[inflight.front().first] = hb_queue_wait(q, inflight.front().second);
            inflight.erase(inflight.begin());
        }
    };
    for (int k = dev; k < nframes; k += nd) {
        if (done[(size_t)k]) continue;                     // went through a batch
        int64_t t = q ? submit(q, k) : (int64_t)HB_ERR_BAD_ARG;
        if (t >= 0) { inflight.push_back({k, t}); retire(MULTI_DEPTH - 1); }
        else rc[k] = one_call(k, dev);                     // not a queue frame: one call (also yields the right error code)
    }
    if (q) { retire(0); hb_queue_destroy(q); }
}
}  // namespace

extern "C" {

int hb_compress_frames_multi(int nframes, const void *const *src, const size_t *n, void *const *dst, const size_t *cap,
                             int64_t *rc, int codec, int level, int shuffle, int typesize, unsigned opts) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (nframes < 0 || (nframes && (!src || !n || !dst || !cap || !rc))) return HB_ERR_BAD_ARG;
    const int nd = hb_device_count();
    std::vector<std::thread> th;
    for (int d = 0; d < nd && d < nframes; d++) {
        th.emplace_back([=]() {
            // small frames of this device: batches (LZ4 / LZ4HC: the codecs the batch kernels carry)
            std::vector<char> batched((size_t)nframes, 0);
            if (codec == HB_LZ4 || codec == HB_LZ4HC) {
                std::vector<int> ks;
                size_t bytes = 0;
                auto flush = [&]() {
                    if (ks.size() < 2) { for (int k : ks) batched[(size_t)k] = 0; ks.clear(); bytes = 0; return; }   // a lone small frame: the queue
                    const int m = (int)ks.size();
                    std::vector<const void *> ps((size_t)m); std::vector<void *> pd((size_t)m); std::vector<size_t> ns((size_t)m), cs((size_t)m); std::vector<int64_t> rs((size_t)m, HB_ERR_HIP);
                    for (int i = 0; i < m; i++) { ps[(size_t)i] = src[ks[(size_t)i]]; pd[(size_t)i] = dst[ks[(size_t)i]]; ns[(size_t)i] = n[ks[(size_t)i]]; cs[(size_t)i] = cap[ks[(size_t)i]]; }
                    const int st = hb_compress_frames_batch(m, ps.data(), ns.data(), pd.data(), cs.data(), rs.data(), codec, level, shuffle, typesize, opts, d);
                    for (int i = 0; i < m; i++) rc[ks[(size_t)i]] = st ? (int64_t)st : rs[(size_t)i];
                    ks.clear(); bytes = 0;
                };
                for (int k = d; k < nframes; k += nd) {
                    if (!src[k] || !dst[k] || n[k] == 0 || n[k] > MULTI_SMALL) continue;
                    batched[(size_t)k] = 1; ks.push_back(k); bytes += n[k];
                    if (bytes >= MULTI_BATCH_BYTES) flush();
                }
                flush();
            }
            size_t max_n = 0;
            if (hb_device_codec(codec))
                for (int k = d; k < nframes; k += nd)
                    if (!batched[(size_t)k] && n[k] <= 0xFFFFFFFFull - HB_HEADER_SIZE - n[k] / 255 - 64) max_n = std::max(max_n, n[k]);
            run_device(d, nd, nframes, max_n, 0u, rc, batched,
                       [&](hb_queue *q, int k) -> int64_t {
                           if (!hb_device_codec(codec) || !src[k] || !dst[k] || n[k] == 0) return HB_ERR_BAD_ARG;
                           return hb_queue_compress(q, src[k], n[k], dst[k], cap[k], codec, level, shuffle, typesize, opts);
                       },
                       [&](int k, int dev) { return hb_compress_frame(src[k], n[k], dst[k], cap[k], codec, level, shuffle, typesize, opts, dev); });
        });
    }
    for (auto &t : th) t.join();
    return HB_OK;
}

int hb_decompress_frames_multi(int nframes, const void *const *frame, const size_t *n, void *const *dst, const size_t *cap,
                               int64_t *rc, int typesize_override) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (nframes < 0 || (nframes && (!frame || !n || !dst || !cap || !rc))) return HB_ERR_BAD_ARG;
    const int nd = hb_device_count();
    std::vector<std::thread> th;
    for (int d = 0; d < nd && d < nframes; d++) {
        th.emplace_back([=]() {
            // small frames of this device: batches (frames the batch entry point cannot carry are answered by the one-frame one inside it)
            std::vector<char> batched((size_t)nframes, 0);
            {
                std::vector<int> ks;
                size_t bytes = 0;
                auto flush = [&]() {
                    if (ks.size() < 2) { for (int k : ks) batched[(size_t)k] = 0; ks.clear(); bytes = 0; return; }
                    const int m = (int)ks.size();
                    std::vector<const void *> ps((size_t)m); std::vector<void *> pd((size_t)m); std::vector<size_t> ns((size_t)m), cs((size_t)m); std::vector<int64_t> rs((size_t)m, HB_ERR_HIP);
                    for (int i = 0; i < m; i++) { ps[(size_t)i] = frame[ks[(size_t)i]]; pd[(size_t)i] = dst[ks[(size_t)i]]; ns[(size_t)i] = n[ks[(size_t)i]]; cs[(size_t)i] = cap[ks[(size_t)i]]; }
                    const int st = hb_decompress_frames_batch(m, ps.data(), ns.data(), pd.data(), cs.data(), rs.data(), typesize_override, d);
                    for (int i = 0; i < m; i++) rc[ks[(size_t)i]] = st ? (int64_t)st : rs[(size_t)i];
                    ks.clear(); bytes = 0;
                };
                for (int k = d; k < nframes; k += nd) {
                    hb_header h;
                    if (!frame[k] || hb_parse_header(frame[k], n[k], &h) != HB_OK || (size_t)h.nbytes > MULTI_SMALL || (size_t)h.nbytes > cap[k] || (!dst[k] && cap[k])) continue;
                    batched[(size_t)k] = 1; ks.push_back(k); bytes += h.nbytes;
                    if (bytes >= MULTI_BATCH_BYTES) flush();
                }
                flush();
            }
            size_t max_n = 0;                              // largest decoded size among this device's well-formed LZ4 frames
            unsigned qflags = 0;                           // any large LZ4 frame without a trailer: it may be somebody else's
            for (int k = d; k < nframes; k += nd) {
                hb_header h;
                if (batched[(size_t)k]) continue;
                // frames whose header asks for more than the caller gave (nbytes above cap[k], cbytes above n[k]) never reach a queue slot:
                // they must not size the slots either (one forged NBytesOrig made every device allocate 3 x ~3 x 4 GiB); the
                // one-call path answers them with the reference's error
                if (frame[k] && hb_parse_header(frame[k], n[k], &h) == HB_OK && (hb_device_codec(h.codec) || (h.flags & HB_FLAG_MEMCPY)) &&
                    (size_t)h.nbytes <= cap[k] && (size_t)h.cbytes <= n[k]) {
                    max_n = std::max(max_n, std::max<size_t>(h.nbytes, 1));
                    if (!(h.flags & HB_FLAG_MEMCPY) && h.cbytes >= HB_HEADER_SIZE && hb_indexless_parallel((size_t)h.cbytes - HB_HEADER_SIZE, h.nbytes) &&
                        n[k] <= (((size_t)h.cbytes + 7) & ~(size_t)7) + 32) qflags = HB_QUEUE_FOREIGN_FRAMES;
                }
            }
            if (max_n > 0xFFFFFFFFull - HB_HEADER_SIZE - max_n / 255 - 64) max_n = 0;
            run_device(d, nd, nframes, max_n, qflags, rc, batched,
                       [&](hb_queue *q, int k) -> int64_t {
                           if (!frame[k] || (!dst[k] && cap[k])) return HB_ERR_BAD_ARG;
                           return hb_queue_decompress(q, frame[k], n[k], dst[k], cap[k], typesize_override);
                       },
                       [&](int k, int dev) { return hb_decompress_frame(frame[k], n[k], dst[k], cap[k], typesize_override, dev); });
        });
    }
    for (auto &t : th) t.join();
    return HB_OK;
}

}  // extern "C"
