// hb_batch.hip — batches of frames in ONE set of launches (SURVEY §8 row f1 "frame batches"; include/hipblosc.h
// hb_compress_frames_batch_dev / hb_decompress_frames_batch_dev).
//
// The reference's own benchmark, and every number it publishes, is a 100 000-byte frame (blosc_test.go:363-413, README.md:113-130).
// One such frame is 25 chunks of work: the one-frame entry points launch four or five kernels for it and leave 99 % of the chip
// idle (1 MiB: 204 us per frame, 5 GB/s).  Frames are independent (blosc.go:37-39, :320-434 share no state), chunks are
// independent, and the per-frame scan is a segmented scan -- so K frames go through the SAME kernels as one frame, flattened:
//
//   compress   hb_lz4_enc.hip: k_bt_map, [batched filter], k_match / k_match_fused, k_tiles, k_scan (one workgroup per frame),
//              [gated batched filter for memcpy frames], k_stitch -- 6 launches for any K.  Every frame is byte-identical to what
//              hb_compress_frame_dev writes for it (same kernels, same per-frame positions; tests/test_gpu_batch.py).
//   decompress frames that carry the restart index: k_bt_dec_plan + k_dec_indexed_batch (hb_lz4_dec.hip) over the units of all
//              frames; frames without one (the default frame shape, and anything another writer produced): ONE wavefront per frame
//              decodes the stream front to back (k_bt_dec_streams: the unit decoder of hb_sym_decode.h, 13 KiB of LDS, a dozen
//              frames per CU in flight) -- slow for one frame, fine for thousands; k_bt_dec_finish settles every frame's result
//              and is the authority for anything that did not check out (the serial block decoder of hb_dec_common.h: bytes
//              and errors of lz4.UncompressBlock, codec.go:77-84); then the batched un-filter.  Error semantics per frame are
//              those of hb_decompress_frame_dev (blosc.go:377-434).
#include "hb_sym_decode.h"
#include <vector>
#include <algorithm>
#include <cstring>

namespace {

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// ---- frames without an index: one wavefront per frame walks the whole block ----
__global__ __launch_bounds__(64) void k_bt_dec_streams(const DecBatchFrame *__restrict__ bf) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    __shared__ __attribute__((aligned(16))) uint8_t s_d[SY_IMG + 64];
    const int lane = threadIdx.x;
    const DecBatchFrame f = bf[blockIdx.x];
    if (f.preset != 1) return;
    if (f.plan->mode == DEC_INDEXED && !f.plan->fail) return;        // the indexed decoder vouches for this frame
    if (f.n_src == 0 || f.n_src > 0xFFFFFFF0ull) return;             // (empty block: lz4.UncompressBlock answers 0, nil -- the finisher's business)
    uint32_t out = 0;
    bool parked;
    const bool ok = sy_decode_unit<false>(f.src, f.n_src, 0u, (uint32_t)f.n_src, 0u, 0u, out, f.serial_dst, nullptr, s_win, s_tq, s_d, nullptr,
                                          lane, 1, 0u, 0u, nullptr, nullptr, nullptr, parked, f.nbytes);
    if (lane == 0) {
        f.plan->pad[0] = ok ? 1u : 0u;                               // verdict: only a clean decode counts; anything else goes to the authority
        f.plan->pad[1] = out;
        if (ok && f.post_needed) f.plan->post = 1;
    }
}

// ---- one workgroup per frame: the result record; frames nobody vouches for are decoded here by the serial block decoder ----
__global__ __launch_bounds__(64) void k_bt_dec_finish(const DecBatchFrame *__restrict__ bf) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[SER_WIN + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_img[SER_HIST + SER_PAGE + 1024];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    const DecBatchFrame f = bf[blockIdx.x];
    hb_result *r = f.result;
    if (f.preset != 1) {                                             // decided on the host: a memcpy frame (blosc.go:398-400, :429-431)
        if (lane == 0) { r->status = f.preset; r->flags = 0; r->bytes = f.n_src; r->total_bytes = f.n_src; r->reserved = 0; }
        return;
    }
    if (f.plan->mode == DEC_INDEXED && !f.plan->fail) {
        if (lane == 0) {
            const uint64_t got = f.plan->nbytes;
            r->flags = 1; r->bytes = got; r->total_bytes = got; r->reserved = 0;
            r->status = got != f.nbytes ? HB_ERR_SIZE_MISMATCH : HB_OK;                        // blosc.go:429-431
        }
        return;
    }
    if (f.plan->pad[0] == 1u) {                                      // the stream decoder's clean decode
        if (lane == 0) {
            const uint64_t got = f.plan->pad[1];
            r->flags = 0; r->bytes = got; r->total_bytes = 0; r->reserved = 0;
            r->status = got != f.nbytes ? HB_ERR_SIZE_MISMATCH : HB_OK;
        }
        return;
    }
    if (f.post_needed && lane == 0) f.plan->post = 1;
    int err;
    const uint64_t got = dec_serial_core(f.src, f.n_src, f.serial_dst, (uint64_t)f.nbytes, s_win, s_img, s_tq, lane, err);
    if (lane == 0) {
        r->flags = 0; r->total_bytes = 0; r->reserved = 0;
        if (err) { r->status = HB_ERR_DECOMPRESSION_FAILED; r->bytes = 0; }                    // blosc.go:411-413
        else if (got != f.nbytes) { r->status = HB_ERR_SIZE_MISMATCH; r->bytes = got; }        // blosc.go:429-431
        else { r->status = HB_OK; r->bytes = got; }
    }
}

// plain copies of a batch (memcpy frames without a filter): job blockIdx.y
__global__ __launch_bounds__(256) void k_bt_copy(const hb_filter_job *__restrict__ jobs) {
    const hb_filter_job j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
    for (uint64_t off = (uint64_t)wave * 16384u; off < j.n; off += (uint64_t)nw * 16384u)
        wave_copy_g2g(j.dst + off, j.src + off, (uint32_t)std::min<uint64_t>(16384u, j.n - off), lane);
}

// the 16 header bytes of every frame, gathered into one contiguous buffer
__global__ void k_bt_gather_headers(const uint8_t *const *__restrict__ frames, const uint8_t *__restrict__ valid, u32x4 *__restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { u32x4 z; z.x = z.y = z.z = z.w = 0; out[i] = valid[i] ? ld16u(frames[i]) : z; }
}

struct DecBatchLayout { size_t frames, plans, jobs, unit_frame, staged, rgjobs, rgwork, rgidx, total; };
// rg_bytes / rgi_bytes: scratch / rebuilt index of the frames that come without a restart index and get one from the token discovery (hb_lz4_region.hip)
DecBatchLayout dec_batch_layout(int nframes, size_t total_units, size_t staged_bytes, size_t rg_bytes, size_t rgi_bytes) {
    DecBatchLayout L{};
    size_t o = 0;
    auto take = [&](size_t b) { size_t at = o; o += al256(b); return at; };
    L.frames = take((size_t)nframes * sizeof(DecBatchFrame));
    L.plans = take((size_t)nframes * sizeof(DecPlan));
    L.jobs = take((size_t)nframes * sizeof(hb_filter_job));
    L.unit_frame = take(total_units * 4 + 64);
    L.staged = take(staged_bytes + 256);
    L.rgjobs = take((size_t)nframes * sizeof(RgJob));
    L.rgwork = take(rg_bytes + 256);
    L.rgidx = take(rgi_bytes + 256);
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

size_t hb_compress_frames_batch_workspace(int nframes, const size_t *n, int typesize) { return hb_lz4_enc_batch_workspace(nframes, n, typesize); }

int hb_compress_frames_batch_dev(int nframes, const void *const *d_src, const size_t *n, void *const *d_frame, const size_t *cap,
                                 int codec, int level, int shuffle, int typesize, unsigned opts,
                                 void *d_work, size_t work_bytes, hb_result *d_results, void *stream) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (nframes < 0 || (nframes && (!d_src || !n || !d_frame || !cap || !d_work || ((uintptr_t)d_work & 255u) || !d_results))) return HB_ERR_BAD_ARG;
    if (nframes == 0) return HB_OK;
    if (typesize <= 0) typesize = 1;                                  // blosc.go:274-276
    if (level < 1) level = 1;                                         // :277-282
    if (level > 9) level = 9;
    if (codec != HB_LZ4 && codec != HB_LZ4HC) return HB_ERR_INVALID_CODEC;      // the batch carries the LZ4 block format; Snappy / ZSTD: one call per frame
    std::vector<hb_batch_frame> fr((size_t)nframes);
    for (int k = 0; k < nframes; k++) {
        if (n[k] == 0) return HB_ERR_INVALID_DATA;                    // blosc.go:269-271 (the whole batch is refused: nothing has been launched)
        if (!d_src[k] || !d_frame[k]) return HB_ERR_BAD_ARG;
        if (n[k] > 0xFFFFFFFFull - HB_HEADER_SIZE - n[k] / 255 - 64) return HB_ERR_DATA_TOO_LARGE;
        if (cap[k] < hb_frame_bound(n[k])) return HB_ERR_SHORT_BUFFER;
        fr[(size_t)k] = hb_batch_frame{(const uint8_t *)d_src[k], n[k], (uint8_t *)d_frame[k], cap[k], d_results + k};
    }
    return hb_launch_lz4_encode_batch(nframes, fr.data(), codec, level, shuffle, typesize, opts, (uint8_t *)d_work, work_bytes, (hipStream_t)stream);
}

// gathers the 16 header bytes of `nframes` device-resident frames and parses them: ONE small D2H and one stream synchronisation for the whole
// batch (hb_decompress_frame_dev pays one per frame).  d_scratch: >= 32 * nframes + 256 bytes of device memory.  hdrs[k] is valid where rc[k] == HB_OK.
int hb_frames_batch_headers_dev(int nframes, const void *const *d_frame, const size_t *n, hb_header *hdrs, int *rc,
                                void *d_scratch, size_t scratch_bytes, void *stream) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (nframes < 0 || (nframes && (!d_frame || !n || !hdrs || !rc || !d_scratch))) return HB_ERR_BAD_ARG;
    if (nframes == 0) return HB_OK;
    if (scratch_bytes < (size_t)nframes * 32 + 256) return HB_ERR_SHORT_BUFFER;
    hipStream_t s = (hipStream_t)stream;
    uint8_t *w = (uint8_t *)d_scratch;
    const uint8_t **d_ptrs = (const uint8_t **)w;
    uint8_t *d_valid = w + al256((size_t)nframes * 8);
    u32x4 *d_out = (u32x4 *)(d_valid + al256((size_t)nframes));
    if ((size_t)((uint8_t *)(d_out + nframes) - w) > scratch_bytes) return HB_ERR_SHORT_BUFFER;
    std::vector<uint8_t> valid((size_t)nframes), raw((size_t)nframes * HB_HEADER_SIZE);
    for (int k = 0; k < nframes; k++) {
        rc[k] = (n[k] < HB_HEADER_SIZE) ? HB_ERR_INVALID_HEADER : (d_frame[k] ? HB_OK : HB_ERR_BAD_ARG);      // blosc.go:297-299
        valid[(size_t)k] = rc[k] == HB_OK;
    }
    HB_HIP_TRY(hipMemcpyAsync(d_ptrs, d_frame, (size_t)nframes * 8, hipMemcpyHostToDevice, s));
    HB_HIP_TRY(hipMemcpyAsync(d_valid, valid.data(), (size_t)nframes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bt_gather_headers, dim3((unsigned)((nframes + 255) / 256)), dim3(256), 0, s, (const uint8_t *const *)d_ptrs, (const uint8_t *)d_valid, d_out, nframes);
    HB_HIP_TRY(hipMemcpyAsync(raw.data(), d_out, raw.size(), hipMemcpyDeviceToHost, s));
    HB_HIP_TRY(hipStreamSynchronize(s));
    for (int k = 0; k < nframes; k++)
        if (rc[k] == HB_OK) rc[k] = hb_parse_header(raw.data() + (size_t)k * HB_HEADER_SIZE, HB_HEADER_SIZE, &hdrs[k]);
    return HB_OK;
}

size_t hb_decompress_frames_batch_workspace(int nframes, const hb_header *hdrs) {
    if (nframes <= 0 || !hdrs) return 256;
    size_t units = 0, staged = 0, rg = 0, rgi = 0;
    for (int k = 0; k < nframes; k++) {
        units = (units + 31) / 32 * 32 + ((size_t)hdrs[k].nbytes + HB_CHUNK - 1) / HB_CHUNK;
        staged += al256((size_t)hdrs[k].nbytes + 64);
        // (whether a frame brings its index is not in the header: room for every LZ4 frame's discovery)
        if (!(hdrs[k].flags & HB_FLAG_MEMCPY) && hdrs[k].cbytes >= HB_HEADER_SIZE && hb_lz4_region_batch_wanted((size_t)hdrs[k].cbytes - HB_HEADER_SIZE, hdrs[k].nbytes)) {
            rg += rg_batch_layout((size_t)hdrs[k].cbytes - HB_HEADER_SIZE).total;
            rgi += al256(hb_lz4_index_bound(hdrs[k].nbytes));
        }
    }
    return dec_batch_layout(nframes, units + 32, staged, rg, rgi).total;
}

// Per-frame outcome in d_results[k] (as hb_decompress_frame_dev reports it); frames whose header the host can already refuse get their
// error there too.  The call itself returns HB_OK unless its arguments are unusable.
int hb_decompress_frames_batch_dev(int nframes, const hb_header *hdrs, const void *const *d_frame, const size_t *n,
                                   void *const *d_dst, const size_t *cap, int typesize_override,
                                   void *d_work, size_t work_bytes, hb_result *d_results, void *stream) {
    if (hb_init() != HB_OK) return HB_ERR_NO_DEVICE;
    if (nframes < 0 || (nframes && (!hdrs || !d_frame || !n || !d_dst || !cap || !d_work || ((uintptr_t)d_work & 255u) || !d_results))) return HB_ERR_BAD_ARG;
    if (nframes == 0) return HB_OK;
    if (work_bytes < hb_decompress_frames_batch_workspace(nframes, hdrs)) return HB_ERR_SHORT_BUFFER;
    hipStream_t s = (hipStream_t)stream;
    uint8_t *w = (uint8_t *)d_work;

    // frames the host refuses (blosc.go:385-390, :403-407; short destination) keep their place in the arrays as presets
    std::vector<DecBatchFrame> h((size_t)nframes);
    std::vector<int> unf((size_t)nframes, -1), tsv((size_t)nframes, 1);
    size_t staged_total = 0, rg_total = 0, rgi_total = 0, rg_stream = 0;
    std::vector<size_t> rg_off((size_t)nframes, (size_t)-1), rgi_off((size_t)nframes, 0);      // frames whose index is rebuilt on the device: offsets of their scratch / index
    for (int k = 0; k < nframes; k++) {
        const hb_header &hd = hdrs[k];
        DecBatchFrame &f = h[(size_t)k];
        f = DecBatchFrame{};
        f.result = d_results + k;
        f.nbytes = hd.nbytes;
        int st = HB_OK;
        if (!d_frame[k] || (!d_dst[k] && cap[k])) st = HB_ERR_BAD_ARG;
        else if (n[k] < HB_HEADER_SIZE) st = HB_ERR_INVALID_HEADER;
        else if (hd.version != HB_FORMAT_VERSION) st = HB_ERR_INVALID_VERSION;
        else if ((size_t)hd.cbytes > n[k] || hd.cbytes < HB_HEADER_SIZE) st = HB_ERR_INVALID_DATA;           // blosc.go:385-390
        else if (!(hd.flags & HB_FLAG_MEMCPY) && hd.codec != HB_LZ4 && hd.codec != HB_LZ4HC) st = HB_ERR_INVALID_CODEC;   // (Snappy / ZSTD frames: one call per frame)
        else if ((size_t)hd.nbytes > cap[k]) st = HB_ERR_SHORT_BUFFER;
        if (st != HB_OK) { f.preset = st; f.nbytes = 0; continue; }
        const int ts = typesize_override > 0 ? typesize_override : (int)hd.typesize;                          // blosc.go:417-419
        int u = -1;
        if ((hd.flags & HB_FLAG_BITSHUFFLE) && ts > 1) u = HB_OP_BITUNSHUFFLE;                                // blosc.go:422-426
        else if ((hd.flags & HB_FLAG_SHUFFLE) && ts > 1) u = HB_OP_UNSHUFFLE;
        if (u >= 0 && hd.nbytes < (uint32_t)ts) u = -1;                                                       // shuffle.go:17-19: identity
        unf[(size_t)k] = u; tsv[(size_t)k] = ts;
        f.src = (const uint8_t *)d_frame[k] + HB_HEADER_SIZE;
        f.n_src = hd.cbytes - HB_HEADER_SIZE;
        staged_total += al256((size_t)hd.nbytes + 64);
        if (hd.flags & HB_FLAG_MEMCPY) {
            f.preset = f.n_src != hd.nbytes ? HB_ERR_SIZE_MISMATCH : HB_OK;                                   // blosc.go:398-400, :429-431
            continue;
        }
        f.preset = 1;
        const size_t ioff = ((size_t)hd.cbytes + 7) & ~(size_t)7;
        const bool stored_index = n[k] > ioff + 32;
        if (stored_index) { f.index = (const uint8_t *)d_frame[k] + ioff; f.index_bytes = n[k] - ioff; f.nunits = (uint32_t)(((size_t)hd.nbytes + HB_CHUNK - 1) / HB_CHUNK); }
        // no index behind NBytesComp (the default frame shape, blosc.go:369-371): the token discovery rebuilds it for all such frames of the batch
        // in one set of launches (hb_lz4_region.hip `_b` kernels); it is trusted no more than a stored one, and a frame whose chain does not check
        // out (or that was not written chunk-locally) is left to the stream decoder below, as before
        const bool rebuilt = !stored_index && hb_lz4_region_batch_wanted((size_t)f.n_src, hd.nbytes);
        if (rebuilt) {
            rg_off[(size_t)k] = rg_total; rg_total += rg_batch_layout((size_t)f.n_src).total;      // (sized for the smallest regions; the job may use fewer, longer ones)
            rg_stream += (size_t)f.n_src;
            rgi_off[(size_t)k] = rgi_total; rgi_total += al256(hb_lz4_index_bound(hd.nbytes));
            f.nunits = (uint32_t)(((size_t)hd.nbytes + HB_CHUNK - 1) / HB_CHUNK);
        }
        const bool have_index = stored_index || rebuilt;
        f.bun4 = (u == HB_OP_BITUNSHUFFLE && ts == 4 && (hd.nbytes % 32u) == 0 && ((uintptr_t)d_dst[k] & 15u) == 0 && have_index) ? 1 : 0;
        f.ush = (u == HB_OP_UNSHUFFLE && ts <= 4 && (hd.nbytes % (uint32_t)ts) == 0 && ((hd.nbytes / (uint32_t)ts) % HB_CHUNK) == 0 && have_index) ? ts : 0;
        f.post_needed = (u >= 0) ? 1 : 0;
    }
    std::vector<uint32_t> unit0((size_t)nframes);
    const size_t total_units = hb_lz4_dec_batch_units(nframes, h.data(), unit0.data());
    const DecBatchLayout L = dec_batch_layout(nframes, total_units, staged_total, rg_total, rgi_total);
    if (L.total > work_bytes) return HB_ERR_SHORT_BUFFER;
    // jobs of the token discovery: scratch and index of frame k at rgwork + rg_off[k]
    std::vector<RgJob> rgj;
    uint32_t rg_maxreg = 0;
    // region size of the batch: about 16384 regions in all (what one large frame gets), never below the one-frame path's 4 KiB, at most 64 KiB
    const uint64_t rs_min = std::min<uint64_t>(65536, std::max<uint64_t>(4096, rg_stream / 16384));
    for (int k = 0; k < nframes; k++) {
        if (rg_off[(size_t)k] == (size_t)-1) continue;
        DecBatchFrame &f = h[(size_t)k];
        uint8_t *base = w + L.rgwork + rg_off[(size_t)k];
        uint8_t *idx = w + L.rgidx + rgi_off[(size_t)k];
        RgJob j;
        hb_lz4_region_batch_job(base, idx, f.src, (size_t)f.n_src, (size_t)f.nbytes, &j, rs_min);
        rg_maxreg = std::max(rg_maxreg, j.nreg);
        rgj.push_back(j);
        f.index = idx; f.index_bytes = hb_lz4_index_bound((size_t)f.nbytes);
    }
    DecBatchFrame *d_bf = (DecBatchFrame *)(w + L.frames);
    DecPlan *d_plans = (DecPlan *)(w + L.plans);
    hb_filter_job *d_jobs = (hb_filter_job *)(w + L.jobs);
    uint32_t *d_unit_frame = (uint32_t *)(w + L.unit_frame);
    uint8_t *staged = w + L.staged;

    // un-filter / copy jobs, grouped by (operation, typesize): usually one group
    struct Group { int op, ts; std::vector<hb_filter_job> jobs; size_t max_n; };
    std::vector<Group> groups;
    auto add_job = [&](int op, int ts, const hb_filter_job &j) {
        for (auto &g : groups) if (g.op == op && g.ts == ts) { g.jobs.push_back(j); g.max_n = std::max(g.max_n, (size_t)j.n); return; }
        groups.push_back(Group{op, ts, {j}, (size_t)j.n});
    };
    size_t soff = 0;
    int max_ush = 0;
    for (int k = 0; k < nframes; k++) {
        DecBatchFrame &f = h[(size_t)k];
        f.unit0 = unit0[(size_t)k];
        f.plan = d_plans + k;
        if (f.preset != 1 && f.preset != HB_OK) continue;             // refused on the host: nothing to do on the device but the record
        const hb_header &hd = hdrs[k];
        uint8_t *st = staged + soff;
        soff += al256((size_t)hd.nbytes + 64);
        const int u = unf[(size_t)k], ts = tsv[(size_t)k];
        uint8_t *final_dst = (uint8_t *)d_dst[k];
        if (f.preset == HB_OK) {                                      // memcpy frame: payload -> (un-filter) -> dst, straight
            add_job(u >= 0 ? u : 4, u >= 0 ? ts : 1, hb_filter_job{final_dst, f.src, (uint64_t)hd.nbytes, nullptr});
            continue;
        }
        if (u < 0) { f.dst = final_dst; f.serial_dst = final_dst; f.post_needed = 0; continue; }
        const bool fused = f.ush || f.bun4;
        f.dst = fused ? final_dst : st;                               // fused: the indexed decoder writes final bytes, fallbacks stage + gated pass
        f.serial_dst = st;
        max_ush = std::max(max_ush, (int)f.ush);
        add_job(u, ts, hb_filter_job{final_dst, st, (uint64_t)hd.nbytes, fused ? &(d_plans + k)->post : nullptr});
    }
    HB_HIP_TRY(hipMemcpyAsync(d_bf, h.data(), h.size() * sizeof(DecBatchFrame), hipMemcpyHostToDevice, s));
    int rc;
    if (!rgj.empty()) {
        RgJob *d_rgj = (RgJob *)(w + L.rgjobs);
        HB_HIP_TRY(hipMemcpyAsync(d_rgj, rgj.data(), rgj.size() * sizeof(RgJob), hipMemcpyHostToDevice, s));
        HB_HIP_TRY(hipMemsetAsync(w + L.rgidx, 0, rgi_total, s));       // (the indexes start zeroed: a build that fails leaves a header k_bt_dec_plan rejects)
        rc = hb_launch_lz4_region_index_batch(d_rgj, (int)rgj.size(), rg_maxreg, s);
        if (rc) return rc;
    }
    rc = hb_launch_lz4_decode_batch_indexed(nframes, d_bf, d_unit_frame, (uint32_t)total_units, max_ush, s);
    if (rc) return rc;
    hb_prof_begin("k_bt_dec_streams", s);
    hipLaunchKernelGGL(k_bt_dec_streams, dim3((unsigned)nframes), dim3(64), 0, s, (const DecBatchFrame *)d_bf);
    hb_prof_end(s);
    hb_prof_begin("k_bt_dec_finish", s);
    hipLaunchKernelGGL(k_bt_dec_finish, dim3((unsigned)nframes), dim3(64), 0, s, (const DecBatchFrame *)d_bf);
    hb_prof_end(s);
    size_t joff = 0;
    for (auto &g : groups) {
        HB_HIP_TRY(hipMemcpyAsync(d_jobs + joff, g.jobs.data(), g.jobs.size() * sizeof(hb_filter_job), hipMemcpyHostToDevice, s));
        if (g.op == 4) {
            hb_prof_begin("k_bt_copy", s);
            for (size_t j0 = 0; j0 < g.jobs.size(); j0 += 65535) {
                const unsigned ny = (unsigned)std::min<size_t>(65535, g.jobs.size() - j0);
                const unsigned gx = (unsigned)std::min<size_t>(64, (g.max_n + 65535) / 65536);
                hipLaunchKernelGGL(k_bt_copy, dim3(gx ? gx : 1, ny), dim3(256), 0, s, (const hb_filter_job *)(d_jobs + joff + j0));
            }
            hb_prof_end(s);
        } else {
            hb_prof_begin(g.op == HB_OP_UNSHUFFLE ? "filter_unshuffle" : "filter_bitunshuffle", s);
            bool all_gated = true;                                 // every job behind a gate (frames whose un-filter ran inside the indexed decoder): a few workgroups per job
            for (const auto &j : g.jobs) if (!j.gate) { all_gated = false; break; }
            rc = hb_launch_filter_batch(g.op, d_jobs + joff, (int)g.jobs.size(), g.max_n, g.ts, s, all_gated ? 1 : 0);
            hb_prof_end(s);
            if (rc) return rc;
        }
        joff += g.jobs.size();
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}

// ---- host pointers: what a Go caller with many small []byte has.  Stages through cached device buffers (one upload and one download
// per frame: PCIe- and call-overhead-bound -- the device-resident rate is bench.py's `small_frame_batches`), per-frame outcome in rc[]
// exactly as hb_compress_frame / hb_decompress_frame would return it.  Frames the batch does not carry (Snappy / ZSTD) take one call each. ----
namespace {
// gathers `m` byte ranges into one contiguous buffer (frames of a batch before ONE download): job blockIdx.y
struct PackJob { const uint8_t *src; uint64_t dst_off; uint64_t n; };
__global__ __launch_bounds__(256) void k_bt_pack(const PackJob *__restrict__ jobs, uint8_t *__restrict__ out) {
    const PackJob j = jobs[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = blockIdx.x * 4u + (threadIdx.x >> 6), nw = gridDim.x * 4u;
    for (uint64_t off = (uint64_t)wave * 16384u; off < j.n; off += (uint64_t)nw * 16384u)
        wave_copy_g2g(out + j.dst_off + off, j.src + off, (uint32_t)std::min<uint64_t>(16384u, j.n - off), lane);
}
// Host buffers that follow each other EXACTLY (buffer k + 1 starts where buffer k ends: slices of one array, a Go shim's staging slab) go up
// in ONE copy instead of one per frame (a copy call costs ~7 us: 4096 frames of 100 KB spent 28 ms there, 6.9 GB/s host to host).  Nothing
// but the buffers themselves is read: adjacency has to be exact.
bool exactly_adjacent(const std::vector<int> &idx, const void *const *p, const size_t *len) {
    for (size_t i = 0; i + 1 < idx.size(); i++)
        if ((const uint8_t *)p[idx[i]] + len[idx[i]] != (const uint8_t *)p[idx[i + 1]]) return false;
    return idx.size() > 1;
}
struct Held {
    int dev; std::vector<std::pair<void *, size_t>> v;
    explicit Held(int d) : dev(d) {}
    uint8_t *get(size_t bytes) { size_t got = 0; void *p = hb_pool_take(dev, bytes, &got); if (p) v.push_back({p, got}); return (uint8_t *)p; }
    ~Held() { for (auto &h : v) hb_pool_give(dev, h.first, h.second); }
};
}

int hb_compress_frames_batch(int nframes, const void *const *src, const size_t *n, void *const *dst, const size_t *cap, int64_t *rc,
                             int codec, int level, int shuffle, int typesize, unsigned opts, int device) {
    if (nframes < 0 || (nframes && (!src || !n || !dst || !cap || !rc))) return HB_ERR_BAD_ARG;
    if (nframes == 0) return HB_OK;
    int st = hb_select_device(device);
    if (st) return st;
    // frames the batch takes; the others get their answer from the one-frame entry point (argument errors, other codecs)
    std::vector<int> idx;
    for (int k = 0; k < nframes; k++) {
        const bool ok = (codec == HB_LZ4 || codec == HB_LZ4HC) && src[k] && dst[k] && n[k] != 0 && n[k] <= 0xFFFFFFFFull - HB_HEADER_SIZE - n[k] / 255 - 64;
        if (ok) idx.push_back(k); else rc[k] = hb_compress_frame(src[k], n[k], dst[k], cap[k], codec, level, shuffle, typesize, opts, device);
    }
    const int m = (int)idx.size();
    if (m == 0) return HB_OK;
    size_t in_bytes = 0, out_bytes = 0;
    std::vector<size_t> ns((size_t)m), caps((size_t)m), ioff((size_t)m), ooff((size_t)m);
    for (int i = 0; i < m; i++) {
        ns[(size_t)i] = n[idx[(size_t)i]]; caps[(size_t)i] = hb_frame_bound(ns[(size_t)i]) + 64;
        ioff[(size_t)i] = in_bytes; ooff[(size_t)i] = out_bytes;
        in_bytes += al256(ns[(size_t)i] + 16); out_bytes += al256(caps[(size_t)i]);
    }
    Held sc(device);
    const size_t wb = hb_compress_frames_batch_workspace(m, ns.data(), typesize);
    const bool span_in = exactly_adjacent(idx, src, n);
    if (span_in) { in_bytes = 0; for (int i = 0; i < m; i++) { ioff[(size_t)i] = in_bytes; in_bytes += ns[(size_t)i]; } }      // the device copy mirrors the host span
    uint8_t *d_in = sc.get(in_bytes + 256), *d_out = sc.get(out_bytes + 256), *d_work = sc.get(wb), *d_res = sc.get((size_t)m * sizeof(hb_result));
    if (!d_in || !d_out || !d_work || !d_res) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    std::vector<const void *> ps((size_t)m); std::vector<void *> pf((size_t)m);
    if (span_in && hipMemcpyAsync(d_in, src[idx[0]], in_bytes, hipMemcpyHostToDevice, nullptr) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    for (int i = 0; i < m; i++) {
        ps[(size_t)i] = d_in + ioff[(size_t)i]; pf[(size_t)i] = d_out + ooff[(size_t)i];
        if (!span_in && hipMemcpyAsync(d_in + ioff[(size_t)i], src[idx[(size_t)i]], ns[(size_t)i], hipMemcpyHostToDevice, nullptr) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    }
    st = hb_compress_frames_batch_dev(m, ps.data(), ns.data(), pf.data(), caps.data(), codec, level, shuffle, typesize, opts, d_work, wb, (hb_result *)d_res, nullptr);
    if (st) { for (int k : idx) rc[k] = st; return HB_OK; }
    std::vector<hb_result> res((size_t)m);
    if (hipMemcpy(res.data(), d_res, (size_t)m * sizeof(hb_result), hipMemcpyDeviceToHost) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    // download: many small frames are packed on the device, come down in ONE copy and are dealt out by the host (a copy call per frame
    // costs more than the bytes it moves below ~256 KiB); large frames one copy each, straight into the caller's buffers
    size_t total_out = 0;
    std::vector<size_t> outs((size_t)m, 0), poff((size_t)m, 0);
    for (int i = 0; i < m; i++) {
        const int k = idx[(size_t)i];
        const hb_result &r = res[(size_t)i];
        if (r.status) { rc[k] = r.status; continue; }
        const size_t out = (opts & HB_OPT_INDEX_TRAILER) ? r.total_bytes : r.bytes;
        if (out > cap[k]) { rc[k] = HB_ERR_SHORT_BUFFER; continue; }
        outs[(size_t)i] = out; poff[(size_t)i] = total_out; total_out += out; rc[k] = (int64_t)out;
    }
    const bool packed = m >= 16 && total_out / (size_t)m < ((size_t)256 << 10) && total_out <= out_bytes;
    if (packed && total_out) {
        std::vector<PackJob> jobs((size_t)m);
        size_t mx = 0;
        for (int i = 0; i < m; i++) { jobs[(size_t)i] = PackJob{(const uint8_t *)pf[(size_t)i], (uint64_t)poff[(size_t)i], (uint64_t)outs[(size_t)i]}; mx = std::max(mx, outs[(size_t)i]); }
        uint8_t *d_pack = sc.get(total_out + 256), *d_jobs = sc.get((size_t)m * sizeof(PackJob));
        std::vector<uint8_t> host(total_out);
        bool good = d_pack && d_jobs && hipMemcpyAsync(d_jobs, jobs.data(), (size_t)m * sizeof(PackJob), hipMemcpyHostToDevice, nullptr) == hipSuccess;
        for (int j0 = 0; good && j0 < m; j0 += 65535) {
            const unsigned ny = (unsigned)std::min(65535, m - j0), gx = (unsigned)std::min<size_t>(16, (mx + 65535) / 65536);
            hipLaunchKernelGGL(k_bt_pack, dim3(gx ? gx : 1, ny), dim3(256), 0, nullptr, (const PackJob *)d_jobs + j0, d_pack);
        }
        good = good && hipMemcpy(host.data(), d_pack, total_out, hipMemcpyDeviceToHost) == hipSuccess;
        for (int i = 0; i < m; i++) {
            const int k = idx[(size_t)i];
            if (rc[k] < 0) continue;
            if (good) memcpy(dst[k], host.data() + poff[(size_t)i], outs[(size_t)i]); else rc[k] = HB_ERR_HIP;
        }
        return HB_OK;
    }
    for (int i = 0; i < m; i++) {
        const int k = idx[(size_t)i];
        if (rc[k] < 0) continue;
        if (hipMemcpyAsync(dst[k], pf[(size_t)i], outs[(size_t)i], hipMemcpyDeviceToHost, nullptr) != hipSuccess) rc[k] = HB_ERR_HIP;
    }
    if (hipStreamSynchronize(nullptr) != hipSuccess) { for (int k : idx) if (rc[k] >= 0) rc[k] = HB_ERR_HIP; }
    return HB_OK;
}

int hb_decompress_frames_batch(int nframes, const void *const *frame, const size_t *n, void *const *dst, const size_t *cap, int64_t *rc,
                               int typesize_override, int device) {
    if (nframes < 0 || (nframes && (!frame || !n || !dst || !cap || !rc))) return HB_ERR_BAD_ARG;
    if (nframes == 0) return HB_OK;
    int st = hb_select_device(device);
    if (st) return st;
    std::vector<int> idx;
    std::vector<hb_header> hd;
    for (int k = 0; k < nframes; k++) {
        hb_header h;
        bool ok = frame[k] && n[k] >= HB_HEADER_SIZE && hb_parse_header(frame[k], n[k], &h) == HB_OK;
        ok = ok && (size_t)h.cbytes <= n[k] && h.cbytes >= HB_HEADER_SIZE && ((h.flags & HB_FLAG_MEMCPY) || h.codec == HB_LZ4 || h.codec == HB_LZ4HC) &&
             (size_t)h.nbytes <= cap[k] && (dst[k] || !h.nbytes);
        if (ok) { idx.push_back(k); hd.push_back(h); } else rc[k] = hb_decompress_frame(frame[k], n[k], dst[k], cap[k], typesize_override, device);
    }
    const int m = (int)idx.size();
    if (m == 0) return HB_OK;
    size_t in_bytes = 0, out_bytes = 0;
    std::vector<size_t> ns((size_t)m), caps((size_t)m), ioff((size_t)m), ooff((size_t)m);
    for (int i = 0; i < m; i++) {
        ns[(size_t)i] = n[idx[(size_t)i]]; caps[(size_t)i] = hd[(size_t)i].nbytes;
        ioff[(size_t)i] = in_bytes; ooff[(size_t)i] = out_bytes;
        in_bytes += al256(ns[(size_t)i] + 64); out_bytes += al256(caps[(size_t)i] + 64);
    }
    Held sc(device);
    const size_t wb = hb_decompress_frames_batch_workspace(m, hd.data());
    // frames that follow each other exactly go up in one copy (see hb_compress_frames_batch); destinations that follow each other inside
    // their own capacities (dst[k+1] in [dst[k] + nbytes, dst[k] + cap[k]]) get a device image of the same layout and come down in one
    // copy: the bytes between two results lie inside the first one's buffer, which the caller handed over for writing (they are zeroed)
    const bool span_in = exactly_adjacent(idx, frame, n);
    if (span_in) { in_bytes = 0; for (int i = 0; i < m; i++) { ioff[(size_t)i] = in_bytes; in_bytes += ns[(size_t)i]; } }
    bool span_out = m > 1;
    for (int i = 0; span_out && i + 1 < m; i++) {
        const uint8_t *a = (const uint8_t *)dst[idx[(size_t)i]], *b = (const uint8_t *)dst[idx[(size_t)i + 1]];
        span_out = a && b && b >= a + hd[(size_t)i].nbytes && b <= a + cap[idx[(size_t)i]];
    }
    size_t span_bytes = 0;
    if (span_out) {
        const uint8_t *base = (const uint8_t *)dst[idx[0]];
        for (int i = 0; i < m; i++) ooff[(size_t)i] = (size_t)((const uint8_t *)dst[idx[(size_t)i]] - base);
        span_bytes = ooff[(size_t)m - 1] + hd[(size_t)m - 1].nbytes;
        out_bytes = span_bytes + 64;
    }
    uint8_t *d_in = sc.get(in_bytes + 256), *d_out = sc.get(out_bytes + 256), *d_work = sc.get(wb), *d_res = sc.get((size_t)m * sizeof(hb_result));
    if (!d_in || !d_out || !d_work || !d_res) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    std::vector<const void *> pf((size_t)m); std::vector<void *> pd((size_t)m);
    if (span_in && hipMemcpyAsync(d_in, frame[idx[0]], in_bytes, hipMemcpyHostToDevice, nullptr) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    if (span_out && hipMemsetAsync(d_out, 0, span_bytes, nullptr) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    for (int i = 0; i < m; i++) {
        pf[(size_t)i] = d_in + ioff[(size_t)i]; pd[(size_t)i] = d_out + ooff[(size_t)i];
        if (!span_in && hipMemcpyAsync(d_in + ioff[(size_t)i], frame[idx[(size_t)i]], ns[(size_t)i], hipMemcpyHostToDevice, nullptr) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    }
    st = hb_decompress_frames_batch_dev(m, hd.data(), pf.data(), ns.data(), pd.data(), caps.data(), typesize_override, d_work, wb, (hb_result *)d_res, nullptr);
    if (st) { for (int k : idx) rc[k] = st; return HB_OK; }
    std::vector<hb_result> res((size_t)m);
    if (hipMemcpy(res.data(), d_res, (size_t)m * sizeof(hb_result), hipMemcpyDeviceToHost) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; return HB_OK; }
    bool all_ok = true;
    for (int i = 0; i < m; i++) { const hb_result &r = res[(size_t)i]; rc[idx[(size_t)i]] = r.status ? (int64_t)r.status : (int64_t)r.bytes; all_ok = all_ok && !r.status && r.bytes == hd[(size_t)i].nbytes; }
    if (span_out && all_ok) {                                           // (a failed frame's buffer must keep what the caller had in it: copy per frame then)
        if (span_bytes && hipMemcpy(dst[idx[0]], d_out, span_bytes, hipMemcpyDeviceToHost) != hipSuccess) { for (int k : idx) rc[k] = HB_ERR_HIP; }
        return HB_OK;
    }
    for (int i = 0; i < m; i++) {
        const int k = idx[(size_t)i];
        if (rc[k] <= 0) continue;
        if (hipMemcpyAsync(dst[k], pd[(size_t)i], (size_t)rc[k], hipMemcpyDeviceToHost, nullptr) != hipSuccess) rc[k] = HB_ERR_HIP;
    }
    if (hipStreamSynchronize(nullptr) != hipSuccess) { for (int k : idx) if (rc[k] >= 0) rc[k] = HB_ERR_HIP; }
    return HB_OK;
}

}  // extern "C"
