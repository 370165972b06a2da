/*
 * hipblosc.h — C ABI of the MI355X-native go-blosc hot path (Shuffle / BitShuffle filters,
 * LZ4 block codec, frame layer).  This is the drop-in boundary: a cgo shim binds exactly
 * these symbols in place of the reference's internal seams (INTEGRATION.md shows the Go
 * side).  Plain pointers and sizes only; no torch, no HIP types in signatures (a stream
 * is passed as an opaque `void*` = hipStream_t, NULL = the default stream).
 *
 * The reference has no FFI.  Each entry point names the reference interface it replaces
 * (file:line into mrjoshuak/go-blosc).
 *
 * Conventions
 *   - All functions are thread-safe and may be called from arbitrary OS threads.
 *   - The library keeps no caller pointer after a call returns (cgo rule).
 *   - int / int64_t returns: >= 0 success (byte counts where stated), < 0 an HB_ERR_* code.
 *   - "host" entry points take host pointers and stage through the device; `_dev` entry
 *     points take DEVICE pointers on the current HIP device, are asynchronous on `stream`,
 *     and report through an hb_result record in device or pinned memory.
 *   - There is NO CPU fallback: without a usable HIP device every compute entry point
 *     returns HB_ERR_NO_DEVICE.
 */
#ifndef HIPBLOSC_H
#define HIPBLOSC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HB_VERSION_STRING "0.1.0"

/* ---- error codes: one per Go sentinel (blosc.go:125-149) + C-side ones ---- */
#define HB_OK                         0
#define HB_ERR_INVALID_DATA         (-1)   /* ErrInvalidData         blosc.go:127; bare for empty input (:269) and bad cbytes (:385-390) */
#define HB_ERR_INVALID_HEADER       (-2)   /* ErrInvalidHeader       blosc.go:130; bare for < 16 bytes (:166, :297) */
#define HB_ERR_INVALID_VERSION      (-3)   /* ErrInvalidVersion      blosc.go:133; wrapped (:181) */
#define HB_ERR_INVALID_CODEC        (-4)   /* ErrInvalidCodec        blosc.go:136; wrapped (:324, :406) */
#define HB_ERR_SIZE_MISMATCH        (-5)   /* ErrSizeMismatch        blosc.go:139; wrapped (:430) */
#define HB_ERR_DATA_TOO_LARGE       (-6)   /* ErrDataTooLarge        blosc.go:142 (declared, never returned by the reference; used here when sizes overflow uint32) */
#define HB_ERR_COMPRESSION_FAILED   (-7)   /* ErrCompressionFailed   blosc.go:145; wrapped (:338) */
#define HB_ERR_DECOMPRESSION_FAILED (-8)   /* ErrDecompressionFailed blosc.go:148; wrapped (:412) */
#define HB_ERR_NO_DEVICE            (-9)   /* no HIP device / HIP runtime unusable */
#define HB_ERR_HIP                  (-10)  /* a HIP runtime call failed */
#define HB_ERR_BAD_ARG              (-11)  /* NULL pointer, overlapping buffers, bad op ... */
#define HB_ERR_SHORT_BUFFER         (-12)  /* caller's dst / workspace too small */

/* ---- enums (values fixed by the wire format) ---- */
enum hb_codec   { HB_BLOSCLZ = 0, HB_LZ4 = 1, HB_LZ4HC = 2, HB_SNAPPY = 3, HB_ZLIB = 4, HB_ZSTD = 5 };   /* blosc.go:57-64 */
enum hb_shuffle { HB_NOSHUFFLE = 0, HB_SHUFFLE = 1, HB_BITSHUFFLE = 2 };                                   /* blosc.go:89-93 */
enum hb_filter_op { HB_OP_SHUFFLE = 0, HB_OP_UNSHUFFLE = 1, HB_OP_BITSHUFFLE = 2, HB_OP_BITUNSHUFFLE = 3 };
/* header flag bits, blosc.go:110-115 */
enum { HB_FLAG_SHUFFLE = 0x1, HB_FLAG_MEMCPY = 0x2, HB_FLAG_BITSHUFFLE = 0x4, HB_FLAG_SPLIT = 0x8 };
#define HB_HEADER_SIZE 16          /* blosc.go:118-121 */
#define HB_FORMAT_VERSION 2        /* blosc.go:51 */

/* ---- compress-side option bits (`opts` argument) ---- */
#define HB_OPT_INDEX_TRAILER     0x1u  /* append the restart index AFTER cbytes (ignored by the reference decoder, blosc.go:385-393);
                                          lets hb_decompress_* decode the frame chunk-parallel */
#define HB_OPT_REFERENCE_MEMCPY  0x2u  /* memcpy frames store the UN-filtered input exactly as blosc.go:342-345 does
                                          (the reference then corrupts them on decode); default stores the
                                          filtered bytes so the reference Decompress reproduces the input */

#define HB_OPT_NO_FUSION         0x4u  /* run the filter as its own kernel pass instead of fusing it into the LZ4 kernels (diagnostics / A-B timing) */

/* 16-byte frame header, blosc.go:154-162 */
typedef struct hb_header {
    uint8_t  version;    /* 2 */
    uint8_t  codec;      /* hb_codec ("VersionLZ") */
    uint8_t  flags;
    uint8_t  typesize;
    uint32_t nbytes;     /* NBytesOrig */
    uint32_t blocksize;  /* == nbytes */
    uint32_t cbytes;     /* NBytesComp, header included */
} hb_header;

/* completion record of the asynchronous `_dev` entry points (lives in device or pinned memory) */
typedef struct hb_result {
    int32_t  status;        /* HB_OK or HB_ERR_* */
    uint32_t flags;         /* frame flags byte (compress) / bit0: parallel index used (decompress) */
    uint64_t bytes;         /* compress: frame bytes per the header (cbytes) or LZ4 block bytes; decompress: decoded bytes */
    uint64_t total_bytes;   /* compress: bytes written to dst including the index trailer */
    uint64_t reserved;
} hb_result;

/* ---- library / device ---- */
int         hb_init(void);                 /* idempotent; HB_OK or HB_ERR_NO_DEVICE.  Replaces package init, shuffle.go:3-5 */
int         hb_device_count(void);         /* 0 when no device */
void        hb_shutdown(void);             /* frees cached workspaces */
void        hb_pool_limit(size_t bytes);   /* idle device scratch the host-pointer entry points may keep cached (default 12 GiB, or
                                              HIPBLOSC_POOL_MAX_MB); what exceeds it is freed at once, largest buffer first */
size_t      hb_pool_cached_bytes(void);    /* idle scratch currently cached (diagnostics) */
const char *hb_strerror(int code);
const char *hb_version(void);
unsigned    hb_last_result_flags(void);    /* hb_result.flags of the last host-pointer frame decode on this thread
                                              (bit0: the restart index was used) — diagnostics for tests */
/* stage timing for the bench harness (single-threaded use): with enable(1) every kernel stage launched by the
 * `_dev` entry points is bracketed by HIP events on its stream; get(i) returns the stage name and its ms. */
int         hb_profile_enable(int on);
int         hb_profile_count(void);
const char *hb_profile_get(int i, float *ms);

/* pinned host buffers a Go caller can wrap with unsafe.Slice (avoids pageable staging) */
void *hb_host_alloc(size_t bytes);
void  hb_host_free(void *p);

/* ---- filters: replace shuffleBytes / unshuffleBytes / bitShuffle / bitUnshuffle (shuffle.go:16-295)
 *      and the SIMD hooks `func xxx(dst, src []byte, typeSize int) bool` (shuffle_amd64.go:21-41,
 *      shuffle_generic.go:15-52).  COMPLETE semantics incl. tails; typesize<=1 || n<typesize -> plain copy.
 *      dst and src must not overlap. ---- */
int hb_filter(int op, void *dst, const void *src, size_t n, int typesize, int device);
int hb_filter_dev(int op, void *d_dst, const void *d_src, size_t n, int typesize, void *stream);

/* ---- LZ4 block codec: replaces lz4Codec.Compress / Decompress (codec.go:63-84), i.e. the
 *      CodecInterface registered for blosc.LZ4 (codec.go:15-38) ---- */
size_t  hb_lz4_bound(size_t n);                                                       /* lz4.CompressBlockBound, codec.go:65 */
int64_t hb_lz4_compress(const void *src, size_t n, void *dst, size_t cap, int device);   /* -> bytes of ONE spec-valid LZ4 block */
int64_t hb_lz4_decompress(const void *src, size_t n, void *dst, size_t cap, int device); /* -> decoded bytes (<= cap) */

size_t  hb_lz4_compress_workspace(size_t n);
size_t  hb_lz4_decompress_workspace(size_t n_out);
size_t  hb_lz4_decompress_workspace_foreign(size_t n_out);   /* see hb_decompress_frame_workspace_foreign */
/* async; d_index (may be NULL) receives the restart index for this block, index_cap bytes available (see hb_index_bound) */
int hb_lz4_compress_dev(const void *d_src, size_t n, void *d_dst, size_t cap,
                        void *d_index, size_t index_cap,
                        void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
/* async; d_index/index_bytes optional (NULL/0 -> serial single-wavefront decode).
 * All `_dev` entry points load with 16-byte vectors: a source buffer may be READ up to 15 bytes past its last byte
 * (never written), so it must not end exactly at the end of a device allocation's last page -- hipMalloc'd buffers
 * with >= 16 bytes of slack, or any sub-range of a larger allocation, are fine.
 * Every `d_work` must be 256-byte aligned (what hipMalloc returns; HB_ERR_BAD_ARG otherwise): the workspaces hold 16-byte records. */
int hb_lz4_decompress_dev(const void *d_src, size_t n, void *d_dst, size_t cap,
                          const void *d_index, size_t index_bytes,
                          void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
size_t  hb_index_bound(size_t n);   /* bytes of restart index for an n-byte block */

/* ---- the same seam for every codec that runs on the device: the CodecInterface of blosc.LZ4 (codec.go:59-84), blosc.LZ4HC
 *      (codec.go:90-128: `level` picks the search depth like the reference's level map) and blosc.Snappy (codec.go:228-244).
 *      Bare blocks, no frame header; Compress always returns the codec's block (no memcpy rule here: that is the frame layer's,
 *      blosc.go:342).  hb_codec_decompress returns the decoded length (Snappy: the length the block declares; a declared length
 *      above `cap` is HB_ERR_SHORT_BUFFER -- the reference would allocate), HB_ERR_DECOMPRESSION_FAILED on a malformed block,
 *      HB_ERR_INVALID_CODEC for a codec that has no device implementation. ---- */
size_t  hb_codec_bound(int codec, size_t n);
int64_t hb_codec_compress(int codec, int level, const void *src, size_t n, void *dst, size_t cap, int device);
int64_t hb_codec_decompress(int codec, const void *src, size_t n, void *dst, size_t cap, int device);

/* ---- frame layer: replaces compressBackend / decompressBackend (blosc.go:320-434) behind
 *      CompressWithOptions / DecompressWithSize (blosc.go:268-303) ---- */
int     hb_parse_header(const void *frame, size_t n, hb_header *out);                 /* ParseHeader, blosc.go:165-185 */
void    hb_header_bytes(const hb_header *h, void *out16);                             /* (*Header).Bytes, blosc.go:188-198 */
size_t  hb_frame_bound(size_t n);                                                     /* 16 + lz4 bound + index trailer */
/* returns bytes written to dst (cbytes, plus the trailer when HB_OPT_INDEX_TRAILER).
 * codec HB_LZ4: everything on the device.  codec HB_ZSTD (BASELINE.json config 5): the filter runs on the device, ZSTD stays
 * a host codec as in the reference (codec.go:173-222) -- one zstd frame per 16 MiB slice, compressed by host threads while
 * the next slices are still being copied back; the concatenation decodes with zstd.Decoder.DecodeAll.  Needs libzstd.so.1
 * at run time (else HB_ERR_INVALID_CODEC); host-pointer entry points only. */
int64_t hb_compress_frame(const void *src, size_t n, void *dst, size_t cap,
                          int codec, int level, int shuffle, int typesize,
                          unsigned opts, int device);
/* returns decoded bytes (== header nbytes); typesize_override <= 0 -> header typesize (blosc.go:417-419) */
int64_t hb_decompress_frame(const void *frame, size_t n, void *dst, size_t cap,
                            int typesize_override, int device);

size_t  hb_compress_frame_workspace(size_t n);
size_t  hb_decompress_frame_workspace(size_t n_out);
/* the same plus ~5 bytes per output byte: with a workspace of this size an LZ4 / LZ4HC frame that has no restart index and was
 * not written chunk-locally (what the reference's lz4.CompressBlock writes, codec.go:63-75) is decoded in parallel as well
 * (symbolic decode from the verified token chain: ~2 bytes per output byte); with the smaller workspace such a frame goes to one
 * wavefront.  It also holds the token store of the discovery (~2.7 bytes per stream byte), with which the index of a frame of THIS
 * library that carries none is rebuilt without a second walk (decode ~6 % faster).  A Snappy frame without the unit index decodes in parallel
 * with either workspace when its encoder compressed 64 KiB blocks that share nothing (golang/snappy, libsnappy); the larger one adds the symbolic
 * decoder for streams whose copies cross those blocks (offsets in 16 bits).  The host-pointer entry points pick the size themselves. */
size_t  hb_decompress_frame_workspace_foreign(size_t n_out);
int hb_compress_frame_dev(const void *d_src, size_t n, void *d_frame, size_t cap,
                          int codec, int level, int shuffle, int typesize, unsigned opts,
                          void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
/* `n` = bytes available at d_frame (>= cbytes; may include the trailer); cap = room at d_dst.  The 16 header bytes are
 * read back to the host first (one small D2H + stream sync: the launch shapes depend on them), so header errors come
 * back as the return value; everything after that is asynchronous and reports through *d_result. */
int hb_decompress_frame_dev(const void *d_frame, size_t n, void *d_dst, size_t cap,
                            int typesize_override,
                            void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
/* the same for a caller that already has the 16 header bytes on the host (every Go caller does: the frame came from host memory):
 * `hdr` = hb_parse_header() of them.  No read-back and no stream synchronisation: the call only enqueues work on `stream`.
 * Header checks of blosc.go:385-390 / :403-407 come back as the return value, everything else through *d_result. */
int hb_decompress_frame_dev_hdr(const hb_header *hdr, const void *d_frame, size_t n, void *d_dst, size_t cap,
                                int typesize_override,
                                void *d_work, size_t work_bytes, hb_result *d_result, void *stream);

/* batches of independent frames, frame k -> device k mod hb_device_count(): what a caller with an
 * 8 GiB array does (8 frames of <= 4 GiB - 1, blosc.go:159-161: the sizes are uint32), one Compress / Decompress call
 * (blosc.go:257-303) per frame.  One host thread per device, each with its own hb_queue of 3 frames in flight; no
 * device-to-device traffic.  Per-frame results in rc[] (what hb_compress_frame / hb_decompress_frame would return);
 * the call itself returns HB_OK unless its arguments are unusable. */
int hb_compress_frames_multi(int nframes, const void *const *src, const size_t *n,
                             void *const *dst, const size_t *cap, int64_t *rc,
                             int codec, int level, int shuffle, int typesize, unsigned opts);
int hb_decompress_frames_multi(int nframes, const void *const *frame, const size_t *n,
                               void *const *dst, const size_t *cap, int64_t *rc, int typesize_override);

/* ---- batches of SMALL frames in one set of launches (SURVEY.md §8 f1 "frame batches") ----
 * The reference's own benchmark is a 100 000-byte frame (blosc_test.go:363-413): one such frame is 25 chunks of work, far too little
 * for four kernel launches of its own.  These entry points put `nframes` independent Compress / Decompress calls (blosc.go:257-303,
 * one frame each, same Options for all) through ONE set of launches: the chunks and index units of all frames form one flat work
 * space, the per-frame scan / header / memcpy rule (blosc.go:342-371) is a segmented scan.  Every frame is byte-identical to what
 * hb_compress_frame_dev would have written for it.  d_src / d_frame / d_dst are HOST arrays of DEVICE pointers; asynchronous on `stream`;
 * d_results: `nframes` hb_result records in device (or pinned) memory, one per frame, as the one-frame entry points fill them.
 * Codecs LZ4 and LZ4HC (HB_ERR_INVALID_CODEC otherwise: Snappy / ZSTD frames go one call per frame).  An argument error of any frame
 * refuses the whole compress batch before anything is launched (the return value says which error). */
size_t hb_compress_frames_batch_workspace(int nframes, const size_t *n, int typesize);
int hb_compress_frames_batch_dev(int nframes, const void *const *d_src, const size_t *n, void *const *d_frame, const size_t *cap,
                                 int codec, int level, int shuffle, int typesize, unsigned opts,
                                 void *d_work, size_t work_bytes, hb_result *d_results, void *stream);
/* headers of device-resident frames: one gather + ONE D2H + one stream synchronisation for the whole batch (hb_decompress_frame_dev pays
 * one per frame); hdrs[k] = ParseHeader (blosc.go:165-185) of frame k where rc[k] == HB_OK.  d_scratch: >= 32 * nframes + 256 bytes. */
int hb_frames_batch_headers_dev(int nframes, const void *const *d_frame, const size_t *n, hb_header *hdrs, int *rc,
                                void *d_scratch, size_t scratch_bytes, void *stream);
/* hdrs: the parsed headers (host memory: a Go caller has them, its frames came from host memory; else hb_frames_batch_headers_dev).
 * Frames that carry the restart index (HB_OPT_INDEX_TRAILER) are decoded chunk-parallel, all frames' units in one launch; frames without
 * one -- the default frame shape, and frames of other writers (the reference) -- by ONE wavefront per frame, all frames at once.  A frame
 * the host can refuse from its header (blosc.go:385-390, :403-407; destination too small) gets its error in d_results[k] and does not
 * disturb the others; everything else reports as hb_decompress_frame_dev does (blosc.go:377-434).  no stream synchronisation. */
size_t hb_decompress_frames_batch_workspace(int nframes, const hb_header *hdrs);
int hb_decompress_frames_batch_dev(int nframes, const hb_header *hdrs, const void *const *d_frame, const size_t *n,
                                   void *const *d_dst, const size_t *cap, int typesize_override,
                                   void *d_work, size_t work_bytes, hb_result *d_results, void *stream);

/* the same with HOST pointers (what a Go caller with many small []byte has): stages through cached device buffers, per-frame outcome in
 * rc[k] exactly as hb_compress_frame / hb_decompress_frame would return it (frames the batch does not carry -- other codecs, argument
 * errors -- are answered by those entry points, one call each).  Returns HB_OK unless the arguments as a whole are unusable. */
int hb_compress_frames_batch(int nframes, const void *const *src, const size_t *n, void *const *dst, const size_t *cap, int64_t *rc,
                             int codec, int level, int shuffle, int typesize, unsigned opts, int device);
int hb_decompress_frames_batch(int nframes, const void *const *frame, const size_t *n, void *const *dst, const size_t *cap, int64_t *rc,
                               int typesize_override, int device);

/* ---- pipelined host API: frames in flight on their own streams ----
 * The one-call entry points above move H2D -> kernels -> D2H back to back, so a caller sees n / (t_h2d + t_k + t_d2h).
 * A queue keeps `depth` frames in flight, each on its own stream with its own device buffers: the upload of frame k+1
 * runs while frame k computes and frame k-1 downloads (PCIe is full duplex), so a stream of frames moves at the
 * speed of the slower PCIe direction.  Same frame semantics as hb_compress_frame / hb_decompress_frame
 * (blosc.go:320-434); LZ4 only.  src / dst must stay valid and untouched until hb_queue_wait() on the ticket
 * returns; use hb_host_alloc() buffers (pageable memory works, but its copies do not overlap).
 * A queue belongs to one thread at a time; different queues are independent. */
typedef struct hb_queue hb_queue;
hb_queue *hb_queue_create(int device, int depth, size_t max_nbytes);   /* NULL on failure; frames up to max_nbytes uncompressed bytes */
/* flags: HB_QUEUE_FOREIGN_FRAMES = every slot gets hb_decompress_frame_workspace_foreign(max_nbytes) bytes of workspace (~5x the
 * frame size more device memory per slot), so that LZ4 frames of other writers -- no restart index, not chunk-local -- decode in parallel */
#define HB_QUEUE_FOREIGN_FRAMES 1u
hb_queue *hb_queue_create_ex(int device, int depth, size_t max_nbytes, unsigned flags);
void hb_queue_destroy(hb_queue *q);                                     /* finishes what is in flight */
/* enqueue one frame; returns a ticket >= 0 (tickets count up from 0) or an HB_ERR_* code.  When all `depth` slots are
 * in flight the oldest one is finished first (its result is kept for its hb_queue_wait). */
int64_t hb_queue_compress(hb_queue *q, const void *src, size_t n, void *dst, size_t cap,
                          int codec, int level, int shuffle, int typesize, unsigned opts);
int64_t hb_queue_decompress(hb_queue *q, const void *frame, size_t n, void *dst, size_t cap, int typesize_override);
/* blocks until the frame is in dst; returns what hb_compress_frame / hb_decompress_frame would have returned.
 * Tickets may be waited for in any order, each once.  A ticket whose slot has been re-used by a later submission (more
 * than `depth` submissions ago) was finished at that moment -- its data is in its dst -- and its return value is kept
 * for the newest 4 * depth such tickets; older ones answer HB_ERR_BAD_ARG. */
int64_t hb_queue_wait(hb_queue *q, int64_t ticket);

/* ---- frames in the C-Blosc-1 wire format (c-blosc 1.x: bstarts table, blocks split into `typesize` streams, the
 *      filter per block) -- what go-blosc's README.md:20 claims to be compatible with and blosc.go does not implement.  Codec
 *      formats LZ4 / LZ4HC and memcpyed frames; byte shuffle, bit shuffle or none, any typesize.  Not a seam of the
 *      reference (it has none for this): an extension next to hb_decompress_frame. ---- */
typedef struct hb_cblosc_header {
    uint8_t  version, versionlz, flags, typesize;   /* flags: 0x01 shuffle, 0x02 memcpyed, 0x04 bitshuffle, 0x10 not split */
    uint32_t nbytes, blocksize, cbytes;
    uint32_t codec_format;                          /* flags >> 5: 0 blosclz, 1 lz4 / lz4hc, 2 snappy, 3 zlib, 4 zstd */
} hb_cblosc_header;
int     hb_cblosc_parse_header(const void *frame, size_t n, hb_cblosc_header *out);          /* host-only */
size_t  hb_cblosc_decompress_workspace(size_t nbytes, size_t blocksize, size_t typesize);
int     hb_cblosc_decompress_dev(const hb_cblosc_header *hdr, const void *d_frame, size_t n, void *d_dst, size_t cap,
                                 void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
/* host pointers: returns the decoded bytes (== nbytes of the header) or HB_ERR_*: HB_ERR_INVALID_CODEC for the codec formats
 * that are not LZ4, HB_ERR_DECOMPRESSION_FAILED for anything blosc_decompress() answers with a negative number */
int64_t hb_cblosc_decompress(const void *frame, size_t n, void *dst, size_t cap, int device);
/* writing the format: a frame that blosc_decompress() of c-blosc 1.x (python-blosc, numcodecs ...) reads.  shuffle: 0 none, 1 byte
 * shuffle, 2 bit shuffle (BLOSC_NOSHUFFLE / BLOSC_SHUFFLE / BLOSC_BITSHUFFLE); LZ4 streams; block size 4096 x typesize (split) or
 * 4096 (not split), so that every stream is one chunk of this library's encoder; n below 2 GiB (c-blosc's limit).  Returns the
 * frame's bytes (its cbytes field). */
size_t  hb_cblosc_bound(size_t n, int typesize);
size_t  hb_cblosc_compress_workspace(size_t n, int shuffle, int typesize);
int     hb_cblosc_compress_dev(const void *d_src, size_t n, void *d_frame, size_t cap, int shuffle, int typesize,
                               void *d_work, size_t work_bytes, hb_result *d_result, void *stream);
int64_t hb_cblosc_compress(const void *src, size_t n, void *dst, size_t cap, int shuffle, int typesize, int device);

#ifdef __cplusplus
}
#endif
#endif /* HIPBLOSC_H */
