//go:build hip

// blosc_hip.go — the cgo shim a go-blosc maintainer adds (build tag `hip`) to run the Shuffle/BitShuffle
// filters and the LZ4 codec on an MI355X through libhipblosc.so (include/hipblosc.h).
//
// It plugs into the reference's own two seams and changes nothing else:
//   * the codec plugin seam      codec.go:15-38   RegisterCodec(LZ4 / LZ4HC / Snappy, ...)
//   * the filter hook seam       shuffle.go:26-57, :154-174 (hooks declared in shuffle_amd64.go:21-41,
//                                stubs in shuffle_generic.go:15-52)  ->  the four xxxHIP functions below
// plus an optional fused fast path (CompressHIP / DecompressHIP) that replaces compressBackend /
// decompressBackend (blosc.go:320-434) in one device round trip.
//
// STATUS: written against the C ABI, NOT compiled — this image has no Go toolchain and the reference's
// module dependencies are not vendored (SURVEY.md §0.9).  The same ABI is exercised from Python (ctypes)
// by tests/ and bench.py.
package blosc

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../lib -lhipblosc -Wl,-rpath,${SRCDIR}/../lib
#include <stdlib.h>
#include "hipblosc.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// MinOffloadBytes: below this the PCIe round trip costs more than the pure-Go path (config 1 of
// BASELINE.json, 100 KB, stays on the CPU).  Tunable by the caller.
var MinOffloadBytes = 1 << 20

// Device used by the host-pointer entry points.
var Device = 0

// ForceDeviceDecode sends the frames only a single wavefront can decode (no restart index AND a payload below 256 KiB -- 16 KiB when it decodes to 2 MiB or more --, or a
// foreign Snappy block) to the device too.
var ForceDeviceDecode = false

// hasRestartIndex: is there an "HBIX" index after NBytesComp (written by CompressHIP(..., withIndex=true))?
// (the LZ4 / LZ4HC trailer starts with "HBIX", the Snappy one with "HBSX": include/hipblosc.h, csrc/hb_format.h)
func hasRestartIndex(data []byte, h *Header) bool {
	off := (int(h.NBytesComp) + 7) &^ 7
	return len(data) >= off+40 && (string(data[off:off+4]) == "HBIX" || string(data[off:off+4]) == "HBSX")
}

var useHIP bool

func init() {
	useHIP = C.hb_init() == C.HB_OK && C.hb_device_count() > 0
	if useHIP {
		RegisterCodec(LZ4, &hipLZ4{fallback: &lz4Codec{}}) // codec.go:36-38
		RegisterCodec(LZ4HC, &hipCodec{id: LZ4HC, name: "lz4hc", fallback: &lz4hcCodec{}})
		RegisterCodec(Snappy, &hipCodec{id: Snappy, name: "snappy", fallback: &snappyCodec{}})
	}
}

// hbError maps a C-ABI code back to the reference's sentinels, keeping the bare-vs-wrapped distinction
// (bare ErrInvalidData / ErrInvalidHeader: blosc.go:269-271, :297-299, :385-390; the rest wrapped with %w).
func hbError(code C.int64_t) error {
	switch code {
	case C.HB_ERR_INVALID_DATA:
		return ErrInvalidData
	case C.HB_ERR_INVALID_HEADER:
		return ErrInvalidHeader
	case C.HB_ERR_INVALID_VERSION:
		return fmt.Errorf("%w: (device path)", ErrInvalidVersion)
	case C.HB_ERR_INVALID_CODEC:
		return fmt.Errorf("%w: (device path)", ErrInvalidCodec)
	case C.HB_ERR_SIZE_MISMATCH:
		return fmt.Errorf("%w: (device path)", ErrSizeMismatch)
	case C.HB_ERR_DATA_TOO_LARGE:
		return ErrDataTooLarge
	case C.HB_ERR_COMPRESSION_FAILED:
		return fmt.Errorf("%w: hipblosc", ErrCompressionFailed)
	case C.HB_ERR_DECOMPRESSION_FAILED:
		return fmt.Errorf("%w: hipblosc", ErrDecompressionFailed)
	default:
		return fmt.Errorf("hipblosc: %s", C.GoString(C.hb_strerror(C.int(code))))
	}
}

func ptr(b []byte) unsafe.Pointer {
	if len(b) == 0 {
		return nil
	}
	return unsafe.Pointer(&b[0]) // Go memory is only borrowed for the duration of the call (cgo rule)
}

// ---------------------------------------------------------------------------------------------
// codec plugin: CodecInterface (codec.go:15-24) for blosc.LZ4
// ---------------------------------------------------------------------------------------------
type hipLZ4 struct{ fallback CodecInterface }

func (c *hipLZ4) Name() string { return "lz4" } // codec.go:61

// Compress: codec.go:63-75.  `level` is ignored, exactly as the reference's LZ4 codec ignores it.
func (c *hipLZ4) Compress(data []byte, level int) ([]byte, error) {
	if !useHIP || len(data) < MinOffloadBytes {
		return c.fallback.Compress(data, level)
	}
	buf := make([]byte, int(C.hb_lz4_bound(C.size_t(len(data))))) // codec.go:65
	n := C.hb_lz4_compress(ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(Device))
	if n < 0 {
		return nil, fmt.Errorf("lz4 compress: %w", hbError(n)) // codec.go:67-69
	}
	return buf[:n], nil
}

// Decompress: codec.go:77-84; returns buf[:n] and lets the frame layer detect a size mismatch.
func (c *hipLZ4) Decompress(data []byte, expectedSize int) ([]byte, error) {
	if !useHIP || expectedSize < MinOffloadBytes {
		return c.fallback.Decompress(data, expectedSize)
	}
	buf := make([]byte, expectedSize)
	n := C.hb_lz4_decompress(ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(Device))
	if n < 0 {
		return nil, fmt.Errorf("lz4 decompress: %w", hbError(n))
	}
	return buf[:n], nil
}

// hipCodec: the same seam for blosc.LZ4HC (codec.go:90-128; `level` picks the search depth) and blosc.Snappy (codec.go:228-244).
type hipCodec struct {
	id       Codec
	name     string
	fallback CodecInterface
}

func (c *hipCodec) Name() string { return c.name } // codec.go:92, :230

func (c *hipCodec) Compress(data []byte, level int) ([]byte, error) {
	if !useHIP || len(data) < MinOffloadBytes {
		return c.fallback.Compress(data, level)
	}
	buf := make([]byte, int(C.hb_codec_bound(C.int(c.id), C.size_t(len(data)))))
	n := C.hb_codec_compress(C.int(c.id), C.int(level), ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(Device))
	if n < 0 {
		return nil, fmt.Errorf("%s compress: %w", c.name, hbError(n)) // codec.go:113-115
	}
	return buf[:n], nil
}

func (c *hipCodec) Decompress(data []byte, expectedSize int) ([]byte, error) {
	if !useHIP || expectedSize < MinOffloadBytes {
		return c.fallback.Decompress(data, expectedSize)
	}
	buf := make([]byte, expectedSize)
	n := C.hb_codec_decompress(C.int(c.id), ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(Device))
	if n == C.HB_ERR_SHORT_BUFFER { // a Snappy block that declares more than expectedSize: snappy.Decode would allocate (codec.go:238)
		return c.fallback.Decompress(data, expectedSize)
	}
	if n < 0 {
		return nil, fmt.Errorf("%s decompress: %w", c.name, hbError(n))
	}
	return buf[:n], nil
}

// ---------------------------------------------------------------------------------------------
// filter hooks: same contract as shuffleBytesAVX2 & co. (shuffle_amd64.go:21-41): dst is pre-allocated with
// len(dst) == len(src); return true = handled.  hb_filter implements the COMPLETE semantics including
// leftover elements and tail bytes, so the scalar finisher of shuffle.go:42-55 / :162-173 has nothing left
// to do; the call sites test `useHIP && n >= MinOffloadBytes` next to `useAVX2` / `useNEON`.
// ---------------------------------------------------------------------------------------------
func filterHIP(op C.int, dst, src []byte, typeSize int) bool {
	if !useHIP || len(src) < MinOffloadBytes || len(dst) != len(src) {
		return false
	}
	return C.hb_filter(op, ptr(dst), ptr(src), C.size_t(len(src)), C.int(typeSize), C.int(Device)) == C.HB_OK
}

func shuffleBytesHIP(dst, src []byte, typeSize int) bool   { return filterHIP(C.HB_OP_SHUFFLE, dst, src, typeSize) }
func unshuffleBytesHIP(dst, src []byte, typeSize int) bool { return filterHIP(C.HB_OP_UNSHUFFLE, dst, src, typeSize) }
func bitShuffleHIP(dst, src []byte, typeSize int) bool     { return filterHIP(C.HB_OP_BITSHUFFLE, dst, src, typeSize) }
func bitUnshuffleHIP(dst, src []byte, typeSize int) bool   { return filterHIP(C.HB_OP_BITUNSHUFFLE, dst, src, typeSize) }

// ---------------------------------------------------------------------------------------------
// fused frame path: one H2D, filter + LZ4 + header on the device, one D2H.  Drop-in for
// compressBackend / decompressBackend (blosc.go:320-374, :377-434) when opts.Codec == LZ4.
// ---------------------------------------------------------------------------------------------

// CompressHIP has CompressWithOptions' semantics (blosc.go:268-286).  withIndex appends the restart index
// after NBytesComp (ignored by every go-blosc decoder, blosc.go:385-393) so DecompressHIP can decode the
// frame chunk-parallel; the returned slice is then longer than NBytesComp.
func CompressHIP(data []byte, opts Options, withIndex bool) ([]byte, error) {
	if len(data) == 0 {
		return nil, ErrInvalidData // blosc.go:269-271
	}
	// device codecs: LZ4 (codec.go:59-84), LZ4HC (codec.go:90-128: the level picks the search depth) and Snappy (codec.go:228-244)
	if !useHIP || !deviceCodec(opts.Codec) || len(data) < MinOffloadBytes {
		return CompressWithOptions(data, opts)
	}
	var o C.uint
	if withIndex {
		o |= C.HB_OPT_INDEX_TRAILER
	}
	buf := make([]byte, int(C.hb_frame_bound(C.size_t(len(data)))))
	n := C.hb_compress_frame(ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)),
		C.int(opts.Codec), C.int(opts.Level), C.int(opts.Shuffle), C.int(opts.TypeSize), o, C.int(Device))
	if n < 0 {
		return nil, hbError(n)
	}
	return buf[:n], nil
}

func deviceCodec(c Codec) bool { return c == LZ4 || c == LZ4HC || c == Snappy }

// DecompressHIP has DecompressWithSize's semantics (blosc.go:296-303).
func DecompressHIP(data []byte, typeSize int) ([]byte, error) {
	if len(data) < HeaderSize {
		return nil, ErrInvalidHeader // blosc.go:297-299
	}
	h, err := ParseHeader(data)
	if err != nil {
		return nil, err
	}
	if !useHIP || int(h.NBytesOrig) < MinOffloadBytes || (!h.IsMemcpy() && !deviceCodec(Codec(h.VersionLZ))) {
		return DecompressWithSize(data, typeSize)
	}
	// An LZ4 block is one serial chain.  With the restart index the device decodes it chunk-parallel; without one it
	// first finds and verifies the token chain itself (payloads from 256 KiB, or from 16 KiB when they decode to 2 MiB and more: csrc/hb_lz4_region.hip) and then either
	// rebuilds the index (frames this library wrote) or decodes symbolically (frames the CPU path wrote, hb_lz4_sym.hip):
	// ~100 / ~240 GB/s device-resident at 1 GiB.  Below that size only ONE wavefront can work on it (~0.15-0.4 GB/s, measured),
	// slower than the pure-Go decoder: those stay on the CPU unless the caller insists.  Snappy frames of other writers stay there
	// too: the device decodes them in parallel when their encoder compressed 64 KiB blocks (golang/snappy, libsnappy: hb_snappy.hip,
	// 178-437 GB/s at 1 GiB), but what this package's own CPU path writes (klauspost's one-block streams) may hold 4-byte offsets, which
	// leave the block to one wavefront -- and a frame does not say who wrote it.  ForceDeviceDecode sends them anyway.
	payload := int(h.NBytesComp) - HeaderSize
	parallel := hasRestartIndex(data, h) || (Codec(h.VersionLZ) != Snappy && (payload >= 256<<10 || (payload >= 16<<10 && h.NBytesOrig >= 2<<20)))
	if !h.IsMemcpy() && !parallel && !ForceDeviceDecode {
		return DecompressWithSize(data, typeSize)
	}
	buf := make([]byte, int(h.NBytesOrig))
	n := C.hb_decompress_frame(ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(typeSize), C.int(Device))
	if n < 0 {
		return nil, hbError(n)
	}
	return buf[:n], nil
}

// CompressFramesHIP: independent frames, frame k on device k mod hb_device_count() (no collective;
// SURVEY.md §8e).  Used for inputs >= 4 GiB, which the uint32 header cannot hold in one frame.
func CompressFramesHIP(frames [][]byte, opts Options, withIndex bool) ([][]byte, []error) {
	n := len(frames)
	out := make([][]byte, n)
	errs := make([]error, n)
	if n == 0 {
		return out, errs
	}
	srcs := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	dsts := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&srcs[0]))
	defer C.free(unsafe.Pointer(&dsts[0]))
	lens := make([]C.size_t, n)
	caps := make([]C.size_t, n)
	rcs := make([]C.int64_t, n)
	pin := make([]unsafe.Pointer, 0, 2*n) // C-allocated staging: Go pointers may not be stored in C memory
	for k, f := range frames {
		lens[k] = C.size_t(len(f))
		caps[k] = C.hb_frame_bound(lens[k])
		srcs[k] = C.hb_host_alloc(lens[k])
		dsts[k] = C.hb_host_alloc(caps[k])
		if srcs[k] == nil || dsts[k] == nil {
			// no pinned memory for this frame (hb_host_alloc answers nil): it stays out of the batch -- a nil source makes
			// hb_compress_frames_multi answer HB_ERR_BAD_ARG for it without touching it -- and goes through the one-call path below
			C.hb_host_free(srcs[k])
			C.hb_host_free(dsts[k])
			srcs[k], dsts[k], lens[k] = nil, nil, 0
			continue
		}
		pin = append(pin, srcs[k], dsts[k])
		copy(unsafe.Slice((*byte)(srcs[k]), len(f)), f)
	}
	var o C.uint
	if withIndex {
		o |= C.HB_OPT_INDEX_TRAILER
	}
	C.hb_compress_frames_multi(C.int(n), &srcs[0], &lens[0], &dsts[0], &caps[0], &rcs[0],
		C.int(opts.Codec), C.int(opts.Level), C.int(opts.Shuffle), C.int(opts.TypeSize), o)
	for k := range frames {
		if srcs[k] == nil { // not in the batch (see above): pageable one-call path
			out[k], errs[k] = CompressHIP(frames[k], opts, withIndex)
		} else if rcs[k] < 0 {
			errs[k] = hbError(rcs[k])
		} else {
			out[k] = append([]byte(nil), unsafe.Slice((*byte)(dsts[k]), int(rcs[k]))...)
		}
	}
	for _, p := range pin {
		C.hb_host_free(p)
	}
	return out, errs
}

// DecompressFrames is the inverse of CompressFrames: independent frames, frame k on device k mod hb_device_count(),
// one Decompress (blosc.go:291-303) per frame; errs[k] carries the reference's sentinel for frame k.
func DecompressFrames(frames [][]byte) (out [][]byte, errs []error) {
	n := len(frames)
	out = make([][]byte, n)
	errs = make([]error, n)
	if n == 0 {
		return out, errs
	}
	srcs := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	dsts := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&srcs[0]))
	defer C.free(unsafe.Pointer(&dsts[0]))
	lens := make([]C.size_t, n)
	caps := make([]C.size_t, n)
	rcs := make([]C.int64_t, n)
	for k, f := range frames {
		lens[k] = C.size_t(len(f))
		caps[k] = 1
		if h, err := ParseHeader(f); err == nil {
			caps[k] = C.size_t(h.NBytesOrig) + 1
		}
		// NBytesOrig is untrusted (up to 4 GiB of pinned memory per frame): when the allocation fails the frame stays out of
		// the batch (a nil frame pointer gets HB_ERR_BAD_ARG from hb_decompress_frames_multi, untouched) and takes the one-call path
		srcs[k] = C.hb_host_alloc(lens[k] + 1)
		dsts[k] = C.hb_host_alloc(caps[k])
		if srcs[k] == nil || dsts[k] == nil {
			C.hb_host_free(srcs[k])
			C.hb_host_free(dsts[k])
			srcs[k], dsts[k], lens[k], caps[k] = nil, nil, 0, 0
			continue
		}
		copy(unsafe.Slice((*byte)(srcs[k]), len(f)), f)
	}
	C.hb_decompress_frames_multi(C.int(n), &srcs[0], &lens[0], &dsts[0], &caps[0], &rcs[0], 0)
	for k := range frames {
		if srcs[k] == nil {
			out[k], errs[k] = DecompressHIP(frames[k], 0)
			continue
		}
		if rcs[k] < 0 {
			errs[k] = hbError(rcs[k])
		} else {
			out[k] = append([]byte(nil), unsafe.Slice((*byte)(dsts[k]), int(rcs[k]))...)
		}
		C.hb_host_free(srcs[k])
		C.hb_host_free(dsts[k])
	}
	return out, errs
}

// ---------------------------------------------------------------------------------------------
// Pipelined frames (hb_queue_*): `depth` frames in flight on one device, uploads / kernels / downloads
// overlapped — for callers that stream many frames (chunked arrays, inputs >= 4 GiB cut into frames).
// Buffers handed to Submit* must be pinned (PinnedBytes) and stay untouched until Wait returns.
// ---------------------------------------------------------------------------------------------

// PinnedBytes returns a []byte backed by hb_host_alloc memory (free with FreePinned).  It holds no Go pointers
// and is not moved by the GC, so the library may keep its address across the Submit/Wait pair.
// A failed allocation (hipHostMalloc refused: n == 0, or the pinned pool is exhausted) returns nil, never a slice over a nil pointer.
func PinnedBytes(n int) []byte {
	if n <= 0 {
		return nil
	}
	p := C.hb_host_alloc(C.size_t(n))
	if p == nil {
		return nil
	}
	return unsafe.Slice((*byte)(p), n)
}

// FreePinned releases a slice returned by PinnedBytes (nil / empty: nothing to do).
func FreePinned(b []byte) {
	if len(b) > 0 {
		C.hb_host_free(unsafe.Pointer(&b[0]))
	}
}

type FrameQueue struct{ q *C.hb_queue }

func NewFrameQueue(device, depth int, maxFrameBytes int) (*FrameQueue, error) {
	q := C.hb_queue_create(C.int(device), C.int(depth), C.size_t(maxFrameBytes))
	if q == nil {
		return nil, fmt.Errorf("%w: hb_queue_create", ErrCompressionFailed)
	}
	return &FrameQueue{q}, nil
}
func (fq *FrameQueue) Close() { C.hb_queue_destroy(fq.q) }

// SubmitCompress enqueues CompressWithOptions(src, opts) into dst (cap >= hb_frame_bound(len(src))); returns a ticket.
func (fq *FrameQueue) SubmitCompress(src, dst []byte, opts Options, withIndex bool) (int64, error) {
	var o C.uint
	if withIndex {
		o |= C.HB_OPT_INDEX_TRAILER
	}
	t := C.hb_queue_compress(fq.q, ptr(src), C.size_t(len(src)), ptr(dst), C.size_t(len(dst)),
		C.int(opts.Codec), C.int(opts.Level), C.int(opts.Shuffle), C.int(opts.TypeSize), o)
	if t < 0 {
		return 0, hbError(t)
	}
	return int64(t), nil
}

// SubmitDecompress enqueues DecompressWithSize(frame, typeSize) into dst (cap >= NBytesOrig).
func (fq *FrameQueue) SubmitDecompress(frame, dst []byte, typeSize int) (int64, error) {
	t := C.hb_queue_decompress(fq.q, ptr(frame), C.size_t(len(frame)), ptr(dst), C.size_t(len(dst)), C.int(typeSize))
	if t < 0 {
		return 0, hbError(t)
	}
	return int64(t), nil
}

// Wait blocks until the ticket's frame is in its dst and returns the byte count (or the reference's error).
func (fq *FrameQueue) Wait(ticket int64) (int, error) {
	n := C.hb_queue_wait(fq.q, C.int64_t(ticket))
	if n < 0 {
		return 0, hbError(n)
	}
	return int(n), nil
}

// ---------------------------------------------------------------------------------------------
// C-Blosc-1 wire format (what README.md:20 promises; blosc.go has no code for it): frames c-blosc 1.x, python-blosc,
// numcodecs read and write.  An extension next to CompressHIP / DecompressHIP, not a seam of the reference.
// ---------------------------------------------------------------------------------------------

// CompressCBlosc writes a C-Blosc-1 frame (LZ4 streams).  shuffle: NoShuffle, Shuffle or BitShuffle; len(data) < 2 GiB.
func CompressCBlosc(data []byte, shuffle Shuffle, typeSize int) ([]byte, error) {
	if !useHIP {
		return nil, fmt.Errorf("%w: no HIP device", ErrCompressionFailed)
	}
	buf := make([]byte, int(C.hb_cblosc_bound(C.size_t(len(data)), C.int(typeSize))))
	n := C.hb_cblosc_compress(ptr(data), C.size_t(len(data)), ptr(buf), C.size_t(len(buf)), C.int(shuffle), C.int(typeSize), C.int(Device))
	if n < 0 {
		return nil, hbError(n)
	}
	return buf[:n], nil
}

// DecompressCBlosc reads a C-Blosc-1 frame with LZ4 / LZ4HC streams (or a memcpyed one); other codec formats: ErrInvalidCodec.
func DecompressCBlosc(frame []byte) ([]byte, error) {
	if !useHIP {
		return nil, fmt.Errorf("%w: no HIP device", ErrDecompressionFailed)
	}
	var h C.hb_cblosc_header
	if rc := C.hb_cblosc_parse_header(ptr(frame), C.size_t(len(frame)), &h); rc != 0 {
		return nil, hbError(C.int64_t(rc))
	}
	buf := make([]byte, int(h.nbytes))
	n := C.hb_cblosc_decompress(ptr(frame), C.size_t(len(frame)), ptr(buf), C.size_t(len(buf)), C.int(Device))
	if n < 0 {
		return nil, hbError(n)
	}
	return buf[:n], nil
}

// ---------------------------------------------------------------------------------------------
// Batches of SMALL frames: many independent Compress / Decompress calls (blosc.go:257-303) through ONE set of kernel
// launches (hb_compress_frames_batch / hb_decompress_frames_batch).  The reference's own benchmark frame is 100 000 bytes
// (blosc_test.go:363-371): one such frame cannot fill a GPU, and below MinOffloadBytes CompressHIP leaves it to the CPU;
// a batch of them is a different matter (4096 x 100 000 B: ~400 GB/s device-resident, include/hipblosc.h).  Every frame is
// byte-identical to what CompressHIP writes for the same input; errs[k] carries the reference's sentinel for frame k.
// Go memory is borrowed for the call only: the pointer arrays live in C memory and hold pinned staging buffers
// (C pointers), never Go pointers (cgo rule).
// ---------------------------------------------------------------------------------------------
func CompressBatchHIP(datas [][]byte, opts Options, withIndex bool) ([][]byte, []error) {
	n := len(datas)
	out := make([][]byte, n)
	errs := make([]error, n)
	if n == 0 {
		return out, errs
	}
	if !useHIP || !(opts.Codec == LZ4 || opts.Codec == LZ4HC) {
		for k, d := range datas {
			out[k], errs[k] = CompressHIP(d, opts, withIndex)
		}
		return out, errs
	}
	srcs := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	dsts := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&srcs[0]))
	defer C.free(unsafe.Pointer(&dsts[0]))
	lens := make([]C.size_t, n)
	caps := make([]C.size_t, n)
	rcs := make([]C.int64_t, n)
	// one pinned slab for all inputs and one for all outputs: two allocations per batch, not two per frame
	var inBytes, outBytes C.size_t
	for k, d := range datas {
		lens[k] = C.size_t(len(d))
		caps[k] = C.hb_frame_bound(lens[k])
		inBytes += lens[k] // tightly packed: inputs that follow each other exactly go up in ONE copy (hb_compress_frames_batch)
		outBytes += (caps[k] + 63) &^ 63
	}
	slabIn, slabOut := C.hb_host_alloc(inBytes+64), C.hb_host_alloc(outBytes+64)
	if slabIn == nil || slabOut == nil {
		C.hb_host_free(slabIn)
		C.hb_host_free(slabOut)
		for k, d := range datas {
			out[k], errs[k] = CompressHIP(d, opts, withIndex)
		}
		return out, errs
	}
	defer C.hb_host_free(slabIn)
	defer C.hb_host_free(slabOut)
	var io, oo C.size_t
	for k, d := range datas {
		srcs[k] = unsafe.Add(slabIn, uintptr(io))
		dsts[k] = unsafe.Add(slabOut, uintptr(oo))
		copy(unsafe.Slice((*byte)(srcs[k]), len(d)), d)
		io += lens[k]
		oo += (caps[k] + 63) &^ 63
	}
	var o C.uint
	if withIndex {
		o |= C.HB_OPT_INDEX_TRAILER
	}
	if rc := C.hb_compress_frames_batch(C.int(n), &srcs[0], &lens[0], &dsts[0], &caps[0], &rcs[0],
		C.int(opts.Codec), C.int(opts.Level), C.int(opts.Shuffle), C.int(opts.TypeSize), o, C.int(Device)); rc != C.HB_OK {
		for k := range datas {
			errs[k] = hbError(C.int64_t(rc))
		}
		return out, errs
	}
	for k := range datas {
		if rcs[k] < 0 {
			errs[k] = hbError(rcs[k])
		} else {
			out[k] = append([]byte(nil), unsafe.Slice((*byte)(dsts[k]), int(rcs[k]))...)
		}
	}
	return out, errs
}

// DecompressBatchHIP: the inverse; frames of any writer (this library with or without the restart index, the pure-Go path).
func DecompressBatchHIP(frames [][]byte) ([][]byte, []error) {
	n := len(frames)
	out := make([][]byte, n)
	errs := make([]error, n)
	if n == 0 {
		return out, errs
	}
	if !useHIP {
		for k, f := range frames {
			out[k], errs[k] = Decompress(f)
		}
		return out, errs
	}
	srcs := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	dsts := (*[1 << 28]unsafe.Pointer)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))))[:n:n]
	defer C.free(unsafe.Pointer(&srcs[0]))
	defer C.free(unsafe.Pointer(&dsts[0]))
	lens := make([]C.size_t, n)
	caps := make([]C.size_t, n)
	rcs := make([]C.int64_t, n)
	var inBytes, outBytes C.size_t
	for k, f := range frames {
		lens[k] = C.size_t(len(f))
		caps[k] = 1
		if h, err := ParseHeader(f); err == nil {
			caps[k] = C.size_t(h.NBytesOrig) + 1 // (untrusted: a forged size only costs pinned memory, hb_host_alloc answers nil when there is none)
		}
		inBytes += lens[k] // frames tightly packed: one upload; results spaced by their capacity: one download (hb_decompress_frames_batch)
		outBytes += caps[k]
	}
	slabIn, slabOut := C.hb_host_alloc(inBytes+64), C.hb_host_alloc(outBytes+64)
	if slabIn == nil || slabOut == nil {
		C.hb_host_free(slabIn)
		C.hb_host_free(slabOut)
		for k, f := range frames {
			out[k], errs[k] = DecompressHIP(f, 0)
		}
		return out, errs
	}
	defer C.hb_host_free(slabIn)
	defer C.hb_host_free(slabOut)
	var io, oo C.size_t
	for k, f := range frames {
		srcs[k] = unsafe.Add(slabIn, uintptr(io))
		dsts[k] = unsafe.Add(slabOut, uintptr(oo))
		copy(unsafe.Slice((*byte)(srcs[k]), len(f)), f)
		io += lens[k]
		oo += caps[k]
	}
	if rc := C.hb_decompress_frames_batch(C.int(n), &srcs[0], &lens[0], &dsts[0], &caps[0], &rcs[0], 0, C.int(Device)); rc != C.HB_OK {
		for k := range frames {
			errs[k] = hbError(C.int64_t(rc))
		}
		return out, errs
	}
	for k := range frames {
		if rcs[k] < 0 {
			errs[k] = hbError(rcs[k])
		} else {
			out[k] = append([]byte(nil), unsafe.Slice((*byte)(dsts[k]), int(rcs[k]))...)
		}
	}
	return out, errs
}
