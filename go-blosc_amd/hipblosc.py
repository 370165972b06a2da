"""Host-side mirror of go-blosc's public API over the hipblosc C ABI (include/hipblosc.h).

Same names, argument meaning and error behaviour as the reference package (blosc.go / codec.go /
shuffle.go), so the parity tests read like the reference's own tests.  Every O(n) operation runs on
the MI355X through libhipblosc.so; there is no CPU fallback — importing works without the library
(so `-m "not gpu"` tests can check symbols), calling anything without it / without a GPU raises.

ctypes only: no torch types cross the boundary (device pointers are plain integers).
"""
import ctypes
import os
from dataclasses import dataclass

_HERE = os.path.dirname(os.path.abspath(__file__))
# HIPBLOSC_LIB: another build of the same library (tools/lab/ab.py times kernel variants side by side); never a different implementation
LIB_PATH = os.environ.get("HIPBLOSC_LIB") or os.path.join(_HERE, "lib", "libhipblosc.so")

# ---- constants, blosc.go:49-52, :57-64, :89-93, :110-121 ----
Version = "1.0.0"
FormatVersion = 2
HeaderSize = MinHeaderSize = 16
BloscLZ, LZ4, LZ4HC, Snappy, ZLIB, ZSTD = range(6)
NoShuffle, Shuffle1, BitShuffle = 0, 1, 2


def indexless_parallel(payload, nbytes):
    """csrc/hb_lz4.h hb_indexless_parallel(): an LZ4 frame without a restart index takes the token discovery (parallel) instead of the
    single wavefront when its payload is 256 KiB or more, or 16 KiB or more and it decodes to 2 MiB or more."""
    return payload >= (256 << 10) or (payload >= (16 << 10) and nbytes >= (2 << 20))

flagShuffle, flagMemcpy, flagBitShuffle, flagSplit = 0x1, 0x2, 0x4, 0x8
OP_SHUFFLE, OP_UNSHUFFLE, OP_BITSHUFFLE, OP_BITUNSHUFFLE = 0, 1, 2, 3
OPT_INDEX_TRAILER, OPT_REFERENCE_MEMCPY, OPT_NO_FUSION = 0x1, 0x2, 0x4

_CODEC_NAMES = {BloscLZ: "blosclz", LZ4: "lz4", LZ4HC: "lz4hc", Snappy: "snappy", ZLIB: "zlib", ZSTD: "zstd"}
_SHUFFLE_NAMES = {NoShuffle: "noshuffle", Shuffle1: "shuffle", BitShuffle: "bitshuffle"}


def codec_string(c):      # Codec.String, blosc.go:67-84
    return _CODEC_NAMES.get(c, f"unknown({c})")


def shuffle_string(s):    # Shuffle.String, blosc.go:96-107
    return _SHUFFLE_NAMES.get(s, f"unknown({s})")


# ---- sentinel errors, blosc.go:125-149 ----
class BloscError(Exception):
    code = 0


class ErrInvalidData(BloscError):
    code = -1


class ErrInvalidHeader(BloscError):
    code = -2


class ErrInvalidVersion(BloscError):
    code = -3


class ErrInvalidCodec(BloscError):
    code = -4


class ErrSizeMismatch(BloscError):
    code = -5


class ErrDataTooLarge(BloscError):
    code = -6


class ErrCompressionFailed(BloscError):
    code = -7


class ErrDecompressionFailed(BloscError):
    code = -8


class HipBloscError(BloscError):
    """C-side failures that have no Go sentinel (no device, HIP error, bad argument, short buffer)."""


_BY_CODE = {c.code: c for c in (ErrInvalidData, ErrInvalidHeader, ErrInvalidVersion, ErrInvalidCodec,
                                ErrSizeMismatch, ErrDataTooLarge, ErrCompressionFailed, ErrDecompressionFailed)}

EXPORTS = [
    "hb_init", "hb_device_count", "hb_shutdown", "hb_pool_limit", "hb_pool_cached_bytes", "hb_strerror", "hb_version", "hb_host_alloc", "hb_host_free",
    "hb_filter", "hb_filter_dev", "hb_lz4_bound", "hb_lz4_compress", "hb_lz4_decompress",
    "hb_lz4_compress_workspace", "hb_lz4_decompress_workspace", "hb_lz4_compress_dev", "hb_lz4_decompress_dev",
    "hb_index_bound", "hb_codec_bound", "hb_codec_compress", "hb_codec_decompress", "hb_parse_header", "hb_header_bytes", "hb_frame_bound", "hb_compress_frame",
    "hb_decompress_frame", "hb_compress_frame_workspace", "hb_decompress_frame_workspace", "hb_decompress_frame_workspace_foreign", "hb_lz4_decompress_workspace_foreign",
    "hb_compress_frame_dev", "hb_decompress_frame_dev", "hb_compress_frames_multi", "hb_decompress_frames_multi",
    "hb_profile_enable", "hb_profile_count", "hb_profile_get", "hb_last_result_flags",
    "hb_decompress_frame_dev_hdr", "hb_cblosc_parse_header", "hb_cblosc_decompress", "hb_cblosc_compress", "hb_cblosc_bound", "hb_cblosc_compress_workspace", "hb_cblosc_compress_dev", "hb_cblosc_decompress_workspace", "hb_cblosc_decompress_dev",
    "hb_compress_frames_batch_workspace", "hb_compress_frames_batch_dev", "hb_frames_batch_headers_dev",
    "hb_decompress_frames_batch_workspace", "hb_decompress_frames_batch_dev", "hb_compress_frames_batch", "hb_decompress_frames_batch",
    "hb_queue_create", "hb_queue_create_ex", "hb_queue_destroy", "hb_queue_compress", "hb_queue_decompress", "hb_queue_wait",
]


class hb_header(ctypes.Structure):
    _fields_ = [("version", ctypes.c_uint8), ("codec", ctypes.c_uint8), ("flags", ctypes.c_uint8),
                ("typesize", ctypes.c_uint8), ("nbytes", ctypes.c_uint32), ("blocksize", ctypes.c_uint32),
                ("cbytes", ctypes.c_uint32)]


class hb_result(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("flags", ctypes.c_uint32), ("bytes", ctypes.c_uint64),
                ("total_bytes", ctypes.c_uint64), ("reserved", ctypes.c_uint64)]


_lib = None


def lib():
    """The loaded C ABI.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipBloscError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        vp, sz, i32, i64, u32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int64, ctypes.c_uint
        sig = {
            "hb_init": (i32, []), "hb_device_count": (i32, []), "hb_shutdown": (None, []),
            "hb_pool_limit": (None, [sz]), "hb_pool_cached_bytes": (sz, []),
            "hb_strerror": (ctypes.c_char_p, [i32]), "hb_version": (ctypes.c_char_p, []),
            "hb_host_alloc": (vp, [sz]), "hb_host_free": (None, [vp]),
            "hb_filter": (i32, [i32, vp, vp, sz, i32, i32]),
            "hb_filter_dev": (i32, [i32, vp, vp, sz, i32, vp]),
            "hb_lz4_bound": (sz, [sz]), "hb_index_bound": (sz, [sz]), "hb_codec_bound": (sz, [i32, sz]),
            "hb_codec_compress": (i64, [i32, i32, vp, sz, vp, sz, i32]), "hb_codec_decompress": (i64, [i32, vp, sz, vp, sz, i32]),
            "hb_lz4_compress": (i64, [vp, sz, vp, sz, i32]),
            "hb_lz4_decompress": (i64, [vp, sz, vp, sz, i32]),
            "hb_lz4_compress_workspace": (sz, [sz]), "hb_lz4_decompress_workspace": (sz, [sz]),
            "hb_lz4_compress_dev": (i32, [vp, sz, vp, sz, vp, sz, vp, sz, vp, vp]),
            "hb_lz4_decompress_dev": (i32, [vp, sz, vp, sz, vp, sz, vp, sz, vp, vp]),
            "hb_parse_header": (i32, [vp, sz, ctypes.POINTER(hb_header)]),
            "hb_header_bytes": (None, [ctypes.POINTER(hb_header), vp]),
            "hb_frame_bound": (sz, [sz]),
            "hb_compress_frame": (i64, [vp, sz, vp, sz, i32, i32, i32, i32, u32, i32]),
            "hb_decompress_frame": (i64, [vp, sz, vp, sz, i32, i32]),
            "hb_compress_frame_workspace": (sz, [sz]), "hb_decompress_frame_workspace": (sz, [sz]), "hb_decompress_frame_workspace_foreign": (sz, [sz]),
            "hb_lz4_decompress_workspace_foreign": (sz, [sz]),
            "hb_compress_frame_dev": (i32, [vp, sz, vp, sz, i32, i32, i32, i32, u32, vp, sz, vp, vp]),
            "hb_decompress_frame_dev": (i32, [vp, sz, vp, sz, i32, vp, sz, vp, vp]),
            "hb_compress_frames_multi": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, u32]),
            "hb_decompress_frames_multi": (i32, [i32, vp, vp, vp, vp, vp, i32]),
            "hb_last_result_flags": (u32, []),
            "hb_profile_enable": (i32, [i32]), "hb_profile_count": (i32, []),
            "hb_profile_get": (ctypes.c_char_p, [i32, ctypes.POINTER(ctypes.c_float)]),
            "hb_decompress_frame_dev_hdr": (i32, [ctypes.POINTER(hb_header), vp, sz, vp, sz, i32, vp, sz, vp, vp]),
            "hb_cblosc_parse_header": (i32, [vp, sz, vp]), "hb_cblosc_compress": (i64, [vp, sz, vp, sz, i32, i32, i32]),
            "hb_cblosc_bound": (sz, [sz, i32]), "hb_cblosc_compress_workspace": (sz, [sz, i32, i32]),
            "hb_cblosc_compress_dev": (i32, [vp, sz, vp, sz, i32, i32, vp, sz, vp, vp]), "hb_cblosc_decompress": (i64, [vp, sz, vp, sz, i32]),
            "hb_cblosc_decompress_workspace": (sz, [sz, sz, sz]), "hb_cblosc_decompress_dev": (i32, [vp, vp, sz, vp, sz, vp, sz, vp, vp]),
            "hb_compress_frames_batch_workspace": (sz, [i32, vp, i32]),
            "hb_compress_frames_batch_dev": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, u32, vp, sz, vp, vp]),
            "hb_frames_batch_headers_dev": (i32, [i32, vp, vp, vp, vp, vp, sz, vp]),
            "hb_decompress_frames_batch_workspace": (sz, [i32, vp]),
            "hb_decompress_frames_batch_dev": (i32, [i32, vp, vp, vp, vp, vp, i32, vp, sz, vp, vp]),
            "hb_compress_frames_batch": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, u32, i32]),
            "hb_decompress_frames_batch": (i32, [i32, vp, vp, vp, vp, vp, i32, i32]),
            "hb_queue_create": (vp, [i32, i32, sz]), "hb_queue_create_ex": (vp, [i32, i32, sz, ctypes.c_uint]), "hb_queue_destroy": (None, [vp]),
            "hb_queue_compress": (i64, [vp, vp, sz, vp, sz, i32, i32, i32, i32, u32]),
            "hb_queue_decompress": (i64, [vp, vp, sz, vp, sz, i32]),
            "hb_queue_wait": (i64, [vp, i64]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _raise(code):
    msg = lib().hb_strerror(int(code)).decode()
    raise _BY_CODE.get(int(code), HipBloscError)(f"{msg} (code {code})")


def _check(rc):
    if rc < 0:
        _raise(rc)
    return rc


def _buf(b):
    """bytes-like -> (ctypes pointer, length, keepalive)"""
    if isinstance(b, (bytes, bytearray)):
        arr = (ctypes.c_char * len(b)).from_buffer_copy(b) if isinstance(b, bytes) else (ctypes.c_char * len(b)).from_buffer(b)
        return ctypes.cast(arr, ctypes.c_void_p), len(b), arr
    mv = memoryview(b).cast("B")
    if mv.readonly:
        arr = (ctypes.c_char * len(mv)).from_buffer_copy(mv)
    else:
        arr = (ctypes.c_char * len(mv)).from_buffer(mv)
    return ctypes.cast(arr, ctypes.c_void_p), len(mv), arr


# ---------------------------------------------------------------------------------------------
# Header / Options, blosc.go:154-245
# ---------------------------------------------------------------------------------------------
@dataclass
class Header:
    Version: int = 0
    VersionLZ: int = 0
    Flags: int = 0
    TypeSize: int = 0
    NBytesOrig: int = 0
    BlockSize: int = 0
    NBytesComp: int = 0

    def Bytes(self):                       # blosc.go:188-198
        h = hb_header(self.Version, self.VersionLZ, self.Flags, self.TypeSize, self.NBytesOrig, self.BlockSize, self.NBytesComp)
        out = ctypes.create_string_buffer(16)
        lib().hb_header_bytes(ctypes.byref(h), ctypes.cast(out, ctypes.c_void_p))
        return out.raw

    def HasShuffle(self):                  # blosc.go:201-203
        return self.Flags & flagShuffle != 0

    def HasBitShuffle(self):               # blosc.go:206-208
        return self.Flags & flagBitShuffle != 0

    def IsMemcpy(self):                    # blosc.go:211-213
        return self.Flags & flagMemcpy != 0

    def ShuffleMode(self):                 # blosc.go:216-224 (bitshuffle wins)
        if self.HasBitShuffle():
            return BitShuffle
        if self.HasShuffle():
            return Shuffle1
        return NoShuffle


def ParseHeader(data):                     # blosc.go:165-185
    p, n, keep = _buf(data)
    h = hb_header()
    _check(lib().hb_parse_header(p, n, ctypes.byref(h)))
    return Header(h.version, h.codec, h.flags, h.typesize, h.nbytes, h.blocksize, h.cbytes)


@dataclass
class Options:                             # blosc.go:227-234
    Codec: int = LZ4
    Level: int = 0
    Shuffle: int = NoShuffle
    TypeSize: int = 0
    BlockSize: int = 0                     # accepted and ignored, as in the reference (blosc.go:232)
    NumThreads: int = 0                    # accepted and ignored (blosc.go:233)


def DefaultOptions():                      # blosc.go:237-245
    return Options(Codec=LZ4, Level=5, Shuffle=Shuffle1, TypeSize=4, BlockSize=0)


# extra knobs that have no counterpart in the reference
device = 0                                 # HIP device used by the host-pointer entry points
default_opts = 0                           # OPT_* bits ORed into every Compress call


def Compress(data, codec, level, shuffle, typeSize, opts=None):     # blosc.go:257-265
    return CompressWithOptions(data, Options(Codec=codec, Level=level, Shuffle=shuffle, TypeSize=typeSize), opts)


def CompressWithOptions(data, o, opts=None):                        # blosc.go:268-286 + compressBackend :320-374
    p, n, keep = _buf(data)
    if n == 0:
        raise ErrInvalidData("blosc: invalid compressed data")      # bare sentinel, blosc.go:269-271
    L = lib()
    cap = L.hb_frame_bound(n)
    out = ctypes.create_string_buffer(cap)
    rc = L.hb_compress_frame(p, n, ctypes.cast(out, ctypes.c_void_p), cap, o.Codec, o.Level, o.Shuffle, o.TypeSize,
                             default_opts if opts is None else opts, device)
    _check(rc)
    return out.raw[:rc]


def Decompress(data):                                               # blosc.go:291-293
    return DecompressWithSize(data, 0)


def DecompressWithSize(data, typeSize):                             # blosc.go:296-303 + decompressBackend :377-434
    p, n, keep = _buf(data)
    if n < HeaderSize:
        raise ErrInvalidHeader("blosc: invalid header")             # bare sentinel, blosc.go:297-299
    h = ParseHeader(data)
    L = lib()
    out = ctypes.create_string_buffer(max(h.NBytesOrig, 1))
    rc = L.hb_decompress_frame(p, n, ctypes.cast(out, ctypes.c_void_p), h.NBytesOrig, typeSize, device)
    _check(rc)
    return out.raw[:rc]


def GetInfo(data):                                                  # blosc.go:306-308
    return ParseHeader(data)


def GetDecompressedSize(data):                                      # blosc.go:311-317
    return ParseHeader(data).NBytesOrig


# ---------------------------------------------------------------------------------------------
# filters, shuffle.go
# ---------------------------------------------------------------------------------------------
def _filter(op, src, typeSize):
    p, n, keep = _buf(src)
    out = ctypes.create_string_buffer(max(n, 1))
    _check(lib().hb_filter(op, ctypes.cast(out, ctypes.c_void_p), p, n, typeSize, device))
    return out.raw[:n]


def shuffleBytes(src, typeSize):           # shuffle.go:16-73
    return _filter(OP_SHUFFLE, src, typeSize)


def unshuffleBytes(src, typeSize):         # shuffle.go:76-133
    return _filter(OP_UNSHUFFLE, src, typeSize)


def bitShuffle(src, typeSize):             # shuffle.go:145-219
    return _filter(OP_BITSHUFFLE, src, typeSize)


def bitUnshuffle(src, typeSize):           # shuffle.go:222-295
    return _filter(OP_BITUNSHUFFLE, src, typeSize)


def ShuffleBuffer(data, typeSize, mode):   # shuffle.go:298-309: in place on a bytearray; unknown mode = no-op
    if mode == Shuffle1:
        data[:] = shuffleBytes(bytes(data), typeSize)
    elif mode == BitShuffle:
        data[:] = bitShuffle(bytes(data), typeSize)


def UnshuffleBuffer(data, typeSize, mode):  # shuffle.go:312-323
    if mode == Shuffle1:
        data[:] = unshuffleBytes(bytes(data), typeSize)
    elif mode == BitShuffle:
        data[:] = bitUnshuffle(bytes(data), typeSize)


# ---------------------------------------------------------------------------------------------
# codec plugin seam, codec.go:15-53 — the device LZ4 codec is what RegisterCodec(LZ4, ...) installs
# ---------------------------------------------------------------------------------------------
class HipLZ4Codec:
    """CodecInterface (codec.go:15-24) backed by hb_lz4_compress / hb_lz4_decompress."""

    def Name(self):                        # codec.go:61
        return "lz4"

    def Compress(self, data, level):       # codec.go:63-75 (level ignored)
        p, n, keep = _buf(data)
        L = lib()
        cap = L.hb_lz4_bound(n)
        out = ctypes.create_string_buffer(cap)
        rc = _check(L.hb_lz4_compress(p, n, ctypes.cast(out, ctypes.c_void_p), cap, device))
        return out.raw[:rc]

    def Decompress(self, data, expectedSize):   # codec.go:77-84: returns buf[:n], n may be < expectedSize
        p, n, keep = _buf(data)
        out = ctypes.create_string_buffer(max(expectedSize, 1))
        rc = _check(lib().hb_lz4_decompress(p, n, ctypes.cast(out, ctypes.c_void_p), expectedSize, device))
        return out.raw[:rc]


class HipDeviceCodec:
    """CodecInterface (codec.go:15-24) for LZ4HC (codec.go:90-128) and Snappy (codec.go:228-244) backed by hb_codec_*."""

    def __init__(self, codec, name):
        self.codec, self.name = codec, name

    def Name(self):
        return self.name

    def Compress(self, data, level):
        p, n, keep = _buf(data)
        L = lib()
        cap = L.hb_codec_bound(self.codec, n)
        out = ctypes.create_string_buffer(cap)
        rc = _check(L.hb_codec_compress(self.codec, level, p, n, ctypes.cast(out, ctypes.c_void_p), cap, device))
        return out.raw[:rc]

    def Decompress(self, data, expectedSize):
        p, n, keep = _buf(data)
        out = ctypes.create_string_buffer(max(expectedSize, 1))
        rc = _check(lib().hb_codec_decompress(self.codec, p, n, ctypes.cast(out, ctypes.c_void_p), expectedSize, device))
        return out.raw[:rc]


# codec.go:27-33: the codecs that live on the device path (ZLIB is not built; ZSTD is a host codec behind hb_compress_frame)
codecs = {LZ4: HipLZ4Codec(), LZ4HC: HipDeviceCodec(LZ4HC, "lz4hc"), Snappy: HipDeviceCodec(Snappy, "snappy")}


def RegisterCodec(id, codec):              # codec.go:36-38
    codecs[id] = codec


def GetCodec(id):                          # codec.go:41-44
    c = codecs.get(id)
    return c, c is not None


def ListCodecs():                          # codec.go:47-53
    return list(codecs.keys())


# ---------------------------------------------------------------------------------------------
# pipelined host API (hb_queue_*): frames in flight, uploads / kernels / downloads overlapped
# ---------------------------------------------------------------------------------------------
class PinnedBuffer:
    """hb_host_alloc() memory as a writable buffer (numpy: np.frombuffer(buf.view, np.uint8))."""

    def __init__(self, nbytes):
        self.nbytes = int(nbytes)
        self.ptr = lib().hb_host_alloc(self.nbytes)
        if not self.ptr:
            raise HipBloscError("hb_host_alloc failed (no HIP device?)")
        self.view = (ctypes.c_char * self.nbytes).from_address(self.ptr)

    def close(self):
        if self.ptr:
            self.view = None
            lib().hb_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.close()


class FrameQueue:
    """`depth` frames in flight on one device.  compress()/decompress() take raw addresses (PinnedBuffer.ptr or any host
    address) and return a ticket; wait(ticket) returns the byte count or raises the reference's sentinel error."""

    def __init__(self, max_nbytes, depth=3, dev=None, foreign_frames=False):
        # foreign_frames: slots get the larger decode workspace, with which frames of other writers decode in parallel too
        self.q = lib().hb_queue_create_ex(device if dev is None else dev, depth, max_nbytes, 1 if foreign_frames else 0)
        if not self.q:
            raise HipBloscError("hb_queue_create failed")

    def compress(self, src_ptr, n, dst_ptr, cap, codec=LZ4, level=5, shuffle=Shuffle1, typesize=4, opts=0):
        return _check(lib().hb_queue_compress(self.q, src_ptr, n, dst_ptr, cap, codec, level, shuffle, typesize, opts))

    def decompress(self, frame_ptr, n, dst_ptr, cap, typesize=0):
        return _check(lib().hb_queue_decompress(self.q, frame_ptr, n, dst_ptr, cap, typesize))

    def wait(self, ticket):
        return _check(lib().hb_queue_wait(self.q, ticket))

    def close(self):
        if self.q:
            lib().hb_queue_destroy(self.q)
            self.q = None

    def __del__(self):
        self.close()


# ---- frames in the C-Blosc-1 wire format (decode only; LZ4 / LZ4HC streams and memcpyed frames) ----
class CBloscHeader(ctypes.Structure):
    _fields_ = [("version", ctypes.c_uint8), ("versionlz", ctypes.c_uint8), ("flags", ctypes.c_uint8), ("typesize", ctypes.c_uint8),
                ("nbytes", ctypes.c_uint32), ("blocksize", ctypes.c_uint32), ("cbytes", ctypes.c_uint32), ("codec_format", ctypes.c_uint32)]


def CBloscParseHeader(frame):
    p, n, keep = _buf(frame)
    h = CBloscHeader()
    _check(lib().hb_cblosc_parse_header(p, n, ctypes.byref(h)))
    return h


def CBloscDecompress(frame):
    """What blosc_decompress() of c-blosc 1.x returns for `frame` (include/hipblosc.h hb_cblosc_decompress)."""
    p, n, keep = _buf(frame)
    h = CBloscParseHeader(frame)
    out = ctypes.create_string_buffer(max(h.nbytes, 1))
    rc = _check(lib().hb_cblosc_decompress(p, n, ctypes.cast(out, ctypes.c_void_p), h.nbytes, device))
    return out.raw[:rc]


def CBloscCompress(data, shuffle=1, typesize=4):
    """A frame blosc_decompress() of c-blosc 1.x reads (include/hipblosc.h hb_cblosc_compress); shuffle 0 / 1 / 2 = none / byte / bit."""
    p, n, keep = _buf(data)
    cap = lib().hb_cblosc_bound(n, typesize)
    out = ctypes.create_string_buffer(cap)
    rc = _check(lib().hb_cblosc_compress(p, n, ctypes.cast(out, ctypes.c_void_p), cap, shuffle, typesize, device))
    return out.raw[:rc]


# ---------------------------------------------------------------------------------------------
# batches of small frames in one set of launches (hb_compress_frames_batch / hb_decompress_frames_batch): the i-th result is what
# Compress / Decompress would have returned for the i-th input -- a frame, or the reference's sentinel error (returned, not raised)
# ---------------------------------------------------------------------------------------------
def CompressBatch(datas, codec=LZ4, level=5, shuffle=Shuffle1, typesize=4, opts=0, dev=None):
    n = len(datas)
    if n == 0:
        return []
    L = lib()
    keep = [_buf(d) for d in datas]
    caps = [L.hb_frame_bound(k[1]) for k in keep]
    outs = [(ctypes.c_char * c)() for c in caps]
    vp, sz, i64 = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_int64 * n
    srcs = vp(*[k[0].value for k in keep])
    dsts = vp(*[ctypes.addressof(o) for o in outs])
    rcs = i64()
    _check(L.hb_compress_frames_batch(n, srcs, sz(*[k[1] for k in keep]), dsts, sz(*caps), rcs, int(codec), int(level), int(shuffle), int(typesize),
                                      int(opts), device if dev is None else dev))
    res = []
    for i in range(n):
        res.append(bytes(outs[i][: rcs[i]]) if rcs[i] >= 0 else _BY_CODE.get(int(rcs[i]), HipBloscError)(f"code {rcs[i]}"))
    return res


def DecompressBatch(frames, typesize=0, dev=None):
    n = len(frames)
    if n == 0:
        return []
    L = lib()
    keep = [_buf(f) for f in frames]
    caps = []
    for f in frames:
        try:
            caps.append(max(ParseHeader(bytes(f[:16])).NBytesOrig, 1))
        except BloscError:
            caps.append(1)
    outs = [(ctypes.c_char * c)() for c in caps]
    vp, sz, i64 = ctypes.c_void_p * n, ctypes.c_size_t * n, ctypes.c_int64 * n
    srcs = vp(*[k[0].value for k in keep])
    dsts = vp(*[ctypes.addressof(o) for o in outs])
    rcs = i64()
    _check(L.hb_decompress_frames_batch(n, srcs, sz(*[k[1] for k in keep]), dsts, sz(*caps), rcs, int(typesize), device if dev is None else dev))
    res = []
    for i in range(n):
        res.append(bytes(outs[i][: rcs[i]]) if rcs[i] >= 0 else _BY_CODE.get(int(rcs[i]), HipBloscError)(f"code {rcs[i]}"))
    return res
