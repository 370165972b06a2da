"""ctypes front-end for the CPU oracle (oracle/blosc_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by anything under go-blosc_amd/.

Also holds an independent numpy twin of the four filters (second restatement of
shuffle.go:16-295, written against the formulas of SURVEY.md Appendix A rather than
against the C code) used to generate / re-derive the KAT fixtures in tests/golden/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libblosc_oracle.so")

ERR = {
    -1: "INVALID_DATA", -2: "INVALID_HEADER", -3: "INVALID_VERSION", -4: "INVALID_CODEC",
    -5: "SIZE_MISMATCH", -6: "DATA_TOO_LARGE", -7: "COMPRESSION_FAILED",
    -8: "DECOMPRESSION_FAILED", -12: "SHORT_BUFFER",
}
LZ4, LZ4HC, SNAPPY = 1, 2, 3
NOSHUFFLE, SHUFFLE, BITSHUFFLE = 0, 1, 2
OP_SHUFFLE, OP_UNSHUFFLE, OP_BITSHUFFLE, OP_BITUNSHUFFLE = 0, 1, 2, 3
POLICY_REFERENCE_MEMCPY = 1
D_F32, D_F64, D_I32, D_RAMP, D_RAND, D_BYTES256 = range(6)
_ELEM = {D_F32: 4, D_F64: 8, D_I32: 4, D_RAMP: 4, D_RAND: 4, D_BYTES256: 1}


def build(force=False):
    src = os.path.join(_HERE, "blosc_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libblosc_oracle.so"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB)
        u8p, sz, i64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int64
        L.ob_filter.argtypes = [ctypes.c_int, u8p, u8p, sz, ctypes.c_int]
        L.ob_filter.restype = None
        L.ob_lz4_bound.argtypes = [sz]; L.ob_lz4_bound.restype = sz
        L.ob_lz4_compress.argtypes = [u8p, sz, u8p, sz]; L.ob_lz4_compress.restype = i64
        L.ob_lz4_decompress.argtypes = [u8p, sz, u8p, sz]; L.ob_lz4_decompress.restype = i64
        L.ob_snappy_bound.argtypes = [sz]; L.ob_snappy_bound.restype = sz
        L.ob_snappy_compress.argtypes = [u8p, sz, u8p, sz]; L.ob_snappy_compress.restype = i64
        L.ob_snappy_decompress.argtypes = [u8p, sz, u8p, sz, ctypes.POINTER(ctypes.c_uint64)]; L.ob_snappy_decompress.restype = i64
        L.ob_frame_bound.argtypes = [sz]; L.ob_frame_bound.restype = sz
        L.ob_compress_frame.argtypes = [u8p, sz, u8p, sz] + [ctypes.c_int] * 4 + [ctypes.c_uint]
        L.ob_compress_frame.restype = i64
        L.ob_decompress_frame.argtypes = [u8p, sz, u8p, sz, ctypes.c_int]
        L.ob_decompress_frame.restype = i64
        L.ob_synth.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, u8p]
        L.ob_synth.restype = sz
        _lib = L
    return _lib


def _u8(a):
    a = np.ascontiguousarray(np.frombuffer(a, dtype=np.uint8) if isinstance(a, (bytes, bytearray, memoryview)) else a)
    return a.view(np.uint8).reshape(-1)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a.size else None


class OracleError(Exception):
    def __init__(self, code):
        super().__init__(f"oracle error {code} ({ERR.get(code, '?')})")
        self.code = code


def filter(op, src, typesize):
    s = _u8(src)
    d = np.empty_like(s)
    lib().ob_filter(op, _ptr(d), _ptr(s), s.size, typesize)
    return d


def lz4_bound(n):
    return lib().ob_lz4_bound(n)


def lz4_compress(src):
    s = _u8(src)
    d = np.empty(lz4_bound(s.size), dtype=np.uint8)
    c = lib().ob_lz4_compress(_ptr(s), s.size, _ptr(d), d.size)
    if c < 0:
        raise OracleError(c)
    return d[:c].copy()


def lz4_decompress(src, cap):
    """Returns the decoded bytes (possibly fewer than cap); raises OracleError on malformed input."""
    s = _u8(src)
    d = np.empty(max(cap, 1), dtype=np.uint8)
    r = lib().ob_lz4_decompress(_ptr(s), s.size, _ptr(d), cap)
    if r < 0:
        raise OracleError(r)
    return d[:r].copy()


def snappy_compress(src):
    s = _u8(src)
    d = np.empty(lib().ob_snappy_bound(s.size), dtype=np.uint8)
    c = lib().ob_snappy_compress(_ptr(s), s.size, _ptr(d), d.size)
    if c < 0:
        raise OracleError(c)
    return d[:c].copy()


def snappy_decompress(src, cap):
    """Returns the decoded bytes (the declared length); raises OracleError on malformed input or when the declared length
    exceeds cap (-12)."""
    s = _u8(src)
    d = np.empty(max(cap, 1), dtype=np.uint8)
    r = lib().ob_snappy_decompress(_ptr(s), s.size, _ptr(d), cap, None)
    if r < 0:
        raise OracleError(r)
    return d[:r].copy()


def compress_frame(src, codec=LZ4, level=5, shuffle=SHUFFLE, typesize=4, policy=0):
    s = _u8(src)
    d = np.empty(lib().ob_frame_bound(s.size), dtype=np.uint8)
    c = lib().ob_compress_frame(_ptr(s), s.size, _ptr(d), d.size, codec, level, shuffle, typesize, policy)
    if c < 0:
        raise OracleError(c)
    return d[:c].copy()


def decompress_frame(frame, typesize_override=0, cap=None):
    f = _u8(frame)
    if cap is None:
        cap = int.from_bytes(f[4:8].tobytes(), "little") if f.size >= 16 else 0
    d = np.empty(max(cap, 1), dtype=np.uint8)
    r = lib().ob_decompress_frame(_ptr(f), f.size, _ptr(d), cap, typesize_override)
    if r < 0:
        raise OracleError(r)
    return d[:r].copy()


def synth(kind, count, frame=0, first=0):
    """SURVEY.md §8(d) workloads; returns a uint8 view of `count` elements."""
    out = np.empty(count * _ELEM[kind], dtype=np.uint8)
    lib().ob_synth(kind, frame, first, count, _ptr(out))
    return out


# ---------------------------------------------------------------------------
# numpy twin of the filters (independent of the C code; small inputs only)
# ---------------------------------------------------------------------------

def np_shuffle(b, ts):
    b = _u8(b)
    n = b.size
    if ts <= 1 or n < ts:
        return b.copy()
    ne = n // ts
    out = b.copy()
    out[: ne * ts] = b[: ne * ts].reshape(ne, ts).T.reshape(-1)   # dst[j*ne+i] = src[i*ts+j]
    return out


def np_unshuffle(b, ts):
    b = _u8(b)
    n = b.size
    if ts <= 1 or n < ts:
        return b.copy()
    ne = n // ts
    out = b.copy()
    out[: ne * ts] = b[: ne * ts].reshape(ts, ne).T.reshape(-1)   # dst[i*ts+j] = src[j*ne+i]
    return out


def np_bitshuffle(b, ts):
    b = _u8(b)
    n = b.size
    if ts <= 1 or n < ts:
        return b.copy()
    g = (n // ts) // 8
    out = b.copy()
    if g:
        a = b[: g * 8 * ts].reshape(g, 8, ts).transpose(0, 2, 1)          # [g][b][e]
        bits = np.unpackbits(a[..., None], axis=3)                        # [g][b][e][k], k=0 is MSB
        out[: g * 8 * ts] = np.packbits(bits.transpose(0, 1, 3, 2), axis=3).reshape(-1)  # [g][b][k] <- bits over e
    return out


def np_bitunshuffle(b, ts):
    b = _u8(b)
    n = b.size
    if ts <= 1 or n < ts:
        return b.copy()
    g = (n // ts) // 8
    out = b.copy()
    if g:
        t = b[: g * 8 * ts].reshape(g, ts, 8)                             # [g][b][i]
        bits = np.unpackbits(t[..., None], axis=3)                        # [g][b][i][e]
        e = np.packbits(bits.transpose(0, 1, 3, 2), axis=3).reshape(g, ts, 8)  # [g][b][e] <- bits over i
        out[: g * 8 * ts] = e.transpose(0, 2, 1).reshape(-1)              # dst[g*8ts + e*ts + b]
    return out


NP_FILTERS = {OP_SHUFFLE: np_shuffle, OP_UNSHUFFLE: np_unshuffle,
              OP_BITSHUFFLE: np_bitshuffle, OP_BITUNSHUFFLE: np_bitunshuffle}
