"""GPU parity for SURVEY §8 row f4: frames in the C-Blosc-1 wire format decoded on the device.

The writer is c-blosc 1.21.0 itself (/opt/conda/lib/libblosc.so.1, the format's own library, via ctypes: the go-blosc reference
does not implement this format -- SURVEY §0.2 -- so there is no reference code to restate; the checker is the library's own
blosc_decompress_ctx and, simpler, the input the frame was made of).  Covered: lz4 / lz4hc, byte shuffle / bit shuffle / none,
typesizes that split (<= 16) and that do not, automatic and explicit block sizes, a last block that is shorter (never split),
element counts that are no multiple of 8 (bit shuffle skipped by the writer), stored streams, memcpyed frames, tiny inputs, and what
must be refused.
"""
import ctypes
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_LIB = "/opt/conda/lib/libblosc.so.1"


@pytest.fixture(scope="module")
def cb():
    if not os.path.exists(_LIB):
        pytest.skip("c-blosc 1.x is not in this image")
    L = ctypes.CDLL(_LIB)
    L.blosc_compress_ctx.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    L.blosc_decompress_ctx.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    class CB:
        def compress(self, x, clevel=5, shuffle=1, typesize=4, cname=b"lz4", blocksize=0):
            x = np.ascontiguousarray(x).view(np.uint8).reshape(-1)
            dst = np.empty(x.size + 16 + 4 * (x.size // 32 + 1024), np.uint8)
            c = L.blosc_compress_ctx(clevel, shuffle, typesize, x.size, x.ctypes.data, dst.ctypes.data, dst.size, cname, blocksize, 1)
            assert c > 0, c
            return dst[:c].tobytes()

        def decompress(self, frame, n):
            out = np.empty(max(n, 1), np.uint8)
            src = np.frombuffer(frame, np.uint8)
            r = L.blosc_decompress_ctx(src.ctypes.data, out.ctypes.data, out.size, 1)
            return r, out[:max(r, 0)].tobytes()

    return CB()


def _sets(O):
    rng = np.random.default_rng(21)
    return {
        "f32": O.synth(O.D_F32, (3 << 20) // 4 + 5), "f64": O.synth(O.D_F64, (2 << 20) // 8 + 1), "i32": O.synth(O.D_I32, 1 << 18),
        "ramp": O.synth(O.D_RAMP, 300000), "random": rng.integers(0, 256, (1 << 20) + 13, dtype=np.uint8),
        "zeros": np.zeros((2 << 20) + 7, np.uint8), "few_valued": (rng.integers(0, 4, 1 << 20, dtype=np.uint8) * 64),
        "text": np.frombuffer(b"".join(bytes(str(i * 7919 % 100003), "ascii") + b", " for i in range(150000)), np.uint8),
    }


def test_frames_of_the_library_decode_exactly(hb, O, cb):
    n_frames = 0
    for name, x in _sets(O).items():
        xb = x.tobytes()
        for ts in (1, 2, 3, 4, 8, 16, 17):
            for shuffle in (0, 1, 2):
                for cname, clevel, bs in ((b"lz4", 5, 0), (b"lz4", 1, 0), (b"lz4hc", 9, 0), (b"lz4", 5, 4096), (b"lz4", 9, 65536 + 8 * ts)):
                    if (ts in (3, 17) or name in ("text", "few_valued")) and (clevel, bs) not in ((5, 0), (5, 4096)):
                        continue                                          # (keep the sweep in seconds)
                    f = cb.compress(x, clevel, shuffle, ts, cname, bs)
                    assert hb.CBloscDecompress(f) == xb, (name, ts, shuffle, cname, clevel, bs)
                    n_frames += 1
    assert n_frames > 400


def test_header_fields_and_odd_sizes(hb, O, cb):
    rng = np.random.default_rng(3)
    x = O.synth(O.D_F32, 100000)
    f = cb.compress(x, 5, 1, 4, b"lz4", 32768)
    h = hb.CBloscParseHeader(f)
    ver, verlz, flags, ts, nbytes, blocksize, cbytes = struct.unpack("<BBBBIII", f[:16])
    assert (h.version, h.versionlz, h.flags, h.typesize, h.nbytes, h.blocksize, h.cbytes, h.codec_format) == (2, verlz, flags, 4, x.size, blocksize, len(f), 1)
    assert blocksize % 32768 == 0                                         # (the library scales a requested block size by the typesize when it splits)
    assert h.flags & 1 and not h.flags & 2
    # sizes around the corners: below the 128-byte minimum buffer, one byte, no whole element, leftovers of every kind
    for n in (1, 2, 3, 5, 15, 16, 17, 127, 128, 129, 4095, 4096, 4097, 32768 * 3 + 1, 32768 * 3 + 29):
        y = rng.integers(0, 7, n, dtype=np.uint8)
        for ts in (1, 4, 8):
            for shuffle in (0, 1, 2):
                f = cb.compress(y, 5, shuffle, ts, b"lz4", 32768 if n > 32768 else 0)
                assert hb.CBloscDecompress(f) == y.tobytes(), (n, ts, shuffle, hb.CBloscParseHeader(f).flags)
    # clevel 0: a memcpyed frame
    f = cb.compress(x, 0, 1, 4)
    assert hb.CBloscParseHeader(f).flags & 2
    assert hb.CBloscDecompress(f) == x.tobytes()
    # incompressible data: stored streams inside a frame that is not memcpyed, or a memcpyed frame -- whatever the library chose
    r = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    for shuffle in (0, 1, 2):
        assert hb.CBloscDecompress(cb.compress(r, 5, shuffle, 4)) == r.tobytes()


def test_what_must_be_refused(hb, O, cb):
    x = O.synth(O.D_F32, 200000)
    f = cb.compress(x, 5, 1, 4)
    with pytest.raises(hb.ErrInvalidCodec):
        hb.CBloscDecompress(cb.compress(x, 5, 1, 4, b"blosclz"))          # codec format 0: not built
    with pytest.raises(hb.BloscError):
        hb.CBloscDecompress(f[:10])                                       # not even a header
    with pytest.raises(hb.BloscError):
        hb.CBloscDecompress(f[:len(f) // 2])                              # cbytes says more than there is
    g = bytearray(f); g[0] = 3
    with pytest.raises(hb.ErrInvalidVersion):
        hb.CBloscDecompress(bytes(g))
    # damage inside the streams: the library answers with a negative number, here ErrDecompressionFailed -- or both decode to the
    # same bytes when the damage happens to leave a valid stream
    rng = np.random.default_rng(8)
    hdr = hb.CBloscParseHeader(f)
    nblocks = (hdr.nbytes + hdr.blocksize - 1) // hdr.blocksize
    refused = 0
    for trial in range(60):
        g = bytearray(f)
        pos = int(rng.integers(16 + 4 * nblocks, len(f)))
        g[pos] ^= 1 << int(rng.integers(0, 8))
        if trial % 3 == 0:
            g[pos:pos + 4] = b"\x00\x00\x00\x00"
        r, out = cb.decompress(bytes(g), hdr.nbytes)
        try:
            got = hb.CBloscDecompress(bytes(g))
        except hb.ErrDecompressionFailed:
            refused += 1
            assert r < 0, (trial, r)
            continue
        assert r == hdr.nbytes and got == out, trial
    assert refused > 5
    # a bstarts entry that points outside the frame
    g = bytearray(f); g[16:20] = struct.pack("<i", len(f) + 100)
    with pytest.raises(hb.ErrDecompressionFailed):
        hb.CBloscDecompress(bytes(g))


def test_blocks_that_the_old_rule_leaves_unsplit(hb, O, cb):
    # blosc_d splits a block only when the not-split bit is clear AND typesize <= 16 AND blocksize / typesize >= 128 (the rule from
    # before the bit existed, c-blosc < 1.15, still applies): c-blosc 1.21 sets the bit on such frames; with the bit cleared by hand
    # they are what an older writer produced, the library reads them as unsplit blocks and so must the device.
    rng = np.random.default_rng(1)
    x = rng.integers(0, 4, 200000, dtype=np.uint8) * 3
    for ts, bs, shuffle in ((17, 0, 1), (8, 512, 1), (32, 0, 1), (8, 512, 2), (17, 0, 0), (4, 256, 1), (16, 2032, 1), (16, 2048, 1)):
        f = cb.compress(x, 5, shuffle, ts, b"lz4", bs)
        g = bytearray(f); g[2] &= 0xEF
        r, out = cb.decompress(bytes(g), x.size)
        assert r == x.size and out == x.tobytes(), (ts, bs, shuffle)      # (what the library does with it; for 16 / 2048 the bit was clear already)
        assert hb.CBloscDecompress(bytes(g)) == x.tobytes(), (ts, bs, shuffle, hex(f[2]))


def test_forged_geometry_is_refused_before_anything_is_sized(hb, O, cb):
    # a 16-byte header with blocksize 1 / typesize 255 / nbytes N asked for ~4080 N bytes of scratch; a header record with
    # blocksize 0 or typesize 0 handed to the _dev entry point divided by zero on the host
    import ctypes
    L = hb.lib()
    n = 1 << 20
    forged = struct.pack("<BBBBIII", 2, 1, 0x21, 255, n, 1, 16) + b""
    before = L.hb_pool_cached_bytes()
    with pytest.raises(hb.ErrInvalidData):
        hb.CBloscDecompress(forged + bytes(64))
    assert L.hb_pool_cached_bytes() - before < 64 * n
    forged = struct.pack("<BBBBIII", 2, 1, 0x21, 8, n, 4, 16 + 64)       # blocksize below typesize
    with pytest.raises(hb.ErrInvalidData):
        hb.CBloscDecompress(forged + bytes(64))
    buf = hb.PinnedBuffer(4096)               # (the record is refused before any of these addresses is used)
    for ts, bsz, want in ((0, 4096, -2), (4, 0, -2)):
        hdr = hb.CBloscHeader()
        hdr.version, hdr.versionlz, hdr.flags, hdr.typesize, hdr.nbytes, hdr.blocksize, hdr.cbytes, hdr.codec_format = 2, 1, 0x21, ts, 1024, bsz, 64, 1
        rc = L.hb_cblosc_decompress_dev(ctypes.byref(hdr), buf.ptr, 64, buf.ptr + 1024, 1024, buf.ptr + 2048, 2048, buf.ptr + 512, None)
        assert rc == want, (ts, bsz, rc)
    buf.close()


def test_frames_written_here_are_read_by_the_library(hb, O, cb):
    # the other direction: hb_cblosc_compress writes, blosc_decompress_ctx of c-blosc 1.21 reads (and the device decoder too)
    sets = _sets(O)
    sets["tiny"] = np.arange(13, dtype=np.uint8)
    sets["one_chunk_and_a_bit"] = O.synth(O.D_F32, 1024 + 3)
    for name, x in sets.items():
        xb = x.tobytes()
        for ts in (1, 2, 3, 4, 8, 16, 17):
            for shuffle in (0, 1, 2):
                f = hb.CBloscCompress(xb, shuffle, ts)
                h = hb.CBloscParseHeader(f)
                assert (h.version, h.nbytes, h.cbytes, h.codec_format, h.typesize) == (2, len(xb), len(f), 1, ts), (name, ts, shuffle)
                r, out = cb.decompress(f, len(xb))
                assert r == len(xb) and out == xb, (name, ts, shuffle, r)
                assert hb.CBloscDecompress(f) == xb, (name, ts, shuffle)
    # ratio: the chunk-local encoder's, a few bytes per 4 KiB stream on top
    x = O.synth(O.D_F32, (8 << 20) // 4)
    f = hb.CBloscCompress(x.tobytes(), 1, 4)
    g = hb.Compress(x.tobytes(), hb.LZ4, 5, hb.Shuffle1, 4, opts=0)
    assert len(f) < len(g) * 1.02, (len(f), len(g))


def test_a_large_frame(hb, O, cb):
    import time
    x = O.synth(O.D_F32, (256 << 20) // 4)
    f = cb.compress(x, 5, 1, 4)
    t0 = time.perf_counter()
    got = hb.CBloscDecompress(f)
    dt = time.perf_counter() - t0
    assert got == x.tobytes()
    h = hb.CBloscParseHeader(f)
    print(f"C-Blosc-1 frame, 256 MiB f32 shuffle+lz4 (blocksize {h.blocksize}, ratio {len(f) / x.size:.3f}): {x.size / dt / 1e9:.2f} GB/s host->host")


def test_more_streams_than_workgroups_and_a_ragged_tail(hb, O):
    # the stream order of the decoders (hb_cblosc.hip k_cb_decode_small / k_cb_decode: 8 * (k P mod m) + (x + k + pass) mod 8) has to
    # visit every stream exactly once also when there are several passes per workgroup and the stream count is no multiple of 8:
    # 300 MiB + 4 KiB + 13 bytes of float32 written here = 19201 blocks = 76804 streams (one of them the short, unsplit last block)
    n = (300 << 20) + 4096 + 13
    x = np.frombuffer(O.synth(O.D_F32, n // 4 + 1).tobytes()[:n], np.uint8)
    f = hb.CBloscCompress(x.tobytes(), 1, 4)
    h = hb.CBloscParseHeader(f)
    assert h.blocksize == 16384 and h.nbytes == n
    assert hb.CBloscDecompress(f) == x.tobytes()
