"""CPU tests of the drop-in boundary: libhipblosc.so loads, exports every symbol include/hipblosc.h declares,
its host-only helpers (bounds, header parse/serialise, error strings) behave like the reference's, and
every compute entry point fails loudly with HB_ERR_NO_DEVICE when there is no GPU (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hbmod():
    import __graft_entry__ as g
    import hipblosc
    if not os.path.exists(hipblosc.LIB_PATH):
        g.build()
    return hipblosc


def test_every_declared_symbol_is_exported(hbmod):
    text = open(os.path.join(ROOT, "include", "hipblosc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(hb_[a-z0-9_]+)\s*\(", text)))
    assert len(declared) >= 25
    L = ctypes.CDLL(hbmod.LIB_PATH)
    missing = [s for s in declared if not hasattr(L, s)]
    assert not missing, missing
    assert sorted(set(hbmod.EXPORTS)) == declared, "hipblosc.py EXPORTS and include/hipblosc.h disagree"


def test_bounds_and_header_helpers(hbmod):
    L = hbmod.lib()
    for n in (0, 1, 254, 255, 1000, 100000, 1 << 30):
        assert L.hb_lz4_bound(n) == n + n // 255 + 16                      # codec.go:65
        assert L.hb_frame_bound(n) >= 16 + L.hb_lz4_bound(n)
        assert L.hb_index_bound(n) == 32 + 16 * ((n + 4095) // 4096 + 1)
    h = hbmod.Header(2, hbmod.LZ4, hbmod.flagShuffle, 4, 1000, 1000, 321)
    raw = h.Bytes()
    assert raw == bytes([2, 1, 1, 4]) + (1000).to_bytes(4, "little") * 2 + (321).to_bytes(4, "little")   # blosc.go:188-198
    assert hbmod.ParseHeader(raw) == h                                      # blosc.go:165-185
    with pytest.raises(hbmod.ErrInvalidHeader):
        hbmod.ParseHeader(raw[:15])
    with pytest.raises(hbmod.ErrInvalidVersion):
        hbmod.ParseHeader(bytes([1]) + raw[1:])
    # flag -> mode priority, blosc.go:216-224 / blosc_test.go:457-478
    assert hbmod.Header(Flags=0x5).ShuffleMode() == hbmod.BitShuffle
    assert hbmod.Header(Flags=0x1).ShuffleMode() == hbmod.Shuffle1
    assert hbmod.Header(Flags=0x2).ShuffleMode() == hbmod.NoShuffle and hbmod.Header(Flags=0x2).IsMemcpy()
    assert hbmod.codec_string(hbmod.LZ4) == "lz4" and hbmod.codec_string(9) == "unknown(9)"               # blosc.go:67-84
    assert hbmod.shuffle_string(hbmod.BitShuffle) == "bitshuffle" and hbmod.shuffle_string(7) == "unknown(7)"
    assert (hbmod.Version, hbmod.FormatVersion, hbmod.HeaderSize, hbmod.MinHeaderSize) == ("1.0.0", 2, 16, 16)
    assert L.hb_strerror(-1) == b"blosc: invalid compressed data" and L.hb_strerror(-8) == b"blosc: decompression failed"
    assert L.hb_version().decode().count(".") == 2


def test_plugin_registry(hbmod):
    # codec_test.go:81-164
    class Mock:
        def Name(self): return "mock"
        def Compress(self, d, level): return d
        def Decompress(self, d, n): return d
    hbmod.RegisterCodec(100, Mock())
    c, ok = hbmod.GetCodec(100)
    assert ok and c.Name() == "mock"
    assert hbmod.GetCodec(77) == (None, False)
    assert hbmod.LZ4 in hbmod.ListCodecs() and 100 in hbmod.ListCodecs()
    assert hbmod.GetCodec(hbmod.LZ4)[0].Name() == "lz4"
    del hbmod.codecs[100]


def test_argument_errors_need_no_device(hbmod):
    with pytest.raises(hbmod.ErrInvalidData):                               # blosc.go:269-271 comes before anything else
        hbmod.Compress(b"", hbmod.LZ4, 5, hbmod.NoShuffle, 1)
    with pytest.raises(hbmod.ErrInvalidHeader):                             # blosc.go:297-299
        hbmod.Decompress(b"\x02\x01")


def test_no_cpu_fallback(hbmod):
    L = hbmod.lib()
    if L.hb_init() == 0:
        pytest.skip("a HIP device is present")
    assert L.hb_device_count() == 0
    for call in (lambda: hbmod.shuffleBytes(bytes(64), 4), lambda: hbmod.Compress(bytes(64), hbmod.LZ4, 5, hbmod.Shuffle1, 4),
                 lambda: hbmod.HipLZ4Codec().Compress(bytes(64), 5), lambda: hbmod.HipLZ4Codec().Decompress(b"\x10A\x01\x00", 10)):
        with pytest.raises(hbmod.HipBloscError) as e:
            call()
        assert "no HIP device" in str(e.value)


def test_sizes_beyond_uint32_are_refused(hbmod):
    # the frame header holds uint32 sizes (blosc.go:159-161); the reference truncates silently (:363-365) and never
    # returns its own ErrDataTooLarge (:142).  Here: HB_ERR_DATA_TOO_LARGE before any byte is touched.
    L = hbmod.lib()
    a = ctypes.create_string_buffer(64)
    b = ctypes.create_string_buffer(64)
    for n in (0xFFFFFFFF, 1 << 32, 4278190300):          # 16 + n + n//255 + slack no longer fits uint32
        rc = L.hb_compress_frame(ctypes.addressof(a), n, ctypes.addressof(b), 64, hbmod.LZ4, 5, hbmod.Shuffle1, 4, 0, 0)
        assert rc == -6, (n, rc)
    assert L.hb_compress_frame(ctypes.addressof(a), 64, ctypes.addressof(b), 64, 77, 5, 0, 1, 0, 0) == -4    # codec first
    assert L.hb_strerror(-6) == b"blosc: data too large"
