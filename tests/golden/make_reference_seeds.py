#!/usr/bin/env python3
"""Generates tests/golden/reference_seeds.json: the deterministic seed inputs the reference's own fuzz targets hold.

These are the only frame-level vectors the reference owns ("What's missing" #6).  The BYTES are built here
exactly as the cited lines of /root/reference/fuzz_test.go build them (the Go expressions are restated as Python next
to each seed); nothing is read from the reference at run time.  The EXPECTED OUTCOME of each seed is not produced by
any implementation: it is derived by hand from blosc.go:165-185 (ParseHeader), blosc.go:296-303 (DecompressWithSize)
and blosc.go:377-434 (decompressBackend), and the derivation is written into the seed's `why`.

  decompress_seeds : FuzzDecompress edge-case seeds, fuzz_test.go:26-133.  Expected result of Decompress(data) AND of
                     DecompressWithSize(data, ts) for ts in {0,1,2,4,8} (fuzz_test.go:155-158): the seeds are built so
                     that the outcome does not depend on ts (stated per seed).
  header_seeds     : FuzzParseHeader seeds, fuzz_test.go:293-352.  Expected ParseHeader outcome and, since they are also
                     legal inputs of Decompress, the expected Decompress outcome.
  compress_seeds   : FuzzCompress seed INPUTS, fuzz_test.go:167-203 (+ makeCompressibleData / makeRandomData,
                     fuzz_test.go:453-471).  The reference asserts for them: Compress(x, codec, 5, NoShuffle, 1) then
                     Decompress gives x back (fuzz_test.go:217-237), same for levels -1,0,1,5,9,10,100 (:256-266); every
                     shuffle x typesize {1,2,4,8} combination must not panic (:241-253).

Error names are the reference's sentinels (blosc.go:125-149); the C ABI code of each is in include/hipblosc.h.
Run from the repo root:  python tests/golden/make_reference_seeds.py
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
HeaderSize = 16            # blosc.go:118-121
FormatVersion = 2          # blosc.go:51
LZ4 = 1                    # blosc.go:57-64
flagShuffle, flagMemcpy, flagBitShuffle = 0x1, 0x2, 0x4     # blosc.go:110-115


def make(n):
    return bytearray(n)


def put32(b, off, v):
    b[off:off + 4] = struct.pack("<I", v)


def decompress_seeds():
    S = []

    def add(name, cite, data, expect, why, out=None):
        S.append(dict(name=name, cite=cite, data=bytes(data).hex(), expect=expect, out=None if out is None else bytes(out).hex(), why=why))

    # fuzz_test.go:28-31  f.Add([]byte{}) ... f.Add([]byte{0x02, 0x01, 0x00, 0x04})
    for k, d in enumerate([b"", b"\x02", b"\x02\x01", b"\x02\x01\x00\x04"]):
        add(f"short_{len(d)}", "fuzz_test.go:28-31", d, "ErrInvalidHeader", "len(data) < HeaderSize -> bare ErrInvalidHeader, blosc.go:297-299")
    # :34-38  wrongVersion := make([]byte, HeaderSize); [0] = 99; PutUint32([4:8], 100); PutUint32([12:16], 116)
    w = make(HeaderSize); w[0] = 99; put32(w, 4, 100); put32(w, 12, 116)
    add("wrong_version_99", "fuzz_test.go:34-38", w, "ErrInvalidVersion", "Version 99 != 2 -> ErrInvalidVersion, blosc.go:180-182 (before any size check)")
    # :41-43  zeroVersion: all zero
    add("version_0", "fuzz_test.go:41-43", make(HeaderSize), "ErrInvalidVersion", "Version 0 != 2, blosc.go:180-182")
    # :46-48  oldVersion[0] = 1
    o = make(HeaderSize); o[0] = 1
    add("version_1", "fuzz_test.go:46-48", o, "ErrInvalidVersion", "Version 1 != 2, blosc.go:180-182")
    # :51-59  validHeaderTruncated: LZ4, flags 0, ts 4, nbytes = blocksize = cbytes = 1000, only 16 bytes present
    t = make(HeaderSize); t[0] = FormatVersion; t[1] = LZ4; t[2] = 0; t[3] = 4
    put32(t, 4, 1000); put32(t, 8, 1000); put32(t, 12, 1000)
    add("truncated_payload", "fuzz_test.go:51-59", t, "ErrInvalidData", "NBytesComp 1000 > len(data) 16 -> bare ErrInvalidData, blosc.go:385-387")
    # :62-70  memcpyHeader: 26 bytes, flags = flagMemcpy, ts 4, nbytes = blocksize = 100, cbytes = 26
    m = make(HeaderSize + 10); m[0] = FormatVersion; m[1] = LZ4; m[2] = flagMemcpy; m[3] = 4
    put32(m, 4, 100); put32(m, 8, 100); put32(m, 12, HeaderSize + 10)
    add("memcpy_wrong_sizes", "fuzz_test.go:62-70", m, "ErrSizeMismatch",
        "memcpy flag: decompressed = the 10 payload bytes (blosc.go:398-400); no filter flag; len 10 != NBytesOrig 100 -> ErrSizeMismatch, blosc.go:429-431 (any typeSize)")
    # :73-81  invalidCodec: 66 bytes, codec 255, flags 0, ts 1, nbytes = blocksize = 50, cbytes = 66
    c = make(HeaderSize + 50); c[0] = FormatVersion; c[1] = 255; c[2] = 0; c[3] = 1
    put32(c, 4, 50); put32(c, 8, 50); put32(c, 12, HeaderSize + 50)
    add("codec_255", "fuzz_test.go:73-81", c, "ErrInvalidCodec", "sizes fine, not memcpy, codecs[Codec(255)] missing -> ErrInvalidCodec, blosc.go:403-407")
    # :84-89  zeroOrig: LZ4, nbytes 0, cbytes 16
    z = make(HeaderSize); z[0] = FormatVersion; z[1] = LZ4; put32(z, 4, 0); put32(z, 12, HeaderSize)
    add("nbytes_0", "fuzz_test.go:84-89", z, "ok",
        "payload is empty; lz4Codec.Decompress(empty, 0): UncompressBlock of an empty source returns 0, nil (codec.go:77-84) -> "
        "buf[:0]; flags 0 -> no filter; len 0 == NBytesOrig 0 -> success with an empty result (any typeSize)", out=b"")
    # :92-98  maxSizes: all three sizes 0xFFFFFFFF
    x = make(HeaderSize); x[0] = FormatVersion; x[1] = LZ4
    put32(x, 4, 0xFFFFFFFF); put32(x, 8, 0xFFFFFFFF); put32(x, 12, 0xFFFFFFFF)
    add("max_sizes", "fuzz_test.go:92-98", x, "ErrInvalidData", "int(NBytesComp) = 4294967295 > len(data) 16 -> ErrInvalidData, blosc.go:385-387 (before any allocation)")
    # :101-111  shuffleHeader for ts in {0,1,2,4,8,16,255}: 36 bytes, flags = flagShuffle, nbytes = blocksize = 20, cbytes = 36
    for ts in (0, 1, 2, 4, 8, 16, 255):
        s = make(HeaderSize + 20); s[0] = FormatVersion; s[1] = LZ4; s[2] = flagShuffle; s[3] = ts
        put32(s, 4, 20); put32(s, 8, 20); put32(s, 12, HeaderSize + 20)
        add(f"shuffle_flag_ts{ts}", "fuzz_test.go:101-111", s, "ErrDecompressionFailed",
            "payload = 20 zero bytes as an LZ4 block: token 0x00 (no literals, match nibble 0), input not exhausted, "
            "offset bytes 00 00 -> offset 0 is invalid -> UncompressBlock error -> ErrDecompressionFailed, blosc.go:411-413; "
            "the un-shuffle (and so the typeSize) is never reached")
    # :114-122  bitshuffleHeader: flags = flagBitShuffle, ts 4, same sizes
    b = make(HeaderSize + 20); b[0] = FormatVersion; b[1] = LZ4; b[2] = flagBitShuffle; b[3] = 4
    put32(b, 4, 20); put32(b, 8, 20); put32(b, 12, HeaderSize + 20)
    add("bitshuffle_flag", "fuzz_test.go:114-122", b, "ErrDecompressionFailed", "as shuffle_flag_*: the zero payload is not a valid LZ4 block (offset 0)")
    # :125-133  allFlags: flags = 0xFF, ts 4, same sizes
    a = make(HeaderSize + 20); a[0] = FormatVersion; a[1] = LZ4; a[2] = 0xFF; a[3] = 4
    put32(a, 4, 20); put32(a, 8, 20); put32(a, 12, HeaderSize + 20)
    add("all_flags", "fuzz_test.go:125-133", a, "ok",
        "0xFF includes flagMemcpy: decompressed = the 20 zero payload bytes (blosc.go:398-400); bitshuffle flag wins "
        "(blosc.go:422-423) and bit-un-shuffling zeros with any typeSize gives zeros; len 20 == NBytesOrig -> 20 zero bytes",
        out=bytes(20))
    return S


def header_seeds():
    S = []

    def add(name, cite, data):
        data = bytes(data)
        # ParseHeader, blosc.go:165-185
        if len(data) < HeaderSize:
            parse = "ErrInvalidHeader"
            fields = None
        else:
            ver, codec, flags, ts = data[0], data[1], data[2], data[3]
            nbytes, bs, cbytes = struct.unpack("<III", data[4:16])
            fields = dict(Version=ver, VersionLZ=codec, Flags=flags, TypeSize=ts, NBytesOrig=nbytes, BlockSize=bs, NBytesComp=cbytes)
            parse = "ok" if ver == FormatVersion else "ErrInvalidVersion"
        # Decompress, blosc.go:296-303 + :377-434, for these seeds (their payloads are empty or absent)
        if parse != "ok":
            dec = parse
            why = "Decompress fails where ParseHeader fails (blosc.go:297-299, :379-382)"
        elif fields["NBytesComp"] > len(data) or fields["NBytesComp"] < HeaderSize:
            dec = "ErrInvalidData"
            why = "NBytesComp > len(data) or < HeaderSize -> bare ErrInvalidData, blosc.go:385-390"
        else:
            # only reachable here with NBytesComp == 16 (empty payload) in a 16-byte seed with VersionLZ 0 and flags 0
            assert fields["NBytesComp"] == HeaderSize and fields["VersionLZ"] == 0 and not (fields["Flags"] & flagMemcpy), name
            dec = "ErrInvalidCodec"
            why = "empty payload, not memcpy, Codec(0) = BloscLZ is not in the registry (codec.go:27-33) -> ErrInvalidCodec, blosc.go:403-407"
        S.append(dict(name=name, cite=cite, data=data.hex(), parse=parse, fields=fields if parse == "ok" else None, decompress=dec, why=why))

    # fuzz_test.go:305-310
    for d in (b"", b"\x02", b"\x02\x01", b"\x02\x01\x00", b"\x02\x01\x00\x04", bytes(15)):
        add(f"short_{len(d)}", "fuzz_test.go:305-310", d)
    v = make(HeaderSize); v[0] = FormatVersion                   # :313-315 validHeader
    add("valid_zero_fields", "fuzz_test.go:313-315", v)
    add("all_zero", "fuzz_test.go:318", make(HeaderSize))        # :318
    add("all_ff", "fuzz_test.go:321-325", b"\xff" * HeaderSize)  # :321-325
    for ver in range(0, 11):                                     # :328-332
        h = make(HeaderSize); h[0] = ver
        add(f"version_{ver}", "fuzz_test.go:328-332", h)
    for flags in range(0, 0x100):                                # :335-340
        h = make(HeaderSize); h[0] = FormatVersion; h[2] = flags
        add(f"flags_{flags:02x}", "fuzz_test.go:335-340", h)
    for ts in range(0, 17):                                      # :343-348
        h = make(HeaderSize); h[0] = FormatVersion; h[3] = ts
        add(f"typesize_{ts}", "fuzz_test.go:343-348", h)
    for size in (0, 1, 15, 16, 17, 100, 1000, 0x7FFFFFFF, 0xFFFFFFFF):   # :351-358
        h = make(HeaderSize); h[0] = FormatVersion
        put32(h, 4, size); put32(h, 8, size); put32(h, 12, size)
        add(f"sizes_{size}", "fuzz_test.go:351-358", h)
    e = make(HeaderSize + 100); e[0] = FormatVersion             # :361-363 extraBytes
    add("extra_bytes", "fuzz_test.go:361-363", e)
    return S


def makeCompressibleData(size):     # fuzz_test.go:453-459
    return bytes(i % 256 for i in range(size))


def makeRandomData(size):           # fuzz_test.go:462-471
    out = bytearray(size)
    x = 12345
    for i in range(size):
        x = (x * 1103515245 + 12345) & 0xFFFFFFFF
        out[i] = (x >> 16) & 0xFF
    return bytes(out)


def compress_seeds():
    S = []

    def add(name, cite, data):
        S.append(dict(name=name, cite=cite, data=bytes(data).hex()))

    add("one_zero", "fuzz_test.go:167", b"\x00")
    add("four_zero", "fuzz_test.go:168", bytes(4))
    add("four_ff", "fuzz_test.go:169", b"\xff" * 4)
    for n in (16, 256, 1024):
        add(f"compressible_{n}", "fuzz_test.go:170-172", makeCompressibleData(n))
    for n in (16, 256, 1024):
        add(f"random_{n}", "fuzz_test.go:173-175", makeRandomData(n))
    add("repeat_aa_100", "fuzz_test.go:178", b"\xaa" * 100)
    add("repeat_00ff_100", "fuzz_test.go:179", b"\x00\xff" * 100)
    add("repeat_1234_100", "fuzz_test.go:180", bytes([1, 2, 3, 4]) * 100)
    add("one_42", "fuzz_test.go:183", bytes([42]))
    for n in (15, 16, 17, 100, 255, 256, 4096):
        add(f"zeros_{n}", "fuzz_test.go:184-190", bytes(n))
    add("aligned4", "fuzz_test.go:193-197", bytes(i & 0xFF for i in range(256)))
    add("aligned8", "fuzz_test.go:199-203", bytes(i % 8 for i in range(256)))
    return S


def main():
    out = dict(
        note="seed inputs of the reference's fuzz targets, restated (see make_reference_seeds.py); expected outcomes derived by hand from blosc.go",
        decompress_with_size_sweep=[0, 1, 2, 4, 8],                       # fuzz_test.go:155-158
        compress_levels=[-1, 0, 1, 5, 9, 10, 100],                        # fuzz_test.go:256
        compress_odd_typesizes=[-1, 0, 3, 7, 16, 32, 1000],               # fuzz_test.go:269
        decompress_seeds=decompress_seeds(), header_seeds=header_seeds(), compress_seeds=compress_seeds())
    with open(os.path.join(HERE, "reference_seeds.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print(len(out["decompress_seeds"]), "decompress seeds,", len(out["header_seeds"]), "header seeds,", len(out["compress_seeds"]), "compress seeds")


if __name__ == "__main__":
    main()
