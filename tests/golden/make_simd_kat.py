#!/usr/bin/env python3
"""make_simd_kat.py — known-answer vectors for the typesize-4 byte shuffle, derived from the only layout data the reference holds:
the constant tables of its SIMD kernels (build container only: reads /root/reference, which does not travel).

The reference pins the output layout of the filters in exactly one place: its AVX2 / NEON kernels must equal the scalar loop byte for
byte (shuffle_amd64_test.go:47-61, shuffle_arm64_test.go:38-52).  Those kernels are a table-driven byte permutation, so the TABLES
(shuffle_amd64.s:35-81 `shuffle4_perm` / `shuffle4_lane`, :86-129 `unshuffle4_lane` / `unshuffle4_perm`; shuffle_arm64.s:28-29
`shuffle4_tbl`, :63-64 `unshuffle4_tbl`) together with the instruction sequences around them (shuffle_amd64.s:183-215, :284-310;
shuffle_arm64.s:107-133, :190-205) ARE a statement of the layout.  This script reads the table constants out of the assembler
text as numbers, runs the documented semantics of the instructions on them (Intel SDM: VPSHUFB — per 128-bit lane, index bit 7
zeroes the byte, low 4 bits select; VPERMD — dst.dword[i] = src.dword[idx.dword[i] & 7]; VPUNPCKLQDQ / VINSERTI128 / VEXTRACTI128
/ VPSRLDQ / VMOVQ moves as written; Arm ARM: TBL — dst.byte[i] = table[idx.byte[i]] if idx < 16 else 0), and writes inputs and
the resulting outputs to simd_tables_kat.json.  tests/test_oracle.py checks the oracle's shuffle / unshuffle against them and
tests/test_gpu_filters.py the device kernels.  The SIMD kernels leave `numElements % 8` (`% 4`) elements and the tail bytes to the
Go caller's scalar finisher (shuffle.go:42-55, :102-115): the vectors record the SIMD-processed prefix and the finisher's part
separately (`simd_elements`), and the tests only hold the oracle / device to the prefix the tables define plus the verbatim tail.

Only numbers leave the reference: table constants and outputs computed from them.  No source text is copied.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "simd_tables_kat.json")


def read_table(path, name):
    """All `DATA name<>+off(SB)/width, $value` constants of one table -> bytes (little-endian, as the assembler lays them out)."""
    pat = re.compile(r"^\s*DATA\s+" + re.escape(name) + r"<>\+(\d+)\(SB\)/(\d+),\s*\$(0x[0-9a-fA-F]+|\d+)")
    size = None
    cells = {}
    for line in open(path):
        m = pat.match(line)
        if m:
            off, width, val = int(m.group(1)), int(m.group(2)), int(m.group(3), 0)
            for k in range(width):
                cells[off + k] = (val >> (8 * k)) & 255
        g = re.match(r"^\s*GLOBL\s+" + re.escape(name) + r"<>\(SB\),\s*RODATA,\s*\$(\d+)", line)
        if g:
            size = int(g.group(1))
    assert size is not None and sorted(cells) == list(range(size)), (name, size, len(cells))
    return bytes(cells[i] for i in range(size))


def dwords(b):
    return [int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(len(b) // 4)]


def vpshufb256(src, mask):
    out = bytearray(32)
    for lane in (0, 16):
        for i in range(16):
            m = mask[lane + i]
            out[lane + i] = 0 if m & 0x80 else src[lane + (m & 15)]
    return bytes(out)


def vpermd(src, idx):
    d = dwords(src)
    ix = dwords(idx)
    return b"".join(d[ix[i] & 7].to_bytes(4, "little") for i in range(8))


def tbl(table16, idx):
    return bytes(table16[i] if i < 16 else 0 for i in idx)


def avx2_shuffle(src, T):
    """shuffle_amd64.s:138-243: 8 elements per iteration; returns (dst with the SIMD-written bytes, elements processed)."""
    n = len(src)
    ne = n // 4
    dst = bytearray(n)
    chunks = ne // 8
    if n < 32 or chunks == 0:
        return bytes(dst), 0
    for c in range(chunks):
        y0 = src[32 * c:32 * c + 32]
        y1 = vpshufb256(y0, T["shuffle4_lane"])
        y0 = vpermd(y1, T["shuffle4_perm"])
        for j in range(4):                                   # qword j -> plane j (VMOVQ / VPSRLDQ $8 / VEXTRACTI128 $1)
            dst[j * ne + 8 * c: j * ne + 8 * c + 8] = y0[8 * j:8 * j + 8]
    return bytes(dst), chunks * 8


def avx2_unshuffle(src, T):
    n = len(src)
    ne = n // 4
    dst = bytearray(n)
    chunks = ne // 8
    if n < 32 or chunks == 0:
        return bytes(dst), 0
    for c in range(chunks):
        y0 = b"".join(src[j * ne + 8 * c: j * ne + 8 * c + 8] for j in range(4))   # VMOVQ x4, VPUNPCKLQDQ x2, VINSERTI128
        y1 = vpermd(y0, T["unshuffle4_perm"])
        dst[32 * c:32 * c + 32] = vpshufb256(y1, T["unshuffle4_lane"])
    return bytes(dst), chunks * 8


def neon_shuffle(src, T):
    """shuffle_arm64.s:65-149: 4 elements per iteration."""
    n = len(src)
    ne = n // 4
    dst = bytearray(n)
    chunks = ne // 4
    if n < 16 or chunks == 0:
        return bytes(dst), 0
    for c in range(chunks):
        v1 = tbl(src[16 * c:16 * c + 16], T["shuffle4_tbl"])
        for j in range(4):                                   # VMOV V1.S[j] -> MOVW to plane j
            dst[j * ne + 4 * c: j * ne + 4 * c + 4] = v1[4 * j:4 * j + 4]
    return bytes(dst), chunks * 4


def neon_unshuffle(src, T):
    n = len(src)
    ne = n // 4
    dst = bytearray(n)
    chunks = ne // 4
    if n < 16 or chunks == 0:
        return bytes(dst), 0
    for c in range(chunks):
        v0 = b"".join(src[j * ne + 4 * c: j * ne + 4 * c + 4] for j in range(4))
        dst[16 * c:16 * c + 16] = tbl(v0, T["unshuffle4_tbl"])
    return bytes(dst), chunks * 4


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    amd, arm = os.path.join(REF, "shuffle_amd64.s"), os.path.join(REF, "shuffle_arm64.s")
    T = {k: read_table(amd, k) for k in ("shuffle4_lane", "shuffle4_perm", "unshuffle4_lane", "unshuffle4_perm")}
    T.update({k: read_table(arm, k) for k in ("shuffle4_tbl", "unshuffle4_tbl")})
    vectors = []
    # inputs: every byte distinct inside the SIMD window (a permutation test), a second pattern with repeats, and the lengths the
    # reference's own SIMD tests use (shuffle_amd64_test.go:23-33: 32, 64, 1000, 1003; arm64: 16, 32, 100, 1000) plus ragged ones
    for n in (16, 20, 32, 35, 64, 100, 127, 250, 1000, 1003):
        for pat in ("ident", "mul37"):
            src = bytes(((i if pat == "ident" else i * 37 + 11) & 255) for i in range(n))
            for isa, fwd, inv in (("avx2", avx2_shuffle, avx2_unshuffle), ("neon", neon_shuffle, neon_unshuffle)):
                out, ne_simd = fwd(src, T)
                if ne_simd == 0:
                    continue                                  # the hook returns false: nothing pinned by the tables
                # unshuffle input: a plane image whose SIMD-covered part is `out`'s; output must give the source elements back
                back, ne2 = inv(out, T)
                assert ne2 == ne_simd and back[:4 * ne_simd] == src[:4 * ne_simd], (isa, n)
                vectors.append({"isa": isa, "n": n, "pattern": pat, "simd_elements": ne_simd, "src": src.hex(),
                                "shuffled_simd_bytes": out.hex(), "unshuffled_prefix": back[:4 * ne_simd].hex()})
    doc = {"source": "constants of shuffle_amd64.s (shuffle4_lane/_perm, unshuffle4_lane/_perm) and shuffle_arm64.s (shuffle4_tbl, unshuffle4_tbl) "
                     "run through the ISA semantics of VPSHUFB+VPERMD / TBL as the kernels sequence them; see make_simd_kat.py",
           "tables": {k: v.hex() for k, v in T.items()},
           "layout": "shuffled_simd_bytes[j*ne + i] for i < simd_elements is byte j of element i (ne = n // 4); bytes the SIMD kernel "
                     "does not write are 00 here and belong to the scalar finisher (shuffle.go:42-55)",
           "vectors": vectors}
    json.dump(doc, open(OUT, "w"), indent=0)
    print(f"{len(vectors)} vectors -> {OUT}")


if __name__ == "__main__":
    main()
