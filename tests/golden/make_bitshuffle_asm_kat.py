#!/usr/bin/env python3
"""make_bitshuffle_asm_kat.py — known-answer vectors for bitShuffle / bitUnshuffle, derived from the only executable statement of the bit
shuffle the reference holds besides its Go loop: the instruction stream of bitShuffleAVX2 / bitUnshuffleAVX2 (shuffle_amd64.s:346-875,
:879-1394 — despite the name, scalar MOVB / SHRQ / ANDQ / SHLQ / ORQ code over general-purpose registers).  Build container only: reads
/root/reference, which does not travel.

The reference's tests require those routines to agree with the scalar Go loop (shuffle.go:156-173: the assembler handles the whole groups
of 8 elements and returns true, the Go caller finishes the partial group and the tail bytes), so running the instruction stream IS a
reference-held statement of the layout for every typesize.  This script parses the two TEXT blocks as text, interprets the subset of Go's
amd64 assembler they use (MOVQ MOVB MOVBQZX LEAQ ADDQ SUBQ INCQ IMULQ XORQ ANDQ ORQ SHLQ SHRQ CMPQ JL JGE JZ JMP RET; operands: registers
and their byte forms, $immediates, off(SP), (base)(index*scale), name+off(FP)) over a flat byte memory — semantics from the Intel SDM and Go's
`cmd/asm` operand order (CMPQ a, b compares a with b; the destination is the last operand) — and runs them on the buffers the reference's own
bitshuffle tests use (makeTestData: byte(i % 256), blosc_test.go:352-359; lengths and typesizes of shuffle_test.go:146-168, :284-316,
:382-435 with n >= 64, where the assembler engages) plus a splitmix byte stream at more typesizes.  Outputs: tests/golden/bitshuffle_asm_kat.json
— inputs as a generator name + length, the routine's return value, the number of whole groups it handled and the bytes it wrote there.

Only numbers leave the reference: what its instructions compute.  No source text is copied.
tests/test_oracle.py holds the CPU oracle to these vectors, tests/test_gpu_filters.py the device kernels.
"""
import json
import os
import re

REF = "/root/reference/shuffle_amd64.s"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bitshuffle_asm_kat.json")
M64 = (1 << 64) - 1

REGS = ["AX", "BX", "CX", "DX", "SI", "DI", "SP", "BP"] + [f"R{i}" for i in range(8, 16)]
BYTE_REGS = {"AL": "AX", "BL": "BX", "CL": "CX", "DL": "DX"}
BYTE_REGS.update({f"R{i}B": f"R{i}" for i in range(8, 16)})


def parse_function(name):
    """Instructions of TEXT ·name as (mnemonic, [operands]) + label -> index."""
    lines = open(REF).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("TEXT") and ("·" + name + "(SB)") in l)
    prog, labels = [], {}
    for l in lines[start + 1:]:
        if l.startswith("TEXT"):
            break
        l = l.split("//")[0].strip()
        if not l:
            continue
        m = re.match(r"^([A-Za-z_][A-Za-z0-9_]*):$", l)
        if m:
            labels[m.group(1)] = len(prog)
            continue
        parts = l.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        prog.append((parts[0], ops))
    return prog, labels


class Machine:
    """Flat little-endian byte memory + the register file; just enough of amd64 for the two routines."""

    def __init__(self, args):
        self.r = {k: 0 for k in REGS}
        self.mem = {}
        self.args = args                     # FP offset -> (value, width)
        self.zf = self.sf_lt = False         # ZF, and "signed less than" of the last CMPQ
        self.r["SP"] = 0x7000_0000
        self.ret = None

    def ld(self, a, w):
        return sum(self.mem.get(a + k, 0) << (8 * k) for k in range(w))

    def st(self, a, w, v):
        for k in range(w):
            self.mem[a + k] = (v >> (8 * k)) & 255

    def ea(self, op):
        m = re.match(r"^(-?\d+)?\((\w+)\)(?:\((\w+)\*(\d+)\))?$", op)
        assert m, op
        off = int(m.group(1) or 0)
        a = off + self.r[m.group(2)]
        if m.group(3):
            a += self.r[m.group(3)] * int(m.group(4))
        return a & M64

    def read(self, op, w):
        if op.startswith("$"):
            return int(op[1:], 0) & ((1 << (8 * w)) - 1)
        if op in self.r:
            return self.r[op] & ((1 << (8 * w)) - 1)
        if op in BYTE_REGS:
            return self.r[BYTE_REGS[op]] & 255
        m = re.match(r"^\w+\+(\d+)\(FP\)$", op)
        if m:
            return self.args[int(m.group(1))] & ((1 << (8 * w)) - 1)
        return self.ld(self.ea(op), w)

    def write(self, op, w, v):
        v &= (1 << (8 * w)) - 1
        if op in self.r:
            assert w == 8
            self.r[op] = v
        elif op in BYTE_REGS:
            k = BYTE_REGS[op]
            self.r[k] = (self.r[k] & ~255) | v
        elif re.match(r"^ret\+\d+\(FP\)$", op):
            self.ret = v
        else:
            self.st(self.ea(op), w, v)

    def run(self, prog, labels, limit=50_000_000):
        pc = steps = 0
        while True:
            steps += 1
            assert steps < limit, "runaway"
            mn, o = prog[pc]
            pc += 1
            if mn == "RET":
                return
            if mn == "JMP":
                pc = labels[o[0]]
            elif mn in ("JL", "JGE", "JZ"):
                take = {"JL": self.sf_lt, "JGE": not self.sf_lt, "JZ": self.zf}[mn]
                if take:
                    pc = labels[o[0]]
            elif mn == "CMPQ":                               # Go order: CMPQ a, b -> flags of a - b
                a, b = self.read(o[0], 8), self.read(o[1], 8)
                sa = a - (1 << 64) if a >> 63 else a
                sb = b - (1 << 64) if b >> 63 else b
                self.sf_lt, self.zf = sa < sb, a == b
            elif mn == "MOVQ":
                self.write(o[1], 8, self.read(o[0], 8))
            elif mn == "MOVB":
                self.write(o[1], 1, self.read(o[0], 1))
            elif mn == "MOVBQZX":
                self.write(o[1], 8, self.read(o[0], 1))
            elif mn == "LEAQ":
                self.write(o[1], 8, self.ea(o[0]))
            elif mn == "INCQ":
                v = (self.read(o[0], 8) + 1) & M64
                self.write(o[0], 8, v); self.zf = v == 0
            elif mn in ("ADDQ", "SUBQ", "IMULQ", "XORQ", "ANDQ", "ORQ", "SHLQ", "SHRQ"):
                s, d = self.read(o[0], 8), self.read(o[1], 8)
                if mn == "ADDQ":
                    v = d + s
                elif mn == "SUBQ":
                    v = d - s
                elif mn == "IMULQ":
                    v = d * s
                elif mn == "XORQ":
                    v = d ^ s
                elif mn == "ANDQ":
                    v = d & s
                elif mn == "ORQ":
                    v = d | s
                elif mn == "SHLQ":
                    v = d << (s & 63)
                else:
                    v = d >> (s & 63)
                v &= M64
                self.write(o[1], 8, v); self.zf = v == 0
            else:
                raise AssertionError(f"instruction outside the subset: {mn} {o}")


def run_routine(name, src, ts):
    """dst, returned bool of ·name(dst, src, typeSize) with len(dst) == len(src)."""
    prog, labels = parse_function(name)
    n = len(src)
    DST, SRC = 0x1000_0000, 0x2000_0000
    m = Machine({0: DST, 8: n, 16: n, 24: SRC, 32: n, 40: n, 48: ts})
    for i, b in enumerate(src):
        m.mem[SRC + i] = b
    m.run(prog, labels)
    touched = sorted(a - DST for a in m.mem if DST <= a < DST + n + 64)
    assert all(t < n for t in touched), "the routine wrote outside dst"
    return bytes(m.mem.get(DST + i, 0) for i in range(n)), bool(m.ret), len(touched)


def gen(kind, n):
    if kind == "i%256":                                     # makeTestData, blosc_test.go:352-359
        return bytes(i % 256 for i in range(n))
    out = bytearray()
    z = 0x9E3779B97F4A7C15
    while len(out) < n:                                      # splitmix64 byte stream (SURVEY.md §8d's h(i), bytes little-endian)
        z = (z + 0x9E3779B97F4A7C15) & M64
        x = z
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
        x ^= x >> 31
        out += x.to_bytes(8, "little")
    return bytes(out[:n])


def main():
    cases = [("i%256", 1003, 4), ("i%256", 127, 8), ("i%256", 64, 4), ("i%256", 128, 8), ("i%256", 97, 4), ("i%256", 100, 4),
             ("i%256", 64, 16),                                   # called (n >= 64) but no whole group: falls back, writes nothing
             ("splitmix", 200, 2), ("splitmix", 300, 6), ("splitmix", 203, 3), ("splitmix", 256, 4), ("splitmix", 321, 5),
             ("splitmix", 512, 8), ("splitmix", 515, 16), ("splitmix", 777, 7)]
    vectors = []
    for kind, n, ts in cases:
        src = gen(kind, n)
        fwd, ok_f, wrote_f = run_routine("bitShuffleAVX2", src, ts)
        ne = n // ts
        assert n >= 64 and ts > 1, "the Go caller only reaches the routine with typeSize > 1 and n >= 64 (shuffle.go:146, :156)"
        groups = ne // 8
        assert ok_f == (groups > 0) and wrote_f == groups * 8 * ts, (kind, n, ts, ok_f, wrote_f, groups)
        prefix = groups * 8 * ts
        inv, ok_i, wrote_i = run_routine("bitUnshuffleAVX2", fwd[:prefix] + src[prefix:], ts)
        assert ok_i == ok_f and wrote_i == prefix
        assert inv[:prefix] == src[:prefix], "the reference's two routines are not inverses of each other on their prefix"
        vectors.append({"input": kind, "n": n, "typesize": ts, "returns": ok_f, "groups": groups, "prefix_bytes": prefix,
                        "bitshuffle_prefix_hex": fwd[:prefix].hex()})
    doc = {"source": "instruction stream of bitShuffleAVX2 / bitUnshuffleAVX2 (shuffle_amd64.s:346-875, :879-1394), interpreted; numbers only",
           "note": "the routines write the first `prefix_bytes` = groups * 8 * typesize bytes and return `returns`; the Go caller finishes the rest "
                   "(shuffle.go:162-173: the partial group and the tail are copied).  bitUnshuffleAVX2 of the prefix restores the input prefix (checked).",
           "inputs": {"i%256": "byte(i % 256), makeTestData of blosc_test.go:352-359",
                      "splitmix": "splitmix64 stream, seed 0x9E3779B97F4A7C15, 8 little-endian bytes per draw (SURVEY.md §8d's finaliser)"},
           "vectors": vectors}
    json.dump(doc, open(OUT, "w"), indent=1)
    print(f"{len(vectors)} vectors -> {OUT}")


if __name__ == "__main__":
    main()
