"""Generator of arbitrary VALID LZ4 blocks for the decoder tests: sequences are drawn at random (literal lengths, match lengths,
offsets) instead of coming from an encoder, so that shapes no encoder of ours emits get decoded too -- offsets up to 65535,
overlapping matches with every small period, matches that reach exactly to the first byte of the block, sequences without
literals, length extensions of one to thousands of bytes, runs of MiB.  The block only has to obey the FORMAT
(token, lengths, offset <= bytes produced so far); the expected output is what the oracle's decoder makes of it."""
import numpy as np


def _ext(n):
    out = bytearray()
    while n >= 255:
        out.append(255); n -= 255
    out.append(n)
    return out


def random_block(rng, target_out, regime_len=1 << 20, align=1):
    """Returns (block bytes, decoded length).  `target_out`: decoded bytes wanted at least (the last sequence is literal-only and
    brings the decoded length to a multiple of `align`)."""
    s = bytearray()
    produced = 0
    regimes = ("dense", "runs", "literal", "far", "mixed", "periodic", "huge")
    while produced < target_out:
        regime = regimes[int(rng.integers(0, len(regimes)))]
        stop = min(target_out, produced + int(rng.integers(regime_len // 4, regime_len * 2)))
        period_tok = None
        while produced < stop:
            if regime == "dense":
                lit = int(rng.integers(0, 3)); ml = int(rng.integers(4, 16)); off = int(rng.integers(1, 4096))
            elif regime == "runs":
                lit = int(rng.integers(0, 8)); ml = int(rng.integers(4, 6000)); off = int(rng.integers(1, 40))
            elif regime == "literal":
                lit = int(rng.integers(100, 70000)); ml = int(rng.integers(4, 12)); off = int(rng.integers(1, 65536))
            elif regime == "far":
                lit = int(rng.integers(0, 20)); ml = int(rng.integers(4, 300)); off = int(rng.integers(30000, 65536))
            elif regime == "periodic":
                if period_tok is None:
                    period_tok = (int(rng.integers(0, 7)), int(rng.integers(3000, 5000)), int(rng.integers(1, 3)))
                lit, ml, off = period_tok
            elif regime == "huge":
                lit = int(rng.integers(0, 3)) * int(rng.integers(0, 600000)); ml = int(rng.integers(4, 900000)); off = int(rng.integers(1, 65536))
                stop = min(stop, produced + lit + ml)
            else:
                lit = int(rng.integers(0, 40)); ml = int(rng.integers(4, 80)); off = int(rng.integers(1, 65536))
            if produced == 0 and lit == 0:
                lit = 1                                            # a match needs something in front of it
            off = max(1, min(off, produced + lit))
            tok = (min(lit, 15) << 4) | min(ml - 4, 15)
            s.append(tok)
            if lit >= 15: s += _ext(lit - 15)
            s += rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
            s += bytes((off & 255, off >> 8))
            if ml - 4 >= 15: s += _ext(ml - 4 - 15)
            produced += lit + ml
    lit = int(rng.integers(0, 30))                                  # the final, literal-only sequence
    lit += (-(produced + lit)) % align
    s.append(min(lit, 15) << 4)
    if lit >= 15: s += _ext(lit - 15)
    s += rng.integers(0, 256, lit, dtype=np.uint8).tobytes()
    return bytes(s), produced + lit
