/*
 * blosc_oracle.c — CPU restatement of go-blosc's Shuffle + LZ4 hot path (see blosc_oracle.h).
 * TEST INFRASTRUCTURE ONLY: the product (go-blosc_amd/) never links this.
 * Plain C11, no dependencies.  Citations are file:line into /root/reference.
 */
#include "blosc_oracle.h"

#include <string.h>
#include <stdlib.h>

/* ------------------------------------------------------------------------- */
/* filters                                                                   */
/* ------------------------------------------------------------------------- */

/* shuffle.go:16-73.  `typeSize <= 1 || len(src) < typeSize` returns src itself (:17-19);
 * the C side has distinct buffers, so that case is a plain copy.  The SIMD branches
 * (:26-57) produce the same bytes as the scalar loop (:60-64); only the loop is restated. */
void ob_shuffle(uint8_t *dst, const uint8_t *src, size_t n, int typesize) {
    if (typesize <= 1 || n < (size_t)typesize) { memcpy(dst, src, n); return; }
    const size_t ts = (size_t)typesize, ne = n / ts;
    for (size_t i = 0; i < ne; i++)                 /* :60 */
        for (size_t j = 0; j < ts; j++)             /* :61 */
            dst[j * ne + i] = src[i * ts + j];      /* :62 */
    if (n % ts) memcpy(dst + ne * ts, src + ne * ts, n - ne * ts);  /* :67-70 */
}

/* shuffle.go:76-133, scalar loop :120-124, tail :127-130 */
void ob_unshuffle(uint8_t *dst, const uint8_t *src, size_t n, int typesize) {
    if (typesize <= 1 || n < (size_t)typesize) { memcpy(dst, src, n); return; }
    const size_t ts = (size_t)typesize, ne = n / ts;
    for (size_t i = 0; i < ne; i++)
        for (size_t j = 0; j < ts; j++)
            dst[i * ts + j] = src[j * ne + i];      /* :122 */
    if (n % ts) memcpy(dst + ne * ts, src + ne * ts, n - ne * ts);
}

/* shuffle.go:145-219.  Groups of 8 elements (:177-178); per byte position gather 8 bytes
 * (:186-189), 8x8 bit transpose MSB-first (:192-200); leftover elements (:206-210) and
 * tail bytes (:213-216) verbatim. */
void ob_bitshuffle(uint8_t *dst, const uint8_t *src, size_t n, int typesize) {
    if (typesize <= 1 || n < (size_t)typesize) { memcpy(dst, src, n); return; }
    const size_t ts = (size_t)typesize, ne = n / ts, ng = ne / 8;
    for (size_t g = 0; g < ng; g++) {
        const size_t base = g * 8 * ts;                                  /* :181-182 */
        for (size_t b = 0; b < ts; b++) {
            uint8_t bytes[8];
            for (int e = 0; e < 8; e++) bytes[e] = src[base + (size_t)e * ts + b];   /* :188 */
            for (int k = 0; k < 8; k++) {                                /* outBit :192 */
                uint8_t out = 0;
                for (int e = 0; e < 8; e++)                              /* inByte :194 */
                    if (bytes[e] & (1u << (7 - k))) out |= (uint8_t)(1u << (7 - e));  /* :195-196 */
                dst[base + b * 8 + (size_t)k] = out;                     /* :199 */
            }
        }
    }
    const size_t done = ng * 8 * ts;
    if (done < n) memcpy(dst + done, src + done, n - done);              /* :206-216 */
}

/* shuffle.go:222-295.  Gather 8 shuffled bytes (:263-266), transpose back (:269-276). */
void ob_bitunshuffle(uint8_t *dst, const uint8_t *src, size_t n, int typesize) {
    if (typesize <= 1 || n < (size_t)typesize) { memcpy(dst, src, n); return; }
    const size_t ts = (size_t)typesize, ne = n / ts, ng = ne / 8;
    for (size_t g = 0; g < ng; g++) {
        const size_t base = g * 8 * ts;
        for (size_t b = 0; b < ts; b++) {
            uint8_t bytes[8];
            for (int i = 0; i < 8; i++) bytes[i] = src[base + b * 8 + (size_t)i];    /* :265 */
            for (int e = 0; e < 8; e++) {                                /* outElem :269 */
                uint8_t out = 0;
                for (int i = 0; i < 8; i++)                              /* inBit :271 */
                    if (bytes[i] & (1u << (7 - e))) out |= (uint8_t)(1u << (7 - i)); /* :272-273 */
                dst[base + (size_t)e * ts + b] = out;                    /* :276 */
            }
        }
    }
    const size_t done = ng * 8 * ts;
    if (done < n) memcpy(dst + done, src + done, n - done);              /* :282-292 */
}

void ob_filter(int op, uint8_t *dst, const uint8_t *src, size_t n, int typesize) {
    switch (op) {
    case 0: ob_shuffle(dst, src, n, typesize); break;
    case 1: ob_unshuffle(dst, src, n, typesize); break;
    case 2: ob_bitshuffle(dst, src, n, typesize); break;
    case 3: ob_bitunshuffle(dst, src, n, typesize); break;
    default: memcpy(dst, src, n);
    }
}

/* ------------------------------------------------------------------------- */
/* LZ4 block codec — restates github.com/pierrec/lz4/v4 v4.1.23 (go.mod:7),   */
/* reached from codec.go:65-66 (compress) and codec.go:79 (decompress).       */
/* The module is not in /root/reference: this follows its published           */
/* algorithm; compressed bytes are PARITY UNPINNED.    */
/* ------------------------------------------------------------------------- */

size_t ob_lz4_bound(size_t n) { return n + n / 255 + 16; }   /* lz4.CompressBlockBound, codec.go:65 */

static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

#define LZ4_MINMATCH 4
#define LZ4_MFLIMIT  14          /* 10 + minMatch: last match cannot start within the last 14 bytes */
#define LZ4_WINSIZE  65536
#define LZ4_HASHLOG  16
#define LZ4_SKIPLOG  7           /* adaptSkipLog: step = 1 + (bytes since last match >> 7) */

/* blockHash: hash of the low 6 bytes, prime6bytes = 227718039650203 */
static inline uint32_t lz4_hash(uint64_t x) {
    return (uint32_t)(((x << 16) * 227718039650203ULL) >> (64 - LZ4_HASHLOG));
}

/* 16-bit position table with window reconstruction (Compressor.get / put) */
static inline int64_t ht_get(const uint16_t *t, uint32_t h, int64_t si) {
    int64_t i = (int64_t)t[h] + (si & ~(int64_t)(LZ4_WINSIZE - 1));
    if (i >= si) i -= LZ4_WINSIZE;
    return i;
}

/* lz4.CompressBlock(src, dst, nil) with len(dst) >= CompressBlockBound(len(src)), the only
 * way codec.go:65-66 calls it; so the "incompressible -> 0" exits are never taken and
 * incompressible input comes back expanded (then blosc.go:342 picks memcpy).
 * Greedy single pass: probe s, s+1, s+2; skip acceleration; backward extension; 64 KiB window. */
int64_t ob_lz4_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (cap < ob_lz4_bound(n)) return OB_ERR_SHORT_BUFFER;
    uint16_t *table = (uint16_t *)calloc((size_t)1 << LZ4_HASHLOG, sizeof(uint16_t));
    if (!table) return OB_ERR_COMPRESSION_FAILED;
    int64_t si = 0, di = 0, anchor = 0;
    const int64_t len = (int64_t)n, sn = len - LZ4_MFLIMIT;

    while (si < sn) {
        const uint64_t match = rd64(src + si);      /* si + 8 <= len - 6 */
        uint32_t h = lz4_hash(match), h2 = lz4_hash(match >> 8);
        int64_t ref = ht_get(table, h, si), ref2 = ht_get(table, h2, si + 1);
        table[h] = (uint16_t)si; table[h2] = (uint16_t)(si + 1);
        int64_t offset = si - ref;
        if (offset <= 0 || offset >= LZ4_WINSIZE || ref < 0 || (uint32_t)match != rd32(src + ref)) {
            h = lz4_hash(match >> 16);
            int64_t ref3 = ht_get(table, h, si + 2);
            si += 1; offset = si - ref2;
            if (offset <= 0 || offset >= LZ4_WINSIZE || ref2 < 0 || (uint32_t)(match >> 8) != rd32(src + ref2)) {
                si += 1; offset = si - ref3;
                table[h] = (uint16_t)si;
                if (offset <= 0 || offset >= LZ4_WINSIZE || ref3 < 0 || (uint32_t)(match >> 16) != rd32(src + ref3)) {
                    si += 2 + ((si - anchor) >> LZ4_SKIPLOG);
                    continue;
                }
            }
        }
        /* match found */
        int64_t llen = si - anchor, mlen = 4;
        int64_t toff = si - offset - 1;
        while (llen > 0 && toff >= 0 && src[si - 1] == src[toff]) { si--; toff--; llen--; mlen++; }
        si += mlen;
        const int64_t mbase = si;
        while (si + 8 <= sn) {
            uint64_t x = rd64(src + si) ^ rd64(src + si - offset);
            if (x == 0) { si += 8; } else { si += __builtin_ctzll(x) >> 3; break; }
        }
        mlen += si - mbase;                 /* total match length */
        int64_t mcode = mlen - LZ4_MINMATCH;
        uint8_t *tok = dst + di++;
        *tok = (uint8_t)(mcode < 15 ? mcode : 15);
        if (llen < 15) { *tok |= (uint8_t)(llen << 4); }
        else {
            *tok |= 0xF0;
            int64_t l = llen - 15;
            for (; l >= 255; l -= 255) dst[di++] = 255;
            dst[di++] = (uint8_t)l;
        }
        memcpy(dst + di, src + anchor, (size_t)llen); di += llen;
        dst[di++] = (uint8_t)offset; dst[di++] = (uint8_t)(offset >> 8);
        if (mcode >= 15) {
            int64_t m = mcode - 15;
            for (; m >= 255; m -= 255) dst[di++] = 255;
            dst[di++] = (uint8_t)m;
        }
        anchor = si;
        if (si >= sn) break;
        table[lz4_hash(rd64(src + si - 2))] = (uint16_t)(si - 2);
    }
    /* last literals */
    int64_t llen = len - anchor;
    if (llen < 15) { dst[di++] = (uint8_t)(llen << 4); }
    else {
        dst[di++] = 0xF0;
        int64_t l = llen - 15;
        for (; l >= 255; l -= 255) dst[di++] = 255;
        dst[di++] = (uint8_t)l;
    }
    memcpy(dst + di, src + anchor, (size_t)llen); di += llen;
    free(table);
    (void)cap;
    return di;
}

/* lz4.UncompressBlock(src, dst) (codec.go:79): returns bytes written (may be < cap; the
 * frame layer turns that into ErrSizeMismatch, blosc.go:429-431), or a negative code for
 * malformed input (ErrInvalidSourceShortBuffer -> ErrDecompressionFailed, blosc.go:411-413).
 *   - empty src            -> 0, no error
 *   - stream may end right after a match (the Go loop simply runs out of input)
 *   - after literals: si == len(src) && match nibble == 0 -> done; si >= len(src) -> error
 *   - offset == 0, offset > bytes produced so far (no dictionary), output overflow,
 *     truncated length bytes / offset -> error
 * The module's "shortcut" copies (16/18-byte over-copies) do not change the produced bytes
 * or the accept/reject set; they are not restated. */
int64_t ob_lz4_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (n == 0) return 0;
    size_t si = 0, di = 0;
    while (si < n) {
        const unsigned b = src[si++];
        size_t llen = b >> 4;
        if (llen > 0) {
            if (llen == 15) {
                for (;;) {
                    if (si >= n) return OB_ERR_DECOMPRESSION_FAILED;
                    unsigned x = src[si++];
                    llen += x;
                    if (x != 255) break;
                }
            }
            if (llen > n - si || llen > cap - di) return OB_ERR_DECOMPRESSION_FAILED;
            memcpy(dst + di, src + si, llen);
            si += llen; di += llen;
        }
        size_t mlen = b & 15;
        if (si == n && mlen == 0) break;
        if (si >= n) return OB_ERR_DECOMPRESSION_FAILED;
        if (n - si < 2) return OB_ERR_DECOMPRESSION_FAILED;
        const size_t offset = (size_t)src[si] | ((size_t)src[si + 1] << 8);
        if (offset == 0) return OB_ERR_DECOMPRESSION_FAILED;
        si += 2;
        mlen += LZ4_MINMATCH;
        if (mlen == LZ4_MINMATCH + 15) {
            for (;;) {
                if (si >= n) return OB_ERR_DECOMPRESSION_FAILED;
                unsigned x = src[si++];
                mlen += x;
                if (x != 255) break;
            }
        }
        if (di < offset) return OB_ERR_DECOMPRESSION_FAILED;      /* no dictionary */
        if (mlen > cap - di) return OB_ERR_DECOMPRESSION_FAILED;
        const uint8_t *m = dst + di - offset;
        for (size_t k = 0; k < mlen; k++) dst[di + k] = m[k];    /* overlapping forward copy */
        di += mlen;
    }
    return (int64_t)di;
}

/* ------------------------------------------------------------------------- */
/* Snappy block codec — codec.go:228-244 calls snappy.Encode / snappy.Decode   */
/* of github.com/klauspost/compress v1.18.2 (go.mod:6), NOT in the reference  */
/* tree.  Restated from the published Snappy block format                     */
/* (format_description.txt) and the published greedy block encoder of the Go  */
/* snappy package (64 KiB blocks, 14-bit hash of 4 bytes, skip heuristic,     */
/* emitLiteral / emitCopy): compressed bytes are PARITY UNPINNED; what is     */
/* pinned is the format (cross-checked against libsnappy 1.1.8 in tests).     */
/* Not restated: S2's repeat-offset extension (copy with 1-byte offset 0).    */
/* ------------------------------------------------------------------------- */

size_t ob_snappy_bound(size_t n) { return 32 + n + n / 6; }      /* snappy.MaxEncodedLen */

static size_t sn_put_uvarint(uint8_t *d, uint64_t v) { size_t i = 0; while (v >= 128) { d[i++] = (uint8_t)(v | 128); v >>= 7; } d[i++] = (uint8_t)v; return i; }

static size_t sn_emit_literal(uint8_t *d, const uint8_t *lit, size_t n) {
    size_t i = 0; const size_t x = n - 1;
    if (x < 60) d[i++] = (uint8_t)(x << 2);
    else if (x < 256) { d[i++] = 60 << 2; d[i++] = (uint8_t)x; }
    else if (x < 65536) { d[i++] = 61 << 2; d[i++] = (uint8_t)x; d[i++] = (uint8_t)(x >> 8); }
    else if (x < 16777216) { d[i++] = 62 << 2; d[i++] = (uint8_t)x; d[i++] = (uint8_t)(x >> 8); d[i++] = (uint8_t)(x >> 16); }
    else { d[i++] = 63 << 2; d[i++] = (uint8_t)x; d[i++] = (uint8_t)(x >> 8); d[i++] = (uint8_t)(x >> 16); d[i++] = (uint8_t)(x >> 24); }
    memcpy(d + i, lit, n);
    return i + n;
}
static size_t sn_emit_copy(uint8_t *d, size_t offset, size_t length) {
    size_t i = 0;
    while (length >= 68) { d[i++] = 63 << 2 | 2; d[i++] = (uint8_t)offset; d[i++] = (uint8_t)(offset >> 8); length -= 64; }
    if (length > 64) { d[i++] = 59 << 2 | 2; d[i++] = (uint8_t)offset; d[i++] = (uint8_t)(offset >> 8); length -= 60; }
    if (length >= 12 || offset >= 2048) { d[i++] = (uint8_t)((length - 1) << 2 | 2); d[i++] = (uint8_t)offset; d[i++] = (uint8_t)(offset >> 8); }
    else { d[i++] = (uint8_t)((offset >> 8) << 5 | (length - 4) << 2 | 1); d[i++] = (uint8_t)offset; }
    return i;
}
/* encodeBlock of the Go snappy package on one block of <= 65536 bytes (offsets fit 16 bits) */
static size_t sn_encode_block(uint8_t *dst, const uint8_t *src, size_t n) {
    enum { TBITS = 14 };
    uint16_t table[1 << TBITS];
    memset(table, 0, sizeof table);
    size_t d = 0;
    if (n < 17) return sn_emit_literal(dst, src, n);              /* minNonLiteralBlockSize */
    const size_t slimit = n - 15;                                 /* inputMargin */
    size_t lit = 0, s = 1;
    uint32_t h = (rd32(src + s) * 0x1e35a7bdu) >> (32 - TBITS);
    for (;;) {
        size_t skip = 32, next_s = s, cand = 0;
        for (;;) {
            s = next_s;
            const size_t step = skip >> 5;
            next_s = s + step; skip += step;
            if (next_s > slimit) goto remainder;
            cand = table[h];
            table[h] = (uint16_t)s;
            h = (rd32(src + next_s) * 0x1e35a7bdu) >> (32 - TBITS);
            if (rd32(src + s) == rd32(src + cand)) break;
        }
        d += sn_emit_literal(dst + d, src + lit, s - lit);
        for (;;) {
            const size_t base = s;
            s += 4;
            size_t i = cand + 4;
            while (s < n && src[i] == src[s]) { i++; s++; }
            d += sn_emit_copy(dst + d, base - cand, s - base);
            lit = s;
            if (s >= slimit) goto remainder;
            const uint64_t x = rd64(src + s - 1);
            const uint32_t hp = ((uint32_t)x * 0x1e35a7bdu) >> (32 - TBITS);
            table[hp] = (uint16_t)(s - 1);
            const uint32_t hc = ((uint32_t)(x >> 8) * 0x1e35a7bdu) >> (32 - TBITS);
            cand = table[hc];
            table[hc] = (uint16_t)s;
            if ((uint32_t)(x >> 8) != rd32(src + cand)) { h = ((uint32_t)(x >> 16) * 0x1e35a7bdu) >> (32 - TBITS); s++; break; }
        }
    }
remainder:
    if (lit < n) d += sn_emit_literal(dst + d, src + lit, n - lit);
    return d;
}
/* snappy.Encode(nil, src): uvarint(len) + the blocks' elements */
int64_t ob_snappy_compress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap) {
    if (cap < ob_snappy_bound(n)) return OB_ERR_SHORT_BUFFER;
    size_t d = sn_put_uvarint(dst, n);
    for (size_t off = 0; off < n; off += 65536) {
        const size_t len = n - off < 65536 ? n - off : 65536;
        d += sn_encode_block(dst + d, src + off, len);
    }
    return (int64_t)d;
}
/* snappy.Decode(buf[:cap], src): *declared = the uvarint length; returns the decoded byte count (== *declared) or an error.
 * A declared length above cap cannot be decoded into the caller's buffer: OB_ERR_SHORT_BUFFER (the frame layer reports it as
 * ErrSizeMismatch: the reference would decode into its own buffer and then fail the NBytesOrig comparison, blosc.go:429-431). */
int64_t ob_snappy_decompress(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, uint64_t *declared) {
    uint64_t dlen = 0; size_t s = 0; int okv = 0;
    for (unsigned i = 0; i < 10 && i < n; i++) {
        const unsigned b = src[i];
        dlen |= (uint64_t)(b & 127) << (7 * i);
        if (!(b & 128)) { s = i + 1; okv = !(i == 9 && b > 1); break; }
    }
    if (!okv || dlen > 0xFFFFFFFFull) return OB_ERR_DECOMPRESSION_FAILED;
    if (declared) *declared = dlen;
    if (dlen > cap) return OB_ERR_SHORT_BUFFER;
    size_t d = 0;
    while (s < n) {
        const unsigned t = src[s];
        size_t length, offset;
        if ((t & 3) == 0) {
            size_t x = t >> 2;
            if (x < 60) s += 1;
            else {
                const size_t nb = x - 59;
                if (n - s < 1 + nb) return OB_ERR_DECOMPRESSION_FAILED;
                x = 0;
                for (size_t i = 0; i < nb; i++) x |= (size_t)src[s + 1 + i] << (8 * i);
                s += 1 + nb;
            }
            length = x + 1;
            if (length > dlen - d || length > n - s) return OB_ERR_DECOMPRESSION_FAILED;
            memcpy(dst + d, src + s, length);
            d += length; s += length;
            continue;
        }
        if ((t & 3) == 1) { if (n - s < 2) return OB_ERR_DECOMPRESSION_FAILED; length = 4 + ((t >> 2) & 7); offset = ((size_t)(t & 0xe0) << 3) | src[s + 1]; s += 2; }
        else if ((t & 3) == 2) { if (n - s < 3) return OB_ERR_DECOMPRESSION_FAILED; length = 1 + (t >> 2); offset = src[s + 1] | ((size_t)src[s + 2] << 8); s += 3; }
        else { if (n - s < 5) return OB_ERR_DECOMPRESSION_FAILED; length = 1 + (t >> 2);
               offset = src[s + 1] | ((size_t)src[s + 2] << 8) | ((size_t)src[s + 3] << 16) | ((size_t)src[s + 4] << 24); s += 5; }
        if (offset == 0 || d < offset || length > dlen - d) return OB_ERR_DECOMPRESSION_FAILED;
        for (size_t k = 0; k < length; k++) dst[d + k] = dst[d - offset + k];
        d += length;
    }
    if (d != dlen) return OB_ERR_DECOMPRESSION_FAILED;
    return (int64_t)d;
}

/* ------------------------------------------------------------------------- */
/* frame layer                                                               */
/* ------------------------------------------------------------------------- */

static inline uint32_t le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
static inline void put32(uint8_t *p, uint32_t v) {
    p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24);
}

/* blosc.go:165-185 */
int ob_parse_header(const uint8_t *f, size_t n, ob_header *h) {
    if (n < OB_HEADER_SIZE) return OB_ERR_INVALID_HEADER;        /* :166-168 */
    h->version = f[0]; h->codec = f[1]; h->flags = f[2]; h->typesize = f[3];
    h->nbytes = le32(f + 4); h->blocksize = le32(f + 8); h->cbytes = le32(f + 12);
    if (h->version != 2) return OB_ERR_INVALID_VERSION;          /* :180-182 */
    return OB_OK;
}

/* blosc.go:188-198 */
void ob_header_bytes(const ob_header *h, uint8_t out[16]) {
    out[0] = h->version; out[1] = h->codec; out[2] = h->flags; out[3] = h->typesize;
    put32(out + 4, h->nbytes); put32(out + 8, h->blocksize); put32(out + 12, h->cbytes);
}

size_t ob_frame_bound(size_t n) { return OB_HEADER_SIZE + (ob_snappy_bound(n) > ob_lz4_bound(n) ? ob_snappy_bound(n) : ob_lz4_bound(n)); }

static int codec_known(int codec) {       /* the registry of codec.go:27-33; LZ4 family and Snappy restated */
    return codec == OB_LZ4 || codec == OB_LZ4HC || codec == OB_SNAPPY;
}

/* CompressWithOptions + compressBackend, blosc.go:268-374.
 * Codecs other than LZ4 are outside the hot path: INVALID_CODEC here.
 * LZ4HC (codec.go:94-118) is restated only on the decode side. */
int64_t ob_compress_frame(const uint8_t *src, size_t n, uint8_t *dst, size_t cap,
                          int codec, int level, int shuffle, int typesize, unsigned policy) {
    if (n == 0) return OB_ERR_INVALID_DATA;                      /* :269-271 */
    if (typesize <= 0) typesize = 1;                             /* :274-276 */
    if (level < 1) level = 1;                                    /* :277-279 */
    if (level > 9) level = 9;                                    /* :280-282 */
    (void)level;                                                 /* LZ4 ignores it, codec.go:63-66 */
    /* LZ4HC (codec.go:94-118): any valid LZ4 block is a valid payload; the greedy encoder stands in for CompressBlockHC */
    if (codec != OB_LZ4 && codec != OB_LZ4HC && codec != OB_SNAPPY) return OB_ERR_INVALID_CODEC;   /* :322-325 */
    if (n > 0xFFFFFFFFu - OB_HEADER_SIZE - n / 6 - 64) return OB_ERR_DATA_TOO_LARGE; /* Appendix D */
    if (cap < ob_frame_bound(n)) return OB_ERR_SHORT_BUFFER;

    uint8_t *filtered = NULL;
    const uint8_t *in = src;                                     /* :328 */
    if ((shuffle == OB_SHUFFLE || shuffle == OB_BITSHUFFLE) && typesize > 1) {   /* :329-333 */
        filtered = (uint8_t *)malloc(n);
        if (!filtered) return OB_ERR_COMPRESSION_FAILED;
        if (shuffle == OB_SHUFFLE) ob_shuffle(filtered, src, n, typesize);
        else ob_bitshuffle(filtered, src, n, typesize);
        in = filtered;
    }
    int64_t c = codec == OB_SNAPPY ? ob_snappy_compress(in, n, dst + OB_HEADER_SIZE, cap - OB_HEADER_SIZE)    /* :336 -> codec.go:232-235 */
                                   : ob_lz4_compress(in, n, dst + OB_HEADER_SIZE, cap - OB_HEADER_SIZE);       /* :336 -> codec.go:63-75 */
    if (c < 0) { free(filtered); return OB_ERR_COMPRESSION_FAILED; }
    const int use_memcpy = (size_t)c >= n;                       /* :342 */
    if (use_memcpy) {
        /* :343-345 stores the UN-filtered input while keeping the filter flags, which the
         * reference's own decoder then un-filters (:398-426) -> corrupt round trip.
         * Default here (and in the product): store the FILTERED bytes so the reference
         * Decompress reproduces the input.  OB_POLICY_REFERENCE_MEMCPY restates :344 exactly. */
        const uint8_t *payload = (policy & OB_POLICY_REFERENCE_MEMCPY) ? src : in;
        memcpy(dst + OB_HEADER_SIZE, payload, n);
        c = (int64_t)n;
    }
    uint8_t flags = 0;                                           /* :348-356 */
    if (shuffle == OB_SHUFFLE) flags |= OB_FLAG_SHUFFLE;
    else if (shuffle == OB_BITSHUFFLE) flags |= OB_FLAG_BITSHUFFLE;
    if (use_memcpy) flags |= OB_FLAG_MEMCPY;
    ob_header h = { 2, (uint8_t)codec, flags, (uint8_t)typesize,                /* :358-366 */
                    (uint32_t)n, (uint32_t)n, (uint32_t)(OB_HEADER_SIZE + c) };
    ob_header_bytes(&h, dst);
    free(filtered);
    return OB_HEADER_SIZE + c;
}

/* DecompressWithSize + decompressBackend, blosc.go:296-303 and :377-434.  Bug-compatible:
 * memcpy payloads are un-filtered too (:398-400 then :422-426), bitshuffle flag wins (:422). */
int64_t ob_decompress_frame(const uint8_t *f, size_t n, uint8_t *dst, size_t cap, int ts_override) {
    ob_header h;
    if (n < OB_HEADER_SIZE) return OB_ERR_INVALID_HEADER;        /* :297-299 */
    int rc = ob_parse_header(f, n, &h);                          /* :379-382 */
    if (rc) return rc;
    if ((size_t)h.cbytes > n) return OB_ERR_INVALID_DATA;        /* :385-387 */
    if (h.cbytes < OB_HEADER_SIZE) return OB_ERR_INVALID_DATA;   /* :388-390 */
    const uint8_t *payload = f + OB_HEADER_SIZE;                 /* :393 */
    const size_t plen = h.cbytes - OB_HEADER_SIZE;
    uint8_t *tmp = (uint8_t *)malloc(h.nbytes > plen ? (size_t)h.nbytes + 1 : plen + 1);
    if (!tmp) return OB_ERR_DECOMPRESSION_FAILED;
    int64_t got;
    if (h.flags & OB_FLAG_MEMCPY) {                              /* :398-400 */
        memcpy(tmp, payload, plen); got = (int64_t)plen;
    } else {
        if (!codec_known(h.codec)) { free(tmp); return OB_ERR_INVALID_CODEC; }   /* :403-407 */
        if (h.codec == OB_SNAPPY) {                              /* :410 -> codec.go:237-244 */
            got = ob_snappy_decompress(payload, plen, tmp, h.nbytes, NULL);
            if (got == OB_ERR_SHORT_BUFFER) { free(tmp); return OB_ERR_SIZE_MISMATCH; }   /* declared > NBytesOrig, see ob_snappy_decompress */
        } else got = ob_lz4_decompress(payload, plen, tmp, h.nbytes);   /* :410 -> codec.go:77-84 */
        if (got < 0) { free(tmp); return OB_ERR_DECOMPRESSION_FAILED; }           /* :411-413 */
    }
    int ts = ts_override > 0 ? ts_override : (int)h.typesize;    /* :417-419 */
    if ((size_t)got != (size_t)h.nbytes) { free(tmp); return OB_ERR_SIZE_MISMATCH; } /* :429-431 (filters keep length) */
    if ((size_t)got > cap) { free(tmp); return OB_ERR_SHORT_BUFFER; }
    if ((h.flags & OB_FLAG_BITSHUFFLE) && ts > 1) ob_bitunshuffle(dst, tmp, (size_t)got, ts);   /* :422-423 */
    else if ((h.flags & OB_FLAG_SHUFFLE) && ts > 1) ob_unshuffle(dst, tmp, (size_t)got, ts);   /* :424-425 */
    else memcpy(dst, tmp, (size_t)got);
    free(tmp);
    return got;
}

/* ------------------------------------------------------------------------- */
/* synthetic workloads, SURVEY.md §8(d)                                      */
/* ------------------------------------------------------------------------- */

static inline uint64_t synth_h(uint64_t i, uint64_t frame) {
    uint64_t z = i + 0x9E3779B97F4A7C15ULL * (frame + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static inline uint64_t synth_tri(uint64_t i) { uint64_t t = i & 8191; return t < 4096 ? t : 8192 - t; }

size_t ob_synth(int kind, uint64_t frame, uint64_t first, uint64_t count, void *out) {
    switch (kind) {
    case OB_D_F32: { float *o = (float *)out;
        for (uint64_t k = 0; k < count; k++) { uint64_t i = first + k;
            o[k] = (float)synth_tri(i) * 0.25f + (float)(synth_h(i, frame) & 0xFF) * (1.0f / 256.0f); }
        return count * 4; }
    case OB_D_F64: { double *o = (double *)out;
        for (uint64_t k = 0; k < count; k++) { uint64_t i = first + k;
            o[k] = (double)synth_tri(i) * 0.25 + (double)(synth_h(i, frame) & 0xFF) * (1.0 / 256.0); }
        return count * 8; }
    case OB_D_I32: { int32_t *o = (int32_t *)out;
        for (uint64_t k = 0; k < count; k++) o[k] = (int32_t)(synth_h(first + k, frame) % 65536);
        return count * 4; }
    case OB_D_RAMP: { float *o = (float *)out;          /* blosc_test.go:109-111 pattern */
        for (uint64_t k = 0; k < count; k++) o[k] = (float)(first + k) * 0.1f;
        return count * 4; }
    case OB_D_RAND: { float *o = (float *)out;
        for (uint64_t k = 0; k < count; k++) o[k] = (float)(synth_h(first + k, frame) >> 40) * (1.0f / 16777216.0f);
        return count * 4; }
    case OB_D_BYTES256: { uint8_t *o = (uint8_t *)out;  /* blosc_test.go:365-368 pattern */
        for (uint64_t k = 0; k < count; k++) o[k] = (uint8_t)((first + k) % 256);
        return count; }
    }
    return 0;
}
