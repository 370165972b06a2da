/* window_model.c — what the reference encoder's WINDOW and its HASH KEY are worth on the headline data, separately.
 * Test tooling (ratio only, emits no bytes); the numbers in DESIGN.md 5.3 come from here.
 *
 * A plain serial greedy LZ4 parse (one candidate per position = the latest position with the same hash, verified on 4 bytes,
 * backward extension, no skip acceleration) is run over each byte plane of the shuffled D-f32 data with
 *   - matches confined to chunks of 4 KiB ... 64 KiB (table reset per chunk), or one 64 KiB sliding window over the whole plane,
 *   - the table keyed by 4, 5 or 6 bytes.
 * Last line: the restated reference encoder (oracle/blosc_oracle.c ob_lz4_compress: 64 KiB window, 6-byte key) on the same
 * planes as one block each, and on independent 4 KiB chunks.
 *
 *   gcc -O2 -o window_model tests/tools/window_model.c oracle/blosc_oracle.c && ./window_model [elements, default 16 Mi]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "../../oracle/blosc_oracle.h"

static uint32_t rd4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint64_t rd8(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }
static uint32_t ext(uint32_t x) { return x < 15 ? 0 : 1 + (x - 15) / 255; }

static uint32_t key_hash(const uint8_t *p, int keybytes, int hlog) {
    if (keybytes == 4) return (rd4(p) * 2654435761u) >> (32 - hlog);
    return (uint32_t)(((rd8(p) << (64 - 8 * keybytes)) * 227718039650203ULL) >> (64 - hlog));
}

/* bytes of the LZ4 block a greedy parse of d[0, n) would produce; chunk = 0: one sliding 64 KiB window */
static uint64_t parse(const uint8_t *d, long n, long chunk, int keybytes, uint64_t *nseq) {
    const int hlog = chunk == 0 || chunk >= 65536 ? 16 : chunk >= 16384 ? 14 : chunk >= 8192 ? 13 : 12;
    int32_t *tab = malloc(sizeof(int32_t) << hlog);
    uint64_t out = 0;
    long anchor = 0;
    *nseq = 0;
    const long step = chunk ? chunk : n;
    for (long start = 0; start < n; start += step) {
        const long end = start + step < n ? start + step : n;
        for (long i = 0; i < (1L << hlog); i++) tab[i] = -1;
        const long ms = end - 12, me = end - 5;
        long p = start;
        while (p <= ms) {
            const uint32_t h = key_hash(d + p, keybytes, hlog);
            long c = tab[h];
            tab[h] = (int32_t)p;
            if (c < 0 || p - c > 65535 || rd4(d + c) != rd4(d + p)) { p++; continue; }
            long ml = 4;
            while (p > anchor && p > start && c > start && d[p - 1] == d[c - 1]) { p--; c--; ml++; }
            while (p + ml < me && d[p + ml] == d[c + ml]) ml++;
            const long lit = p - anchor;
            out += 1 + ext((uint32_t)lit) + lit + 2 + ext((uint32_t)(ml - 4));
            (*nseq)++;
            p += ml; anchor = p;
            if (p - 2 >= start && p + 6 <= end) tab[key_hash(d + p - 2, keybytes, hlog)] = (int32_t)(p - 2);
        }
    }
    const long lit = n - anchor;
    out += 1 + ext((uint32_t)lit) + lit;
    free(tab);
    return out;
}

int main(int argc, char **argv) {
    const long ne = argc > 1 ? atol(argv[1]) : (16L << 20);
    float *x = malloc(ne * 4);
    ob_synth(0, 0, 0, ne, x);                                     /* D-f32, the headline data set */
    uint8_t *s = malloc(ne * 4 + 64);
    memset(s + ne * 4, 0, 64);
    ob_shuffle(s, (const uint8_t *)x, ne * 4, 4);
    const long chunks[] = {4096, 8192, 16384, 32768, 65536, 0};
    const int keys[] = {4, 5, 6};
    printf("%-22s %-5s %8s %8s %8s %8s %8s %14s\n", "matches confined to", "key", "plane 0", "plane 1", "plane 2", "plane 3", "frame", "sequences/GiB");
    for (unsigned ci = 0; ci < sizeof chunks / sizeof chunks[0]; ci++)
        for (unsigned ki = 0; ki < 3; ki++) {
            uint64_t tot = 0, seqs = 0;
            double r[4];
            for (int j = 0; j < 4; j++) {
                uint64_t ns;
                const uint64_t o = parse(s + j * ne, ne, chunks[ci], keys[ki], &ns);
                tot += o; seqs += ns; r[j] = (double)o / ne;
            }
            char name[32];
            if (chunks[ci]) snprintf(name, sizeof name, "%ld KiB chunks", chunks[ci] >> 10); else snprintf(name, sizeof name, "64 KiB sliding window");
            printf("%-22s %-5d %8.4f %8.4f %8.4f %8.4f %8.4f %13.1fM\n", name, keys[ki], r[0], r[1], r[2], r[3], (double)tot / (ne * 4),
                   seqs * (double)(1L << 28) / ne / 1e6);
        }
    uint8_t *c = malloc(ob_lz4_bound(ne));
    uint64_t one = 0, per4k = 0;
    for (int j = 0; j < 4; j++) one += (uint64_t)ob_lz4_compress(s + j * ne, ne, c, ob_lz4_bound(ne));
    for (long o = 0; o < ne * 4; o += 4096) per4k += (uint64_t)ob_lz4_compress(s + o, 4096, c, ob_lz4_bound(ne));
    printf("restated reference encoder: one block per plane %.4f, independent 4 KiB chunks %.4f\n", (double)one / (ne * 4), (double)per4k / (ne * 4));
    return 0;
}
