/* window_model2.c -- round 3: what a shared 8 / 16 / 32 KiB match window would buy at table sizes an LDS-resident matcher can afford.
 * Serial greedy LZ4 parse (latest occurrence per hash, 4-byte verify, optional backward extension) over the four byte planes of the
 * headline data, matches confined to chunks of 4-32 KiB, table 2^8 .. 2^14 entries, key 5 or 6 bytes.  DESIGN.md 5.5 quotes it.
 *   gcc -O2 -o wm2 tests/tools/window_model2.c oracle/blosc_oracle.c \&\& ./wm2 */
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
/* minlen: minimum accepted match length; backext: backward extension on/off */
static uint64_t parse(const uint8_t *d, long n, long chunk, int keybytes, int hlog, int backext, uint64_t *nseq) {
    int32_t *tab = malloc(sizeof(int32_t) << hlog);
    uint64_t out = 0; long anchor = 0; *nseq = 0;
    for (long start = 0; start < n; start += chunk) {
        const long end = start + chunk < n ? start + chunk : n;
        for (long i = 0; i < (1L << hlog); i++) tab[i] = -1;
        const long ms = end - 12, me = end - 5;
        long p = start;
        while (p <= ms) {
            const uint32_t h = key_hash(d + p, keybytes, hlog);
            long c = tab[h]; tab[h] = (int32_t)p;
            if (c < 0 || rd4(d + c) != rd4(d + p)) { p++; continue; }
            long ml = 4;
            if (backext) while (p > anchor && p > start && c > start && d[p - 1] == d[c - 1]) { p--; c--; ml++; }
            while (p + ml < me && d[p + ml] == d[c + ml]) ml++;
            const long lit = p - anchor;
            out += 1 + ext((uint32_t)lit) + lit + 2 + ext((uint32_t)(ml - 4));
            (*nseq)++; p += ml; anchor = p;
        }
    }
    const long lit = n - anchor; out += 1 + ext((uint32_t)lit) + lit; free(tab); return out;
}
int main(int argc, char **argv) {
    const long ne = 16L << 20;
    float *x = malloc(ne * 4); ob_synth(0, 0, 0, ne, x);
    uint8_t *s = malloc(ne * 4 + 64); memset(s + ne * 4, 0, 64);
    ob_shuffle(s, (const uint8_t *)x, ne * 4, 4);
    const long chunks[] = {4096, 8192, 16384, 32768};
    for (unsigned ci = 0; ci < 4; ci++) for (int key = 5; key <= 6; key++) for (int hlog = 8; hlog <= 14; hlog += 2) for (int be = 0; be < 2; be++) {
        uint64_t tot = 0, seqs = 0; double r[4];
        for (int j = 0; j < 4; j++) { uint64_t ns; uint64_t o = parse(s + j * ne, ne, chunks[ci], key, hlog, be, &ns); tot += o; seqs += ns; r[j] = (double)o / ne; }
        printf("chunk %6ld key %d hlog %2d backext %d : %.4f %.4f %.4f %.4f  frame %.4f  seq %.1fM\n", chunks[ci], key, hlog, be, r[0], r[1], r[2], r[3], (double)tot / (ne * 4), seqs * 16.0 / 1e6);
    }
    return 0;
}
