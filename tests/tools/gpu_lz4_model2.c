/* gpu_lz4_model2.c — CPU model of the device LZ4 matcher's PARSE (go-blosc_amd/csrc/hb_lz4_enc.hip match_chunk), step by step as the
 * wavefront takes it, with the policies round 4 asks about as switches.  Offline tooling (ratio, sequence and step counts; no bytes).
 *   gcc -O2 -o gpu_lz4_model2 gpu_lz4_model2.c && ./gpu_lz4_model2 <file> [key=5] [hlog=8] [gate=0] [minlen=4] [lanes=64] [flags=0]
 *   key    bytes hashed into the table key (4 / 5 / 6; 7 = five bytes through the device's 24-bit-multiply hash, hlog 8 only)
 *   gate   run gate: a step whose window has >= gate positions equal to the byte before takes ONLY the runs (offset 1, length >= minlen)
 *          it sees in the wave-wide mask: no hash, no table probe, no insert (0 = off)
 *   lanes  positions per step (64 = one per lane, 128 = two per lane: inserts of the first half are not seen by the second)
 *   flags  bit0: steps under the gate still insert their non-run positions
 *          bit1: adaptive gate -- a run step is only taken for the `aux` steps after a FULL hit step whose greedy parse needed no fewer
 *                sequences than the window has runs (nrun <= nsel + 1); periodic patterns of short runs (a ramp's low mantissa plane), where
 *                one table match spans many runs, then never open the gate
 *          bit2: a run step inserts the first byte of every run it selects
 *   aux    steps a full step's verdict holds (default 7)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static uint32_t rd4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int g_key = 5, g_hlog = 8;
static uint32_t HS(const uint8_t *p) {
    const uint32_t v = rd4(p);
    uint32_t x = v;
    if (g_key == 7) {   /* the device's round-4 hash of five bytes: two 24-bit multiplies (bytes 0..2, bytes 2..4), top byte of the sum */
        const uint32_t z = (uint32_t)p[2] | ((uint32_t)p[3] << 8) | ((uint32_t)p[4] << 16);
        static uint32_t K1 = 0, K2 = 0;
        if (!K1) { K1 = getenv("HK1") ? strtoul(getenv("HK1"), 0, 16) : 0xF85117u; K2 = getenv("HK2") ? strtoul(getenv("HK2"), 0, 16) : 0xE01E5Bu; }
        return (((v & 0xFFFFFFu) * K1 + z * K2) & 0xFFFFFFFFu) >> (32 - g_hlog);
    }
    if (g_key == 5) x = v + (uint32_t)p[4] * 0x50505u;
    else if (g_key == 6) x = v + ((uint32_t)p[4] | ((uint32_t)p[5] << 8)) * 0x50505u;
    return (x * 2246822519u) >> (32 - g_hlog);
}
static uint32_t ext(uint32_t x) { return x < 15 ? 0 : 1 + (x - 15) / 255; }

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(n + 256); memset(buf + n, 0, 256);
    if (fread(buf, 1, n, f) != (size_t)n) return 1;
    g_key = argc > 2 ? atoi(argv[2]) : 5; g_hlog = argc > 3 ? atoi(argv[3]) : 8;
    const int gate = argc > 4 ? atoi(argv[4]) : 0, minlen = argc > 5 ? atoi(argv[5]) : 4, W = argc > 6 ? atoi(argv[6]) : 64;
    const int flags = argc > 7 ? atoi(argv[7]) : 0, aux = argc > 8 ? atoi(argv[8]) : 7;
    const int chunk = 4096, accel = 64;
    const int stride2 = getenv("STRIDE2") ? atoi(getenv("STRIDE2")) : 0;   /* 1: full steps probe every other position (model of a 128-byte step with one position per lane) */
    uint16_t *tab = malloc(sizeof(uint16_t) << g_hlog);
    uint64_t out = 0, nseq = 0, steps = 0, hitsteps = 0, gatesteps = 0, carry = 0, hitlanes = 0, longs = 0;
    uint64_t mlhist[6] = {0};   /* match length: 4-7, 8-11, 12-19, 20-63, 64-255, 256+ */
    for (long start = 0; start < n; start += chunk) {
        const uint8_t *d = buf + start; const int len = n - start < chunk ? (int)(n - start) : chunk;
        memset(tab, 0, sizeof(uint16_t) << g_hlog);
        int pos = 0, anchor = 0, first = 1, miss = 0, gate_left = (flags & 2) ? 0 : 1 << 30, aux_cur = aux;
        const int ms = len - 12, me = len - 5;
        while (pos <= ms) {
            steps++;
            int cand[128], ml[128], ism[128];
            /* wave-wide "equals the byte before" count of the window */
            int neq = 0;
            for (int l = 0; l < W && pos + l < len; l++) if (pos + l >= 1 && d[pos + l] == d[pos + l - 1]) neq++;
            int nrle = 0;
            for (int l = 0; l < W; l++) { const int p = pos + l; if (p > ms || p < 1) continue; const uint32_t v = rd4(d + p);
                if (d[p - 1] == d[p] && v == (v & 255u) * 0x01010101u) nrle++; }
            int nstart = 0;   /* runs the window would select: groups of consecutive run lanes */
            { int prev = 0; for (int l = 0; l < W; l++) { const int p = pos + l; int r = 0; if (p <= ms && p >= 1) { const uint32_t v = rd4(d + p); r = d[p - 1] == d[p] && v == (v & 255u) * 0x01010101u; } if (r && !prev) nstart++; prev = r; } }
            const int maxruns = getenv("MAXRUNS") ? atoi(getenv("MAXRUNS")) : 64;
            const int gated = gate && neq >= gate && gate_left > 0 && nrle > 0 && nstart <= maxruns;
            if (gated) gate_left--;
            for (int half = 0; half < W; half += 64) {          /* table reads of 64 lanes, then their writes */
                for (int l = half; l < half + 64; l++) { const int p = pos + l; cand[l] = 0; ism[l] = 0; ml[l] = 0; if (p > ms) continue; if (!gated) cand[l] = tab[HS(d + p)]; }
                for (int l = half; l < half + 64; l++) { const int p = pos + l; if (p > ms) continue;
                    const uint32_t v = rd4(d + p);
                    const int rle = p >= 1 && d[p - 1] == d[p] && v == (v & 255u) * 0x01010101u;
                    if (rle) continue;
                    if (stride2 && !gated && (l & 1)) continue;
                    if (gated && !(flags & 1)) {
                        /* bit2: a run step still inserts the first byte of every run it selects (the position in front of the first run lane) */
                        if (!(flags & 4)) continue;
                        const uint32_t v1 = rd4(d + p + 1); const int r1 = d[p] == d[p + 1] && v1 == (v1 & 255u) * 0x01010101u && p + 1 <= ms;
                        if (!r1) continue;
                    }
                    tab[HS(d + p)] = (uint16_t)p; }
            }
            int any = 0;
            for (int l = 0; l < W; l++) { const int p = pos + l; if (p > ms) continue;
                if (stride2 && !gated && (l & 1)) continue;            /* STRIDE2: odd positions are neither probed nor run candidates */
                const uint32_t v = rd4(d + p);
                const int rle = p >= 1 && d[p - 1] == d[p] && v == (v & 255u) * 0x01010101u;
                int c = -1;
                if (!gated && cand[l] < p && rd4(d + cand[l]) == v) c = cand[l];
                else if (rle) c = p - 1;
                if (c < 0) continue;
                int k = 0; while (p + k < me && d[p + k] == d[c + k]) k++;
                if (k < minlen) continue;
                ism[l] = 1; cand[l] = c; ml[l] = k; any = 1; hitlanes++; }
            if (any) {
                hitsteps++; if (gated) gatesteps++;
                int l = 0, nsel = 0, nrun = 0;
                for (int q = 0; q < W; q++) { const int p = pos + q; if (p > ms || p < 1) continue; const uint32_t v = rd4(d + p);
                    const int r = d[p - 1] == d[p] && v == (v & 255u) * 0x01010101u;
                    int rprev = 0; if (q > 0) { const uint32_t v1 = rd4(d + p - 1); rprev = p >= 2 && d[p - 2] == d[p - 1] && v1 == (v1 & 255u) * 0x01010101u; }
                    if (r && !rprev) nrun++; }
                while (l < W) {
                    if (!ism[l]) { l++; continue; }
                    const int p = pos + l;
                    const uint32_t lit = p - anchor;
                    if (first) { out += 1 + ext(lit + carry) + lit + carry; carry = 0; first = 0; } else out += 1 + ext(lit) + lit;
                    out += 2 + ext(ml[l] - 4); nseq++;
                    if (ml[l] >= 20) longs++;
                    mlhist[ml[l] < 8 ? 0 : ml[l] < 12 ? 1 : ml[l] < 20 ? 2 : ml[l] < 64 ? 3 : ml[l] < 256 ? 4 : 5]++;
                    anchor = p + ml[l];
                    l += ml[l]; nsel++;
                }
                if ((flags & 2) && !gated) {
                    if (nrun <= nsel + 1) { gate_left = aux_cur; if (flags & 8) aux_cur = 2 * aux_cur + 1 > 15 ? 15 : 2 * aux_cur + 1; }   /* bit3: the verdict holds twice as long each time it is confirmed */
                    else { gate_left = 0; aux_cur = aux; }
                }
                miss = 0;
            } else miss++;
            const int nxt = pos + W + miss * accel;
            pos = anchor > nxt ? anchor : nxt;
        }
        carry += len - anchor;
    }
    out += 1 + ext(carry) + carry;
    printf("key=%d hlog=%d gate=%d min=%d W=%d: ratio=%.4f seqs/GiB=%.1fM steps/chunk=%.1f hit-steps=%.1f gated=%.1f hit-lanes/hit-step=%.1f  len 4-7:%.0f%% 8-11:%.0f%% 12-19:%.0f%% 20-63:%.0f%% 64+:%.0f%%\n",
           g_key, g_hlog, gate, minlen, W, (double)out / n, nseq * (1073741824.0 / n) / 1e6, steps / (n / 4096.0), hitsteps / (n / 4096.0), gatesteps / (n / 4096.0),
           hitsteps ? (double)hitlanes / hitsteps : 0.0, 100.0 * mlhist[0] / nseq, 100.0 * mlhist[1] / nseq, 100.0 * mlhist[2] / nseq, 100.0 * mlhist[3] / nseq, 100.0 * (mlhist[4] + mlhist[5]) / nseq);
    return 0;
}
