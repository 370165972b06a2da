/* lane_model.c — CPU model of the lane-serial device LZ4 matcher (go-blosc_amd/csrc/hb_lz4_enc.hip match_chunk_ls): the 64
 * lanes of a wavefront each parse their own 64-byte segment of a 4 KiB chunk serially, in lockstep, sharing one hash table
 * and the whole chunk image as match source.  Used offline to explore the parse (ratio, sequences, lockstep iterations);
 * it emits no bytes.  Test tooling, not product.
 *   gcc -O2 -o lane_model lane_model.c && ./lane_model <file> [hlog=9] [skipshift=2] [flags=3] [seg=64] [longcap=20]
 * flags: bit0 = offset-1 (run) probe when the table candidate fails, bit1 = runs are not inserted into the table,
 *        bit2 = backward extension, bit3 = offset-1 preferred over the table candidate,
 *        bit4 = table keeps the LOWEST position per slot (atomic min) instead of the last writer,
 *        bit5 = two slots: lowest position + last writer (last writer tried first),
 *        bit6 = with bit5: verify ONE candidate only (the last writer if it lies before p, else the lowest position)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static uint32_t rd4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static uint32_t ext(uint32_t x) { return x < 15 ? 0 : 1 + (x - 15) / 255; }
#define CHUNK 4096
#define MAXL 128
#define MAXE 32

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(n + 64); memset(buf + n, 0, 64);
    if (fread(buf, 1, n, f) != (size_t)n) return 1;
    const int hlog = argc > 2 ? atoi(argv[2]) : 9, skipshift = argc > 3 ? atoi(argv[3]) : 2, flags = argc > 4 ? atoi(argv[4]) : 3;
    const int seg = argc > 5 ? atoi(argv[5]) : 64, longcap = argc > 6 ? atoi(argv[6]) : 20;
    const int nl = CHUNK / seg;
    uint16_t *tab = malloc(sizeof(uint16_t) << hlog);
    uint16_t *tmin = malloc(sizeof(uint16_t) << hlog);
    uint64_t out = 0, nseq = 0, iters = 0, emit_iters = 0, carry = 0, coop = 0, dropped = 0, trimmed = 0, hititers = 0;
    for (long start = 0; start < n; start += CHUNK) {
        const uint8_t *d = buf + start; const int len = n - start < CHUNK ? (int)(n - start) : CHUNK;
        memset(tab, (flags & 16) ? 0xFF : 0, sizeof(uint16_t) << hlog);
        memset(tmin, 0xFF, sizeof(uint16_t) << hlog);
        const int ms = len - 12, me = len - 5;
        int p[MAXL], miss[MAXL], ne[MAXL], send[MAXL];
        int ep[MAXL][MAXE], el[MAXL][MAXE], eo[MAXL][MAXE];
        for (int l = 0; l < nl; l++) { p[l] = l * seg; send[l] = (l + 1) * seg < len ? (l + 1) * seg : len; miss[l] = 0; ne[l] = 0; }
        for (;;) {
            int act[MAXL], any = 0, cand[MAXL], cand2[MAXL];
            for (int l = 0; l < nl; l++) { act[l] = p[l] < send[l] && p[l] <= ms; any |= act[l]; }
            if (!any) break;
            iters++;
            for (int l = 0; l < nl; l++) if (act[l]) { const uint32_t h = (rd4(d + p[l]) * 2654435761u) >> (32 - hlog); cand[l] = tab[h]; cand2[l] = tmin[h]; }
            for (int l = 0; l < nl; l++) if (act[l]) {               /* highest lane wins */
                const uint32_t v = rd4(d + p[l]);
                if ((flags & 2) && p[l] >= 1 && rd4(d + p[l] - 1) == v) continue;
                const uint32_t h = (v * 2654435761u) >> (32 - hlog);
                if (flags & 16) { if (p[l] < tab[h]) tab[h] = (uint16_t)p[l]; } else tab[h] = (uint16_t)p[l];
                if (p[l] < tmin[h]) tmin[h] = (uint16_t)p[l]; }
            int anyhit = 0;
            for (int l = 0; l < nl; l++) if (act[l]) {
                const int q = p[l]; const uint32_t v = rd4(d + q);
                int c = -1;
                const int rle = (flags & 1) && q >= 1 && rd4(d + q - 1) == v;
                if ((flags & 8) && rle) c = q - 1;
                else if ((flags & 64) && cand[l] >= q) { if (cand2[l] < q && rd4(d + cand2[l]) == v) c = cand2[l]; else if (rle) c = q - 1; }
                else if (cand[l] < q && rd4(d + cand[l]) == v) c = cand[l];
                else if ((flags & 32) && !(flags & 64) && cand2[l] < q && rd4(d + cand2[l]) == v) c = cand2[l];
                else if (rle) c = q - 1;
                if (c < 0) { p[l] = q + 1 + (miss[l] >> skipshift); miss[l]++; continue; }
                int mp = q, mc = c;
                if (flags & 4) { const int lo = ne[l] ? ep[l][ne[l] - 1] + el[l][ne[l] - 1] : l * seg;
                    while (mp > lo && mc > 0 && d[mp - 1] == d[mc - 1]) { mp--; mc--; } }
                int ml = 0; while (mp + ml < me && d[mp + ml] == d[mc + ml]) ml++;
                if (ml < 4) { p[l] = q + 1; continue; }
                if (ml > longcap) coop++;
                anyhit = 1;
                ep[l][ne[l]] = mp; el[l][ne[l]] = ml; eo[l][ne[l]] = mp - mc; ne[l]++;
                p[l] = mp + ml; miss[l] = 0;
            }
            hititers += anyhit;
        }
        /* resolution: matches of earlier lanes win; later lanes drop / trim what they cover */
        int cover = 0, anchor = 0, first = 1, maxkept = 0;
        for (int l = 0; l < nl; l++) {
            int kept = 0, lastend = 0;
            for (int i = 0; i < ne[l]; i++) {
                int mp = ep[l][i], ml = el[l][i];
                lastend = mp + ml;
                if (mp + ml <= cover) { dropped++; continue; }
                if (mp < cover) { ml -= cover - mp; mp = cover; if (ml < 4) { dropped++; continue; } trimmed++; }
                const uint32_t lit = mp - anchor;
                if (first) { out += 1 + ext(lit + carry) + lit + carry; carry = 0; first = 0; } else out += 1 + ext(lit) + lit;
                out += 2 + ext(ml - 4); nseq++; kept++;
                anchor = mp + ml;
            }
            if (lastend > cover) cover = lastend;
            if (kept > maxkept) maxkept = kept;
        }
        emit_iters += maxkept;
        carry += len - anchor;
    }
    out += 1 + ext(carry) + carry;
    const double kib = n / 1024.0;
    printf("n=%ld out=%llu ratio=%.4f seqs=%llu iters/chunk=%.1f (hit iters %.1f) emit_iters/chunk=%.1f long=%llu dropped=%llu trimmed=%llu\n", n,
           (unsigned long long)out, (double)out / n, (unsigned long long)nseq, iters / (kib / 4), hititers / (kib / 4), emit_iters / (kib / 4),
           (unsigned long long)coop, (unsigned long long)dropped, (unsigned long long)trimmed);
    return 0;
}
