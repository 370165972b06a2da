/* gpu_lz4_model.c — CPU model of the device LZ4 matcher (go-blosc_amd/csrc/hb_lz4_enc.hip k_match), used to
 * explore parse heuristics offline (compression ratio only; it emits no bytes).  Test tooling, not product.
 *   gcc -O2 -o gpu_lz4_model gpu_lz4_model.c && ./gpu_lz4_model <file> [chunk] [hlog] [flags] [aux] [min]
 * flags: bit0 = probe offset 1, bit1 = probe offset `aux`, bit2 = backward extension, bit3 = skip acceleration,
 *        bit4 = positions whose 4 bytes equal those at p-1 (inside a run) are not inserted,
 *        bit5 = prefer the offset-1 candidate over the table candidate when both verify,
 *        bit6 = a position whose table candidate verifies is NOT inserted (the table keeps the oldest verified occurrence: match
 *               sources are then old data, which the decoder's dependency rounds like; prints the mean offset as a proxy),
 *        bit7 = like bit6, but only when the candidate verifies for >= 12 bytes (the "keep_long" policy of k_match)
 *        bit8 = the table is keyed by FIVE bytes, exactly as the device hashes them (hb_lz4_enc.hip): same ratio as the device to 3
 *               digits on the headline planes; against the 4-byte key: 0.5233 -> 0.5164 and 22 % fewer sequences at 256 entries
 * [min] = shortest match taken (default 4)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

static uint32_t rd4(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int g_flags, g_hlog, g_min = 4;
static uint32_t HS(const uint8_t *p) {
    const uint32_t v = rd4(p);
    if (g_flags & 256) return ((v + (uint32_t)p[4] * 0x50505u) * 2246822519u) >> (32 - g_hlog);
    return (v * 2654435761u) >> (32 - g_hlog);
}
static uint32_t ext(uint32_t x) { return x < 15 ? 0 : 1 + (x - 15) / 255; }

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *buf = malloc(n + 64); memset(buf + n, 0, 64);
    if (fread(buf, 1, n, f) != (size_t)n) return 1;
    int chunk = argc > 2 ? atoi(argv[2]) : 4096, hlog = argc > 3 ? atoi(argv[3]) : 11, flags = argc > 4 ? atoi(argv[4]) : 0;
    int aux = argc > 5 ? atoi(argv[5]) : 4; g_flags = flags; g_hlog = hlog; if (argc > 6) g_min = atoi(argv[6]);
    uint16_t *tab = malloc(sizeof(uint16_t) << hlog);
    uint64_t out = 0, nseq = 0, steps = 0; uint64_t carry = 0; double offsum = 0; uint64_t near = 0;
    for (long start = 0; start < n; start += chunk) {
        const uint8_t *d = buf + start; int len = n - start < chunk ? (int)(n - start) : chunk;
        memset(tab, 0, sizeof(uint16_t) << hlog);
        int pos = 0, anchor = 0, ms = len - 12, me = len - 5, first = 1, miss = 0;
        while (pos <= ms) {
            steps++;
            int cand[64]; int ism[64];
            for (int l = 0; l < 64; l++) { int p = pos + l; cand[l] = 0; ism[l] = 0; if (p > ms) continue;
                cand[l] = tab[HS(d + p)]; }
            for (int l = 0; l < 64; l++) { int p = pos + l; if (p > ms) continue; uint32_t v = rd4(d + p);
                if ((flags & 16) && p >= 1 && rd4(d + p - 1) == v) continue;
                if ((flags & 64) && cand[l] < p && rd4(d + cand[l]) == v) continue;
                if ((flags & 128) && cand[l] < p && memcmp(d + cand[l], d + p, 12) == 0) continue;
                tab[HS(d + p)] = (uint16_t)p; }   /* highest lane wins */
            for (int l = 0; l < 64; l++) { int p = pos + l; if (p > ms) continue; uint32_t v = rd4(d + p);
                int rle = (flags & 1) && p >= 1 && rd4(d + p - 1) == v;
                if ((flags & 32) && rle) { ism[l] = 1; cand[l] = p - 1; }
                else if (cand[l] < p && rd4(d + cand[l]) == v) ism[l] = 1;
                else if ((flags & 1) && p >= 1 && rd4(d + p - 1) == v) { ism[l] = 1; cand[l] = p - 1; }
                else if ((flags & 2) && p >= aux && rd4(d + p - aux) == v) { ism[l] = 1; cand[l] = p - aux; } }
            int any = 0;
            for (int l = 0; l < 64; l++) { int p = pos + l; if (!ism[l] || p < anchor) continue;
                int mp = p, mc = cand[l]; any = 1;
                if (flags & 4) while (mp > anchor && mc > 0 && d[mp - 1] == d[mc - 1]) { mp--; mc--; }
                int ml = 0; while (mp + ml < me && d[mp + ml] == d[mc + ml]) ml++;
                if (ml < g_min) continue;
                uint32_t lit = mp - anchor;
                if (first) { out += 1 + ext(lit + carry) + lit + carry; carry = 0; first = 0; } else out += 1 + ext(lit) + lit;
                out += 2 + ext(ml - 4); nseq++; offsum += mp - mc; if (mp - mc < 416) near++;
                anchor = mp + ml; }
            if (any) miss = 0; else miss++;
            int nxt = pos + 64;
            if ((flags & 8) && !any) nxt += 64 * (miss >> 2);
            pos = anchor > nxt ? anchor : nxt;
        }
        carry += len - anchor;
    }
    out += 1 + ext(carry) + carry;
    printf("n=%ld out=%llu ratio=%.4f seqs=%llu steps=%llu (%.2f steps/KiB) mean offset %.0f, %.1f%% of matches reach < 416 B back\n", n, (unsigned long long)out, (double)out / n,
           (unsigned long long)nseq, (unsigned long long)steps, steps / (n / 1024.0), nseq ? offsum / nseq : 0.0, nseq ? 100.0 * near / nseq : 0.0);
    return 0;
}
