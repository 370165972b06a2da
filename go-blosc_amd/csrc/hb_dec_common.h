// hb_dec_common.h — lane-parallel copy machinery shared by the LZ4 and Snappy block decoders (hb_lz4_dec.hip,
// hb_snappy.hip): a queue of parsed tokens {literal run, match} in LDS, one token per lane, output positions from a wave
// scan, every lane copies its own literals and its own match into an LDS image of the output, matches in dependency rounds.
#pragma once
#include "hb_lz4.h"

#define DTQ 96                           // token queue slots: < 64 queued before a window is parsed; a 64-byte window adds <= 22 LZ4 tokens or <= 32 Snappy elements

#define DLITCAP 16u
#define DMCAP 32u             // matches up to this long are copied by their own lane

// 4 bytes at any byte address of a 4-byte aligned LDS array: two aligned dwords + v_alignbyte.
// gfx950 LDS also takes misaligned ds_read/ds_write_b16/b32/b64 (hipcc emits them for align-1 types) but
// replays them; they still beat byte loops for the short exact-length copies below.
__device__ __forceinline__ uint32_t dec_read4(const uint8_t *base, uint32_t a) {
    const uint32_t *w = (const uint32_t *)base;
    const uint32_t w0 = w[a >> 2], w1 = w[(a >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
}

// exact-length copy inside LDS, no overlap between [d, d+len) and [s, s+len): 8/4/2/1-byte pieces
__device__ __forceinline__ void lds_copy_exact(uint8_t *d, const uint8_t *s, uint32_t len) {
    uint32_t k = 0;
    for (; k + 8u <= len; k += 8u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(s + k))->v;
    if (len & 4u) { ((hb_u32u *)(d + k))->v = ((const hb_u32u *)(s + k))->v; k += 4u; }
    if (len & 2u) { ((hb_u16u *)(d + k))->v = ((const hb_u16u *)(s + k))->v; k += 2u; }
    if (len & 1u) d[k] = s[k];
}
// a match copied by one lane: out[md + k] = out[md - off + k]; pieces never read bytes they have not written yet
__device__ __forceinline__ void lds_match_lane(uint8_t *out, uint32_t md, uint32_t off, uint32_t len) {
    if (off >= 8u) { lds_copy_exact(out + md, out + md - off, len); return; }
    if (off == 1u) {                                            // run of one byte
        const uint64_t pat = 0x0101010101010101ull * out[md - 1u];
        uint32_t k = 0;
        for (; k + 8u <= len; k += 8u) ((hb_u64u *)(out + md + k))->v = pat;
        if (len & 4u) { ((hb_u32u *)(out + md + k))->v = (uint32_t)pat; k += 4u; }
        if (len & 2u) { ((hb_u16u *)(out + md + k))->v = (uint16_t)pat; k += 2u; }
        if (len & 1u) out[md + k] = (uint8_t)pat;
        return;
    }
    for (uint32_t k = 0; k < len; k++) out[md + k] = out[md - off + k];   // offsets 2..7: byte by byte
}
__device__ __forceinline__ uint32_t dec_incl_scan(uint32_t v, int lane) { (void)lane; return wave_incl_scan_dpp(v); }

// copy a match inside the LDS image of the chunk; all arguments wave-uniform, source fully produced
__device__ __forceinline__ void dec_match_copy(uint8_t *s_out, uint32_t md, uint32_t off, uint32_t ml, int lane) {
    if (off >= 64u || off >= ml) {
        for (uint32_t c0 = 0; c0 < ml; c0 += 64) {
            const uint32_t i = c0 + lane;
            uint8_t v = 0;
            if (i < ml) v = s_out[md + i - off];
            if (i < ml) s_out[md + i] = v;
        }
    } else if (off <= 4u && off != 3u && ml >= 128u) {
        // a long run with period 1, 2 or 4 (what a shuffled plane of slowly varying values is made of): the period
        // divides 4, so every aligned dword of the destination holds the same rotated pattern -- 16 bytes per lane
        uint8_t *d = s_out + md;
        uint32_t w;                                             // the period, repeated to 4 bytes (md >= off: no read before the image)
        if (off == 1u) w = (uint32_t)d[-1] * 0x01010101u;
        else if (off == 2u) w = ((uint32_t)d[-2] | ((uint32_t)d[-1] << 8)) * 0x00010001u;
        else w = (uint32_t)d[-4] | ((uint32_t)d[-3] << 8) | ((uint32_t)d[-2] << 16) | ((uint32_t)d[-1] << 24);
        // w = bytes for positions j = 0..3 (mod 4) relative to md; destination dwords start at j = head (mod 4)
        const uint32_t head = (uint32_t)((16u - ((uintptr_t)d & 15u)) & 15u);
        const uint32_t wr = __builtin_amdgcn_alignbyte(w, w, head & 3u);
        if ((uint32_t)lane < head) d[lane] = (uint8_t)(w >> (8u * ((uint32_t)lane & 3u)));
        const uint32_t nblk = (ml - head) >> 4;
        u32x4 pat; pat.x = wr; pat.y = wr; pat.z = wr; pat.w = wr;
        for (uint32_t b = lane; b < nblk; b += 64) *(u32x4 *)(d + head + 16u * b) = pat;
        const uint32_t done = head + 16u * nblk;
        if (done + (uint32_t)lane < ml) d[done + lane] = (uint8_t)(wr >> (8u * ((uint32_t)lane & 3u)));
    } else {                                                   // overlapping: the source is [md-off, md), repeated
        uint32_t m = (uint32_t)lane % off;
        const uint32_t step = 64u % off;
        for (uint32_t c0 = 0; c0 < ml; c0 += 64) {
            const uint32_t i = c0 + lane;
            if (i < ml) s_out[md + i] = s_out[md - off + m];
            m += step; if (m >= off) m -= off;
        }
    }
}

// DRAIN: one queued token per lane, while 64 are queued (or `stop` and any are).  out[0] is the first byte of the image,
// `hist` bytes before it are valid match sources, `outlen` is the room.  A token that does not fit is not decoded:
// `rewound` is set and si goes back to that token.  Returns false on a match that reaches before out[-hist] or has offset 0.
__device__ __forceinline__ bool dec_drain(const uint8_t *in, const int inoff, uint8_t *s_out, const uint32_t outlen, const uint32_t hist,
                                          uint32_t &di, uint32_t &si, uint32_t &nq, uint2 *s_tq, const bool stop,
                                          bool &rewound, const int lane) {
    bool ok = true;
    while (nq >= 64u || (stop && nq > 0u)) {
        const uint32_t cntb = nq < 64u ? nq : 64u;
        const uint2 e = s_tq[lane];
        const uint32_t lsrc = (uint32_t)((int)(e.x & 0x1FFFu) + inoff), lit = (e.x >> 13) & 0x1FFu, mlen = e.x >> 22;
        const uint32_t offv = e.y & 0xFFFFu, tp = e.y >> 16;
        const uint32_t olen = (uint32_t)lane < cntb ? lit + mlen : 0u;
        const uint32_t incl = dec_incl_scan(olen, lane);
        const uint32_t dpos = di + incl - olen;
        uint32_t total = __builtin_amdgcn_readlane(incl, 63);
        unsigned long long amask = cntb >= 64u ? ~0ull : ((1ull << cntb) - 1ull);
        const unsigned long long om = hb_ballot(olen != 0u && dpos + olen > outlen);
        if (om) {                                        // this sequence passes the end of the unit: slow path from its token
            const int jx = __builtin_ctzll(om);
            amask &= (1ull << jx) - 1ull;
            total = __builtin_amdgcn_readlane(dpos, jx) - di;
            si = __builtin_amdgcn_readlane(tp, jx);
            rewound = true;
        }
        const bool istok = (amask >> lane) & 1ull;
        // a match may only read what this unit has produced (else: not ours to decide -> serial decoder)
        // (a token with mlen == 0 carries no match: the Snappy decoder queues literal elements that way)
        if (hb_ballot(istok && mlen != 0u && (offv == 0u || offv > dpos + lit + hist))) { ok = false; break; }
        // literals: short runs by their own lane, long runs by the whole wave
        if (istok && lit <= DLITCAP) lds_copy_exact(s_out + dpos, in + lsrc, lit);
        unsigned long long lm = hb_ballot(istok && lit > DLITCAP);
        while (lm) {
            const int l = __builtin_ctzll(lm);
            const uint32_t sp = __builtin_amdgcn_readlane(lsrc, l), dp = __builtin_amdgcn_readlane(dpos, l);
            const uint32_t ln = __builtin_amdgcn_readlane(lit, l);
            for (uint32_t k = lane; k < ln; k += 64) s_out[dp + k] = in[sp + k];
            lm &= lm - 1;
        }
        // matches: every lane copies its own short match as soon as its source is final.  Everything before the
        // first pending match is final, so each round retires at least that one; matches longer than DMCAP
        // bytes are copied by the whole wave when they come first.
        const uint32_t mdv = dpos + lit;                               // where my match goes
        const int src0 = (int)mdv - (int)offv;                             // < 0: the source starts in the history
        const int srcend = src0 + (int)(mlen < offv ? mlen : offv);        // end of the source that is not my own output
        const uint32_t mend = mdv + mlen;                              // end of my match
        unsigned long long pend = amask & hb_ballot(mlen != 0u);
        while (pend) {
            const int f = __builtin_ctzll(pend);
            const uint32_t X = __builtin_amdgcn_readlane(mdv, f);
            const uint32_t mlf = __builtin_amdgcn_readlane(mlen, f);
            if (mlf > DMCAP) {
                dec_match_copy(s_out, X, __builtin_amdgcn_readlane(offv, f), mlf, lane);
                pend &= pend - 1;
                continue;
            }
            // ready: the source ends before the first pending match, or starts at / after the end of the
            // nearest pending match before me (everything between that and my own match is final: literals
            // and retired matches; with no pending predecessor that is simply "anything before me")
            const unsigned long long below = pend & ((1ull << lane) - 1ull);
            const uint32_t pj = below ? 63u - (uint32_t)__builtin_clzll(below) : 0u;
            const uint32_t pe = (uint32_t)__shfl((int)mend, (int)pj);
            const bool ready = ((pend >> lane) & 1ull) && mlen <= DMCAP &&
                               (srcend <= (int)X || below == 0ull || src0 >= (int)pe);
            if (ready) lds_match_lane(s_out, mdv, offv, mlen);
            pend &= ~hb_ballot(ready);
        }
        di += total;
        if (rewound) { nq = 0; break; }
        const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];   // keep what is queued beyond the 64 just decoded
        nq -= cntb;
        if ((uint32_t)lane < nq) s_tq[lane] = rest;
    }
    return ok;
}
