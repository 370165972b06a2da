// hb_snappy.hip — Snappy block decoder for gfx950: replaces snappyCodec.Decompress (codec.go:237-244, i.e. snappy.Decode of
// klauspost/compress v1.18.2, which is not in the reference tree) on the device.  The encoder side is the LZ4 matcher with a
// Snappy emitter (hb_lz4_enc.hip: match_chunk<1, true>, k_sn_tiles / k_sn_scan / k_sn_pack).
//
// Block format (Snappy format_description.txt): uvarint(decoded length), then elements.  Tag byte, low two bits:
//   00 literal : len-1 = tag >> 2 if < 60, else 60..63 -> the next 1..4 bytes hold len-1 (little endian)
//   01 copy    : len = 4 + ((tag >> 2) & 7), offset = (tag >> 5) << 8 | next byte          (2 bytes)
//   10 copy    : len = 1 + (tag >> 2), offset = next 2 bytes                               (3 bytes)
//   11 copy    : len = 1 + (tag >> 2), offset = next 4 bytes                               (5 bytes)
// Decoder contract restated from the format + the Go decoder's published behaviour: offset 0, offset beyond the bytes produced,
// an element running past the input, output beyond the declared length, or fewer bytes than declared -> corrupt
// (-> ErrDecompressionFailed, blosc.go:411-413); the result always has the DECLARED length, which the frame layer then compares
// with NBytesOrig (ErrSizeMismatch, blosc.go:429-431).  NOT restated: S2's "repeat offset" reading of a 1-byte-offset copy with
// offset 0 (a klauspost extension that no Snappy encoder emits); such an element is corrupt here.
//
//   k_sn_dec_indexed : blocks with a unit index (HBSX, hb_format.h) -- every block this library encodes.  One wavefront per
//       4 KiB unit of output, the unit's slice of the stream staged in LDS; 64 lanes parse 64 stream bytes "as if an element
//       started at my byte", the real chain is followed with one s_bitset1 + one v_readlane per element, elements are queued and
//       copied one per lane by the machinery shared with the LZ4 decoder (hb_dec_common.h: a literal element is a token without
//       a match, a copy a token without literals).  The index is not trusted: a unit must consume exactly its slice and
//       produce exactly its 4 KiB, and no copy may reach before the unit; anything else raises a flag ...
//   k_sn_dec_serial  : ... and the whole block is decoded by one wavefront front to back: 64 KiB of history in LDS, copies that reach
//       further back read the output in HBM.  Also the authority for every block the parallel paths do not vouch for.
//   blocks WITHOUT an index (any other writer's; round 4, further down): the element chain is found and verified by the token discovery of
//       hb_lz4_region.hip (k_snr_*), cut at every 64 KiB of output (k_snr_units_fast / k_snr_units) and decoded one unit per wavefront
//       (k_sn_dec_units); the symbolic decoder of hb_lz4_sym.hip behind it for streams whose copies cross those units (larger workspace).
#include "hb_lz4.h"
#include "hb_dec_common.h"
#include "hb_lz4_region.h"
#define SY_IMG   2048u          // (the unit decoder of hb_sym_decode.h as hb_lz4_sym.hip configures it: 2 KiB image, 512 bytes of history)
#define SY_HIST  512u
#include "hb_sym_decode.h"

struct SnPlan {
    uint32_t mode;        // 0 = serial, 1 = indexed (4 KiB units, stored index), 2 = blocks (64 KiB units found by the discovery)
    uint32_t fail;
    uint32_t nunits;
    uint32_t nbytes;      // declared (uvarint) length
    uint32_t hdr;         // bytes of the uvarint
    uint32_t pad[3];
};

#define SN_IN_MAX 4608u      // largest unit slice the indexed decoder stages (4096 literal bytes + headers + slack)

__device__ __forceinline__ bool sn_uvarint(const uint8_t *src, uint64_t n, uint64_t &v, uint32_t &used) {
    v = 0;
    for (uint32_t i = 0; i < 10u && i < n; i++) {
        const uint32_t b = src[i];
        v |= (uint64_t)(b & 127u) << (7u * i);
        if (!(b & 128u)) { used = i + 1u; return !(i == 9u && b > 1u); }
    }
    return false;
}

__global__ void k_sn_dec_plan(const uint8_t *__restrict__ src, uint64_t n_src, const uint8_t *__restrict__ index, uint64_t index_bytes,
                              uint64_t cap, SnPlan *plan, hb_result *result) {
    plan->mode = 0; plan->fail = 0; plan->nunits = 0; plan->nbytes = 0; plan->hdr = 0;
    result->status = HB_OK; result->flags = 0; result->bytes = 0; result->total_bytes = 0; result->reserved = 0;
    uint64_t dlen; uint32_t hl;
    if (!sn_uvarint(src, n_src, dlen, hl) || dlen > 0xFFFFFFFFull) return;      // the serial kernel reports the error
    if (dlen <= cap) { plan->nbytes = (uint32_t)dlen; plan->hdr = hl; }           // (the block-parallel path of index-less blocks starts from these)
    if (!index || index_bytes < HB_SNX_HDR_BYTES + 2 * HB_SNX_ENTRY) return;
    uint32_t h[8];
    for (int i = 0; i < 8; i++) h[i] = ld4u(index + 4 * i);
    if (h[0] != HB_SNX_MAGIC || h[1] != (HB_IDX_VERSION | (HB_SNX_ENTRY << 16))) return;
    if (h[7] != (h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5] ^ h[6])) return;
    const uint64_t nunits = h[2];
    if (nunits == 0 || HB_SNX_HDR_BYTES + (nunits + 1) * HB_SNX_ENTRY > index_bytes) return;
    if (h[3] != HB_CHUNK || nunits != ((uint64_t)h[5] + HB_CHUNK - 1) / HB_CHUNK) return;     // one unit per 4 KiB of output, exactly
    if (h[4] != n_src || h[5] != dlen || dlen > cap || h[6] != hl) return;
    plan->nunits = (uint32_t)nunits; plan->nbytes = (uint32_t)dlen; plan->hdr = hl;
    plan->mode = 1;
}

// (the window parser sn_fill and the one-element parser sn_parse_uniform: hb_lz4_region.h -- the symbolic decoder of foreign blocks walks elements too)

__global__ __launch_bounds__(64) void k_sn_dec_indexed(const uint8_t *__restrict__ src, uint64_t n_src, uint8_t *__restrict__ dst,
                                                       const uint8_t *__restrict__ index, SnPlan *plan) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[SN_IN_MAX + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[HB_CHUNK + 64];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    if (plan->mode != 1) return;
    const int lane = threadIdx.x;
    const uint32_t nunits = plan->nunits, nbytes = plan->nbytes, hl = plan->hdr;
    const uint8_t *ent = index + HB_SNX_HDR_BYTES;
    for (uint32_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const uint32_t s0 = RFL(ld4u(ent + 4 * (size_t)u)), s1 = RFL(ld4u(ent + 4 * (size_t)(u + 1)));
        const uint32_t d0 = u * HB_CHUNK, outlen = min(HB_CHUNK, nbytes - d0);
        bool ok = s0 >= hl && s0 <= s1 && s1 <= n_src && (s1 - s0) <= SN_IN_MAX;
        if (u == 0) ok = ok && s0 == hl;
        if (u + 1 == nunits) ok = ok && s1 == n_src;
        if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); continue; }
        const uint32_t slen = s1 - s0;
        const uint8_t *g = src + s0;
        const uint32_t sh = (uint32_t)((uintptr_t)g & 15u);
        wave_sync();
        {
            const u32x4 *ga = (const u32x4 *)(g - sh);
            const uint32_t nv = (sh + slen + 15u) >> 4;
            for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_in)[i] = ga[i];
        }
        wave_sync();
        uint32_t si = 0, di = 0, nq = 0;
        bool done = false;
        while (ok && !done) {
            const bool stop = sn_fill(s_in, sh, slen, si, nq, s_tq, lane);
            bool rewound = false;
            ok = dec_drain<false, true>(s_in, (int)sh, s_out, outlen, 0u, di, si, nq, s_tq, stop, rewound, lane);
            if (!ok || rewound) { ok = false; break; }            // an element that passes the end of the unit: not ours to decide
            if (stop) {
                if (si == slen) { done = true; break; }
                SnElem e;                                        // one element the slow way: a literal longer than 511 bytes
                if (!sn_parse_uniform(s_in + sh + si, slen - si, e) || e.kind != 0u) { ok = false; break; }
                if (e.lit > (uint64_t)(slen - si - e.hdr) || e.lit > (uint64_t)(outlen - di)) { ok = false; break; }
                const uint32_t ls = sh + si + e.hdr, ln = (uint32_t)e.lit;
                for (uint32_t k = lane; k < ln; k += 64) s_out[di + k] = s_in[ls + k];
                si += e.hdr + ln; di += ln;
                wave_sync();
            }
        }
        if (ok) ok = (si == slen) && (di == outlen);
        if (!ok) { if (lane == 0) atomicExch(&plan->fail, 1u); wave_sync(); continue; }
        wave_sync();
        uint8_t *o = dst + d0;                                    // flush the unit image: 16-byte stores on an aligned body
        uint32_t head = (uint32_t)((16u - ((uintptr_t)o & 15u)) & 15u);
        if (head > outlen) head = outlen;
        if ((uint32_t)lane < head) o[lane] = s_out[lane];
        const uint32_t body = (outlen - head) >> 4;
        if (head == 0) { for (uint32_t i = lane; i < body; i += 64) *(u32x4 *)(o + i * 16u) = *(const u32x4 *)(s_out + i * 16u); }
        else {
            for (uint32_t i = lane; i < body; i += 64) {
                const uint8_t *q = s_out + head + i * 16u;
                u32x4 v;
                v.x = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
                v.y = (uint32_t)q[4] | ((uint32_t)q[5] << 8) | ((uint32_t)q[6] << 16) | ((uint32_t)q[7] << 24);
                v.z = (uint32_t)q[8] | ((uint32_t)q[9] << 8) | ((uint32_t)q[10] << 16) | ((uint32_t)q[11] << 24);
                v.w = (uint32_t)q[12] | ((uint32_t)q[13] << 8) | ((uint32_t)q[14] << 16) | ((uint32_t)q[15] << 24);
                *(u32x4 *)(o + head + i * 16u) = v;
            }
        }
        const uint32_t done_b = head + body * 16u;
        if (done_b + lane < outlen) o[done_b + lane] = s_out[done_b + lane];
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------------------
// k_sn_dec_serial: one wavefront, whole block, front to back
// ------------------------------------------------------------------------------------------------------------
#define SNS_WIN   8192u
#define SNS_HIST  65536u
#define SNS_PAGE  32768u
#define SNS_SOFT  16384u

__global__ __launch_bounds__(64) void k_sn_dec_serial(const uint8_t *__restrict__ src, uint64_t n_src, uint8_t *__restrict__ dst, uint64_t cap,
                                                      SnPlan *plan, hb_result *result, int frame, uint32_t expect) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[SNS_WIN + 128];
    __shared__ __attribute__((aligned(16))) uint8_t s_img[SNS_HIST + SNS_PAGE + 1024];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    if ((plan->mode == 1 || plan->mode == 2) && !plan->fail) {
        if (lane == 0) {
            const uint64_t got = plan->nbytes;
            result->flags = 1; result->bytes = got; result->total_bytes = got;
            result->status = (frame && got != expect) ? HB_ERR_SIZE_MISMATCH : HB_OK;      // blosc.go:429-431
        }
        return;
    }
    int err = 0;
    uint64_t dlen = 0; uint32_t hl = 0;
    if (!sn_uvarint(src, n_src, dlen, hl) || dlen > 0xFFFFFFFFull) err = 1;
    if (!err && dlen > cap) {
        // more declared bytes than the caller's buffer (the frame's NBytesOrig): the reference decodes into a buffer of its
        // own and the frame layer then reports ErrSizeMismatch; without that buffer the stream cannot be validated here
        if (lane == 0) { result->flags = 0; result->total_bytes = 0; result->bytes = dlen; result->status = frame ? HB_ERR_SIZE_MISMATCH : HB_ERR_SHORT_BUFFER; }
        return;
    }
    uint8_t *out = s_img + SNS_HIST;
    uint64_t gbase = 0, si = hl, wpos = 0;
    uint32_t di = 0, wlen = 0, wsh = 0, nq = 0;
    auto refill = [&](uint64_t at) __attribute__((always_inline)) {
        const uint8_t *g = src + at;
        wsh = (uint32_t)((uintptr_t)g & 15u);
        const uint64_t left = n_src - at;
        wlen = (uint32_t)(left < (uint64_t)(SNS_WIN - 16u) ? left : (uint64_t)(SNS_WIN - 16u));
        const u32x4 *ga = (const u32x4 *)(g - wsh);
        const uint32_t nv = (wsh + wlen + 15u) >> 4;
        wave_sync();
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
        wpos = at;
        wave_sync();
    };
    auto flush = [&](bool all) __attribute__((always_inline)) {
        const uint32_t fl = all ? di : (di & ~15u);
        if (fl == 0) return;
        wave_sync();
        uint8_t *o = dst + gbase;
        for (uint32_t i = lane * 16u; i + 16u <= fl; i += 1024u) st16u(o + i, *(const u32x4 *)(out + i));
        const uint32_t tail0 = fl & ~15u;
        if (tail0 + lane < fl) o[tail0 + lane] = out[tail0 + lane];
        if (!all) {
            const uint32_t mv = SNS_HIST + (di - fl);
            for (uint32_t k = lane * 16u; k < mv; k += 1024u) { const u32x4 v = *(const u32x4 *)(s_img + fl + k); *(u32x4 *)(s_img + k) = v; }
            wave_sync();
        }
        gbase += fl; di -= fl;
    };
    bool fin = false;
    if (!err) { if (si < n_src) refill(si); else fin = true; }
    while (!err && !fin) {
        if (gbase + di > dlen) { err = 1; break; }
        if (di >= SNS_SOFT) flush(false);
        if (si == n_src) { fin = true; break; }
        if (si < wpos || si - wpos + 1024u > wlen) { if (si != wpos || wlen == 0) refill(si); }
        uint32_t rel = (uint32_t)(si - wpos);
        const uint32_t hist = (uint32_t)(gbase < (uint64_t)SNS_HIST ? gbase : (uint64_t)SNS_HIST);
        // ---- fast path: window-parallel parse, lane-parallel copies (offsets inside the LDS history) ----
        const uint64_t left_out = dlen - gbase;
        const uint32_t room = (uint32_t)(left_out < (uint64_t)SNS_PAGE ? left_out : (uint64_t)SNS_PAGE);
        const bool stop = sn_fill(s_win, wsh, wlen, rel, nq, s_tq, lane);
        bool rewound = false;
        const uint32_t di0 = di;
        const bool dok = dec_drain<false, true>(s_win + wsh, 0, out, room, hist, di, rel, nq, s_tq, true, rewound, lane);
        const bool moved = (wpos + rel) != si;
        if (!dok) {
            // a copy that reaches beyond the LDS history (> 64 KiB back) or is malformed: decided one element at a time below,
            // from the first element of the batch (the batch copied nothing that later elements could not overwrite again)
            di = di0; nq = 0;
        } else {
            si = wpos + rel;
            if (rewound) {                                       // an element that does not fit: the page is full, or it passes the declared length
                if (room < SNS_PAGE) { err = 1; break; }
                flush(false);
                continue;
            }
            else if (moved && !stop) continue;
            else if (si == n_src) { fin = true; break; }
            else if (moved && si - wpos + 1024u > wlen && wpos + wlen < n_src) continue;     // stopped at the window edge: refill first
        }
        // ---- one element the slow way: literal of any size, copy with any offset ----
        if (si < wpos || si - wpos + 8u > wlen) refill(si);
        rel = (uint32_t)(si - wpos);
        SnElem e;
        if (!sn_parse_uniform(s_win + wsh + rel, n_src - si, e)) { err = 1; break; }
        si += e.hdr;
        if (e.kind == 0u) {
            if (e.lit > n_src - si || e.lit > dlen - (gbase + di)) { err = 1; break; }
            uint64_t lrem = e.lit;
            while (lrem) {
                const uint32_t rm = SNS_PAGE - di;
                const uint32_t take = (uint32_t)(lrem < (uint64_t)rm ? lrem : (uint64_t)rm);
                if (take == 0) { flush(false); continue; }
                const uint8_t *g = src + si;
                uint32_t k0 = 0;
                for (; k0 + 4096u <= take; k0 += 4096u) {
                    u32x4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = ld16u(g + k0 + (uint32_t)q * 1024u + (uint32_t)lane * 16u);
#pragma unroll
                    for (int q = 0; q < 4; q++) ((hb_u128u *)(out + di + k0 + (uint32_t)q * 1024u + (uint32_t)lane * 16u))->v = v[q];
                }
                for (uint32_t k = k0 + lane; k < take; k += 64) out[di + k] = g[k];
                si += take; di += take; lrem -= take;
                wave_sync();
                if (di >= SNS_SOFT) flush(false);
            }
        } else {
            const uint64_t produced = gbase + di;
            if (e.off == 0 || e.off > produced || (uint64_t)e.mlen > dlen - produced) { err = 1; break; }
            if (SNS_PAGE - di < e.mlen) flush(false);
            if (e.off <= (uint64_t)di + hist) {
                wave_sync();
                dec_match_copy(out, di, (uint32_t)e.off, e.mlen, lane);
            } else {
                // beyond the LDS history: the source lies in output this wave flushed to HBM long ago (off > 64 KiB >= mlen: no overlap)
                const uint8_t *gs = dst + (produced - e.off);
                if ((uint32_t)lane < e.mlen) out[di + lane] = __builtin_nontemporal_load(gs + lane);
            }
            di += e.mlen;
            wave_sync();
        }
    }
    if (!err && gbase + di != dlen) err = 1;                           // fewer (or more) bytes than declared
    if (!err) flush(true);
    if (lane == 0) {
        result->flags = 0; result->total_bytes = 0;
        if (err) { result->status = HB_ERR_DECOMPRESSION_FAILED; result->bytes = 0; }                       // blosc.go:411-413
        else if (frame && dlen != expect) { result->status = HB_ERR_SIZE_MISMATCH; result->bytes = dlen; }    // blosc.go:429-431
        else { result->status = HB_OK; result->bytes = dlen; }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Blocks without an index (written by any other Snappy encoder -- the reference's frames, codec.go:228-235): block-parallel decode.
// Every Snappy encoder in use compresses its input in blocks of 64 KiB that share nothing (golang/snappy, libsnappy: the hash table is reset per
// block and the remaining literals are emitted at its end), so the element stream is a concatenation of sub-streams that decode to exactly 64 KiB
// each and copy only from themselves -- but nothing in the stream says where they begin.  The token discovery of hb_lz4_region.hip finds and
// VERIFIES the element chain (same kernels, element parser: hb_launch_snappy_region_chain); k_snr_units walks it once more with output positions
// and notes the element that starts at every multiple of 64 KiB of output; k_sn_dec_units decodes one such unit per wavefront (at first with a 64 KiB image in
// LDS).  Nothing is assumed: a unit must consume exactly its slice, produce exactly its bytes, and no copy may reach in front of it -- an encoder
// that matches across 64 KiB (klauspost's s2.EncodeSnappy on one large block does) fails that test, and the single wavefront decodes the block.
// ------------------------------------------------------------------------------------------------------------
#define SNB_UNIT 65536u

// the verified chain of region r, elements 64 at a time: f(cnt, pos, olen, opos) -- lane < cnt holds the element at stream position pos that
// produces olen bytes at output position opos (64-bit); returns false when the walk leaves the stream (cannot happen on a verified chain)
template <class F>
__device__ __forceinline__ bool snr_walk(const uint8_t *__restrict__ src, const uint64_t n_src, const uint64_t start, const uint64_t exitp, const uint64_t opos0,
                                         uint8_t *s_win, uint2 *s_tq, const int lane, F &&f) {
    uint64_t si = start, wpos = 0, out = opos0;
    uint32_t wlen = 0, wsh = 0, nq = 0;
    auto refill = [&](uint64_t at) __attribute__((always_inline)) {
        const uint8_t *g = src + at;
        wsh = (uint32_t)((uintptr_t)g & 15u);
        const uint64_t left = n_src - at;
        wlen = (uint32_t)(left < (uint64_t)(RG_PWIN - 16u) ? left : (uint64_t)(RG_PWIN - 16u));
        const u32x4 *ga = (const u32x4 *)(g - wsh);
        const uint32_t nv = (wsh + wlen + 15u) >> 4;
        wave_sync();
        for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_win)[i] = ga[i];
        wpos = at;
        wave_sync();
    };
    wave_sync();
    for (;;) {
        if (si >= exitp) return true;
        if (si < wpos || si - wpos + 1024u > wlen) { if (si != wpos || wlen == 0) refill(si); }
        uint32_t rel = (uint32_t)(si - wpos);
        const bool stop = sn_rg_fill(s_win, wsh, wlen, rel, nq, s_tq, lane);
        while (nq > 0u) {
            const uint32_t cntb = nq < 64u ? nq : 64u;
            const uint2 e = s_tq[lane];
            const uint64_t ap = wpos + e.x;
            const unsigned long long over = hb_ballot((uint32_t)lane < cntb && ap >= exitp);
            const uint32_t cnt = over ? (uint32_t)__builtin_ctzll(over) : cntb;
            const uint32_t olen = (uint32_t)lane < cnt ? e.y : 0u;
            const uint32_t incl = wave_incl_scan_dpp(olen);
            f(cnt, ap, olen, out + (incl - olen));
            out += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
            if (over) return true;
            const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];
            nq -= cntb;
            if ((uint32_t)lane < nq) s_tq[lane] = rest;
        }
        const bool moved = (wpos + rel) != si;
        si = wpos + rel;
        if (si > n_src) return false;
        if (moved && !stop) continue;
        if (si >= exitp) continue;
        if (moved && si - wpos + 1024u > wlen && wpos + wlen < n_src) continue;
        if (si < wpos || si + 8u > wpos + wlen) refill(si);
        rel = (uint32_t)(si - wpos);
        SnElem e;
        if (!sn_parse_uniform(s_win + wsh + rel, n_src - si, e)) return false;
        const uint64_t p = si + e.hdr;
        if (e.lit > n_src - p || e.lit > 0xFFFFFFF0ull) return false;
        f(1u, si, (uint32_t)e.lit + e.mlen, out);
        out += e.lit + e.mlen;
        si = p + e.lit;
    }
}

// units[u] = 1 + stream position of the element that starts at output position u * SNB_UNIT (0: none); an element that crosses such a position: fail
__global__ __launch_bounds__(64) void k_snr_units(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *rgplan, const RgRegion *__restrict__ reg, uint32_t *__restrict__ units, uint32_t nunits_max,
                                                  const uint32_t *__restrict__ done) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    if (!rgplan->ok) return;
    const uint32_t nreg = rgplan->nreg;
    for (uint32_t r = blockIdx.x; r < nreg; r += gridDim.x) {
        const uint32_t entry = RFL(reg[r].entry), exitp = RFL(reg[r].exit);
        if (entry >= exitp) continue;                                    // no element of the chain starts in this region
        if (done && RFL(done[r]) != 0u) continue;                         // k_snr_units_fast found this region's units from its records
        const uint64_t opos = reg[r].opos;
        bool bad = false;
        const bool through = snr_walk(src, n_src, entry, exitp, opos, s_win, s_tq, lane, [&](uint32_t cnt, uint64_t pos, uint32_t olen, uint64_t o) __attribute__((always_inline)) {
            if ((uint32_t)lane < cnt && olen) {
                if (o / SNB_UNIT >= (uint64_t)nunits_max) bad = true;         // (more output than the chain's verified total allows: the table ends here)
                else if ((o & (SNB_UNIT - 1u)) == 0u) units[o / SNB_UNIT] = (uint32_t)pos + 1u;
                else if ((o / SNB_UNIT) != ((o + olen - 1u) / SNB_UNIT)) bad = true;
            }
        });
        if (hb_ballot(bad) || !through) { if (lane == 0) atomicExch(&rgplan->fail, 1u); }
        wave_sync();
    }
}

// The same from the discovery's records, without a walk of the whole stream (the wave walk above costs 1.2 ms per GiB only to find 16 384 elements).  A
// region's first parse left {position, output so far} of the first element in each of 128 slices of its stream range (`traces`, hb_lz4_region.h), and
// from RgRegion.pad0 on that record lies on the verified chain when the recorded parse ends where the chain does: an element at recorded output c
// then starts at output opos + outlen - (outlen0 - c).  One wavefront per region, a LANE per record: the lane whose stretch of output -- from its
// record to the next one's -- holds a multiple of 64 KiB walks the elements from its record (a few dozen, straight from memory) and notes the one that
// starts there, or finds that an element crosses it (no units: rgplan->fail).  Lane 0 also takes the head, from the region's entry to the first
// usable record.  A region whose records are not usable, or whose walk would be long, stays with the wave walk (done[r] = 0).
#define SNU_WALKCAP 256u           // elements a lane walks before it leaves the region to the wave walk (a lane's element costs a memory round trip, the wave's a sixtieth)
__global__ __launch_bounds__(64) void k_snr_units_fast(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *rgplan, const RgRegion *__restrict__ reg, const uint2 *__restrict__ traces,
                                                       uint32_t *__restrict__ units, uint32_t nunits_max, uint32_t *__restrict__ done) {
    const int lane = threadIdx.x;
    const uint32_t r = blockIdx.x;
    if (lane == 0) done[r] = 0u;
    if (!rgplan->ok || r >= rgplan->nreg) return;
    const uint32_t entry = RFL(reg[r].entry), exitp = RFL(reg[r].exit), outlen = RFL(reg[r].outlen);
    if (entry >= exitp || outlen == 0u) { if (lane == 0) done[r] = 1u; return; }
    const uint64_t opos = reg[r].opos, oend = opos + outlen;
    if (((opos + SNB_UNIT - 1u) & ~(uint64_t)(SNB_UNIT - 1u)) >= oend) { if (lane == 0) done[r] = 1u; return; }      // no multiple of 64 KiB in this region's output
    const uint32_t pad0 = RFL(reg[r].pad0), exit0 = RFL(reg[r].exit0), outlen0 = RFL(reg[r].outlen0), rec0 = RFL(reg[r].entry0);
    if (pad0 == RG_INVALID || exit0 != exitp || pad0 >= exitp) return;      // records unusable: the wave walk
    const uint2 *tr = traces + (size_t)r * RG_TRACE + RG_DENSE;
    // my two records (slices lane and lane + 64), as {position, absolute output}
    uint32_t x[2]; uint64_t fo[2]; bool v[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const uint2 t = tr[lane + 64 * q];
        x[q] = t.x;
        v[q] = t.x != RG_INVALID && t.x >= pad0 && t.x >= rec0 && t.x < exitp && t.y <= outlen0 && (uint64_t)(outlen0 - t.y) <= (uint64_t)outlen;
        fo[q] = oend - (uint64_t)(outlen0 - (v[q] ? t.y : 0u));
    }
    const unsigned long long V0 = hb_ballot(v[0]), V1 = hb_ballot(v[1]);
    // where each stretch ends: the next usable record's output (records are in stream order: slice k before slice k + 1), or the region's end
    auto next_fo = [&](const int q) __attribute__((always_inline)) -> uint64_t {
        unsigned long long a = q == 0 ? (lane < 63 ? V0 >> (lane + 1) : 0ull) : 0ull;
        int src_lane, src_q;
        if (q == 0 && a) { src_lane = lane + 1 + __builtin_ctzll(a); src_q = 0; }
        else {
            const unsigned long long b = q == 0 ? V1 : (lane < 63 ? V1 >> (lane + 1) : 0ull);
            if (!b) { src_lane = -1; src_q = 0; }
            else { src_lane = (q == 0 ? 0 : lane + 1) + __builtin_ctzll(b); src_q = 1; }
        }
        // (every lane takes part in the shuffles)
        const uint32_t lo0 = (uint32_t)__shfl((int)(uint32_t)fo[0], src_lane < 0 ? 0 : src_lane), hi0 = (uint32_t)__shfl((int)(uint32_t)(fo[0] >> 32), src_lane < 0 ? 0 : src_lane);
        const uint32_t lo1 = (uint32_t)__shfl((int)(uint32_t)fo[1], src_lane < 0 ? 0 : src_lane), hi1 = (uint32_t)__shfl((int)(uint32_t)(fo[1] >> 32), src_lane < 0 ? 0 : src_lane);
        if (src_lane < 0) return oend;
        return src_q == 0 ? (((uint64_t)hi0 << 32) | lo0) : (((uint64_t)hi1 << 32) | lo1);
    };
    const uint64_t e0 = next_fo(0), e1 = next_fo(1);
    // the head: from the region's entry to the first usable record (lane 0's extra stretch)
    const bool anyv = (V0 | V1) != 0ull;
    uint64_t hend = oend;
    {
        int fl = V0 ? __builtin_ctzll(V0) : (V1 ? __builtin_ctzll(V1) : 0);
        const uint32_t lo0 = (uint32_t)__shfl((int)(uint32_t)fo[0], fl), hi0 = (uint32_t)__shfl((int)(uint32_t)(fo[0] >> 32), fl);
        const uint32_t lo1 = (uint32_t)__shfl((int)(uint32_t)fo[1], fl), hi1 = (uint32_t)__shfl((int)(uint32_t)(fo[1] >> 32), fl);
        if (anyv) hend = V0 ? (((uint64_t)hi0 << 32) | lo0) : (((uint64_t)hi1 << 32) | lo1);
    }
    bool bad = false, capped = false;
    auto stretch = [&](uint64_t p, uint64_t o, const uint64_t oe) __attribute__((always_inline)) {
        // elements from stream position p (output position o) while they start in front of oe: note those that start on a multiple of 64 KiB
        uint64_t B = (o + SNB_UNIT - 1u) & ~(uint64_t)(SNB_UNIT - 1u);
        uint32_t steps = 0;
        while (B < oe && !bad && !capped) {
            if (o == B) {
                if (B / SNB_UNIT >= (uint64_t)nunits_max) { bad = true; break; }
                units[B / SNB_UNIT] = (uint32_t)p + 1u;
                B += SNB_UNIT;
                continue;
            }
            if (o > B) { bad = true; break; }                             // an element crossed it
            if (++steps > SNU_WALKCAP) { capped = true; break; }
            uint64_t cum = 0;
            if (!rg_step_serial<RG_SNAPPY>(src, n_src, p, cum)) { bad = true; break; }
            o += cum;
        }
    };
    if (lane == 0 && hend > opos) stretch(entry, opos, hend);
#pragma unroll
    for (int q = 0; q < 2; q++) if (v[q]) stretch(x[q], fo[q], q == 0 ? e0 : e1);
    const bool anybad = hb_ballot(bad) != 0ull, anycap = hb_ballot(capped) != 0ull;
    if (lane == 0) {
        if (anybad) atomicExch(&rgplan->fail, 1u);
        done[r] = (anybad || !anycap) ? 1u : 0u;                          // (capped: the wave walk does this region again -- same entries)
    }
}

// 1 thread: the chain verified, its output is what the block declares, no element crosses a unit boundary -> mode 2
__global__ void k_snr_gate(const RgPlan *__restrict__ rgplan, SnPlan *plan, uint32_t *__restrict__ units, uint64_t n_src) {
    if (plan->mode != 0 || !rgplan->ok || rgplan->fail || plan->hdr == 0u || plan->nbytes == 0u) return;
    if (rgplan->total != (uint64_t)plan->nbytes) return;                 // fewer / more bytes than declared: the serial decoder reports it
    const uint32_t nunits = (plan->nbytes + SNB_UNIT - 1u) / SNB_UNIT;
    units[nunits] = (uint32_t)n_src + 1u;
    plan->nunits = nunits;
    plan->mode = 2;
}

// One 64 KiB unit per wavefront with the unit decoder of the symbolic pass, without the symbols (hb_sym_decode.h, sy_decode_unit<false>: a 2 KiB image in
// LDS, older sources fetched from the output in HBM, all lanes at once) fed by sn_walk: 5 KiB of LDS per wavefront -- 16 per CU.  (Round 4's first
// version kept the whole unit in a 64 KiB LDS image: two wavefronts per CU, 16.5 ms for 1 GiB where this one takes ~5.)  `base` = the unit's first
// byte: a copy from in front of it is not this decoder's to resolve and fails the unit, as does a unit that does not produce exactly its bytes.
#define SNU_PWIN 2048u
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4))) void k_sn_dec_units(const uint8_t *__restrict__ src, uint64_t n_src, uint8_t *__restrict__ dst,
                                                                                              const uint32_t *__restrict__ units, SnPlan *plan) {
    __shared__ __attribute__((aligned(16))) uint8_t s_win[SNU_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    __shared__ __attribute__((aligned(16))) uint8_t s_d[SY_IMG + 64];
    if (plan->mode != 2) return;
    const int lane = threadIdx.x;
    const uint32_t nunits = plan->nunits, nbytes = plan->nbytes;
    for (uint32_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        const uint32_t a0 = RFL(units[u]), a1 = RFL(units[u + 1]);
        const uint32_t O = u * SNB_UNIT;                                  // (nbytes < 2^32: so is every unit's start)
        const uint32_t limit = nbytes - O < SNB_UNIT ? nbytes : O + SNB_UNIT;
        bool ok = a0 != 0u && a1 != 0u && a0 < a1 && (uint64_t)(a1 - 1u) <= n_src;
        uint32_t out = O;
        if (ok) {
            bool parked;
            ok = sy_decode_unit<false, SNU_PWIN, false, RG_SNAPPY>(src, n_src, a0 - 1u, a1 - 1u, O, O, out, dst, nullptr, s_win, s_tq, s_d, nullptr,
                                                                   lane, 1, 0u, 0u, nullptr, nullptr, nullptr, parked, limit);
        }
        if (!ok || out != limit) { if (lane == 0) atomicExch(&plan->fail, 1u); }
        wave_sync();
    }
}

size_t hb_snappy_dec_workspace(size_t) { return 256; }

int hb_launch_snappy_decode(const hb_dec_args &a, hipStream_t s) {
    SnPlan *plan = (SnPlan *)a.work;
    hb_prof_begin("k_sn_dec_plan", s);
    hipLaunchKernelGGL(k_sn_dec_plan, dim3(1), dim3(1), 0, s, a.src, (uint64_t)a.n, a.index, (uint64_t)a.index_bytes, (uint64_t)a.cap, plan, a.result);
    hb_prof_end(s);
    // a block without an index that is worth a dozen launches: find the element chain, decode 64 KiB units in parallel (a.work is sized for the LZ4
    // discovery of a block of a.cap bytes -- hb_lz4_dec_workspace -- which this one fits into)
    static const bool no_blocks = [] { const char *e = getenv("HIPBLOSC_DEBUG_NO_SNAPPY_BLOCKS"); return e && *e && *e != '0'; }();   // A/B: the single wavefront
    if (!a.index && !no_blocks && hb_indexless_parallel(a.n, a.cap) && a.n < 0xFFFFFFF0ull && a.cap < 0xFFFFFFF0ull && a.n <= a.cap + a.cap / 255 + 16) {
        uint8_t *w = a.work + 256;
        const RgLayout L = rg_layout(a.cap);
        RgPlan *rgplan = (RgPlan *)(w + L.plan);
        const RgRegion *reg = (const RgRegion *)(w + L.reg);
        uint32_t *units = (uint32_t *)(w + L.total);
        const size_t nunits = (a.cap + SNB_UNIT - 1) / SNB_UNIT;
        uint64_t rs; uint32_t nreg;
        rg_regions(a.n, &rs, &nreg);
        int rc = hb_launch_snappy_region_chain(a.src, a.n, a.cap, w, &plan->hdr, s);
        if (rc) return rc;
        // units first: an encoder that compresses 64 KiB at a time (every Snappy encoder in wide use) leaves units that share nothing ...
        HB_HIP_TRY(hipMemsetAsync(units, 0, (nunits + 2) * 4, s));
        hb_prof_begin("k_snr_units", s);
        static const bool slow_units = [] { const char *e = getenv("HIPBLOSC_DEBUG_SLOW_UNITS"); return e && *e && *e != '0'; }();   // A/B: the wave walk only
        uint32_t *done = (uint32_t *)(w + L.pmax);                          // (k_rg_pmax / k_snr_fix are through with it)
        // (regions of a few KiB -- highly compressed blocks -- are walked faster by their wavefront than lane by lane: measured on a float ramp at ratio 0.05)
        const bool fast_units = !slow_units && rs >= 12288u;
        if (fast_units) hipLaunchKernelGGL(k_snr_units_fast, dim3(nreg), dim3(64), 0, s, a.src, (uint64_t)a.n, rgplan, reg, (const uint2 *)(w + L.trace), units, (uint32_t)nunits, done);
        // (behind the fast kernel few regions are left: a small grid strides over them -- 16 377 workgroups that look at a flag and leave took 0.39 ms)
        hipLaunchKernelGGL(k_snr_units, dim3(fast_units && nreg > 2048u ? 2048u : nreg), dim3(64), 0, s, a.src, (uint64_t)a.n, rgplan, reg, units, (uint32_t)nunits,
                           fast_units ? (const uint32_t *)done : (const uint32_t *)nullptr);
        hipLaunchKernelGGL(k_snr_gate, dim3(1), dim3(1), 0, s, (const RgPlan *)rgplan, plan, units, (uint64_t)a.n);
        hb_prof_end(s);
        hb_prof_begin("k_sn_dec_units", s);
        hipLaunchKernelGGL(k_sn_dec_units, dim3((unsigned)(nunits < 16384 ? nunits : 16384)), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, (const uint32_t *)units, plan);
        hb_prof_end(s);
        if (a.sym_work) {
            // ... and with the larger workspace a stream whose copies do cross them (or whose elements straddle them) still decodes in parallel: the
            // symbolic decoder of foreign blocks (hb_lz4_sym.hip) takes the verified chain -- any stream whose offsets fit 16 bits; idle when the units held
            rc = hb_launch_lz4_sym_decode(a, a.dst, a.sym_work, 0, s, RG_SNAPPY);
            if (rc) return rc;
        }
    }
    if (a.index) {
        const uint64_t units = (a.cap + HB_CHUNK - 1) / HB_CHUNK;
        const unsigned grid = (unsigned)(units < 1 ? 1 : (units < 256u * 256u ? units : 256u * 256u));
        hb_prof_begin("k_sn_dec_indexed", s);
        hipLaunchKernelGGL(k_sn_dec_indexed, dim3(grid), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, a.index, plan);
        hb_prof_end(s);
    }
    hb_prof_begin("k_sn_dec_serial", s);
    hipLaunchKernelGGL(k_sn_dec_serial, dim3(1), dim3(64), 0, s, a.src, (uint64_t)a.n, a.dst, (uint64_t)a.cap, plan, a.result, a.frame, a.expect);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
