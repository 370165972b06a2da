// hb_lz4_enc.hip — LZ4 block encoder for gfx950: replaces lz4Codec.Compress (codec.go:63-75, i.e.
// lz4.CompressBlock of pierrec/lz4 v4.1.23) on the device.
//
// The reference decodes a frame payload with ONE UncompressBlock call (codec.go:79), so the output has to
// be one spec-valid LZ4 block.  A block is a serial chain of sequences; it is built in parallel like this:
//
//   k_match / k_match_fused<TS> : one wavefront per 4 KiB chunk, chunk image resident in LDS (k_match_fused builds
//              the image straight from the un-filtered source: byte shuffle fused, no filtered buffer in HBM).
//              Every step the 64 lanes look at 64 consecutive positions: 12 bytes at the position and 12 at the
//              table candidate (aligned dwords + v_alignbyte: a misaligned ds_read_b32 is replayed), hash of the first
//              4, probe / insert into a 256-entry u16 table in LDS, verify, and -- when any lane hit -- a branch-free
//              extension to at most 20 bytes.  Positions inside a run of equal 4-grams are not inserted, so the table
//              keeps run STARTS (a candidate at the end of a run cannot be extended), and such positions fall back to
//              the offset-1 candidate, whose length comes from the wave-wide "equals the byte before" mask as far
//              as the 64-position window goes.  Greedy left-to-right selection: every hit lane computes its successor
//              (first hit at or after the end of its match; the last one points at itself), then the chain is walked
//              with one s_bitset1_b64 + one v_readlane per sequence, unrolled by 4; matches still going after 20 bytes
//              (or runs that leave the window) are extended cooperatively, 256 bytes per step.  Selected matches are
//              compacted (v_mbcnt) into an LDS queue and emitted up to 64 at a time, lane-parallel: sizes -> DPP
//              prefix sum -> every lane writes its own token / extension / literals / offset into a 768-byte staging
//              buffer, which is drained to the record with 16-byte stores after every flush (a 4 KiB record image
//              would cost the occupancy the kernel lives on: time x resident waves is constant up to ~16 waves per
//              CU; 6 KiB of LDS and 64 VGPRs give 26).  The emission state lives in LDS between flushes.  Every
//              step without a hit widens the stride by 64 bytes (LZ4's skip acceleration).  Matches never
//              leave the chunk; the last 5 bytes of a chunk stay literals and no match starts in its last 12 (LZ4
//              end-of-block rules, applied per chunk so the very last chunk satisfies them).
//              Record:  [lead literals][rest of sequence 0][sequence 1]...[sequence m-1]  (+ [trailing literals] when fused)
//              Sequence 0 has no token yet: its literal run also contains whatever the previous chunks
//              left un-matched, which only the scan knows.
//   k_tiles, k_scan : an associative scan over chunk summaries (has-match, first match position F, bytes,
//              last match end E) gives every chunk the stream offset of its segment and the length of the
//              literal run that closes it; a suffix scan gives the position of the NEXT match (the header
//              of a literal run depends on its total length, which is only known at its end).
//   k_stitch : every chunk writes exactly the bytes it owns in the final block: its segment header
//              (token + 255-extension), its encoded bytes (from the record), and its trailing literals
//              (from the source) at their place inside the literal run that a later chunk (or the end of
//              the block) closes.  Also the memcpy fallback (blosc.go:342-345) and the restart index.
//
// Algorithmic HBM bytes: n (read) + C (write).  Extra traffic: records (~C written + read) and the literal
// bytes of the source read a second time by k_stitch; see DESIGN.md.
#include "hb_lz4.h"
#include <cstdlib>
#include <vector>
#include <type_traits>

#ifndef HLOG_HC
#define HLOG_HC 8
#endif
#ifndef ENC_HLOG
#define ENC_HLOG 8
#endif
#ifndef ENC_KEY
#define ENC_KEY 5           // bytes of a position that key the LZ4 matcher's table (5 or 6)
#endif
#define HLOGW(WAYS) ((WAYS) == 1 ? ENC_HLOG : HLOG_HC)      // log2 of the hash table's buckets: LZ4 / LZ4HC
#define HSIZEW(WAYS) (1u << HLOGW(WAYS))
#define LITCAP 32u          // literal runs up to this long are copied by the owning lane, longer ones by the wave

struct __attribute__((aligned(16))) ChunkDesc {
    uint32_t lead;      // literals before the first match (chunk-relative)
    uint32_t enc_len;   // record bytes (lead literals + encoded sequences)
    uint32_t last_end;  // chunk-relative end of the last match; 0 = chunk has no match
    uint32_t mcode0;    // match-length nibble of sequence 0
};

// summary of a run of chunks; combine() is associative (not commutative)
struct Agg {
    int64_t fixed;      // stream bytes of all segments in the run, minus [ext(F - a) - a] of the first one
    uint32_t F;         // position of the first match in the run
    uint32_t E;         // end of the last match in the run
    uint32_t has;       // run contains a match
    uint32_t pad;
};

__device__ __forceinline__ Agg agg_identity() { Agg r; r.fixed = 0; r.F = 0; r.E = 0; r.has = 0; r.pad = 0; return r; }
__device__ __forceinline__ Agg agg_combine(const Agg &x, const Agg &y) {
    if (!y.has) return x;
    if (!x.has) return y;
    Agg r;
    r.has = 1; r.pad = 0; r.F = x.F; r.E = y.E;
    r.fixed = x.fixed + y.fixed + (int64_t)lz4_ext_bytes(y.F - x.E) - (int64_t)x.E;
    return r;
}
__device__ __forceinline__ Agg agg_of_chunk(const ChunkDesc &d, uint32_t start) {
    Agg r = agg_identity();
    if (d.last_end) {
        r.has = 1; r.F = start + d.lead; r.E = start + d.last_end;
        r.fixed = 1 + (int64_t)start + (int64_t)d.enc_len;     // segment = 1 + ext(F-a) + (start-a) + enc_len
    }
    return r;
}
// stream bytes of a whole prefix summarised by p, the stream starting at position 0
__device__ __forceinline__ int64_t agg_bytes(const Agg &p) { return p.has ? p.fixed + (int64_t)lz4_ext_bytes(p.F) : 0; }

struct EncPlan {
    uint64_t cbytes_block;   // LZ4 block bytes
    uint32_t use_memcpy;
    uint32_t nchunks;
    uint64_t index_off;      // byte offset of the index from the frame start (0: none / external buffer)
    uint64_t pad[4];
};

// ---- batches of frames in ONE set of launches (hb_compress_frames_batch_dev; SURVEY §8 f1 "frame batches") ----
// Chunks are independent and the scan is per frame, so K frames are K segments of one flat chunk space: global chunk g belongs to
// frame chunk_frame[g] and is that frame's chunk g - chunk0; scan tiles never span two frames (tile_frame[t]); descriptors, records
// and tile summaries are indexed globally, positions and stream offsets stay frame-local.  The single-frame launches pass bf = NULL
// and compile to what they were.
struct BatchFrame {
    const uint8_t *src;          // what the matcher reads: the filtered bytes, or the raw input when the filter is fused
    uint8_t *dst;                // frame start
    const uint8_t *memcpy_src;   // what a memcpy frame stores (NULL: the gated batch filter writes the payload)
    hb_result *result;
    EncPlan *plan;
    uint64_t n;
    uint32_t chunk0, nchunks;    // first global chunk / chunks of this frame
    uint32_t tile0, ntiles;      // first global scan tile / tiles of this frame
    uint32_t nblk, pad;          // fused byte shuffle: element blocks of the frame (nchunks = nblk * typesize)
};

// workspace layout
struct EncLayout {
    size_t plan, desc, tile_agg, tile_nf, tile_pre, tile_suf, records, total;
    uint32_t nchunks, ntiles;
};
static inline EncLayout enc_layout(size_t n) {
    EncLayout L;
    L.nchunks = (uint32_t)((n + HB_CHUNK - 1) / HB_CHUNK);
    L.ntiles = (L.nchunks + HB_TILE_CHUNKS - 1) / HB_TILE_CHUNKS;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.plan = take(sizeof(EncPlan));
    L.desc = take((size_t)L.nchunks * sizeof(ChunkDesc));
    L.tile_agg = take((size_t)L.ntiles * sizeof(Agg));
    L.tile_nf = take((size_t)L.ntiles * 4);
    L.tile_pre = take((size_t)L.ntiles * sizeof(Agg));
    L.tile_suf = take((size_t)L.ntiles * 4);
    L.records = take((size_t)L.nchunks * HB_RSTRIDE + 256);
    L.total = o;
    return L;
}
size_t hb_lz4_enc_workspace(size_t n) { return enc_layout(n).total; }

// ----------------------------------------------------------------------------------------------
// k_match
// ----------------------------------------------------------------------------------------------
// 4 bytes at any byte address of an LDS array whose base is 4-byte aligned: two aligned dwords + v_alignbyte.
// (gfx950 accepts a misaligned ds_read_b32, but replays it: measured 35 % slower for this kernel.)
__device__ __forceinline__ uint32_t lds_read4(const uint8_t *base, uint32_t a) {
    const uint32_t *w = (const uint32_t *)base;
    const uint32_t w0 = w[a >> 2], w1 = w[(a >> 2) + 1];
    return __builtin_amdgcn_alignbyte(w1, w0, a & 3u);
}

// exact-length LDS -> LDS copy in 8/4/2/1-byte pieces (ranges do not overlap)
__device__ __forceinline__ void lds_copy_exact(uint8_t *d, const uint8_t *s, uint32_t len) {
    uint32_t k = 0;
    for (; k + 8u <= len; k += 8u) ((hb_u64u *)(d + k))->v = ((const hb_u64u *)(s + k))->v;
    if (len & 4u) { ((hb_u32u *)(d + k))->v = ((const hb_u32u *)(s + k))->v; k += 4u; }
    if (len & 2u) { ((hb_u16u *)(d + k))->v = ((const hb_u16u *)(s + k))->v; k += 2u; }
    if (len & 1u) d[k] = s[k];
}

#ifndef ENC_EXT_GATE
#define ENC_EXT_GATE 1      // bytes 12..19 of a candidate are only compared when some lane of the step still matches at 12
#endif
#ifndef ENC_HASH24
#define ENC_HASH24 1
#endif
#ifndef ENC_NOMUL
#define ENC_NOMUL 1
#endif
#ifndef ENC_INNER_STEP
#define ENC_INNER_STEP 1
#endif
#ifndef ENC_WAVES
#define ENC_WAVES 6         // waves per SIMD the LZ4 matchers are compiled for (measured with 5.5 KiB of LDS per wave: 6 -> 1.39 ms, 7 -> 1.42, 8 -> 1.44)
#endif
#ifndef ENC_GATE_ADAPT
#define ENC_GATE_ADAPT 0    // 1: the verdict / hold / run-start-insert protection of the run step (measured: costs the step loop 0.13 ms of the 0.22 the run step saves)
#endif
#ifndef ENC_GATE_MAXRUNS
#define ENC_GATE_MAXRUNS 7  // a run step takes at most this many runs: a window of eight or more short runs goes through the table, which sees periodic
                            // patterns of short runs (the low mantissa plane of a float ramp: runs of 8 that repeat every 1.3 KB -- 2.7x larger as runs)
#endif
#ifndef ENC_GATE_HOLD
#define ENC_GATE_HOLD 3     // run steps a full step's verdict allows (modelled: 3 keeps the far end of a float ramp at its ratio, 7 costs it 20 %)
#endif
#ifndef ENC_RUN_GATE
#define ENC_RUN_GATE 48     // run step (match_chunk): taken when this many of a window's 64 bytes equal the byte before them; 0 = never
#endif
#ifdef ENC_DEBUG_TIMES     /* lab build only (tools/lab): clocks of the matcher's phases, summed over all wavefronts, per byte plane */
__device__ unsigned long long g_enc_t[8][16];
extern "C" int hblab_enc_times(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_t), sizeof(g_enc_t)) != hipSuccess) return -1;
    if (reset) { static unsigned long long z[8][16]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_enc_t), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define DBG_CLK() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define DBG_ADD(k, v) do { dbg_acc[k] += (unsigned long long)(v); } while (0)
#define DBG_DECL() unsigned long long dbg_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define DBG_COMMIT() do { if (lane == 0) { for (int k_ = 0; k_ < 14; k_++) if (dbg_acc[k_]) atomicAdd(&g_enc_t[dbg_plane & 7][k_], dbg_acc[k_]); } } while (0)
#else
#define DBG_DECL() do { } while (0)
#define DBG_COMMIT() do { } while (0)
#define DBG_CLK() 0ull
#define DBG_ADD(k, v) do { } while (0)
#endif
#define QCAP 80             // sequence queue slots: flushed once 64 are queued, and a step adds at most 16 (matches are >= 4 bytes)
#define SOUT 768u           // bytes of record staging per wave: drained to the record in HBM after every flush (LDS per wave
                            // sets the number of resident waves, and the matcher is latency-bound: time ~ 1 / waves)

// Table index of the FIVE bytes at a position (v = bytes 0..3, v4 = bytes 4..7).  LZ4 (WAYS == 1): two 24-bit multiplies -- bytes 0..2 and
// bytes 2..4 -- added, top byte of the sum: four full-rate instructions (a 32-bit v_mul_lo issues at a quarter of the rate; the step loop is
// bound by instruction issue).  LZ4HC keeps round 2's multiplicative hash (its tables are larger than 256 buckets).
template <int WAYS>
__device__ __forceinline__ uint32_t enc_hash(const uint32_t v, const uint32_t v4) {
#if ENC_HASH24
    if constexpr (WAYS == 1 && ENC_HLOG == 8 && ENC_KEY == 5) {
        const uint32_t z = __builtin_amdgcn_alignbyte(v4, v, 2u);                   // bytes 2..5
        return ((uint32_t)__umul24(v, 0xF85117u) + (uint32_t)__umul24(z, 0xE01E5Bu)) >> 24;   // (this HIP declares __umul24 as int: without the casts the shift is arithmetic)
    }
#endif
    return ((v + (v4 & (ENC_KEY == 6 && WAYS == 1 ? 0xFFFFu : 255u)) * 0x50505u) * 2246822519u) >> (32 - HLOGW(WAYS));
}

// the same index for a position whose five bytes are all the byte b (b4 = that byte four times)
template <int WAYS>
__device__ __forceinline__ uint32_t enc_hash_run(const uint32_t b4) {
#if ENC_HASH24
    if constexpr (WAYS == 1 && ENC_HLOG == 8 && ENC_KEY == 5)
        return ((uint32_t)__umul24(b4, 0xF85117u) + (uint32_t)__umul24(b4, 0xE01E5Bu)) >> 24;
#endif
    return enc_hash<WAYS>(b4, b4);
}

// One chunk, one wavefront.  The chunk image is in LDS (byte i of the chunk at s_data[sh + i]); s_out / s_tab /
// s_q are this wave's scratch; the record goes to `rec`, the summary to *dsc.  with_trailing: also append the
// un-matched tail of the chunk to the record (fused filter: there is no filtered buffer in HBM for k_stitch to
// take literals from; a chunk without any match then stores its whole image).
//
// WAYS: candidates per hash bucket.  1 = the LZ4 matcher (lz4.CompressBlock's single most-recent candidate, codec.go:63-75).
//       2 / 4 = the LZ4HC levels (codec.go:96-106 maps level 1-3 / 4-5 / 6-7 / 8-9 to CompressBlockHC Level1 / 5 / 7 / 9, i.e. to
//       deeper searches): the bucket keeps the last WAYS positions of its hash, every one is verified and extended to 20 bytes,
//       the longest wins.
// SNAPPY: emit Snappy elements (codec.go:228-244 -> snappy.Encode) instead of LZ4 sequences; the record is then the chunk's
//       complete, self-contained element stream (a Snappy block is a plain concatenation of elements: nothing to stitch).
// accel: bytes by which a step without any hit widens the stride (LZ4-style skip acceleration; Options.Level as a speed knob:
//       128 for levels 1-3, 64 for 4-6 (the default level 5), 0 for 7-9 and for every LZ4HC level).
template <int WAYS, int MODE>
__device__ __forceinline__ void match_chunk(const uint8_t *s_data, const uint32_t sh, const int len,
                                            uint8_t *s_out /* SOUT + 16 */, uint16_t *s_tab, uint2 *s_q /* {pos | len << 16, offset} */, uint32_t *s_st /* 8 words */,
                                            ChunkDesc *dsc, uint8_t *rec, const bool with_trailing, const bool keep_long, const int accel, const int lane,
                                            const int dbg_plane = 0) {
    constexpr bool SNAPPY = MODE == 1;        // MODE 0: LZ4 sequences for k_stitch, 1: Snappy elements, 2: a self-contained LZ4 block per chunk
    constexpr bool SELF = MODE == 2;
    {
        {
            u32x4 z; z.x = 0; z.y = 0; z.z = 0; z.w = 0;
            for (uint32_t i = lane; i < HSIZEW(WAYS) * WAYS * 2 / 16; i += 64) ((u32x4 *)s_tab)[i] = z;
            if (lane < 2) ((u32x4 *)s_st)[lane] = z;
        }
        wave_sync();

        int pos = 0, anchor = 0, nq = 0, miss = 0, gate_left = 0;
#ifdef LAB_ENC
        unsigned long long lab_dummy = 0;
#endif
        // Emission state.  It changes once per flush (every ~64 sequences), so between flushes it lives in LDS (s_st): as
        // loop-carried scalars each of these costs the step loop two register copies per step.
        int nseq = 0;                                     // sequences emitted so far
        uint32_t rec_done = 0, opend = 0;                 // record bytes already in HBM (multiple of 16) / pending in s_out
        uint32_t batch_anchor = 0, lead = 0, mcode0 = 0;
        auto ld_state = [&]() __attribute__((always_inline)) {
            const u32x4 a = ((const u32x4 *)s_st)[0], b = ((const u32x4 *)s_st)[1];
            nseq = (int)__builtin_amdgcn_readfirstlane(a.x); batch_anchor = __builtin_amdgcn_readfirstlane(a.y);
            lead = __builtin_amdgcn_readfirstlane(a.z); mcode0 = __builtin_amdgcn_readfirstlane(a.w);
            rec_done = __builtin_amdgcn_readfirstlane(b.x); opend = __builtin_amdgcn_readfirstlane(b.y);
        };
        auto st_state = [&]() __attribute__((always_inline)) {
            if (lane == 0) {
                u32x4 a, b;
                a.x = (uint32_t)nseq; a.y = batch_anchor; a.z = lead; a.w = mcode0; b.x = rec_done; b.y = opend; b.z = 0; b.w = 0;
                ((u32x4 *)s_st)[0] = a; ((u32x4 *)s_st)[1] = b;
            }
            wave_sync();
        };
        const int mstart_max = len - 12;                  // last position a match may start at
        const int mend_max = len - 5;                     // matches end at or before this
        const uint8_t *data = s_data + sh;
#define RD4(x) lds_read4(s_data, sh + (uint32_t)(x))

        // write the complete 16-byte blocks of s_out to the record and keep the tail (all: the tail too, padded)
        auto drain = [&](const bool all) __attribute__((always_inline)) {
            wave_sync();
            const uint32_t full = all ? ((opend + 15u) & ~15u) : (opend & ~15u);
            if (full) {
                for (uint32_t i = lane * 16u; i < full; i += 1024u) *(u32x4 *)(rec + rec_done + i) = *(const u32x4 *)(s_out + i);
                if (!all && lane == 0) { const u32x4 t = *(const u32x4 *)(s_out + full); *(u32x4 *)s_out = t; }
                rec_done += full;
                opend = all ? 0u : opend - full;
                wave_sync();
            }
        };
        // wave copy of `cnt` chunk bytes from position `from` into the record, through s_out
        auto stream_literals = [&](uint32_t from, uint32_t cnt) __attribute__((always_inline)) {
            while (cnt) {
                const uint32_t piece = cnt < SOUT - opend ? cnt : SOUT - opend;
                for (uint32_t k = lane * 4u; k < piece; k += 256u) {
                    if (k + 4u <= piece) ((hb_u32u *)(s_out + opend + k))->v = RD4(from + k);
                    else for (uint32_t r = k; r < piece; r++) s_out[opend + r] = data[from + r];
                }
                opend += piece; from += piece; cnt -= piece;
                drain(false);
            }
        };

        // emit queued sequences, one per lane: as many of the first min(nq, 64) as fit into s_out
        auto flush = [&]() __attribute__((always_inline)) {
            ld_state();
            const int take = nq < 64 ? nq : 64;
            const uint2 e = s_q[lane];
            const uint32_t q_mp = e.x & 0xFFFFu, q_ml = e.x >> 16, q_off = e.y;
            const uint32_t end = q_mp + q_ml;
            const uint32_t prev = wave_shr1(end, batch_anchor);
            const uint32_t lit = q_mp - prev, mcode = q_ml - 4u;
            const bool first = !SELF && (nseq == 0) && lane == 0;  // sequence 0 of the chunk: token comes from k_stitch (not when the chunk is a block of its own)
            const uint32_t nbl = first ? 0u : lz4_ext_bytes16(lit), nbm = lz4_ext_bytes16(mcode);      // (both below 4096 + 15: chunk-local)
            const uint32_t size = lane < take ? ((first ? 0u : 1u) + nbl + lit + 2u + nbm) : 0u;
            const uint32_t incl = wave_incl_scan_dpp(size);
            int cnt = __builtin_popcountll(hb_ballot(lane < take && incl <= SOUT - opend));
            // what is queued beyond the sequences emitted now is read BEFORE anything is staged: the queue lives in the staging buffer's bytes
            // (s_q = s_out + 16: the two are never in use at the same time, and 640 bytes less LDS per wave are three more waves per CU)
            const uint32_t i0 = (uint32_t)(lane + (cnt ? cnt : 1)), i1 = i0 + 64u;
            const uint2 r0 = s_q[i0 < QCAP ? i0 : 0], r1 = s_q[i1 < QCAP ? i1 : 0];
            wave_sync();
            if (nseq == 0) {
                lead = __builtin_amdgcn_readlane(lit, 0);
                const uint32_t m0 = __builtin_amdgcn_readlane(mcode, 0);
                mcode0 = m0 < 15u ? m0 : 15u;
            }
            if (cnt == 0) {
                // the first queued sequence alone is larger than the staging buffer (a literal run of > 1 KiB):
                // header, then the literals in pieces, then offset + match extension
                cnt = 1;
                if (lane == 0) {
                    uint32_t q = opend;
                    if (!first) {
                        s_out[q++] = (uint8_t)((15u << 4) | (mcode < 15u ? mcode : 15u));
                        for (uint32_t k = 0; k + 1 < nbl; k++) s_out[q++] = 255;
                        s_out[q++] = (uint8_t)((lit - 15u) - 255u * (nbl - 1));
                    }
                }
                opend += (uint32_t)__builtin_amdgcn_readlane((first ? 0u : 1u) + nbl, 0);
                wave_sync();
                stream_literals(__builtin_amdgcn_readlane(prev, 0), __builtin_amdgcn_readlane(lit, 0));
                if (lane == 0) {
                    uint32_t q = opend;
                    ((hb_u16u *)(s_out + q))->v = (uint16_t)q_off;
                    q += 2;
                    if (nbm) {
                        for (uint32_t k = 0; k + 1 < nbm; k++) s_out[q++] = 255;
                        s_out[q++] = (uint8_t)((mcode - 15u) - 255u * (nbm - 1));
                    }
                }
                opend += 2u + (uint32_t)__builtin_amdgcn_readlane(nbm, 0);
            } else {
                const bool act = lane < cnt;
                uint32_t q = opend + incl - size;
                const uint32_t litdst = q + (first ? 0u : 1u + nbl);
                // (measured and not kept: runs of 1..8 literals copied blind -- one unaligned 8-byte read, one or two unaligned 4-byte stores in front
                // of the token / offset stores -- instead of the exact-length pieces: 1.405 against 1.383 ms, the replayed unaligned accesses cost
                // what the saved exec-mask branches gain)
                if (act && lit <= LITCAP) lds_copy_exact(s_out + litdst, data + prev, lit);
                if (act) {
                    if (!first) {
                        s_out[q++] = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (mcode < 15u ? mcode : 15u));
                        if (nbl) {
                            for (uint32_t k = 0; k + 1 < nbl; k++) s_out[q++] = 255;
                            s_out[q++] = (uint8_t)((lit - 15u) - 255u * (nbl - 1));
                        }
                    }
                    q += lit;
                    ((hb_u16u *)(s_out + q))->v = (uint16_t)q_off;
                    q += 2;
                    if (nbm) {
                        for (uint32_t k = 0; k + 1 < nbm; k++) s_out[q++] = 255;
                        s_out[q++] = (uint8_t)((mcode - 15u) - 255u * (nbm - 1));
                    }
                }
                unsigned long long lm = hb_ballot(act && lit > LITCAP);
                while (lm) {                                   // long literal runs: the whole wave copies
                    const int l = __builtin_ctzll(lm);
                    const uint32_t s = __builtin_amdgcn_readlane(prev, l), dq = __builtin_amdgcn_readlane(litdst, l);
                    const uint32_t ln = __builtin_amdgcn_readlane(lit, l);
                    for (uint32_t k = lane * 4u; k < ln; k += 256u) {
                        if (k + 4u <= ln) ((hb_u32u *)(s_out + dq + k))->v = ((const hb_u32u *)(data + s + k))->v;
                        else for (uint32_t r = k; r < ln; r++) s_out[dq + r] = data[s + r];
                    }
                    lm &= lm - 1;
                }
                opend += (uint32_t)__builtin_amdgcn_readlane(incl, cnt - 1);
            }
            batch_anchor = __builtin_amdgcn_readlane(end, cnt - 1);
            nseq += cnt;
            drain(false);
            nq -= cnt;
            if (lane < nq) s_q[lane] = r0;
            if (lane + 64 < nq) s_q[lane + 64] = r1;
            st_state();
        };

        // ---- Snappy emitter (SNAPPY): [literal element][copy elements] per queued sequence; format_description.txt of Snappy:
        // literal tag = (len-1) << 2 | 0 (len-1 >= 60: 60 / 61 = one / two extra length bytes), copy with 1-byte offset
        // tag = (off >> 8) << 5 | (len-4) << 2 | 1 (len 4..11, off < 2048), copy with 2-byte offset tag = (len-1) << 2 | 2 (len 1..64).
        // Long matches are cut as the reference encoder's emitCopy cuts them: 64-byte copies while >= 68 remain, then 60 if > 64.
        auto sn_lit_header = [&](uint32_t q, uint32_t lit) __attribute__((always_inline)) -> uint32_t {     // lit >= 1
            const uint32_t x = lit - 1u;
            if (x < 60u) { s_out[q] = (uint8_t)(x << 2); return 1u; }
            if (x < 256u) { s_out[q] = (uint8_t)(60u << 2); s_out[q + 1] = (uint8_t)x; return 2u; }
            s_out[q] = (uint8_t)(61u << 2); s_out[q + 1] = (uint8_t)x; s_out[q + 2] = (uint8_t)(x >> 8); return 3u;
        };
        auto sn_copies = [&](uint32_t q, uint32_t ml, uint32_t off) __attribute__((always_inline)) -> uint32_t {
            const uint32_t q0 = q;
            while (ml >= 68u) { s_out[q] = (uint8_t)((63u << 2) | 2u); s_out[q + 1] = (uint8_t)off; s_out[q + 2] = (uint8_t)(off >> 8); q += 3u; ml -= 64u; }
            if (ml > 64u) { s_out[q] = (uint8_t)((59u << 2) | 2u); s_out[q + 1] = (uint8_t)off; s_out[q + 2] = (uint8_t)(off >> 8); q += 3u; ml -= 60u; }
            if (ml >= 12u || off >= 2048u) { s_out[q] = (uint8_t)(((ml - 1u) << 2) | 2u); s_out[q + 1] = (uint8_t)off; s_out[q + 2] = (uint8_t)(off >> 8); q += 3u; }
            else { s_out[q] = (uint8_t)(((off >> 8) << 5) | ((ml - 4u) << 2) | 1u); s_out[q + 1] = (uint8_t)off; q += 2u; }
            return q - q0;
        };
        auto flush_sn = [&]() __attribute__((always_inline)) {
            ld_state();
            const int take = nq < 64 ? nq : 64;
            const uint2 e = s_q[lane];
            const uint32_t q_mp = e.x & 0xFFFFu, q_ml = e.x >> 16, q_off = e.y;
            const uint32_t end = q_mp + q_ml;
            const uint32_t prev = wave_shr1(end, batch_anchor);
            const uint32_t lit = q_mp - prev;
            const uint32_t lh = lit == 0u ? 0u : (lit <= 60u ? 1u : (lit <= 256u ? 2u : 3u));
            const uint32_t n64 = q_ml >= 68u ? (q_ml - 68u) / 64u + 1u : 0u;
            const uint32_t rem = q_ml - 64u * n64;                 // 4..67
            const uint32_t r60 = rem > 64u ? 1u : 0u;
            const uint32_t fin = rem - 60u * r60;                  // 4..64
            const uint32_t csz = 3u * (n64 + r60) + ((fin >= 12u || q_off >= 2048u) ? 3u : 2u);
            const uint32_t size = lane < take ? lh + lit + csz : 0u;
            const uint32_t incl = wave_incl_scan_dpp(size);
            int cnt = __builtin_popcountll(hb_ballot(lane < take && incl <= SOUT - opend));
            const uint32_t i0 = (uint32_t)(lane + (cnt ? cnt : 1)), i1 = i0 + 64u;     // (as in flush(): the queue shares the staging buffer's bytes)
            const uint2 r0 = s_q[i0 < QCAP ? i0 : 0], r1 = s_q[i1 < QCAP ? i1 : 0];
            wave_sync();
            if (cnt == 0) {                                        // a literal run longer than the staging buffer: header, pieces, copies
                cnt = 1;
                uint32_t hl = 0;
                if (lane == 0) hl = sn_lit_header(opend, lit);
                opend += (uint32_t)__builtin_amdgcn_readlane(hl, 0);
                wave_sync();
                stream_literals(__builtin_amdgcn_readlane(prev, 0), __builtin_amdgcn_readlane(lit, 0));
                uint32_t cl = 0;
                if (lane == 0) cl = sn_copies(opend, q_ml, q_off);
                opend += (uint32_t)__builtin_amdgcn_readlane(cl, 0);
            } else {
                const bool act = lane < cnt;
                uint32_t q = opend + incl - size;
                uint32_t litdst = 0;
                if (act) {
                    if (lit) q += sn_lit_header(q, lit);
                    litdst = q;
                    if (lit <= LITCAP) lds_copy_exact(s_out + q, data + prev, lit);
                    q += lit;
                    sn_copies(q, q_ml, q_off);
                }
                unsigned long long lm = hb_ballot(act && lit > LITCAP);
                while (lm) {                                       // long literal runs: the whole wave copies
                    const int l = __builtin_ctzll(lm);
                    const uint32_t sp = __builtin_amdgcn_readlane(prev, l), dq = __builtin_amdgcn_readlane(litdst, l);
                    const uint32_t ln = __builtin_amdgcn_readlane(lit, l);
                    for (uint32_t k = lane * 4u; k < ln; k += 256u) {
                        if (k + 4u <= ln) ((hb_u32u *)(s_out + dq + k))->v = ((const hb_u32u *)(data + sp + k))->v;
                        else for (uint32_t r = k; r < ln; r++) s_out[dq + r] = data[sp + r];
                    }
                    lm &= lm - 1;
                }
                opend += (uint32_t)__builtin_amdgcn_readlane(incl, cnt - 1);
            }
            batch_anchor = __builtin_amdgcn_readlane(end, cnt - 1);
            nseq += cnt;
            drain(false);
            nq -= cnt;
            if (lane < nq) s_q[lane] = r0;
            if (lane + 64 < nq) s_q[lane + 64] = r1;
            st_state();
        };
        DBG_DECL();
        unsigned long long dbg_flush = 0;
        auto flush_any = [&]() __attribute__((always_inline)) {
            const unsigned long long c0 = DBG_CLK();
            if constexpr (SNAPPY) flush_sn(); else flush();
            const unsigned long long c1 = DBG_CLK();
            dbg_flush += c1 - c0; DBG_ADD(8, c1 - c0); DBG_ADD(9, 1);
        };
        (void)dbg_flush;
        const unsigned long long dbg_t0 = DBG_CLK(); (void)dbg_t0;

        // One step = 64 positions.  The body exists twice: INNER for the steps whose positions and 20-byte extensions all lie inside the chunk
        // (all but the first and the last one or two of a chunk: no validity masks, no clamping against the end), and the general one.
        auto step = [&](auto inner_tag) __attribute__((always_inline)) {
            constexpr bool INNER = decltype(inner_tag)::value;
            const unsigned long long dbg_s0 = DBG_CLK(); const unsigned long long dbg_f0 = dbg_flush; (void)dbg_s0; (void)dbg_f0;
            const int p = pos + lane;
            // ---- 12 bytes at my position: 4 aligned dwords + v_alignbyte (one LDS round trip) ----
            const uint32_t ap = sh + (uint32_t)p;
            const uint32_t *wp = (const uint32_t *)s_data + (ap >> 2);
            const uint32_t p0 = wp[0], p1 = wp[1], p2 = wp[2], p3 = wp[3];
            const uint32_t prevb = s_data[sh + (uint32_t)(pos > 0 ? pos - 1 : 0)];   // wave-uniform: byte before lane 0
            const uint32_t v = __builtin_amdgcn_alignbyte(p1, p0, ap & 3u);
            const uint32_t v4 = __builtin_amdgcn_alignbyte(p2, p1, ap & 3u);
            const uint32_t v8 = __builtin_amdgcn_alignbyte(p3, p2, ap & 3u);
            // predicates are kept as wave masks (one v_cmp each, combined on the scalar side): a ballot of a compound
            // bool would first be materialised per lane
            const bool valid = INNER || p <= mstart_max;       // INNER: a step whose 64 positions and their 20-byte extensions lie inside the chunk
            // inside a run of equal 4-grams <=> data[p-1] == data[p] == ... == data[p+3]
            const uint32_t vb = v & 255u;
            const uint32_t before = wave_shr1(vb, prevb);
#if ENC_NOMUL
            const uint32_t b4 = __builtin_amdgcn_perm(v, v, 0u);                   // byte 0 four times (a 32-bit multiply issues at a quarter of the rate)
#else
            const uint32_t b4 = vb * 0x01010101u;
#endif
            const unsigned long long m_first = INNER || pos > 0 ? ~0ull : ~1ull;   // p >= 1
            // bit l: data[pos+l] == data[pos+l-1]; positions past the end of the chunk hold whatever was in LDS: masked out
            const unsigned long long m_eqprev = hb_ballot(before == vb) & (INNER || len - pos >= 64 ? ~0ull : ((1ull << (len - pos)) - 1ull));
            const unsigned long long m_rle = hb_ballot(valid) & hb_ballot(v == b4) & m_eqprev & m_first;
            const bool rle = valid && (INNER || pos > 0 || lane > 0) && v == b4 && before == vb;
            bool runny = false; (void)runny;
#if ENC_RUN_GATE
            if constexpr (WAYS == 1) {
                // ---- RUN STEP.  A window in which >= ENC_RUN_GATE of the 64 bytes equal the byte before them is made of runs (a byte plane of
                // slowly varying values): its matches are the runs themselves (offset 1), and the wave-wide mask already says where every run
                // starts and ends -- no hash, no table probe, no candidate read, no extension, no selection walk: the selected matches are the
                // first lanes of the groups of consecutive run lanes.  Modelled in tests/tools/gpu_lz4_model2.c: on the headline data's plane 2
                // 51 of 55 hit steps per chunk are such steps and the ratio does not move (0.5293 -> 0.5301; all four planes 0.5169 -> 0.5161).
                // The gate is only open for the ENC_GATE_HOLD steps behind a FULL hit step whose greedy parse needed no fewer sequences than the
                // window has runs: where one table match spans many short runs (a periodic pattern: the low mantissa plane of a ramp, 2.7x
                // larger with runs alone) the full steps keep it shut; and a run step still enters the first byte of every run it takes
                // into the table, so that a full step finds such repeats at all.
                // (population counts through the scalar instruction by hand: the compiler turns `popcount(mask) >= 48` into a 64-bit VECTOR compare)
                uint32_t n_eq, n_runs;
                asm("s_bcnt1_i32_b64 %0, %1" : "=s"(n_eq) : "s"(m_eqprev) : "scc");
                runny = n_eq >= (uint32_t)ENC_RUN_GATE;
                const unsigned long long sel = m_rle & ~(m_rle << 1);
                n_runs = 64u;
                if (runny) asm volatile("s_bcnt1_i32_b64 %0, %1" : "=s"(n_runs) : "s"(sel) : "scc");      // (volatile: keeps the test behind the first branch -- most steps leave there)
                if (runny && (!ENC_GATE_ADAPT || gate_left > 0) && n_runs - 1u < (uint32_t)ENC_GATE_MAXRUNS) {     // (1 .. MAXRUNS runs: sel != 0 <=> m_rle != 0)
                    gate_left--;
#if ENC_GATE_ADAPT && !defined(ENC_GATE_NOINS)
                    // (the byte in front of the first run lane starts the run: its five bytes are the run's byte, so is its key)
                    if (hb_lane_in(sel)) s_tab[enc_hash_run<WAYS>(b4)] = (uint16_t)(p - 1);
#endif
                    const unsigned long long nrun = ~(m_eqprev >> lane);
                    const uint32_t run = nrun ? (uint32_t)__builtin_ctzll(nrun) : 64u;
                    uint32_t ml = min(run, (uint32_t)(mend_max - p));
                    const int lastj = 63 - __builtin_clzll(sel);
                    const int mpj = pos + lastj, maxlj = mend_max - mpj;
                    const int runj = (int)__builtin_amdgcn_readlane(run, lastj);
                    int mlj = runj < maxlj ? runj : maxlj;
                    if (lastj + runj >= 64 && mlj < maxlj) {       // the last run reaches the end of the window: go on, 256 bytes per step
                        const uint32_t rb4 = __builtin_amdgcn_readlane(b4, lastj);
                        for (;;) {
                            const int i = mlj + 4 * lane;
                            const int ic = i < maxlj ? i : maxlj;  // keep the reads inside the chunk image
                            const uint32_t x = RD4(mpj + ic) ^ rb4;
                            int eq = x ? (__builtin_ctz(x) >> 3) : 4;
                            const int avail = maxlj - i;
                            if (avail < eq) eq = avail > 0 ? avail : 0;
                            const unsigned long long part = hb_ballot(eq != 4);
                            if (part) {
                                const int f = __builtin_ctzll(part);
                                mlj += 4 * f + (int)__builtin_amdgcn_readlane((uint32_t)eq, f);
                                break;
                            }
                            mlj += 256;
                        }
                        if (lane == lastj) ml = (uint32_t)mlj;
                    }
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
                    {   // every lane stores: the lanes that are not selected to the spare slot behind the queue (a v_cndmask on the mask itself where
                        // a predicated store costs three scalar instructions -- the step loop issues as many scalar as vector instructions)
                        uint2 e; e.x = (uint32_t)p | (ml << 16); e.y = 1u;
                        s_q[hb_select_lane(sel, (uint32_t)nq + rank, (uint32_t)QCAP)] = e;
                    }
                    nq += __builtin_popcountll(sel);
                    anchor = mpj + mlj;
                    while (nq >= 64) flush_any();
                    miss = 0;
                    pos = anchor > pos + 64 ? anchor : pos + 64;
                    DBG_ADD(2, DBG_CLK() - dbg_s0 - (dbg_flush - dbg_f0)); DBG_ADD(3, 1);
                    return;
                }
            }
#endif
            // the table is keyed by FIVE bytes (one v_and + one v_mad_u32_u24 more than a 4-byte key): the entry of a 4-gram that
            // occurs in many contexts is then not the latest of them but the latest with the same next byte -- longer matches, 20 %
            // fewer sequences and a better ratio at the same table size (modelled in tests/tools/gpu_lz4_model.c)
            const uint32_t h = enc_hash<WAYS>(v, v4);
            uint32_t cand, xa, xb, ac = 0, hc_fbit = 0;
            const uint32_t *wc = nullptr;
            uint32_t c3 = 0;
            bool hit;
            unsigned long long m_hit;
            if constexpr (WAYS == 1) {
                cand = s_tab[h];
                if constexpr (INNER) s_tab[hb_lane_in(m_rle) ? HSIZEW(WAYS) * WAYS : h] = (uint16_t)p;      // (run lanes store to the spare entry behind the table)
                else if (valid && !rle) s_tab[h] = (uint16_t)p;
                // ---- 12 bytes at the candidate (second round trip) ----
                ac = sh + cand;
                wc = (const uint32_t *)s_data + (ac >> 2);
                const uint32_t c0 = wc[0], c1 = wc[1], c2 = wc[2];
                c3 = wc[3];
                const uint32_t cv = __builtin_amdgcn_alignbyte(c1, c0, ac & 3u);
                const uint32_t cv4 = __builtin_amdgcn_alignbyte(c2, c1, ac & 3u);
                const uint32_t cv8 = __builtin_amdgcn_alignbyte(c3, c2, ac & 3u);
                m_hit = hb_ballot(valid) & hb_ballot((int)cand < p) & hb_ballot(cv == v);
                hit = valid && (int)cand < p && cv == v;
                // the offset-1 candidate of a run needs no second read: its match is the rest of the run
                xa = hit ? (cv4 ^ v4) : (b4 ^ v4); xb = hit ? (cv8 ^ v8) : (b4 ^ v8);
                cand = hit ? cand : (uint32_t)p - 1u;
            } else {
                // ---- LZ4HC: the bucket holds the last WAYS positions of this hash; the new one goes in front ----
                uint32_t cw[WAYS];
                if constexpr (WAYS == 2) {
                    uint32_t *b = (uint32_t *)s_tab + h;
                    const uint32_t w = *b;
                    cw[0] = w & 0xFFFFu; cw[1] = w >> 16;
                    if (valid && !rle) *b = (uint32_t)p | (cw[0] << 16);
                } else {
                    u32x2 *b = (u32x2 *)s_tab + h;
                    const u32x2 w = *b;
                    cw[0] = w.x & 0xFFFFu; cw[1] = w.x >> 16; cw[2] = w.y & 0xFFFFu; cw[3] = w.y >> 16;
                    if (valid && !rle) { u32x2 nw; nw.x = (uint32_t)p | (cw[0] << 16); nw.y = cw[1] | (cw[2] << 16); *b = nw; }
                }
                // 20 bytes at my position and at every candidate; the longest verified candidate wins (nearest first on ties)
                const uint32_t p4 = wp[4], p5 = wp[5];
                const uint32_t v12 = __builtin_amdgcn_alignbyte(p4, p3, ap & 3u), v16 = __builtin_amdgcn_alignbyte(p5, p4, ap & 3u);
                hit = false; cand = (uint32_t)p - 1u;
#pragma unroll
                for (int k = 0; k < WAYS; k++) {
                    const uint32_t a2 = sh + cw[k];
                    const uint32_t *w2 = (const uint32_t *)s_data + (a2 >> 2);
                    const uint32_t d0 = w2[0], d1 = w2[1], d2 = w2[2], d3 = w2[3], d4 = w2[4], d5 = w2[5];
                    const uint32_t e0 = __builtin_amdgcn_alignbyte(d1, d0, a2 & 3u), e4 = __builtin_amdgcn_alignbyte(d2, d1, a2 & 3u);
                    const uint32_t e8 = __builtin_amdgcn_alignbyte(d3, d2, a2 & 3u), e12 = __builtin_amdgcn_alignbyte(d4, d3, a2 & 3u);
                    const uint32_t e16 = __builtin_amdgcn_alignbyte(d5, d4, a2 & 3u);
                    uint32_t fa, fb, fc, fd;
                    asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(e4 ^ v4));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(e8 ^ v8));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fc) : "v"(e12 ^ v12));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fd) : "v"(e16 ^ v16));
                    const uint32_t f = min(fa, min(fb, min(fc, min(fd, 32u) + 32u) + 32u) + 32u);
                    const bool ok = valid && (int)cw[k] < p && e0 == v;
                    if (ok && (!hit || f > hc_fbit)) { hit = true; hc_fbit = f; cand = cw[k]; }
                }
                m_hit = hb_ballot(hit);
                if (!hit) {                                     // the offset-1 candidate of a run: its match is the rest of the run
                    uint32_t fa, fb, fc, fd;
                    asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(b4 ^ v4));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(b4 ^ v8));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fc) : "v"(b4 ^ v12));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fd) : "v"(b4 ^ v16));
                    hc_fbit = min(fa, min(fb, min(fc, min(fd, 32u) + 32u) + 32u) + 32u);
                }
                xa = 0; xb = 0;
            }
            unsigned long long mask = m_hit | m_rle;
#if defined(LAB_ENC) && LAB_ENC == 2       /* lab ablation: the probe alone (no extension, selection, queue); garbage frames, timing only */
            lab_dummy ^= mask; mask = 0; miss = -1;
#endif
            if (mask) {
                // An entry whose candidate still matches 12 bytes or more is put BACK: repeated content then keeps pointing
                // at its first occurrence instead of at the previous repeat, so a decoder never finds a chain of matches
                // that each copy the one before (bitshuffled integers: 27 dependency rounds per 64 tokens otherwise, 5
                // with this); the ratio does not move.  Not in the fused byte-shuffle kernels: their long matches are runs
                // (offset 1), and the extra LDS write costs their step loop 3 %.
                uint32_t fbit;
                if constexpr (WAYS == 1) {
                    if (keep_long && hit && (xa | xb) == 0u) s_tab[h] = (uint16_t)cand;
                    // every lane extends its own match to at most 20 bytes, branch-free (two more dwords on each side, read
                    // on this path only): v_ffbl_b32 gives -1 for 0, so the first differing bit of the 16 bytes is
                    // min(ffbl(xa), 32 + min(ffbl(xb), 32 + min(ffbl(xc), 32 + min(ffbl(xd), 32))))
                    uint32_t fa, fb;
                    asm("v_ffbl_b32 %0, %1" : "=v"(fa) : "v"(xa));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fb) : "v"(xb));
#if ENC_EXT_GATE
                    // bytes 12..19 are only looked at when some lane still matches at 12 (few-symbol planes: matches of 4-7 bytes, never)
                    if ((hb_ballot((xa | xb) == 0u) & mask) == 0ull) {
                        fbit = min(fa, min(fb, 32u) + 32u);
                    } else
#endif
                    {
                        const uint32_t p4 = wp[4], p5 = wp[5], c4 = wc[4], c5 = wc[5];
                        const uint32_t v12 = __builtin_amdgcn_alignbyte(p4, p3, ap & 3u), v16 = __builtin_amdgcn_alignbyte(p5, p4, ap & 3u);
                        const uint32_t cv12 = __builtin_amdgcn_alignbyte(c4, c3, ac & 3u), cv16 = __builtin_amdgcn_alignbyte(c5, c4, ac & 3u);
                        const uint32_t xc = hit ? (cv12 ^ v12) : (b4 ^ v12), xd = hit ? (cv16 ^ v16) : (b4 ^ v16);
                        uint32_t fc, fd;
                        asm("v_ffbl_b32 %0, %1" : "=v"(fc) : "v"(xc));
                        asm("v_ffbl_b32 %0, %1" : "=v"(fd) : "v"(xd));
                        fbit = min(fa, min(fb, min(fc, min(fd, 32u) + 32u) + 32u) + 32u);
                    }
                } else {
                    fbit = hc_fbit;
                }
                uint32_t ml = 4u + (fbit >> 3);
                const uint32_t maxl = (uint32_t)(mend_max - p);
                unsigned long long lmask = mask & hb_ballot(fbit == 128u);
                if constexpr (!INNER) { lmask &= hb_ballot(ml < maxl); ml = min(ml, maxl); }      // (INNER: ml <= 20 < maxl)
                if (lmask & m_rle & ~m_hit) {
                    // a run: the offset-1 match is the rest of the run, and the wave already knows where runs end as far
                    // as this window goes -- only a run that leaves the window needs the cooperative extension
                    const unsigned long long nrun = ~(m_eqprev >> lane);           // (0 for lane 0 of a window that is one run)
                    const uint32_t run = nrun ? (uint32_t)__builtin_ctzll(nrun) : 64u;
                    // a run that reaches the end of the window stops there if the next byte differs (bitshuffled data has
                    // runs that end on 32-byte window boundaries all the time)
                    const bool endstop = data[pos + 64] != data[pos + 63];
                    const bool inwin = !hit && fbit == 128u && ((uint32_t)lane + run < 64u || ((uint32_t)lane + run == 64u && endstop));
                    if (inwin) ml = min(run, maxl);
                    lmask &= ~(hb_ballot(inwin) & m_rle);
                }
                // greedy left-to-right selection: scalar walk over the hit mask
                unsigned long long sel = 0;
                int last_end = anchor;
                if (lmask == 0) {
                    // all lengths known: every hit lane computes its successor (first hit at or after the end of
                    // its match; the last one points at itself) and the walk is one bit-set + one readlane per
                    // sequence, unrolled by 4 (setting the bit of the last lane again is harmless)
                    const uint32_t x = (uint32_t)lane + ml;
                    uint32_t succ = (uint32_t)lane;
                    if (x < 64u) {
                        const unsigned long long m = mask >> x;
                        if (m) succ = x + (uint32_t)__builtin_ctzll(m);
                    }
                    uint32_t j = (uint32_t)__builtin_ctzll(mask);
                    uint32_t lastj;
                    for (;;) {
                        asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(j));
                        const uint32_t j1 = __builtin_amdgcn_readlane(succ, (int)j);
                        asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(j1));
                        const uint32_t j2 = __builtin_amdgcn_readlane(succ, (int)j1);
                        asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(j2));
                        const uint32_t j3 = __builtin_amdgcn_readlane(succ, (int)j2);
                        asm volatile("s_bitset1_b64 %0, %1" : "+s"(sel) : "s"(j3));
                        j = __builtin_amdgcn_readlane(succ, (int)j3);
                        lastj = j3;
                        if (j == j3) break;
                    }
                    last_end = pos + (int)lastj + (int)__builtin_amdgcn_readlane(ml, (int)lastj);
                    mask = 0;
                }
                while (mask) {
                    const int j = __builtin_ctzll(mask);
                    int mlj = (int)__builtin_amdgcn_readlane(ml, j);
                    if ((lmask >> j) & 1ull) {             // still matching after 20 bytes: 256 bytes per step
                        const int mp = pos + j, mc = (int)__builtin_amdgcn_readlane(cand, j);
                        const int maxl = mend_max - mp;
                        for (;;) {
                            const int i = mlj + 4 * lane;
                            const int ic = i < maxl ? i : maxl;        // keep the reads inside the chunk image
                            const uint32_t x = RD4(mp + ic) ^ RD4(mc + ic);
                            int eq = x ? (__builtin_ctz(x) >> 3) : 4;
                            const int avail = maxl - i;
                            if (avail < eq) eq = avail > 0 ? avail : 0;
                            const unsigned long long part = hb_ballot(eq != 4);
                            if (part) {
                                const int f = __builtin_ctzll(part);
                                mlj += 4 * f + (int)__builtin_amdgcn_readlane((uint32_t)eq, f);
                                break;
                            }
                            mlj += 256;
                        }
                        if (lane == j) ml = (uint32_t)mlj;
                    }
                    sel |= 1ull << j;
                    const int e = j + mlj;
                    last_end = pos + e;
                    mask = e >= 64 ? 0ull : (mask & (~0ull << e));
                }
#if ENC_RUN_GATE && ENC_GATE_ADAPT
                if constexpr (WAYS == 1) {                       // the run gate's verdict (see the run step): only a window of runs asks for one
                    if (runny) {
                        const unsigned long long rstart = m_rle & ~(m_rle << 1);
                        gate_left = (int)__builtin_popcountll(rstart) <= (int)__builtin_popcountll(sel) + 1 ? ENC_GATE_HOLD : 0;
                    }
                }
#endif
                // queue the selected matches, compacted in position order
#if defined(LAB_ENC) && LAB_ENC == 1       /* lab ablation: no queue, no emission; garbage frames, timing only */
                lab_dummy ^= sel ^ ml ^ cand;
                anchor = last_end;
#else
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(sel >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sel, 0u));
                {
                    uint2 e; e.x = (uint32_t)p | (ml << 16); e.y = (uint32_t)p - cand;
                    s_q[hb_select_lane(sel, (uint32_t)nq + rank, (uint32_t)QCAP)] = e;       // (the spare slot: see the run step)
                }
                nq += __builtin_popcountll(sel);
                anchor = last_end;
                while (nq >= 64) flush_any();
#endif
                miss = 0;
                pos = anchor > pos + 64 ? anchor : pos + 64;
                DBG_ADD(0, DBG_CLK() - dbg_s0 - (dbg_flush - dbg_f0)); DBG_ADD(1, 1);
            } else {
                miss++;
                const int nxt = pos + 64 + miss * accel;        // every step without a hit widens the stride by `accel` bytes
                pos = anchor > nxt ? anchor : nxt;
                DBG_ADD(4, DBG_CLK() - dbg_s0); DBG_ADD(5, 1);
            }
                };
        if (ENC_INNER_STEP) {
            if (pos <= mstart_max) step(std::false_type{});                              // (pos == 0)
            while (pos + 80 <= mstart_max) step(std::true_type{});                       // (pos > 0 from here on: a step advances by at least 64)
        }
        while (pos <= mstart_max) step(std::false_type{});
        const unsigned long long dbg_t1 = DBG_CLK(); (void)dbg_t1;
        DBG_ADD(6, dbg_t1 - dbg_t0); DBG_ADD(7, 1);                 // the whole step loop (incl. its flushes) / chunks
        while (nq > 0) flush_any();
        ld_state();
#ifdef LAB_ENC
        if (lab_dummy == 0x1234567ull) s_out[lane] = 1;     // keeps the ablated computations alive
#endif
        if constexpr (SNAPPY) {
            // the rest of the chunk is one literal element; the record is the chunk's complete element stream
            const uint32_t tl = (uint32_t)(len - anchor);
            if (tl) {
                uint32_t hl = 0;
                if (lane == 0) hl = sn_lit_header(opend, tl);
                opend += (uint32_t)__builtin_amdgcn_readlane(hl, 0);
                wave_sync();
                stream_literals((uint32_t)anchor, tl);
            }
            const uint32_t total = rec_done + opend;
            drain(true);
            if (lane == 0) {
                ChunkDesc d;
                d.lead = 0u; d.enc_len = total; d.last_end = (uint32_t)len; d.mcode0 = 0u;
                *dsc = d;
            }
            wave_sync();
            return;
        }
        if constexpr (SELF) {
            // the chunk is an LZ4 block of its own (a stream of a C-Blosc-1 frame, hb_cblosc.hip): the rest of the chunk is the final,
            // literal-only sequence (never empty: the last 5 bytes of a chunk are literals, a chunk below 13 bytes has no match)
            const uint32_t tl = (uint32_t)(len - anchor);
            const uint32_t nb = lz4_ext_bytes(tl);
            if (lane == 0) {
                uint32_t q = opend;
                s_out[q++] = (uint8_t)((tl < 15u ? tl : 15u) << 4);
                if (nb) { for (uint32_t k = 0; k + 1 < nb; k++) s_out[q++] = 255; s_out[q++] = (uint8_t)((tl - 15u) - 255u * (nb - 1)); }
            }
            opend += 1u + nb;
            wave_sync();
            stream_literals((uint32_t)anchor, tl);
            const uint32_t total = rec_done + opend;
            drain(true);
            if (lane == 0) {
                ChunkDesc d;
                d.lead = 0u; d.enc_len = total; d.last_end = (uint32_t)len; d.mcode0 = 0u;
                *dsc = d;
            }
            wave_sync();
            return;
        }
        const uint32_t enc = rec_done + opend;             // record bytes without the trailing literals
        if (with_trailing) {
            if (nseq == 0 && sh == 0) {            // no match at all: the record is the image itself
                for (int i = lane * 16; i < len; i += 64 * 16) *(u32x4 *)(rec + i) = *(const u32x4 *)(s_data + i);
            } else {
                stream_literals((uint32_t)anchor, (uint32_t)(len - anchor));
            }
        }
        drain(true);
        if (lane == 0) {
            ChunkDesc d;
            d.lead = nseq ? lead : 0u; d.enc_len = nseq ? enc : 0u;
            d.last_end = nseq ? (uint32_t)anchor : 0u; d.mcode0 = mcode0;
            *dsc = d;
        }
        wave_sync();
        DBG_ADD(10, DBG_CLK() - dbg_t1);                            // tail: last flushes, trailing literals, descriptor
        DBG_COMMIT();
    }
#undef RD4
}

// un-fused: the chunk comes from a linear (already filtered, or never filtered) buffer in HBM
// bits4 != 0: src is the UN-filtered input and go-blosc's bitshuffle for typesize 4 (an in-place transform of every
// 32-byte window, shuffle.go:184-200) is applied to the chunk image in LDS -- filter fused, no filtered buffer in
// HBM; the caller guarantees n % 32 == 0 and a 16-byte aligned src.
template <int WAYS, int MODE>
__global__ __launch_bounds__(64) void k_match(const uint8_t *__restrict__ src_, uint64_t n_,
                                              ChunkDesc *__restrict__ desc, uint8_t *__restrict__ records,
                                              uint32_t nchunks, int bits4, int keep_long, int accel,
                                              const BatchFrame *__restrict__ bf, const uint32_t *__restrict__ chunk_frame) {
    __shared__ __attribute__((aligned(16))) uint8_t s_data[HB_CHUNK + 112];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[SOUT + 16];       // record staging; its bytes 16.. double as the sequence queue between flushes
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[HSIZEW(WAYS) * WAYS + 8];         // (+ a spare entry: the store of lanes that insert nothing)
    static_assert(16 + (QCAP + 1) * 8 <= SOUT + 16, "the sequence queue (+ its spare slot) must fit into the staging buffer");
    uint2 *const s_q = (uint2 *)(s_out + 16);
    // emission state: the last 32 bytes of s_data's slack (read as data only by lanes past the end of the chunk, never
    // staged over) -- LDS is allocated in 512-byte granules and 13 of them give 24 waves per CU
    uint32_t *const s_st = (uint32_t *)(s_data + HB_CHUNK + 80);
    const int lane = threadIdx.x;
    for (uint32_t ck = blockIdx.x; ck < nchunks; ck += gridDim.x) {
        const uint8_t *src = src_;
        uint64_t n = n_, start = (uint64_t)ck * HB_CHUNK;
        if (bf) {                                          // batch: global chunk ck is chunk ck - chunk0 of its frame
            const uint32_t fid = chunk_frame[ck];
            if (fid == 0xFFFFFFFFu) continue;              // (a gap between two frames: nobody's chunk)
            const BatchFrame &f = bf[fid];
            src = f.src; n = f.n; start = (uint64_t)(ck - f.chunk0) * HB_CHUNK;
        }
        const int len = (int)((n - start) < HB_CHUNK ? (n - start) : HB_CHUNK);
        const uint8_t *g = src + start;
        const uint32_t sh = (uint32_t)((uintptr_t)g & 15u);
        // stage the chunk: 16-byte aligned vectors (over-reads stay inside the first/last 16-byte block)
        const u32x4 *ga = (const u32x4 *)(g - sh);
        const uint32_t nv = (sh + (uint32_t)len + 15u) >> 4;
        if (!bits4) {
            for (uint32_t i = lane; i < nv; i += 64) ((u32x4 *)s_data)[i] = ga[i];
        } else {                                           // sh == 0, len % 32 == 0: one 32-byte window per lane and step
            for (uint32_t w = lane; w < (uint32_t)len / 32u; w += 64) {
                u32x4 oa, ob;
                bitshuffle4_window<false>(ga[2 * w], ga[2 * w + 1], oa, ob);
                ((u32x4 *)s_data)[2 * w] = oa;
                ((u32x4 *)s_data)[2 * w + 1] = ob;
            }
        }
        match_chunk<WAYS, MODE>(s_data, sh, len, s_out, s_tab, s_q, s_st, desc + ck, records + (size_t)ck * HB_RSTRIDE, bits4 != 0, keep_long != 0, accel, lane);
    }
}

// Fused byte-shuffle + match (north star: the filter never makes a round trip through HBM).  Chunk  j * nblk + b  of
// the (never materialised) shuffled buffer is byte j of the elements [b * HB_CHUNK, (b+1) * HB_CHUNK): the wave reads
// those HB_CHUNK*TS source bytes coalesced (16 B per lane), picks its byte plane with three v_perm_b32 per 4
// elements, and builds the chunk image in LDS.  The TS waves of a block read the same bytes; they are given
// workgroup ids that are equal mod 8 (same XCD under round-robin placement: they share the L2 lines -- speed
// only) and are otherwise independent: planes differ a lot in cost, a barrier between them would idle the cheap ones.
template <int TS, int WAYS, int MODE>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAYS == 1 ? ENC_WAVES : 4))) void k_match_fused(const uint8_t *__restrict__ src_, ChunkDesc *__restrict__ desc,
                                                    uint8_t *__restrict__ records, uint32_t nblk_, uint32_t plane_mask, int accel,
                                                    const BatchFrame *__restrict__ bf, const uint32_t *__restrict__ chunk_frame, uint32_t total_) {
    __shared__ __attribute__((aligned(16))) uint8_t s_data[HB_CHUNK + 112];
    __shared__ __attribute__((aligned(16))) uint8_t s_out[SOUT + 16];       // record staging; its bytes 16.. double as the sequence queue between flushes
    __shared__ __attribute__((aligned(16))) uint16_t s_tab[HSIZEW(WAYS) * WAYS + 8];         // (+ a spare entry: the store of lanes that insert nothing)
    static_assert(16 + (QCAP + 1) * 8 <= SOUT + 16, "the sequence queue (+ its spare slot) must fit into the staging buffer");
    uint2 *const s_q = (uint2 *)(s_out + 16);
    // emission state: the last 32 bytes of s_data's slack (read as data only by lanes past the end of the chunk, never
    // staged over) -- LDS is allocated in 512-byte granules and 13 of them give 24 waves per CU
    uint32_t *const s_st = (uint32_t *)(s_data + HB_CHUNK + 80);
    const int lane = threadIdx.x;
    const uint32_t total = bf ? total_ : nblk_ * TS;
    for (uint32_t gi = blockIdx.x; gi < total; gi += gridDim.x) {
        const uint8_t *src = src_;
        uint32_t nblk = nblk_, i = gi, ck0 = 0;
        if (bf) {                                          // batch: global work item gi is item gi - chunk0 of its frame (chunk0 is a
            const uint32_t fid = chunk_frame[gi];          // multiple of 8 * TS there: the XCD of a block's planes is kept)
            if (fid == 0xFFFFFFFFu) continue;              // (a gap between two frames: nobody's chunk)
            const BatchFrame &f = bf[fid];
            src = f.src; nblk = f.nblk; ck0 = f.chunk0; i = gi - ck0;
        }
        // i -> (block b, plane j): within a group of 8 blocks, work item (j, b % 8) has index j * 8 + b % 8
        const uint32_t grp = i / (8u * TS), r = i % (8u * TS);
        // rotate the plane with the pass number: a workgroup's id fixes r, and planes differ a lot in cost
        uint32_t b = grp * 8u + (r & 7u), j = ((r >> 3) + gi / gridDim.x) % TS;
        if (grp * 8u + 8u > nblk) {                        // ragged last group: plain (b, j) order
            const uint32_t k = i - grp * 8u * TS;
            const uint32_t nb = nblk - grp * 8u;
            b = grp * 8u + k % nb; j = k / nb;
        }
        if (!((plane_mask >> j) & 1u)) continue;           // hb_debug_plane_mask: per-plane timing
        const uint64_t e0 = (uint64_t)b * HB_CHUNK;        // first element of the block
        wave_sync();
        const int dbg_plane = (int)j; (void)dbg_plane;
        DBG_DECL();
        const unsigned long long dbg_k0 = DBG_CLK(); (void)dbg_k0;
        // all loads of a batch are issued before the first use: one HBM round trip per 16 vectors, not 16
        if constexpr (TS == 2) {
            const uint32_t sel = j ? 0x07050301u : 0x06040200u;
            u32x4 v[HB_CHUNK / 512];
#pragma unroll
            for (int it = 0; it < (int)(HB_CHUNK / 512); it++)              // 8 elements (16 B) per lane per step
                v[it] = ld16u(src + (e0 + (uint32_t)it * 512u + (uint32_t)lane * 8u) * 2);
#pragma unroll
            for (int it = 0; it < (int)(HB_CHUNK / 512); it++) {
                u32x2 pl;
                pl.x = __builtin_amdgcn_perm(v[it].y, v[it].x, sel);
                pl.y = __builtin_amdgcn_perm(v[it].w, v[it].z, sel);
                *(u32x2 *)&s_data[(uint32_t)it * 512u + (uint32_t)lane * 8u] = pl;
            }
        } else {
            const uint32_t q = j >> 2, rb = j & 3u;
            const uint32_t sel = 0x0c0c0000u | ((4u + rb) << 8) | rb;      // {lo.byte rb, hi.byte rb, 0, 0}
            constexpr int NV = TS / 4;                                      // 16-byte vectors per 4 elements
            constexpr int STEPS = (int)(HB_CHUNK / 256);
#ifndef ENC_STAGE_VECS
#define ENC_STAGE_VECS 8
#endif
            constexpr int BATCH = (ENC_STAGE_VECS / NV) < STEPS ? (ENC_STAGE_VECS / NV) : STEPS;     // steps per batch (<= 8 vectors in flight: 32 VGPRs, the kernel is capped at 80)
            static_assert(STEPS % BATCH == 0, "chunk size must be a multiple of the staging batch");
#pragma unroll 1
            for (int it0 = 0; it0 < (int)(HB_CHUNK / 256); it0 += BATCH) {  // 4 elements (4*TS bytes) per lane per step
                u32x4 v[BATCH][NV];
#pragma unroll
                for (int k = 0; k < BATCH; k++)
#pragma unroll
                    for (int c = 0; c < NV; c++)
                        v[k][c] = ld16u(src + (e0 + (uint32_t)(it0 + k) * 256u + (uint32_t)lane * 4u) * TS + c * 16);
#pragma unroll
                for (int k = 0; k < BATCH; k++) {
                    uint32_t w0, w1, w2, w3;                                // the dword holding byte j of elements 0..3
                    if constexpr (TS == 4) { w0 = v[k][0].x; w1 = v[k][0].y; w2 = v[k][0].z; w3 = v[k][0].w; }
                    else {                                                  // TS == 8: element = 2 dwords
                        w0 = q ? v[k][0].y : v[k][0].x; w1 = q ? v[k][0].w : v[k][0].z;
                        w2 = q ? v[k][1].y : v[k][1].x; w3 = q ? v[k][1].w : v[k][1].z;
                    }
                    const uint32_t t = __builtin_amdgcn_perm(w1, w0, sel), u = __builtin_amdgcn_perm(w3, w2, sel);
                    *(uint32_t *)&s_data[(uint32_t)(it0 + k) * 256u + (uint32_t)lane * 4u] = __builtin_amdgcn_perm(u, t, 0x05040100u);
                }
            }
        }
        const uint32_t ck = ck0 + j * nblk + b;
#if defined(LAB_ENC) && LAB_ENC == 3          /* lab ablation: staging only (garbage frames, timing only) */
        wave_sync();
        if (lane == 0) { ChunkDesc d0; d0.lead = 0; d0.enc_len = s_data[ck & 4095u]; d0.last_end = 0; d0.mcode0 = 0; desc[ck] = d0; }
        continue;
#endif
        DBG_ADD(11, DBG_CLK() - dbg_k0);                            // staging (source reads, plane extraction, LDS writes issued)
        match_chunk<WAYS, MODE>(s_data, 0u, (int)HB_CHUNK, s_out, s_tab, s_q, s_st, desc + ck, records + (size_t)ck * HB_RSTRIDE, true, false, accel, lane, (int)j);
        DBG_ADD(12, DBG_CLK() - dbg_k0); DBG_ADD(13, 1);            // the whole chunk
        DBG_COMMIT();
    }
}

// ----------------------------------------------------------------------------------------------
// scans
// ----------------------------------------------------------------------------------------------
// inclusive scan of 256 Aggs in LDS (Hillis-Steele, ordered); threads >= 256 only take part in the barriers
__device__ __forceinline__ void block_scan_agg(Agg *s, int t) {
    for (int d = 1; d < 256; d <<= 1) {
        Agg x = agg_identity(), y = agg_identity();
        if (t < 256) { x = s[t]; if (t >= d) y = s[t - d]; }
        __syncthreads();
        if (t < 256) s[t] = agg_combine(y, x);
        __syncthreads();
    }
}

// per tile: aggregate of its chunks + position of its first match
__global__ __launch_bounds__(256) void k_tiles(const ChunkDesc *__restrict__ desc, uint32_t nchunks,
                                               Agg *__restrict__ tile_agg, uint32_t *__restrict__ tile_nf,
                                               const BatchFrame *__restrict__ bf, const uint32_t *__restrict__ tile_frame) {
    __shared__ Agg s[256];
    __shared__ uint32_t s_nf[256];
    const int t = threadIdx.x;
    uint32_t ck = blockIdx.x * HB_TILE_CHUNKS + t;         // chunk inside its frame
    if (bf) {                                              // batch: tile blockIdx.x is tile blockIdx.x - tile0 of its frame
        const BatchFrame &f = bf[tile_frame[blockIdx.x]];
        ck = (blockIdx.x - f.tile0) * HB_TILE_CHUNKS + t; nchunks = f.nchunks; desc += f.chunk0;
    }
    Agg a = agg_identity();
    if (ck < nchunks) a = agg_of_chunk(desc[ck], ck * HB_CHUNK);
    s[t] = a;
    s_nf[t] = a.has ? a.F : 0xFFFFFFFFu;
    __syncthreads();
    block_scan_agg(s, t);
    for (int d = 128; d > 0; d >>= 1) {
        if (t < d) s_nf[t] = min(s_nf[t], s_nf[t + d]);
        __syncthreads();
    }
    if (t == 0) { tile_agg[blockIdx.x] = s[255]; tile_nf[blockIdx.x] = s_nf[0]; }
}

struct FrameInfo { int frame, codec, shuffle, typesize; unsigned opts; };

// one workgroup: exclusive prefix of tile aggregates, exclusive suffix-min of tile first-match positions,
// total size, memcpy decision, frame header, result record.
__global__ __launch_bounds__(256) void k_scan(const Agg *__restrict__ tile_agg, const uint32_t *__restrict__ tile_nf,
                                              Agg *__restrict__ tile_pre, uint32_t *__restrict__ tile_suf,
                                              uint32_t ntiles, uint32_t nchunks, uint64_t n, EncPlan *plan,
                                              uint8_t *dst, FrameInfo fi, hb_result *result, int has_index,
                                              const BatchFrame *__restrict__ bf) {
    __shared__ Agg s[256];
    __shared__ uint32_t s_nf[256];
    __shared__ Agg carry;
    __shared__ uint32_t carry_nf;
    const int t = threadIdx.x;
    if (bf) {                                              // batch: one workgroup per frame, over that frame's tiles
        const BatchFrame &f = bf[blockIdx.x];
        tile_agg += f.tile0; tile_nf += f.tile0; tile_pre += f.tile0; tile_suf += f.tile0;
        ntiles = f.ntiles; nchunks = f.nchunks; n = f.n; plan = f.plan; dst = f.dst; result = f.result;
    }
    if (t == 0) { carry = agg_identity(); carry_nf = 0xFFFFFFFFu; }
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {       // forward
        const uint32_t i = base + t;
        s[t] = (i < ntiles) ? tile_agg[i] : agg_identity();
        __syncthreads();
        block_scan_agg(s, t);
        const Agg excl = agg_combine(carry, t ? s[t - 1] : agg_identity());
        if (i < ntiles) tile_pre[i] = excl;
        __syncthreads();
        if (t == 0) carry = agg_combine(carry, s[255]);
        __syncthreads();
    }
    for (int64_t base = (int64_t)((ntiles + 255) / 256) * 256 - 256; base >= 0; base -= 256) {   // backward
        const uint32_t i = (uint32_t)base + t;
        s_nf[t] = (i < ntiles) ? tile_nf[i] : 0xFFFFFFFFu;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {      // inclusive suffix min
            const uint32_t x = s_nf[t], y = (t + d < 256) ? s_nf[t + d] : 0xFFFFFFFFu;
            __syncthreads();
            s_nf[t] = min(x, y);
            __syncthreads();
        }
        const uint32_t excl = min(carry_nf, (t + 1 < 256) ? s_nf[t + 1] : 0xFFFFFFFFu);
        if (i < ntiles) tile_suf[i] = excl;
        __syncthreads();
        if (t == 0) carry_nf = min(carry_nf, s_nf[0]);
        __syncthreads();
    }
    if (t == 0) {
        const Agg all = carry;
        const uint64_t a2 = all.has ? all.E : 0;
        const uint64_t tl = n - a2;                                   // final literal run
        const uint64_t C = (uint64_t)agg_bytes(all) + 1 + lz4_ext_bytes((uint32_t)tl) + tl;
        plan->cbytes_block = C;
        plan->nchunks = nchunks;
        uint32_t use_memcpy = 0;
        uint64_t payload = C;
        if (fi.frame) {
            use_memcpy = !(fi.opts & HB_OPT_INTERNAL_BLOCK) && C >= n;  // blosc.go:342 (the codec seam wants the block itself, always)
            if (use_memcpy) payload = n;
            uint8_t flags = 0;                                        // blosc.go:348-356
            if (fi.shuffle == HB_SHUFFLE) flags |= HB_FLAG_SHUFFLE;
            else if (fi.shuffle == HB_BITSHUFFLE) flags |= HB_FLAG_BITSHUFFLE;
            if (use_memcpy) flags |= HB_FLAG_MEMCPY;
            const uint32_t cbytes = (uint32_t)(HB_HEADER_SIZE + payload);
            dst[0] = HB_FORMAT_VERSION; dst[1] = (uint8_t)fi.codec; dst[2] = flags; dst[3] = (uint8_t)fi.typesize;   // :358-366
            ((uint32_t *)dst)[1] = (uint32_t)n; ((uint32_t *)dst)[2] = (uint32_t)n; ((uint32_t *)dst)[3] = cbytes;
            result->flags = flags;
            result->bytes = cbytes;
            uint64_t total = cbytes;
            plan->index_off = 0;
            if (has_index && !use_memcpy) {
                plan->index_off = ((uint64_t)cbytes + 7) & ~7ull;
                total = plan->index_off + HB_IDX_HDR_BYTES + (uint64_t)HB_IDX_ENTRY * (nchunks + 1);
                for (uint64_t i = cbytes; i < plan->index_off; i++) dst[i] = 0;   // pad: frames are deterministic byte for byte
            }
            result->total_bytes = total;
        } else {
            if (nchunks == 0) dst[0] = 0;                             // empty input: a single zero token
            result->flags = 0;
            result->bytes = C;
            result->total_bytes = C;
            plan->index_off = 0;
        }
        plan->use_memcpy = use_memcpy;
        result->status = HB_OK;
        result->reserved = 0;
    }
}

// ----------------------------------------------------------------------------------------------
// k_stitch: one workgroup (16 waves) per tile of 256 chunks; wave w places chunks w, w+16, ...
// ----------------------------------------------------------------------------------------------
#ifndef STITCH_NT
#define STITCH_NT 0     // (measured: nontemporal copies in k_stitch 0.249 -> 0.265 ms)
#endif
#if STITCH_NT
#define STITCH_COPY wave_copy_g2g_nt
#else
#define STITCH_COPY wave_copy_g2g
#endif
#define STITCH_THREADS 512     /* measured: 1024 -> 0.29 ms, 512 -> 0.25 ms, 256 -> 0.36 ms per GiB */
__global__ __launch_bounds__(STITCH_THREADS) void k_stitch(
        const ChunkDesc *__restrict__ desc, const uint8_t *__restrict__ records, const uint8_t *__restrict__ src,
        const Agg *__restrict__ tile_pre, const uint32_t *__restrict__ tile_suf, const EncPlan *__restrict__ plan,
        uint32_t nchunks, uint64_t n, uint8_t *__restrict__ out /* block start */,
        uint8_t *__restrict__ index_base_ext, uint8_t *__restrict__ frame_base, const uint8_t *__restrict__ memcpy_src,
        int lit_from_records, const BatchFrame *__restrict__ bf, const uint32_t *__restrict__ tile_frame) {
    __shared__ Agg s[256];
    __shared__ uint32_t s_nf[256];
    __shared__ ChunkDesc s_desc[256];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    constexpr int NW = STITCH_THREADS / 64;
    uint32_t tile = blockIdx.x;                                       // tile inside its frame
    if (bf) {                                                         // batch: everything frame-local from here on
        const BatchFrame &f = bf[tile_frame[blockIdx.x]];
        tile = blockIdx.x - f.tile0;
        desc += f.chunk0; records += (size_t)f.chunk0 * HB_RSTRIDE; src = f.src; tile_pre += f.tile0; tile_suf += f.tile0;
        plan = f.plan; nchunks = f.nchunks; n = f.n; out = f.dst + HB_HEADER_SIZE; frame_base = f.dst; memcpy_src = f.memcpy_src;
    }
    const uint32_t ck0 = tile * HB_TILE_CHUNKS;
    const uint32_t cnt = min(HB_TILE_CHUNKS, nchunks - ck0);

    if (plan->use_memcpy) {                                           // blosc.go:343-345: payload = (filtered) input
        const uint64_t b0 = (uint64_t)ck0 * HB_CHUNK;
        const uint64_t b1 = min((uint64_t)(ck0 + cnt) * HB_CHUNK, n);
        for (uint64_t off = b0 + (uint64_t)wave * 16384u; off < b1; off += (uint64_t)NW * 16384u)
            if (memcpy_src) wave_copy_g2g(out + off, memcpy_src + off, (uint32_t)min((uint64_t)16384u, b1 - off), lane);
        return;
    }

    if (t < 256) {
        ChunkDesc d; d.lead = 0; d.enc_len = 0; d.last_end = 0; d.mcode0 = 0;
        if ((uint32_t)t < cnt) d = desc[ck0 + t];
        s_desc[t] = d;
        const Agg mine = ((uint32_t)t < cnt) ? agg_of_chunk(d, (ck0 + t) * HB_CHUNK) : agg_identity();
        s[t] = mine;
        s_nf[t] = mine.has ? mine.F : 0xFFFFFFFFu;
    }
    __syncthreads();
    block_scan_agg(s, t);                                             // inclusive prefix within the tile
    for (int dd = 1; dd < 256; dd <<= 1) {                            // inclusive suffix-min of F within the tile
        uint32_t x = 0, y = 0xFFFFFFFFu;
        if (t < 256) { x = s_nf[t]; if (t + dd < 256) y = s_nf[t + dd]; }
        __syncthreads();
        if (t < 256) s_nf[t] = min(x, y);
        __syncthreads();
    }
    const Agg tpre = tile_pre[tile];                                  // (uniform address: scalar loads)
    const uint32_t tsuf = tile_suf[tile];
    // values read from LDS are wave-uniform here but arrive in vector registers: move them to scalar ones, or every
    // address computation and copy loop below runs on the vector side under exec masks
#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))     /* the builtin is int -> int: no sign extension later */
    auto rfl_agg = [](const Agg &x) __attribute__((always_inline)) {
        Agg r;
        r.fixed = (int64_t)(((uint64_t)RFL((uint32_t)((uint64_t)x.fixed >> 32)) << 32) | RFL((uint32_t)(uint64_t)x.fixed));
        r.F = RFL(x.F); r.E = RFL(x.E); r.has = RFL(x.has); r.pad = 0;
        return r;
    };
    uint8_t *index = nullptr;
    if (index_base_ext) index = index_base_ext;
    else if (frame_base && plan->index_off) index = frame_base + plan->index_off;

    for (uint32_t c = wave; c < cnt; c += NW) {
        const uint32_t ck = ck0 + c;
        ChunkDesc cd = s_desc[c];
        cd.lead = RFL(cd.lead); cd.enc_len = RFL(cd.enc_len); cd.last_end = RFL(cd.last_end); cd.mcode0 = RFL(cd.mcode0);
        const uint32_t start = ck * HB_CHUNK;
        const uint32_t end = (uint32_t)min((uint64_t)start + HB_CHUNK, n);
        const Agg P = agg_combine(tpre, c ? rfl_agg(s[c - 1]) : agg_identity()); // everything before this chunk
        uint32_t NF = min(tsuf, (c + 1 < 256) ? RFL(s_nf[c + 1]) : 0xFFFFFFFFu);  // next match after this chunk
        if (NF == 0xFFFFFFFFu) NF = (uint32_t)n;                                  // ... or the end of the block
        const uint32_t a = P.has ? P.E : 0u;
        const uint64_t O = (uint64_t)agg_bytes(P);
        uint64_t I = O;
        uint32_t a2 = a, tpos = start;
        if (cd.last_end) {
            const uint32_t tl = start + cd.lead - a;                  // literal run closed by this chunk's first match
            const uint32_t hdr = 1 + lz4_ext_bytes(tl);
            const uint32_t carry_lits = start - a;
            wave_write_lit_header(out + O, tl, cd.mcode0, lane);
            STITCH_COPY(out + O + hdr + carry_lits, records + (size_t)ck * HB_RSTRIDE, cd.enc_len, lane);
            I = O + hdr + carry_lits + cd.enc_len;
            a2 = start + cd.last_end; tpos = a2;
        }
        // trailing literals (from the source, or from the record when the filter was fused): part of the run that ends at NF
        const uint32_t hdr2 = 1 + lz4_ext_bytes(NF - a2);
        const uint8_t *lsrc = lit_from_records ? records + (size_t)ck * HB_RSTRIDE + (cd.last_end ? cd.enc_len : 0u) : src + tpos;
        STITCH_COPY(out + I + hdr2 + (tpos - a2), lsrc, end - tpos, lane);
        if (ck + 1 == nchunks) wave_write_lit_header(out + I, NF - a2, 0, lane);  // final literal-only sequence
        if (index && lane == 0) {
            uint32_t *e = (uint32_t *)(index + HB_IDX_HDR_BYTES) + 4 * (size_t)ck;
            if (ck == 0) { e[0] = 0; e[1] = 0; e[2] = HB_IDX_AT_TOKEN; e[3] = 0; }
            else {
                const uint32_t Fc = cd.last_end ? start + cd.lead : NF;           // end of the run containing `start`
                const uint32_t hdrc = 1 + lz4_ext_bytes(Fc - a);
                e[0] = (uint32_t)(O + hdrc + (start - a)); e[1] = start; e[2] = Fc - start; e[3] = (uint32_t)O;
            }
            if (ck + 1 == nchunks) {
                e[4] = (uint32_t)plan->cbytes_block; e[5] = (uint32_t)n; e[6] = 0; e[7] = 0;
                uint32_t *h = (uint32_t *)index;
                h[0] = HB_IDX_MAGIC; h[1] = HB_IDX_VERSION | (HB_IDX_ENTRY << 16); h[2] = nchunks; h[3] = HB_CHUNK;
                h[4] = (uint32_t)plan->cbytes_block; h[5] = (uint32_t)n; h[6] = 0;
                h[7] = h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// Snappy block assembly (codec.go:228-244: snappy.Encode): uvarint(n) + the chunks' element streams, concatenated.
// A Snappy block has no sequence chain to repair: every chunk's record is already final, so this is a prefix sum + a copy.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t uvarint_len(uint64_t v) { uint32_t k = 1; while (v >= 128) { v >>= 7; k++; } return k; }

__global__ __launch_bounds__(256) void k_sn_tiles(const ChunkDesc *__restrict__ desc, uint32_t nchunks, Agg *__restrict__ tile_agg) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x;
    const uint32_t ck = blockIdx.x * HB_TILE_CHUNKS + t;
    s[t] = ck < nchunks ? desc[ck].enc_len : 0u;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) { if (t < d) s[t] += s[t + d]; __syncthreads(); }
    if (t == 0) { Agg a = agg_identity(); a.fixed = s[0]; tile_agg[blockIdx.x] = a; }
}

// one workgroup: exclusive prefix of the tile sums, total, memcpy decision (blosc.go:342), frame header, result record
__global__ __launch_bounds__(256) void k_sn_scan(const Agg *__restrict__ tile_agg, Agg *__restrict__ tile_pre, uint32_t ntiles, uint32_t nchunks,
                                                 uint64_t n, EncPlan *plan, uint8_t *dst, FrameInfo fi, hb_result *result, int has_index) {
    __shared__ uint64_t s[256];
    __shared__ uint64_t carry;
    const int t = threadIdx.x;
    if (t == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < ntiles; base += 256) {
        const uint32_t i = base + t;
        const uint64_t mine = i < ntiles ? (uint64_t)tile_agg[i].fixed : 0ull;
        s[t] = mine;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint64_t y = t >= d ? s[t - d] : 0ull;
            __syncthreads();
            s[t] += y;
            __syncthreads();
        }
        if (i < ntiles) { Agg a = agg_identity(); a.fixed = (int64_t)(carry + s[t] - mine); tile_pre[i] = a; }
        __syncthreads();
        if (t == 0) carry += s[255];
        __syncthreads();
    }
    if (t == 0) {
        const uint32_t hl = uvarint_len(n);
        const uint64_t C = hl + carry;
        plan->cbytes_block = C;
        plan->nchunks = nchunks;
        const uint32_t use_memcpy = !(fi.opts & HB_OPT_INTERNAL_BLOCK) && C >= n;      // blosc.go:342
        const uint64_t payload = use_memcpy ? n : C;
        uint8_t flags = 0;                                            // blosc.go:348-356
        if (fi.shuffle == HB_SHUFFLE) flags |= HB_FLAG_SHUFFLE;
        else if (fi.shuffle == HB_BITSHUFFLE) flags |= HB_FLAG_BITSHUFFLE;
        if (use_memcpy) flags |= HB_FLAG_MEMCPY;
        const uint32_t cbytes = (uint32_t)(HB_HEADER_SIZE + payload);
        dst[0] = HB_FORMAT_VERSION; dst[1] = (uint8_t)fi.codec; dst[2] = flags; dst[3] = (uint8_t)fi.typesize;   // :358-366
        ((uint32_t *)dst)[1] = (uint32_t)n; ((uint32_t *)dst)[2] = (uint32_t)n; ((uint32_t *)dst)[3] = cbytes;
        if (!use_memcpy) {                                            // uvarint(n), little-endian base 128
            uint64_t v = n; uint8_t *o = dst + HB_HEADER_SIZE;
            while (v >= 128) { *o++ = (uint8_t)(v | 128u); v >>= 7; }
            *o = (uint8_t)v;
        }
        result->flags = flags; result->bytes = cbytes;
        uint64_t total = cbytes;
        plan->index_off = 0;
        if (has_index && !use_memcpy) {
            plan->index_off = ((uint64_t)cbytes + 7) & ~7ull;
            total = plan->index_off + HB_SNX_HDR_BYTES + (uint64_t)HB_SNX_ENTRY * (nchunks + 1);
            for (uint64_t i = cbytes; i < plan->index_off; i++) dst[i] = 0;
        }
        result->total_bytes = total;
        plan->use_memcpy = use_memcpy;
        result->status = HB_OK; result->reserved = 0;
    }
}

// one workgroup per tile of 256 chunks: every wave copies its chunks' records to their final place; unit index (stream offsets)
__global__ __launch_bounds__(STITCH_THREADS) void k_sn_pack(const ChunkDesc *__restrict__ desc, const uint8_t *__restrict__ records,
                                                            const Agg *__restrict__ tile_pre, const EncPlan *__restrict__ plan, uint32_t nchunks,
                                                            uint64_t n, uint8_t *__restrict__ out /* payload start */, uint8_t *__restrict__ frame_base,
                                                            const uint8_t *__restrict__ memcpy_src) {
    __shared__ uint32_t s[256];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    constexpr int NW = STITCH_THREADS / 64;
    const uint32_t ck0 = blockIdx.x * HB_TILE_CHUNKS;
    const uint32_t cnt = min(HB_TILE_CHUNKS, nchunks - ck0);
    if (plan->use_memcpy) {                                           // blosc.go:343-345: payload = (filtered) input
        const uint64_t b0 = (uint64_t)ck0 * HB_CHUNK, b1 = min((uint64_t)(ck0 + cnt) * HB_CHUNK, n);
        for (uint64_t off = b0 + (uint64_t)wave * 16384u; off < b1; off += (uint64_t)NW * 16384u)
            if (memcpy_src) wave_copy_g2g(out + off, memcpy_src + off, (uint32_t)min((uint64_t)16384u, b1 - off), lane);
        return;
    }
    uint32_t mine = 0;
    if (t < 256) { mine = (uint32_t)t < cnt ? desc[ck0 + t].enc_len : 0u; s[t] = mine; }
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {                               // inclusive prefix within the tile
        uint32_t y = 0;
        if (t < 256 && t >= d) y = s[t - d];
        __syncthreads();
        if (t < 256) s[t] += y;
        __syncthreads();
    }
    const uint32_t hl = uvarint_len(n);
    const uint64_t tbase = (uint64_t)hl + (uint64_t)tile_pre[blockIdx.x].fixed;
    uint8_t *index = (frame_base && plan->index_off) ? frame_base + plan->index_off : nullptr;
    for (uint32_t c = wave; c < cnt; c += NW) {
        const uint32_t ck = ck0 + c;
        const uint32_t incl = (uint32_t)__builtin_amdgcn_readfirstlane((int)s[c]);
        const uint32_t len = (uint32_t)__builtin_amdgcn_readfirstlane((int)desc[ck].enc_len);
        const uint64_t O = tbase + incl - len;
        wave_copy_g2g(out + O, records + (size_t)ck * HB_RSTRIDE, len, lane);
        if (index && lane == 0) {
            uint32_t *e = (uint32_t *)(index + HB_SNX_HDR_BYTES);
            e[ck] = (uint32_t)O;
            if (ck + 1 == nchunks) {
                e[nchunks] = (uint32_t)plan->cbytes_block;
                uint32_t *h = (uint32_t *)index;
                h[0] = HB_SNX_MAGIC; h[1] = HB_IDX_VERSION | (HB_SNX_ENTRY << 16); h[2] = nchunks; h[3] = HB_CHUNK;
                h[4] = (uint32_t)plan->cbytes_block; h[5] = (uint32_t)n; h[6] = hl;
                h[7] = h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5] ^ h[6];
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------
// launch
// ----------------------------------------------------------------------------------------------
// Search depth and skip acceleration from (codec, level).  LZ4: one candidate (lz4.CompressBlock); the reference ignores the level
// there (codec.go:63-66), here it is a speed knob that never changes validity: 1-3 skip harder, 4-6 the default, 7-9 no skipping.
// LZ4HC: codec.go:96-106 maps level <= 3 / <= 5 / <= 7 / else to lz4.Level1 / 5 / 7 / 9; here 1 / 2 / 4 / 4 candidates per
// bucket, no skipping.  Snappy has no levels (codec.go:232-235).
static inline void enc_policy(const hb_enc_args &a, int &ways, int &accel) {
    const int level = a.level < 1 ? 1 : (a.level > 9 ? 9 : a.level);          // blosc.go:277-282
    ways = 1; accel = 64;
    if (a.codec == HB_LZ4HC) { ways = level <= 3 ? 1 : (level <= 5 ? 2 : 4); accel = 0; }
    else if (a.codec == HB_LZ4 && a.frame) accel = level <= 3 ? 128 : (level <= 6 ? 64 : 0);
}

#define HB_LAUNCH_FUSED(TS, W, SN) hipLaunchKernelGGL((k_match_fused<TS, W, SN>), dim3(grid), dim3(64), 0, s, a.src, desc, records, nblk, hb_dbg_plane_mask(), accel, bf, chunk_frame, total_items)
static void launch_match_fused(const hb_enc_args &a, unsigned grid, ChunkDesc *desc, uint8_t *records, uint32_t nblk, hipStream_t s,
                               const BatchFrame *bf = nullptr, const uint32_t *chunk_frame = nullptr, uint32_t total_items = 0) {
    int ways, accel;
    enc_policy(a, ways, accel);
    const bool sn = a.codec == HB_SNAPPY;
    switch (a.fused_ts) {
    case 2: if (sn) HB_LAUNCH_FUSED(2, 1, true); else if (ways == 1) HB_LAUNCH_FUSED(2, 1, false); else if (ways == 2) HB_LAUNCH_FUSED(2, 2, false); else HB_LAUNCH_FUSED(2, 4, false); break;
    case 4: if (sn) HB_LAUNCH_FUSED(4, 1, true); else if (ways == 1) HB_LAUNCH_FUSED(4, 1, false); else if (ways == 2) HB_LAUNCH_FUSED(4, 2, false); else HB_LAUNCH_FUSED(4, 4, false); break;
    default: if (sn) HB_LAUNCH_FUSED(8, 1, true); else if (ways == 1) HB_LAUNCH_FUSED(8, 1, false); else if (ways == 2) HB_LAUNCH_FUSED(8, 2, false); else HB_LAUNCH_FUSED(8, 4, false); break;
    }
}
#define HB_LAUNCH_MATCH(W, SN) hipLaunchKernelGGL((k_match<W, SN>), dim3(grid), dim3(64), 0, s, a.src, (uint64_t)a.n, desc, records, nchunks, a.fused_bits, keep_long, accel, bf, chunk_frame)
static void launch_match(const hb_enc_args &a, unsigned grid, ChunkDesc *desc, uint8_t *records, uint32_t nchunks, hipStream_t s,
                         const BatchFrame *bf = nullptr, const uint32_t *chunk_frame = nullptr) {
    int ways, accel;
    enc_policy(a, ways, accel);
    // byte-shuffled frames keep the fused kernels' table policy, fused or not (identical frames either way)
    const int keep_long = (a.frame && a.shuffle == HB_SHUFFLE && a.typesize > 1) ? 0 : 1;
    if (a.codec == HB_SNAPPY) HB_LAUNCH_MATCH(1, true);
    else if (ways == 1) HB_LAUNCH_MATCH(1, false);
    else if (ways == 2) HB_LAUNCH_MATCH(2, false);
    else HB_LAUNCH_MATCH(4, false);
}

// every HB_CHUNK bytes of src become an LZ4 block of their own (hb_cblosc.hip): records + descriptors only, nothing is stitched
void hb_launch_match_selfcontained(const uint8_t *src, size_t n, void *desc, uint8_t *records, uint32_t nchunks, int accel, hipStream_t s) {
    const unsigned grid = nchunks < 256u * 256u ? nchunks : 256u * 256u;
    hipLaunchKernelGGL((k_match<1, 2>), dim3(grid), dim3(64), 0, s, src, (uint64_t)n, (ChunkDesc *)desc, records, nchunks, 0, 0, accel, (const BatchFrame *)nullptr, (const uint32_t *)nullptr);
}

// the same with the byte shuffle fused (typesize 2 / 4 / 8): chunk (plane j, element block b) lands at index j * nblk + b
bool hb_launch_match_fused_selfcontained(const uint8_t *src, int typesize, void *desc, uint8_t *records, uint32_t nblk, int accel, hipStream_t s) {
    const uint64_t items = (uint64_t)nblk * (uint64_t)typesize;
    const unsigned grid = (unsigned)(items < 256u * 256u ? items : 256u * 256u);
    switch (typesize) {
    case 2: hipLaunchKernelGGL((k_match_fused<2, 1, 2>), dim3(grid), dim3(64), 0, s, src, (ChunkDesc *)desc, records, nblk, 0xFFFFFFFFu, accel, (const BatchFrame *)nullptr, (const uint32_t *)nullptr, 0u); return true;
    case 4: hipLaunchKernelGGL((k_match_fused<4, 1, 2>), dim3(grid), dim3(64), 0, s, src, (ChunkDesc *)desc, records, nblk, 0xFFFFFFFFu, accel, (const BatchFrame *)nullptr, (const uint32_t *)nullptr, 0u); return true;
    case 8: hipLaunchKernelGGL((k_match_fused<8, 1, 2>), dim3(grid), dim3(64), 0, s, src, (ChunkDesc *)desc, records, nblk, 0xFFFFFFFFu, accel, (const BatchFrame *)nullptr, (const uint32_t *)nullptr, 0u); return true;
    default: return false;
    }
}

int hb_launch_lz4_encode(const hb_enc_args &a, hipStream_t s) {
    const EncLayout L = enc_layout(a.n);
    uint8_t *w = a.work;
    EncPlan *plan = (EncPlan *)(w + L.plan);
    ChunkDesc *desc = (ChunkDesc *)(w + L.desc);
    Agg *tile_agg = (Agg *)(w + L.tile_agg), *tile_pre = (Agg *)(w + L.tile_pre);
    uint32_t *tile_nf = (uint32_t *)(w + L.tile_nf), *tile_suf = (uint32_t *)(w + L.tile_suf);
    uint8_t *records = w + L.records;
    uint8_t *out = a.frame ? a.dst + HB_HEADER_SIZE : a.dst;
    FrameInfo fi{a.frame, a.codec, a.shuffle, a.typesize, a.opts};
    const int has_index = a.frame ? ((a.opts & HB_OPT_INDEX_TRAILER) ? 1 : 0) : (a.index ? 1 : 0);

    if (L.nchunks) {
        hb_prof_begin(a.fused_ts ? "k_match_fused" : "k_match", s);
        if (a.fused_ts) {
            const uint32_t nblk = L.nchunks / (uint32_t)a.fused_ts;
            // one chunk per workgroup for small frames; beyond that each workgroup makes `typesize` passes where possible -- it
            // then meets every plane exactly once (planes differ 10x in cost) and the workgroups of a block drift apart the
            // least (they share the source lines in L2): measured 1.99 ms with 16384 workgroups, 1.79 ms with 65536,
            // 3.06 ms with one chunk per workgroup (1 GiB, typesize 4).  A capped grid is a multiple of 8 * typesize.
            unsigned grid = L.nchunks;                                   // up to 16 MiB: latency first, one chunk per workgroup
            if (grid > 256u * 16u) {
                const unsigned gran = 8u * (unsigned)a.fused_ts;
                grid = nblk / gran * gran;                               // `typesize` passes
                if (L.nchunks > 256u * 64u && grid < 256u * 64u) grid = 256u * 64u;   // 64 - 256 MiB: at least 16384 workgroups
                if (grid > 256u * 256u) grid = 256u * 256u;
                if (grid == 0) grid = L.nchunks;
            }
            launch_match_fused(a, grid, desc, records, nblk, s);
        } else {
            const unsigned grid = L.nchunks < 256u * 256u ? L.nchunks : 256u * 256u;
            launch_match(a, grid, desc, records, L.nchunks, s);
        }
        hb_prof_end(s);
        if (a.codec == HB_SNAPPY) {
            hb_prof_begin("k_sn_tiles", s);
            hipLaunchKernelGGL(k_sn_tiles, dim3(L.ntiles), dim3(256), 0, s, desc, L.nchunks, tile_agg);
            hb_prof_end(s);
        } else {
            hb_prof_begin("k_tiles", s);
            hipLaunchKernelGGL(k_tiles, dim3(L.ntiles), dim3(256), 0, s, desc, L.nchunks, tile_agg, tile_nf, (const BatchFrame *)nullptr, (const uint32_t *)nullptr);
            hb_prof_end(s);
        }
    }
    if (a.codec == HB_SNAPPY) {                                        // (frame only; n > 0)
        hb_prof_begin("k_sn_scan", s);
        hipLaunchKernelGGL(k_sn_scan, dim3(1), dim3(256), 0, s, tile_agg, tile_pre, L.ntiles, L.nchunks, (uint64_t)a.n, plan, a.dst, fi, a.result, has_index);
        hb_prof_end(s);
        if ((a.fused_ts || a.fused_bits) && !a.memcpy_src) {
            const int rc = hb_launch_filter_gated(a.fused_bits ? HB_OP_BITSHUFFLE : HB_OP_SHUFFLE, out, a.src, a.n, a.fused_bits ? a.fused_bits : a.fused_ts, &plan->use_memcpy, s);
            if (rc) return rc;
        }
        hb_prof_begin("k_sn_pack", s);
        hipLaunchKernelGGL(k_sn_pack, dim3(L.ntiles), dim3(STITCH_THREADS), 0, s, desc, records, tile_pre, plan, L.nchunks, (uint64_t)a.n, out, a.dst, a.memcpy_src);
        hb_prof_end(s);
        HB_HIP_TRY(hipGetLastError());
        return HB_OK;
    }
    hb_prof_begin("k_scan", s);
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(256), 0, s, tile_agg, tile_nf, tile_pre, tile_suf, L.ntiles, L.nchunks,
                       (uint64_t)a.n, plan, a.dst, fi, a.result, has_index, (const BatchFrame *)nullptr);
    hb_prof_end(s);
    if ((a.fused_ts || a.fused_bits) && !a.memcpy_src) {
        // memcpy fallback of a fused frame (blosc.go:342-345 with the filtered payload): the shuffled bytes were never
        // written, so shuffle straight into the payload -- the kernel exits at once unless k_scan chose memcpy
        const int rc = hb_launch_filter_gated(a.fused_bits ? HB_OP_BITSHUFFLE : HB_OP_SHUFFLE, out, a.src, a.n, a.fused_bits ? a.fused_bits : a.fused_ts, &plan->use_memcpy, s);
        if (rc) return rc;
    }
    if (L.nchunks) {
        hb_prof_begin("k_stitch", s);
        hipLaunchKernelGGL(k_stitch, dim3(L.ntiles), dim3(STITCH_THREADS), 0, s, desc, records, a.src, tile_pre, tile_suf,
                           plan, L.nchunks, (uint64_t)a.n, out, a.frame ? (uint8_t *)nullptr : a.index,
                           a.frame ? a.dst : (uint8_t *)nullptr, a.memcpy_src, (a.fused_ts || a.fused_bits) ? 1 : 0,
                           (const BatchFrame *)nullptr, (const uint32_t *)nullptr);
        hb_prof_end(s);
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}

// ----------------------------------------------------------------------------------------------
// batches: K frames, one set of launches (hb_compress_frames_batch_dev)
// ----------------------------------------------------------------------------------------------
struct EncBatchLayout {
    size_t frames, plans, jobs, chunk_frame, tile_frame, desc, tile_agg, tile_nf, tile_pre, tile_suf, records, filtered, total;
    uint32_t total_chunks, total_tiles;
};
// granule of a frame's first global chunk: with the fused byte shuffle the planes of an element block sit 8 work items apart and must
// keep their workgroup id mod 8 (XCD), so every frame starts at a multiple of 8 * typesize (the gap chunks belong to no frame)
static inline uint32_t batch_granule(bool fused_ts, int typesize) { return fused_ts ? 8u * (uint32_t)typesize : 1u; }

static EncBatchLayout enc_batch_layout(int nframes, const size_t *n, uint32_t granule, bool need_filtered) {
    EncBatchLayout L{};
    uint64_t chunks = 0, tiles = 0, fbytes = 0;
    for (int k = 0; k < nframes; k++) {
        const uint64_t c = (n[k] + HB_CHUNK - 1) / HB_CHUNK;
        chunks = (chunks + granule - 1) / granule * granule + c;
        tiles += (c + HB_TILE_CHUNKS - 1) / HB_TILE_CHUNKS;
        fbytes += (n[k] + 255) & ~(size_t)255;
    }
    L.total_chunks = (uint32_t)chunks; L.total_tiles = (uint32_t)tiles;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    L.frames = take((size_t)nframes * sizeof(BatchFrame));
    L.plans = take((size_t)nframes * sizeof(EncPlan));
    L.jobs = take((size_t)nframes * sizeof(hb_filter_job) * 2);
    L.chunk_frame = take((size_t)chunks * 4 + 4);
    L.tile_frame = take((size_t)tiles * 4 + 4);
    L.desc = take((size_t)chunks * sizeof(ChunkDesc));
    L.tile_agg = take((size_t)tiles * sizeof(Agg));
    L.tile_nf = take((size_t)tiles * 4);
    L.tile_pre = take((size_t)tiles * sizeof(Agg));
    L.tile_suf = take((size_t)tiles * 4);
    L.records = take((size_t)chunks * HB_RSTRIDE + 256);
    L.filtered = take(need_filtered ? fbytes + 256 : 0);
    L.total = o;
    return L;
}
// (sized for the larger of the two shapes a batch can take: gaps for the fused shuffle, a filtered copy otherwise)
size_t hb_lz4_enc_batch_workspace(int nframes, const size_t *n, int typesize) {
    if (nframes <= 0 || !n) return 256;
    const size_t a = enc_batch_layout(nframes, n, batch_granule(true, typesize > 0 && typesize <= 8 ? typesize : 8), false).total;
    const size_t b = enc_batch_layout(nframes, n, 1u, true).total;
    return a > b ? a : b;
}

// frame of every global chunk and of every scan tile (gap chunks: no frame)
__global__ __launch_bounds__(64) void k_bt_map(const BatchFrame *__restrict__ bf, uint32_t nframes, uint32_t *__restrict__ chunk_frame, uint32_t *__restrict__ tile_frame,
                                               uint32_t total_chunks) {
    const uint32_t f = blockIdx.x;
    const BatchFrame b = bf[f];
    const uint32_t next = f + 1 < nframes ? bf[f + 1].chunk0 : total_chunks;
    for (uint32_t c = threadIdx.x; c < next - b.chunk0; c += 64) chunk_frame[b.chunk0 + c] = c < b.nchunks ? f : 0xFFFFFFFFu;
    for (uint32_t t = threadIdx.x; t < b.ntiles; t += 64) tile_frame[b.tile0 + t] = f;
}

int hb_launch_lz4_encode_batch(int nframes, const hb_batch_frame *fr, int codec, int level, int shuffle, int typesize, unsigned opts,
                               uint8_t *work, size_t work_bytes, hipStream_t s) {
    if (nframes <= 0) return HB_OK;
    if (codec != HB_LZ4 && codec != HB_LZ4HC) return HB_ERR_INVALID_CODEC;
    const bool filt = (shuffle == HB_SHUFFLE || shuffle == HB_BITSHUFFLE) && typesize > 1;          // blosc.go:329-333
    // the filter is fused into the matcher when EVERY frame of the batch allows it (same rules as hb_compress_frame_dev); else all
    // frames take the two-pass path: one batched filter launch into a filtered copy, then the plain matcher
    bool fused = filt && shuffle == HB_SHUFFLE && (typesize == 2 || typesize == 4 || typesize == 8) && !(opts & HB_OPT_NO_FUSION);
    bool fused_bits = filt && shuffle == HB_BITSHUFFLE && typesize == 4 && !(opts & HB_OPT_NO_FUSION);
    size_t max_n = 0;
    std::vector<size_t> ns((size_t)nframes);
    for (int k = 0; k < nframes; k++) {
        ns[(size_t)k] = fr[k].n;
        max_n = fr[k].n > max_n ? fr[k].n : max_n;
        if (fused && fr[k].n % ((size_t)typesize * HB_CHUNK) != 0) fused = false;
        if (fused_bits && (fr[k].n % 32 != 0 || ((uintptr_t)fr[k].src & 15u))) fused_bits = false;
        // (a frame shorter than one element is not filtered at all, shuffle.go:17-19: the two-pass path points the matcher at its source)
    }
    const uint32_t granule = batch_granule(fused, typesize);
    const bool need_filtered = filt && !fused && !fused_bits;
    const EncBatchLayout L = enc_batch_layout(nframes, ns.data(), granule, need_filtered);
    if (work_bytes < L.total) return HB_ERR_SHORT_BUFFER;
    BatchFrame *d_bf = (BatchFrame *)(work + L.frames);
    EncPlan *d_plans = (EncPlan *)(work + L.plans);
    hb_filter_job *d_jobs = (hb_filter_job *)(work + L.jobs);
    uint32_t *chunk_frame = (uint32_t *)(work + L.chunk_frame), *tile_frame = (uint32_t *)(work + L.tile_frame);
    ChunkDesc *desc = (ChunkDesc *)(work + L.desc);
    Agg *tile_agg = (Agg *)(work + L.tile_agg), *tile_pre = (Agg *)(work + L.tile_pre);
    uint32_t *tile_nf = (uint32_t *)(work + L.tile_nf), *tile_suf = (uint32_t *)(work + L.tile_suf);
    uint8_t *records = work + L.records, *filtered = work + L.filtered;

    std::vector<BatchFrame> h((size_t)nframes);
    std::vector<hb_filter_job> jobs((size_t)nframes * 2);          // [0, K): the filter pass; [K, 2K): the gated memcpy-fallback shuffle of fused frames
    uint64_t chunks = 0, tiles = 0, foff = 0;
    for (int k = 0; k < nframes; k++) {
        const size_t n = fr[k].n;
        const uint32_t c = (uint32_t)((n + HB_CHUNK - 1) / HB_CHUNK);
        chunks = (chunks + granule - 1) / granule * granule;
        BatchFrame &b = h[(size_t)k];
        const bool f_filt = filt && n >= (size_t)typesize;
        const uint8_t *in = fr[k].src;
        if (need_filtered && f_filt) in = filtered + foff;
        b.src = (fused || fused_bits) ? fr[k].src : in;
        b.dst = fr[k].dst;
        b.memcpy_src = (opts & HB_OPT_REFERENCE_MEMCPY) ? fr[k].src : ((fused || fused_bits) ? nullptr : in);
        b.result = fr[k].result;
        b.plan = d_plans + k;
        b.n = n;
        b.chunk0 = (uint32_t)chunks; b.nchunks = c;
        b.tile0 = (uint32_t)tiles; b.ntiles = (c + HB_TILE_CHUNKS - 1) / HB_TILE_CHUNKS;
        b.nblk = fused ? c / (uint32_t)typesize : 0u; b.pad = 0;
        jobs[(size_t)k] = hb_filter_job{filtered + foff, fr[k].src, (need_filtered && f_filt) ? (uint64_t)n : 0ull, nullptr};
        jobs[(size_t)(nframes + k)] = hb_filter_job{fr[k].dst + HB_HEADER_SIZE, fr[k].src, ((fused || fused_bits) && !b.memcpy_src) ? (uint64_t)n : 0ull, &(d_plans + k)->use_memcpy};
        chunks += c; tiles += b.ntiles; foff += (n + 255) & ~(size_t)255;
    }
    // (pageable host memory: the runtime stages these copies before the call returns, so the vectors may go out of scope)
    HB_HIP_TRY(hipMemcpyAsync(d_bf, h.data(), h.size() * sizeof(BatchFrame), hipMemcpyHostToDevice, s));
    HB_HIP_TRY(hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(hb_filter_job), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_bt_map, dim3((unsigned)nframes), dim3(64), 0, s, d_bf, (uint32_t)nframes, chunk_frame, tile_frame, L.total_chunks);
    if (need_filtered) {
        hb_prof_begin(shuffle == HB_SHUFFLE ? "filter_shuffle" : "filter_bitshuffle", s);
        const int rc = hb_launch_filter_batch(shuffle == HB_SHUFFLE ? HB_OP_SHUFFLE : HB_OP_BITSHUFFLE, d_jobs, nframes, max_n, typesize, s);
        hb_prof_end(s);
        if (rc) return rc;
    }
    hb_enc_args a{};                                                 // what the launch helpers read: policy and fusion only
    a.frame = 1; a.codec = codec; a.shuffle = shuffle; a.typesize = typesize; a.opts = opts; a.level = level;
    a.fused_ts = fused ? typesize : 0; a.fused_bits = fused_bits ? 4 : 0;
    FrameInfo fi{1, codec, shuffle, typesize, opts};
    const int has_index = (opts & HB_OPT_INDEX_TRAILER) ? 1 : 0;
    unsigned grid = L.total_chunks < 256u * 256u ? L.total_chunks : 256u * 256u;
    if (fused && L.total_chunks > 256u * 16u) {                      // as for one frame: `typesize` passes per workgroup where the batch is large
        grid = L.total_chunks / (unsigned)typesize / granule * granule;   // enough (the plane rotates with the pass: every workgroup meets every plane)
        if (L.total_chunks > 256u * 64u && grid < 256u * 64u) grid = 256u * 64u;
        if (grid > 256u * 256u) grid = 256u * 256u;
        if (grid == 0) grid = L.total_chunks;
    }
    hb_prof_begin(fused ? "k_match_fused" : "k_match", s);
    if (fused) launch_match_fused(a, grid, desc, records, 0u, s, d_bf, chunk_frame, L.total_chunks);
    else launch_match(a, grid, desc, records, L.total_chunks, s, d_bf, chunk_frame);
    hb_prof_end(s);
    hb_prof_begin("k_tiles", s);
    hipLaunchKernelGGL(k_tiles, dim3(L.total_tiles), dim3(256), 0, s, desc, 0u, tile_agg, tile_nf, (const BatchFrame *)d_bf, (const uint32_t *)tile_frame);
    hb_prof_end(s);
    hb_prof_begin("k_scan", s);
    hipLaunchKernelGGL(k_scan, dim3((unsigned)nframes), dim3(256), 0, s, tile_agg, tile_nf, tile_pre, tile_suf, 0u, 0u, (uint64_t)0, (EncPlan *)nullptr,
                       (uint8_t *)nullptr, fi, (hb_result *)nullptr, has_index, (const BatchFrame *)d_bf);
    hb_prof_end(s);
    if (fused || fused_bits) {                                       // memcpy frames of a fused batch: the payload is filtered in place, frames that compressed leave at once
        const int rc = hb_launch_filter_batch(fused_bits ? HB_OP_BITSHUFFLE : HB_OP_SHUFFLE, d_jobs + nframes, nframes, max_n, typesize, s, 1);
        if (rc) return rc;
    }
    hb_prof_begin("k_stitch", s);
    hipLaunchKernelGGL(k_stitch, dim3(L.total_tiles), dim3(STITCH_THREADS), 0, s, desc, records, (const uint8_t *)nullptr, tile_pre, tile_suf,
                       (const EncPlan *)nullptr, 0u, (uint64_t)0, (uint8_t *)nullptr, (uint8_t *)nullptr, (uint8_t *)nullptr, (const uint8_t *)nullptr,
                       (fused || fused_bits) ? 1 : 0, (const BatchFrame *)d_bf, (const uint32_t *)tile_frame);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
