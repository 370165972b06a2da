// hb_lz4_region.hip — token discovery for an LZ4 block that comes WITHOUT a restart index (a frame written without
// HB_OPT_INDEX_TRAILER -- the drop-in default, since the reference's frames end at NBytesComp; blosc.go:369-371 -- or a frame the
// reference itself wrote), so that it decodes in parallel instead of on one wavefront.  Input contract: that of
// lz4.UncompressBlock (codec.go:77-84), nothing more.
//
// Two things make such a block serial: (1) nothing says where its sequences start -- token k+1 begins where token k ends;
// (2) a match may copy from up to 65535 bytes back, i.e. from output that some other part of the stream produces.
// Both are resolved by GUESSING in parallel and then VERIFYING exactly; wrong guesses are repaired by iteration, and anything
// that does not check out leaves the block to the single-wavefront decoder (k_dec_serial), so the result never depends on luck.
//
//   (1) Sequence boundaries (this file).  The stream is cut into <= 16384 regions (>= 8 KiB) at fixed byte positions b_r.  k_rg_parse: one
//       wavefront per region parses (window-parallel token parser, hb_lz4_region.h, no copies) to the first token at or after
//       b_{r+1}, starting at a token it spotted by its long length extension (a run of FF bytes) or else at b_r AS IF a token
//       started there; it keeps that exit, the output length, and a record {position, output so far} of its first 128 tokens and
//       of the first token in each of 128 slices of the region.  A parse started at a wrong byte is garbage, but LZ4 parses
//       synchronise: it usually falls onto a real token and stays on the real chain.  Beliefs "my first token = the furthest
//       position any region in front of me reaches" are then iterated (k_rg_pmax + k_rg_fix chip-wide, k_rg_settle with the
//       regions' state in LDS): a region whose belief changed walks a few tokens looking for a position of its record (the
//       parses have merged: exit unchanged, output length corrected by the difference) or is re-parsed in merge mode; what is
//       still moving after the chip-wide rounds (a stretch whose parses never synchronise) is finished by the last k_rg_settle
//       launch itself.  k_rg_scan checks the whole chain (entry[r] == exit[r-1] for every r, last exit == end of block): a
//       chain that passes is the true token chain, however it was found; the prefix sum of the output lengths gives every
//       region its place.
//   (2) Match sources.  A block whose matches never leave their 4 KiB chunk of output -- every block THIS library writes -- gets
//       its restart index rebuilt: k_rg_index walks the verified chain with output positions and writes the HBIX entry of every
//       4 KiB unit, or gives up at the first unit boundary that falls inside a match; the rebuilt index goes through k_dec_plan /
//       k_dec_indexed like a stored one, which trust neither.  Any other block (the reference's: one block, 64 KiB window) is
//       decoded symbolically from the same chain: hb_lz4_sym.hip.
//
// Everything malformed (offset 0, offset before the block, lengths running off the stream, output beyond cap) only raises
// plan->fail; k_dec_serial then decodes the block and reports what lz4.UncompressBlock would report.
#include "hb_lz4_region.h"
#include <cstdlib>

size_t hb_lz4_region_workspace(size_t n_out) { return rg_layout(n_out).total + ((hb_lz4_index_bound(n_out) + 255) & ~(size_t)255); }
// blocks below 256 KiB stay with the single wavefront (a dozen launches cost more than they save)
// ... and a stream that is longer than any block of a.cap bytes can be is malformed anyway (the workspace has regions for a.cap)
bool hb_lz4_region_wanted(const hb_dec_args &a) {
    return !a.index && !a.memcpy_payload && hb_indexless_parallel(a.n, a.cap) && a.n < 0xFFFFFFF0ull && a.cap < 0xFFFFFFF0ull && a.n <= a.cap + a.cap / 255 + 16;
}

// entry0: where the chain begins (an LZ4 block: its first byte; a Snappy block: behind the uvarint, SnPlan.hdr)
__device__ __forceinline__ void k_rg_init_body(RgPlan *plan, RgRegion *reg, uint32_t nreg, uint32_t rs, const uint32_t bx_, const uint32_t gx_, const uint32_t entry0 = 0u) {
    (void)bx_; (void)gx_;
    const uint32_t r = bx_ * blockDim.x + threadIdx.x;
    if (r == 0) {
        plan->pad[1] = 0; plan->pad[2] = 0; plan->pad[3] = entry0;
        plan->ok = 0; plan->fail = 0; plan->nreg = nreg; plan->rs = rs; plan->total = 0;
        uint32_t sh = 0; while (((uint64_t)RG_BUCKETS << sh) < rs) sh++;      // RG_BUCKETS << sh >= rs: every position of a region has a bucket
        plan->pad[0] = sh;
    }
    if (r >= nreg) return;
    RgRegion R;
    R.b = r * rs; R.entry = R.b; R.exit = RG_INVALID; R.outlen = 0; R.entry0 = R.b; R.exit0 = RG_INVALID; R.outlen0 = 0; R.ntrace = 0;
    R.needfull = 1; R.pad0 = R.b; R.opos = 0; R.pad1[0] = RG_INVALID; R.pad1[1] = R.pad1[2] = R.pad1[3] = 0;
    if (r == 0) { R.entry = entry0; R.entry0 = entry0; R.pad0 = entry0; }
    reg[r] = R;
}
__global__ void k_rg_init_sn(RgPlan *plan, RgRegion *reg, uint32_t nreg, uint32_t rs, const uint32_t *__restrict__ entry0) { k_rg_init_body(plan, reg, nreg, rs, blockIdx.x, gridDim.x, *entry0 < rs ? *entry0 : 0u); }
__global__ void k_rg_init(RgPlan *plan, RgRegion *reg, uint32_t nreg, uint32_t rs) { k_rg_init_body(plan, reg, nreg, rs, blockIdx.x, gridDim.x); }
__global__ void k_rg_init_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_init_body(j.plan, j.reg, j.nreg, j.rs, blockIdx.x, gridDim.x); }

#ifndef RG_LEAN_PARSE
#define RG_LEAN_PARSE 1
#endif
#ifndef RG_MM_RESTART
#define RG_MM_RESTART 1024u        // tokens a re-parse walks in merge mode before it starts over and records (0: never)
#endif
#ifndef RG_HEADPARSE
#define RG_HEADPARSE 0xFFFFFFFFu     // a spotted token further into its region than this is a checkpoint of the parse from the first byte, not its start (OFF: see below)
#endif
#ifndef RG_SN_GIVEUP
#define RG_SN_GIVEUP 4096u          // one-byte hops after which a guessed Snappy parse gives up (a region inside one 64 KiB literal makes ~2000 and then hands on an exit nobody needs)
#endif
#ifndef RG_FAT_HOLD
#define RG_FAT_HOLD 8u
#endif
// ---- (1a) parse a region from its entry to the first token at / after the next region's start; no copies ----
// parses region r from its believed first token (reg[r].entry) to the first token at / after the next region's start; one wavefront.
// first: nothing is on record yet.  s_win: RG_PWIN + 128 bytes, s_tq: DTQ entries, both this wave's own.
// tok / tokcap: the token store (hb_lz4_region.h), or NULL.
template <int CODEC = RG_LZ4>
__device__ __forceinline__ void rg_parse_region(const uint8_t *__restrict__ src, const uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces, const uint32_t r,
                                                const int first, uint8_t *s_win, uint2 *s_tq, const int lane, uint2 *tok = nullptr, const uint32_t tokcap = 0) {
    const uint32_t nreg = plan->nreg;
    RgRegion *R = reg + r;
    {
        uint32_t start = RFL(R->entry);
        const uint64_t bnext = (r + 1 < nreg) ? (uint64_t)RFL(reg[r + 1].b) : n_src;
        uint2 *tr = traces + (size_t)r * RG_TRACE;
        uint2 *const tk = tok ? tok + (size_t)r * tokcap : nullptr;
        const uint32_t rb = RFL(R->b), bsh = plan->pad[0];                 // bucket = (position - b) >> bsh
        // A region that was parsed before is re-parsed in MERGE mode: nothing is recorded, the first token of every bucket is compared
        // with the one on record, and at the first match the two parses have merged -- exit unchanged, output length corrected by the
        // difference -- typically a few hundred tokens in instead of the whole region.  A parse that never meets the recorded one
        // just delivers its exit and output length; the record stays what the region's FIRST parse left: that one started at
        // the region's first byte and has, as a rule, fallen onto the true chain, while a later first token may come from a stray
        // parse further up (in periodic data stray and true chains run side by side and never meet: a record overwritten by a
        // stray parse would make the correction that follows one region behind pay a full parse per region too).
        bool mm = !first && RFL(R->exit0) != RG_INVALID;
        const uint32_t exit0 = RFL(R->exit0), outlen0 = RFL(R->outlen0), rec0 = RFL(R->entry0);     // (rec0: where the recorded parse began -- see the guess rules of the Snappy parser)
        if (!mm) for (uint32_t i = lane; i < RG_BUCKETS; i += 64) { uint2 t; t.x = RG_INVALID; t.y = 0; tr[RG_DENSE + i] = t; }
        uint64_t si = start, wpos = 0, out = 0;
        uint32_t wlen = 0, wsh = 0, nq = 0, ntok = 0, exitp = RG_INVALID, lastbk = RG_INVALID;
        uint32_t last_ntok = DEC_BPERM_MIN;                              // (dec_fill_lean: how the previous window's chain was followed)
        uint32_t fat_hold = 0;                                           // windows the fat parser still has
        uint32_t checkpoint = RG_INVALID;                                // (LZ4, first parse) a spotted token the parse from the first byte has to land on
        uint32_t clean = 0, nevents = 0;                                 // (Snappy, a guessed parse) elements since the last one that cannot be on a chain; how many of those so far
        bool invalid = false, merged = false;
        uint32_t mpos = 0, mcum = 0, mc0 = 0;
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
        if (CODEC == RG_SNAPPY && first && r != 0u) {
            // A better first guess than "an element starts at my first byte" where it matters.  In a stretch the encoder found little in, the chain is
            // literals of kilobytes -- a whole 64 KiB block as ONE literal, F4 FF FF + the block, where it found nothing -- with a copy here and there, and
            // a parse that starts inside literal bytes hops through them (any byte is a tag) and past the next header with them: stray and true chain
            // do not meet, the region's exit is garbage, and the belief rounds repair such stretches region by region (measured, 1 GiB of shuffled
            // floats: a third of the exits wrong after the first parse, the chain settled by halves over six rounds of re-parses: 4-5 ms; float64 with
            // three noise planes: 13 ms).  The headers of long literals are easy to tell from noise though: a tag F4 whose chain -- the literal, then
            // whatever follows it -- runs for 24 elements without meeting one that no 64 KiB-block encoder writes (a copy-4, a literal of more than
            // 64 KiB, an element that leaves the stream).  Garbage meets one every fourth element: 0.74^24 is one false header in 1400 candidates, and a
            // region holds ~125 bytes F4.  So: every F4 of the region is a candidate, 64 at a time are followed by a lane each (straight from memory: the
            // chain leaves the region at once), and the parse starts at the first one that holds.  Only a guess, like the other one: the chain is
            // verified as a whole.  Where the stream is dense with copies no candidate holds and the parse starts at the region's first byte, as before.
            uint32_t found = RG_INVALID, ncand = 0;
            uint32_t *const cl = (uint32_t *)s_tq;                         // candidate positions (DTQ * 2 words: room for 128)
            auto validate = [&](const uint32_t cnt) __attribute__((always_inline)) {
                const uint32_t q = (uint32_t)lane < cnt ? cl[lane] : RG_INVALID;
                bool ok = q != RG_INVALID;
                uint64_t p = q, cum = 0;
                uint32_t nlong = 0;                                        // literals of more than 256 bytes on the way
                for (int i = 0; i < 24; i++) {
                    const bool go = ok && p < n_src;
                    if (!hb_ballot(go)) break;
                    if (go) { const uint64_t p0 = p; ok = rg_step_serial<RG_SNAPPY>(src, n_src, p, cum); if (ok && p - p0 > 259u) nlong++; }
                }
                // ... and the chain must be a chain of LITERALS: where the stream is dense with copies everything synchronises -- a stray F4 jumps, lands,
                // falls onto the chain within a few elements and runs on without a fault (measured: 15 957 of 16 377 regions "found" a header that way
                // and started in their middle; every one of them had to be parsed again).  Four long literals among the 24 elements: half of a noise
                // stretch's elements are, next to none of a copy-dense one's.
                ok = ok && nlong >= 4u;
                const unsigned long long m = hb_ballot(ok);
                if (m) found = (uint32_t)__builtin_amdgcn_readlane(q, (int)__builtin_ctzll(m));   // (the list is in stream order)
            };
            for (uint64_t at = start; at < bnext && found == RG_INVALID; at += RG_PWIN - 64u) {
                refill(at);
                const uint32_t span = (uint32_t)((bnext - at) < (uint64_t)wlen ? (bnext - at) : (uint64_t)wlen);
                for (uint32_t k0 = 0; k0 < span && found == RG_INVALID; k0 += 64u) {
                    const uint32_t k = k0 + (uint32_t)lane;
                    const bool hit = k < span && s_win[wsh + k] == 0xF4u;
                    const unsigned long long m = hb_ballot(hit);
                    if (!m) continue;
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (hit) cl[ncand + rank] = (uint32_t)at + k;
                    ncand += (uint32_t)__builtin_popcountll(m);
                    if (ncand >= 64u) {
                        wave_sync();
                        validate(64u);
                        const uint32_t rest = cl[64u + (uint32_t)lane];
                        wave_sync();
                        ncand -= 64u;
                        if ((uint32_t)lane < ncand) cl[lane] = rest;
                    }
                }
            }
            if (found == RG_INVALID && ncand) { wave_sync(); validate(ncand); }
            wave_sync();
            if (found != RG_INVALID && found >= start && found < bnext) { start = found; si = start; clean = 1000u; }     // (a header that held: on the chain from the first element)
            wlen = 0; wpos = 0;                                              // (the window was only the scan's)
        }
        if (CODEC == RG_LZ4 && first && r != 0u) {
            // A better first guess than "a token starts at my first byte".  Where literal runs are long, tokens are rare and a parse that
            // starts inside literal bytes may not meet one for the whole region (measured on this library's own streams: the exits
            // of 36 % of the regions wrong after the first pass, in chains of up to 15 regions); where the stream is periodic (a
            // plane of long runs) a stray parse never falls onto the true chain at all.  Both kinds of stream are full of LONG
            // LENGTH EXTENSIONS, and those are easy to spot and say where a token is: a run of >= 15 bytes FF (found 1 KiB per step
            // as an aligned all-FF quadword) is the extension of a literal length when the byte in front of it is F? -- that byte
            // is the token -- and of a match length otherwise -- the next token starts behind the byte that ends the run.  The first
            // such run in the region decides where the parse starts.  Only a guess, like the other one: the chain is verified as a
            // whole, and a pattern in the data itself costs what a stray start costs anyway.
            uint32_t found = RG_INVALID;                                  // token position, relative to `start`
            int tries = 4;
            // The region BEGINS inside such a run (a match of 128 MiB has 526 KB of them: 64 regions): no token starts before the run
            // ends.  Parsing "as if" from an FF byte would read the rest of the run as a literal length and land up to 255 bytes per
            // FF further on -- a stray exit far behind the region, which the first belief round (a prefix maximum) would hand to
            // every region in between, whose repair then walks region by region (measured, D-f64 as the reference writes it: 47 ms
            // in k_rg_settle, then the single wavefront after all).  So: start behind the byte that ends the run, or -- when the
            // whole region is FF -- record a region without a token (exit == entry: it says nothing).
            {
                bool inrun = RFL((uint32_t)src[start - 1u]) == 255u;
                if (inrun) {
                    const uint64_t i0 = (uint64_t)start + (uint32_t)lane;
                    inrun = hb_ballot((uint32_t)lane < 16u && i0 < n_src && src[i0] != 255u) == 0ull;
                }
                if (inrun) {
                    uint64_t e = bnext;                                   // the first byte that is not FF
                    for (uint64_t at = start; at < bnext; at += 1024u) {
                        const uint64_t idx = at + (uint64_t)lane * 16u;
                        uint32_t fb = 16u;                                // my first byte that is not FF
                        if (idx + 16u <= n_src) {
                            const u32x4 v = ld16u(src + idx);
                            const uint32_t w[4] = {~v.x, ~v.y, ~v.z, ~v.w};
#pragma unroll
                            for (int k = 3; k >= 0; k--) if (w[k]) fb = 4u * (uint32_t)k + ((uint32_t)__builtin_ctz(w[k]) >> 3);
                        } else {
                            for (uint32_t k = 0; k < 16u; k++) if (idx + k >= n_src || src[idx + k] != 255u) { fb = k; break; }
                        }
                        const unsigned long long m = hb_ballot(fb != 16u);
                        if (m) { const int j = __builtin_ctzll(m); e = at + 16u * (uint64_t)j + __builtin_amdgcn_readlane(fb, j); break; }
                    }
                    start = (uint32_t)(e + 1u < bnext ? e + 1u : bnext);
                    si = start;
                    tries = 0;                                            // (no search below)
                }
            }
            for (uint64_t at = start; at < bnext && found == RG_INVALID && tries > 0; at += RG_PWIN - 512u) {
                refill(at);
                const uint32_t endo = wsh + wlen;                          // LDS offsets [wsh, endo) hold the window
                uint32_t from = wsh;                                       // runs that end in front of this offset were looked at
                for (uint32_t g0 = 0; 16u * g0 < endo && found == RG_INVALID && tries > 0; g0 += 64) {
                    const uint32_t g = g0 + (uint32_t)lane;
                    u32x4 v; v.x = 0; v.y = 0; v.z = 0; v.w = 0;
                    if (16u * g + 16u <= endo) v = ((const u32x4 *)s_win)[g];
                    const bool h0 = (v.x & v.y) == 0xFFFFFFFFu && 16u * g >= from, h1 = (v.z & v.w) == 0xFFFFFFFFu && 16u * g + 8u >= from;
                    const unsigned long long m = hb_ballot(h0 || h1);
                    if (!m) continue;
                    const int j = __builtin_ctzll(m);
                    const uint32_t q = 16u * (g0 + (uint32_t)j) + (__builtin_amdgcn_readlane((uint32_t)h0, j) ? 0u : 8u);   // an all-FF quadword
                    // where the run starts ...
                    uint32_t rs0 = q;
                    bool open = true;
                    while (open && rs0 > wsh) {
                        const uint32_t idx = rs0 - 1u - (uint32_t)lane;
                        const uint32_t byte = (rs0 >= 1u + (uint32_t)lane && idx >= wsh) ? (uint32_t)s_win[idx] : 0u;
                        const unsigned long long nm = hb_ballot(byte != 255u);
                        if (nm) { rs0 -= (uint32_t)__builtin_ctzll(nm); open = false; } else rs0 -= 64u;
                    }
                    // ... and where it ends (the first byte that is not FF belongs to the extension)
                    uint32_t re = q + 8u;
                    bool eopen = true;
                    while (eopen && re < endo) {
                        const uint32_t idx = re + (uint32_t)lane;
                        const uint32_t byte = idx < endo ? (uint32_t)s_win[idx] : 255u;
                        const unsigned long long nm = hb_ballot(byte != 255u);
                        if (nm) { re += (uint32_t)__builtin_ctzll(nm); eopen = false; } else re += 64u;
                    }
                    tries--;
                    if (!open && rs0 > wsh && ((uint32_t)s_win[rs0 - 1u] >> 4) == 15u) found = (uint32_t)(at - start) + (rs0 - 1u - wsh);
                    else if (!open && !eopen && re + 1u < endo) found = (uint32_t)(at - start) + (re + 1u - wsh);
                    from = eopen ? endo : re + 1u;                        // (a run that leaves the window on either side is skipped)
                    if (found == RG_INVALID) g0 = (from >> 4) / 64u * 64u - 64u;     // go on behind it (the loop adds 64)
                }
            }
            // (parsing from the first byte as well and keeping that parse when it comes by the spotted token was measured: the dense
            // regions that miss it pay twice, +2 ms per GiB; starting at the token costs the fix round 0.4 ms instead)
            if (found != RG_INVALID && (uint64_t)start + found < bnext) {
                // A token spotted DEEP in the region costs two serial walks of the head in front of it later on, everybody else idle (the reference-written
                // headline frame: the sequence with the 256 MiB literal run sits 24 KB into its region behind 6000 dense tokens -- 0.42 ms in the re-parse
                // behind k_rg_fix).  -DRG_HEADPARSE=2048 parses that head HERE, from the region's first byte, with the spotted token as a checkpoint (a
                // parse that is on the chain by then lands on it and the record is one piece; one that passes over it starts over at the token).
                // Measured in round 4 and left OFF: the reference-written frame 7.88 -> 7.62 ms (k_rg_fix 0.42 -> 0.12, k_rg_parse +0.13), but this
                // library's own frames lose -- their deep tokens sit behind literal bytes, the head parse passes over them and is wasted: D-f32
                // k_rg_parse 1.28 -> 1.44 ms, D-f64 0.69 -> 0.77 and its fix round 0.24 -> 0.43.
                if (found > RG_HEADPARSE) checkpoint = start + found;
                else { start += found; si = start; }
            }
        }
        for (;;) {
            if (si >= bnext) { exitp = (uint32_t)si; break; }           // (si <= n_src always; bnext <= n_src)
            if (si < wpos || si - wpos + 1024u > wlen) { if (si != wpos || wlen == 0) refill(si); }
            uint32_t rel = (uint32_t)(si - wpos);
            // Two window parsers.  LEAN (dec_fill_lean, hb_dec_common.h: the indexed decoder's): a lane only works out how long the sequence at its
            // byte would be, taking a length extension for ONE byte; the queue gets the positions of the real tokens and their fields are parsed
            // below, by lanes that all hold one.  FAT (rg_fill): every lane parses its "as if" token in full, extensions of up to 24 bytes included.
            // Lean is the faster one per window (bit-shuffled integers: 0.82 against 1.39 ms per GiB) but loses the rest of a window at every
            // token with a longer extension, and where the stream is made of those (a plane of runs: one 19-byte sequence per 4 KiB chunk) a region
            // is a latency chain of window passes per token (measured with lean alone: 8.5 ms instead of 1.2 -- 40 such regions of 16 384 are
            // the launch).  So a longer extension switches to the fat parser, which stays until RG_FAT_HOLD windows in a row met none.
            const bool fat = CODEC != RG_LZ4 || !RG_LEAN_PARSE || fat_hold != 0u;
            bool stop, refat = false, recp = false;
            if constexpr (CODEC == RG_SNAPPY) stop = sn_rg_fill(s_win, wsh, wlen, rel, nq, s_tq, lane, first && r != 0u);        // (elements: entry = {position, output bytes})
            else if (fat) stop = rg_fill(s_win, wsh, wlen, rel, nq, s_tq, lane);
            else stop = dec_fill_lean(s_win, wsh, wlen, wlen, rel, nq, s_tq, lane, last_ntok);
            if (fat_hold) fat_hold--;
            bool done = false;
            while (nq > 0u) {                                           // account for the queued tokens, 64 at a time
                uint32_t cntb = nq < 64u ? nq : 64u;
                uint2 e;
                if (!fat) {
                    const uint32_t tp = (uint32_t)lane < cntb ? (uint32_t)((const uint16_t *)s_tq)[lane] : 0u;
                    const uint32_t w = dec_read4(s_win, wsh + tp);
                    const uint32_t t = w & 255u, b1 = (w >> 8) & 255u;
                    const bool big = t >= 0xF0u;
                    const uint32_t lit = big ? 15u + b1 : (t >> 4);
                    const uint32_t x = dec_read4(s_win, wsh + tp + (big ? 2u : 1u) + lit);
                    const uint32_t mb = (x >> 16) & 255u, mn = t & 15u;
                    const uint32_t mlen = mn == 15u ? 19u + mb : 4u + mn;
                    e.x = tp; e.y = lit | (mlen << 16);
                    const unsigned long long xm = hb_ballot((uint32_t)lane < cntb && ((big && b1 == 255u) || (mn == 15u && mb == 255u)));
                    if (xm) {                                           // what was queued behind that token is not on the chain: go on from it, fat
                        const uint32_t jx = (uint32_t)__builtin_ctzll(xm);
                        rel = (uint32_t)__builtin_amdgcn_readlane(tp, (int)jx);
                        cntb = jx; nq = jx; refat = true; fat_hold = RG_FAT_HOLD;
                        if (jx == 0u) break;
                    }
                } else {
                    e = s_tq[lane];
                    if (hb_ballot((uint32_t)lane < cntb && ((e.y & 0xFFFFu) >= 270u || (e.y >> 16) >= 274u))) fat_hold = RG_FAT_HOLD;
                }
                const uint32_t lit = e.y & 0xFFFFu, mlen = e.y >> 16;
                const uint64_t ap = wpos + (e.x & 0xFFFFu);
                const unsigned long long over = hb_ballot((uint32_t)lane < cntb && ap >= bnext);
                const uint32_t cnt = over ? (uint32_t)__builtin_ctzll(over) : cntb;
                uint32_t lo = 0;                                         // first lane of the batch that goes on record
                if constexpr (CODEC == RG_SNAPPY) {
                    // one-byte hops of a guess (sn_rg_fill): the record starts over behind the last of them (see the one-element path below for why)
                    const unsigned long long evm = hb_ballot((uint32_t)lane < cnt && (e.x >> 31) != 0u);
                    if (evm) {
                        const int last = 63 - __builtin_clzll(evm);
                        nevents += (uint32_t)__builtin_popcountll(evm);
                        if (nevents >= RG_SN_GIVEUP) { invalid = true; done = true; nq = 0; break; }
                        lo = (uint32_t)last + 1u;
                        start = RFL(__builtin_amdgcn_readlane((uint32_t)ap, last)) + 1u; out = 0; ntok = 0; lastbk = RG_INVALID; clean = 0;
                    }
                }
                if (CODEC == RG_LZ4 && checkpoint != RG_INVALID) {
                    const unsigned long long ge = hb_ballot((uint32_t)lane < cntb && (uint32_t)ap >= checkpoint);
                    if (ge) {
                        const int j = __builtin_ctzll(ge);
                        if (RFL(__builtin_amdgcn_readlane((uint32_t)ap, j)) == checkpoint) checkpoint = RG_INVALID;       // landed on it: one record
                        else { rel = checkpoint - (uint32_t)wpos; recp = true; nq = 0; break; }                              // passed over it: start over there
                    }
                }
                const bool mine = (uint32_t)lane >= lo && (uint32_t)lane < cnt;
                const uint32_t olen = mine ? (CODEC == RG_SNAPPY ? e.y : lit + mlen) : 0u;
                const uint32_t incl = wave_incl_scan_dpp(olen);
                const uint32_t idx = ntok + (uint32_t)lane - lo;
                // trace: the first RG_DENSE tokens, and the first token that starts in each bucket of the region's stream range
                const uint32_t bk = mine ? ((uint32_t)ap - rb) >> bsh : RG_INVALID;
                const uint32_t pbk = wave_shr1(bk, lastbk);
                if (mm) {
                    bool hit = false; uint32_t c0 = 0;
                    if ((uint32_t)lane < cnt && bk != pbk && bk < RG_BUCKETS) { const uint2 o = tr[RG_DENSE + bk]; hit = o.x == (uint32_t)ap && o.x >= rec0; c0 = o.y; }
                    const unsigned long long hm = hb_ballot(hit);
                    if (hm) {
                        const int j = __builtin_ctzll(hm);
                        mpos = RFL(__builtin_amdgcn_readlane((uint32_t)ap, j));
                        mcum = (uint32_t)out + (uint32_t)__builtin_amdgcn_readlane(incl - olen, j);
                        mc0 = (uint32_t)__builtin_amdgcn_readlane(c0, j);
                        merged = true; done = true; nq = 0;
                        break;
                    }
                } else if (mine) {
                    uint2 t; t.x = (uint32_t)ap; t.y = (uint32_t)(out + incl - olen);
                    if (idx < RG_DENSE) tr[idx] = t;
                    if (bk != pbk && bk < RG_BUCKETS) tr[RG_DENSE + bk] = t;
                    if (tk && idx < tokcap) { uint2 k; k.x = (uint32_t)ap; k.y = e.y; tk[idx] = k; }
                }
                if (cnt > lo) lastbk = (uint32_t)__builtin_amdgcn_readlane(bk, (int)cnt - 1);
                out += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
                ntok += cnt - lo; clean += cnt - lo;
                if (over) { exitp = RFL(__builtin_amdgcn_readlane((uint32_t)ap, (int)__builtin_ctzll(over))); done = true; nq = 0; break; }
                if (!fat) {
                    const uint16_t rest = ((const uint16_t *)s_tq)[64 + lane < DTQ ? 64 + lane : 0];
                    nq -= cntb;
                    ((uint16_t *)s_tq)[lane] = rest;
                } else {
                    const uint2 rest = s_tq[64 + lane < DTQ ? 64 + lane : 0];
                    nq -= cntb;
                    if ((uint32_t)lane < nq) s_tq[lane] = rest;
                }
            }
            if (done) break;
            if (refat) { si = wpos + rel; continue; }
            if (CODEC == RG_SNAPPY && RG_MM_RESTART && mm && ntok >= RG_MM_RESTART) {
                // A re-parse that has not met the region's record after this many tokens will hardly meet it at all: that record is a stray parse's (the
                // guess never fell onto the chain), and left as it is nobody behind this launch can use it -- the region's restart index entries / unit starts
                // would be found by a wave walk of the whole region, everybody else idle (measured: 518 of 16 378 regions on this library's own headline
                // frame, 0.16 ms in k_rg_index; 0.39 ms in k_snr_units on the foreign Snappy frame).  So the parse starts over from its entry and RECORDS
                // (dense tokens, bucket entries, token store): the record is then this chain's from its first token on.  (The tokens walked so far are
                // walked twice: ~a tenth of a region.)  Snappy only -- measured for LZ4 too and worse there: the re-parse launch lasts as long as its
                // slowest region, and the regions that start over are the slowest (own headline frame: k_rg_index 0.38 -> 0.35 ms but the launches behind
                // k_rg_fix 0.29 -> 0.36 and k_rg_settle 0.15 -> 0.18; the reference-written frame 7.75 -> 8.16 ms).
                mm = false;
                si = start; out = 0; ntok = 0; lastbk = RG_INVALID; nq = 0; wlen = 0; wpos = 0;
                wave_sync();
                for (uint32_t i = lane; i < RG_BUCKETS; i += 64) { uint2 t; t.x = RG_INVALID; t.y = 0; tr[RG_DENSE + i] = t; }
                wave_sync();
                continue;
            }
            if (recp || (CODEC == RG_LZ4 && checkpoint != RG_INVALID && wpos + rel > checkpoint)) {
                // the record starts over at the spotted token (bucket entries in front of it are ignored by their readers: RgRegion.entry0)
                si = checkpoint; start = checkpoint; checkpoint = RG_INVALID;
                out = 0; ntok = 0; lastbk = RG_INVALID; nq = 0;
                continue;
            }
            const bool moved = (wpos + rel) != si;
            si = wpos + rel;
            if (CODEC == RG_SNAPPY && si > n_src) { invalid = true; break; }     // (a literal that runs off the stream: a stray parse, or a corrupt block)
            if (moved && !stop) continue;
            if (si >= bnext) continue;
            if (moved && si - wpos + 1024u > wlen && wpos + wlen < n_src) continue;      // stopped at the window edge: refill first
            // ---- one token the slow way: length extensions of any size ----
            uint32_t tokstart, ll, ml = 0;
            uint64_t p;
            if constexpr (CODEC == RG_SNAPPY) {
                // (an element near the window's edge, or a literal whose length takes three or four bytes)
                if (si < wpos || si + 8u > wpos + wlen) refill(si);
                rel = (uint32_t)(si - wpos);
                tokstart = (uint32_t)si;
                SnElem e;
                // Literals of more than 256 bytes and copies with 4-byte offsets come here one at a time, and the discovery does not take every one at its
                // word.  The Snappy encoders whose streams can decode in parallel at all compress 64 KiB at a time: no literal of more than 64 KiB, no
                // offset above 65535 -- and so no copy-4 element at all.  A stray parse meets those all the time: one byte in four reads as a copy-4 tag,
                // F8 / FC as "a literal, its length in the next 3-4 bytes" (up to 4 GiB: the end of the parse).  Measured (1 GiB of shuffled floats as the
                // oracle's encoder writes it; a CPU model of the parse agrees): a third of the regions' first parses died on F8 / FC before they fell onto
                // the chain -- Snappy parses synchronise less readily than LZ4's -- and the chain then settled by halves, round after round.  So:
                // (a) no chain goes through such an element: a parse from a handed-down entry that meets one ends as invalid (a stream that really holds
                //     them -- klauspost's one-block streams -- does not verify and goes to the single wavefront, where its copies across 64 KiB would have
                //     sent it anyway);
                // (b) a parse that is only a GUESS (a region's first parse, from its first byte) takes one for what it almost surely is -- proof that it
                //     is not on the chain: it moves on by ONE byte and keeps looking.  That ends a stray episode after four elements on average, before it
                //     meets the tag that does the damage: F4, "a literal of up to 64 KiB", an exit regions further on that the belief rounds hand on.
                const bool guess = first && r != 0u;
                bool okp = sn_parse_uniform(s_win + wsh + rel, n_src - si, e);
                if (okp && (e.kind == 3u || e.lit > 65536u || e.lit > n_src - (si + e.hdr))) okp = false;
                // (c) ... and while a guess has not run 16 elements in a row without such proof, it does not follow a literal of more than 256 bytes either:
                //     the few elements a stray parse reads before it falls onto the chain hold the tag F4 once in 256, and that jump -- up to two regions
                //     ahead, landing on the chain -- makes the regions it flies over look empty until the next round takes it back (measured: 2200 of
                //     16 377).  A real literal passed over that way is hopped through byte by byte and the chain met again behind it.
                if (okp && guess && clean < 16u && e.lit > 256u) okp = false;
                if (!okp) {
                    if (guess) {
                        // ... and its RECORD starts over: what the belief rounds merge with must be a parse that follows the rules of every other parse
                        // from its first token on ("once two parses meet they stay together"), and this one has just broken them.  Dense tokens and the
                        // token count restart; bucket entries in front of the new start are ignored by whoever reads them (RgRegion.entry0).
                        // A guess that has broken them 64 times is inside literal bytes (one element in four of noise is impossible; a region inside a 64 KiB
                        // literal would grind through 2000 of these one-element steps: measured, the first parse 1.6 -> 3.9 ms): it gives up -- no exit is
                        // better than a wrong one, and the belief rounds hand such a region the exit of the literal it lies in.
                        if (++nevents >= RG_SN_GIVEUP) { invalid = true; break; }
                        si += 1u; clean = 0;
                        start = (uint32_t)si; out = 0; ntok = 0; lastbk = RG_INVALID;
                        if (si >= bnext) { exitp = (uint32_t)si; break; }
                        continue;
                    }
                    invalid = true; break;
                }
                clean++;
                p = si + e.hdr;
                p += e.lit;
                ll = (uint32_t)e.lit; ml = e.mlen;
            } else {
            if (si < wpos || si >= wpos + wlen) refill(si);
            rel = (uint32_t)(si - wpos);
            tokstart = (uint32_t)si;
            if (checkpoint != RG_INVALID && tokstart == checkpoint) checkpoint = RG_INVALID;      // (landed on the spotted token)
            const uint32_t tok = RFL((uint32_t)s_win[wsh + rel]);
            rel++;
            ll = tok >> 4;
            {
                const uint64_t span = n_src - wpos;
                if (ll == 15u && !dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel, ll, lane)) { invalid = true; break; }
            }
            p = wpos + rel;
            if ((uint64_t)ll > n_src - p) { invalid = true; break; }
            p += ll;
            if (p != n_src) {                                           // (p == n_src: the block's final, literal-only sequence)
                if (n_src - p < 2) { invalid = true; break; }
                p += 2;
                ml = (tok & 15u) + 4u;
                if ((tok & 15u) == 15u) {
                    if (p < wpos || p - wpos + 64u > wlen) refill(p);
                    uint32_t rel2 = (uint32_t)(p - wpos);
                    const uint64_t span = n_src - wpos;
                    if (!dec_read_ext(s_win + wsh, 0, 0u, wlen, src + wpos, (uint32_t)(span < 0xFFFFFFF0ull ? span : 0xFFFFFFF0ull), rel2, ml, lane)) { invalid = true; break; }
                    p = wpos + rel2;
                }
            }
            }
            {
                const uint32_t bk = (tokstart - rb) >> bsh;
                if (mm) {
                    if (bk != lastbk && bk < RG_BUCKETS) {
                        const uint32_t ox = RFL(tr[RG_DENSE + bk].x), oy = RFL(tr[RG_DENSE + bk].y);
                        if (ox == tokstart && ox >= rec0) { mpos = tokstart; mcum = (uint32_t)out; mc0 = oy; merged = true; break; }
                    }
                } else if (lane == 0) {
                    uint2 t; t.x = tokstart; t.y = (uint32_t)out;
                    if (ntok < RG_DENSE) tr[ntok] = t;
                    if (bk != lastbk && bk < RG_BUCKETS) tr[RG_DENSE + bk] = t;
                    if (tk && ntok < tokcap) { uint2 k; k.x = tokstart; k.y = (ll <= 0xFFFFu && ml <= 0xFFFFu) ? (ll | (ml << 16)) : RG_INVALID; tk[ntok] = k; }
                }
                lastbk = bk;
            }
            out += (uint64_t)ll + ml;
            ntok++;
            si = p;
            if (CODEC == RG_LZ4 && (ll >= 270u || ml >= 274u)) fat_hold = RG_FAT_HOLD;
            if (out > 0xFFFFFFF0ull) { invalid = true; break; }
        }
        if (out > 0xFFFFFFF0ull) invalid = true;
        if (lane == 0) {
            const uint32_t ex = invalid ? RG_INVALID : exitp;
            if (merged) { R->exit = exit0; R->outlen = mcum + (outlen0 - mc0); R->pad0 = mpos; }
            else if (mm) { R->exit = ex; R->outlen = (uint32_t)out; R->pad0 = RG_INVALID; }       // (nothing on record lies on this chain)
            else {
                R->exit = ex; R->outlen = (uint32_t)out;
                R->entry0 = start; R->exit0 = ex; R->outlen0 = (uint32_t)out; R->ntrace = ntok < RG_DENSE ? ntok : RG_DENSE;
                R->pad0 = start;                                         // the whole record lies on this parse
                R->pad1[0] = (tk && ntok <= tokcap && !invalid) ? ntok : RG_INVALID;
                if (first) R->entry = start;                             // (the guess may have moved to a spotted token)
            }
            R->needfull = 0;
        }
    }
}

template <int CODEC = RG_LZ4>
__device__ __forceinline__ void k_rg_parse_body(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces, int first, uint2 *tok, uint32_t tokcap, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    const int lane = threadIdx.x;
    const uint32_t nreg = plan->nreg;
    if (!first && plan->pad[1] == 0u) return;                            // no region asked for a re-parse
    for (uint32_t r = bx_; r < nreg; r += gx_) {
        if (!RFL(reg[r].needfull)) continue;
#ifdef RG_DEBUG_TIMES
        const uint64_t t0 = wall_clock64();
        const uint32_t had0 = RFL(reg[r].exit0);
#endif
        rg_parse_region<CODEC>(src, n_src, plan, reg, traces, r, first, s_win, s_tq, lane, tok, tokcap);
        wave_sync();
#ifdef RG_DEBUG_TIMES
        if (!first && lane == 0) { reg[r].pad1[3] += (uint32_t)((wall_clock64() - t0) / 100); reg[r].pad1[2] = had0 == RG_INVALID ? 2u : 1u; }     // microseconds; re-parse kind
#endif
    }
}
__global__ __launch_bounds__(64) void k_rg_parse(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces, int first, uint2 *tok, uint32_t tokcap) { k_rg_parse_body(src, n_src, plan, reg, traces, first, tok, tokcap, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_snr_parse(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces, int first) { k_rg_parse_body<RG_SNAPPY>(src, n_src, plan, reg, traces, first, nullptr, 0u, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_rg_parse_b(const RgJob *__restrict__ jobs, int first) { const RgJob j = jobs[blockIdx.y]; k_rg_parse_body(j.src, j.n_src, j.plan, j.reg, j.traces, first, j.tok, j.tokcap, blockIdx.x, gridDim.x); }

// ---- (1b) settle the chain.  Belief of every region about its first token: the furthest position any predecessor's parse reaches (an
// exclusive prefix maximum of the exits).  On the true chain exits are monotone, so this is the predecessor's exit; a token that
// spans many regions (a literal run of MiB) reaches all of them in ONE step instead of one region per step, and the garbage parses
// of the regions inside such a run (they started at bytes that are literals) are out-voted as long as they stay local.  A region
// whose belief changed re-parses from the new entry one token at a time until it lands on a position of its recorded trace (the
// parses have merged: exit unchanged, output length corrected by the difference) or asks for a full parse.  One workgroup, the
// regions' state in LDS, iterated until nothing moves. ----
// FINISH: the last launch.  What is still moving by then is a long thin chain -- a stretch of the stream whose parses never fall onto the
// true chain by themselves (periodic data: stray and true chains run side by side), so that every region has to wait for its
// predecessor's exit and be parsed from there.  Launch pairs would cost more than the hops: this kernel parses the regions that
// ask for it itself (RG_FPARSERS wavefronts) and goes on, until nothing moves or RG_MAXHOPS.
template <bool FINISH, uint32_t MAXR, int CODEC = RG_LZ4>
__device__ __forceinline__ void k_rg_settle_body(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    // (LDS: entry + exit of every region, one bit per region for "waits for a full parse"; output lengths stay in global memory --
    // they are written a few times per launch, read never)
    __shared__ uint32_t s_entry[MAXR], s_exit[MAXR], s_needb[MAXR / 32];
    __shared__ uint32_t s_pm[1024];
    __shared__ uint32_t s_changed, s_pend, s_nlist;
    __shared__ __attribute__((aligned(16))) uint8_t s_pwin[FINISH ? RG_FPARSERS : 1][FINISH ? RG_PWIN + 128 : 16];
    __shared__ __attribute__((aligned(16))) uint2 s_ptq[FINISH ? RG_FPARSERS : 1][FINISH ? DTQ : 2];
    __shared__ uint32_t s_list[FINISH ? RG_FLIST : 1];
    const int t = threadIdx.x;
    const uint32_t nreg = plan->nreg, bsh = plan->pad[0], rs = plan->rs;
    if (plan->pad[2]) return;                                           // an earlier launch came to a standstill with nothing pending
    constexpr uint32_t PER = MAXR / 1024;
    auto need = [&](uint32_t r) __attribute__((always_inline)) -> bool { return (s_needb[r >> 5] >> (r & 31u)) & 1u; };
    auto set_need = [&](uint32_t r, bool v) __attribute__((always_inline)) { if (v) atomicOr(&s_needb[r >> 5], 1u << (r & 31u)); else atomicAnd(&s_needb[r >> 5], ~(1u << (r & 31u))); };
    for (uint32_t k = (uint32_t)t; k < MAXR / 32; k += 1024u) s_needb[k] = 0u;
    if (t == 0) { s_changed = 0; s_pend = 0; s_nlist = 0; }
    __syncthreads();
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        if (r < nreg) { s_entry[r] = reg[r].entry; s_exit[r] = reg[r].exit; if (reg[r].needfull) atomicOr(&s_needb[r >> 5], 1u << (r & 31u)); }
    }
    __syncthreads();
    bool capped = true;                                                 // left the loop because of the iteration cap, still moving
  for (int hop = 0; hop < (FINISH ? RG_MAXHOPS : 1); hop++) {
    capped = true;
    for (int it = 0; it < 48; it++) {
        uint32_t mx = 0;
        // (a region without a token of its own -- exit == entry -- has nothing to say: if it repeated the position it was handed, a wrong
        // one would outlive its source, the stray parse of a region inside a long literal run, by one region per iteration)
        for (uint32_t k = 0; k < PER; k++) { const uint32_t r = (uint32_t)t * PER + k; if (r < nreg && s_exit[r] != RG_INVALID && s_exit[r] != s_entry[r]) mx = max(mx, s_exit[r]); }
        s_pm[t] = mx;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const uint32_t y = t >= d ? s_pm[t - d] : 0u;
            __syncthreads();
            s_pm[t] = max(s_pm[t], y);
            __syncthreads();
        }
        uint32_t run = t ? s_pm[t - 1] : 0u;                            // the furthest exit in front of my first region
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t r = (uint32_t)t * PER + k;
            if (r >= nreg) break;
            const uint32_t a = run, rb = r * rs;                        // (= reg[r].b)
            bool work = r != 0u && a >= rb;                             // (a < b: no predecessor reaches me yet -- unsettled exits in front of me)
            if (work && need(r) && s_entry[r] == a) work = false;     // a full parse from this entry is already pending
            if (work && s_entry[r] == a && s_exit[r] != RG_INVALID) work = false;      // belief unchanged
            if (work) {
                const uint64_t bnext = (r + 1 < nreg) ? (uint64_t)(r + 1) * rs : n_src;
                s_entry[r] = a;
                s_changed = 1;
                if ((uint64_t)a >= bnext) { s_exit[r] = a; reg[r].outlen = 0; set_need(r, false); }     // no token of the chain starts in this region
                else {
                    const uint2 *tr = traces + (size_t)r * RG_TRACE;
                    const uint32_t nt = reg[r].ntrace, exit0 = reg[r].exit0, outlen0 = reg[r].outlen0, rec0 = reg[r].entry0;
                    uint64_t p = a, cum = 0;
                    uint32_t ti = 0;
                    bool settled = false;
                    for (int iter = 0; iter < RG_WALKCAP && exit0 != RG_INVALID; iter++) {       // (a longer walk is cheaper as a wave-parallel re-parse)
                        if (p >= bnext) { s_exit[r] = (uint32_t)p; reg[r].outlen = (uint32_t)cum; set_need(r, false); settled = true; break; }
                        // merged with the recorded parse?  (one of its first tokens, or the first token of a bucket: once the parses have
                        // merged, this walk visits every token of the recorded one, so it meets a recorded position within a bucket)
                        while (ti < nt && tr[ti].x < (uint32_t)p) ti++;
                        uint32_t cum0 = RG_INVALID;
                        if (ti < nt && tr[ti].x == (uint32_t)p) cum0 = tr[ti].y;
                        else if (p >= rb && (uint32_t)p >= rec0) { const uint32_t bk = ((uint32_t)p - rb) >> bsh; if (bk < RG_BUCKETS && tr[RG_DENSE + bk].x == (uint32_t)p) cum0 = tr[RG_DENSE + bk].y; }
                        if (cum0 != RG_INVALID) { s_exit[r] = exit0; reg[r].outlen = (uint32_t)(cum + (outlen0 - cum0)); set_need(r, false); reg[r].pad0 = (uint32_t)p; settled = true; break; }
                        if (!rg_step_serial<CODEC>(src, n_src, p, cum)) break;     // one token, serially
                    }
                    if (!settled) { s_exit[r] = RG_INVALID; set_need(r, true); s_pend = 1; }     // k_rg_parse takes it from `entry`
                }
            }
            if (s_exit[r] != RG_INVALID && s_exit[r] != s_entry[r]) run = max(run, s_exit[r]);
        }
        __syncthreads();
        const bool go = s_changed != 0u;                                // (regions waiting for a re-parse sit out; the others go on)
        __syncthreads();
        if (t == 0) s_changed = 0;
        __syncthreads();
        if (!go) { capped = false; break; }
    }
    if constexpr (FINISH) {
        // the regions that wait for a parse: listed, handed to the parser wavefronts through global memory, results read back
        for (uint32_t k = 0; k < PER; k++) {
            const uint32_t r = (uint32_t)t * PER + k;
            if (r < nreg && need(r)) { const uint32_t i = atomicAdd(&s_nlist, 1u); if (i < RG_FLIST) { s_list[i] = r; reg[r].entry = s_entry[r]; reg[r].needfull = 1; } }
        }
        __threadfence_block();
        __syncthreads();
        const uint32_t nl = s_nlist < RG_FLIST ? s_nlist : RG_FLIST;
        if (nl == 0u && !capped) break;                                 // nothing waits, nothing moves: done
        const uint32_t w = (uint32_t)t >> 6;
        if (w < RG_FPARSERS)
            for (uint32_t i = w; i < nl; i += RG_FPARSERS) { rg_parse_region<CODEC>(src, n_src, plan, reg, traces, s_list[i], 0, s_pwin[w], s_ptq[w], t & 63); wave_sync(); }
        __threadfence_block();
        __syncthreads();
        if ((uint32_t)t < nl) {
            const uint32_t r = s_list[t];
            s_exit[r] = __builtin_nontemporal_load(&reg[r].exit); set_need(r, false);           // (the parser wrote reg[r].outlen itself)
        }
        if (t == 0) { s_nlist = 0; s_pend = 0; s_changed = 0; }
        __syncthreads();
    }
  }
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        if (r < nreg) { reg[r].entry = s_entry[r]; reg[r].exit = s_exit[r]; reg[r].needfull = need(r) ? 1u : 0u; }
    }
    if (t == 0) { plan->pad[1] = s_pend; if (!s_pend && !capped) plan->pad[2] = 1; }   // parses pending: the next k_rg_parse has work; else: settled
}
template <bool FINISH>
__global__ __launch_bounds__(1024) void k_rg_settle(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces) { k_rg_settle_body<FINISH, RG_MAXREG>(src, n_src, plan, reg, traces, blockIdx.x, gridDim.x); }
template <bool FINISH>
__global__ __launch_bounds__(1024) void k_snr_settle(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, uint2 *traces) { k_rg_settle_body<FINISH, RG_MAXREG, RG_SNAPPY>(src, n_src, plan, reg, traces, blockIdx.x, gridDim.x); }
template <bool FINISH>
__global__ __launch_bounds__(1024) void k_rg_settle_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_settle_body<FINISH, RGB_MAXR>(j.src, j.n_src, j.plan, j.reg, j.traces, blockIdx.x, gridDim.x); }

// the first round of (1b) has nearly every region re-walk its head: one lane per region over the whole chip instead of one workgroup
__device__ __forceinline__ void k_rg_pmax_body(RgPlan *plan, const RgRegion *reg, uint32_t *pmax, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    __shared__ uint32_t s[1024];
    const int t = threadIdx.x;
    const uint32_t nreg = plan->nreg;
    constexpr uint32_t PER = RG_MAXREG / 1024;
    uint32_t mine[PER], mx = 0;
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        uint32_t e = 0;
        if (r < nreg) { e = reg[r].exit; if (e == RG_INVALID || e == reg[r].entry) e = 0; }
        mine[k] = e;
        mx = max(mx, e);
    }
    s[t] = mx;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t y = t >= d ? s[t - d] : 0u;
        __syncthreads();
        s[t] = max(s[t], y);
        __syncthreads();
    }
    uint32_t run = t ? s[t - 1] : 0u;
    for (uint32_t k = 0; k < PER; k++) {
        const uint32_t r = (uint32_t)t * PER + k;
        if (r < nreg) pmax[r] = run;
        run = max(run, mine[k]);
    }
}
__global__ __launch_bounds__(1024) void k_rg_pmax(RgPlan *plan, const RgRegion *reg, uint32_t *pmax) { k_rg_pmax_body(plan, reg, pmax, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(1024) void k_rg_pmax_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_pmax_body(j.plan, j.reg, j.pmax, blockIdx.x, gridDim.x); }
template <int CODEC = RG_LZ4>
__device__ __forceinline__ void k_rg_fix_body(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, const uint2 *traces, const uint32_t *__restrict__ pmax, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    const uint32_t nreg = plan->nreg, bsh = plan->pad[0];
    const uint32_t r = bx_ * blockDim.x + threadIdx.x;
    if (r == 0 || r >= nreg) return;
    RgRegion *R = reg + r;
    const uint32_t a = pmax[r], rb = R->b;
    if (a < rb) return;
    if (R->entry == a && R->exit != RG_INVALID) return;
    const uint64_t bnext = (r + 1 < nreg) ? (uint64_t)reg[r + 1].b : n_src;
    R->entry = a;
    if ((uint64_t)a >= bnext) { R->exit = a; R->outlen = 0; R->needfull = 0; return; }
    const uint2 *tr = traces + (size_t)r * RG_TRACE;
    const uint32_t nt = R->ntrace, exit0 = R->exit0, outlen0 = R->outlen0, rec0 = R->entry0;
    uint64_t p = a, cum = 0;
    uint32_t ti = 0;
    for (int iter = 0; iter < RG_WALKCAP && exit0 != RG_INVALID; iter++) {       // (a longer walk is cheaper as a wave-parallel re-parse)
        if (p >= bnext) { R->exit = (uint32_t)p; R->outlen = (uint32_t)cum; R->needfull = 0; return; }
        while (ti < nt && tr[ti].x < (uint32_t)p) ti++;
        uint32_t cum0 = RG_INVALID;
        if (ti < nt && tr[ti].x == (uint32_t)p) cum0 = tr[ti].y;
        else if (p >= rb && (uint32_t)p >= rec0) { const uint32_t bk = ((uint32_t)p - rb) >> bsh; if (bk < RG_BUCKETS && tr[RG_DENSE + bk].x == (uint32_t)p) cum0 = tr[RG_DENSE + bk].y; }
        if (cum0 != RG_INVALID) { R->exit = exit0; R->outlen = (uint32_t)(cum + (outlen0 - cum0)); R->needfull = 0; R->pad0 = (uint32_t)p; return; }
        if (!rg_step_serial<CODEC>(src, n_src, p, cum)) break;
    }
    R->exit = RG_INVALID; R->needfull = 1; plan->pad[1] = 1;            // k_rg_parse takes it from `entry`
}
__global__ __launch_bounds__(64) void k_rg_fix(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, const uint2 *traces, const uint32_t *__restrict__ pmax) { k_rg_fix_body(src, n_src, plan, reg, traces, pmax, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_snr_fix(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, RgRegion *reg, const uint2 *traces, const uint32_t *__restrict__ pmax) { k_rg_fix_body<RG_SNAPPY>(src, n_src, plan, reg, traces, pmax, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_rg_fix_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_fix_body(j.src, j.n_src, j.plan, j.reg, j.traces, j.pmax, blockIdx.x, gridDim.x); }

// ---- (1c) verify the chain, give every region its output position ----
__device__ __forceinline__ void k_rg_scan_body(RgPlan *plan, RgRegion *reg, uint64_t n_src, uint64_t cap, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    __shared__ uint64_t s[1024];
    __shared__ uint32_t bad;
    const int t = threadIdx.x;
    const uint32_t nreg = plan->nreg;
    if (t == 0) bad = 0;
    __syncthreads();
    uint64_t mine[RG_MAXREG / 1024];
    uint64_t sum = 0;
    for (uint32_t k = 0; k < RG_MAXREG / 1024; k++) {
        const uint32_t r = (uint32_t)t * (RG_MAXREG / 1024) + k;
        mine[k] = 0;
        if (r < nreg) {
            const RgRegion R = reg[r];
            bool ok = !R.needfull && R.exit != RG_INVALID;
            ok = ok && (r == 0 ? R.entry == plan->pad[3] : R.entry == reg[r - 1].exit);
            if (r + 1 == nreg) ok = ok && R.exit == (uint32_t)n_src;
            if (!ok) atomicOr(&bad, 1u);
            mine[k] = R.outlen;
        }
        sum += mine[k];
    }
    s[t] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint64_t y = t >= d ? s[t - d] : 0ull;
        __syncthreads();
        s[t] += y;
        __syncthreads();
    }
    uint64_t o = s[t] - sum;
    for (uint32_t k = 0; k < RG_MAXREG / 1024; k++) {
        const uint32_t r = (uint32_t)t * (RG_MAXREG / 1024) + k;
        if (r < nreg) reg[r].opos = o;
        o += mine[k];
    }
    __syncthreads();
    if (t == 0) {
        const uint64_t total = s[1023];
        plan->total = total;
        if (bad || total > cap || total > 0xFFFFFFF0ull) plan->fail = 1; else plan->ok = 1;
    }
}
__global__ __launch_bounds__(1024) void k_rg_scan(RgPlan *plan, RgRegion *reg, uint64_t n_src, uint64_t cap) { k_rg_scan_body(plan, reg, n_src, cap, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(1024) void k_rg_scan_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_scan_body(j.plan, j.reg, j.n_src, j.cap, blockIdx.x, gridDim.x); }

// ---- (2) with the chain known, write the restart index the encoder would have appended (HBIX, hb_format.h): the decoder's state at
// every HB_CHUNK bytes of output.  One wavefront per region walks its tokens (window-parallel parser again) with the region's
// output position in hand; a unit boundary that falls on a token start or inside a literal run becomes an entry; one that falls
// inside a match (or at its start) means the block was not written chunk-locally: no index, the single wavefront decodes. ----
__device__ __forceinline__ void rg_emit(uint8_t *ents, uint64_t U, uint32_t s_off, uint32_t lit_rem, uint32_t tok_off) {
    u32x4 v; v.x = s_off; v.y = (uint32_t)U; v.z = lit_rem; v.w = tok_off;
    *(u32x4 *)(ents + (U / HB_CHUNK) * HB_IDX_ENTRY) = v;                  // (entries are HB_IDX_ENTRY = 16 bytes, the table is 16-byte aligned)
}
// Round 3: the walk below costs a wavefront about a microsecond per 64 stream bytes (~0.5 ms for a 34 KiB region, 1.1 ms per GiB for
// the launch) only to find the unit boundaries inside the region.  The first parse left a record of where it was at the start of each
// of 128 buckets of the region's stream range ({position, output so far}: `traces`), and from reg[r].pad0 on that record lies on the
// final chain.  k_rg_index_fast: one wavefront per region, one LANE per recorded token: lane k walks the tokens from its record to the
// next one (a bucket: <= 512 stream bytes of token starts) and writes the entry of every unit boundary it comes
// by -- the same three cases as the wave walk (at a token, inside a literal run, inside a match = not chunk-local).  What the records do
// not cover -- the region's head up to its first usable record (the stretch the first parse needed to fall onto the chain), regions
// whose record is not on the chain at all -- is left to k_rg_index: done[r] = where its walk may stop (0: walk everything).
// Same entries either way (tests/test_gpu_codec.py, test_gpu_foreign.py at 16 MiB - 1 GiB; A/B: HIPBLOSC_DEBUG_SLOW_INDEX=1).
// (Measured on the way, 1 GiB D-f32: with the buckets staged in 16 KiB of LDS per wave the launch took 1.02 ms -- 9 waves per CU, each
// lane's byte-serial walk ~140 us per region: no better than the wave walk at 32 waves per CU.  The lanes read the stream straight
// from memory instead, a dword per access -- token + first extension bytes, offset + first extension bytes -- at full occupancy.)
__device__ __forceinline__ uint32_t rg_rd4(const uint8_t *__restrict__ src, const uint64_t n_src, const uint64_t at) {
    if (at + 4u <= n_src) return ld4u(src + at);
    uint32_t w = 0;
    for (uint32_t b = 0; b < 4u; b++) if (at + b < n_src) w |= (uint32_t)src[at + b] << (8u * b);
    return w;
}
// a length extension (bytes 255 ... 255 r) at stream position q: adds it to len, moves q behind it; false: runs off the stream / absurd
__device__ __forceinline__ bool rg_ext(const uint8_t *__restrict__ src, const uint64_t n_src, uint64_t &q, uint64_t &len) {
    // (64 steps = a run of 64 KiB: every step is a memory round trip of its own, and a lane that crawls through the 1 MiB extension in front
    // of an incompressible byte plane keeps its whole launch waiting -- 4096 steps were 0.5 - 1.2 ms per GiB; longer runs go to the wave walk)
    for (int k = 0; k < 64; k++) {
        if (q >= n_src) return false;
        const uint32_t w = rg_rd4(src, n_src, q);
        if (w == 0xFFFFFFFFu && q + 4u <= n_src) { len += 1020u; q += 4u; continue; }
        const uint32_t nff = (uint32_t)__builtin_ctz(~w) >> 3;
        if (q + nff >= n_src) return false;
        len += 255u * nff + ((w >> (8u * nff)) & 255u);
        q += nff + 1u;
        return true;
    }
    return false;
}
// The same entries from the token store (hb_lz4_region.h), when the workspace has one: no walk at all.  One wavefront per region reads the stored
// tokens 64 at a time; their output lengths add up backwards from the region's end to the output position of the first token on the chain (the
// one at RgRegion.pad0), forwards to every token's own, and each lane writes the entries of the unit boundaries its sequence holds.  done[r] as
// k_rg_index_fast leaves it (which then only takes the regions this kernel left at 0: no usable store, a token with 32-bit lengths).
__device__ __forceinline__ void k_rg_index_tok_body(const uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint2 *__restrict__ tok, const uint32_t tokcap, uint8_t *__restrict__ index, uint32_t *__restrict__ done, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    const int lane = threadIdx.x;
    const uint32_t r = bx_;
    if (lane == 0) done[r] = 0u;
    if (!plan->ok || __hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || r >= plan->nreg) return;
    const uint64_t N = plan->total;
    uint8_t *ents = index + HB_IDX_HDR_BYTES;
    const uint32_t start = RFL(reg[r].entry), exitp = RFL(reg[r].exit), outlen = RFL(reg[r].outlen), pad0 = RFL(reg[r].pad0), nt = RFL(reg[r].pad1[0]);
    if (outlen == 0u || start >= exitp) { if (lane == 0) done[r] = 0xFFFFFFFFu; return; }
    if (pad0 == RG_INVALID || RFL(reg[r].exit0) != exitp || nt == RG_INVALID || nt == 0u || pad0 >= exitp) return;
    const uint64_t opos = reg[r].opos;
    if (opos / HB_CHUNK == (opos + outlen - 1) / HB_CHUNK && (opos & (HB_CHUNK - 1)) != 0) { if (lane == 0) done[r] = 0xFFFFFFFFu; return; }   // no boundary in here
    const uint2 *tk = tok + (size_t)r * tokcap;
    // the first stored token on the chain, and what the tokens from there on put out (a token with 32-bit lengths: not here).  Every lane
    // adds up its own tokens, the sums meet once at the end: no step waits for the one before.
    uint32_t k0 = RG_INVALID;
    uint64_t mysum = 0;
    bool odd = false;
    for (uint32_t b = 0; b < nt; b += 64u) {
        const uint32_t k = b + (uint32_t)lane;
        uint2 t; t.x = 0; t.y = 0;
        if (k < nt) t = tk[k];
        const bool on = k < nt && t.x >= pad0;
        if (k0 == RG_INVALID) { const unsigned long long m = hb_ballot(on); if (m) k0 = b + (uint32_t)__builtin_ctzll(m); }
        odd = odd || (on && t.y == RG_INVALID);
        mysum += on ? (t.y & 0xFFFFu) + (t.y >> 16) : 0u;
        if ((b & 0x3FFu) == 0x3C0u && __hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // (every 16th step) no index
    }
    if (hb_ballot(odd) || k0 == RG_INVALID) return;
    uint64_t tail = mysum;
    for (int d = 32; d; d >>= 1) tail += (uint64_t)__shfl_xor((unsigned long long)tail, d);
    if (tail > outlen) return;
    if (RFL(tk[k0].x) != pad0) return;                                  // (pad0 is a token of the recorded parse: it is there)
    if (__hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;         // somebody found a boundary inside a match: no index
    uint64_t d = (uint64_t)outlen - tail;                               // output position (from `entry`) of the token at pad0
    bool inmatch = false;
    // four tokens per lane and step (32 bytes, two 16-byte reads): a quarter of the dependent steps (the running position is the only thing a
    // step needs from the one before)
    for (uint32_t b = k0; b < nt && !inmatch; b += 256u) {
        const uint32_t kb = b + 4u * (uint32_t)lane;
        uint2 t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { t[q].x = 0; t[q].y = 0; if (kb + (uint32_t)q < nt) t[q] = tk[kb + (uint32_t)q]; }
        uint32_t o[4], tot = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { o[q] = kb + (uint32_t)q < nt ? (t[q].y & 0xFFFFu) + (t[q].y >> 16) : 0u; tot += o[q]; }
        const uint32_t incl = wave_incl_scan_dpp(tot);
        uint64_t g0 = opos + d + (incl - tot);                          // absolute output position of my first sequence
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (kb + (uint32_t)q < nt) {
                const uint32_t ll = t[q].y & 0xFFFFu, ml = t[q].y >> 16, tp = t[q].x;
                const uint32_t ls = tp + 1u + (ll >= 15u ? (((ll - 15u) * 0x8081u) >> 23) + 1u : 0u);     // (x / 255 for x < 65536)
                uint64_t U = (g0 + HB_CHUNK - 1) & ~(uint64_t)(HB_CHUNK - 1);
                if (U == g0 && U < N) { rg_emit(ents, U, tp, HB_IDX_AT_TOKEN, 0u); U += HB_CHUNK; }
                for (; U < g0 + ll && U < N; U += HB_CHUNK) rg_emit(ents, U, (uint32_t)(ls + (U - g0)), (uint32_t)(g0 + ll - U), tp);
                if (U < g0 + ll + ml && U < N) inmatch = true;          // a unit boundary inside a match: the block was not written chunk-locally
                g0 += o[q];
            }
        }
        inmatch = hb_ballot(inmatch) != 0ull;
        d += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
    }
    if (inmatch) { if (lane == 0) { done[r] = 6u; if (!__hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicExch(&plan->fail, 1u); } return; }
    if (lane == 0) done[r] = pad0 > start ? pad0 : 0xFFFFFFFFu;        // the head [entry, pad0) is still to do (nothing, when the chain starts on the record)
}
__global__ __launch_bounds__(64) void k_rg_index_tok(const uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint2 *__restrict__ tok, const uint32_t tokcap, uint8_t *__restrict__ index, uint32_t *__restrict__ done) { k_rg_index_tok_body(n_src, plan, reg, tok, tokcap, index, done, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_rg_index_tok_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_index_tok_body(j.n_src, j.plan, j.reg, j.tok, j.tokcap, j.idx, j.pmax, blockIdx.x, gridDim.x); }

__device__ __forceinline__ void k_rg_index_fast_body(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint2 *__restrict__ traces, uint8_t *__restrict__ index, uint32_t *__restrict__ done, int after_tok, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    __shared__ uint2 s_tr[RG_BUCKETS + 1];
    const int lane = threadIdx.x;
    const uint32_t r = bx_;
    if (after_tok) { if (r < plan->nreg && RFL(done[r]) != 0u) return; }                // k_rg_index_tok did this region (or gave the verdict)
    else if (lane == 0) done[r] = 0u;
    if (!plan->ok || __hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) || r >= plan->nreg) return;
    const uint64_t N = plan->total;
    uint8_t *ents = index + HB_IDX_HDR_BYTES;
    const uint32_t start = RFL(reg[r].entry), exitp = RFL(reg[r].exit), outlen = RFL(reg[r].outlen), pad0 = RFL(reg[r].pad0);
    if (outlen == 0u || start >= exitp) { if (lane == 0) done[r] = 0xFFFFFFFFu; return; }
    const uint32_t adj = outlen - RFL(reg[r].outlen0);                  // output position (from `entry`) of a recorded token = its recorded value + adj
    const uint64_t opos = reg[r].opos;
    // the record is usable where it lies on the final chain: from pad0 on, and only when the recorded parse ends where the chain's does
    // (the rule of hb_lz4_sym.hip k_sy_units)
    if (pad0 == RG_INVALID || RFL(reg[r].exit0) != exitp) return;
    if (opos / HB_CHUNK == (opos + outlen - 1) / HB_CHUNK && (opos & (HB_CHUNK - 1)) != 0) { if (lane == 0) done[r] = 0xFFFFFFFFu; return; }   // no boundary in here
    // the usable records, compacted in stream order: { position, output position from `entry` }
    uint32_t nu = 0;
    {
        const uint2 *tr = traces + (size_t)r * RG_TRACE + RG_DENSE;
        for (uint32_t b0 = 0; b0 < RG_BUCKETS; b0 += 64) {
            uint2 t = tr[b0 + lane];
            const bool ok = t.x != RG_INVALID && t.x >= pad0 && t.x > start && t.x < exitp;
            const unsigned long long m = hb_ballot(ok);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (ok) { t.y += adj; s_tr[nu + rank] = t; }
            nu += (uint32_t)__builtin_popcountll(m);
        }
    }
    if (nu == 0u) return;
    if (lane == 0) { uint2 e; e.x = exitp; e.y = outlen; s_tr[nu] = e; }
    wave_sync();
    const uint32_t headend = s_tr[0].x;
    bool bad = false, inmatch = false;
    // (Measured and not kept, round 3: a 16-byte register window per lane so that 3-4 byte sequences share a memory round trip -- no gain, the
    // walk is bound by its dependent steps, not by the loads' count; the whole wavefront emitting the entries of a long literal run -- the
    // lanes' extra state cost the launch 0.14 ms, and runs of more than 64 KiB never get here.)
    for (uint32_t k = (uint32_t)lane; k < nu; k += 64u) {
        const uint32_t send = s_tr[k + 1].x;                            // my segment: the tokens that START in [s_tr[k].x, send)
        uint64_t q = s_tr[k].x;                                         // stream position
        uint64_t d0 = s_tr[k].y;                                        // output position (from `entry`) of the token at q
        for (uint32_t steps = 0; q < send; steps++) {
            if (steps > 1024u) { bad = true; break; }                   // (a bucket holds <= 512 stream bytes of token starts: cannot happen on the chain)
            // somebody found a unit boundary inside a match: there will be no index, stop walking (another writer's frame says so within
            // microseconds, and every lane of the launch would otherwise finish its bucket first)
            // (the poll is ISSUED in front of the token's read and LOOKED AT behind it: both are in flight together and return in order, so the
            // step waits once; a poll that is branched on at once adds a memory round trip to every polling step: +20 % for a frame that has an index)
            uint32_t failseen = 0;
            if ((steps & 3u) == 3u) failseen = __hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t tp = (uint32_t)q;
            const uint32_t w = rg_rd4(src, n_src, q);
            const uint32_t tok = w & 255u;
            if (failseen) { bad = true; break; }
            uint64_t ll = tok >> 4;
            q++;
            if (ll == 15u) {
                const uint32_t b1 = (w >> 8) & 255u;
                if (b1 != 255u && q < n_src) { ll += b1; q++; }
                else if (!rg_ext(src, n_src, q, ll)) { bad = true; break; }
            }
            const uint64_t ls = q;
            if (ll > n_src - ls) { bad = true; break; }
            q += ll;
            uint64_t ml = 0;
            if (q != n_src) {
                if (n_src - q < 2) { bad = true; break; }
                ml = (tok & 15u) + 4u;
                if ((tok & 15u) == 15u) {
                    const uint32_t x = rg_rd4(src, n_src, q);
                    const uint32_t b2 = (x >> 16) & 255u;
                    q += 2;
                    if (b2 != 255u && q < n_src) { ml += b2; q++; }
                    else if (!rg_ext(src, n_src, q, ml)) { bad = true; break; }
                } else q += 2;
            }
            // unit boundaries of this sequence: at its token, inside its literal run (any number of them), never in its match
            const uint64_t g0 = opos + d0;                              // absolute output position of the sequence
            uint64_t U = (g0 + HB_CHUNK - 1) & ~(uint64_t)(HB_CHUNK - 1);
            if (U == g0 && U < N) { rg_emit(ents, U, tp, HB_IDX_AT_TOKEN, 0u); U += HB_CHUNK; }
            for (; U < g0 + ll && U < N; U += HB_CHUNK) rg_emit(ents, U, (uint32_t)(ls + (U - g0)), (uint32_t)(g0 + ll - U), tp);
            if (U < g0 + ll + ml && U < N) {                            // a unit boundary inside a match: the block was not written chunk-locally
                // (said at once: the other lanes of this wavefront, and every other wavefront of the launch, are still walking and poll the flag)
                // (one writer is enough: 100 000 lanes of a foreign frame get here, and as many atomics on one word take milliseconds)
                bad = true; inmatch = true;
                if (!__hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicExch(&plan->fail, 1u);
                break;
            }
            d0 += ll + ml;
        }
        if (!bad && (q != send || (uint32_t)d0 != s_tr[k + 1].y)) bad = true;           // my walk must land on the next record exactly
        if (bad) break;
    }
    // A boundary inside a match, seen from a record of the settled chain (every usable record is a token of it, with its final output
    // position): the frame has no index, whatever the other regions find -- a frame of another writer says so in its first regions,
    // and the rest of this launch and the wave walk behind it have nothing left to do (1 GiB as the reference writes it: 1.4 ms).
    if (hb_ballot(inmatch)) { if (lane == 0) { done[r] = 6u; atomicExch(&plan->fail, 1u); } return; }
    // anything odd: the wave walk does the whole region again, with the machinery the other decoders share (it also decides what a
    // boundary inside a match means for the frame)
    if (hb_ballot(bad)) { if (lane == 0) done[r] = 6u; return; }
    if (lane == 0) done[r] = headend;                                   // the head [entry, first usable record) is still to do
}
__global__ __launch_bounds__(64) void k_rg_index_fast(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint2 *__restrict__ traces, uint8_t *__restrict__ index, uint32_t *__restrict__ done, int after_tok) { k_rg_index_fast_body(src, n_src, plan, reg, traces, index, done, after_tok, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_rg_index_fast_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_index_fast_body(j.src, j.n_src, j.plan, j.reg, j.traces, j.idx, j.pmax, 1, blockIdx.x, gridDim.x); }

__device__ __forceinline__ void k_rg_index_body(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint32_t *__restrict__ done, uint8_t *__restrict__ index, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    __shared__ __attribute__((aligned(16))) uint8_t s_win[RG_PWIN + 128];
    __shared__ __attribute__((aligned(16))) uint2 s_tq[DTQ];
    if (!plan->ok || plan->fail) return;
    const int lane = threadIdx.x;
    const uint32_t nreg = plan->nreg;
    const uint64_t N = plan->total;
    uint8_t *ents = index + HB_IDX_HDR_BYTES;
    for (uint32_t r = bx_; r < nreg; r += gx_) {
        if (__hip_atomic_load(&plan->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;     // somebody found a boundary inside a match
        const uint32_t start = RFL(reg[r].entry), exitp = RFL(reg[r].exit);
        if (RFL(reg[r].outlen) == 0u || start >= exitp) continue;
        // k_rg_index_fast: 0 = nothing done, 0xFFFFFFFF = all of this region's entries written, else only the head [entry, that position) is left
        const uint32_t dn = done ? RFL(done[r]) : 0u;
        if (dn == 0xFFFFFFFFu) continue;
        const uint32_t walk_end = dn >= 16u ? dn : exitp;
        uint64_t out = reg[r].opos;
        // unit boundaries of a sequence: at its token, inside its literal run (any number of them), never in its match
        auto batch = [&](uint32_t cnt, uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t lp) __attribute__((always_inline)) -> bool {
            (void)off; (void)lp;
            const uint32_t olen = (uint32_t)lane < cnt ? lit + mlen : 0u;
            const uint32_t incl = wave_incl_scan_dpp(olen);
            const uint64_t d0 = out + incl - olen;                       // where my sequence's output starts
            uint64_t U = (d0 + HB_CHUNK - 1) & ~(uint64_t)(HB_CHUNK - 1);                // first unit boundary at / after it
            bool inm = false;
            if ((uint32_t)lane < cnt) {
                if (U == d0 && U < N) { rg_emit(ents, U, tp, HB_IDX_AT_TOKEN, 0u); U += HB_CHUNK; }
                for (; U < d0 + lit && U < N; U += HB_CHUNK) rg_emit(ents, U, ls + (uint32_t)(U - d0), (uint32_t)(d0 + lit - U), tp);
                if (U < d0 + lit + mlen && U < N) inm = true;           // a unit boundary at the start of / inside a match
            }
            if (hb_ballot(inm)) return false;
            out += (uint32_t)__builtin_amdgcn_readlane(incl, 63);
            return true;
        };
        auto single = [&](uint32_t tp, uint32_t ls, uint32_t lit, uint32_t mlen, uint32_t off, uint32_t tok) __attribute__((always_inline)) -> bool {
            (void)off; (void)tok;
            const uint64_t d0 = out, U0 = (d0 + HB_CHUNK - 1) & ~(uint64_t)(HB_CHUNK - 1);
            if (U0 == d0 && U0 < N && lane == 0) rg_emit(ents, U0, tp, HB_IDX_AT_TOKEN, 0u);
            for (uint64_t U = (U0 == d0 ? U0 + HB_CHUNK : U0) + (uint64_t)lane * HB_CHUNK; U < d0 + lit && U < N; U += 64ull * HB_CHUNK)
                rg_emit(ents, U, ls + (uint32_t)(U - d0), (uint32_t)(d0 + lit - U), tp);
            const uint64_t m0 = d0 + lit, Um = (m0 + HB_CHUNK - 1) & ~(uint64_t)(HB_CHUNK - 1);
            if (mlen && Um < m0 + mlen && Um < N) return false;         // a boundary at the start of / inside the match
            out += (uint64_t)lit + mlen;
            return true;
        };
#ifdef RG_DEBUG_TIMES
        const uint64_t t0 = wall_clock64();
#endif
        if (!rg_walk(src, n_src, start, walk_end, s_win, s_tq, lane, batch, single) && lane == 0) atomicExch(&plan->fail, 1u);
        wave_sync();
#ifdef RG_DEBUG_TIMES
        if (lane == 0 && done) ((uint32_t *)done)[r] = 0x80000000u | (uint32_t)((wall_clock64() - t0) / 100);      // microseconds (100 MHz clock)
#endif
    }
}
__global__ __launch_bounds__(64) void k_rg_index(const uint8_t *__restrict__ src, uint64_t n_src, RgPlan *plan, const RgRegion *__restrict__ reg, const uint32_t *__restrict__ done, uint8_t *__restrict__ index) { k_rg_index_body(src, n_src, plan, reg, done, index, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(64) void k_rg_index_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_index_body(j.src, j.n_src, j.plan, j.reg, j.pmax, j.idx, blockIdx.x, gridDim.x); }

// first entry, terminator and header (written last: a failed build leaves the zeroed header, which k_dec_plan rejects)
__device__ __forceinline__ void k_rg_index_head_body(RgPlan *plan, uint8_t *__restrict__ index, uint64_t n_src, const uint32_t bx_, const uint32_t gx_) {
    (void)bx_; (void)gx_;
    if (!plan->ok || plan->fail) return;
    const uint64_t N = plan->total;
    const uint32_t nunits = (uint32_t)((N + HB_CHUNK - 1) / HB_CHUNK);
    if (nunits == 0) return;
    uint32_t *e = (uint32_t *)(index + HB_IDX_HDR_BYTES);
    e[0] = 0; e[1] = 0; e[2] = HB_IDX_AT_TOKEN; e[3] = 0;
    uint32_t *t = e + 4 * (size_t)nunits;
    t[0] = (uint32_t)n_src; t[1] = (uint32_t)N; t[2] = 0; t[3] = 0;
    uint32_t *h = (uint32_t *)index;
    h[0] = HB_IDX_MAGIC; h[1] = HB_IDX_VERSION | (HB_IDX_ENTRY << 16); h[2] = nunits; h[3] = HB_CHUNK;
    h[4] = (uint32_t)n_src; h[5] = (uint32_t)N; h[6] = 0;
    h[7] = h[0] ^ h[1] ^ h[2] ^ h[3] ^ h[4] ^ h[5];
}
__global__ void k_rg_index_head(RgPlan *plan, uint8_t *__restrict__ index, uint64_t n_src) { k_rg_index_head_body(plan, index, n_src, blockIdx.x, gridDim.x); }
__global__ void k_rg_index_head_b(const RgJob *__restrict__ jobs) { const RgJob j = jobs[blockIdx.y]; k_rg_index_head_body(j.plan, j.idx, j.n_src, blockIdx.x, gridDim.x); }

// Builds the restart index of an index-less block in the workspace; *index / *index_bytes then go to k_dec_plan / k_dec_indexed as if
// the frame had carried them (an index that could not be built stays zeroed and is rejected there: serial decode).
// workgroups of the FIRST re-parse behind k_rg_fix: almost every region moved its entry off the guess then (14 500 of 16 378 on the headline
// frames) and re-parses a few windows until it meets its own record; 1024 workgroups did 14 regions each, one after the other (0.34 -> 0.28 ms).
// (Also measured in round 3 and not kept: a table of announced literal runs of >= 1 MiB, so that the first parse of the regions INSIDE such a run --
// the incompressible plane of shuffled floats is a quarter to a half of all regions, and parsing its bytes as tokens costs what parsing tokens costs --
// does not start or stops under way: the run's token is read late in ITS region's parse, the regions it covers are parsing at the same time, and
// only the second generation of workgroups profits: k_rg_parse 1.88 -> 1.74 ms on the reference-written frame, nothing on this library's own.)
#ifndef RG_FIXGRID
#define RG_FIXGRID 16384u
#endif
int hb_launch_lz4_region_index(const hb_dec_args &a, const uint8_t **index, size_t *index_bytes, hipStream_t s) {
    const RgLayout L = rg_layout(a.cap);
    uint8_t *w = a.work + 256;                                          // behind the DecPlan
    RgPlan *plan = (RgPlan *)(w + L.plan);
    RgRegion *reg = (RgRegion *)(w + L.reg);
    uint2 *traces = (uint2 *)(w + L.trace);
    uint8_t *idx = w + L.total;                                         // hb_lz4_index_bound(cap) bytes behind the fixed part
    const size_t ib = hb_lz4_index_bound(a.cap);
    uint64_t rs; uint32_t nreg;
    rg_regions(a.n, &rs, &nreg);
    // the token store: the first rg_tok_bytes(cap) bytes of the symbolic decoder's scratch, when the caller's workspace has that (hb_lz4_sym.hip)
    static const bool no_tok = [] { const char *e = getenv("HIPBLOSC_DEBUG_NO_TOKEN_STORE"); return e && *e && *e != '0'; }();   // A/B
    uint2 *tok = (a.sym_work && !no_tok) ? (uint2 *)a.sym_work : nullptr;
    const uint32_t tokcap = rg_tokcap(rs);
    HB_HIP_TRY(hipMemsetAsync(idx, 0, ib, s));
    hb_prof_begin("k_rg_parse", s);
    hipLaunchKernelGGL(k_rg_init, dim3((nreg + 255) / 256), dim3(256), 0, s, plan, reg, nreg, (uint32_t)rs);
    hipLaunchKernelGGL(k_rg_parse, dim3(nreg), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, traces, 1, tok, tokcap);
    hb_prof_end(s);
    hb_prof_begin("k_rg_fix", s);
    hipLaunchKernelGGL(k_rg_pmax, dim3(1), dim3(1024), 0, s, plan, reg, (uint32_t *)(w + L.pmax));
    hipLaunchKernelGGL(k_rg_fix, dim3((nreg + 63) / 64), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, traces, (const uint32_t *)(w + L.pmax));
    hipLaunchKernelGGL(k_rg_parse, dim3(nreg < RG_FIXGRID ? nreg : RG_FIXGRID), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, traces, 0, tok, tokcap);
    hb_prof_end(s);
    hb_prof_begin("k_rg_settle", s);
    for (int k = 0; k < RG_FIXROUNDS; k++) {                           // (both return at once when an earlier round has settled the chain)
        hipLaunchKernelGGL(k_rg_settle<false>, dim3(1), dim3(1024), 0, s, a.src, (uint64_t)a.n, plan, reg, traces);
        hipLaunchKernelGGL(k_rg_parse, dim3(nreg < 1024u ? nreg : 1024u), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, traces, 0, tok, tokcap);
    }
    hipLaunchKernelGGL(k_rg_settle<true>, dim3(1), dim3(1024), 0, s, a.src, (uint64_t)a.n, plan, reg, traces);
    hipLaunchKernelGGL(k_rg_scan, dim3(1), dim3(1024), 0, s, plan, reg, (uint64_t)a.n, (uint64_t)a.cap);
    hb_prof_end(s);
    hb_prof_begin("k_rg_index", s);
    static const bool slow_index = [] { const char *e = getenv("HIPBLOSC_DEBUG_SLOW_INDEX"); return e && *e && *e != '0'; }();   // A/B: the wave-parallel walk only
    uint32_t *done = (uint32_t *)(w + L.pmax);                          // (k_rg_pmax / k_rg_fix are through with it)
    if (!slow_index && tok) hipLaunchKernelGGL(k_rg_index_tok, dim3(nreg), dim3(64), 0, s, (uint64_t)a.n, plan, reg, (const uint2 *)tok, tokcap, idx, done);
    if (!slow_index) hipLaunchKernelGGL(k_rg_index_fast, dim3(nreg), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, (const uint2 *)traces, idx, done, tok ? 1 : 0);
    hipLaunchKernelGGL(k_rg_index, dim3(nreg), dim3(64), 0, s, a.src, (uint64_t)a.n, plan, reg, slow_index ? (const uint32_t *)nullptr : (const uint32_t *)done, idx);
    hipLaunchKernelGGL(k_rg_index_head, dim3(1), dim3(1), 0, s, plan, idx, (uint64_t)a.n);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    *index = idx; *index_bytes = ib;
    return HB_OK;
}


// ---- the chain of a Snappy block: the same kernels with the element parser; what follows (units, decode) is hb_snappy.hip's ----
int hb_launch_snappy_region_chain(const uint8_t *src, size_t n, size_t cap, uint8_t *w, const uint32_t *entry0, hipStream_t s) {
    const RgLayout L = rg_layout(cap);
    RgPlan *plan = (RgPlan *)(w + L.plan);
    RgRegion *reg = (RgRegion *)(w + L.reg);
    uint2 *traces = (uint2 *)(w + L.trace);
    uint32_t *pmax = (uint32_t *)(w + L.pmax);
    uint64_t rs; uint32_t nreg;
    rg_regions(n, &rs, &nreg);
    hb_prof_begin("k_snr_parse", s);
    hipLaunchKernelGGL(k_rg_init_sn, dim3((nreg + 255) / 256), dim3(256), 0, s, plan, reg, nreg, (uint32_t)rs, entry0);
    hipLaunchKernelGGL(k_snr_parse, dim3(nreg), dim3(64), 0, s, src, (uint64_t)n, plan, reg, traces, 1);
    hb_prof_end(s);
    static const int stop_at = [] { const char *e = getenv("HIPBLOSC_DEBUG_SNR_STOP"); return e && *e ? atoi(e) : 1000; }();   // lab: leave the chain as it is after this many stages
    if (stop_at < 1) return HB_OK;
    hb_prof_begin("k_snr_settle", s);
    // two chip-wide belief rounds (a lane per region: ~65 us each) before the one-workgroup rounds (0.2 ms while many regions still move)
    for (int k = 0; k < 2; k++) {
        hipLaunchKernelGGL(k_rg_pmax, dim3(1), dim3(1024), 0, s, plan, reg, pmax);
        hipLaunchKernelGGL(k_snr_fix, dim3((nreg + 63) / 64), dim3(64), 0, s, src, (uint64_t)n, plan, reg, traces, (const uint32_t *)pmax);
        hipLaunchKernelGGL(k_snr_parse, dim3(nreg), dim3(64), 0, s, src, (uint64_t)n, plan, reg, traces, 0);
        if (stop_at < 2) break;
    }
    for (int k = 0; k < RG_FIXROUNDS && k + 2 <= stop_at; k++) {
        hipLaunchKernelGGL(k_snr_settle<false>, dim3(1), dim3(1024), 0, s, src, (uint64_t)n, plan, reg, traces);
        hipLaunchKernelGGL(k_snr_parse, dim3(nreg), dim3(64), 0, s, src, (uint64_t)n, plan, reg, traces, 0);     // (a workgroup per region: those that wait for a parse wait for nothing else)
    }
    if (stop_at < 100) { hb_prof_end(s); return HB_OK; }
    hipLaunchKernelGGL(k_snr_settle<true>, dim3(1), dim3(1024), 0, s, src, (uint64_t)n, plan, reg, traces);
    hipLaunchKernelGGL(k_rg_scan, dim3(1), dim3(1024), 0, s, plan, reg, (uint64_t)n, (uint64_t)cap);
    hb_prof_end(s);
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}


// ---- the index of MANY index-less blocks in one set of launches (hb_decompress_frames_batch_dev) ----
// One job per block; a job's scratch is laid out by rg_batch_layout (plan, regions, pmax / done, traces, token store), its index goes to `idx`.
// A block whose chain does not verify, or that was not written chunk-locally, ends with a zeroed index header: k_bt_dec_plan rejects it and the
// stream decoder takes the frame, as before.
bool hb_lz4_region_batch_wanted(size_t n_src, size_t cap) {
    if (!hb_indexless_parallel(n_src, cap) || n_src >= 0xFFFFFFF0ull || cap >= 0xFFFFFFF0ull || n_src > cap + cap / 255 + 16) return false;
    uint64_t rs; uint32_t nreg;
    rg_regions(n_src, &rs, &nreg);
    return nreg <= RGB_MAXR;
}
size_t hb_lz4_region_batch_bytes(size_t n_src, size_t cap) {
    if (!hb_lz4_region_batch_wanted(n_src, cap)) return 0;
    return rg_batch_layout(n_src).total + ((hb_lz4_index_bound(cap) + 255) & ~(size_t)255);
}
void hb_lz4_region_batch_job(uint8_t *w, uint8_t *idx, const uint8_t *src, size_t n_src, size_t cap, RgJob *j, uint64_t rs_min) {
    const RgBatchLayout L = rg_batch_layout(n_src, rs_min);
    uint64_t rs; uint32_t nreg;
    rg_regions_min(n_src, rs_min, &rs, &nreg);
    j->src = src; j->n_src = n_src; j->cap = cap;
    j->plan = (RgPlan *)(w + L.plan); j->reg = (RgRegion *)(w + L.reg); j->pmax = (uint32_t *)(w + L.pmax);
    j->traces = (uint2 *)(w + L.trace); j->tok = (uint2 *)(w + L.tok); j->idx = idx;
    j->tokcap = rg_tokcap(rs); j->nreg = nreg; j->rs = (uint32_t)rs; j->pad = 0;
}
int hb_launch_lz4_region_index_batch(const RgJob *d_jobs, int njobs, uint32_t max_nreg, hipStream_t s) {
    if (njobs <= 0) return HB_OK;
    for (int j0 = 0; j0 < njobs; j0 += 65535) {                       // gridDim.y <= 65535
        const unsigned ny = (unsigned)(njobs - j0 < 65535 ? njobs - j0 : 65535);
        const RgJob *jb = d_jobs + j0;
        hb_prof_begin("k_rg_parse", s);
        hipLaunchKernelGGL(k_rg_init_b, dim3((max_nreg + 255) / 256, ny), dim3(256), 0, s, jb);
        hipLaunchKernelGGL(k_rg_parse_b, dim3(max_nreg, ny), dim3(64), 0, s, jb, 1);
        hb_prof_end(s);
        hb_prof_begin("k_rg_fix", s);
        hipLaunchKernelGGL(k_rg_pmax_b, dim3(1, ny), dim3(1024), 0, s, jb);
        hipLaunchKernelGGL(k_rg_fix_b, dim3((max_nreg + 63) / 64, ny), dim3(64), 0, s, jb);
        hipLaunchKernelGGL(k_rg_parse_b, dim3(max_nreg, ny), dim3(64), 0, s, jb, 0);
        hb_prof_end(s);
        hb_prof_begin("k_rg_settle", s);
        for (int k = 0; k < RG_FIXROUNDS; k++) {
            hipLaunchKernelGGL(k_rg_settle_b<false>, dim3(1, ny), dim3(1024), 0, s, jb);
            hipLaunchKernelGGL(k_rg_parse_b, dim3(max_nreg < 64u ? max_nreg : 64u, ny), dim3(64), 0, s, jb, 0);
        }
        hipLaunchKernelGGL(k_rg_settle_b<true>, dim3(1, ny), dim3(1024), 0, s, jb);
        hipLaunchKernelGGL(k_rg_scan_b, dim3(1, ny), dim3(1024), 0, s, jb);
        hb_prof_end(s);
        hb_prof_begin("k_rg_index", s);
        hipLaunchKernelGGL(k_rg_index_tok_b, dim3(max_nreg, ny), dim3(64), 0, s, jb);
        hipLaunchKernelGGL(k_rg_index_fast_b, dim3(max_nreg, ny), dim3(64), 0, s, jb);
        hipLaunchKernelGGL(k_rg_index_b, dim3(max_nreg, ny), dim3(64), 0, s, jb);
        hipLaunchKernelGGL(k_rg_index_head_b, dim3(1, ny), dim3(1), 0, s, jb);
        hb_prof_end(s);
    }
    HB_HIP_TRY(hipGetLastError());
    return HB_OK;
}
