// hb_format.h — sizes and the restart-index layout shared by the device kernels and the host-only helpers (hb_host.cpp).
// Plain C++, no HIP: the sanitizer build of the host helpers (tests/tools/host_asan_check.cpp) includes it too.
#pragma once
#include <stddef.h>
#include <stdint.h>

// One wavefront encodes / decodes one chunk of the (filtered) buffer.
#define HB_CHUNK        4096u            // bytes of input per chunk; matches never leave their chunk
#define HB_RSTRIDE      (HB_CHUNK + 64u) // bytes reserved per chunk record in the workspace
#define HB_TILE_CHUNKS  256u             // chunks per scan tile (one workgroup of the stitch kernel)

// Restart index ("HBIX"), written AFTER cbytes of a frame (the reference decoder never looks there,
// blosc.go:385-393) or into a caller buffer for a bare block.
//   header  : 8 x u32 { magic, version|entry_size<<16, nunits, chunk_bytes, payload_bytes, nbytes, 0, check }
//   entries : (nunits + 1) x { u32 src_off, dst_off, lit_rem, tok_off }
// Entry k says: when the serial decoder has produced dst_off bytes it is at payload offset src_off, inside a
// literal run with lit_rem bytes still to copy, whose token sits at tok_off.  lit_rem == HB_IDX_AT_TOKEN
// means "src_off is a token".  The last entry is the terminator { payload_bytes, nbytes, 0, 0 }.
#define HB_IDX_MAGIC     0x58494248u     // "HBIX"
#define HB_IDX_VERSION   1u
#define HB_IDX_HDR_BYTES 32u
#define HB_IDX_ENTRY     16u
#define HB_IDX_AT_TOKEN  0xFFFFFFFFu


// Unit index of a Snappy payload ("HBSX"), same place as HBIX (after cbytes).  A Snappy block is a plain concatenation of
// elements and this encoder never lets an element straddle two 4 KiB chunks of output, so a unit needs one number only:
//   header  : 8 x u32 { magic, version|entry_size<<16, nunits, chunk_bytes, payload_bytes, nbytes, uvarint_bytes, check }
//   entries : (nunits + 1) x u32 payload offset of the first element of unit k (entry nunits = payload_bytes)
#define HB_SNX_MAGIC     0x58534248u     // "HBSX"
#define HB_SNX_HDR_BYTES 32u
#define HB_SNX_ENTRY     4u
