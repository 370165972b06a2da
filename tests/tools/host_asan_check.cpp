// Host-only parts of the library under AddressSanitizer + UBSan (sanitizers run on the CPU build only; the GPU pool
// refuses them).  Built by tests/test_host_cpp.py from the SAME sources the product compiles: csrc/hb_host.cpp (header
// parse / serialise, bounds, error strings) and csrc/hb_ticket_ring.h (the queue's kept-result ring).
//   host_asan_check <blobs.bin>     blobs.bin = repeated { u32 len, bytes } : the reference's fuzz seeds + mutations
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../go-blosc_amd/csrc/hb_host.cpp"
#include "../../go-blosc_amd/csrc/hb_ticket_ring.h"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    FILE *f = std::fopen(argv[1], "rb");
    if (!f) return 2;
    size_t nblobs = 0, ok = 0;
    for (;;) {
        uint32_t len;
        if (std::fread(&len, 4, 1, f) != 1) break;
        // exact-size heap copy: any read past the end is an ASan error
        uint8_t *b = (uint8_t *)std::malloc(len ? len : 1);
        if (len && std::fread(b, 1, len, f) != len) return 2;
        hb_header h;
        std::memset(&h, 0xA5, sizeof h);
        const int rc = hb_parse_header(len ? b : nullptr, len, &h);
        if (len < 16) REQUIRE(rc == HB_ERR_INVALID_HEADER);                  // blosc.go:166-168
        else if (b[0] != 2) REQUIRE(rc == HB_ERR_INVALID_VERSION);           // blosc.go:180-182
        else {
            REQUIRE(rc == HB_OK);
            uint8_t out[16];
            hb_header_bytes(&h, out);
            REQUIRE(std::memcmp(out, b, 16) == 0);                           // Bytes() round trip, fuzz_test.go:400-421
            ok++;
        }
        std::free(b);
        nblobs++;
    }
    std::fclose(f);
    REQUIRE(nblobs > 300 && ok > 250);
    REQUIRE(hb_parse_header(nullptr, 16, nullptr) == HB_ERR_BAD_ARG);
    // bounds: codec.go:65 and monotone growth up to the uint32 limit of the format
    REQUIRE(hb_lz4_bound(0) == 16 && hb_lz4_bound(255) == 255 + 1 + 16 && hb_lz4_bound(1u << 30) == (1u << 30) + (1u << 30) / 255 + 16);
    size_t prev = 0;
    for (size_t n = 0; n < ((size_t)1 << 33); n = n * 3 + 1) {
        const size_t fb = hb_frame_bound(n);
        REQUIRE(fb > prev && fb >= 16 + hb_lz4_bound(n) + hb_index_bound(n));
        prev = fb;
    }
    for (int c = -20; c <= 1; c++) REQUIRE(hb_strerror(c) != nullptr && std::strlen(hb_strerror(c)) > 1);
    REQUIRE(std::strcmp(hb_strerror(HB_ERR_INVALID_DATA), "blosc: invalid compressed data") == 0);   // blosc.go:127
    // ticket ring: newest `cap` kept, each answers once
    hb_ticket_ring ring(12);
    for (int64_t t = 0; t < 100; t++) ring.put(t, t * 7);
    int64_t rc = -1;
    REQUIRE(!ring.take(0, &rc) && !ring.take(87, &rc));
    REQUIRE(ring.take(88, &rc) && rc == 88 * 7 && !ring.take(88, &rc));
    REQUIRE(ring.take(99, &rc) && rc == 99 * 7);
    for (int64_t t = 89; t < 99; t++) REQUIRE(ring.take(t, &rc) && rc == t * 7);
    REQUIRE(ring.kept.empty());
    hb_ticket_ring none(0);
    none.put(1, 2);
    REQUIRE(!none.take(1, &rc));
    std::puts("host helpers ok under ASan + UBSan");
    return 0;
}
