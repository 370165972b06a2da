// Compiles go-blosc_amd/host/blosc.hpp against libhipblosc.so and exercises it the way the reference's
// blosc_test.go exercises the Go API.  Exit 0 = ok.  With no GPU it checks the no-fallback behaviour.
#include <cstdio>
#include <cstring>
#include "../../go-blosc_amd/host/blosc.hpp"

#define REQUIRE(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

int main() {
    using namespace blosc;
    Header h{2, LZ4, HB_FLAG_SHUFFLE, 4, 1000, 1000, 321};
    Bytes raw = h.bytes();
    REQUIRE(raw.size() == 16 && raw[0] == 2 && raw[3] == 4);
    Header p = ParseHeader(raw);
    REQUIRE(p.NBytesOrig == 1000 && p.NBytesComp == 321 && p.ShuffleMode() == Shuffle1);
    REQUIRE(to_string(LZ4) == "lz4" && to_string((Codec)9) == "unknown(9)" && to_string(BitShuffle) == "bitshuffle");
    try { Compress(Bytes{}, LZ4, 5, NoShuffle, 1); REQUIRE(false); } catch (const Error &e) { REQUIRE(is(e, ErrInvalidData)); }   // blosc_test.go:211-215
    try { Decompress(Bytes{2, 1}); REQUIRE(false); } catch (const Error &e) { REQUIRE(is(e, ErrInvalidHeader)); }                 // blosc_test.go:217-225
    REQUIRE(GetCodec(LZ4) && GetCodec(LZ4)->Name() == "lz4" && !GetCodec(ZSTD) && ListCodecs().size() >= 1);

    Bytes x(100000);
    for (size_t i = 0; i < x.size(); i++) x[i] = (uint8_t)(i % 256);                          // blosc_test.go:363-371
    if (hb_init() != HB_OK) {
        try { Compress(x, LZ4, 5, Shuffle1, 4); REQUIRE(false); } catch (const Error &e) { REQUIRE(is(e, ErrNoDevice)); }
        std::puts("host mirror ok (no device: compute entry points fail loudly)");
        return 0;
    }
    Options o = DefaultOptions();
    Bytes f = CompressWithOptions(x.data(), x.size(), o);
    REQUIRE(f.size() < x.size() && GetDecompressedSize(f) == (int)x.size() && GetInfo(f).TypeSize == 4);
    REQUIRE(Decompress(f) == x);
    Bytes y = x;
    ShuffleBuffer(y, 4, Shuffle1); REQUIRE(y != x); UnshuffleBuffer(y, 4, Shuffle1); REQUIRE(y == x);
    ShuffleBuffer(y, 4, BitShuffle); UnshuffleBuffer(y, 4, BitShuffle); REQUIRE(y == x);
    ShuffleBuffer(y, 4, (Shuffle)9); REQUIRE(y == x);
    Bytes c = GetCodec(LZ4)->Compress(x, 5);
    REQUIRE(GetCodec(LZ4)->Decompress(c, (int)x.size()) == x);
    std::puts("host mirror ok (device round trips)");
    return 0;
}
