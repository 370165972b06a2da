// blosc.hpp — header-only C++17 mirror of go-blosc's public API over the hipblosc C ABI.
//
// The Go toolchain is absent in this image, so this is the compiled-language host side above the boundary
// (go/blosc_hip.go is the cgo shim a maintainer would add; it binds the same symbols).  Same names, argument
// meaning and error behaviour as the reference package: blosc.go:49-317, shuffle.go:298-323, codec.go:15-53.
// Every O(n) operation runs on the MI355X through libhipblosc.so; there is no CPU fallback.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hipblosc.h"

namespace blosc {

using Bytes = std::vector<uint8_t>;

// ---- constants, blosc.go:49-52, :118-121 ----
inline constexpr const char *Version = "1.0.0";
inline constexpr int FormatVersion = 2;
inline constexpr int HeaderSize = 16, MinHeaderSize = 16;

enum Codec : uint8_t { BloscLZ = 0, LZ4 = 1, LZ4HC = 2, Snappy = 3, ZLIB = 4, ZSTD = 5 };   // blosc.go:57-64
enum Shuffle : uint8_t { NoShuffle = 0, Shuffle1 = 1, BitShuffle = 2 };                      // blosc.go:89-93

inline std::string to_string(Codec c) {                                                     // blosc.go:67-84
    switch (c) {
    case BloscLZ: return "blosclz"; case LZ4: return "lz4"; case LZ4HC: return "lz4hc";
    case Snappy: return "snappy"; case ZLIB: return "zlib"; case ZSTD: return "zstd";
    default: return "unknown(" + std::to_string((int)c) + ")";
    }
}
inline std::string to_string(Shuffle s) {                                                   // blosc.go:96-107
    switch (s) {
    case NoShuffle: return "noshuffle"; case Shuffle1: return "shuffle"; case BitShuffle: return "bitshuffle";
    default: return "unknown(" + std::to_string((int)s) + ")";
    }
}

// ---- errors: one exception type carrying the sentinel (blosc.go:125-149).  is(e, ErrX) plays errors.Is ----
enum Sentinel { ErrInvalidData = HB_ERR_INVALID_DATA, ErrInvalidHeader = HB_ERR_INVALID_HEADER,
                ErrInvalidVersion = HB_ERR_INVALID_VERSION, ErrInvalidCodec = HB_ERR_INVALID_CODEC,
                ErrSizeMismatch = HB_ERR_SIZE_MISMATCH, ErrDataTooLarge = HB_ERR_DATA_TOO_LARGE,
                ErrCompressionFailed = HB_ERR_COMPRESSION_FAILED, ErrDecompressionFailed = HB_ERR_DECOMPRESSION_FAILED,
                ErrNoDevice = HB_ERR_NO_DEVICE, ErrHip = HB_ERR_HIP, ErrBadArg = HB_ERR_BAD_ARG, ErrShortBuffer = HB_ERR_SHORT_BUFFER };

class Error : public std::runtime_error {
  public:
    explicit Error(int code) : std::runtime_error(hb_strerror(code)), code_(code) {}
    int code() const { return code_; }
  private:
    int code_;
};
inline bool is(const Error &e, Sentinel s) { return e.code() == (int)s; }
inline int64_t check(int64_t rc) { if (rc < 0) throw Error((int)rc); return rc; }

// ---- Header, blosc.go:154-224 ----
struct Header {
    uint8_t Version = 0, VersionLZ = 0, Flags = 0, TypeSize = 0;
    uint32_t NBytesOrig = 0, BlockSize = 0, NBytesComp = 0;
    Bytes bytes() const {                                                                   // blosc.go:188-198
        hb_header h{Version, VersionLZ, Flags, TypeSize, NBytesOrig, BlockSize, NBytesComp};
        Bytes out(HeaderSize);
        hb_header_bytes(&h, out.data());
        return out;
    }
    bool HasShuffle() const { return Flags & HB_FLAG_SHUFFLE; }                             // blosc.go:201-203
    bool HasBitShuffle() const { return Flags & HB_FLAG_BITSHUFFLE; }                       // blosc.go:206-208
    bool IsMemcpy() const { return Flags & HB_FLAG_MEMCPY; }                                // blosc.go:211-213
    Shuffle ShuffleMode() const { return HasBitShuffle() ? BitShuffle : HasShuffle() ? Shuffle1 : NoShuffle; }   // :216-224
};

inline Header ParseHeader(const uint8_t *data, size_t n) {                                  // blosc.go:165-185
    hb_header h;
    check(hb_parse_header(data, n, &h));
    return Header{h.version, h.codec, h.flags, h.typesize, h.nbytes, h.blocksize, h.cbytes};
}
inline Header ParseHeader(const Bytes &d) { return ParseHeader(d.data(), d.size()); }

// ---- Options, blosc.go:227-245 ----
struct Options {
    Codec codec = LZ4;
    int Level = 0;
    Shuffle shuffle = NoShuffle;
    int TypeSize = 0;
    int BlockSize = 0;      // accepted and ignored, as in the reference (blosc.go:232)
    int NumThreads = 0;     // accepted and ignored (blosc.go:233)
    unsigned hip_opts = 0;  // HB_OPT_* (no counterpart in the reference)
    int device = 0;
};
inline Options DefaultOptions() { Options o; o.codec = LZ4; o.Level = 5; o.shuffle = Shuffle1; o.TypeSize = 4; return o; }

// ---- Compress / Decompress, blosc.go:257-317 ----
inline Bytes CompressWithOptions(const uint8_t *data, size_t n, const Options &o) {         // blosc.go:268-286 + :320-374
    if (n == 0) throw Error(HB_ERR_INVALID_DATA);
    Bytes out(hb_frame_bound(n));
    const int64_t rc = check(hb_compress_frame(data, n, out.data(), out.size(), o.codec, o.Level, o.shuffle, o.TypeSize,
                                               o.hip_opts, o.device));
    out.resize((size_t)rc);
    return out;
}
inline Bytes Compress(const Bytes &data, Codec codec, int level, Shuffle shuffle, int typeSize) {   // blosc.go:257-265
    Options o; o.codec = codec; o.Level = level; o.shuffle = shuffle; o.TypeSize = typeSize;
    return CompressWithOptions(data.data(), data.size(), o);
}
inline Bytes DecompressWithSize(const uint8_t *data, size_t n, int typeSize, int device = 0) {      // blosc.go:296-303 + :377-434
    if (n < (size_t)HeaderSize) throw Error(HB_ERR_INVALID_HEADER);
    const Header h = ParseHeader(data, n);
    Bytes out(h.NBytesOrig ? h.NBytesOrig : 1);
    const int64_t rc = check(hb_decompress_frame(data, n, out.data(), h.NBytesOrig, typeSize, device));
    out.resize((size_t)rc);
    return out;
}
inline Bytes Decompress(const Bytes &data) { return DecompressWithSize(data.data(), data.size(), 0); }          // blosc.go:291-293
inline Header GetInfo(const Bytes &data) { return ParseHeader(data); }                                            // blosc.go:306-308
inline int GetDecompressedSize(const Bytes &data) { return (int)ParseHeader(data).NBytesOrig; }                   // blosc.go:311-317

// ---- filters, shuffle.go:298-323 (in place; unknown mode / NoShuffle = no-op) ----
inline void filter_in_place(int op, Bytes &data, int typeSize, int device) {
    Bytes out(data.size());
    check(hb_filter(op, out.data(), data.data(), data.size(), typeSize, device));
    data.swap(out);
}
inline void ShuffleBuffer(Bytes &data, int typeSize, Shuffle mode, int device = 0) {
    if (mode == Shuffle1) filter_in_place(HB_OP_SHUFFLE, data, typeSize, device);
    else if (mode == BitShuffle) filter_in_place(HB_OP_BITSHUFFLE, data, typeSize, device);
}
inline void UnshuffleBuffer(Bytes &data, int typeSize, Shuffle mode, int device = 0) {
    if (mode == Shuffle1) filter_in_place(HB_OP_UNSHUFFLE, data, typeSize, device);
    else if (mode == BitShuffle) filter_in_place(HB_OP_BITUNSHUFFLE, data, typeSize, device);
}

// ---- codec plugin seam, codec.go:15-53 ----
struct CodecInterface {
    virtual ~CodecInterface() = default;
    virtual Bytes Compress(const Bytes &data, int level) = 0;
    virtual Bytes Decompress(const Bytes &data, int expectedSize) = 0;
    virtual std::string Name() const = 0;
};
struct HipLZ4Codec : CodecInterface {                                                      // replaces lz4Codec, codec.go:59-84
    int device = 0;
    std::string Name() const override { return "lz4"; }
    Bytes Compress(const Bytes &data, int) override {
        Bytes out(hb_lz4_bound(data.size()));
        out.resize((size_t)check(hb_lz4_compress(data.data(), data.size(), out.data(), out.size(), device)));
        return out;
    }
    Bytes Decompress(const Bytes &data, int expectedSize) override {
        Bytes out(expectedSize > 0 ? expectedSize : 1);
        out.resize((size_t)check(hb_lz4_decompress(data.data(), data.size(), out.data(), (size_t)expectedSize, device)));
        return out;
    }
};
struct HipDeviceCodec : CodecInterface {                                  // replaces lz4hcCodec (codec.go:90-128) / snappyCodec (:228-244)
    Codec codec; std::string name; int device = 0;
    HipDeviceCodec(Codec c, std::string n) : codec(c), name(std::move(n)) {}
    std::string Name() const override { return name; }
    Bytes Compress(const Bytes &data, int level) override {
        Bytes out(hb_codec_bound(codec, data.size()));
        out.resize((size_t)check(hb_codec_compress(codec, level, data.data(), data.size(), out.data(), out.size(), device)));
        return out;
    }
    Bytes Decompress(const Bytes &data, int expectedSize) override {
        Bytes out(expectedSize > 0 ? expectedSize : 1);
        out.resize((size_t)check(hb_codec_decompress(codec, data.data(), data.size(), out.data(), (size_t)expectedSize, device)));
        return out;
    }
};
inline std::map<Codec, std::shared_ptr<CodecInterface>> &registry() {
    static std::map<Codec, std::shared_ptr<CodecInterface>> r{{LZ4, std::make_shared<HipLZ4Codec>()},
                                                              {LZ4HC, std::make_shared<HipDeviceCodec>(LZ4HC, "lz4hc")},
                                                              {Snappy, std::make_shared<HipDeviceCodec>(Snappy, "snappy")}};
    return r;
}
inline std::mutex &registry_mu() { static std::mutex m; return m; }                        // the reference's map is unguarded (codec.go:36-38)
inline void RegisterCodec(Codec id, std::shared_ptr<CodecInterface> c) { std::lock_guard<std::mutex> l(registry_mu()); registry()[id] = std::move(c); }
inline std::shared_ptr<CodecInterface> GetCodec(Codec id) { std::lock_guard<std::mutex> l(registry_mu()); auto it = registry().find(id); return it == registry().end() ? nullptr : it->second; }
inline std::vector<Codec> ListCodecs() { std::lock_guard<std::mutex> l(registry_mu()); std::vector<Codec> v; for (auto &kv : registry()) v.push_back(kv.first); return v; }

}  // namespace blosc
