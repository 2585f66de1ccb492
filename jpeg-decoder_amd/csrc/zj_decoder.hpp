// zj_decoder.hpp -- C++ host mirror of the reference's public API (src/lib.rs:5, src/decoder.rs:148-156,
// :322-330, src/error.rs:3-12) on top of the C ABI in include/zj.h.  The reference is a compiled Rust
// crate; no Rust toolchain exists in the build image, so the host side above the C ABI is C++ with
// the same names, argument meaning and error behaviour:
//     Decoder::new(file)      -> zj::Decoder(FILE*)   (takes ownership of the open file)
//     decoder.decode()        -> zj::Decoder::decode() returns DecodeError (Ok on success) and
//                                writes out.ppm into the current directory (src/decoder.rs:334,580,648)
#pragma once
#include <stdint.h>
#include <stdio.h>

#include <string>
#include <vector>

#include "../../include/zj.h"

namespace zj {

// Counterpart of `pub enum DecodeError { Io(io::Error) }` (src/error.rs:4-6).  Everything the
// reference reports by panicking is reported here as one of the additional kinds instead.
struct DecodeError {
    enum Kind { Ok = 0, Io, Format, Unsupported, Device, Corrupt } kind = Ok;
    std::string message;
    uint32_t status = 0; // per-image status word (ZJ_ST_*) when kind == Corrupt
    bool ok() const { return kind == Ok; }
};

class Decoder {
public:
    explicit Decoder(FILE *file, int device = 0); // Decoder::new (src/decoder.rs:323-328)
    ~Decoder();
    Decoder(const Decoder &) = delete;
    Decoder &operator=(const Decoder &) = delete;

    // Decoder::decode (src/decoder.rs:330): parse on the host, decode on the GPU, write out.ppm.
    DecodeError decode();
    // Same, but keep the pixels in memory and optionally choose the output path ("" = no file).
    DecodeError decode_to(std::vector<uint8_t> &rgb, const char *ppm_path);

    uint16_t width() const { return w_; }  // Dimensions (src/decoder.rs:153-156)
    uint16_t height() const { return h_; }

private:
    FILE *file_;
    int device_;
    uint16_t w_ = 0, h_ = 0;
};

} // namespace zj
