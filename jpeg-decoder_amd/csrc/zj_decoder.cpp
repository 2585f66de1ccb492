// zj_decoder.cpp -- see zj_decoder.hpp.
#include "zj_decoder.hpp"

namespace zj {

Decoder::Decoder(FILE *file, int device) : file_(file), device_(device) {}

Decoder::~Decoder()
{
    if (file_) fclose(file_);
}

DecodeError Decoder::decode()
{
    std::vector<uint8_t> rgb;
    return decode_to(rgb, "out.ppm");
}

DecodeError Decoder::decode_to(std::vector<uint8_t> &rgb, const char *ppm_path)
{
    DecodeError e;
    if (!file_) { e.kind = DecodeError::Io; e.message = "file not open"; return e; }
    std::vector<uint8_t> data;
    uint8_t chunk[1 << 16];
    size_t got;
    while ((got = fread(chunk, 1, sizeof chunk, file_)) > 0) data.insert(data.end(), chunk, chunk + got);
    if (ferror(file_)) { e.kind = DecodeError::Io; e.message = "read error"; return e; }

    zj_parsed parsed;
    int rc = zj_parse(data.data(), data.size(), &parsed);
    if (rc == ZJ_E_UNSUPPORTED) { e.kind = DecodeError::Unsupported; e.message = "JPEG variant outside the baseline 3-component path"; return e; }
    if (rc != ZJ_OK) { e.kind = DecodeError::Format; e.message = "malformed JPEG container"; return e; }
    w_ = parsed.image.width;
    h_ = parsed.image.height;

    zj_ctx *ctx = zj_create(device_, nullptr);
    if (!ctx) { e.kind = DecodeError::Device; e.message = "no usable GPU (the decode path has no CPU fallback)"; return e; }
    rgb.assign((size_t)w_ * h_ * 3, 0);
    zj_output out = {rgb.data(), 0, nullptr, nullptr, ZJ_MEM_HOST};
    uint32_t status = 0;
    rc = zj_decode_batch(ctx, &parsed.image, 1, &out, &status);
    if (rc != ZJ_OK) { e.kind = DecodeError::Device; e.message = zj_last_error(ctx); zj_destroy(ctx); return e; }
    zj_destroy(ctx);
    if (status) { e.kind = DecodeError::Corrupt; e.status = status; e.message = "entropy-coded data is corrupt or truncated"; return e; }
    if (ppm_path && *ppm_path) {
        if (zj_write_ppm(ppm_path, w_, h_, rgb.data(), 0) != ZJ_OK) { e.kind = DecodeError::Io; e.message = "cannot write PPM"; return e; }
    }
    return e;
}

} // namespace zj
