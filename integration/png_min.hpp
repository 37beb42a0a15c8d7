// png_min.hpp — PNG decoding to RGBA8 (rows top-down) and a plain RGB8 PNG writer, on zlib.
// Role of lodepng::decode in tex_image::createImage (tex_image.h:25-31), whose results it reproduces byte for
// byte: colour types 0/2/3/4/6 at every legal bit depth, tRNS keys compared at full sample width, 16-bit
// samples reduced to their high byte, sub-byte grey scaled by 255/(2^depth - 1), Adam7 interlacing, a palette
// index beyond the palette read as opaque black; chunk CRCs are verified and an unknown critical chunk, a bad
// filter type, a colour type / depth pair the format does not allow, truncated data or a header that announces more
// than 2^28 - 1 pixels (or a 32-bit pixel count that wraps) are errors.
#pragma once
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace rt_hip {

struct ImageRGBA8 { uint32_t width = 0, height = 0; std::vector<uint8_t> rgba; };

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline std::vector<uint8_t> read_file(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<uint8_t> data;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + n);
    fclose(f);
    return data;
}

inline ImageRGBA8 decode_png(const std::vector<uint8_t> &file, const std::string &name = "png") {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto fail = [&](const char *what) -> void { throw std::runtime_error(name + ": " + what); };
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) fail("not a PNG file");
    uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    size_t pos = 8;
    bool end = false;
    while (!end && pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *body = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) fail("truncated chunk");
        if ((uint32_t)crc32(0L, type, 4 + len) != be32(body + len)) fail("chunk checksum mismatch");
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) fail("bad IHDR");
            W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) end = true;
        else if (!(type[0] & 0x20)) fail("unknown critical chunk");
        pos += 12 + (size_t)len;
    }
    if (W == 0 || H == 0 || ctype < 0) fail("missing IHDR");
    // Size limit of the reference's decoder (lodepng.cpp:4547-4553): the pixel count is formed in 32 bits, a product that
    // wraps is rejected, and so is anything above 2^28 - 1 pixels.  Every size computed below then fits 64 bits with room
    // to spare (at most 2^28 pixels x 8 bytes + 2^28 filter bytes), and no crafted header can make the buffers disagree.
    {
        const uint32_t numpixels = W * H;                       // wraps like the reference's `unsigned` product
        if (numpixels / H != W || numpixels > 268435455u) fail("image too large");
    }
    if (interlace > 1) fail("unknown interlace method");
    int channels;
    switch (ctype) {
    case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break;
    case 4: channels = 2; break; case 6: channels = 4; break;
    default: fail("unknown colour type"); return {};
    }
    bool sub_byte = depth == 1 || depth == 2 || depth == 4;
    if (!((depth == 8) || (depth == 16 && ctype != 3) || (sub_byte && (ctype == 0 || ctype == 3)))) fail("unsupported bit depth");
    const size_t bits_pp = (size_t)channels * depth, bpp = (bits_pp + 7) / 8;
    // sub-images of the stream: the whole picture, or the seven Adam7 passes {x0, y0, dx, dy}
    struct Pass { uint32_t x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (!interlace) passes.push_back({0, 0, 1, 1, W, H});
    else {
        static const uint32_t A7[7][4] = {{0, 0, 8, 8}, {4, 0, 8, 8}, {0, 4, 4, 8}, {2, 0, 4, 4}, {0, 2, 2, 4}, {1, 0, 2, 2}, {0, 1, 1, 2}};
        for (const auto &a : A7) {
            uint32_t w = W > a[0] ? (W - a[0] + a[2] - 1) / a[2] : 0, h = H > a[1] ? (H - a[1] + a[3] - 1) / a[3] : 0;
            if (w && h) passes.push_back({a[0], a[1], a[2], a[3], w, h});
        }
    }
    size_t raw_size = 0;
    for (const Pass &p : passes) raw_size += (((size_t)p.w * bits_pp + 7) / 8 + 1) * p.h;
    std::vector<uint8_t> raw(raw_size);
    uLongf out_len = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || out_len != raw.size()) fail("corrupt image data");
    ImageRGBA8 img; img.width = W; img.height = H; img.rgba.resize((size_t)W * H * 4);
    size_t pass_off = 0;
    for (const Pass &ps : passes) {
    const size_t stride = ((size_t)ps.w * bits_pp + 7) / 8;
    // undo the per-row filters of this sub-image in place
    std::vector<uint8_t> prev(stride, 0);
    for (uint32_t y = 0; y < ps.h; y++) {
        uint8_t *row = &raw[pass_off + (stride + 1) * (size_t)y];
        uint8_t ft = row[0];
        uint8_t *px = row + 1;
        for (size_t i = 0; i < stride; i++) {
            int a = i >= bpp ? px[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, pred;
            switch (ft) {
            case 0: pred = 0; break; case 1: pred = a; break; case 2: pred = b; break; case 3: pred = (a + b) >> 1; break;
            case 4: { int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                      pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: fail("bad filter type"); pred = 0;
            }
            px[i] = (uint8_t)(px[i] + pred);
        }
        prev.assign(px, px + stride);
    }
    auto sample = [&](const uint8_t *px, size_t x, int ch) -> uint32_t {       // sample scaled to 8 bits
        if (depth == 8) return px[x * channels + ch];
        if (depth == 16) return px[(x * channels + ch) * 2];                   // high byte
        size_t bit = x * (size_t)depth;
        uint32_t v = (px[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
        return ctype == 3 ? v : v * 255u / ((1u << depth) - 1u);
    };
    for (uint32_t y = 0; y < ps.h; y++) {
        const uint8_t *px = &raw[pass_off + (stride + 1) * (size_t)y + 1];
        for (uint32_t x = 0; x < ps.w; x++) {
            uint8_t *dst = &img.rgba[((size_t)(ps.y0 + y * ps.dy) * W + (ps.x0 + x * ps.dx)) * 4];
            switch (ctype) {
            case 0: {
                uint32_t g = sample(px, x, 0);
                dst[0] = dst[1] = dst[2] = (uint8_t)g; dst[3] = 255;
                if (trns.size() >= 2) {
                    uint32_t key = ((uint32_t)trns[0] << 8) | trns[1];
                    uint32_t rawv = depth == 16 ? (((uint32_t)px[x * 2] << 8) | px[x * 2 + 1])
                                  : (depth == 8 ? px[x] : ((px[(x * depth) >> 3] >> (8 - depth - ((x * depth) & 7))) & ((1u << depth) - 1u)));
                    if (rawv == key) dst[3] = 0;
                }
                break;
            }
            case 2:
                dst[0] = (uint8_t)sample(px, x, 0); dst[1] = (uint8_t)sample(px, x, 1); dst[2] = (uint8_t)sample(px, x, 2); dst[3] = 255;
                if (trns.size() >= 6) {
                    bool same = true;
                    for (int ch = 0; ch < 3 && same; ch++) {
                        uint32_t key = ((uint32_t)trns[2 * ch] << 8) | trns[2 * ch + 1];
                        uint32_t rawv = depth == 16 ? (((uint32_t)px[(x * 3 + ch) * 2] << 8) | px[(x * 3 + ch) * 2 + 1]) : px[x * 3 + ch];
                        same = rawv == key;
                    }
                    if (same) dst[3] = 0;
                }
                break;
            case 3: {
                uint32_t i = sample(px, x, 0);
                if ((size_t)i * 3 + 2 >= plte.size()) { dst[0] = dst[1] = dst[2] = 0; dst[3] = 255; break; }   // as the reference's decoder does
                dst[0] = plte[i * 3]; dst[1] = plte[i * 3 + 1]; dst[2] = plte[i * 3 + 2];
                dst[3] = i < trns.size() ? trns[i] : 255;
                break;
            }
            case 4:
                dst[0] = dst[1] = dst[2] = (uint8_t)sample(px, x, 0); dst[3] = (uint8_t)sample(px, x, 1);
                break;
            default:
                for (int ch = 0; ch < 4; ch++) dst[ch] = (uint8_t)sample(px, x, ch);
            }
        }
    }
    pass_off += (stride + 1) * (size_t)ps.h;
    }
    return img;
}

inline ImageRGBA8 decode_png_file(const std::string &path) { return decode_png(read_file(path), path); }

// RGB8 rows top-down -> PNG file (filter 0, zlib default compression)
inline void write_png_rgb8(const std::string &path, uint32_t W, uint32_t H, const uint8_t *rgb) {
    std::vector<uint8_t> raw(((size_t)W * 3 + 1) * H);
    for (uint32_t y = 0; y < H; y++) {
        raw[((size_t)W * 3 + 1) * y] = 0;
        memcpy(&raw[((size_t)W * 3 + 1) * y + 1], rgb + (size_t)y * W * 3, (size_t)W * 3);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress(comp.data(), &clen, raw.data(), (uLong)raw.size()) != Z_OK) throw std::runtime_error("png: compression failed");
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    auto put32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; };
    auto chunk = [&](const char *type, const uint8_t *body, uint32_t len) {
        uint8_t head[8]; put32(head, len); memcpy(head + 4, type, 4);
        fwrite(head, 1, 8, f);
        if (len) fwrite(body, 1, len, f);
        uLong crc = crc32(0L, (const Bytef *)type, 4);
        if (len) crc = crc32(crc, body, len);
        uint8_t tail[4]; put32(tail, (uint32_t)crc);
        fwrite(tail, 1, 4, f);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    fwrite(sig, 1, 8, f);
    uint8_t ihdr[13]; put32(ihdr, W); put32(ihdr + 4, H); ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    fclose(f);
}

}  // namespace rt_hip
