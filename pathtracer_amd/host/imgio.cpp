// Image files for the host scene layer: what `image::open(path)?.to_rgb8()` (src/textures/image.rs:17-26) and
// `RgbImage::save("strat.png")` (src/main.rs:93) do in the reference through the `image` crate (0.25.2, a
// crates.io dependency that is not part of the reference tree).  Restated here for the two formats the scenes
// need: PNG (read: 8- and 16-bit grey, grey + alpha, RGB, RGBA and palette images, not interlaced; written: 8-bit
// RGB) and binary / plain PPM.  The deflate stream is zlib's; chunk layout, CRC and the five scan-line filters
// follow the PNG specification.  Conversion to RGB8 as `to_rgb8` does it: grey is replicated, alpha is dropped,
// 16-bit samples are scaled by (v + 128) / 257.
#include <zlib.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pthost.hpp"

namespace pt {

namespace {

bool read_file(const std::string &path, std::vector<uint8_t> &out, std::string &err) {
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) {
        err = "cannot open " + path;
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (n < 0) {
        std::fclose(f);
        err = "cannot size " + path;
        return false;
    }
    out.resize((size_t)n);
    const bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
    std::fclose(f);
    if (!ok) err = "cannot read " + path;
    return ok;
}

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}

const uint8_t PNG_MAGIC[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};

int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}

bool decode_png(const std::vector<uint8_t> &file, RgbImage &img, std::string &err) {
    size_t at = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool have_header = false, done = false;
    while (!done && at + 12 <= file.size()) {
        const uint32_t len = be32(&file[at]);
        const uint8_t *type = &file[at + 4];
        if (at + 12 + (size_t)len > file.size()) {
            err = "truncated chunk";
            return false;
        }
        const uint8_t *data = &file[at + 8];
        if (be32(data + len) != (uint32_t)crc32(crc32(0L, type, 4), data, len)) {
            err = "chunk checksum mismatch";
            return false;
        }
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len != 13) {
                err = "bad IHDR";
                return false;
            }
            w = be32(data);
            h = be32(data + 4);
            depth = data[8];
            color = data[9];
            interlace = data[12];
            if (data[10] != 0 || data[11] != 0) {
                err = "unknown compression or filter method";
                return false;
            }
            have_header = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        at += 12 + (size_t)len;
    }
    if (!have_header || !done) {
        err = "missing IHDR or IEND";
        return false;
    }
    if (interlace != 0) {
        err = "interlaced PNG files are not read";
        return false;
    }
    const int channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
    const bool depth_ok = (color == 0 && (depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16)) ||
                          (color == 3 && (depth == 1 || depth == 2 || depth == 4 || depth == 8)) ||
                          ((color == 2 || color == 4 || color == 6) && (depth == 8 || depth == 16));
    if (!channels || !depth_ok || w == 0 || h == 0 || (uint64_t)w * h > (1ull << 28)) {
        err = "unsupported colour type, bit depth or size";
        return false;
    }
    const size_t bits_per_pixel = (size_t)channels * depth, stride = ((size_t)w * bits_per_pixel + 7) / 8;
    const size_t bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;  // the filter's "previous pixel" distance in bytes
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) {
        err = "bad deflate stream";
        return false;
    }
    // undo the scan-line filters in place
    std::vector<uint8_t> zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        uint8_t *row = &raw[(stride + 1) * (size_t)y + 1];
        const uint8_t *up = y ? &raw[(stride + 1) * (size_t)(y - 1) + 1] : zero.data();
        const int filter = row[-1];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? row[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
            int v = row[x];
            switch (filter) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) / 2; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "unknown scan-line filter"; return false;
            }
            row[x] = (uint8_t)v;
        }
    }
    if (color == 3 && palette.size() < 3) {
        err = "palette image without a palette";
        return false;
    }
    img = RgbImage(w, h);
    auto sample = [&](const uint8_t *row, size_t index) -> uint32_t {  // sample `index` of the row, at the image's bit depth
        if (depth == 8) return row[index];
        if (depth == 16) return ((uint32_t)row[2 * index] << 8) | row[2 * index + 1];
        const size_t bit = index * (size_t)depth;
        return (row[bit / 8] >> (8 - depth - (int)(bit % 8))) & ((1u << depth) - 1u);
    };
    auto to8 = [&](uint32_t v) -> uint8_t {
        if (depth == 8) return (uint8_t)v;
        if (depth == 16) return (uint8_t)((v + 128u) / 257u);
        return (uint8_t)(v * 255u / ((1u << depth) - 1u));  // 1, 2 and 4 bits: spread over 0..255
    };
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t *row = &raw[(stride + 1) * (size_t)y + 1];
        uint8_t *out = &img.data[(size_t)y * w * 3];
        for (uint32_t x = 0; x < w; x++) {
            if (color == 3) {
                const uint32_t k = sample(row, x);
                if (3 * (size_t)k + 2 >= palette.size()) {
                    err = "palette index out of range";
                    return false;
                }
                std::memcpy(out + 3 * x, &palette[3 * (size_t)k], 3);
            } else if (channels <= 2) {
                out[3 * x] = out[3 * x + 1] = out[3 * x + 2] = to8(sample(row, (size_t)x * channels));
            } else {
                for (int c = 0; c < 3; c++) out[3 * x + c] = to8(sample(row, (size_t)x * channels + c));
            }
        }
    }
    return true;
}

// P6 (binary) and P3 (plain) with a maximum value of at most 65535; comments allowed in the header.
bool decode_ppm(const std::vector<uint8_t> &file, RgbImage &img, std::string &err) {
    size_t at = 2;
    const bool binary = file[1] == '6';
    auto next_int = [&](uint32_t &out) {
        for (;;) {
            while (at < file.size() && std::isspace(file[at])) at++;
            if (at < file.size() && file[at] == '#') {
                while (at < file.size() && file[at] != '\n') at++;
                continue;
            }
            break;
        }
        if (at >= file.size() || !std::isdigit(file[at])) return false;
        uint64_t v = 0;
        while (at < file.size() && std::isdigit(file[at])) {
            v = v * 10 + (file[at++] - '0');
            if (v > 0xFFFFFFFFull) return false;
        }
        out = (uint32_t)v;
        return true;
    };
    uint32_t w, h, maxv;
    if (!next_int(w) || !next_int(h) || !next_int(maxv) || w == 0 || h == 0 || maxv == 0 || maxv > 65535 || (uint64_t)w * h > (1ull << 28)) {
        err = "bad PPM header";
        return false;
    }
    img = RgbImage(w, h);
    const size_t n = (size_t)w * h * 3;
    auto scale = [&](uint32_t v) { return (uint8_t)(maxv == 255 ? v : (v * 255u + maxv / 2) / maxv); };
    if (binary) {
        at++;  // the single whitespace byte after the maximum value
        const size_t bytes = maxv < 256 ? 1 : 2;
        if (at + n * bytes > file.size()) {
            err = "truncated PPM data";
            return false;
        }
        for (size_t i = 0; i < n; i++) img.data[i] = scale(bytes == 1 ? file[at + i] : ((uint32_t)file[at + 2 * i] << 8) | file[at + 2 * i + 1]);
    } else {
        for (size_t i = 0; i < n; i++) {
            uint32_t v;
            if (!next_int(v) || v > maxv) {
                err = "bad PPM sample";
                return false;
            }
            img.data[i] = scale(v);
        }
    }
    return true;
}

}  // namespace

std::shared_ptr<const RgbImage> RgbImage::load(const std::string &path, std::string *why) {
    std::vector<uint8_t> file;
    std::string err;
    auto img = std::make_shared<RgbImage>();
    bool ok = read_file(path, file, err);
    if (ok) {
        if (file.size() >= 8 && !std::memcmp(file.data(), PNG_MAGIC, 8)) ok = decode_png(file, *img, err);
        else if (file.size() >= 2 && file[0] == 'P' && (file[1] == '6' || file[1] == '3')) ok = decode_ppm(file, *img, err);
        else {
            ok = false;
            err = "neither a PNG nor a PPM file";
        }
    }
    if (!ok) {
        if (why) *why = path + ": " + err;
        return nullptr;
    }
    return img;
}

bool RgbImage::save_png(const std::string &path) const {
    if (width == 0 || height == 0 || data.size() != (size_t)width * height * 3) return false;
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)width * 3 + 1) * height);
    for (uint32_t y = 0; y < height; y++) {
        raw.push_back(0);  // filter type None
        raw.insert(raw.end(), data.begin() + (size_t)y * width * 3, data.begin() + (size_t)(y + 1) * width * 3);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    z.resize(zlen);
    std::vector<uint8_t> out(PNG_MAGIC, PNG_MAGIC + 8);
    auto chunk = [&](const char *type, const std::vector<uint8_t> &body) {
        put_be32(out, (uint32_t)body.size());
        const size_t start = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), body.begin(), body.end());
        put_be32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};  // 8 bits, RGB, deflate, adaptive filtering, not interlaced
    ihdr.insert(ihdr.end(), tail, tail + 5);
    chunk("IHDR", ihdr);
    chunk("IDAT", z);
    chunk("IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    return ok;
}

namespace Image {
// Image::load (src/textures/image.rs:17-26): a file that cannot be read gives the texture without an image, which
// shows the reference's built-in checker (:54-63), and the failure is reported on stderr as the reference logs it.
TextureRef load(const std::string &path) {
    std::string why;
    auto img = RgbImage::load(path, &why);
    if (!img) std::fprintf(stderr, "Failed to load image: '%s'... Using default texture instead... (%s)\n", path.c_str(), why.c_str());
    return make(std::move(img));
}
}  // namespace Image

}  // namespace pt

// ---- C entry points (include/pthost.h)
extern "C" int pth_image_load(const char *path, uint32_t *width, uint32_t *height, uint8_t *rgb8, size_t capacity, char *err, size_t errlen) {
    std::string why;
    auto img = path ? pt::RgbImage::load(path, &why) : nullptr;
    if (!img) {
        if (err && errlen) std::snprintf(err, errlen, "%s", path ? why.c_str() : "path is NULL");
        return 1;
    }
    if (width) *width = img->width;
    if (height) *height = img->height;
    if (rgb8) {
        if (capacity < img->data.size()) {
            if (err && errlen) std::snprintf(err, errlen, "the image needs %zu bytes", img->data.size());
            return 1;
        }
        std::memcpy(rgb8, img->data.data(), img->data.size());
    }
    return 0;
}

extern "C" int pth_image_save_png(const char *path, uint32_t width, uint32_t height, const uint8_t *rgb8) {
    if (!path || !rgb8) return 1;
    pt::RgbImage img(width, height);
    std::memcpy(img.data.data(), rgb8, img.data.size());
    return img.save_png(path) ? 0 : 1;
}
