#include "image.hpp"

#include <zlib.h>

#include <cstdlib>
#include <cstring>
#include <fstream>

namespace {

bool ReadAll(const std::filesystem::path &path, std::vector<uint8_t> &out) {
    std::ifstream f(path, std::ios::binary);
    if (!f) return false;
    f.seekg(0, std::ios::end);
    std::streamoff n = f.tellg();
    f.seekg(0, std::ios::beg);
    out.resize((size_t)n);
    if (n > 0) f.read(reinterpret_cast<char *>(out.data()), n);
    return true;
}

inline uint32_t Be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint32_t Pack(uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }

// PNG specification, section 9: the five scanline filters, undone in place
bool Unfilter(std::vector<uint8_t> &raw, size_t rows, size_t stride, size_t bpp) {
    for (size_t y = 0; y < rows; y++) {
        uint8_t *line = raw.data() + y * (stride + 1);
        const uint8_t type = line[0];
        uint8_t *cur = line + 1;
        const uint8_t *up = y ? cur - (stride + 1) : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int add;
            switch (type) {
                case 0: add = 0; break;
                case 1: add = a; break;
                case 2: add = b; break;
                case 3: add = (a + b) >> 1; break;
                case 4: {
                    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                    add = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default: return false;
            }
            cur[i] = (uint8_t)(cur[i] + add);
        }
    }
    return true;
}

std::shared_ptr<Image> DecodePng(const std::vector<uint8_t> &file, std::string &error) {
    size_t pos = 8;
    uint32_t width = 0, height = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette, trns;
    bool seen_header = false, seen_end = false;
    while (pos + 12 <= file.size() && !seen_end) {
        const uint32_t len = Be32(&file[pos]);
        const char *type = reinterpret_cast<const char *>(&file[pos + 4]);
        if (pos + 12 + (size_t)len > file.size()) { error = "truncated PNG chunk"; return nullptr; }
        const uint8_t *data = &file[pos + 8];
        if (!std::memcmp(type, "IHDR", 4) && len >= 13) {
            width = Be32(data); height = Be32(data + 4);
            depth = data[8]; color = data[9]; interlace = data[12];
            seen_header = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(data, data + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            seen_end = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!seen_header || width == 0 || height == 0) { error = "PNG without a header"; return nullptr; }
    if (interlace != 0) { error = "interlaced PNG files are not read"; return nullptr; }
    int channels;
    switch (color) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: error = "unknown PNG colour type"; return nullptr;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) || (color == 3 && depth == 16) || ((color == 2 || color == 4 || color == 6) && depth < 8)) {
        error = "unsupported PNG bit depth";
        return nullptr;
    }
    const size_t bits_per_pixel = (size_t)channels * depth;
    const size_t stride = ((size_t)width * bits_per_pixel + 7) / 8;
    const size_t bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;
    // zlib expands by at most about 1032 : 1, so a header that announces more pixels than the compressed data can
    // hold is damaged; refuse it before it sizes an allocation
    if (width == 0 || height == 0 || (stride + 1) * (uint64_t)height > 1040ull * idat.size() + 4096) { error = "PNG header does not match its image data"; return nullptr; }
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    uLongf out_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size()) { error = "PNG image data does not inflate"; return nullptr; }
    if (!Unfilter(raw, height, stride, bpp)) { error = "unknown PNG filter type"; return nullptr; }

    auto image = std::make_shared<Image>();
    image->width = width;
    image->height = height;
    image->rgba.resize((size_t)width * height);
    auto sample = [&](const uint8_t *line, size_t index) -> uint32_t {       // sample `index` of a scanline, scaled to 8 bits except palette indices
        if (depth == 8) return line[index];
        if (depth == 16) return line[2 * index];
        const size_t bit = index * (size_t)depth;
        const uint32_t v = (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1u);
        return color == 3 ? v : v * 255u / ((1u << depth) - 1u);
    };
    for (uint32_t y = 0; y < height; y++) {
        const uint8_t *line = raw.data() + (size_t)y * (stride + 1) + 1;
        for (uint32_t x = 0; x < width; x++) {
            uint32_t r, g, b, a = 255;
            switch (color) {
                case 0: r = g = b = sample(line, x); break;
                case 2: r = sample(line, 3 * (size_t)x); g = sample(line, 3 * (size_t)x + 1); b = sample(line, 3 * (size_t)x + 2); break;
                case 3: {
                    const uint32_t i = sample(line, x);
                    if (3 * (size_t)i + 2 >= palette.size()) { error = "PNG palette index out of range"; return nullptr; }
                    r = palette[3 * i]; g = palette[3 * i + 1]; b = palette[3 * i + 2];
                    if (i < trns.size()) a = trns[i];
                    break;
                }
                case 4: r = g = b = sample(line, 2 * (size_t)x); a = sample(line, 2 * (size_t)x + 1); break;
                default: r = sample(line, 4 * (size_t)x); g = sample(line, 4 * (size_t)x + 1); b = sample(line, 4 * (size_t)x + 2); a = sample(line, 4 * (size_t)x + 3); break;
            }
            image->rgba[(size_t)y * width + x] = Pack(r, g, b, a);
        }
    }
    return image;
}

// netpbm P2/P3 (ASCII) and P5/P6 (binary), maxval up to 65535
std::shared_ptr<Image> DecodePnm(const std::vector<uint8_t> &file, std::string &error) {
    size_t pos = 2;
    auto token = [&](uint32_t &out) -> bool {
        for (;;) {
            while (pos < file.size() && std::isspace(file[pos])) ++pos;
            if (pos < file.size() && file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') ++pos; continue; }
            break;
        }
        if (pos >= file.size() || !std::isdigit(file[pos])) return false;
        uint64_t v = 0;
        while (pos < file.size() && std::isdigit(file[pos])) { v = v * 10 + (file[pos++] - '0'); if (v > 0xffffffffull) return false; }
        out = (uint32_t)v;
        return true;
    };
    const char kind = (char)file[1];
    const int channels = (kind == '3' || kind == '6') ? 3 : 1;
    const bool binary = kind == '5' || kind == '6';
    uint32_t width, height, maxval;
    if (!token(width) || !token(height) || !token(maxval) || width == 0 || height == 0 || maxval == 0 || maxval > 65535) { error = "bad PNM header"; return nullptr; }
    auto image = std::make_shared<Image>();
    image->width = width;
    image->height = height;
    image->rgba.resize((size_t)width * height);
    if (binary) ++pos;      // the single whitespace byte after maxval
    const size_t bytes = maxval > 255 ? 2 : 1;
    for (size_t i = 0; i < (size_t)width * height; i++) {
        uint32_t c[3];
        for (int k = 0; k < channels; k++) {
            uint32_t v;
            if (binary) {
                if (pos + bytes > file.size()) { error = "truncated PNM data"; return nullptr; }
                v = bytes == 2 ? ((uint32_t)file[pos] << 8) | file[pos + 1] : file[pos];
                pos += bytes;
            } else if (!token(v)) {
                error = "truncated PNM data";
                return nullptr;
            }
            c[k] = maxval == 255 ? v : (v * 255u + maxval / 2) / maxval;
            if (c[k] > 255) c[k] = 255;
        }
        image->rgba[i] = channels == 3 ? Pack(c[0], c[1], c[2], 255) : Pack(c[0], c[0], c[0], 255);
    }
    return image;
}

}  // namespace

std::shared_ptr<Image> LoadImage(const std::filesystem::path &path, std::string &error) {
    std::vector<uint8_t> file;
    if (!ReadAll(path, file)) { error = "cannot open '" + path.string() + "'"; return nullptr; }
    static const uint8_t kPngSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (file.size() >= 8 && !std::memcmp(file.data(), kPngSignature, 8)) return DecodePng(file, error);
    if (file.size() >= 3 && file[0] == 'P' && (file[1] == '2' || file[1] == '3' || file[1] == '5' || file[1] == '6')) return DecodePnm(file, error);
    error = "'" + path.string() + "' is neither a PNG nor a PPM/PGM file";
    return nullptr;
}

bool SavePng(const std::filesystem::path &path, const Image &image) {
    std::vector<uint8_t> raw;
    raw.reserve(((size_t)image.width * 4 + 1) * image.height);
    for (uint32_t y = 0; y < image.height; y++) {
        raw.push_back(0);
        const uint8_t *row = reinterpret_cast<const uint8_t *>(&image.rgba[(size_t)y * image.width]);
        raw.insert(raw.end(), row, row + (size_t)image.width * 4);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    auto be32 = [](uint32_t v, uint8_t *o) { o[0] = (uint8_t)(v >> 24); o[1] = (uint8_t)(v >> 16); o[2] = (uint8_t)(v >> 8); o[3] = (uint8_t)v; };
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t head[8];
        be32(len, head);
        std::memcpy(head + 4, type, 4);
        f.write(reinterpret_cast<const char *>(head), 8);
        if (len) f.write(reinterpret_cast<const char *>(data), len);
        uLong crc = crc32(0L, reinterpret_cast<const Bytef *>(type), 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t tail[4];
        be32((uint32_t)crc, tail);
        f.write(reinterpret_cast<const char *>(tail), 4);
    };
    static const uint8_t kPngSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    f.write(reinterpret_cast<const char *>(kPngSignature), 8);
    uint8_t ihdr[13];
    be32(image.width, ihdr);
    be32(image.height, ihdr + 4);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    return (bool)f;
}
