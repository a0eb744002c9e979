#include "film.hpp"

#include <cstdio>
#include <cstring>
#include <string>

Film::Film(uint32_t width, uint32_t height) { Init(width, height); }

void Film::Init(uint32_t width, uint32_t height) {
    width_ = width;
    height_ = height;
    allocated_ = false;      // device memory is taken on first use, so a Film can exist without a GPU
    zero_generation_++;
    mapped_since_zeroed_ = false;
}

void Film::Resize(uint32_t width, uint32_t height) {
    if (width_ != width || height_ != height) Init(width, height);
}

float *Film::DeviceMap(hipStream_t stream) {
    if (!allocated_) {
        buffer_.Allocate((size_t)width_ * height_ * 16);
        // the running mean multiplies the previous value by 0 at the first sample: a NaN left in
        // fresh memory would survive that, so the buffer starts from zeros.  The clear is finished
        // before this returns, whichever stream the caller renders on.
        CPT_HIP(hipMemsetAsync(buffer_.GpuData(), 0, (size_t)width_ * height_ * 16, stream));
        CPT_HIP(hipStreamSynchronize(stream));
        allocated_ = true;
    }
    mapped_since_zeroed_ = true;
    return buffer_.TypedGpuData<float>();
}

void Film::Clear(hipStream_t stream) {
    float *p = DeviceMap(stream);
    CPT_HIP(hipMemsetAsync(p, 0, (size_t)width_ * height_ * 16, stream));
    zero_generation_++;
    mapped_since_zeroed_ = false;
}

void Film::ReadBack(float *dst, hipStream_t stream) {
    float *p = DeviceMap(stream);
    CPT_HIP(hipMemcpyAsync(dst, p, (size_t)width_ * height_ * 16, hipMemcpyDeviceToHost, stream));
    CPT_HIP(hipStreamSynchronize(stream));
}

bool Film::SaveTo(const char *path) {
    std::vector<float> host((size_t)width_ * height_ * 4);
    ReadBack(host.data());
    return WriteExrRgba(path, host.data(), width_, height_);
}

namespace {
void Put32(std::string &s, uint32_t v) { s.append(reinterpret_cast<const char *>(&v), 4); }
void PutStr(std::string &s, const char *z) { s.append(z, std::strlen(z) + 1); }
void Attr(std::string &s, const char *name, const char *type, const std::string &value) {
    PutStr(s, name);
    PutStr(s, type);
    Put32(s, (uint32_t)value.size());
    s += value;
}
}  // namespace

// OpenEXR 2.0 single-part scanline file, NO_COMPRESSION, four FLOAT channels.  Channels are
// stored in alphabetical order (A, B, G, R), which is also what tinyexr's SaveEXR emits for
// 4-component input (thirdparty/tinyexr/tinyexr.h:8438-8475).
bool WriteExrRgba(const char *path, const float *rgba, uint32_t width, uint32_t height) {
    if (width == 0 || height == 0) return false;
    std::string hdr;
    Put32(hdr, 20000630u);     // magic
    Put32(hdr, 2u);            // version 2, no flags
    {
        std::string ch;
        const char *names[4] = {"A", "B", "G", "R"};
        for (auto n : names) {
            PutStr(ch, n);
            Put32(ch, 2u);                         // FLOAT
            ch.append("\0\0\0\0", 4);              // pLinear + reserved
            Put32(ch, 1u); Put32(ch, 1u);          // sampling
        }
        ch.push_back('\0');
        Attr(hdr, "channels", "chlist", ch);
    }
    Attr(hdr, "compression", "compression", std::string(1, '\0'));
    {
        std::string box;
        Put32(box, 0u); Put32(box, 0u); Put32(box, width - 1); Put32(box, height - 1);
        Attr(hdr, "dataWindow", "box2i", box);
        Attr(hdr, "displayWindow", "box2i", box);
    }
    Attr(hdr, "lineOrder", "lineOrder", std::string(1, '\0'));     // increasing y
    { float one = 1.0f; Attr(hdr, "pixelAspectRatio", "float", std::string(reinterpret_cast<char *>(&one), 4)); }
    { float z[2] = {0.0f, 0.0f}; Attr(hdr, "screenWindowCenter", "v2f", std::string(reinterpret_cast<char *>(z), 8)); }
    { float one = 1.0f; Attr(hdr, "screenWindowWidth", "float", std::string(reinterpret_cast<char *>(&one), 4)); }
    hdr.push_back('\0');

    FILE *f = std::fopen(path, "wb");
    if (!f) return false;
    const uint64_t line_bytes = (uint64_t)width * 16;
    const uint64_t table_start = hdr.size();
    const uint64_t data_start = table_start + (uint64_t)height * 8;
    bool ok = std::fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
    for (uint32_t y = 0; ok && y < height; y++) {
        uint64_t off = data_start + (uint64_t)y * (8 + line_bytes);
        ok = std::fwrite(&off, 8, 1, f) == 1;
    }
    std::vector<float> line((size_t)width * 4);
    for (uint32_t y = 0; ok && y < height; y++) {
        const float *src = rgba + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) {
            line[x] = src[4 * x + 3];                          // A
            line[(size_t)width + x] = src[4 * x + 2];          // B
            line[(size_t)2 * width + x] = src[4 * x + 1];      // G
            line[(size_t)3 * width + x] = src[4 * x + 0];      // R
        }
        int32_t yy = (int32_t)y;
        uint32_t size = (uint32_t)line_bytes;
        ok = std::fwrite(&yy, 4, 1, f) == 1 && std::fwrite(&size, 4, 1, f) == 1 && std::fwrite(line.data(), 1, line_bytes, f) == line_bytes;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok;
}
