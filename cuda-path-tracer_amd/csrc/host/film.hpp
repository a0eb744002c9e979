// Accumulation buffer of the renderer (src/pathtracer/film.hpp:6-30): W*H float4 in device
// memory, row 0 = top of the image.  The reference backs it with an OpenGL pixel-unpack buffer
// mapped into CUDA; here it is a plain HIP allocation and there is no GL texture.
#pragma once
#include <cstdint>
#include <vector>

#include "buffer.hpp"

class Film {
public:
    Film(uint32_t width, uint32_t height);

    uint32_t Width() const { return width_; }
    uint32_t Height() const { return height_; }
    void Resize(uint32_t width, uint32_t height);

    // Device pointer to the accumulation buffer (the reference's CudaMap / CudaUnmap pair).
    // `stream`: the stream whose kernels will write the buffer; the first-use clear is ordered on it
    float *DeviceMap(hipStream_t stream = nullptr);
    void DeviceUnmap() {}
    void Clear(hipStream_t stream = nullptr);
    // Counts the times the buffer was made all zeros (created, resized, cleared), and whether its pointer has been handed
    // out since the last of them: what a renderer needs to know that the content is still what it left there.
    uint64_t ZeroGeneration() const { return zero_generation_; }
    bool MappedSinceZeroed() const { return mapped_since_zeroed_; }

    // host copy of the buffer, W*H*4 floats
    void ReadBack(float *dst, hipStream_t stream = nullptr);
    // fp32 RGBA OpenEXR, scanline, channels stored as A,B,G,R like tinyexr's SaveEXR (film.cpp:61-68)
    bool SaveTo(const char *path);

private:
    void Init(uint32_t width, uint32_t height);
    uint32_t width_ = 0, height_ = 0;
    bool allocated_ = false;
    uint64_t zero_generation_ = 0;
    bool mapped_since_zeroed_ = false;
    cpt::HipBuffer buffer_;
};

// Writes W*H RGBA floats (row 0 = top) as an uncompressed scanline OpenEXR file.
bool WriteExrRgba(const char *path, const float *rgba, uint32_t width, uint32_t height);
