// RAII device allocation on the HIP runtime (the role src/cuda_helpers/buffer.hpp plays in the reference).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <stdexcept>
#include <string>

namespace cpt {

inline void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
#define CPT_HIP(expr) ::cpt::hip_check((expr), #expr)

class HipBuffer {
public:
    HipBuffer() = default;
    explicit HipBuffer(size_t size, const void *data = nullptr, hipStream_t stream = nullptr) { Allocate(size, data, stream); }
    ~HipBuffer() { Release(); }
    HipBuffer(const HipBuffer &) = delete;
    HipBuffer &operator=(const HipBuffer &) = delete;
    HipBuffer(HipBuffer &&o) noexcept : ptr_(o.ptr_), size_(o.size_) { o.ptr_ = nullptr; o.size_ = 0; }
    HipBuffer &operator=(HipBuffer &&o) noexcept {
        if (this != &o) { Release(); ptr_ = o.ptr_; size_ = o.size_; o.ptr_ = nullptr; o.size_ = 0; }
        return *this;
    }

    // grows when needed, never shrinks
    void Allocate(size_t size, const void *data = nullptr, hipStream_t stream = nullptr) {
        if (size > size_) {
            Release();
            CPT_HIP(hipMalloc(&ptr_, size ? size : 1));
            size_ = size;
        }
        if (data && size) SetData(data, size, stream);
    }
    void SetData(const void *data, size_t size, hipStream_t stream = nullptr) {
        // pageable host memory: the copy has left the host buffer when this returns
        CPT_HIP(hipMemcpyAsync(ptr_, data, size, hipMemcpyHostToDevice, stream));
        CPT_HIP(hipStreamSynchronize(stream));
    }
    void Zero(hipStream_t stream = nullptr) { if (size_) CPT_HIP(hipMemsetAsync(ptr_, 0, size_, stream)); }
    void Release() {
        if (ptr_) (void)hipFree(ptr_);
        ptr_ = nullptr;
        size_ = 0;
    }
    void *GpuData() const { return ptr_; }
    template <typename T> T *TypedGpuData() const { return static_cast<T *>(ptr_); }
    size_t Size() const { return size_; }

private:
    void *ptr_ = nullptr;
    size_t size_ = 0;
};

}  // namespace cpt
