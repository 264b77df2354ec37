// Internal helpers shared by the libwagg translation units (gfx950 only; no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/wagg.h"

namespace wagg {

void set_error(const char *fmt, ...);

#define WAGG_HIP(expr)                                                                        \
    do {                                                                                      \
        hipError_t e__ = (expr);                                                              \
        if (e__ != hipSuccess) {                                                              \
            wagg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr,                     \
                            hipGetErrorString(e__));                                          \
            return WAGG_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

#define WAGG_REQUIRE(cond, ...)                                                               \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            wagg::set_error(__VA_ARGS__);                                                     \
            return WAGG_EINVAL;                                                               \
        }                                                                                     \
    } while (0)

// counter hash shared bit for bit with oracle/wagg_oracle.c and oracle/ref_numpy.py
__host__ __device__ inline uint32_t hash32(uint64_t idx, uint32_t seed) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t x = lo ^ (hi * 0x85EBCA6Bu) ^ (seed * 0x9E3779B9u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__host__ __device__ inline float hash_u01(uint64_t idx, uint32_t seed) {
    return (float)(hash32(idx, seed) >> 8) * (1.0f / 16777216.0f);
}

// event ring behind wagg_profile_enable / wagg_profile_read (wagg_util.hip)
void profile_mark(hipStream_t stream, bool begin);

template <typename T>
struct DevBuf {  // owning device buffer, freed in the destructor (plan lifetime)
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) {
        if (p) { (void)hipFree(p); p = nullptr; }
        n = count;
        return hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

}  // namespace wagg
