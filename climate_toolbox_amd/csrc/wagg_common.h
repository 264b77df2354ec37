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

// ---- grid-level transforms evaluated while the data is loaded ---------------------
constexpr int XF_EDD = -1;

// transformations.py:64-87, evaluated in the data type like the reference:
//   tmin < e ? (tmax > e ? ((M - e)(pi/2 - theta) + w cos(theta)) / pi : 0) : M - e
// (a NaN tasmin gives M - e = NaN, a NaN tasmax with tasmin < e gives 0 -- exactly what the two
// nested xr.where calls select)
template <typename T> __device__ __forceinline__ T snyder_edd1(T tmin, T tmax, T e) {
    const T M = (tmax + tmin) / T(2), w = (tmax - tmin) / T(2);
    if (!(tmin < e)) return M - e;
    if (!(tmax > e)) return T(0);
    const T pi = T(3.14159265358979323846);
    // theta = arcsin(z) lies in [-pi/2, pi/2], where cos(theta) = sqrt((1 - z)(1 + z)) exactly
    // (tmin < e < tmax puts z strictly inside (-1, 1)); this spares the general-argument cosine
    const T z = (e - M) / w;
    T theta, c;
    if constexpr (sizeof(T) == 4) { theta = asinf(z); c = sqrtf((T(1) - z) * (T(1) + z)); }
    else { theta = asin(z); c = sqrt((T(1) - z) * (T(1) + z)); }
    return ((M - e) * (pi / T(2) - theta) + w * c) / pi;
}

// fp32: the same function with the slow parts replaced -- v_rcp_f32 / v_sqrt_f32 (1 ulp) instead of the IEEE
// division and square-root sequences, and arcsin by Abramowitz & Stegun 4.4.46,
//   asin|z| = pi/2 - sqrt(1 - |z|) (a0 + a1 |z| + ... + a7 |z|^7),  |error| <= 2e-8,
// all three cases evaluated and selected (no divergent branches): ~35 vector instructions instead of ~100.
// The degree days differ from the libm evaluation by ~1e-7 relative (tolerance of the fp32 path: 1e-4);
// NaN / inf behave as in the generic form (a NaN tasmin gives NaN, a NaN tasmax below the threshold 0).
template <> __device__ __forceinline__ float snyder_edd1<float>(float tmin, float tmax, float e) {
    const float M = 0.5f * (tmax + tmin), w = 0.5f * (tmax - tmin);
    const float d = M - e;
    const float z = -d * __builtin_amdgcn_rcpf(w);                     // (e - M) / w, strictly inside (-1, 1) in the band
    const float az = __builtin_fabsf(z);
    float p = -0.0012624911f;
    p = __builtin_fmaf(p, az, 0.0066700901f);
    p = __builtin_fmaf(p, az, -0.0170881256f);
    p = __builtin_fmaf(p, az, 0.0308918810f);
    p = __builtin_fmaf(p, az, -0.0501743046f);
    p = __builtin_fmaf(p, az, 0.0889789874f);
    p = __builtin_fmaf(p, az, -0.2145988016f);
    p = __builtin_fmaf(p, az, 1.5707963050f);
    const float acos_az = __builtin_amdgcn_sqrtf(__builtin_fmaxf(1.0f - az, 0.0f)) * p;   // pi/2 - asin|z|
    const float pi = 3.14159265358979323846f;
    const float quarter = z >= 0.0f ? acos_az : pi - acos_az;         // pi/2 - theta
    const float c = __builtin_amdgcn_sqrtf(__builtin_fmaxf((1.0f - z) * (1.0f + z), 0.0f));   // cos(theta)
    const float inner = (d * quarter + w * c) * (1.0f / pi);
    return !(tmin < e) ? d : (!(tmax > e) ? 0.0f : inner);
}

template <typename T> __device__ __forceinline__ T xform1(T x, T off, int pw) {
    const T y = x + off;
    T r = y;
    for (int i = 1; i < pw; ++i) r *= y;
    return r;
}
template <typename V, typename T> __device__ __forceinline__ V xform4(V v, T off, int pw) {
    V r;
#pragma unroll
    for (int c = 0; c < 4; ++c) r[c] = xform1<T>(v[c], off, pw);
    return r;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per apply
hipError_t allow_dynamic_lds(const void *kern, size_t bytes);

// event ring behind wagg_profile_enable / wagg_profile_read (wagg_util.hip)
void profile_mark(hipStream_t stream, bool begin);

// ---- host-resident (time, gridcell) data: row-block pipeline ----------------------------
// X_host is cut into blocks of whole rows; the H2D copy of block i+1 (copy stream) overlaps the kernels
// of block i (compute stream), results return block by block, and the device holds two blocks instead of
// the whole field.  With WAGG_HOST_PIN the caller's arrays are page-locked in place for the duration of
// the call (hipHostRegister), which makes the copies truly asynchronous; pageable arrays are staged by
// the runtime and overlap only partly.  apply(X_dev, rows, out_dev, stream) launches one block.
// elements from the first element of row 0 to the last of row rows-1 (the last row of a pitched host array need
// not be followed by its padding: never touch more than this)
inline size_t host_span(int64_t rows, int64_t ld, int64_t cols) { return rows > 0 ? (size_t)((rows - 1) * ld + cols) : 0; }

// Small host buffers (below 32 MiB) never reach an asynchronous runtime copy as they are: the runtime would page-lock
// them on the fly and drop that pin later on its own schedule, possibly after the caller has freed the array (small
// arrays are the ones that die right after the call).  They go through the library's own pair of page-locked staging
// buffers instead (wagg_util.hip; blocking, ~10 GB/s -- irrelevant at these sizes).  Both return when the user memory
// is no longer needed (h2d) / completely written (d2h).
constexpr size_t HOST_STAGE_MAX = (size_t)32 << 20;
hipError_t staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
hipError_t staged_d2h_rows(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, hipStream_t st);

// blocking host -> device copy (small buffers through the staging pair, see above)
inline hipError_t copy_to_device(void *dst_dev, const void *src_host, size_t bytes) {
    if (bytes == 0) return hipSuccess;
    if (bytes < HOST_STAGE_MAX) return staged_h2d(dst_dev, src_host, bytes, nullptr);
    return hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice);
}

// device (rows x ld, same pitch) -> pitched host array: only the `cols` used elements of every row are written, the
// caller's padding between rows is left alone
template <typename T>
inline hipError_t copy_rows_to_host(T *dst_host, const T *src_dev, int64_t rows, int64_t ld, int64_t cols, hipStream_t st,
                                    bool async) {
    if (rows <= 0 || cols <= 0) return hipSuccess;
    if (!async && sizeof(T) * host_span(rows, ld, cols) < HOST_STAGE_MAX)
        return staged_d2h_rows(dst_host, src_dev, rows, sizeof(T) * (size_t)ld, sizeof(T) * (size_t)cols, st);
    if (ld == cols)
        return async ? hipMemcpyAsync(dst_host, src_dev, sizeof(T) * (size_t)(rows * cols), hipMemcpyDeviceToHost, st)
                     : hipMemcpy(dst_host, src_dev, sizeof(T) * (size_t)(rows * cols), hipMemcpyDeviceToHost);
    return async ? hipMemcpy2DAsync(dst_host, sizeof(T) * (size_t)ld, src_dev, sizeof(T) * (size_t)ld, sizeof(T) * (size_t)cols,
                                    (size_t)rows, hipMemcpyDeviceToHost, st)
                 : hipMemcpy2D(dst_host, sizeof(T) * (size_t)ld, src_dev, sizeof(T) * (size_t)ld, sizeof(T) * (size_t)cols,
                               (size_t)rows, hipMemcpyDeviceToHost);
}

template <typename T, typename ApplyFn>
int stream_host_rows(const T *X_host, int64_t Tn, int64_t ldx, int64_t G, T *out_host, int64_t ldo, int64_t R, int flags,
                     int64_t quantum, ApplyFn apply) {
    if (Tn == 0) return WAGG_OK;
    // ~256 MiB of X per block in whole multiples of `quantum` rows (the row count one launch handles well:
    // 64 for the segment-table kernels, a full 368 / 176-row block for the MFMA forms, whose W is streamed
    // once per launch), at least two blocks when there are >= 2 quanta of rows
    int64_t B = ((int64_t)256 << 20) / (int64_t)(ldx * sizeof(T));
    B = B < quantum ? quantum : B / quantum * quantum;
    if (Tn >= 2 * quantum && B > (Tn + 1) / 2) B = ((Tn + 1) / 2 + quantum - 1) / quantum * quantum;
    if (B > Tn) B = Tn;
    const int64_t nb = (Tn + B - 1) / B;
    const size_t xbytes = sizeof(T) * host_span(Tn, ldx, G), obytes = sizeof(T) * host_span(Tn, ldo, R);
    bool pin_x = false, pin_o = false;
    if (flags & WAGG_HOST_PIN) {
        // Registration works on whole pages and costs ~0.1 ms per MiB.  Below 32 MiB (glibc's largest mmap threshold)
        // an array may live in the brk heap and share its first and last page with unrelated heap objects, which would
        // then be page-locked and GPU-mapped along with it: such buffers are staged instead (they are small anyway).
        constexpr size_t PIN_MIN = (size_t)32 << 20;
        pin_x = xbytes >= PIN_MIN && hipHostRegister(const_cast<T *>(X_host), xbytes, hipHostRegisterDefault) == hipSuccess;
        pin_o = obytes >= PIN_MIN && hipHostRegister(out_host, obytes, hipHostRegisterDefault) == hipSuccess;
        (void)hipGetLastError();            // a failed registration is not an error: the copies are staged instead
    }
    struct Guard {              // everything acquired here is released on every exit path
        const void *rx = nullptr; void *ro = nullptr;
        hipStream_t sc = nullptr, sk = nullptr;
        hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
        void *dx[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
        ~Guard() {
            if (sc) (void)hipStreamSynchronize(sc);
            if (sk) (void)hipStreamSynchronize(sk);
            for (int b = 0; b < 2; ++b) {
                if (ready[b]) (void)hipEventDestroy(ready[b]);
                if (done[b]) (void)hipEventDestroy(done[b]);
                if (dx[b]) (void)hipFree(dx[b]);
                if (dout[b]) (void)hipFree(dout[b]);
            }
            if (sc) (void)hipStreamDestroy(sc);
            if (sk) (void)hipStreamDestroy(sk);
            if (rx) (void)hipHostUnregister(const_cast<void *>(rx));
            if (ro) (void)hipHostUnregister(ro);
        }
    } g;
    if (pin_x) g.rx = X_host;
    if (pin_o) g.ro = out_host;
    WAGG_HIP(hipStreamCreateWithFlags(&g.sc, hipStreamNonBlocking));
    WAGG_HIP(hipStreamCreateWithFlags(&g.sk, hipStreamNonBlocking));
    for (int b = 0; b < 2 && b < nb; ++b) {
        WAGG_HIP(hipEventCreateWithFlags(&g.ready[b], hipEventDisableTiming));
        WAGG_HIP(hipEventCreateWithFlags(&g.done[b], hipEventDisableTiming));
        WAGG_HIP(hipMalloc(&g.dx[b], sizeof(T) * (size_t)(B * ldx)));
        WAGG_HIP(hipMalloc(&g.dout[b], sizeof(T) * (size_t)(B * ldo)));
    }
    for (int64_t i = 0; i < nb; ++i) {
        const int b = (int)(i & 1);
        const int64_t r0 = i * B, rows = Tn - r0 < B ? Tn - r0 : B;
        if (i >= 2) WAGG_HIP(hipStreamWaitEvent(g.sc, g.done[b], 0));       // block i-2 no longer uses this buffer
        if (!pin_x && xbytes < HOST_STAGE_MAX)
            WAGG_HIP(staged_h2d(g.dx[b], X_host + r0 * ldx, sizeof(T) * host_span(rows, ldx, G), g.sc));
        else
            WAGG_HIP(hipMemcpyAsync(g.dx[b], X_host + r0 * ldx, sizeof(T) * host_span(rows, ldx, G), hipMemcpyHostToDevice, g.sc));
        WAGG_HIP(hipEventRecord(g.ready[b], g.sc));
        WAGG_HIP(hipStreamWaitEvent(g.sk, g.ready[b], 0));
        const int rc = apply(static_cast<const T *>(g.dx[b]), rows, static_cast<T *>(g.dout[b]), g.sk);
        if (rc != WAGG_OK) return rc;
        if (!pin_o && obytes < HOST_STAGE_MAX)
            WAGG_HIP(staged_d2h_rows(out_host + r0 * ldo, g.dout[b], rows, sizeof(T) * (size_t)ldo, sizeof(T) * (size_t)R, g.sk));
        else
            WAGG_HIP(copy_rows_to_host<T>(out_host + r0 * ldo, static_cast<const T *>(g.dout[b]), rows, ldo, R, g.sk, true));
        WAGG_HIP(hipEventRecord(g.done[b], g.sk));
    }
    WAGG_HIP(hipStreamSynchronize(g.sc));
    WAGG_HIP(hipStreamSynchronize(g.sk));
    return WAGG_OK;
}

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
