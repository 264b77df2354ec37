// Internal helpers shared by the libwagg translation units (gfx950 only; no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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

// fp32: the same function in a cheaper form.  With z = (e - M) / w in (-1, 1), theta = asin z and d = M - e = -z w the
// band value is  w (sqrt(1 - z^2) - z acos z) / pi =: w f(z),  f(-z) = f(z) + z,  and on [0, 1]
//   f(x) = (1 - x)^(3/2) P(x),  P smooth between 1/pi (x = 0) and 2 sqrt(2) / (3 pi) (x = 1):
// a degree-6 minimax fit of P (|error| <= 5e-9, tools/fit_edd_poly.py) leaves ONE v_sqrt_f32, v_rcp_f32 and ~15 vector
// instructions per value (round 2: Abramowitz & Stegun 4.4.46 for the arcsine, two square roots, ~27).  With |z| clamped to 1
// the expression  w s t P(|z|) + max(d, 0)  (t = 1 - |z|, s = sqrt t) also gives the two outer cases by itself for finite
// data: M - e where tasmin >= e (z <= -1) and 0 where tasmax <= e (z >= 1); w = 0 is covered by the clamp (v_min returns
// the number when z is 0 * inf).  snyder_edd1_finite is that expression alone; snyder_edd1<float> adds the reference's two
// selections, which also decide what a NaN in either field gives (a NaN tasmin NaN, a NaN tasmax below the threshold 0).
// The degree days differ from the fp64 libm evaluation by <= 2e-6 absolute (tolerance of the fp32 path: 1e-4 relative).
__device__ __forceinline__ float snyder_edd1_finite(float tmin, float tmax, float e) {
    const float M = 0.5f * (tmax + tmin), w = 0.5f * (tmax - tmin);
    const float d = M - e;
    const float z = -d * __builtin_amdgcn_rcpf(w);                     // (e - M) / w
    const float az = __builtin_fminf(__builtin_fabsf(z), 1.0f);
    const float t = 1.0f - az;
    float p = 7.577220821985975e-05f;
    p = __builtin_fmaf(p, az, -0.0003951705584768206f);
    p = __builtin_fmaf(p, az, 0.0010792884277179837f);
    p = __builtin_fmaf(p, az, -0.002406827174127102f);
    p = __builtin_fmaf(p, az, 0.005977150052785873f);
    p = __builtin_fmaf(p, az, -0.022534651681780815f);
    p = __builtin_fmaf(p, az, 0.31830987334251404f);
    const float q = __builtin_amdgcn_sqrtf(t) * t;                     // (1 - |z|)^(3/2)
    return __builtin_fmaf(w, q * p, __builtin_fmaxf(d, 0.0f));
}
// fp64: the same expression with a degree-13 fit of P (|error| <= 4.4e-15, tools/fit_edd_poly.py): a reciprocal, a reciprocal
// square root and 13 fused multiply-adds where asin + sqrt of libm take several hundred instructions; used by the
// loader/consumer kernel for finite fields (everything else keeps the libm form above, from which it differs by ~1e-14).
__device__ __forceinline__ double snyder_edd1_finite(double tmin, double tmax, double e) {
    const double M = 0.5 * (tmax + tmin), w = 0.5 * (tmax - tmin);
    const double d = M - e;
    // 1 / w and sqrt(t) from v_rcp_f64 / v_rsq_f64 with one Newton step each (~2^-52 relative; the IEEE division and
    // square-root sequences are a third of this function's instructions)
    double rw = __builtin_amdgcn_rcp(w);
    rw = __builtin_fma(rw, __builtin_fma(-w, rw, 1.0), rw);
    const double az = __builtin_fmin(__builtin_fabs(d * rw), 1.0);     // |z|, z = (e - M) / w (w = 0: inf or NaN -> 1)
    const double t = 1.0 - az;
    const double ty = __builtin_amdgcn_rsq(__builtin_fmax(t, 1e-300));  // (t = 0: 0 * 1e150 = 0 below)
    double st = t * ty;                                                 // sqrt(t) to ~2^-26 ...
    st = __builtin_fma(__builtin_fma(-st, st, t), 0.5 * ty, st);        // ... and one step: s + (t - s^2) / (2 s)
    double p = -8.644295123170258e-07;
    p = __builtin_fma(p, az, 7.147532263874047e-06);
    p = __builtin_fma(p, az, -2.807399717529102e-05);
    p = __builtin_fma(p, az, 7.126284961634285e-05);
    p = __builtin_fma(p, az, -0.0001359898123558899);
    p = __builtin_fma(p, az, 0.0002182750865254441);
    p = __builtin_fma(p, az, -0.0003242640574299061);
    p = __builtin_fma(p, az, 0.000480605991421148);
    p = __builtin_fma(p, az, -0.0007477719612718598);
    p = __builtin_fma(p, az, 0.0012691227365052833);
    p = __builtin_fma(p, az, -0.0024647062510063378);
    p = __builtin_fma(p, az, 0.005985979570242935);
    p = __builtin_fma(p, az, -0.0225351707225776);
    p = __builtin_fma(p, az, 0.3183098861837864);
    return __builtin_fma(w, st * t * p, __builtin_fmax(d, 0.0));
}
template <> __device__ __forceinline__ float snyder_edd1<float>(float tmin, float tmax, float e) {
    const float d = 0.5f * (tmax + tmin) - e;
    const float inner = snyder_edd1_finite(tmin, tmax, e);
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

// a struct that crosses the C boundary by layout, into a caller's buffer of `dst_size` bytes: min(dst_size, src_size) bytes are
// written, what the caller has beyond the library's size is zeroed (wagg_desc.hip; include/wagg.h "structs that cross ...")
void copy_sized(void *dst, uint64_t dst_size, const void *src, size_t src_size);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device) instead of once per apply
hipError_t allow_dynamic_lds(const void *kern, size_t bytes);

// event ring behind wagg_profile_enable / wagg_profile_read (wagg_util.hip).  profile_slot hands out the next pair of the
// ring (false: profiling is off or the ring is full); launch_timed / launch_timed_ptr give that pair to hipExtLaunchKernel,
// so the two events stamp the START and the END OF THE DISPATCH ITSELF.  (Rounds 1-4 recorded an event on either side of the
// launch call: whatever the host did between the first record and the dispatch -- a page fault, a descheduled thread -- was
// booked as kernel time.)
bool profile_slot(hipEvent_t *start, hipEvent_t *stop);

inline hipError_t launch_timed_ptr(bool timed, const void *kern, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t st) {
    hipEvent_t a = nullptr, b = nullptr;
    if (timed && profile_slot(&a, &b)) return hipExtLaunchKernel(kern, grid, block, args, shmem, st, a, b, 0);
    return hipLaunchKernel(kern, grid, block, args, shmem, st);
}
template <typename... Args, typename F = void (*)(Args...)>
inline void launch_timed(bool timed, F kern, dim3 grid, dim3 block, size_t shmem, hipStream_t st, Args... args) {
    hipEvent_t a = nullptr, b = nullptr;
    if (timed && profile_slot(&a, &b)) hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)shmem, st, a, b, 0u, args...);
    else hipLaunchKernelGGL(kern, grid, block, shmem, st, args...);
}

// host <-> device copies of library- or caller-owned pageable memory never go through a runtime copy of the pageable
// pointer (wagg_host.h says why): small ones pass through the library's page-locked staging pieces
hipError_t staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
hipError_t staged_d2h_rows(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, hipStream_t st);
inline hipError_t staged_d2h(void *dst_host, const void *src_dev, size_t bytes, hipStream_t st = nullptr) {
    return staged_d2h_rows(dst_host, src_dev, 1, bytes, bytes, st);
}
void clear_error();

// wagg_scratch.hip: device blocks and streams needed for the length of one call come from (and return to) a small pool --
// see there for why hipMalloc / hipFree per call is not an option.  A block / stream is returned only when nothing on the
// device can still be using it.
hipError_t scratch_alloc(void **p, size_t bytes);        // on the current device
void scratch_free(void *p, bool keep = true);             // keep = false: hipFree now (with its implicit wait for the device)
// an idle non-blocking stream of the current device.  `role`: what the taker will use it for -- 0 kernels / anything, 1 host ->
// device copies, 2 device -> host copies: a stream comes back with the role it had, and a taker gets a stream of its own role
// (so a copy-in stream never carries kernels or copies out in another call)
hipError_t scratch_stream(hipStream_t *st, int role = 0);
void scratch_stream_done(hipStream_t st, int role = 0, bool keep = true);    // keep = false: destroy it
void release_scratch(int what = 7);                                  // everything kept goes back to the driver
template <typename T>
struct ScratchBuf {                                      // RAII over scratch_alloc (call-lifetime buffers)
    T *p = nullptr;
    ScratchBuf() = default;
    ScratchBuf(const ScratchBuf &) = delete;
    ScratchBuf &operator=(const ScratchBuf &) = delete;
    hipError_t alloc(size_t count) { return scratch_alloc(reinterpret_cast<void **>(&p), (count ? count : 1) * sizeof(T)); }
    // (the block may reach its next taker at once, so whatever was queued on it -- also on an error path that returned
    //  without waiting -- is drained first; hipFree used to do that implicitly)
    ~ScratchBuf() { if (p) { (void)hipDeviceSynchronize(); scratch_free(p); } }
};

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
        hipError_t e = hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
        if (e == hipErrorOutOfMemory) {          // what the scratch pool keeps for later calls (wagg_scratch.hip) goes first
            (void)hipGetLastError();
            release_scratch();
            e = hipMalloc((void **)&p, (count ? count : 1) * sizeof(T));
        }
        return e;
    }
    hipError_t upload(const std::vector<T> &h, hipStream_t st = nullptr) {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return staged_h2d(p, h.data(), h.size() * sizeof(T), st);
    }
};

}  // namespace wagg
