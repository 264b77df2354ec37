// Process-level entry points, the materialised gather (aggregations.py:27) and the synthetic
// field generator shared with the oracle.
#include "wagg_common.h"

namespace wagg {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct ProfRing {
    bool on = false;
    int count = 0;                      // pairs begun
    hipEvent_t a[WAGG_PROFILE_SLOTS], b[WAGG_PROFILE_SLOTS];
    bool created = false;
};
static ProfRing g_prof;

void profile_mark(hipStream_t stream, bool begin) {
    if (!g_prof.on) return;
    if (begin) {
        if (g_prof.count >= WAGG_PROFILE_SLOTS) return;
        (void)hipEventRecord(g_prof.a[g_prof.count], stream);
    } else {
        if (g_prof.count >= WAGG_PROFILE_SLOTS) return;
        (void)hipEventRecord(g_prof.b[g_prof.count], stream);
        ++g_prof.count;
    }
}

// out[t, i] = X[t, cell_idx[i]]; one thread per (segment, timestep) with the contiguous axis of
// the OUTPUT on the lanes so stores coalesce.
template <typename T>
__global__ void gather_kernel(const T *__restrict__ X, int64_t Ttot, int64_t ldx, int layout,
                              const int32_t *__restrict__ cell_idx, int64_t nseg,
                              T *__restrict__ out, int64_t ldo, int out_layout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * Ttot) return;
    int64_t t, s;
    if (out_layout == WAGG_OUT_TR) { t = i / nseg; s = i % nseg; }
    else { s = i / Ttot; t = i % Ttot; }
    const int64_t g = cell_idx[s];
    const T v = layout == WAGG_LAYOUT_TG ? X[t * ldx + g] : X[g * ldx + t];
    if (out_layout == WAGG_OUT_TR) out[t * ldo + s] = v;
    else out[s * ldo + t] = v;
}

template <typename T>
static int gather(const T *X, int64_t Ttot, int64_t ldx, int layout, const int32_t *cell_idx,
                  int64_t nseg, T *out, int64_t ldo, int out_layout, void *stream) {
    WAGG_REQUIRE(Ttot >= 0 && nseg >= 0, "negative size");
    WAGG_REQUIRE(layout == WAGG_LAYOUT_TG || layout == WAGG_LAYOUT_GT, "bad layout");
    WAGG_REQUIRE(out_layout == WAGG_OUT_TR || out_layout == WAGG_OUT_RT, "bad out_layout");
    const int64_t n = Ttot * nseg;
    if (n == 0) return WAGG_OK;
    WAGG_REQUIRE(X && cell_idx && out, "NULL pointer");
    WAGG_REQUIRE(ldo >= (out_layout == WAGG_OUT_TR ? nseg : Ttot), "ldo too small");
    WAGG_REQUIRE((n + 255) / 256 < (int64_t)0x7fffffff, "gather too large");
    hipLaunchKernelGGL((gather_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, X, Ttot, ldx, layout, cell_idx, nseg, out, ldo, out_layout);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}

// separately rounded multiply and add (no FMA contraction) so that the field equals the
// oracle's numpy expression bit for bit
template <typename T> __device__ __forceinline__ T mul_add_rn(T b, T a, T h) {
#pragma clang fp contract(off)
    const T p = a * h;
    return b + p;
}

template <typename T>
__global__ void synth_field_kernel(T *__restrict__ X, int64_t Ttot, int64_t G, int64_t ldx,
                                   uint32_t seed, T base, T amp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < Ttot * G; i += stride) {
        const int64_t t = i / G, g = i % G;
        X[t * ldx + g] = mul_add_rn(base, amp, (T)hash_u01((uint64_t)i, seed) - (T)0.5);
    }
}

template <typename T>
static int synth_field(T *X, int64_t Ttot, int64_t G, int64_t ldx, uint32_t seed, T base, T amp,
                       void *stream) {
    WAGG_REQUIRE(Ttot >= 0 && G >= 0 && ldx >= G, "bad sizes");
    if (Ttot * G == 0) return WAGG_OK;
    WAGG_REQUIRE(X != nullptr, "X is NULL");
    hipLaunchKernelGGL((synth_field_kernel<T>), dim3(256 * 16), dim3(256), 0, (hipStream_t)stream, X,
                       Ttot, G, ldx, seed, base, amp);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}

}  // namespace wagg

extern "C" int wagg_version(void) { return 10000 * 0 + 100 * 1 + 0; }

extern "C" int wagg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" const char *wagg_last_error(void) { return wagg::g_err; }

extern "C" int wagg_gather_f32(const float *X, int64_t T, int64_t ldx, int layout,
                               const int32_t *cell_idx, int64_t nseg, float *out, int64_t ldo,
                               int out_layout, void *stream) {
    return wagg::gather<float>(X, T, ldx, layout, cell_idx, nseg, out, ldo, out_layout, stream);
}
extern "C" int wagg_gather_f64(const double *X, int64_t T, int64_t ldx, int layout,
                               const int32_t *cell_idx, int64_t nseg, double *out, int64_t ldo,
                               int out_layout, void *stream) {
    return wagg::gather<double>(X, T, ldx, layout, cell_idx, nseg, out, ldo, out_layout, stream);
}
extern "C" int wagg_synth_field_f32(float *X, int64_t T, int64_t G, int64_t ldx, uint32_t seed,
                                    float base, float amp, void *stream) {
    return wagg::synth_field<float>(X, T, G, ldx, seed, base, amp, stream);
}
extern "C" int wagg_synth_field_f64(double *X, int64_t T, int64_t G, int64_t ldx, uint32_t seed,
                                    double base, double amp, void *stream) {
    return wagg::synth_field<double>(X, T, G, ldx, seed, base, amp, stream);
}

extern "C" int wagg_profile_enable(int on) {
    using namespace wagg;
    if (on && !g_prof.created) {
        for (int i = 0; i < WAGG_PROFILE_SLOTS; ++i) {
            WAGG_HIP(hipEventCreate(&g_prof.a[i]));
            WAGG_HIP(hipEventCreate(&g_prof.b[i]));
        }
        g_prof.created = true;
    }
    g_prof.on = on != 0;
    g_prof.count = 0;
    return WAGG_OK;
}

extern "C" int wagg_profile_read(float *ms_out, int max_out, int *n_out) {
    using namespace wagg;
    WAGG_REQUIRE(n_out != nullptr && (ms_out != nullptr || max_out == 0), "NULL argument");
    int n = g_prof.count < max_out ? g_prof.count : max_out;
    for (int i = 0; i < n; ++i) {
        WAGG_HIP(hipEventSynchronize(g_prof.b[i]));
        WAGG_HIP(hipEventElapsedTime(&ms_out[i], g_prof.a[i], g_prof.b[i]));
    }
    *n_out = n;
    return WAGG_OK;
}
