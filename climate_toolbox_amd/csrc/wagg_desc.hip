// wagg_apply(): the one entry point of every apply (include/wagg.h, "one entry point for every apply"), the exported
// wagg_*apply* symbols as wrappers that fill a descriptor for it, and the size / ordinal self-description of the structs that
// cross the boundary by layout.
//
// wagg_apply copies min(desc->struct_size, sizeof) bytes of the caller's descriptor into a zeroed one of the library's own
// layout (a caller built against an older header leaves the newer fields 0 = "not given"), checks that the combination of
// plan kind, element type, source and transform is one the library has a kernel path for -- anything else is
// WAGG_EUNSUPPORTED with a message that names the combination, nothing is emulated -- and calls the typed entry
// (wagg_entry.h), which validates shapes and pointers as it always did.
#include <cstddef>
#include <cstring>
#include <type_traits>

#include "wagg_common.h"
#include "wagg_dense_int.h"
#include "wagg_entry.h"
#include "wagg_host.h"
#include "wagg_sparse_int.h"

namespace wagg {
namespace {

const char *src_name(int s) {
    switch (s) {
        case WAGG_SRC_DEVICE: return "device";
        case WAGG_SRC_HOST: return "host";
        case WAGG_SRC_HOST_MULTI: return "host-multi";
        case WAGG_SRC_SHARDED: return "sharded";
        default: return "?";
    }
}
const char *xf_name(int x) { return x == WAGG_XF_NONE ? "none" : x == WAGG_XF_POLY ? "poly" : x == WAGG_XF_EDD ? "edd" : "?"; }

int unsupported(const wagg_apply_desc &d, const char *why) {
    set_error("wagg_apply: no kernel path for %s plan / %s / source %s / transform %s: %s",
              d.plan_kind == WAGG_PLAN_DENSE ? "dense-family" : "segment-table", d.elem == WAGG_T_F64 ? "f64" : "f32", src_name(d.source),
              xf_name(d.transform), why);
    return WAGG_EUNSUPPORTED;
}

template <typename T>
int run_segment(const wagg_apply_desc &d) {
    constexpr bool f32 = sizeof(T) == 4;
    const T *x = static_cast<const T *>(d.x), *x2 = static_cast<const T *>(d.x2);
    T *out = static_cast<T *>(d.out);
    const wagg_plan *plan = static_cast<const wagg_plan *>(d.plan);
    switch (d.source) {
        case WAGG_SRC_DEVICE:
            if (d.transform == WAGG_XF_NONE) {
                if constexpr (f32) return entry::apply_f32(plan, x, d.T, d.ldx, d.layout, out, d.ldo, d.out_layout, d.stream);
                else return entry::apply_f64(plan, x, d.T, d.ldx, d.layout, out, d.ldo, d.out_layout, d.stream);
            }
            if (d.transform == WAGG_XF_POLY) {
                if constexpr (f32) return entry::apply_poly_f32(plan, x, d.T, d.ldx, d.layout, d.offset, d.pow_first, d.n_pow, out, d.ldo, d.out_pstride, d.out_layout, d.stream);
                else return entry::apply_poly_f64(plan, x, d.T, d.ldx, d.layout, d.offset, d.pow_first, d.n_pow, out, d.ldo, d.out_pstride, d.out_layout, d.stream);
            }
            if constexpr (f32) return entry::apply_edd_f32(plan, x, x2, d.T, d.ldx, d.layout, d.offset, d.thresholds, d.n_thr, out, d.ldo, d.out_pstride, d.out_layout, d.stream);
            else return entry::apply_edd_f64(plan, x, x2, d.T, d.ldx, d.layout, d.offset, d.thresholds, d.n_thr, out, d.ldo, d.out_pstride, d.out_layout, d.stream);
        case WAGG_SRC_HOST:
            if (d.transform == WAGG_XF_NONE) {
                if constexpr (f32) return entry::apply_host_ex_f32(plan, x, d.T, d.ldx, d.layout, out, d.ldo, d.out_layout, d.flags);
                else return entry::apply_host_ex_f64(plan, x, d.T, d.ldx, d.layout, out, d.ldo, d.out_layout, d.flags);
            }
            if (d.layout != WAGG_LAYOUT_TG || d.out_layout != WAGG_OUT_TR)
                return unsupported(d, "fused transforms of host-resident fields take (time, gridcell) data and give (time, region) results");
            if (d.transform == WAGG_XF_POLY) {
                if constexpr (f32) return entry::apply_poly_host_f32(plan, x, d.T, d.ldx, d.offset, d.pow_first, d.n_pow, out, d.ldo, d.out_pstride, d.flags);
                else return entry::apply_poly_host_f64(plan, x, d.T, d.ldx, d.offset, d.pow_first, d.n_pow, out, d.ldo, d.out_pstride, d.flags);
            }
            if constexpr (f32) return entry::apply_edd_host_f32(plan, x, x2, d.T, d.ldx, d.offset, d.thresholds, d.n_thr, out, d.ldo, d.out_pstride, d.flags);
            else return entry::apply_edd_host_f64(plan, x, x2, d.T, d.ldx, d.offset, d.thresholds, d.n_thr, out, d.ldo, d.out_pstride, d.flags);
        case WAGG_SRC_HOST_MULTI: {
            if (d.transform != WAGG_XF_NONE) return unsupported(d, "the multi-device host pipeline has no fused transforms");
            if (d.layout != WAGG_LAYOUT_TG || d.out_layout != WAGG_OUT_TR) return unsupported(d, "(time, gridcell) data and (time, region) results only");
            const wagg_plan *const *plans = static_cast<const wagg_plan *const *>(d.plan);
            if constexpr (f32) return entry::apply_host_multi_f32(plans, d.devices, d.n_plans, x, d.T, d.ldx, out, d.ldo, d.flags);
            else return entry::apply_host_multi_f64(plans, d.devices, d.n_plans, x, d.T, d.ldx, out, d.ldo, d.flags);
        }
        default: {
            if (d.transform != WAGG_XF_NONE) return unsupported(d, "the sharded form has no fused transforms");
            if (d.layout != WAGG_LAYOUT_TG || d.out_layout != WAGG_OUT_TR) return unsupported(d, "(time, gridcell) data and (time, region) results only");
            wagg_shard_group *g = static_cast<wagg_shard_group *>(d.group);
            const wagg_plan *const *plans = static_cast<const wagg_plan *const *>(d.plan);
            const T *const *xs = static_cast<const T *const *>(d.x);
            if constexpr (f32) return entry::apply_sharded_f32(g, plans, xs, d.rows, d.ldx, out, d.ldo, d.root);
            else return entry::apply_sharded_f64(g, plans, xs, d.rows, d.ldx, out, d.ldo, d.root);
        }
    }
}

template <typename T>
int run_dense(const wagg_apply_desc &d) {
    constexpr bool f32 = sizeof(T) == 4;
    if (d.layout != WAGG_LAYOUT_TG || d.out_layout != WAGG_OUT_TR)
        return unsupported(d, "dense-family plans take (time, gridcell) data and give (time, region) results");
    const T *x = static_cast<const T *>(d.x), *x2 = static_cast<const T *>(d.x2);
    T *out = static_cast<T *>(d.out);
    wagg_dense *plan = static_cast<wagg_dense *>(const_cast<void *>(d.plan));
    switch (d.source) {
        case WAGG_SRC_DEVICE:
            if (d.transform == WAGG_XF_NONE) {
                if constexpr (f32) return entry::dense_apply_f32(plan, x, d.T, d.ldx, out, d.ldo, d.ksplit, d.stream);
                else return entry::dense_apply_f64(plan, x, d.T, d.ldx, out, d.ldo, d.ksplit, d.stream);
            }
            if (d.transform == WAGG_XF_POLY) {
                if (d.n_pow != 1) return unsupported(d, "one power per call (n_pow == 1, the power in pow_first)");
                if constexpr (f32) return entry::dense_apply_poly_f32(plan, x, d.T, d.ldx, d.offset, d.pow_first, out, d.ldo, d.ksplit, d.stream);
                else return entry::dense_apply_poly_f64(plan, x, d.T, d.ldx, d.offset, d.pow_first, out, d.ldo, d.ksplit, d.stream);
            }
            if (d.n_thr != 1 || d.thresholds == nullptr) return unsupported(d, "one threshold per call (n_thr == 1)");
            if constexpr (f32) return entry::dense_apply_edd_f32(plan, x, x2, d.T, d.ldx, d.offset, d.thresholds[0], out, d.ldo, d.ksplit, d.stream);
            else return entry::dense_apply_edd_f64(plan, x, x2, d.T, d.ldx, d.offset, d.thresholds[0], out, d.ldo, d.ksplit, d.stream);
        case WAGG_SRC_HOST:
            if (d.transform != WAGG_XF_NONE) return unsupported(d, "host-resident fields through a dense-family plan have no fused transforms");
            if constexpr (f32) return entry::dense_apply_host_f32(plan, x, d.T, d.ldx, out, d.ldo, d.flags);
            else return entry::dense_apply_host_f64(plan, x, d.T, d.ldx, out, d.ldo, d.flags);
        case WAGG_SRC_HOST_MULTI: {
            if (d.transform != WAGG_XF_NONE) return unsupported(d, "the multi-device host pipeline has no fused transforms");
            wagg_dense *const *plans = static_cast<wagg_dense *const *>(d.plan);
            if constexpr (f32) return entry::dense_apply_host_multi_f32(plans, d.devices, d.n_plans, x, d.T, d.ldx, out, d.ldo, d.flags);
            else return entry::dense_apply_host_multi_f64(plans, d.devices, d.n_plans, x, d.T, d.ldx, out, d.ldo, d.flags);
        }
        default: {
            if (d.transform != WAGG_XF_NONE) return unsupported(d, "the sharded form has no fused transforms");
            wagg_shard_group *g = static_cast<wagg_shard_group *>(d.group);
            wagg_dense *const *plans = static_cast<wagg_dense *const *>(d.plan);
            const T *const *xs = static_cast<const T *const *>(d.x);
            if constexpr (f32) return entry::dense_apply_sharded_f32(g, plans, xs, d.rows, d.ldx, out, d.ldo, d.root);
            else return entry::dense_apply_sharded_f64(g, plans, xs, d.rows, d.ldx, out, d.ldo, d.root);
        }
    }
}

// descriptor of the common shape; the wrappers below set what differs
wagg_apply_desc base(int plan_kind, int elem, int source, const void *plan, const void *x, int64_t T, int64_t ldx, void *out, int64_t ldo) {
    wagg_apply_desc d;
    std::memset(&d, 0, sizeof(d));
    d.struct_size = sizeof(d);
    d.plan_kind = plan_kind; d.elem = elem; d.source = source; d.transform = WAGG_XF_NONE;
    d.plan = plan; d.x = x; d.T = T; d.ldx = ldx; d.out = out; d.ldo = ldo;
    d.layout = WAGG_LAYOUT_TG; d.out_layout = WAGG_OUT_TR;
    d.pow_first = 1; d.n_pow = 1;
    return d;
}
wagg_apply_desc &poly(wagg_apply_desc &d, double offset, int pow_first, int n_pow, int64_t pstride) {
    d.transform = WAGG_XF_POLY; d.offset = offset; d.pow_first = pow_first; d.n_pow = n_pow; d.out_pstride = pstride;
    return d;
}
wagg_apply_desc &edd(wagg_apply_desc &d, const void *tasmax, double offset, const double *thr, int n_thr, int64_t pstride) {
    d.transform = WAGG_XF_EDD; d.x2 = tasmax; d.offset = offset; d.thresholds = thr; d.n_thr = n_thr; d.out_pstride = pstride;
    return d;
}
wagg_apply_desc &lay(wagg_apply_desc &d, int layout, int out_layout) { d.layout = layout; d.out_layout = out_layout; return d; }

}  // namespace
}  // namespace wagg

extern "C" int wagg_apply(const wagg_apply_desc *desc) {
    using namespace wagg;
    WAGG_REQUIRE(desc != nullptr, "wagg_apply: descriptor is NULL");
    WAGG_REQUIRE(desc->struct_size >= offsetof(wagg_apply_desc, out_pstride) && desc->struct_size <= (uint64_t)1 << 20,
                 "wagg_apply: struct_size %llu is not the size of a wagg_apply_desc (this library: %zu)",
                 (unsigned long long)desc->struct_size, sizeof(wagg_apply_desc));
    wagg_apply_desc d;
    std::memset(&d, 0, sizeof(d));
    std::memcpy(&d, desc, desc->struct_size < sizeof(d) ? (size_t)desc->struct_size : sizeof(d));
    WAGG_REQUIRE(d.plan_kind == WAGG_PLAN_SEGMENT || d.plan_kind == WAGG_PLAN_DENSE, "wagg_apply: unknown plan_kind %d", d.plan_kind);
    WAGG_REQUIRE(d.elem == WAGG_T_F32 || d.elem == WAGG_T_F64, "wagg_apply: elem must be WAGG_T_F32 or WAGG_T_F64, got %d", d.elem);
    WAGG_REQUIRE(d.source >= WAGG_SRC_DEVICE && d.source <= WAGG_SRC_SHARDED, "wagg_apply: unknown source %d", d.source);
    WAGG_REQUIRE(d.transform >= WAGG_XF_NONE && d.transform <= WAGG_XF_EDD, "wagg_apply: unknown transform %d", d.transform);
    WAGG_REQUIRE(d.plan != nullptr, "wagg_apply: plan is NULL");
    if (d.source == WAGG_SRC_HOST_MULTI) WAGG_REQUIRE(d.n_plans >= 1, "wagg_apply: the multi-device host form needs n_plans >= 1 plan handles in `plan`");
    if (d.source == WAGG_SRC_SHARDED) WAGG_REQUIRE(d.group != nullptr, "wagg_apply: the sharded form needs a wagg_shard_group");
    if (d.plan_kind == WAGG_PLAN_SEGMENT) return d.elem == WAGG_T_F32 ? run_segment<float>(d) : run_segment<double>(d);
    return d.elem == WAGG_T_F32 ? run_dense<float>(d) : run_dense<double>(d);
}

// ---- the exported apply symbols of rounds 1-5: fill a descriptor, call wagg_apply ----------------------------------------
#define SEG WAGG_PLAN_SEGMENT
#define DEN WAGG_PLAN_DENSE
#define F32 WAGG_T_F32
#define F64 WAGG_T_F64
using wagg::base; using wagg::edd; using wagg::lay; using wagg::poly;

extern "C" int wagg_apply_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout, float *out_dev, int64_t ldo,
                              int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_DEVICE, plan, X_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(d, layout, out_layout));
}
extern "C" int wagg_apply_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout, double *out_dev, int64_t ldo,
                              int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_DEVICE, plan, X_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(d, layout, out_layout));
}
extern "C" int wagg_apply_host_ex_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, int layout, float *out_host,
                                      int64_t ldo, int out_layout, int flags) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_HOST, plan, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&lay(d, layout, out_layout));
}
extern "C" int wagg_apply_host_ex_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, int layout, double *out_host,
                                      int64_t ldo, int out_layout, int flags) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_HOST, plan, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&lay(d, layout, out_layout));
}
extern "C" int wagg_apply_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, int layout, float *out_host,
                                   int64_t ldo, int out_layout) {
    return wagg_apply_host_ex_f32(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, WAGG_HOST_PIN | WAGG_HOST_LINES);
}
extern "C" int wagg_apply_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, int layout, double *out_host,
                                   int64_t ldo, int out_layout) {
    return wagg_apply_host_ex_f64(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, WAGG_HOST_PIN | WAGG_HOST_LINES);
}
extern "C" int wagg_apply_host_multi_f32(const wagg_plan *const *plans, const int *devices, int n_devices, const float *X_host, int64_t T,
                                         int64_t ldx, float *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_HOST_MULTI, plans, X_host, T, ldx, out_host, ldo);
    d.devices = devices; d.n_plans = n_devices; d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_apply_host_multi_f64(const wagg_plan *const *plans, const int *devices, int n_devices, const double *X_host, int64_t T,
                                         int64_t ldx, double *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_HOST_MULTI, plans, X_host, T, ldx, out_host, ldo);
    d.devices = devices; d.n_plans = n_devices; d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_apply_poly_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout, double offset, int pow_first,
                                   int n_pow, float *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_DEVICE, plan, X_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(poly(d, offset, pow_first, n_pow, out_pstride), layout, out_layout));
}
extern "C" int wagg_apply_poly_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout, double offset, int pow_first,
                                   int n_pow, double *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_DEVICE, plan, X_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(poly(d, offset, pow_first, n_pow, out_pstride), layout, out_layout));
}
extern "C" int wagg_apply_poly_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, double offset, int pow_first, int n_pow,
                                        float *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_HOST, plan, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&poly(d, offset, pow_first, n_pow, out_pstride));
}
extern "C" int wagg_apply_poly_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, double offset, int pow_first, int n_pow,
                                        double *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_HOST, plan, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&poly(d, offset, pow_first, n_pow, out_pstride));
}
extern "C" int wagg_apply_edd_f32(const wagg_plan *plan, const float *tasmin_dev, const float *tasmax_dev, int64_t T, int64_t ldx, int layout,
                                  double offset, const double *thresholds, int n_thr, float *out_dev, int64_t ldo, int64_t out_pstride,
                                  int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_DEVICE, plan, tasmin_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(edd(d, tasmax_dev, offset, thresholds, n_thr, out_pstride), layout, out_layout));
}
extern "C" int wagg_apply_edd_f64(const wagg_plan *plan, const double *tasmin_dev, const double *tasmax_dev, int64_t T, int64_t ldx, int layout,
                                  double offset, const double *thresholds, int n_thr, double *out_dev, int64_t ldo, int64_t out_pstride,
                                  int out_layout, void *stream) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_DEVICE, plan, tasmin_dev, T, ldx, out_dev, ldo);
    d.stream = stream;
    return wagg_apply(&lay(edd(d, tasmax_dev, offset, thresholds, n_thr, out_pstride), layout, out_layout));
}
extern "C" int wagg_apply_edd_host_f32(const wagg_plan *plan, const float *tasmin_host, const float *tasmax_host, int64_t T, int64_t ldx, double offset,
                                       const double *thresholds, int n_thr, float *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_HOST, plan, tasmin_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&edd(d, tasmax_host, offset, thresholds, n_thr, out_pstride));
}
extern "C" int wagg_apply_edd_host_f64(const wagg_plan *plan, const double *tasmin_host, const double *tasmax_host, int64_t T, int64_t ldx, double offset,
                                       const double *thresholds, int n_thr, double *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_HOST, plan, tasmin_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&edd(d, tasmax_host, offset, thresholds, n_thr, out_pstride));
}
extern "C" int wagg_apply_sharded_f32(wagg_shard_group *g, const wagg_plan *const *plans, const float *const *X_dev, const int64_t *rows, int64_t ldx,
                                      float *out_root, int64_t ldo, int root) {
    wagg_apply_desc d = base(SEG, F32, WAGG_SRC_SHARDED, plans, X_dev, 0, ldx, out_root, ldo);
    d.group = g; d.rows = rows; d.root = root;
    return wagg_apply(&d);
}
extern "C" int wagg_apply_sharded_f64(wagg_shard_group *g, const wagg_plan *const *plans, const double *const *X_dev, const int64_t *rows, int64_t ldx,
                                      double *out_root, int64_t ldo, int root) {
    wagg_apply_desc d = base(SEG, F64, WAGG_SRC_SHARDED, plans, X_dev, 0, ldx, out_root, ldo);
    d.group = g; d.rows = rows; d.root = root;
    return wagg_apply(&d);
}

extern "C" int wagg_dense_apply_f32(wagg_dense *p, const float *X_dev, int64_t T, int64_t ldx, float *out_dev, int64_t ldo, int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_DEVICE, p, X_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_f64(wagg_dense *p, const double *X_dev, int64_t T, int64_t ldx, double *out_dev, int64_t ldo, int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_DEVICE, p, X_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_poly_f32(wagg_dense *p, const float *X_dev, int64_t T, int64_t ldx, double offset, int power, float *out_dev, int64_t ldo,
                                         int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_DEVICE, p, X_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&poly(d, offset, power, 1, 0));
}
extern "C" int wagg_dense_apply_poly_f64(wagg_dense *p, const double *X_dev, int64_t T, int64_t ldx, double offset, int power, double *out_dev, int64_t ldo,
                                         int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_DEVICE, p, X_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&poly(d, offset, power, 1, 0));
}
extern "C" int wagg_dense_apply_edd_f32(wagg_dense *p, const float *tasmin_dev, const float *tasmax_dev, int64_t T, int64_t ldx, double offset,
                                        double threshold, float *out_dev, int64_t ldo, int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_DEVICE, p, tasmin_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&edd(d, tasmax_dev, offset, &threshold, 1, 0));
}
extern "C" int wagg_dense_apply_edd_f64(wagg_dense *p, const double *tasmin_dev, const double *tasmax_dev, int64_t T, int64_t ldx, double offset,
                                        double threshold, double *out_dev, int64_t ldo, int ksplit, void *stream) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_DEVICE, p, tasmin_dev, T, ldx, out_dev, ldo);
    d.ksplit = ksplit; d.stream = stream;
    return wagg_apply(&edd(d, tasmax_dev, offset, &threshold, 1, 0));
}
extern "C" int wagg_dense_apply_host_f32(wagg_dense *p, const float *X_host, int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_HOST, p, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_host_f64(wagg_dense *p, const double *X_host, int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_HOST, p, X_host, T, ldx, out_host, ldo);
    d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_host_multi_f32(wagg_dense *const *plans, const int *devices, int n_devices, const float *X_host, int64_t T, int64_t ldx,
                                               float *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_HOST_MULTI, plans, X_host, T, ldx, out_host, ldo);
    d.devices = devices; d.n_plans = n_devices; d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_host_multi_f64(wagg_dense *const *plans, const int *devices, int n_devices, const double *X_host, int64_t T, int64_t ldx,
                                               double *out_host, int64_t ldo, int flags) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_HOST_MULTI, plans, X_host, T, ldx, out_host, ldo);
    d.devices = devices; d.n_plans = n_devices; d.flags = flags;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_sharded_f32(wagg_shard_group *g, wagg_dense *const *plans, const float *const *X_dev, const int64_t *rows, int64_t ldx,
                                            float *out_root, int64_t ldo, int root) {
    wagg_apply_desc d = base(DEN, F32, WAGG_SRC_SHARDED, plans, X_dev, 0, ldx, out_root, ldo);
    d.group = g; d.rows = rows; d.root = root;
    return wagg_apply(&d);
}
extern "C" int wagg_dense_apply_sharded_f64(wagg_shard_group *g, wagg_dense *const *plans, const double *const *X_dev, const int64_t *rows, int64_t ldx,
                                            double *out_root, int64_t ldo, int root) {
    wagg_apply_desc d = base(DEN, F64, WAGG_SRC_SHARDED, plans, X_dev, 0, ldx, out_root, ldo);
    d.group = g; d.rows = rows; d.root = root;
    return wagg_apply(&d);
}
#undef SEG
#undef DEN
#undef F32
#undef F64

// ---- the structs that cross the boundary by layout: their sizes here, and a pattern that shows their field order ----------
namespace wagg {
namespace {
template <typename F> void put(int &k, F &f) {
    ++k;
    if constexpr (std::is_pointer_v<F>) f = reinterpret_cast<F>(static_cast<uintptr_t>(k));
    else f = static_cast<F>(k);
}
template <typename F, size_t N> void put(int &k, F (&f)[N]) { for (F &e : f) put(k, e); }
template <typename... F> void number(F &...f) { int k = 0; (put(k, f), ...); }

// (structured bindings bind in DECLARATION order and fail to compile when the count is wrong: the pattern follows the
//  struct of include/wagg.h itself, not a second list of names kept by hand)
void ordinals(wagg_plan_info &s) {
    auto &[a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19] = s;
    number(a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19);
}
void ordinals(wagg_dense_info &s) {
    auto &[a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16] = s;
    number(a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16);
}
void ordinals(wagg_host_stats &s) {
    auto &[a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18] = s;
    number(a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18);
}
void ordinals(wagg_apply_desc &s) {
    auto &[a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27, a28, a29] = s;
    number(a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11, a12, a13, a14, a15, a16, a17, a18, a19, a20, a21, a22, a23, a24, a25, a26, a27, a28, a29);
}
template <typename S> int fill_ordinals(void *out, uint64_t size) {
    S s;
    std::memset(&s, 0, sizeof(s));
    ordinals(s);
    copy_sized(out, size, &s, sizeof(s));
    return WAGG_OK;
}
}  // namespace

void copy_sized(void *dst, uint64_t dst_size, const void *src, size_t src_size) {
    const size_t n = dst_size < src_size ? (size_t)dst_size : src_size;
    std::memcpy(dst, src, n);
    if (dst_size > src_size) std::memset(static_cast<char *>(dst) + src_size, 0, (size_t)(dst_size - src_size));
}
}  // namespace wagg

extern "C" int wagg_struct_size(int which) {
    switch (which) {
        case WAGG_STRUCT_PLAN_INFO: return (int)sizeof(wagg_plan_info);
        case WAGG_STRUCT_DENSE_INFO: return (int)sizeof(wagg_dense_info);
        case WAGG_STRUCT_HOST_STATS: return (int)sizeof(wagg_host_stats);
        case WAGG_STRUCT_APPLY_DESC: return (int)sizeof(wagg_apply_desc);
        default: wagg::set_error("wagg_struct_size: unknown struct id %d", which); return WAGG_EINVAL;
    }
}

extern "C" int wagg_struct_ordinals(int which, void *out, uint64_t size) {
    using namespace wagg;
    WAGG_REQUIRE(out != nullptr && size <= ((uint64_t)1 << 20), "wagg_struct_ordinals: NULL buffer or absurd size");
    switch (which) {
        case WAGG_STRUCT_PLAN_INFO: return fill_ordinals<wagg_plan_info>(out, size);
        case WAGG_STRUCT_DENSE_INFO: return fill_ordinals<wagg_dense_info>(out, size);
        case WAGG_STRUCT_HOST_STATS: return fill_ordinals<wagg_host_stats>(out, size);
        case WAGG_STRUCT_APPLY_DESC: return fill_ordinals<wagg_apply_desc>(out, size);
        default: set_error("wagg_struct_ordinals: unknown struct id %d", which); return WAGG_EINVAL;
    }
}
