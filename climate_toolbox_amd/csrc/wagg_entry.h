// The apply entry points by their old C names, as plain C++ functions (namespace wagg::entry): what wagg_apply()
// (wagg_desc.hip) dispatches a descriptor to, and what code INSIDE the library calls when it needs an apply (the host
// pipelines hand them on as per-block launchers, the shard group calls them per shard).  The exported wagg_*apply* symbols of
// include/wagg.h are wrappers in wagg_desc.hip that fill a wagg_apply_desc and call wagg_apply(); argument meaning: wagg.h.
#pragma once
#include "../../include/wagg.h"

namespace wagg {
namespace entry {
int apply_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout, float *out_dev, int64_t ldo, int out_layout, void *stream);
int apply_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout, double *out_dev, int64_t ldo, int out_layout, void *stream);
int apply_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, int layout, float *out_host, int64_t ldo, int out_layout);
int apply_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, int layout, double *out_host, int64_t ldo, int out_layout);
int apply_host_ex_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, int layout, float *out_host, int64_t ldo, int out_layout, int flags);
int apply_host_ex_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, int layout, double *out_host, int64_t ldo, int out_layout, int flags);
int apply_host_multi_f32(const wagg_plan *const *plans, const int *devices, int n_devices, const float *X_host, int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags);
int apply_host_multi_f64(const wagg_plan *const *plans, const int *devices, int n_devices, const double *X_host, int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags);
int apply_poly_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout, double offset, int pow_first, int n_pow, float *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
int apply_poly_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout, double offset, int pow_first, int n_pow, double *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
int apply_poly_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, double offset, int pow_first, int n_pow, float *out_host, int64_t ldo, int64_t out_pstride, int flags);
int apply_poly_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, double offset, int pow_first, int n_pow, double *out_host, int64_t ldo, int64_t out_pstride, int flags);
int apply_edd_f32(const wagg_plan *plan, const float *tasmin_dev, const float *tasmax_dev, int64_t T, int64_t ldx, int layout, double offset, const double *thresholds, int n_thr, float *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
int apply_edd_f64(const wagg_plan *plan, const double *tasmin_dev, const double *tasmax_dev, int64_t T, int64_t ldx, int layout, double offset, const double *thresholds, int n_thr, double *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
int apply_edd_host_f32(const wagg_plan *plan, const float *tasmin_host, const float *tasmax_host, int64_t T, int64_t ldx, double offset, const double *thresholds, int n_thr, float *out_host, int64_t ldo, int64_t out_pstride, int flags);
int apply_edd_host_f64(const wagg_plan *plan, const double *tasmin_host, const double *tasmax_host, int64_t T, int64_t ldx, double offset, const double *thresholds, int n_thr, double *out_host, int64_t ldo, int64_t out_pstride, int flags);
int apply_sharded_f32(wagg_shard_group *g, const wagg_plan *const *plans, const float *const *X_dev, const int64_t *rows, int64_t ldx, float *out_root, int64_t ldo, int root);
int apply_sharded_f64(wagg_shard_group *g, const wagg_plan *const *plans, const double *const *X_dev, const int64_t *rows, int64_t ldx, double *out_root, int64_t ldo, int root);
int dense_apply_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx, float *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx, double *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_poly_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx, double offset, int power, float *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_poly_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx, double offset, int power, double *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_edd_f32(wagg_dense *d, const float *tasmin_dev, const float *tasmax_dev, int64_t T, int64_t ldx, double offset, double threshold, float *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_edd_f64(wagg_dense *d, const double *tasmin_dev, const double *tasmax_dev, int64_t T, int64_t ldx, double offset, double threshold, double *out_dev, int64_t ldo, int ksplit, void *stream);
int dense_apply_host_f32(wagg_dense *d, const float *X_host, int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags);
int dense_apply_host_f64(wagg_dense *d, const double *X_host, int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags);
int dense_apply_host_multi_f32(wagg_dense *const *plans, const int *devices, int n_devices, const float *X_host, int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags);
int dense_apply_host_multi_f64(wagg_dense *const *plans, const int *devices, int n_devices, const double *X_host, int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags);
int dense_apply_sharded_f32(wagg_shard_group *g, wagg_dense *const *plans, const float *const *X_dev, const int64_t *rows, int64_t ldx, float *out_root, int64_t ldo, int root);
int dense_apply_sharded_f64(wagg_shard_group *g, wagg_dense *const *plans, const double *const *X_dev, const int64_t *rows, int64_t ldx, double *out_root, int64_t ldo, int root);
}  // namespace entry
}  // namespace wagg
