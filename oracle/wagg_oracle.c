/* CPU oracle (plain C) for the weighted grid->region aggregation path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product path (climate_toolbox_amd/, include/wagg.h) never links or calls it.
 *
 * It restates, on the coded segment table that crosses the C-ABI, the arithmetic of
 *   /root/reference/climate_toolbox/aggregations/aggregations.py
 *     :24-27  gather of the grid cells listed in the segment table
 *     :78     (ds[variable] * ds[aggwt]).groupby(agglev).sum(dim="reshape_index")   (fp64, skipna)
 *     :79     ds[aggwt].groupby(agglev).sum(dim="reshape_index")
 *     :77-80  numerator / denominator (IEEE division, no guard)
 * Label resolution (:27 exact match), backup fill (:73) and label factorisation happen on the
 * host before this point, exactly as for the HIP engine (see oracle/ref_numpy.py for those).
 * Parity status: "parity unpinned" in the strict sense (no reference-produced number exists); see
 * the header of oracle/ref_numpy.py for what pins the numbers instead (restatement agreement).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define WAGG_LAYOUT_TG 0 /* X[t*ldx + g] */
#define WAGG_LAYOUT_GT 1 /* X[g*ldx + t] */

int wagg_oracle_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* A fixed team for the threaded legs: no dynamic adjustment of the thread count between calls (bench.py also sets
 * OMP_PROC_BIND=spread OMP_PLACES=cores before this library loads, and warms the team with untimed calls). */
static void wagg_oracle_fixed_team(void) {
#ifdef _OPENMP
    static int done = 0;
    if (!done) { omp_set_dynamic(0); done = 1; }
#endif
}

/* Faithful, single-threaded like the reference: per timestep gather -> multiply -> group-sum.
 * out is (T, R) row-major double.  Returns 0, or -1 on a bad index. */
#define DEFINE_SEGMENTS(NAME, TYPE)                                                              \
    int NAME(const TYPE *X, int64_t T, int64_t ldx, int layout, const int32_t *cell_idx,          \
             const int32_t *region_code, const double *w_eff, int64_t nseg, int64_t G, int32_t R, \
             double *out) {                                                                       \
        double *den = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));                  \
        if (!den) return -2;                                                                      \
        for (int64_t i = 0; i < nseg; ++i) {                                                      \
            int32_t r = region_code[i];                                                           \
            if (r < 0) continue;                       /* null label: row dropped (S3) */         \
            if (r >= R || cell_idx[i] < 0 || cell_idx[i] >= G) { free(den); return -1; }          \
            if (!isnan(w_eff[i])) den[r] += w_eff[i];  /* :79, skipna */                          \
        }                                                                                         \
        for (int64_t t = 0; t < T; ++t) {                                                         \
            double *row = out + t * (int64_t)R;                                                   \
            for (int32_t r = 0; r < R; ++r) row[r] = 0.0;                                         \
            for (int64_t i = 0; i < nseg; ++i) {                                                  \
                int32_t r = region_code[i];                                                       \
                if (r < 0) continue;                                                              \
                int64_t g = cell_idx[i];                                                          \
                double x = (double)(layout == WAGG_LAYOUT_TG ? X[t * ldx + g] : X[g * ldx + t]);  \
                double p = x * w_eff[i];               /* :78 fp64 product (S8) */                \
                if (!isnan(p)) row[r] += p;            /* skipna (S6) */                          \
            }                                                                                     \
            for (int32_t r = 0; r < R; ++r) row[r] = row[r] / den[r]; /* :77-80 (S7) */           \
        }                                                                                         \
        free(den);                                                                                \
        return 0;                                                                                 \
    }

DEFINE_SEGMENTS(wagg_oracle_segments_f32, float)
DEFINE_SEGMENTS(wagg_oracle_segments_f64, double)

/* counter hash shared with the device generator (climate_toolbox_amd/csrc/wagg_synth.hip) */
static inline uint32_t hash32(uint64_t idx, uint32_t seed) {
    uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t x = lo ^ (hi * 0x85EBCA6Bu) ^ (seed * 0x9E3779B9u);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
float wagg_oracle_hash_u01(uint64_t idx, uint32_t seed) {
    return (float)(hash32(idx, seed) >> 8) * (1.0f / 16777216.0f);
}

/* Dense form with structured sparsity: only a fraction `fill` of the entries is non-zero (kept where
 * hash_u01(g*R + r, seed ^ 0x9e3779b9) < fill; fill >= 1 keeps all) -- BASELINE configs[4], the c5
 * "uniform-random columns" structure of include/wagg.h wagg_dense_create_synth_sparse -- and with
 * blocklocal != 0 additionally only inside the 256-region column tile (97 * (g / 64)) mod
 * ceil(R_total / 256) of each 64-cell run (wagg_dense_create_synth_blocklocal).  Unlike the engine's
 * MFMA forms this multiplies kept pairs only, so +-inf data stays with the regions that own the cell
 * (S6).  den comes from ALL G cells of the column window only when g0 == 0 and Gw == G; it is the
 * window's own column sum otherwise (callers compare windows that span the whole grid). */
int wagg_oracle_dense_synth2_f32(const float *X, int64_t T, int64_t ldx, int64_t g0, int64_t Gw,
                                 int64_t R_total, int64_t r0, int64_t Rw, uint32_t seed, double fill,
                                 int blocklocal, double *out) {
    double *den = (double *)calloc((size_t)Rw, sizeof(double));
    if (!den) return -2;
    memset(out, 0, sizeof(double) * (size_t)T * (size_t)Rw);
    enum { GB2 = 64 };
    float *wblk = (float *)malloc(sizeof(float) * GB2 * (size_t)Rw);
    if (!wblk) { free(den); return -2; }
    const int64_t n_nt = (R_total + 255) / 256;
    const float ffill = (float)fill;
    for (int64_t gb = 0; gb < Gw; gb += GB2) {
        int64_t gn = Gw - gb < GB2 ? Gw - gb : GB2;
#pragma omp parallel for schedule(static)
        for (int64_t gi = 0; gi < gn; ++gi) {
            const int64_t g = g0 + gb + gi;
            const int64_t owner = (97 * (g / 64)) % n_nt;
            for (int64_t r = 0; r < Rw; ++r) {
                const uint64_t id = (uint64_t)g * (uint64_t)R_total + (uint64_t)(r0 + r);
                int keep = fill >= 1.0 || wagg_oracle_hash_u01(id, seed ^ 0x9e3779b9u) < ffill;
                if (blocklocal && (r0 + r) / 256 != owner) keep = 0;
                wblk[gi * Rw + r] = keep ? wagg_oracle_hash_u01(id, seed) : 0.0f;
            }
        }
        for (int64_t gi = 0; gi < gn; ++gi)
            for (int64_t r = 0; r < Rw; ++r) den[r] += (double)wblk[gi * Rw + r];
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < T; ++t) {
            double *row = out + t * Rw;
            for (int64_t gi = 0; gi < gn; ++gi) {
                const float xf = X[t * ldx + g0 + gb + gi];
                const float *wr = wblk + gi * Rw;
                for (int64_t r = 0; r < Rw; ++r) {
                    if (wr[r] == 0.0f) continue;             /* no segment row for this pair */
                    const double p = (double)xf * (double)wr[r];
                    if (!isnan(p)) row[r] += p;              /* skipna (S6) */
                }
            }
        }
    }
    for (int64_t t = 0; t < T; ++t)
        for (int64_t r = 0; r < Rw; ++r) out[t * Rw + r] /= den[r];
    free(wblk);
    free(den);
    return 0;
}

/* Dense form (X . W) / (1^T W) with W[g,r] = hash_u01(g*R_total + r0 + r, seed) generated on the
 * fly for the column window [r0, r0+Rw) and the row window [g0, g0+Gw) -- the CPU comparator of
 * the c2-dense workload on a bounded sample.  fp32 inputs, fp64 accumulation, threads = OpenMP.
 * X is (T, ldx) with the Gw cells of the window at columns g0.. ; out is (T, Rw) double. */
int wagg_oracle_dense_synth_f32(const float *X, int64_t T, int64_t ldx, int64_t g0, int64_t Gw,
                                int64_t R_total, int64_t r0, int64_t Rw, uint32_t seed,
                                double *out) {
    double *den = (double *)calloc((size_t)Rw, sizeof(double));
    if (!den) return -2;
    memset(out, 0, sizeof(double) * (size_t)T * (size_t)Rw);
    enum { GB = 64 };
    float *wblk = (float *)malloc(sizeof(float) * GB * (size_t)Rw);
    if (!wblk) { free(den); return -2; }
    for (int64_t gb = 0; gb < Gw; gb += GB) {
        int64_t gn = Gw - gb < GB ? Gw - gb : GB;
#pragma omp parallel for schedule(static)
        for (int64_t gi = 0; gi < gn; ++gi)
            for (int64_t r = 0; r < Rw; ++r)
                wblk[gi * Rw + r] =
                    wagg_oracle_hash_u01((uint64_t)(g0 + gb + gi) * (uint64_t)R_total + (uint64_t)(r0 + r), seed);
        for (int64_t gi = 0; gi < gn; ++gi)
            for (int64_t r = 0; r < Rw; ++r) den[r] += (double)wblk[gi * Rw + r];
#pragma omp parallel for schedule(static)
        for (int64_t t = 0; t < T; ++t) {
            double *row = out + t * Rw;
            for (int64_t gi = 0; gi < gn; ++gi) {
                float xf = X[t * ldx + g0 + gb + gi];
                double x = isnan(xf) ? 0.0 : (double)xf;   /* S6 */
                const float *wr = wblk + gi * Rw;
                for (int64_t r = 0; r < Rw; ++r) row[r] += x * (double)wr[r];
            }
        }
    }
    for (int64_t t = 0; t < T; ++t)
        for (int64_t r = 0; r < Rw; ++r) out[t * Rw + r] /= den[r];
    free(wblk);
    free(den);
    return 0;
}

/* The same dense-form restatement (fp32 inputs, fp64 accumulation, kept pairs only, skipna) for an
 * ARBITRARY LIST of output columns over all G cells: out[t * ncols + j] is region cols[j].  This is what lets a
 * parity test touch every 256-region column tile of a full-size result (one short window per tile) in seconds:
 * each thread owns whole columns, so there is one parallel region instead of two per 64 cells. */
int wagg_oracle_dense_synth_cols_f32(const float *X, int64_t T, int64_t ldx, int64_t G, int64_t R_total,
                                     const int64_t *cols, int64_t ncols, uint32_t seed, double fill,
                                     int blocklocal, double *out) {
    const int64_t n_nt = (R_total + 255) / 256;
    const float ffill = (float)fill;
    int failed = 0;
    enum { CB = 8 };                                      /* columns per work item */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t c0 = 0; c0 < ncols; c0 += CB) {
        const int64_t cn = ncols - c0 < CB ? ncols - c0 : CB;
        double *acc = (double *)calloc((size_t)(T + 1) * CB, sizeof(double));   /* [T][CB] sums, then [CB] den */
        if (!acc) { failed = 1; continue; }
        double *den = acc + T * CB;
        float w[CB];
        for (int64_t g = 0; g < G; ++g) {
            const int64_t owner = (97 * (g / 64)) % n_nt;
            int any = 0;
            for (int64_t j = 0; j < cn; ++j) {
                const int64_t r = cols[c0 + j];
                const uint64_t id = (uint64_t)g * (uint64_t)R_total + (uint64_t)r;
                int keep = fill >= 1.0 || wagg_oracle_hash_u01(id, seed ^ 0x9e3779b9u) < ffill;
                if (blocklocal && r / 256 != owner) keep = 0;
                w[j] = keep ? wagg_oracle_hash_u01(id, seed) : 0.0f;
                den[j] += (double)w[j];
                any |= w[j] != 0.0f;
            }
            if (!any) continue;
            for (int64_t t = 0; t < T; ++t) {
                const float xf = X[t * ldx + g];
                for (int64_t j = 0; j < cn; ++j) {
                    if (w[j] == 0.0f) continue;                       /* no segment row for this pair */
                    const double p = (double)xf * (double)w[j];
                    if (!isnan(p)) acc[t * CB + j] += p;              /* skipna (S6) */
                }
            }
        }
        for (int64_t t = 0; t < T; ++t)
            for (int64_t j = 0; j < cn; ++j) out[t * ncols + c0 + j] = acc[t * CB + j] / den[j];
        free(acc);
    }
    return failed ? -2 : 0;
}

/* The c5 weight tables (uniform-random, or block-local with blocklocal != 0; see wagg_oracle_dense_synth2_f32) as the
 * CSR table a caller would hold: rows = grid cells, columns = region codes ascending, fp64 values = the fp32 hashes.
 * Rows [g0, g1).  Call with col == NULL to get rowptr (g1 - g0 + 1 offsets, rowptr[0] = 0) and the entry count;
 * then with arrays of that capacity.  Returns the number of entries, or -2 when capacity is too small.
 * (Test infrastructure: the input of the "caller-supplied table" parity tests, generated independently of the
 * device generator wagg_synth_table_csr, which must agree with it bit for bit.) */
/* keep[r - r0] = hash_u01(g R + r, seed2) < fill for r in [r0, r1), in a form the compiler vectorises: the high word of
 * the 64-bit counter changes at most once inside a row, and hash_u01 < fill compares 24-bit integers
 * (hash_u01 = (h >> 8) 2^-24 exactly, so it is below fill iff h >> 8 < ceil(fill 2^24)). */
static int64_t keep_flags(int64_t g, int64_t R, int64_t r0, int64_t r1, uint32_t seed2, uint32_t thr, uint8_t *keep) {
    const uint64_t base = (uint64_t)g * (uint64_t)R;
    int64_t n = 0, r = r0;
    while (r < r1) {
        const uint64_t id0 = base + (uint64_t)r;
        const uint32_t lo0 = (uint32_t)id0, hi = (uint32_t)(id0 >> 32);
        int64_t seg = (int64_t)(0x100000000ull - (uint64_t)lo0);          /* counters left before the low word wraps */
        if (seg > r1 - r) seg = r1 - r;
        const uint32_t hmix = (hi * 0x85EBCA6Bu) ^ (seed2 * 0x9E3779B9u);
        uint8_t *k = keep + (r - r0);
        int64_t m = 0;
        for (int64_t i = 0; i < seg; ++i) {
            uint32_t x = (lo0 + (uint32_t)i) ^ hmix;
            x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
            const uint8_t f = (x >> 8) < thr;
            k[i] = f;
            m += f;
        }
        n += m;
        r += seg;
    }
    return n;
}

int64_t wagg_oracle_synth_csr(int64_t G, int64_t R, uint32_t seed, double fill, int blocklocal, int64_t g0, int64_t g1,
                              int64_t *rowptr, int32_t *col, double *val, int64_t capacity) {
    const int64_t n_nt = (R + 255) / 256;
    const float ffill = (float)fill;
    const double scaled = (double)ffill * 16777216.0;
    const uint32_t thr = scaled >= 16777216.0 ? 0x1000000u : (uint32_t)ceil(scaled);
    const int64_t width = blocklocal ? 256 : R;
    int failed = 0;
    (void)G;
    rowptr[0] = 0;
    for (int pass = 0; pass < (col ? 2 : 1); ++pass) {
        if (pass == 1 && capacity < rowptr[g1 - g0]) return -2;
#pragma omp parallel
        {
            uint8_t *keep = (uint8_t *)malloc((size_t)width + 8);
            if (!keep) {
#pragma omp atomic write
                failed = 1;
            }
#pragma omp for schedule(static, 64)
            for (int64_t g = g0; g < g1; ++g) {
                if (!keep) continue;
                int64_t r0 = 0, r1 = R;
                if (blocklocal) { r0 = ((97 * (g / 64)) % n_nt) * 256; r1 = r0 + 256 < R ? r0 + 256 : R; }
                const int64_t n = keep_flags(g, R, r0, r1, seed ^ 0x9e3779b9u, thr, keep);
                if (pass == 0) { rowptr[g - g0 + 1] = n; continue; }
                int64_t at = rowptr[g - g0];
                for (int64_t r = r0; r < r1; ++r) {
                    if (!keep[r - r0]) continue;
                    col[at] = (int32_t)r;
                    val[at] = (double)wagg_oracle_hash_u01((uint64_t)g * (uint64_t)R + (uint64_t)r, seed);
                    ++at;
                }
            }
            free(keep);
        }
        if (failed) return -3;
        if (pass == 0)
            for (int64_t i = 0; i < g1 - g0; ++i) rowptr[i + 1] += rowptr[i];
    }
    return rowptr[g1 - g0];
}

/* Best-effort CPU form of the segment-table aggregation (SURVEY 8d "threaded C++ SpMM over all host cores"): the same
 * arithmetic as wagg_oracle_segments_* (fp64 products, skipna, IEEE division), the timesteps dealt to OpenMP threads;
 * per timestep the sums run in table order, so the result equals the single-threaded form bit for bit. */
#define DEFINE_SEGMENTS_OMP(NAME, TYPE)                                                           \
    int NAME(const TYPE *X, int64_t T, int64_t ldx, int layout, const int32_t *cell_idx,          \
             const int32_t *region_code, const double *w_eff, int64_t nseg, int64_t G, int32_t R, \
             double *out) {                                                                       \
        double *den = (double *)calloc((size_t)(R > 0 ? R : 1), sizeof(double));                  \
        if (!den) return -2;                                                                      \
        for (int64_t i = 0; i < nseg; ++i) {                                                      \
            int32_t r = region_code[i];                                                           \
            if (r < 0) continue;                                                                  \
            if (r >= R || cell_idx[i] < 0 || cell_idx[i] >= G) { free(den); return -1; }          \
            if (!isnan(w_eff[i])) den[r] += w_eff[i];                                             \
        }                                                                                         \
        wagg_oracle_fixed_team();                                                                 \
        _Pragma("omp parallel for schedule(static) proc_bind(spread)")                            \
        for (int64_t t = 0; t < T; ++t) {                                                         \
            double *row = out + t * (int64_t)R;                                                   \
            for (int32_t r = 0; r < R; ++r) row[r] = 0.0;                                         \
            for (int64_t i = 0; i < nseg; ++i) {                                                  \
                int32_t r = region_code[i];                                                       \
                if (r < 0) continue;                                                              \
                int64_t g = cell_idx[i];                                                          \
                double x = (double)(layout == WAGG_LAYOUT_TG ? X[t * ldx + g] : X[g * ldx + t]);  \
                double p = x * w_eff[i];                                                          \
                if (!isnan(p)) row[r] += p;                                                       \
            }                                                                                     \
            for (int32_t r = 0; r < R; ++r) row[r] = row[r] / den[r];                             \
        }                                                                                         \
        free(den);                                                                                \
        return 0;                                                                                 \
    }

DEFINE_SEGMENTS_OMP(wagg_oracle_segments_omp_f32, float)
DEFINE_SEGMENTS_OMP(wagg_oracle_segments_omp_f64, double)
