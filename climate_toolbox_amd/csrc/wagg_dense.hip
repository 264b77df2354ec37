// Dense path: out = (nan0(X) . W) / (1^T W) with W a (gridcell x region) fp32 matrix resident in
// HBM (c2-dense: 1,036,800 x 24,378 = 101 GB).  This is the algebraic form of
// aggregations.py:78-80 when every (cell, region) pair carries a weight.
//
// Shape: M = T (365) is skinny, K = G (1e6) is huge, N = R (24k).  fp32 MFMA runs at the vector
// rate (256 flop/clk/CU), so the contraction is MFMA-issue-bound by ~7x over HBM; the design
// therefore spends nothing on bandwidth tricks and everything on keeping the matrix pipe fed:
//   * one workgroup (8 waves, 2 per SIMD) owns ALL 365 rows (padded to 23 x 16 = 368, 0.8 %
//     waste) x 128 columns of the output, so every W element is read from HBM exactly once;
//   * wave w owns columns [16w, 16w+16): 23 independent 16x16 accumulators (92 AGPR/VGPR), one B
//     fragment per k-step feeds 23 back-to-back v_mfma_f32_16x16x4_f32 (no dependent-issue
//     stalls: 40-cycle latency vs 23 x 32 cycles between reuses of an accumulator);
//   * K is split into S slices (multiple of 8): blocks with equal blockIdx % 8 (one XCD under
//     round-robin placement -- speed only) walk the same k-slice over neighbouring column
//     tiles, so the 23.5 KB X panel of each k-step is served by that XCD's L2;
//   * global -> register -> LDS staging, double-buffered, ONE barrier per 16-deep k-step; both
//     LDS images are k-major with row strides 368 and 144 words (= 16 mod 32) so that the
//     ds_read_b32 fragment reads (lanes 0-15: k, lanes 16-31: k+1) are bank-conflict free;
//   * fp32 partial slabs per (tile, k-slice), then one reduce kernel fuses the division by
//     den[r] (deterministic, no atomics).
#include <algorithm>
#include <cmath>

#include "wagg_common.h"

namespace wagg {

constexpr int D_MT = 23;            // 16-row MFMA tiles per workgroup
constexpr int D_BM = D_MT * 16;     // 368 rows
constexpr int D_BN = 128;           // 8 waves x 16 columns
constexpr int D_LDA = 368;          // 368 % 32 == 16
constexpr int D_LDB = 144;          // 144 % 32 == 16
constexpr int D_THREADS = 512;

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float nan0(float v) { return v == v ? v : 0.0f; }

template <int BK> struct DenseCfg {
    static constexpr int STAGE = BK * D_LDA + BK * D_LDB;                         // floats per LDS buffer
    static constexpr int XQ = D_BM * (BK / 4);                                    // float4 pieces of the X tile
    static constexpr int XLOADS = (XQ + D_THREADS - 1) / D_THREADS;               // per thread
    static constexpr int WLOADS = BK * D_BN / 4 / D_THREADS;                      // per thread
    static_assert(BK % 16 == 0 && WLOADS >= 1, "BK must be a multiple of 16");
};

template <int BK, bool ALIGNED>
__global__ __launch_bounds__(D_THREADS, 2) void dense_mfma_kernel(
    const float *__restrict__ X, int64_t Ttot, int64_t ldx, const float *__restrict__ W,
    int64_t ldw, int64_t G, int n_nt, int n_mb, int S, int64_t k_per_slice,
    float *__restrict__ slabs) {
    using C = DenseCfg<BK>;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][STAGE]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, kq = lane >> 4;

    // work item: blocks with equal (blockIdx % 8) share a k-slice (XCD L2 affinity, speed only)
    int j = blockIdx.x >> 3;
    const int nt = j % n_nt; j /= n_nt;
    const int mb = j % n_mb;
    const int ks = (blockIdx.x & 7) + 8 * (j / n_mb);
    const int64_t k_begin = (int64_t)ks * k_per_slice;
    const int64_t k_end = (k_begin + k_per_slice < G) ? k_begin + k_per_slice : G;
    const int64_t klen = k_end > k_begin ? k_end - k_begin : 0;
    const int nfull = (int)(klen / BK);
    const int ntiles = (int)((klen + BK - 1) / BK);
    const int64_t n0 = (int64_t)nt * D_BN;
    const int64_t m0 = (int64_t)mb * D_BM;

    // staging coordinates.  Rows past T are clamped to the last row: their accumulators hold
    // garbage that the reduce kernel never reads, and the loads need no row predicate.
    const float *xp[C::XLOADS];
    int xoff[C::XLOADS];      // LDS word offset of element c=0 of the piece, -1 = this lane idles
    int xk[C::XLOADS];        // k offset of the piece inside the tile
#pragma unroll
    for (int i = 0; i < C::XLOADS; ++i) {
        const int idx = tid + D_THREADS * i;
        const int cidx = idx < C::XQ ? idx : C::XQ - 1;
        const int row = cidx % D_BM, q = cidx / D_BM;
        int64_t grow = m0 + row;
        grow = grow < Ttot ? grow : Ttot - 1;
        xp[i] = X + grow * ldx + k_begin + q * 4;
        xoff[i] = idx < C::XQ ? (q * 4) * D_LDA + row : -1;
        xk[i] = q * 4;
    }
    const float *wp[C::WLOADS];
    int woff[C::WLOADS];
#pragma unroll
    for (int i = 0; i < C::WLOADS; ++i) {
        const int idx = tid + D_THREADS * i;
        const int row = idx >> 5, c4 = idx & 31;
        wp[i] = W + (k_begin + row) * ldw + n0 + c4 * 4;
        woff[i] = row * D_LDB + c4 * 4;
    }

    f32x4 acc[D_MT];
#pragma unroll
    for (int m = 0; m < D_MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 xr[C::XLOADS], wr[C::WLOADS];
    // full tile: no k predicate anywhere -> straight-line loads that stay in flight under the MFMAs
    auto load_full = [&](int tile) {
#pragma unroll
        for (int i = 0; i < C::XLOADS; ++i) {
            const float *p = xp[i] + (int64_t)tile * BK;
            if (ALIGNED) xr[i] = *reinterpret_cast<const f32x4 *>(p);
            else xr[i] = f32x4{p[0], p[1], p[2], p[3]};
        }
#pragma unroll
        for (int i = 0; i < C::WLOADS; ++i)
            wr[i] = *reinterpret_cast<const f32x4 *>(wp[i] + (int64_t)tile * BK * ldw);
    };
    // the (at most one) ragged last tile of a slice
    auto load_tail = [&](int tile) {
        const int64_t krem = klen - (int64_t)tile * BK;      // 1 .. BK-1 valid k
#pragma unroll
        for (int i = 0; i < C::XLOADS; ++i) {
            const float *p = xp[i] + (int64_t)tile * BK;
#pragma unroll
            for (int c = 0; c < 4; ++c) xr[i][c] = (xk[i] + c < krem) ? p[c] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < C::WLOADS; ++i) {
            const int row = (tid + D_THREADS * i) >> 5;
            wr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row < krem) wr[i] = *reinterpret_cast<const f32x4 *>(wp[i] + (int64_t)tile * BK * ldw);
        }
    };
    auto load_tile = [&](int tile) { if (tile < nfull) load_full(tile); else load_tail(tile); };
    auto store_tile = [&](int buf) {
        float *xs = lds + buf * C::STAGE;
        float *ws = xs + BK * D_LDA;
#pragma unroll
        for (int i = 0; i < C::XLOADS; ++i) {
            if (xoff[i] >= 0) {
#pragma unroll
                for (int c = 0; c < 4; ++c) xs[xoff[i] + c * D_LDA] = nan0(xr[i][c]);   // S6
            }
        }
#pragma unroll
        for (int i = 0; i < C::WLOADS; ++i) *reinterpret_cast<f32x4 *>(ws + woff[i]) = wr[i];
    };

    if (ntiles > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int tile = 0; tile < ntiles; ++tile) {
        const int cur = tile & 1;
        if (tile + 1 < ntiles) load_tile(tile + 1);       // global loads fly under the MFMAs
        const float *xs = lds + cur * C::STAGE;
        const float *ws = xs + BK * D_LDA;
        // fragment double buffer: the ds_reads of k-step s+1 are issued before the 23 MFMAs of
        // k-step s, so LDS latency never sits between two MFMAs
        float af[2][D_MT], bf[2];
        auto load_frag = [&](int kk, float (&a)[D_MT], float &b) {
            b = ws[(kk + kq) * D_LDB + wave * 16 + lr];
            const float *xa = xs + (kk + kq) * D_LDA + lr;
#pragma unroll
            for (int m = 0; m < D_MT; ++m) a[m] = xa[m * 16];
        };
        load_frag(0, af[0], bf[0]);
#pragma unroll
        for (int s4 = 0; s4 < BK / 4; ++s4) {
            // hipcc otherwise sinks every ds_read next to its MFMA and waits lgkmcnt(0) between
            // each pair; the sched_barriers keep the next step's reads ahead of this step's MFMAs
            if (s4 + 1 < BK / 4) load_frag((s4 + 1) * 4, af[(s4 + 1) & 1], bf[(s4 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < D_MT; ++m)
                acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s4 & 1][m], bf[s4 & 1], acc[m], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tile + 1 < ntiles) store_tile(cur ^ 1);
        __syncthreads();
    }

    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    float *slab = slabs + ((((int64_t)mb * n_nt + nt) * S + ks) * D_BM) * D_BN;
#pragma unroll
    for (int m = 0; m < D_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            slab[(m * 16 + kq * 4 + r) * D_BN + wave * 16 + lr] = acc[m][r];
}

// out[t, r] = sum_s slab[mb][nt][s][t_local][c] / den[r]        (aggregations.py:77-80 fused)
__global__ void dense_reduce_kernel(const float *__restrict__ slabs, int n_nt, int S, int64_t Ttot,
                                    int32_t R, const float *__restrict__ den,
                                    float *__restrict__ out, int64_t ldo) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = blockIdx.y;
    if (r >= R) return;
    const int mb = (int)(t / D_BM), tl = (int)(t % D_BM);
    const int nt = (int)(r / D_BN), c = (int)(r % D_BN);
    const float *p = slabs + ((((int64_t)mb * n_nt + nt) * S) * D_BM + tl) * D_BN + c;
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += p[(int64_t)k * D_BM * D_BN];
    out[t * ldo + r] = s / den[r];
}

__global__ void dense_synth_w_kernel(float *__restrict__ W, int64_t G, int32_t R, int64_t ldw,
                                     uint32_t seed) {
    const int64_t n4 = ldw / 4;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < G * n4; i += stride) {
        const int64_t g = i / n4, r = (i % n4) * 4;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            v[c] = (r + c < R) ? hash_u01((uint64_t)g * (uint64_t)R + (uint64_t)(r + c), seed) : 0.f;
        *reinterpret_cast<f32x4 *>(W + g * ldw + r) = v;
    }
}

// column sums in fp64 (plan time): block = 256 columns x a strip of rows
__global__ void dense_colsum_kernel(const float *__restrict__ W, int64_t G, int32_t R, int64_t ldw,
                                    int64_t rows_per_block, double *__restrict__ den) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t g0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t g1 = g0 + rows_per_block < G ? g0 + rows_per_block : G;
    double s = 0.0;
    for (int64_t g = g0; g < g1; ++g) s += (double)W[g * ldw + r];
    atomicAdd(&den[r], s);
}

__global__ void dense_den32_kernel(const double *__restrict__ den64, float *__restrict__ den32, int32_t R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) den32[r] = (float)den64[r];
}

__global__ void dense_scatter_kernel(float *__restrict__ W, int64_t ldw, const int32_t *__restrict__ cell,
                                     const int32_t *__restrict__ region, const float *__restrict__ w,
                                     int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) W[(int64_t)cell[i] * ldw + region[i]] = w[i];
}

}  // namespace wagg

struct wagg_dense {
    int64_t G = 0, ldw = 0;
    int32_t R = 0;
    wagg::DevBuf<float> W, den32, slabs;
    wagg::DevBuf<double> den64;
    std::vector<double> den_host;
    bool den_exact_host = false;
};

namespace wagg {

static int dense_alloc(int64_t G, int32_t R, wagg_dense **out) {
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(G > 0 && R > 0, "bad sizes G=%lld R=%d", (long long)G, R);
    wagg_dense *d = new (std::nothrow) wagg_dense();
    if (!d) { set_error("host allocation failed"); return WAGG_ENOMEM; }
    d->G = G; d->R = R;
    d->ldw = ((int64_t)R + D_BN - 1) / D_BN * D_BN;
    hipError_t e = d->W.alloc((size_t)(G * d->ldw));
    if (e == hipSuccess) e = d->den32.alloc((size_t)R);
    if (e == hipSuccess) e = d->den64.alloc((size_t)R);
    if (e != hipSuccess) {
        set_error("dense W allocation of %.1f GB failed: %s", (double)G * d->ldw * 4e-9, hipGetErrorString(e));
        delete d;
        return e == hipErrorOutOfMemory ? WAGG_ENOMEM : WAGG_EHIP;
    }
    *out = d;
    return WAGG_OK;
}

static int dense_finish_den(wagg_dense *d) {
    WAGG_HIP(hipMemset(d->den64.p, 0, sizeof(double) * (size_t)d->R));
    const int64_t rows_per_block = 4096;
    dim3 grid((unsigned)((d->R + 255) / 256), (unsigned)((d->G + rows_per_block - 1) / rows_per_block));
    hipLaunchKernelGGL(dense_colsum_kernel, grid, dim3(256), 0, nullptr, d->W.p, d->G, d->R, d->ldw,
                       rows_per_block, d->den64.p);
    hipLaunchKernelGGL(dense_den32_kernel, dim3((unsigned)((d->R + 255) / 256)), dim3(256), 0, nullptr,
                       d->den64.p, d->den32.p, d->R);
    WAGG_HIP(hipGetLastError());
    d->den_host.resize((size_t)d->R);
    WAGG_HIP(hipMemcpy(d->den_host.data(), d->den64.p, sizeof(double) * (size_t)d->R, hipMemcpyDeviceToHost));
    return WAGG_OK;
}

static int pick_ksplit(int64_t items, int64_t G) {
    int best = 8;
    double best_eff = 0.0;
    for (int S = 8; S <= 64; S += 8) {
        if (S > 8 && G / S < 64 * 16) break;           // keep >= 64 k-steps per slice
        const double w = (double)items * S / 256.0;
        const double eff = w / std::ceil(w);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = S; }
        if (eff >= 0.985) break;
    }
    return best;
}

}  // namespace wagg

extern "C" int wagg_dense_create_synth(int64_t G, int32_t R, uint32_t seed, wagg_dense **out) {
    using namespace wagg;
    int rc = dense_alloc(G, R, out);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    hipLaunchKernelGGL(dense_synth_w_kernel, dim3(256 * 32), dim3(256), 0, nullptr, d->W.p, G, R, d->ldw, seed);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) rc = dense_finish_den(d); else { set_error("synth launch: %s", hipGetErrorString(e)); rc = WAGG_EHIP; }
    if (rc != WAGG_OK) { delete d; *out = nullptr; }
    return rc;
}

extern "C" int wagg_dense_create_host(const float *W_host, int64_t G, int32_t R, wagg_dense **out) {
    using namespace wagg;
    WAGG_REQUIRE(W_host != nullptr, "W_host is NULL");
    int rc = dense_alloc(G, R, out);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    hipError_t e = hipMemset(d->W.p, 0, sizeof(float) * (size_t)(G * d->ldw));
    if (e == hipSuccess)
        e = hipMemcpy2D(d->W.p, sizeof(float) * (size_t)d->ldw, W_host, sizeof(float) * (size_t)R,
                        sizeof(float) * (size_t)R, (size_t)G, hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("dense upload: %s", hipGetErrorString(e)); delete d; *out = nullptr; return WAGG_EHIP; }
    rc = dense_finish_den(d);
    if (rc != WAGG_OK) { delete d; *out = nullptr; }
    return rc;
}

extern "C" int wagg_dense_create_from_segments(const int32_t *cell_idx, const int32_t *region_code,
                                               const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                               wagg_dense **out) {
    using namespace wagg;
    WAGG_REQUIRE(nseg == 0 || (cell_idx && region_code && w_eff), "NULL segment arrays");
    struct Seg { int32_t region, cell; double w; };
    std::vector<Seg> segs;
    std::vector<double> den((size_t)(R > 0 ? R : 0), 0.0);
    for (int64_t i = 0; i < nseg; ++i) {
        const int32_t r = region_code[i];
        if (r < 0) continue;
        WAGG_REQUIRE(r < R && cell_idx[i] >= 0 && cell_idx[i] < G, "segment %lld out of range", (long long)i);
        if (std::isnan(w_eff[i])) continue;
        den[(size_t)r] += w_eff[i];
        segs.push_back({r, cell_idx[i], w_eff[i]});
    }
    std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) {
        return a.region != b.region ? a.region < b.region : a.cell < b.cell; });
    std::vector<int32_t> hc, hr; std::vector<float> hw;
    for (size_t i = 0; i < segs.size();) {
        double s = 0; size_t j = i;
        while (j < segs.size() && segs[j].region == segs[i].region && segs[j].cell == segs[i].cell) s += segs[j++].w;
        hc.push_back(segs[i].cell); hr.push_back(segs[i].region); hw.push_back((float)s);
        i = j;
    }
    int rc = dense_alloc(G, R, out);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    DevBuf<int32_t> dc, dr; DevBuf<float> dw;
    hipError_t e = hipMemset(d->W.p, 0, sizeof(float) * (size_t)(G * d->ldw));
    if (e == hipSuccess) e = dc.upload(hc);
    if (e == hipSuccess) e = dr.upload(hr);
    if (e == hipSuccess) e = dw.upload(hw);
    if (e == hipSuccess && !hc.empty()) {
        hipLaunchKernelGGL(dense_scatter_kernel, dim3((unsigned)((hc.size() + 255) / 256)), dim3(256), 0, nullptr,
                           d->W.p, d->ldw, dc.p, dr.p, dw.p, (int64_t)hc.size());
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { set_error("densify: %s", hipGetErrorString(e)); delete d; *out = nullptr; return WAGG_EHIP; }
    // denominators from the fp64 segment sums (aggregations.py:79), not from the fp32 matrix
    std::vector<float> den32(den.size());
    for (size_t i = 0; i < den.size(); ++i) den32[i] = (float)den[i];
    e = hipMemcpy(d->den64.p, den.data(), sizeof(double) * den.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d->den32.p, den32.data(), sizeof(float) * den32.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) { set_error("densify den: %s", hipGetErrorString(e)); delete d; *out = nullptr; return WAGG_EHIP; }
    d->den_host = den;
    return WAGG_OK;
}

extern "C" int wagg_dense_destroy(wagg_dense *d) {
    delete d;
    return WAGG_OK;
}

extern "C" int wagg_dense_get_den(const wagg_dense *d, double *den_host) {
    WAGG_REQUIRE(d && den_host, "NULL argument");
    std::memcpy(den_host, d->den_host.data(), sizeof(double) * (size_t)d->R);
    return WAGG_OK;
}

extern "C" int wagg_dense_apply_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx,
                                    float *out_dev, int64_t ldo, int ksplit, void *stream) {
    using namespace wagg;
    WAGG_REQUIRE(d != nullptr, "dense plan is NULL");
    WAGG_REQUIRE(T >= 0, "T < 0");
    if (T == 0) return WAGG_OK;
    WAGG_REQUIRE(X_dev && out_dev, "X/out is NULL");
    WAGG_REQUIRE(ldx >= d->G && ldo >= d->R, "ldx/ldo too small");
    WAGG_REQUIRE(ksplit >= 0 && ksplit % 8 == 0, "ksplit must be 0 or a multiple of 8");
    const int n_nt = (int)(d->ldw / D_BN);
    const int n_mb = (int)((T + D_BM - 1) / D_BM);
    const int S = ksplit ? ksplit : pick_ksplit((int64_t)n_nt * n_mb, d->G);
    constexpr int BK = 16;
    const int64_t k_per_slice = ((d->G + S - 1) / S + BK - 1) / BK * BK;
    const int64_t nblk = (int64_t)n_nt * n_mb * S;
    WAGG_REQUIRE(nblk < (int64_t)0x7fffffff && T <= 65535, "grid too large");
    const size_t need = (size_t)nblk * D_BM * D_BN;
    if (d->slabs.n < need) WAGG_HIP(d->slabs.alloc(need));   // first call (or larger T) only
    const bool aligned = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X_dev) & 15) == 0);
    const size_t shmem = sizeof(float) * 2 * DenseCfg<BK>::STAGE;
    hipStream_t st = (hipStream_t)stream;
    auto kern = aligned ? dense_mfma_kernel<BK, true> : dense_mfma_kernel<BK, false>;
    WAGG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    profile_mark(st, true);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(D_THREADS), shmem, st, X_dev, T, ldx, d->W.p, d->ldw,
                       d->G, n_nt, n_mb, S, k_per_slice, d->slabs.p);
    profile_mark(st, false);
    WAGG_HIP(hipGetLastError());
    hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((d->R + 255) / 256), (unsigned)T), dim3(256), 0, st,
                       d->slabs.p, n_nt, S, T, d->R, d->den32.p, out_dev, ldo);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}
