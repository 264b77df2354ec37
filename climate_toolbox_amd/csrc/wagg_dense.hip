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
#include <cstdlib>
#include <type_traits>

#include "wagg_common.h"

namespace wagg {

constexpr int D_MT = 23;            // 16-row MFMA tiles per workgroup
constexpr int D_BM = D_MT * 16;     // 368 rows
constexpr int D_BN = 128;           // 8 waves x 16 columns
constexpr int D_BK = 32;            // k depth of one LDS tile = 8 MFMA k-steps
constexpr int D_KS = D_BK / 4;
constexpr int D_LDA = 370;          // words per k-row of the X image (370 % 32 == 2, see below)
constexpr int D_LDB = 132;          // words per k-row of the W image (16-byte aligned rows)
constexpr int D_THREADS = 512;
constexpr int D_STAGE = D_BK * D_LDA + D_BK * D_LDB;          // floats per LDS buffer (64,256 B)
constexpr int D_XQ = D_BM * (D_BK / 4);                       // 16-byte pieces of the X tile (2944)
constexpr int D_XLOADS = (D_XQ + D_THREADS - 1) / D_THREADS;  // 6 per thread
constexpr int D_WLOADS = D_BK * D_BN / 4 / D_THREADS;         // 2 per thread

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float nan0(float v) { return v == v ? v : 0.0f; }

// LDS images (both k-major):  xs[k][row] with row stride 370 words, ws[k][col] with 132.
//  * X is fetched as whole 128-byte lines: 8 consecutive lanes take the 8 16-byte pieces of one
//    row's 32 k-values (8 rows per wave instruction).  Piece q of row r is stored as 4
//    ds_write_b32 to xs[4q+c][r]: bank = (8q + 2c + r) mod 32 -> each bank is hit by exactly two
//    lanes of a 32-lane half (2-way is free for ds_write_b32).
//  * MFMA k-step s (0..7) takes k = 8*kq + s for lane group kq = lane>>4 (any 4 distinct k per
//    step work as long as A and B agree).  The two lane groups of a 32-lane half then read rows
//    8 apart: 8*370 mod 32 = 16 -> lanes 0-15 and 16-31 sit on disjoint banks, no conflict.
//
// DBG is a diagnostic knob (WAGG_DENSE_DBG env, never set in production): bit0 = skip the global
// loads of the k-loop, bit1 = skip the LDS restage, bit2 = skip the per-tile barrier.  Results are
// wrong with any bit set; only the timing is of interest.
template <bool ALIGNED, int DBG = 0>
__global__ __launch_bounds__(D_THREADS, 2) void dense_mfma_kernel(
    const float *__restrict__ X, int64_t Ttot, int64_t ldx, const float *__restrict__ W,
    int64_t ldw, int64_t G, int n_nt, int n_mb, int S, int64_t k_per_slice,
    float *__restrict__ slabs) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][D_STAGE]

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
    // G here is the part of the gridcell axis that is a whole number of LDS tiles; the ragged
    // remainder (< 32 cells) is contracted by dense_ktail_kernel into one extra slab
    const int64_t k_end = (k_begin + k_per_slice < G) ? k_begin + k_per_slice : G;
    const int64_t klen = k_end > k_begin ? k_end - k_begin : 0;
    const int ntiles = (int)(klen / D_BK);
    const int64_t n0 = (int64_t)nt * D_BN;
    const int64_t m0 = (int64_t)mb * D_BM;

    // staging coordinates: wave-uniform 64-bit bases + 32-bit per-lane byte offsets (saddr form),
    // LDS offsets that differ between a thread's pieces only by immediates.  Rows past T are
    // clamped to the last row: their accumulators hold garbage that the reduce kernel never
    // reads, and the loads need no row predicate.
    const int xrow0 = tid >> 3, xq = tid & 7;          // piece i of this thread: row xrow0 + 64 i
    unsigned xvoff[D_XLOADS];
#pragma unroll
    for (int i = 0; i < D_XLOADS; ++i) {
        int64_t grow = m0 + xrow0 + 64 * i;
        grow = grow < Ttot ? grow : Ttot - 1;
        xvoff[i] = (unsigned)(((grow - m0) * ldx + xq * 4) * 4);
    }
    const bool x_last_ok = tid + D_THREADS * (D_XLOADS - 1) < D_XQ;    // piece 5 exists for tid < 384
    const int xoff0 = (xq * 4) * D_LDA + xrow0;        // + 64 i (+ c * LDA)
    const int wrow0 = tid >> 5, wc4 = tid & 31;        // piece i: k-row wrow0 + 16 i
    const unsigned wvoff = (unsigned)((wrow0 * ldw + wc4 * 4) * 4);
    const int woff0 = D_BK * D_LDA + wrow0 * D_LDB + wc4 * 4;          // + 16 i * LDB
    const char *xbase = reinterpret_cast<const char *>(X + m0 * ldx + k_begin);      // uniform
    const char *wbase = reinterpret_cast<const char *>(W + k_begin * ldw + n0);      // uniform

    f32x4 acc[D_MT];
#pragma unroll
    for (int m = 0; m < D_MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging registers are NAMED scalars (not arrays): arrays indexed inside the unrolled MFMA
    // stream end up in scratch memory with hipcc (ROCm 7.2)
    f32x4 xr0, xr1, xr2, xr3, xr4, xr5, wr0, wr1;
    static_assert(D_XLOADS == 6 && D_WLOADS == 2, "staging registers are named for 6 + 2 pieces");
#define WAGG_XPTR(i, tile) (xbase + (int64_t)(tile) * (D_BK * 4) + xvoff[i])
#define WAGG_WPTR(i, tile) (wbase + ((int64_t)(tile) * D_BK + 16 * (i)) * ldw * 4 + wvoff)
#define WAGG_LOAD_X(i, tile)                                                                     \
    do {                                                                                         \
        if (ALIGNED) xr##i = *reinterpret_cast<const f32x4 *>(WAGG_XPTR(i, tile));               \
        else { const float *f_ = reinterpret_cast<const float *>(WAGG_XPTR(i, tile));            \
               xr##i = f32x4{f_[0], f_[1], f_[2], f_[3]}; }                                      \
    } while (0)
#define WAGG_LOAD_W(i, tile)                                                                     \
    do {                                                                                         \
        if (DBG & 32) wr##i = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(WAGG_WPTR(i, tile))); \
        else wr##i = *reinterpret_cast<const f32x4 *>(WAGG_WPTR(i, tile));                        \
    } while (0)
#define WAGG_STORE_X(i, buf)                                                                     \
    do {                                                                                         \
        float *xs_ = lds + (buf) * D_STAGE + xoff0 + 64 * (i);                                   \
        if ((i) + 1 < D_XLOADS || x_last_ok) {                                                   \
            xs_[0] = nan0(xr##i[0]); xs_[D_LDA] = nan0(xr##i[1]);          /* S6 */              \
            xs_[2 * D_LDA] = nan0(xr##i[2]); xs_[3 * D_LDA] = nan0(xr##i[3]);                    \
        }                                                                                        \
    } while (0)
#define WAGG_STORE_W(i, buf)                                                                     \
    *reinterpret_cast<f32x4 *>(lds + (buf) * D_STAGE + woff0 + 16 * (i) * D_LDB) = wr##i
#define WAGG_LOAD_ALL(tile)                                                                      \
    do { WAGG_LOAD_X(0, tile); WAGG_LOAD_X(1, tile); WAGG_LOAD_X(2, tile); WAGG_LOAD_X(3, tile); \
         WAGG_LOAD_X(4, tile); WAGG_LOAD_X(5, tile); WAGG_LOAD_W(0, tile); WAGG_LOAD_W(1, tile); } while (0)
#define WAGG_STORE_ALL(buf)                                                                      \
    do { WAGG_STORE_X(0, buf); WAGG_STORE_X(1, buf); WAGG_STORE_X(2, buf); WAGG_STORE_X(3, buf); \
         WAGG_STORE_X(4, buf); WAGG_STORE_X(5, buf); WAGG_STORE_W(0, buf); WAGG_STORE_W(1, buf); } while (0)

    xr0 = xr1 = xr2 = xr3 = xr4 = xr5 = wr0 = wr1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ntiles > 0) {
        WAGG_LOAD_ALL(0);
        WAGG_STORE_ALL(0);
        if (ntiles > 1) WAGG_LOAD_ALL(1);
    }
    __syncthreads();

    const int a_lane = (8 * kq) * D_LDA + lr;                    // + s*LDA + m*16 (immediates)
    const int b_lane = D_BK * D_LDA + (8 * kq) * D_LDB + wave * 16 + lr;

    // One LDS tile = 8 MFMA k-steps = 184 MFMAs per wave.  A fragments run D_AHEAD MFMAs ahead of
    // their use (a rolling window of ~8 VGPRs instead of a 46-register double buffer); the
    // fragment reads and (PF) the next tile's global loads are spread between the MFMAs -- one
    // small group after every second MFMA, pinned with sched_barriers (hipcc otherwise sinks each
    // ds_read next to its MFMA and waits lgkmcnt(0) between every pair, and a burst of reads per
    // k-step leaves the matrix pipe idle when both waves of a SIMD burst together).
    constexpr int D_AHEAD = 6;
    constexpr int NMF = D_KS * D_MT;
    // MODE 2: inside tile t's MFMA stream, store the staged registers (tile t+1) into the other
    //         LDS buffer and refill each register with its piece of tile t+2 right after;
    // MODE 1: store only (t+2 does not exist);  MODE 0: neither (last tile).
    // Loads therefore run a whole tile (~5 us) ahead of their LDS store, and the LDS stores
    // (ds_write_b32 runs at 64 B/clk/CU: ~950 cycles per tile) hide under the MFMAs.
    auto tile_body = [&](auto mode_tag, int tile) {
        constexpr int MODE = decltype(mode_tag)::value;
        const float *xs = lds + (tile & 1) * D_STAGE;
        const int nbuf = (tile & 1) ^ 1;
        float a[8], b[2];                                  // rolling windows, static indices
        b[0] = xs[b_lane];
#pragma unroll
        for (int jj = 0; jj < D_AHEAD; ++jj) a[jj] = xs[a_lane + (jj / D_MT) * D_LDA + (jj % D_MT) * 16];
#pragma unroll
        for (int s4 = 0; s4 < D_KS; ++s4) {
#pragma unroll
          for (int m = 0; m < D_MT; ++m) {
            const int j = s4 * D_MT + m;
            if ((j & 1) == 0) {
                __builtin_amdgcn_sched_barrier(0);
                if (j + D_AHEAD < NMF)
                    a[(j + D_AHEAD) & 7] = xs[a_lane + ((j + D_AHEAD) / D_MT) * D_LDA + ((j + D_AHEAD) % D_MT) * 16];
                if (j + D_AHEAD + 1 < NMF)
                    a[(j + D_AHEAD + 1) & 7] =
                        xs[a_lane + ((j + D_AHEAD + 1) / D_MT) * D_LDA + ((j + D_AHEAD + 1) % D_MT) * 16];
                if ((m == 0 || m == 1) && s4 + 1 < D_KS) b[(s4 + 1) & 1] = xs[b_lane + (s4 + 1) * D_LDB];
                if (MODE >= 1 && (m == 4 || m == 5) && !(DBG & 2)) {      // one piece per k-step: store ...
                    if (s4 == 0) WAGG_STORE_X(0, nbuf);
                    if (s4 == 1) WAGG_STORE_X(1, nbuf);
                    if (s4 == 2) WAGG_STORE_X(2, nbuf);
                    if (s4 == 3) WAGG_STORE_X(3, nbuf);
                    if (s4 == 4) WAGG_STORE_X(4, nbuf);
                    if (s4 == 5) WAGG_STORE_X(5, nbuf);
                    if (s4 == 6) WAGG_STORE_W(0, nbuf);
                    if (s4 == 7) WAGG_STORE_W(1, nbuf);
                }
                if (MODE == 2 && (m == 10 || m == 11) && !(DBG & 1)) {    // ... then refill its register
                    if (s4 == 0 && !(DBG & 8)) WAGG_LOAD_X(0, tile + 2);
                    if (s4 == 1 && !(DBG & 8)) WAGG_LOAD_X(1, tile + 2);
                    if (s4 == 2 && !(DBG & 8)) WAGG_LOAD_X(2, tile + 2);
                    if (s4 == 3 && !(DBG & 8)) WAGG_LOAD_X(3, tile + 2);
                    if (s4 == 4 && !(DBG & 8)) WAGG_LOAD_X(4, tile + 2);
                    if (s4 == 5 && !(DBG & 8)) WAGG_LOAD_X(5, tile + 2);
                    if (s4 == 6) WAGG_LOAD_W(0, tile + 2);
                    if (s4 == 7) WAGG_LOAD_W(1, tile + 2);
                }
                if ((DBG & 64) && MODE == 2 && (m == 16 || m == 17)) {    // experiment: stagger waves 4-7
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 7], b[s4 & 1], acc[m], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
    };

    // single hot body in the loop; the two drain tiles are peeled (merging differently shaped
    // bodies at a loop join makes hipcc copy and spill the accumulators)
    int tile = 0;
    for (; tile + 2 < ntiles; ++tile) {
        tile_body(std::integral_constant<int, 2>{}, tile);
        if (!(DBG & 4)) __syncthreads();
    }
    if (tile + 1 < ntiles) {
        tile_body(std::integral_constant<int, 1>{}, tile);
        __syncthreads();
        ++tile;
    }
    if (tile < ntiles) tile_body(std::integral_constant<int, 0>{}, tile);

    // C/D map of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
    float *slab = slabs + ((((int64_t)mb * n_nt + nt) * (S + 1) + ks) * D_BM) * D_BN;
#pragma unroll
    for (int m = 0; m < D_MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            slab[(m * 16 + kq * 4 + r) * D_BN + wave * 16 + lr] = acc[m][r];
}

// ragged remainder of the gridcell axis (k in [Gfull, G), fewer than 32 cells): plain FMA into the
// extra slab S of every (row block, column tile); zero when there is no remainder
__global__ void dense_ktail_kernel(const float *__restrict__ X, int64_t Ttot, int64_t ldx,
                                   const float *__restrict__ W, int64_t ldw, int64_t Gfull, int64_t G,
                                   int n_nt, int S, float *__restrict__ slabs) {
    const int c = threadIdx.x & (D_BN - 1);
    const int nt = blockIdx.x;
    const int64_t t = (int64_t)blockIdx.y * 2 + (threadIdx.x >> 7);     // 256 threads = 2 rows x 128 cols
    const int mb = (int)(t / D_BM), tl = (int)(t % D_BM);
    if (t >= (int64_t)gridDim.y * 2) return;
    float s = 0.f;
    if (t < Ttot) {
        for (int64_t k = Gfull; k < G; ++k)
            s = fmaf(nan0(X[t * ldx + k]), W[k * ldw + (int64_t)nt * D_BN + c], s);
    }
    slabs[((((int64_t)mb * n_nt + nt) * (S + 1) + S) * D_BM + tl) * D_BN + c] = s;
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
    const float *p = slabs + ((((int64_t)mb * n_nt + nt) * (S + 1)) * D_BM + tl) * D_BN + c;
    float s = 0.f;
    for (int k = 0; k < S + 1; ++k) s += p[(int64_t)k * D_BM * D_BN];     // slab S = ragged k tail
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
        if (S > 8 && G / S < 32 * D_BK) break;         // keep >= 32 LDS tiles per slice
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
    constexpr int BK = D_BK;
    const int64_t Gfull = d->G / BK * BK;
    const int64_t k_per_slice = ((Gfull + S - 1) / S + BK - 1) / BK * BK;
    const int64_t nblk = (int64_t)n_nt * n_mb * S;
    WAGG_REQUIRE(nblk < (int64_t)0x7fffffff && T <= 65535, "grid too large");
    if ((int64_t)D_BM * ldx * 4 >= ((int64_t)1 << 32)) {
        set_error("dense path: 368 rows x ldx x 4 bytes must stay below 4 GiB (ldx = %lld)", (long long)ldx);
        return WAGG_EUNSUPPORTED;
    }
    const size_t need = (size_t)n_nt * n_mb * (S + 1) * D_BM * D_BN;
    if (d->slabs.n < need) WAGG_HIP(d->slabs.alloc(need));   // first call (or larger T) only
    const bool aligned = (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(X_dev) & 15) == 0);
    const size_t shmem = sizeof(float) * 2 * D_STAGE;
    hipStream_t st = (hipStream_t)stream;
    auto kern = aligned ? dense_mfma_kernel<true> : dense_mfma_kernel<false>;
    if (const char *dbg = getenv("WAGG_DENSE_DBG")) {
        switch (atoi(dbg)) {
            case 1: kern = dense_mfma_kernel<true, 1>; break;
            case 2: kern = dense_mfma_kernel<true, 2>; break;
            case 3: kern = dense_mfma_kernel<true, 3>; break;
            case 7: kern = dense_mfma_kernel<true, 7>; break;
            case 8: kern = dense_mfma_kernel<true, 8>; break;
            case 32: kern = dense_mfma_kernel<true, 32>; break;
            default: break;
        }
    }
    WAGG_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    profile_mark(st, true);
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(D_THREADS), shmem, st, X_dev, T, ldx, d->W.p, d->ldw,
                       Gfull, n_nt, n_mb, S, k_per_slice, d->slabs.p);
    profile_mark(st, false);
    hipLaunchKernelGGL(dense_ktail_kernel, dim3((unsigned)n_nt, (unsigned)(n_mb * D_BM / 2)), dim3(256), 0, st,
                       X_dev, T, ldx, d->W.p, d->ldw, Gfull, d->G, n_nt, S, d->slabs.p);
    WAGG_HIP(hipGetLastError());
    hipLaunchKernelGGL(dense_reduce_kernel, dim3((unsigned)((d->R + 255) / 256), (unsigned)T), dim3(256), 0, st,
                       d->slabs.p, n_nt, S, T, d->R, d->den32.p, out_dev, ldo);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}
