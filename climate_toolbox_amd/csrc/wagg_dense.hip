// Dense path: out = (nan0(X) . W) / (1^T W) with W a (gridcell x region) fp32 matrix resident in
// HBM (c2-dense: 1,036,800 x 24,378 = 101 GB).  This is the algebraic form of
// aggregations.py:78-80 when every (cell, region) pair carries a weight.
//
// Shape: M = T (365) is skinny, K = G (1e6) is huge, N = R (24k).  fp32 MFMA runs at the vector
// rate (256 flop/clk/CU), so the contraction is MFMA-issue-bound by ~7x over HBM; the design
// spends nothing on bandwidth tricks and everything on keeping the matrix pipe fed:
//   * one workgroup (8 waves, 2 per SIMD, all 160 KB of LDS) owns ALL 365 rows (23 x 16 = 368,
//     0.8 % padding) x 256 columns of the output, so every W element leaves HBM exactly once and
//     the X panel is re-read only once per 256 columns; wave w owns columns [32w, 32w+32):
//     46 independent 16x16 accumulators (184 registers);
//   * both operands are kept in HBM in the exact byte order of their LDS tile image
//     ("packed": W at plan time, X by dense_pack_x_kernel at the start of every apply, which also
//     applies NaN -> 0 (S6) and zero-pads rows/cells), so a tile is filled by 1-KiB LDS-DMA pieces
//     (global_load_lds_dwordx4: contiguous 1 KiB in HBM -> contiguous 1 KiB in LDS, no staging
//     registers, no ds_write) -- 10 per wave per tile against 368 MFMAs;
//   * tile image = [row][8 x 16-byte pieces] (32 k-values per row), piece p stored at position
//     p ^ ((row >> 1) & 7): every ds_read_b128 fragment read is bank-conflict free (see the lane
//     groups of ds_read_b128 on gfx950), and one b128 read feeds four MFMA k-steps;
//   * K is split into S slices (multiple of 8): blocks with equal blockIdx % 8 (one XCD under
//     round-robin placement -- speed only) walk the same k-slice over neighbouring column tiles,
//     so the 46 KB X tile of each k-step is served by that XCD's L2;
//   * two LDS buffers, ONE barrier per tile: the DMA of tile t+1 is issued piecewise inside tile
//     t's MFMA stream and has ~10 us to land;
//   * fp32 partial slabs per (tile, k-slice), then one reduce kernel fuses the division by den[r]
//     (deterministic, no atomics).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdlib>
#include <type_traits>

#include "wagg_dense_int.h"
#include "wagg_entry.h"

namespace wagg {

constexpr int D_MT = 23;                 // most 16-row MFMA tiles per workgroup (fp32; T = 365 -> 23)
constexpr int D_BN = 256;                // 8 waves x 32 columns
constexpr int D_ROWB = 128;              // bytes of one tile row = 8 pieces of 16 bytes: 32 floats or 16 doubles of k
constexpr int D_THREADS = 512;
constexpr int D_WTB = D_BN * D_ROWB;     // bytes per packed W tile (32,768 B)
constexpr int D_WSLOTS = D_WTB / 16;     // 16-byte slots per W tile (2,048)
static_assert(D_WTB / 1024 == 32, "4 one-KiB W pieces per wave");
// a workgroup owns MT x 16 rows (chosen per T so that the row blocks are evenly filled)
constexpr int d_xt_bytes(int mt) { return mt * 16 * D_ROWB; }             // packed X tile (MT = 23: 47,104 B)
constexpr int d_buf_bytes(int mt) { return d_xt_bytes(mt) + D_WTB; }      // per LDS buffer, two buffers (MT = 23: 79,872 B)
static_assert(2 * d_buf_bytes(D_MT) <= 160 * 1024, "LDS budget");

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1))) *gptr_t;
typedef void __attribute__((address_space(3))) *lptr_t;

// Everything below is written once for both data types.  The tile geometry is the same in BYTES
// (rows of 8 x 16-byte pieces, 256-column W tiles, the DMA schedule, the bank-conflict-free piece
// swizzle); what changes with the type is the k depth of a tile (32 floats / 16 doubles), the
// matrix instruction (v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64: 4 k per instruction either
// way, half the rate in fp64), the accumulator size (4 / 8 registers per 16x16 block, hence at most
// 23 / 11 row blocks per workgroup) and the C/D register -> row map.
template <typename T> struct DT;
template <> struct DT<float> {
    typedef f32x4 vec;                                   // one 16-byte piece
    typedef f32x4 acc;
    static constexpr int EPP = 4, BK = 32, MT_MAX = 23;  // elements per piece, k per tile, row blocks
    static __device__ __forceinline__ acc mfma(float a, float b, acc c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int kq, int reg) { return kq * 4 + reg; }   // C/D: row = 4 (lane >> 4) + reg
};
template <> struct DT<double> {
    typedef f64x2 vec;
    typedef f64x4 acc;
    static constexpr int EPP = 2, BK = 16, MT_MAX = 11;
    static __device__ __forceinline__ acc mfma(double a, double b, acc c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int kq, int reg) { return kq + 4 * reg; }   // f64 C/D: row = (lane >> 4) + 4 reg
};

// Position of (row, piece p) inside a packed tile, in 16-byte slots.  The same involution is
// applied by the packers (source side) and by the fragment reads.
__host__ __device__ __forceinline__ int tile_slot(int row, int p) { return row * 8 + (p ^ ((row >> 1) & 7)); }

// MFMA k mapping: lane group kq = lane >> 4, fragment read h (0/1) fetches piece p = kq + 4h, i.e.
// k = 4 kq + 16 h + c (c = 0..3); MFMA k-step s = 4h + c then sums k in {c, 4+c, 8+c, 12+c} + 16h.
// A and B use the same mapping, so any permutation of k inside the tile is fine.
//
// Bank check for ds_read_b128 (banks = (byte/4) % 64, lane groups {0-3,12-15,20-27},
// {4-11,16-19,28-31} and their +32 mirrors): a group holds rows i in {0-3,12-15} of one kq and
// rows {4-11} of the next; with f = i >> 1 the 16-byte positions (i & 1) * 8 + ((kq + 4h) ^ f)
// are all distinct within each group.
//
// DBG is a diagnostic knob (WAGG_DENSE_DBG env, read by the -DWAGG_DIAG build only): bit0 = skip the LDS-DMA
// of the k-loop, bit2 = skip the per-tile barrier, bit3 = both waves of a SIMD issue their DMA
// pieces at the same point, bit4 = X pieces before the W pieces.  Results are wrong with bit0 or
// bit2 set.
//
// TILED (tile-sparse W, e.g. c5 "block-local" weights): only the non-empty (32-cell x 256-region)
// tiles of W are stored, compacted per column tile; tile_kt[i] is the k-tile (= X tile) of stored
// tile i.  Since round 5 tile-sparse plans run dense_pieces_kernel (below: the same body, a workgroup
// walks pieces of the stored-tile runs); the TILED branch of dense_mfma_kernel is what rounds 2-4 ran
// (blocks of equal k-slices) and is no longer instantiated.  The k index of tile t+2 is fetched by a plain
// vector load at the start of tile t (it retires in order ahead of the DMA pieces) and moved to an SGPR
// after the end-of-tile wait.
//
// RM (tile-sparse only, "pack-free"): Xp is the caller's ROW-MAJOR X (row stride ldxB bytes, 16-byte aligned,
// no NaN -> 0 pass): the LDS-DMA builds the same tile image straight from it -- lane l of 1-KiB piece q
// fetches piece ((l & 7) ^ (l >> 4) ^ 4 (q & 1)) of row 8 q + (l >> 3), i.e. eight whole 128-byte lines per
// instruction; rows >= T repeat row T - 1 (their results are never read).  A NaN or +-inf in X then
// reaches the accumulators; dense_reduce_kernel notices the non-finite numerator and the launcher's gated
// second pass (pack + the packed kernel, `gate` != 0) redoes the apply the exact way.
template <typename T, int DBG = 0, bool TILED = false, int MT = D_MT, bool RM = false>
__global__ __launch_bounds__(D_THREADS, 2) void dense_mfma_kernel(
    const T *__restrict__ Xp, const T *__restrict__ Wp, int n_kt, int n_nt, int n_mb, int S,
    int kt_per_slice, T *__restrict__ slabs, const int32_t *__restrict__ tile_kt = nullptr,
    const int32_t *__restrict__ tile_off = nullptr, int64_t ldxB = 0, int Tn = 0, const int *__restrict__ gate = nullptr) {
#define WAGG_PIECES 0
#include "wagg_dense_kernel.inc"
#undef WAGG_PIECES
}

// The tile-sparse form's kernel (round 5): the same body, a workgroup walks PIECES.  The stored tiles of all (row block,
// column tile) pairs, laid end to end, are cut into one equal share per workgroup -- a piece is the part of one pair's run
// that falls into a share -- so every CU gets the same number of tiles whatever the pair count (rounds 2-4: n_nt x n_mb x S
// blocks of equal k-slices; 10.5 rounds of 256 on c5-block, the last one half empty).  `tile_off` is the piece table of the
// launch (csrc/wagg_dense.hip: tile_pieces_for); n_kt / n_mb / S / kt_per_slice are unused.
template <typename T, int MT, bool RM>
__global__ __launch_bounds__(D_THREADS, 2) void dense_pieces_kernel(
    const T *__restrict__ Xp, const T *__restrict__ Wp, int n_kt, int n_nt, int n_mb, int S,
    int kt_per_slice, T *__restrict__ slabs, const int32_t *__restrict__ tile_kt, const int32_t *__restrict__ tile_off,
    int64_t ldxB, int Tn, const int *__restrict__ gate) {
    constexpr bool TILED = true;
    constexpr int DBG = 0;
    (void)n_kt; (void)n_mb; (void)S; (void)kt_per_slice;
#define WAGG_PIECES 1
#include "wagg_dense_kernel.inc"
#undef WAGG_PIECES
}

// X (T x G, row stride ldx) -> packed tiles Xp[mb][kt][slot] (one 16-byte piece per slot; a row block has
// bm = 16 MT rows), transform (tas_poly / snyder_edd), NaN -> 0 (S6), zero for rows >= T and cells >= G.
// One thread per slot; the 8 slots of a row read one 128-byte line of X.
template <typename T>
__global__ void dense_pack_x_kernel(const T *__restrict__ X, int64_t Tn, int64_t ldx, int64_t G,
                                    int n_kt, int bm, int64_t n_slots, int aligned, typename DT<T>::vec *__restrict__ Xp,
                                    PackXfT<T> xf, int *__restrict__ inf_flag, const int *__restrict__ gate = nullptr) {
    typedef typename DT<T>::vec vec_t;
    constexpr int E = DT<T>::EPP;
    if (gate != nullptr && *gate == 0) return;              // second pass of a pack-free apply: not needed
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int tile_slots = bm * 8;                         // bm rows x 8 pieces of 16 bytes
    bool inf_seen = false;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
        const int slot = (int)(s % tile_slots);
        const int64_t tk = s / tile_slots;
        const int kt = (int)(tk % n_kt);
        const int64_t mb = tk / n_kt;
        const int row = slot >> 3, p = (slot & 7) ^ ((row >> 1) & 7);
        const int64_t t = mb * bm + row, k0 = (int64_t)kt * DT<T>::BK + E * p;
        vec_t v, h;
#pragma unroll
        for (int c = 0; c < E; ++c) { v[c] = T(0); h[c] = T(0); }
        if (t < Tn) {
            const T *src = X + t * ldx + k0;
            const T *src2 = xf.mode == XF_EDD ? xf.X2 + t * ldx + k0 : src;
            if (aligned && k0 + E <= G) {
                v = *reinterpret_cast<const vec_t *>(src);
                if (xf.mode == XF_EDD) h = *reinterpret_cast<const vec_t *>(src2);
            } else {
#pragma unroll
                for (int c = 0; c < E; ++c) if (k0 + c < G) { v[c] = src[c]; h[c] = src2[c]; }
            }
            // transform, then NaN -> 0 (S6); cells >= G stay 0
#pragma unroll
            for (int c = 0; c < E; ++c) if (k0 + c < G) v[c] = pack_xf<T>(xf, v[c], h[c], inf_seen);
        }
        Xp[s] = v;
    }
    // the MFMA forms multiply every (cell, region) pair of a stored tile: +-inf data would turn the
    // zero weights of other regions into NaN, so the caller is told (wagg_dense_saw_inf)
    if (inf_seen) __hip_atomic_store(inf_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// element index of W[g][r] inside the packed matrix Wp[nt][kt][slot][EPP]
template <typename T>
__host__ __device__ __forceinline__ int64_t wp_index(int64_t g, int64_t r, int n_kt) {
    constexpr int E = DT<T>::EPP, BK = DT<T>::BK;
    const int64_t nt = r / D_BN, kt = g / BK;
    const int cl = (int)(r % D_BN), kk = (int)(g % BK);
    return ((nt * n_kt + kt) * D_WSLOTS + tile_slot(cl, kk / E)) * E + (kk % E);
}

// out[t, r] = sum_s slab[mb][nt][s][t_local][c] / den[r]        (aggregations.py:77-80 fused)
// `nonfinite` (pack-free first pass): set when a numerator is NaN / +-inf; `gate` (second pass): run only if set.
template <typename T>
__global__ void dense_reduce_kernel(const T *__restrict__ slabs, int n_nt, int S, int bm, int64_t Ttot,
                                    int32_t R, const T *__restrict__ den, T *__restrict__ out, int64_t ldo,
                                    int *__restrict__ nonfinite = nullptr, const int *__restrict__ gate = nullptr,
                                    int *__restrict__ sticky = nullptr, const int32_t *__restrict__ slab_first = nullptr) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = blockIdx.y;
    if (gate != nullptr && *gate == 0) return;
    if (r >= R) return;
    const int mb = (int)(t / bm), tl = (int)(t % bm);
    const int nt = (int)(r / D_BN), c = (int)(r % D_BN);
    // slabs of the pair (row block, column tile): S of them in a row, or (tile-sparse launches) slab_first[2 pair + 1] of them
    // from slab_first[2 pair] on -- as many as pieces of the pair's run, added in that (fixed) order
    int64_t first = ((int64_t)mb * n_nt + nt) * S;
    if (slab_first != nullptr) { first = slab_first[2 * (mb * n_nt + nt)]; S = slab_first[2 * (mb * n_nt + nt) + 1]; }
    const T *p = slabs + (first * bm + tl) * D_BN + c;
    T s = T(0);
    for (int k = 0; k < S; ++k) s += p[(int64_t)k * bm * D_BN];
    if (nonfinite != nullptr && !(fabs(s) <= std::numeric_limits<T>::max())) {
        __hip_atomic_store(nonfinite, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (sticky != nullptr) __hip_atomic_store(sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    out[t * ldo + r] = s / den[r];
}

// (slot, element) of a packed W tile -> (column inside the tile, k inside the tile)
template <typename T> __device__ __forceinline__ void slot_to_ck(int slot, int &cl, int &k0) {
    cl = slot >> 3;
    k0 = DT<T>::EPP * ((slot & 7) ^ ((cl >> 1) & 7));
}

// synthetic W[g][r] = hash_u01(g R + r, seed) written straight into the packed order
template <typename T>
__global__ void dense_synth_w_kernel(typename DT<T>::vec *__restrict__ Wp, int64_t G, int32_t R, int n_kt,
                                     int64_t n_slots, uint32_t seed, float fill) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
        const int64_t tk = s / D_WSLOTS;
        const int64_t kt = tk % n_kt, nt = tk / n_kt;
        int cl, kl;
        slot_to_ck<T>((int)(s % D_WSLOTS), cl, kl);
        const int64_t r = nt * D_BN + cl, g0 = kt * DT<T>::BK + kl;
        typename DT<T>::vec v;
#pragma unroll
        for (int c = 0; c < DT<T>::EPP; ++c) {
            const uint64_t id = (uint64_t)(g0 + c) * (uint64_t)R + (uint64_t)r;
            const bool keep = r < R && g0 + c < G && (fill >= 1.0f || hash_u01(id, seed ^ 0x9e3779b9u) < fill);
            v[c] = keep ? (T)hash_u01(id, seed) : T(0);
        }
        Wp[s] = v;
    }
}

// column sums in fp64 (plan time): one block per (column tile, strip of k tiles); thread = slot.  Every slot
// writes its own partial (no atomics: the denominators are the same bits on every build); slot s of a tile
// holds piece s & 7 of column s >> 3, so a column has 8 partials per strip, added in a fixed order below.
template <typename T>
__global__ void dense_colsum_kernel(const typename DT<T>::vec *__restrict__ Wp, int32_t R, int n_kt, int kt_per_block,
                                    double *__restrict__ part /* [strips][n_nt * 2048] */) {
    const int nt = blockIdx.x;
    const int ktb = blockIdx.y * kt_per_block;
    const int kte = ktb + kt_per_block < n_kt ? ktb + kt_per_block : n_kt;
    for (int slot = threadIdx.x; slot < D_WSLOTS; slot += blockDim.x) {
        double s = 0.0;
        for (int kt = ktb; kt < kte; ++kt) {
            const typename DT<T>::vec v = Wp[((int64_t)nt * n_kt + kt) * D_WSLOTS + slot];
            double sv = 0.0;
#pragma unroll
            for (int c = 0; c < DT<T>::EPP; ++c) sv += (double)v[c];
            s += sv;
        }
        part[((int64_t)blockIdx.y * gridDim.x + nt) * D_WSLOTS + slot] = s;
    }
    (void)R;
}

__global__ void dense_colsum_combine_kernel(const double *__restrict__ part, int n_strips, int n_nt, int32_t R,
                                            double *__restrict__ den) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t nt = r / D_BN, c = r % D_BN;
    double s = 0.0;
    for (int st = 0; st < n_strips; ++st)
        for (int p = 0; p < 8; ++p) s += part[((int64_t)st * n_nt + nt) * D_WSLOTS + c * 8 + p];
    den[r] = s;
}

__global__ void dense_den32_kernel(const double *__restrict__ den64, float *__restrict__ den32, int32_t R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) den32[r] = (float)den64[r];
}

template <typename T>
__global__ void dense_scatter_kernel(T *__restrict__ Wp, int n_kt, const int32_t *__restrict__ cell,
                                     const int32_t *__restrict__ region, const T *__restrict__ w, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Wp[wp_index<T>(cell[i], region[i], n_kt)] = w[i];
}

template <typename T>
__global__ void dense_scatter_at_kernel(T *__restrict__ Wp, const int64_t *__restrict__ at,
                                        const T *__restrict__ w, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) Wp[at[i]] = w[i];
}

// synthetic "block-local" weights: every run of 64 cells touches the 256 regions of
// ONE column tile, nt = (97 run) mod n_nt; inside, W[g][r] = hash_u01(g R + r, seed) where a second
// hash is below `fill`, else 0.  Stored tile i holds k tile kt_of[i] of column tile nt_of[i].
template <typename T>
__global__ void dense_synth_blocklocal_kernel(typename DT<T>::vec *__restrict__ Wp, const int32_t *__restrict__ tile_kt,
                                              const int32_t *__restrict__ tile_nt, int64_t n_tiles, int64_t G,
                                              int32_t R, uint32_t seed, float fill) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, n_slots = n_tiles * D_WSLOTS;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
        const int64_t ti = s / D_WSLOTS;
        const int64_t kt = tile_kt[ti], nt = tile_nt[ti];
        int cl, kl;
        slot_to_ck<T>((int)(s % D_WSLOTS), cl, kl);
        const int64_t r = nt * D_BN + cl, g0 = kt * DT<T>::BK + kl;
        typename DT<T>::vec v;
#pragma unroll
        for (int c = 0; c < DT<T>::EPP; ++c) {
            const uint64_t id = (uint64_t)(g0 + c) * (uint64_t)R + (uint64_t)r;
            v[c] = (r < R && g0 + c < G && hash_u01(id, seed ^ 0x9e3779b9u) < fill) ? (T)hash_u01(id, seed) : T(0);
        }
        Wp[s] = v;
    }
}

// column sums of the tile-sparse form: one thread per region walks the stored tiles of its column tile in
// order (tile_first[nt] .. tile_first[nt + 1]) -- fixed summation order, no atomics
template <typename T>
__global__ void dense_colsum_tiled_kernel(const typename DT<T>::vec *__restrict__ Wp, const int32_t *__restrict__ tile_first,
                                          int32_t R, double *__restrict__ den) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const int64_t nt = r / D_BN, c = r % D_BN;
    double s = 0.0;
    for (int64_t ti = tile_first[nt]; ti < tile_first[nt + 1]; ++ti)
        for (int p = 0; p < 8; ++p) {
            const typename DT<T>::vec v = Wp[ti * D_WSLOTS + c * 8 + p];
#pragma unroll
            for (int e = 0; e < DT<T>::EPP; ++e) s += (double)v[e];
        }
    den[r] = s;
}

// plain row-major W (G x R) -> packed order (small matrices handed over by the host)
template <typename T>
__global__ void dense_pack_w_kernel(const T *__restrict__ W, int64_t G, int32_t R, int n_kt,
                                    int64_t n_slots, typename DT<T>::vec *__restrict__ Wp) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_slots; s += stride) {
        const int64_t tk = s / D_WSLOTS;
        const int64_t kt = tk % n_kt, nt = tk / n_kt;
        int cl, kl;
        slot_to_ck<T>((int)(s % D_WSLOTS), cl, kl);
        const int64_t r = nt * D_BN + cl, g0 = kt * DT<T>::BK + kl;
        typename DT<T>::vec v;
#pragma unroll
        for (int c = 0; c < DT<T>::EPP; ++c) v[c] = (r < R && g0 + c < G) ? W[(g0 + c) * R + r] : T(0);
        Wp[s] = v;
    }
}

}  // namespace wagg

namespace wagg {

template <typename T>
static int dense_alloc(int64_t G, int32_t R, wagg_dense **out, int64_t stored_tiles = -1, bool spmm = false) {
    constexpr int BK = DT<T>::BK;
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(G > 0 && R > 0, "bad sizes G=%lld R=%d", (long long)G, R);
    WAGG_REQUIRE((G + BK - 1) / BK < (int64_t)0x7fffffff / 8, "G too large");
    wagg_dense *d = new (std::nothrow) wagg_dense();
    if (!d) { set_error("host allocation failed"); return WAGG_ENOMEM; }
    d->G = G; d->R = R;
    d->f64 = sizeof(T) == 8;
    d->n_kt = (int)((G + BK - 1) / BK);
    d->n_nt = (int)(((int64_t)R + D_BN - 1) / D_BN);
    d->n_tiles = spmm ? 0 : (stored_tiles >= 0 ? stored_tiles : (int64_t)d->n_nt * d->n_kt);
    d->tiled = !spmm && stored_tiles >= 0;
    d->spmm = spmm;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    d->device = dev;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&d->ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e == hipSuccess) e = d->nonfinite.alloc(1);
    if (e == hipSuccess) e = hipHostMalloc((void **)&d->inf_host, 2 * sizeof(int), hipHostMallocMapped);
    if (e == hipSuccess) { d->inf_host[0] = d->inf_host[1] = 0; e = hipHostGetDevicePointer((void **)&d->inf_dev, d->inf_host, 0); }
    if (e == hipSuccess) e = d->W.alloc((size_t)(d->w_slots() > 0 ? d->w_slots() : 1) * 4);
    if (e == hipSuccess) e = d->den32.alloc((size_t)R);
    if (e == hipSuccess) e = d->den64.alloc((size_t)R);
    if (e != hipSuccess) {
        set_error("dense W allocation of %.1f GB failed: %s", (double)d->w_slots() * 16e-9, hipGetErrorString(e));
        delete d;
        return e == hipErrorOutOfMemory ? WAGG_ENOMEM : WAGG_EHIP;
    }
    *out = d;
    return WAGG_OK;
}

static int dense_den_to_host(wagg_dense *d, hipStream_t st = nullptr) {
    hipLaunchKernelGGL(dense_den32_kernel, dim3((unsigned)((d->R + 255) / 256)), dim3(256), 0, st,
                       d->den64.p, d->den32.p, d->R);
    WAGG_HIP(hipGetLastError());
    d->den_host.resize((size_t)d->R);
    WAGG_HIP(staged_d2h(d->den_host.data(), d->den64.p, sizeof(double) * (size_t)d->R, st));
    return WAGG_OK;
}

template <typename T>
static int dense_finish_den(wagg_dense *d) {
    typedef typename DT<T>::vec vec_t;
    const int kt_per_block = 128;
    const int n_strips = (d->n_kt + kt_per_block - 1) / kt_per_block;
    dim3 grid((unsigned)d->n_nt, (unsigned)n_strips);
    DevBuf<double> part;
    WAGG_HIP(part.alloc((size_t)n_strips * (size_t)d->n_nt * D_WSLOTS));
    hipLaunchKernelGGL((dense_colsum_kernel<T>), grid, dim3(512), 0, nullptr,
                       reinterpret_cast<const vec_t *>(d->W.p), d->R, d->n_kt, kt_per_block, part.p);
    hipLaunchKernelGGL(dense_colsum_combine_kernel, dim3((unsigned)((d->R + 255) / 256)), dim3(256), 0, nullptr,
                       (const double *)part.p, n_strips, d->n_nt, d->R, d->den64.p);
    WAGG_HIP(hipGetLastError());
    WAGG_HIP(hipDeviceSynchronize());            // `part` is freed on return
    return dense_den_to_host(d);
}

// stored-tile lists of the tile-sparse form from the sorted keys nt * n_kt + kt
static hipError_t dense_set_tiles(wagg_dense *d, const std::vector<int64_t> &tiles, std::vector<int32_t> *nt_out = nullptr,
                                  hipStream_t st = nullptr) {
    std::vector<int32_t> kt(tiles.size() + 2, 0), ntv(tiles.size() + 1, 0);
    std::vector<int64_t> first((size_t)d->n_nt + 1, 0);
    for (size_t i = 0; i < tiles.size(); ++i) {
        kt[i] = (int32_t)(tiles[i] % d->n_kt);
        ntv[i] = (int32_t)(tiles[i] / d->n_kt);
        first[(size_t)ntv[i] + 1]++;
    }
    for (int nt = 0; nt < d->n_nt; ++nt) first[(size_t)nt + 1] += first[(size_t)nt];
    d->nt_first.assign(first.size(), 0);
    for (size_t i = 0; i < first.size(); ++i) d->nt_first[i] = (int32_t)first[i];
    hipError_t e = d->tile_kt.upload(kt, st);
    if (nt_out) *nt_out = ntv;
    return e;
}

// The piece table of a tile-sparse launch over n_mb row blocks (see dense_mfma_kernel).  Pairs in (row block, column tile)
// order; workgroup w takes the tiles [w total / n_wg, (w + 1) total / n_wg) of their concatenation; a pair's pieces get
// consecutive slabs, in run order (the reduce kernel adds them in that order: fixed, hence reproducible).  Built once per
// row-block count and kept with the plan.
static int tile_pieces_for(wagg_dense *d, int n_mb, hipStream_t st, const wagg_dense::TilePieces **out) {
    for (const auto &tp : d->pieces)
        if (tp->n_mb == n_mb) { *out = tp.get(); return WAGG_OK; }
    try {
        std::unique_ptr<wagg_dense::TilePieces> tp(new wagg_dense::TilePieces());
        const int n_nt = d->n_nt;
        const int64_t per_mb = d->n_tiles, total = per_mb * n_mb;
        // at least ~8 tiles per workgroup (a piece costs a pipeline fill and a slab), at most one workgroup per CU
        int64_t n_wg = total / 8;
        if (n_wg > d->ncu) n_wg = d->ncu;
        if (n_wg < 1) n_wg = 1;
        std::vector<int32_t> first_piece((size_t)n_wg + 1, 0), rec, slab_first(2 * (size_t)n_mb * n_nt, 0);
        int64_t pos = 0;                               // position in the concatenation
        int slab = 0, w = 0;
        int64_t w_end = total * 1 / n_wg;              // end of workgroup 0's share
        // pair order: row block by row block (diagnostic build, WAGG_PIECES_ORDER=nt: column tile by column tile, so that the
        // row blocks of one column tile -- which contract the same W tiles -- run side by side)
        bool nt_major = false;
#ifdef WAGG_DIAG
        if (const char *e = getenv("WAGG_PIECES_ORDER")) nt_major = e[0] == 'n';
#endif
        for (int64_t pair = 0; pair < (int64_t)n_mb * n_nt; ++pair) {
            const int mb = nt_major ? (int)(pair % n_mb) : (int)(pair / n_nt), nt = nt_major ? (int)(pair / n_mb) : (int)(pair % n_nt);
            {
                const int slab0 = slab;
                slab_first[2 * ((size_t)mb * n_nt + nt)] = slab;
                int64_t t0 = d->nt_first[(size_t)nt];
                const int64_t t1 = d->nt_first[(size_t)nt + 1];
                while (t0 < t1) {
                    while (pos >= w_end && w + 1 < n_wg) { ++w; first_piece[(size_t)w] = (int32_t)(rec.size() / 8); w_end = total * (w + 1) / n_wg; }
                    const int64_t room = w + 1 < n_wg ? w_end - pos : t1 - t0;
                    const int64_t take = room < t1 - t0 ? room : t1 - t0;
                    const int32_t r8[8] = {nt, mb, (int32_t)t0, (int32_t)take, slab, 0, 0, 0};
                    rec.insert(rec.end(), r8, r8 + 8);
                    ++slab;
                    t0 += take;
                    pos += take;
                }
                slab_first[2 * ((size_t)mb * n_nt + nt) + 1] = slab - slab0;
            }
        }
        for (int64_t k = w + 1; k <= n_wg; ++k) first_piece[(size_t)k] = (int32_t)(rec.size() / 8);      // (workgroups without a share: none)
        tp->n_mb = n_mb; tp->n_wg = (int)n_wg; tp->n_slabs = slab;
        std::vector<int32_t> all;
        all.reserve(first_piece.size() + rec.size() + slab_first.size());
        all.insert(all.end(), first_piece.begin(), first_piece.end());
        all.insert(all.end(), rec.begin(), rec.end());
        tp->slab_first_at = (int64_t)all.size();
        all.insert(all.end(), slab_first.begin(), slab_first.end());
        WAGG_HIP(tp->tab.upload(all, st));
        *out = tp.get();
        d->pieces.push_back(std::move(tp));
        return WAGG_OK;
    } catch (const std::bad_alloc &) { set_error("host allocation failed"); return WAGG_ENOMEM; }
}

// Generated tables (wagg_dense_create_synth*): below this share of non-zeros a scattered W goes to the entry-list form.
constexpr double SPMM_MAX_FILL = 0.10;

// ---- which form a caller's table takes (wagg_dense_create_from_csr* / _from_segments*) ---------------------------------
// The reference knows one weights type (aggregations.py:64-73) and so does the caller here: the form is the library's
// business.  It is chosen by the estimated time per row of X of the three forms, each priced at the rate its kernel was
// MEASURED at on the c5 grid (tools/form_crossover.py -> profiles/r05_form_crossover.txt; docs/HISTORY.md (b) has the table):
//   full matrix     2 x (all tiles x BK x 256) flop     at the full-form MFMA rate (padding of G and R to whole tiles included)
//   tile-sparse     2 x (stored tiles x BK x 256) flop  at the tile-sparse MFMA rate (a little lower: the tile list is walked)
//   entry lists     2 x walked entries flop at the entry-loop rate of their mean list length, but never faster than the X
//                   stream that every one of the n_rb region blocks pulls through the LDS-DMA path (b G n_rb bytes per row)
// Rates in flop/s and byte/s; fp32 / fp64.
struct FormRates { double full, tiled, entries_scale, dma; };
// fp32: full 204.3 ms for 2,282 rows x 8,100 x 96 tiles; tile-sparse (a workgroup walks an equal share of the stored tiles:
// dense_pieces_kernel) 144 TF on the stored tiles' flops (c5-block: 8.39 ms), fp64 73 TF (16.58 ms); c5-uniform-f64 of
// bench.py for the fp64 entry-list scale
constexpr FormRates FORM_RATES_F32 = {142e12, 144e12, 1.0, 10.1e12};
constexpr FormRates FORM_RATES_F64 = {70e12, 73e12, 0.44, 10.1e12};
// The entry loop slows down as the lists grow (the first 16 groups of a wave's list are preloaded across the previous chunk;
// what follows is fetched inside the loop): flop/s on the WALKED entries against the mean list length per wave and chunk,
// fp32, measured at uniform fills of 1, 3, 6, 10, 15 and 30 % (mean lengths 54 ... 1625)
static double entry_loop_rate(double mean_list) {
    static const double L[] = {170, 325, 541, 812, 1625}, Rt[] = {30e12, 23e12, 19.5e12, 17.3e12, 15.9e12};
    if (mean_list <= L[0]) return Rt[0];
    for (int i = 1; i < 5; ++i)
        if (mean_list <= L[i]) return Rt[i - 1] + (Rt[i] - Rt[i - 1]) * (mean_list - L[i - 1]) / (L[i] - L[i - 1]);
    return Rt[4];
}

struct FormCost { double t_full, t_tiled, t_entries; };
// seconds per row of X.  The MFMA forms multiply whole (BK x 256) tiles, padding included; the entry-list kernel walks
// `walked` entries (spmm_list_cost: 16 x the longest per-wave list of every item, in whole groups) spread over n_lists
// (region block, chunk, wave) lists.
static FormCost table_form_cost(int64_t G, int elem_bytes, int64_t n_tiles_stored, int64_t n_tiles_all, int64_t walked, int n_rb,
                                int64_t n_lists, int n_nt) {
    const FormRates &rt = elem_bytes == 8 ? FORM_RATES_F64 : FORM_RATES_F32;
    const double tile_flop = 2.0 * (128.0 / elem_bytes) * 256.0;        // BK = 32 / 16 cells x 256 regions
    FormCost c;
    // (full form with few column tiles: a launch has at most n_nt x 64 k-slices of workgroups per row block -- R = 600 fills
    //  192 of 256 CUs; the tile-sparse form hands every CU an equal share of the stored tiles whatever their layout)
    const double util = n_nt * 64.0 < 256.0 ? n_nt * 64.0 / 256.0 : 1.0;
    c.t_full = tile_flop * (double)n_tiles_all / (rt.full * util);
    c.t_tiled = tile_flop * (double)n_tiles_stored / rt.tiled;
    const double rate = rt.entries_scale * entry_loop_rate((double)walked / (double)(n_lists > 0 ? n_lists : 1));
    const double t_loop = 2.0 * (double)walked / rate, t_stream = (double)elem_bytes * (double)G * (double)n_rb / rt.dma;
    c.t_entries = t_loop > t_stream ? t_loop : t_stream;
    return c;
}
static void pick_table_form(const FormCost &c, bool *tiled, bool *entries) {
    *entries = c.t_entries < c.t_full && c.t_entries < c.t_tiled;
    *tiled = !*entries && c.t_tiled < c.t_full;
}

static int pick_ksplit(int64_t items, int n_kt) {
    int best = 8;
    double best_eff = 0.0;
    for (int S = 8; S <= 64; S += 8) {
        if (S > 8 && n_kt / S < 32) break;             // keep >= 32 LDS tiles per slice
        const double w = (double)items * S / 256.0;
        const double eff = w / std::ceil(w);
        if (eff > best_eff + 1e-9) { best_eff = eff; best = S; }
        if (eff >= 0.985) break;
    }
    return best;
}

template <typename T>
static int create_synth(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    typedef typename DT<T>::vec vec_t;
    WAGG_REQUIRE(fill > 0.0 && fill <= 1.0, "fill must be in (0, 1]");
    if (fill < SPMM_MAX_FILL) {                         // scattered and sparse: entry lists instead of a matrix
        int rc = dense_alloc<T>(G, R, out, -1, true);
        if (rc != WAGG_OK) return rc;
        rc = spmm_build_synth<T>(*out, seed, fill);
        if (rc != WAGG_OK) { delete *out; *out = nullptr; }
        return rc;
    }
    int rc = dense_alloc<T>(G, R, out);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    hipLaunchKernelGGL((dense_synth_w_kernel<T>), dim3(256 * 32), dim3(256), 0, nullptr,
                       reinterpret_cast<vec_t *>(d->W.p), G, R, d->n_kt, d->w_slots(), seed, (float)fill);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) rc = dense_finish_den<T>(d); else { set_error("synth launch: %s", hipGetErrorString(e)); rc = WAGG_EHIP; }
    if (rc != WAGG_OK) { delete d; *out = nullptr; }
    return rc;
}

template <typename T>
static int create_synth_blocklocal(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    typedef typename DT<T>::vec vec_t;
    constexpr int BK = DT<T>::BK;
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(G > 0 && R > 0 && fill > 0.0 && fill <= 1.0, "bad arguments");
    const int n_kt = (int)((G + BK - 1) / BK), n_nt = (int)(((int64_t)R + D_BN - 1) / D_BN);
    // run j (64 cells = 64 / BK consecutive k tiles) touches column tile (97 j) mod n_nt
    std::vector<int64_t> tiles;
    tiles.reserve((size_t)n_kt);
    for (int kt = 0; kt < n_kt; ++kt) tiles.push_back((int64_t)(((int64_t)97 * (kt / (64 / BK))) % n_nt) * n_kt + kt);
    std::sort(tiles.begin(), tiles.end());
    int rc = dense_alloc<T>(G, R, out, (int64_t)tiles.size());
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    std::vector<int32_t> ntv;
    DevBuf<int32_t> dnt, dfirst;
    hipError_t e = dense_set_tiles(d, tiles, &ntv);
    if (e == hipSuccess) e = dnt.upload(ntv);
    if (e == hipSuccess) {
        std::vector<int32_t> first((size_t)n_nt + 1, 0);
        for (size_t i = 0; i < tiles.size(); ++i) first[(size_t)ntv[i] + 1]++;
        for (int nt = 0; nt < n_nt; ++nt) first[(size_t)nt + 1] += first[(size_t)nt];
        e = dfirst.upload(first);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL((dense_synth_blocklocal_kernel<T>), dim3(256 * 16), dim3(256), 0, nullptr,
                           reinterpret_cast<vec_t *>(d->W.p), d->tile_kt.p, dnt.p, d->n_tiles, G, R, seed, (float)fill);
        e = hipGetLastError();
    }
    if (e == hipSuccess) {
        // the slice table for one slice = first stored tile of every column tile (+ the total): [nt][0..1]
        hipLaunchKernelGGL((dense_colsum_tiled_kernel<T>), dim3((unsigned)((R + 255) / 256)), dim3(256), 0, nullptr,
                           reinterpret_cast<const vec_t *>(d->W.p), (const int32_t *)dfirst.p, R, d->den64.p);
        e = hipGetLastError();
    }
    if (e != hipSuccess) { set_error("block-local synth: %s", hipGetErrorString(e)); delete d; *out = nullptr; return WAGG_EHIP; }
    rc = dense_den_to_host(d);
    if (rc != WAGG_OK) { delete d; *out = nullptr; }
    return rc;
}

template <typename T>
static int create_host(const T *W_host, int64_t G, int32_t R, wagg_dense **out) {
    typedef typename DT<T>::vec vec_t;
    WAGG_REQUIRE(W_host != nullptr, "W_host is NULL");
    int rc = dense_alloc<T>(G, R, out);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    DevBuf<T> plain;
    hipError_t e = plain.alloc((size_t)(G * R));
    if (e == hipSuccess && copy_to_device(plain.p, W_host, sizeof(T) * (size_t)(G * R), true) != WAGG_OK) { delete d; *out = nullptr; return WAGG_EHIP; }
    if (e == hipSuccess) {
        hipLaunchKernelGGL((dense_pack_w_kernel<T>), dim3(256 * 8), dim3(256), 0, nullptr, (const T *)plain.p, G, R, d->n_kt,
                           d->w_slots(), reinterpret_cast<vec_t *>(d->W.p));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { set_error("dense upload: %s", hipGetErrorString(e)); delete d; *out = nullptr; return WAGG_EHIP; }
    rc = dense_finish_den<T>(d);
    if (rc != WAGG_OK) { delete d; *out = nullptr; }
    return rc;
}

// ---- plans from a caller's table (COO segment rows or CSR), built on the device ---------------------------------------
// which (BK-cell x 256-region) tiles of W hold a pair: one bit per tile.  Key order walks a chunk's cells inside one wave's
// regions, so the 64 pairs of a wavefront lie in a handful of tiles: one lane per distinct tile of the wave sets its bit
// (evenly spread weights ask for every bit some eight hundred times -- the atomics on 24,300 words were most of this
// kernel while every head of a run of equal tiles issued one).
template <typename T>
__global__ __launch_bounds__(256) void table_tiles_kernel(const uint64_t *__restrict__ key, int64_t n, EntryKeyGeom geom, int n_kt,
                                                          uint32_t *__restrict__ bitmap) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int64_t t = -1;
    if (i < n) {
        int64_t cell; int32_t region; int cic, j;
        geom.decode(key[i], cell, region, cic, j);
        t = (int64_t)(region / D_BN) * n_kt + cell / DT<T>::BK;
    }
    uint64_t todo = __ballot(i < n);
    while (todo) {                                     // (wave-uniform)
        const int leader = __ffsll((unsigned long long)todo) - 1;
        const int64_t tl = __shfl(t, leader, 64);
        if (lane == leader) atomicOr(&bitmap[tl >> 5], 1u << (tl & 31));      // idempotent: the same bitmap on every build
        todo &= ~__ballot(t == tl);
    }
}

// distinct pairs -> packed W (full form: word_rank == NULL; tile-sparse: stored tile = rank of the pair's tile among the set bits)
template <typename T>
__global__ __launch_bounds__(256) void table_scatter_kernel(const uint64_t *__restrict__ key, const double *__restrict__ w, int64_t n,
                                                            EntryKeyGeom geom, int n_kt, const uint32_t *__restrict__ bitmap,
                                                            const uint32_t *__restrict__ word_rank, T *__restrict__ Wp) {
    constexpr int E = DT<T>::EPP, BK = DT<T>::BK;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t cell; int32_t region; int cic, j;
    geom.decode(key[i], cell, region, cic, j);
    int64_t at;
    if (word_rank) {
        const int64_t t = (int64_t)(region / D_BN) * n_kt + cell / BK;
        const int64_t ti = (int64_t)word_rank[t >> 5] + __popc(bitmap[t >> 5] & ((1u << (t & 31)) - 1u));
        const int cl = region % D_BN, kk = (int)(cell % BK);
        at = (ti * D_WSLOTS + tile_slot(cl, kk / E)) * E + (kk % E);
    } else {
        at = wp_index<T>(cell, region, n_kt);
    }
    Wp[at] = (T)w[i];
}

static double wall_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// cell_idx (COO) or rowptr (CSR: G + 1 offsets, rows = cells) -- host arrays, like region_code and w.  The host validates
// the row offsets and uploads; sorting, coalescing duplicate rows (S5), the denominators (aggregations.py:79), the choice
// of form and the packing all run on the device (wagg_build.hip): extra host memory is the 8-MiB staging pieces (or none:
// large arrays are page-locked in place for the copy).
template <typename T>
static int create_from_table(const int32_t *cell_idx, const int64_t *rowptr, const int32_t *region_code, const double *w_eff,
                             int64_t n, int64_t G, int32_t R, int flags, wagg_dense **out) {
    constexpr int BK = DT<T>::BK;
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(G > 0 && R > 0, "bad sizes G=%lld R=%d", (long long)G, R);
    const bool general_sort = (flags & WAGG_DENSE_GENERAL_SORT) != 0;
    flags &= ~WAGG_DENSE_GENERAL_SORT;
    WAGG_REQUIRE(flags >= WAGG_DENSE_FORM_AUTO && flags <= WAGG_DENSE_FORCE_ENTRIES, "unknown form flag %d", flags);
    if (rowptr) {
        WAGG_REQUIRE(rowptr[0] == 0, "rowptr[0] must be 0");
        for (int64_t g = 0; g < G; ++g) WAGG_REQUIRE(rowptr[g] <= rowptr[g + 1], "rowptr decreases at row %lld", (long long)g);
        n = rowptr[G];
    }
    WAGG_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff, "table of %lld rows: at most 2^31 - 1", (long long)n);
    WAGG_REQUIRE(n == 0 || ((cell_idx || rowptr) && region_code && w_eff), "NULL table arrays");
    const double t0 = wall_s();
    SpmmPlan geo;
    spmm_geometry(G, R, geo);
    EntryKeyGeom kg;
    kg.rw = geo.rw; kg.n_rb = geo.n_rb; kg.n_chunks = geo.n_chunks;
    const int n_kt = (int)((G + BK - 1) / BK), n_nt = (int)(((int64_t)R + D_BN - 1) / D_BN);
    const int64_t n_tiles_all = (int64_t)n_kt * n_nt, n_words = (n_tiles_all + 31) / 32;
    const int64_t n_buckets = (int64_t)geo.n_rb * geo.n_chunks * SP_WAVES;
    // one stream and one arena for the whole build (wagg_build.h): the sort's scratch first, and once that is released the
    // tile bitmap with its ranks or the list bounds of the entry-list packing
    const size_t after_sort = 2 * (sizeof(uint32_t) * (size_t)n_words + 512) + 2 * (sizeof(int32_t) * ((size_t)n_buckets + 1) + 512) +
                              sizeof(uint32_t) * ((size_t)n_buckets / 2048 + 2) + 8192;
    size_t arena = build_arena_bytes(n, G, R, rowptr != nullptr);
    if (rowptr) arena += chunk_sort_scratch_bytes(kg);
    if (arena < after_sort) arena = after_sort;
    BuildCtx ctx;
    WAGG_BUILD_STAMP(ctx, "^start");
    {
        const hipError_t e = ctx.init(arena);
        if (e != hipSuccess) {
            set_error("plan build: %.2f GB of device scratch for a table of %lld rows -> %s", (double)arena * 1e-9, (long long)n, hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? WAGG_ENOMEM : WAGG_EHIP;
        }
    }
    WAGG_BUILD_STAMP(ctx, "arena");
    SortedEntries se;
    double t_up = 0.0;
    {
        // the table goes to the input side of the arena and is dropped by build_sorted_entries once the keys exist
        int32_t *dcell = nullptr, *dreg = nullptr;
        int64_t *drow = nullptr;
        double *dw = nullptr;
        if (n > 0) {
            if (rowptr) { drow = ctx.take_input<int64_t>((size_t)G + 1); } else { dcell = ctx.take_input<int32_t>((size_t)n); }
            dreg = ctx.take_input<int32_t>((size_t)n);
            dw = ctx.take_input<double>((size_t)n);
            if (!(drow || dcell) || !dreg || !dw) { set_error("build arena too small for the table"); return WAGG_ENOMEM; }
            int rc = rowptr ? copy_to_device(drow, rowptr, sizeof(int64_t) * ((size_t)G + 1), true, ctx.st)
                            : copy_to_device(dcell, cell_idx, sizeof(int32_t) * (size_t)n, true, ctx.st);
            if (rc == WAGG_OK) rc = copy_to_device(dreg, region_code, sizeof(int32_t) * (size_t)n, true, ctx.st);
            if (rc == WAGG_OK) rc = copy_to_device(dw, w_eff, sizeof(double) * (size_t)n, true, ctx.st);
            if (rc != WAGG_OK) return rc;
        }
        t_up = wall_s() - t0;
        WAGG_BUILD_STAMP(ctx, "upload");
        if (int rc = build_sorted_entries(ctx, dcell, drow, dreg, dw, n, G, R, kg, &se, general_sort)) return rc;
    }
    // which tiles of W hold anything?  Few -> tile-sparse form; (almost) all but few pairs in them -> entry lists
    uint32_t *bitmap, *word_rank = nullptr;
    WAGG_TAKE(bitmap, ctx, uint32_t, n_words);
    WAGG_HIP(hipMemsetAsync(bitmap, 0, sizeof(uint32_t) * (size_t)n_words, ctx.st));
    const unsigned nblk = (unsigned)((se.n_u + 255) / 256);
    if (se.n_u > 0) {
        hipLaunchKernelGGL((table_tiles_kernel<T>), dim3(nblk), dim3(256), 0, ctx.st, (const uint64_t *)se.key.p, se.n_u, kg, n_kt, bitmap);
        WAGG_HIP(hipGetLastError());
    }
    WAGG_BUILD_STAMP(ctx, "den + tile census");
    std::vector<uint32_t> hbits, hrank;
    std::vector<int64_t> tiles;
    try {
        hbits.resize((size_t)n_words);
        hrank.resize((size_t)n_words);
        WAGG_HIP(staged_d2h(hbits.data(), bitmap, sizeof(uint32_t) * hbits.size(), ctx.st));     // (the host picks the form from it)
        int64_t cnt = 0;
        for (int64_t i = 0; i < n_words; ++i) { hrank[(size_t)i] = (uint32_t)cnt; cnt += __builtin_popcount(hbits[(size_t)i]); }
        tiles.reserve((size_t)cnt);
        for (int64_t i = 0; i < n_words; ++i)
            for (uint32_t m = hbits[(size_t)i]; m; m &= m - 1) tiles.push_back(i * 32 + __builtin_ctz(m));
    } catch (const std::bad_alloc &) { set_error("host allocation failed"); return WAGG_ENOMEM; }
    // the form: measured crossovers (tools/form_crossover.py, docs/HISTORY.md (b)); a WAGG_DENSE_FORCE_* flag overrides the choice
    const double fill_all = (double)se.n_u / ((double)G * (double)R);
    int64_t walked = 0;
    if (se.n_u > 0) { if (int rc = spmm_list_cost(ctx, se, &walked)) return rc; }
    WAGG_BUILD_STAMP(ctx, "bitmap to host + list cost");
    const FormCost cost = table_form_cost(G, (int)sizeof(T), (int64_t)tiles.size(), n_tiles_all, walked, geo.n_rb, n_buckets, n_nt);
    bool tiled, entries;
    pick_table_form(cost, &tiled, &entries);
    if (flags == WAGG_DENSE_FORCE_FULL) { tiled = false; entries = false; }
    else if (flags == WAGG_DENSE_FORCE_TILES) { tiled = true; entries = false; }
    else if (flags == WAGG_DENSE_FORCE_ENTRIES) { tiled = false; entries = true; }
#ifdef WAGG_DIAG
    if (getenv("WAGG_DENSE_NO_TILED") && tiled) { tiled = false; entries = fill_all < SPMM_MAX_FILL; }
#else
    (void)fill_all;
#endif
    int rc = dense_alloc<T>(G, R, out, tiled ? (int64_t)tiles.size() : -1, entries);
    if (rc != WAGG_OK) return rc;
    wagg_dense *d = *out;
    WAGG_BUILD_STAMP(ctx, "plan alloc");
    auto fail = [&](int code) { (void)ctx.sync(); delete d; *out = nullptr; return code; };
    if (entries) {
        ctx.release_to(0);                            // (the bitmap has been read; the list bounds take its place, in stream order)
        rc = spmm_build_from_sorted<T>(ctx, d, se);
        if (rc != WAGG_OK) return fail(rc);
    } else {
        hipError_t e = hipMemsetAsync(d->W.p, 0, 16 * (size_t)d->w_slots(), ctx.st);
        if (e == hipSuccess && tiled) e = dense_set_tiles(d, tiles, nullptr, ctx.st);
        if (e == hipSuccess && tiled) {
            word_rank = ctx.take<uint32_t>((size_t)n_words);
            e = word_rank ? staged_h2d(word_rank, hrank.data(), sizeof(uint32_t) * hrank.size(), ctx.st) : hipErrorOutOfMemory;
        }
        if (e == hipSuccess && se.n_u > 0) {
            hipLaunchKernelGGL((table_scatter_kernel<T>), dim3(nblk), dim3(256), 0, ctx.st, (const uint64_t *)se.key.p, (const double *)se.w.p,
                               se.n_u, kg, n_kt, (const uint32_t *)bitmap, (const uint32_t *)word_rank, reinterpret_cast<T *>(d->W.p));
            e = hipGetLastError();
        }
        if (e != hipSuccess) { set_error("densify: %s", hipGetErrorString(e)); return fail(WAGG_EHIP); }
    }
    // denominators from the fp64 sums of the table (aggregations.py:79), not from the stored (rounded) weights
    hipError_t e = hipMemcpyAsync(d->den64.p, se.den.p, sizeof(double) * (size_t)R, hipMemcpyDeviceToDevice, ctx.st);
    if (e != hipSuccess) { set_error("densify den: %s", hipGetErrorString(e)); return fail(WAGG_EHIP); }
    rc = dense_den_to_host(d, ctx.st);
    if (rc != WAGG_OK) return fail(rc);
    e = ctx.sync();                                   // the plan is complete before anyone applies it on another stream
    if (e != hipSuccess) { set_error("plan build: %s", hipGetErrorString(e)); return fail(WAGG_EHIP); }
    WAGG_BUILD_STAMP(ctx, "pack + den to host");
    if (entries) d->sp.nnz = se.n_u;
    d->nnz_table = se.n_u;
    d->est_row_s[0] = cost.t_full; d->est_row_s[1] = cost.t_tiled; d->est_row_s[2] = cost.t_entries;
    d->walked_entries = walked;
    d->one_pass_sort = se.chunkwise;
    d->build.upload_s = t_up;
    d->build.total_s = wall_s() - t0;
    d->build.device_s = d->build.total_s - t_up;
    return WAGG_OK;
}

template <typename T, bool TILED, int MT, bool RM = false>
static const void *mfma_kernel_ptr() {
    if constexpr (TILED) return (const void *)dense_pieces_kernel<T, MT, RM>;        // tile-sparse form: a workgroup walks pieces
    else return (const void *)dense_mfma_kernel<T, 0, false, MT, RM>;
}
// full form, pack-free: instantiated for the tall row blocks only (fp32 MT >= 20, fp64 MT >= 10), where the packing
// pass is worth skipping; rm = first pass reading X in place, !rm = its gated packed second pass (DBG = 256)
template <typename T>
static const void *pick_full_rm_kernel(int MT, bool rm) {
#define WAGG_PICK(M) case M: return rm ? (const void *)dense_mfma_kernel<T, 0, false, M, true> \
                                       : (const void *)dense_mfma_kernel<T, 256, false, M, false>
    if constexpr (sizeof(T) == 4) {
        switch (MT) { WAGG_PICK(20); WAGG_PICK(21); WAGG_PICK(22); WAGG_PICK(23); default: return nullptr; }
    } else {
        switch (MT) { WAGG_PICK(10); WAGG_PICK(11); default: return nullptr; }
    }
#undef WAGG_PICK
}

// the kernel instantiated for MT row blocks (fp32: 1..6, 8, 10, ..., 20, 21, 22, 23; fp64: 1..6, 8, 10, 11)
template <typename T>
static const void *pick_mfma_kernel(int MT, bool tiled, bool rm = false) {
#define WAGG_PICK(M) case M: return tiled ? (rm ? mfma_kernel_ptr<T, true, M, true>() : mfma_kernel_ptr<T, true, M>()) \
                                          : mfma_kernel_ptr<T, false, M>()
    if constexpr (sizeof(T) == 4) {
        switch (MT) {
            WAGG_PICK(1); WAGG_PICK(2); WAGG_PICK(3); WAGG_PICK(4); WAGG_PICK(5); WAGG_PICK(6); WAGG_PICK(8);
            WAGG_PICK(10); WAGG_PICK(12); WAGG_PICK(14); WAGG_PICK(16); WAGG_PICK(18); WAGG_PICK(20);
            WAGG_PICK(21); WAGG_PICK(22); WAGG_PICK(23);
            default: return nullptr;
        }
    } else {
        switch (MT) {
            WAGG_PICK(1); WAGG_PICK(2); WAGG_PICK(3); WAGG_PICK(4); WAGG_PICK(5); WAGG_PICK(6); WAGG_PICK(8);
            WAGG_PICK(10); WAGG_PICK(11);
            default: return nullptr;
        }
    }
#undef WAGG_PICK
}

template <typename T>
static int dense_apply(wagg_dense *d, const T *X_dev, int64_t Tn, int64_t ldx, const PackXfT<T> &xf,
                       T *out_dev, int64_t ldo, int ksplit, void *stream) {
    typedef typename DT<T>::vec vec_t;
    WAGG_REQUIRE(d != nullptr, "dense plan is NULL");
    WAGG_REQUIRE(d->f64 == (sizeof(T) == 8), "this plan holds %s weights: use the matching wagg_dense_apply_*",
                 d->f64 ? "fp64" : "fp32");
    WAGG_REQUIRE(Tn >= 0, "T < 0");
    if (Tn == 0) return WAGG_OK;
    WAGG_REQUIRE(X_dev && out_dev, "X/out is NULL");
    WAGG_REQUIRE(xf.mode != XF_EDD || xf.X2 != nullptr, "tasmax is NULL");
    WAGG_REQUIRE(ldx >= d->G && ldo >= d->R, "ldx/ldo too small");
    WAGG_REQUIRE(ksplit >= 0 && ksplit % 8 == 0, "ksplit must be 0 or a multiple of 8");
    if (d->spmm) return spmm_apply<T>(d, X_dev, Tn, ldx, xf, out_dev, ldo, (hipStream_t)stream);
    // Long batches (ensemble x time rows of a small grid: 50 members x 30 years = 547,500 rows) go through in groups of
    // whole row blocks: the reduce kernel's grid has one y block per row (<= 65,535), and the packed copy of X and the
    // partial slabs are sized by the rows of one launch.  Same stream, same workspaces, one group after the other.
    constexpr int64_t ROWS_MAX = (int64_t)(65535 / (DT<T>::MT_MAX * 16)) * (DT<T>::MT_MAX * 16);
    if (Tn > ROWS_MAX) {
        for (int64_t t0 = 0; t0 < Tn; t0 += ROWS_MAX) {
            PackXfT<T> xg = xf;
            if (xg.X2) xg.X2 += t0 * ldx;
            const int64_t rows = Tn - t0 < ROWS_MAX ? Tn - t0 : ROWS_MAX;
            if (int rc = dense_apply<T>(d, X_dev + t0 * ldx, rows, ldx, xg, out_dev + t0 * ldo, ldo, ksplit, stream)) return rc;
        }
        return WAGG_OK;
    }
    const int n_nt = d->n_nt, n_kt = d->n_kt;
    // row blocks: as few as possible (<= MT_MAX x 16 rows each), evenly filled, 16 MT rows with MT from the
    // instantiated set -- fp32: T = 365 -> one block of 23 x 16; T = 1369 -> four of 22 x 16; T = 31 -> 2 x 16
    constexpr int BM_MAX = DT<T>::MT_MAX * 16;
    const int n_mb = (int)((Tn + BM_MAX - 1) / BM_MAX);
    const int rows = (int)((Tn + n_mb - 1) / n_mb);
    static const int mts32[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 14, 16, 18, 20, 21, 22, 23};
    static const int mts64[] = {1, 2, 3, 4, 5, 6, 8, 10, 11};
    int MT = DT<T>::MT_MAX;
    if constexpr (sizeof(T) == 4) { for (int m : mts32) if (m * 16 >= rows) { MT = m; break; } }
    else { for (int m : mts64) if (m * 16 >= rows) { MT = m; break; } }
    const int bm = MT * 16;
    // A ragged batch: with equal row blocks the padding of EVERY block is paid (c4's rank shard: 1,369 rows = 4 x 343 -> 4 blocks
    // of 22 x 16 = 1,408 rows, 2.8 % of the MFMA work on zero rows).  Full blocks first, the remainder as a launch of its own with
    // the smallest row-block count that holds it (3 x 352 + 313 -> 22, 22, 22, 20: 1,376 rows) -- when that saves at least two
    // sixteen-row units and the remainder is tall enough to stay MFMA-bound against its own pass over W.
    if (n_mb >= 2) {
        const int64_t head = (int64_t)(n_mb - 1) * bm, tail = Tn - head;
        if (tail > 0) {
            int MT_tail = DT<T>::MT_MAX;
            if constexpr (sizeof(T) == 4) { for (int m : mts32) if ((int64_t)m * 16 >= tail) { MT_tail = m; break; } }
            else { for (int m : mts64) if ((int64_t)m * 16 >= tail) { MT_tail = m; break; } }
            if (MT_tail >= 4 && MT - MT_tail >= 2) {
                PackXfT<T> xt = xf;
                if (xt.X2) xt.X2 += head * ldx;
                if (int rc = dense_apply<T>(d, X_dev, head, ldx, xf, out_dev, ldo, ksplit, stream)) return rc;
                return dense_apply<T>(d, X_dev + head * ldx, tail, ldx, xt, out_dev + head * ldo, ldo, ksplit, stream);
            }
        }
    }
    int S = ksplit ? ksplit : pick_ksplit((int64_t)n_nt * n_mb, n_kt);
    // tile-sparse form: no k-slices -- the launch walks PIECES, an equal share of all stored tiles per workgroup (see the kernel)
    const wagg_dense::TilePieces *tp = nullptr;
    if (d->tiled) {
        if (int rc = tile_pieces_for(d, n_mb, (hipStream_t)stream, &tp)) return rc;
        S = 1;
    }
    const int kt_per_slice = (n_kt + S - 1) / S;
    const int64_t nblk = tp ? (int64_t)tp->n_wg : (int64_t)n_nt * n_mb * S;
    WAGG_REQUIRE(nblk < (int64_t)0x7fffffff, "grid too large");
    const size_t n_slabs = tp ? (size_t)tp->n_slabs : (size_t)n_nt * n_mb * S;
    const size_t need = (n_slabs ? n_slabs : 1) * bm * D_BN * (sizeof(T) / 4);      // DevBuf<float>: 4-byte units
    if (d->slabs.n < need) WAGG_HIP(d->slabs.alloc(need));   // first call (or larger T) only
    const int64_t x_slots = (int64_t)n_mb * n_kt * bm * 8;
    if (d->xp.n < (size_t)x_slots * 4) WAGG_HIP(d->xp.alloc((size_t)x_slots * 4));
    const int aligned = ((ldx * sizeof(T)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(X_dev) & 15) == 0) &&
                        (xf.mode != XF_EDD || (reinterpret_cast<uintptr_t>(xf.X2) & 15) == 0);
    const size_t shmem = 2 * (size_t)d_buf_bytes(MT);
    hipStream_t st = (hipStream_t)stream;
    const void *kern = pick_mfma_kernel<T>(MT, d->tiled);
    if (!kern) { set_error("no kernel for MT=%d", MT); return WAGG_EINVAL; }
    // pack-free first pass (plain aggregation of 16-byte-aligned rows, grid = whole k tiles):
    // the MFMA kernel reads X where it lies; the packed pass below then runs only if a numerator came out
    // non-finite (NaN / +-inf somewhere in the data), gated on the device so the stream never waits for the host
    // ... and only while no earlier pack-free pass of this plan met such data: fields with NaN in them (ocean cells of
    // land-only variables) tend to stay that way, and for them the first pass is pure overhead.  The note is a
    // host-mapped word written by the reduce kernel; reading it here without synchronising is a heuristic only.
    // (G must be a whole number of k tiles: the last tile is then read from inside every row, whatever the caller's
    // pitch and however short the buffer behind the last row is)
    bool rm = xf.mode == 0 && aligned && d->G % DT<T>::BK == 0 && ((volatile int *)d->inf_host)[1] == 0;
    const void *kern_rm = nullptr;
    if (rm && d->tiled) kern_rm = pick_mfma_kernel<T>(MT, true, true);
    // (full form: only with two or more row blocks -- with one, A/B on one GPU shows the step unchanged: the kernel
    // loses reading in place what the skipped packing pass saves)
    else if (rm && n_mb >= 2 && (kern_rm = pick_full_rm_kernel<T>(MT, true)) != nullptr) kern = pick_full_rm_kernel<T>(MT, false);
    rm = kern_rm != nullptr;
    if (rm) WAGG_HIP(allow_dynamic_lds(kern_rm, shmem));
#ifdef WAGG_DIAG      // ablation variants (timing only; results are wrong with bit0 or bit2): tools/dense_ablate.sh
    // WAGG_DENSE_PACKED=1 (or any WAGG_DENSE_DBG variant): always take the packed pass, for A/B runs on one GPU
    if (getenv("WAGG_DENSE_PACKED") || (!d->tiled && getenv("WAGG_DENSE_DBG"))) {
        rm = false; kern_rm = nullptr;
        kern = pick_mfma_kernel<T>(MT, d->tiled);
    }
    if constexpr (sizeof(T) == 4) {
        if (const char *dbg = (d->tiled || MT != D_MT) ? nullptr : getenv("WAGG_DENSE_DBG")) {
            switch (atoi(dbg)) {
                case 1: kern = (const void *)dense_mfma_kernel<float, 1>; break;
                case 4: kern = (const void *)dense_mfma_kernel<float, 4>; break;
                case 5: kern = (const void *)dense_mfma_kernel<float, 5>; break;
                case 8: kern = (const void *)dense_mfma_kernel<float, 8>; break;
                case 16: kern = (const void *)dense_mfma_kernel<float, 16>; break;
                case 32: kern = (const void *)dense_mfma_kernel<float, 32>; break;
                case 64: kern = (const void *)dense_mfma_kernel<float, 64>; break;
                case 96: kern = (const void *)dense_mfma_kernel<float, 96>; break;
                case 128: kern = (const void *)dense_mfma_kernel<float, 128>; break;
                case 160: kern = (const void *)dense_mfma_kernel<float, 160>; break;
                case 192: kern = (const void *)dense_mfma_kernel<float, 192>; break;
                case 224: kern = (const void *)dense_mfma_kernel<float, 224>; break;
                default: break;
            }
        }
    }
#endif
    WAGG_HIP(allow_dynamic_lds(kern, shmem));
    const T *xp = reinterpret_cast<const T *>(d->xp.p), *wp = reinterpret_cast<const T *>(d->W.p);
    T *slabs = reinterpret_cast<T *>(d->slabs.p);
    int n_kt_a = n_kt, n_nt_a = n_nt, n_mb_a = n_mb, S_a = S, kps = kt_per_slice;
    const int32_t *tkt = d->tile_kt.p;
    const int32_t *toff = tp ? tp->tab.p : nullptr;                       // tile-sparse: the launch's piece table
    const int32_t *slab_first = tp ? tp->tab.p + tp->slab_first_at : nullptr;
    int64_t ldxB = ldx * (int64_t)sizeof(T);
    int Tn_a = (int)Tn;
    const T *den;
    if constexpr (sizeof(T) == 4) den = d->den32.p; else den = d->den64.p;
    const dim3 rgrid((unsigned)((d->R + 255) / 256), (unsigned)Tn);
    const int *gate = nullptr;
    if (rm) {
        gate = d->nonfinite.p;
        WAGG_HIP(hipMemsetAsync(d->nonfinite.p, 0, sizeof(int), st));
        const int *no_gate = nullptr;
        void *args_rm[] = {&X_dev, &wp, &n_kt_a, &n_nt_a, &n_mb_a, &S_a, &kps, &slabs, &tkt, &toff, &ldxB, &Tn_a, &no_gate};
        WAGG_HIP(launch_timed_ptr(true, kern_rm, dim3((unsigned)nblk), dim3(D_THREADS), args_rm, shmem, st));
        hipLaunchKernelGGL((dense_reduce_kernel<T>), rgrid, dim3(256), 0, st, (const T *)slabs, n_nt, S, bm, Tn, d->R, den,
                           out_dev, ldo, d->nonfinite.p, (const int *)nullptr, d->inf_dev + 1, slab_first);
        WAGG_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((dense_pack_x_kernel<T>), dim3(256 * 16), dim3(256), 0, st, X_dev, Tn, ldx, d->G, n_kt, bm, x_slots,
                       aligned, reinterpret_cast<vec_t *>(d->xp.p), xf, d->inf_dev, gate);
    WAGG_HIP(hipGetLastError());
    void *args[] = {&xp, &wp, &n_kt_a, &n_nt_a, &n_mb_a, &S_a, &kps, &slabs, &tkt, &toff, &ldxB, &Tn_a, &gate};
    WAGG_HIP(launch_timed_ptr(!rm, kern, dim3((unsigned)nblk), dim3(D_THREADS), args, shmem, st));
    hipLaunchKernelGGL((dense_reduce_kernel<T>), rgrid, dim3(256), 0, st, (const T *)slabs, n_nt, S, bm, Tn, d->R, den,
                       out_dev, ldo, (int *)nullptr, gate, (int *)nullptr, slab_first);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}

template <typename T>
static int apply_poly(wagg_dense *d, const T *X_dev, int64_t Tn, int64_t ldx, double offset, int power, T *out_dev,
                      int64_t ldo, int ksplit, void *stream) {
    WAGG_REQUIRE(power >= 1 && power <= 16, "power must lie in [1, 16], got %d", power);
    PackXfT<T> xf;
    xf.mode = power; xf.off = (T)offset;
    return dense_apply<T>(d, X_dev, Tn, ldx, xf, out_dev, ldo, ksplit, stream);
}

template <typename T>
static int apply_edd(wagg_dense *d, const T *tasmin, const T *tasmax, int64_t Tn, int64_t ldx, double offset,
                     double threshold, T *out_dev, int64_t ldo, int ksplit, void *stream) {
    PackXfT<T> xf;
    xf.mode = XF_EDD; xf.off = (T)offset; xf.thr = (T)threshold; xf.X2 = tasmax;
    return dense_apply<T>(d, tasmin, Tn, ldx, xf, out_dev, ldo, ksplit, stream);
}

}  // namespace wagg

extern "C" int wagg_dense_create_synth(int64_t G, int32_t R, uint32_t seed, wagg_dense **out) {
    return wagg::create_synth<float>(G, R, seed, 1.0, out);
}
extern "C" int wagg_dense_create_synth_sparse(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    return wagg::create_synth<float>(G, R, seed, fill, out);
}
extern "C" int wagg_dense_create_synth_f64(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    return wagg::create_synth<double>(G, R, seed, fill, out);
}
extern "C" int wagg_dense_create_synth_blocklocal(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    return wagg::create_synth_blocklocal<float>(G, R, seed, fill, out);
}
extern "C" int wagg_dense_create_synth_blocklocal_f64(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out) {
    return wagg::create_synth_blocklocal<double>(G, R, seed, fill, out);
}
extern "C" int wagg_dense_create_host(const float *W_host, int64_t G, int32_t R, wagg_dense **out) {
    return wagg::create_host<float>(W_host, G, R, out);
}
extern "C" int wagg_dense_create_host_f64(const double *W_host, int64_t G, int32_t R, wagg_dense **out) {
    return wagg::create_host<double>(W_host, G, R, out);
}
extern "C" int wagg_dense_create_from_segments(const int32_t *cell_idx, const int32_t *region_code,
                                               const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                               int flags, wagg_dense **out) {
    return wagg::create_from_table<float>(cell_idx, nullptr, region_code, w_eff, nseg, G, R, flags, out);
}
extern "C" int wagg_dense_create_from_segments_f64(const int32_t *cell_idx, const int32_t *region_code,
                                                   const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                                   int flags, wagg_dense **out) {
    return wagg::create_from_table<double>(cell_idx, nullptr, region_code, w_eff, nseg, G, R, flags, out);
}

extern "C" int wagg_dense_create_from_csr(const int64_t *rowptr, const int32_t *col, const double *val, int64_t G, int32_t R,
                                          int flags, wagg_dense **out) {
    using namespace wagg;
    WAGG_REQUIRE(rowptr != nullptr, "rowptr is NULL");
    return create_from_table<float>(nullptr, rowptr, col, val, 0, G, R, flags, out);
}
extern "C" int wagg_dense_create_from_csr_f64(const int64_t *rowptr, const int32_t *col, const double *val, int64_t G, int32_t R,
                                              int flags, wagg_dense **out) {
    using namespace wagg;
    WAGG_REQUIRE(rowptr != nullptr, "rowptr is NULL");
    return create_from_table<double>(nullptr, rowptr, col, val, 0, G, R, flags, out);
}

extern "C" int wagg_dense_get_info_sized(const wagg_dense *d, void *info_buf, uint64_t size) {
    using namespace wagg;
    WAGG_REQUIRE(d && info_buf, "NULL argument");
    wagg_dense_info st;
    std::memset(&st, 0, sizeof(st));
    wagg_dense_info *info = &st;
    info->G = d->G; info->R = d->R; info->n_kt = d->n_kt; info->n_nt = d->n_nt;
    info->n_tiles = d->n_tiles; info->tiled = d->tiled ? 1 : 0;
    info->w_bytes = d->spmm ? (int64_t)d->sp.n_groups * 4 * (d->f64 ? wagg::SpT<double>::GW : wagg::SpT<float>::GW) : d->w_slots() * 16;
    info->form = d->spmm ? WAGG_FORM_ENTRIES : (d->tiled ? WAGG_FORM_TILES : WAGG_FORM_FULL);
    info->elem_bytes = d->f64 ? 8 : 4;
    info->nnz = d->spmm ? d->sp.nnz : d->nnz_table;
    info->build_s = d->build.total_s;
    info->build_upload_s = d->build.upload_s;
    for (int k = 0; k < 3; ++k) info->est_row_s[k] = d->est_row_s[k];
    info->walked_entries = d->walked_entries;
    info->one_pass_sort = d->one_pass_sort ? 1 : 0;
    info->reserved0 = 0;
    copy_sized(info_buf, size, &st, sizeof(st));
    return WAGG_OK;
}
extern "C" int wagg_dense_get_info(const wagg_dense *d, wagg_dense_info *info) {
    return wagg_dense_get_info_sized(d, info, sizeof(wagg_dense_info));
}

namespace wagg {
template <typename E>
static hipError_t clone_buf(DevBuf<E> &dst, int dst_dev, const DevBuf<E> &src, int src_dev) {
    if (!src.p) return hipSuccess;
    hipError_t e = dst.alloc(src.n);
    if (e == hipSuccess && src.n) e = hipMemcpyPeer(dst.p, dst_dev, src.p, src_dev, src.n * sizeof(E));
    return e;
}
}  // namespace wagg

extern "C" int wagg_dense_clone(const wagg_dense *src, int device, wagg_dense **out) {
    using namespace wagg;
    clear_error();
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(src != nullptr, "dense plan is NULL");
    int n_dev = 0, cur = 0;
    WAGG_HIP(hipGetDeviceCount(&n_dev));
    WAGG_REQUIRE(device >= 0 && device < n_dev, "no device %d (%d visible)", device, n_dev);
    WAGG_HIP(hipGetDevice(&cur));
    wagg_dense *d = new (std::nothrow) wagg_dense();
    if (!d) { set_error("host allocation failed"); return WAGG_ENOMEM; }
    hipError_t e = hipSetDevice(device);
    try {
        d->G = src->G; d->R = src->R; d->n_kt = src->n_kt; d->n_nt = src->n_nt; d->f64 = src->f64;
        d->tiled = src->tiled; d->n_tiles = src->n_tiles; d->spmm = src->spmm;
        d->sp.rw = src->sp.rw; d->sp.n_rb = src->sp.n_rb; d->sp.n_chunks = src->sp.n_chunks;
        d->sp.nnz = src->sp.nnz; d->sp.n_groups = src->sp.n_groups;
        d->nnz_table = src->nnz_table; d->walked_entries = src->walked_entries; d->one_pass_sort = src->one_pass_sort;
        for (int k = 0; k < 3; ++k) d->est_row_s[k] = src->est_row_s[k];
        d->den_host = src->den_host;
        d->nt_first = src->nt_first;
        d->device = device;
        const double t0 = wall_s();
        if (e == hipSuccess) e = hipDeviceGetAttribute(&d->ncu, hipDeviceAttributeMultiprocessorCount, device);
        if (e == hipSuccess) e = d->nonfinite.alloc(1);
        if (e == hipSuccess) e = hipHostMalloc((void **)&d->inf_host, 2 * sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) { d->inf_host[0] = d->inf_host[1] = 0; e = hipHostGetDevicePointer((void **)&d->inf_dev, d->inf_host, 0); }
        if (e == hipSuccess) e = clone_buf(d->W, device, src->W, src->device);
        if (e == hipSuccess) e = clone_buf(d->den32, device, src->den32, src->device);
        if (e == hipSuccess) e = clone_buf(d->den64, device, src->den64, src->device);
        if (e == hipSuccess) e = clone_buf(d->tile_kt, device, src->tile_kt, src->device);
        if (e == hipSuccess) e = clone_buf(d->sp.ent, device, src->sp.ent, src->device);
        if (e == hipSuccess) e = clone_buf(d->sp.grp_off, device, src->sp.grp_off, src->device);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);   // (the copies are complete before anyone applies the clone)
        d->build.total_s = d->build.device_s = wall_s() - t0;
    } catch (const std::bad_alloc &) {
        (void)hipSetDevice(cur);
        delete d;
        set_error("host allocation failed");
        return WAGG_ENOMEM;
    }
    const hipError_t back = hipSetDevice(cur);
    if (e != hipSuccess || back != hipSuccess) {
        set_error("clone of a dense plan onto device %d: %s", device, hipGetErrorString(e != hipSuccess ? e : back));
        delete d;
        return e == hipErrorOutOfMemory ? WAGG_ENOMEM : WAGG_EHIP;
    }
    *out = d;
    return WAGG_OK;
}

extern "C" int wagg_dense_destroy(wagg_dense *d) {
    delete d;
    return WAGG_OK;
}

extern "C" int wagg_dense_get_den(const wagg_dense *d, double *den_host) {
    using namespace wagg;
    WAGG_REQUIRE(d && den_host, "NULL argument");
    std::memcpy(den_host, d->den_host.data(), sizeof(double) * (size_t)d->R);
    return WAGG_OK;
}

int wagg::entry::dense_apply_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx,
                                    float *out_dev, int64_t ldo, int ksplit, void *stream) {
    return wagg::dense_apply<float>(d, X_dev, T, ldx, wagg::PackXfT<float>{}, out_dev, ldo, ksplit, stream);
}
int wagg::entry::dense_apply_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx,
                                    double *out_dev, int64_t ldo, int ksplit, void *stream) {
    return wagg::dense_apply<double>(d, X_dev, T, ldx, wagg::PackXfT<double>{}, out_dev, ldo, ksplit, stream);
}
int wagg::entry::dense_apply_poly_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx, double offset,
                                         int power, float *out_dev, int64_t ldo, int ksplit, void *stream) {
    return wagg::apply_poly<float>(d, X_dev, T, ldx, offset, power, out_dev, ldo, ksplit, stream);
}
int wagg::entry::dense_apply_poly_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx, double offset,
                                         int power, double *out_dev, int64_t ldo, int ksplit, void *stream) {
    return wagg::apply_poly<double>(d, X_dev, T, ldx, offset, power, out_dev, ldo, ksplit, stream);
}
int wagg::entry::dense_apply_edd_f32(wagg_dense *d, const float *tasmin_dev, const float *tasmax_dev, int64_t T,
                                        int64_t ldx, double offset, double threshold, float *out_dev, int64_t ldo,
                                        int ksplit, void *stream) {
    return wagg::apply_edd<float>(d, tasmin_dev, tasmax_dev, T, ldx, offset, threshold, out_dev, ldo, ksplit, stream);
}
int wagg::entry::dense_apply_edd_f64(wagg_dense *d, const double *tasmin_dev, const double *tasmax_dev, int64_t T,
                                        int64_t ldx, double offset, double threshold, double *out_dev, int64_t ldo,
                                        int ksplit, void *stream) {
    return wagg::apply_edd<double>(d, tasmin_dev, tasmax_dev, T, ldx, offset, threshold, out_dev, ldo, ksplit, stream);
}

namespace wagg {
static int check_dense_device(const wagg_dense *d) {
    int cur = 0;
    WAGG_HIP(hipGetDevice(&cur));
    WAGG_REQUIRE(cur == d->device, "the dense plan was created on device %d, the current device is %d", d->device, cur);
    return WAGG_OK;
}

template <typename T>
static int dense_apply_host(wagg_dense *d, const T *X_host, int64_t Tn, int64_t ldx, T *out_host, int64_t ldo, int flags) {
    clear_error();
    WAGG_REQUIRE(d != nullptr, "dense plan is NULL");
    WAGG_REQUIRE(Tn >= 0, "T < 0");
    if (Tn == 0) return WAGG_OK;
    WAGG_REQUIRE(X_host && out_host, "X/out is NULL");
    WAGG_REQUIRE(ldx >= d->G && ldo >= d->R, "ldx/ldo too small");
    WAGG_REQUIRE((flags & ~(WAGG_HOST_PIN | WAGG_HOST_WHOLE)) == 0, "unknown host flags 0x%x", flags);
    if (int rc = check_dense_device(d)) return rc;
    if (flags & WAGG_HOST_WHOLE) {
        ScratchBuf<T> dx, dout;                  // (call-lifetime blocks: from the pool, wagg_scratch.hip; the device is drained below)
        WAGG_HIP(dx.alloc((size_t)(Tn * ldx)));
        WAGG_HIP(dout.alloc((size_t)(Tn * ldo)));
        const bool pin = (flags & WAGG_HOST_PIN) != 0;
        if (int rc = copy_to_device(dx.p, X_host, sizeof(T) * (size_t)((Tn - 1) * ldx + d->G), pin)) return rc;
        const int rc = dense_apply<T>(d, dx.p, Tn, ldx, PackXfT<T>{}, dout.p, ldo, 0, nullptr);
        if (rc != WAGG_OK) return rc;
        WAGG_HIP(hipDeviceSynchronize());
        return copy_rows_to_host(out_host, dout.p, Tn, sizeof(T) * (size_t)ldo, sizeof(T) * (size_t)d->R, pin);
    }
    return stream_host_rows<T>(X_host, Tn, ldx, d->G, out_host, ldo, d->R, flags, d->spmm ? SpT<T>::TB : DT<T>::MT_MAX * 16, 1, nullptr,
                               [&](int, const T *xd, int64_t rows, T *od, hipStream_t st) {
                                   return dense_apply<T>(d, xd, rows, ldx, PackXfT<T>{}, od, ldo, 0, (void *)st);
                               },
                               [](int, hipStream_t) {});
}

// multi-device form: one replica of the dense-family plan per device (wagg_apply_host_multi_* says how the blocks travel)
template <typename T>
static int dense_apply_host_multi(wagg_dense *const *plans, const int *devices, int n, const T *X_host, int64_t Tn, int64_t ldx,
                                  T *out_host, int64_t ldo, int flags) {
    clear_error();
    WAGG_REQUIRE(plans && devices && n >= 1 && n <= 64, "need 1..64 plan replicas and their devices");
    WAGG_REQUIRE((flags & ~WAGG_HOST_PIN) == 0, "unknown host flags 0x%x", flags);
    WAGG_REQUIRE(Tn >= 0, "T < 0");
    for (int s = 0; s < n; ++s) {
        WAGG_REQUIRE(plans[s] != nullptr, "plan replica %d is NULL", s);
        WAGG_REQUIRE(plans[s]->device == devices[s], "plan replica %d lives on device %d, not %d", s, plans[s]->device, devices[s]);
        WAGG_REQUIRE(plans[s]->G == plans[0]->G && plans[s]->R == plans[0]->R && plans[s]->f64 == (sizeof(T) == 8) &&
                     plans[s]->spmm == plans[0]->spmm, "plan replica %d has another shape, element type or form", s);
        for (int q = 0; q < s; ++q) WAGG_REQUIRE(plans[q] != plans[s], "replicas %d and %d are the same plan (a dense-family plan owns "
                                                 "its workspaces: one replica per pipeline)", q, s);
    }
    if (Tn == 0) return WAGG_OK;
    WAGG_REQUIRE(X_host && out_host, "X/out is NULL");
    wagg_dense *d0 = plans[0];
    WAGG_REQUIRE(ldx >= d0->G && ldo >= d0->R, "ldx/ldo too small");
    const int rc = stream_host_rows<T>(X_host, Tn, ldx, d0->G, out_host, ldo, d0->R, flags, d0->spmm ? SpT<T>::TB : DT<T>::MT_MAX * 16, n, devices,
                                       [&](int s, const T *xd, int64_t rows, T *od, hipStream_t st) {
                                           return dense_apply<T>(plans[s], xd, rows, ldx, PackXfT<T>{}, od, ldo, 0, (void *)st);
                                       },
                                       [](int, hipStream_t) {});
    // Every pipeline has drained.  A replica's pack stage leaves its +-inf note in the replica's own word: the caller asks
    // plans[0] (wagg_dense_saw_inf), so the notes of the other replicas move there (and are cleared: a replica is reused).
    for (int s = 1; s < n; ++s) {
        volatile int *note = plans[s]->inf_host;
        if (note[0]) { *(volatile int *)d0->inf_host = 1; note[0] = 0; }
    }
    return rc;
}
}  // namespace wagg

int wagg::entry::dense_apply_host_f32(wagg_dense *d, const float *X_host, int64_t T, int64_t ldx,
                                         float *out_host, int64_t ldo, int flags) {
    return wagg::dense_apply_host<float>(d, X_host, T, ldx, out_host, ldo, flags);
}
int wagg::entry::dense_apply_host_f64(wagg_dense *d, const double *X_host, int64_t T, int64_t ldx,
                                         double *out_host, int64_t ldo, int flags) {
    return wagg::dense_apply_host<double>(d, X_host, T, ldx, out_host, ldo, flags);
}
int wagg::entry::dense_apply_host_multi_f32(wagg_dense *const *plans, const int *devices, int n_devices, const float *X_host,
                                               int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags) {
    return wagg::dense_apply_host_multi<float>(plans, devices, n_devices, X_host, T, ldx, out_host, ldo, flags);
}
int wagg::entry::dense_apply_host_multi_f64(wagg_dense *const *plans, const int *devices, int n_devices, const double *X_host,
                                               int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags) {
    return wagg::dense_apply_host_multi<double>(plans, devices, n_devices, X_host, T, ldx, out_host, ldo, flags);
}

extern "C" int wagg_dense_saw_inf(wagg_dense *d, void *stream, int *saw) {
    using namespace wagg;
    WAGG_REQUIRE(d && saw, "NULL argument");
    WAGG_HIP(hipStreamSynchronize((hipStream_t)stream));
    *saw = *(volatile int *)d->inf_host;
    *(volatile int *)d->inf_host = 0;
    return WAGG_OK;
}
