// Sparse path: coded segment table -> region-grouped gather plan -> one HIP kernel.
//
// Replaces aggregations.py:24-27 (gather) and :78-80 (grouped sums + division) of the reference
// with   out[t,r] = sum_{i in r} X[t, cell_i] * w_i / den[r]   computed WITHOUT materialising
// the (T x nseg) gathered copy.  Design (MI355X-first, HBM-bound):
//
//   * host plan builder (below): coalesce duplicate (cell, region) rows, order regions along
//     8-row latitude bands, pack neighbouring regions into GROUPS whose union of cells is at
//     most UC=256 unique cells (one LDS "chunk"); regions larger than a chunk become "giant"
//     groups that walk several chunks.
//   * kernel: one 256-thread workgroup per (group, time block).  Thread u gathers the TB
//     timesteps of unique cell u (consecutive lanes -> consecutive cells -> coalesced lines)
//     with all TB loads in flight (64 KB of HBM requests per workgroup), parks them in an LDS
//     image xs[u][t] (row stride TB+1 words: conflict-free for the lane-per-cell store and the
//     lane-per-timestep read).  Then each wave owns one region at a time: lane = timestep, the
//     segment list (u, w) is wave-uniform and comes through the scalar cache, one ds_read +
//     one FMA per (segment, timestep).  The division by den[r] (aggregations.py:79-80) and the
//     skipna rule (NaN product counts 0, S6) are fused; results are stored once, no atomics,
//     bitwise reproducible.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

#include "wagg_common.h"

namespace wagg {

constexpr int UC = 256;      // cell slots per LDS chunk == workgroup size
constexpr int UQ = UC / 4;   // aligned 4-cell quads per chunk (one per lane of a wave)
constexpr int UROW = UC + 4; // LDS row stride in elements (t-major image, 16-byte aligned rows)
constexpr int RG_MAX = 255;  // regions per group (bounded by cells anyway)
constexpr int SEG_MAX = 768; // segments per chunk staged in LDS
constexpr int NWAVE = 4;

struct SparsePlanDev {
    DevBuf<int32_t> grp_chunk_begin, grp_giant;   // [n_groups+1], [n_groups]
    DevBuf<int32_t> chunk_u_begin, chunk_e_begin; // [n_chunks+1]
    DevBuf<int32_t> ucell;                        // [n_ucells]
    DevBuf<int32_t> ent_region, ent_seg_begin;    // [n_ent], [n_ent+1]
    DevBuf<int32_t> seg_u;                        // [nnz]
    DevBuf<float> seg_w32;
    DevBuf<double> seg_w64;
    DevBuf<float> den32;
    DevBuf<double> den64;
    DevBuf<int32_t> empty_regions;
};

}  // namespace wagg

struct wagg_plan {
    wagg_plan_info info{};
    std::vector<double> den_host;
    wagg::SparsePlanDev d;
    int device = 0;
};

namespace wagg {

template <typename T> struct PlanView {
    const int32_t *grp_chunk_begin, *grp_giant, *chunk_u_begin, *chunk_e_begin, *ucell;
    const int32_t *ent_region, *ent_seg_begin, *seg_u;
    const T *seg_w, *den;
    int n_groups;
};

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
// LDS carve-up (bytes):  xs [TB][UROW] T | red [NWAVE][TB] T | seg_w [SEG_MAX] T |
//                        ent_r [RG_MAX] i32 | seg_u [SEG_MAX] u16 | ent_s [RG_MAX+1] u16
// f32/TB=64: 74.8 KB, f64/TB=32: 79.9 KB  ->  two workgroups per CU (160 KB).
template <typename T, int TB> struct SparseLds {
    static constexpr size_t xs = 0;
    static constexpr size_t red = xs + sizeof(T) * TB * UROW;
    static constexpr size_t seg_w = red + sizeof(T) * NWAVE * TB;
    static constexpr size_t ent_r = seg_w + sizeof(T) * SEG_MAX;
    static constexpr size_t seg_u = ent_r + sizeof(int32_t) * RG_MAX;
    static constexpr size_t ent_s = seg_u + sizeof(uint16_t) * SEG_MAX;
    static constexpr size_t total = (ent_s + sizeof(uint16_t) * (RG_MAX + 1) + 15) / 16 * 16;
    static_assert(total <= 80 * 1024, "two workgroups must fit one CU's LDS");
};

// DBG: diagnostic knob (WAGG_SPARSE_DBG env, never set in production; results are wrong with
// any bit set): bit0 = no gather loads, bit1 = no LDS image stores, bit2 = no segment loop.
// VEC: rows of X are 16-byte aligned (base and ldx), so a quad is one aligned vector load.
template <typename T, int TB, int LAYOUT, int OUT_LAYOUT, bool VEC, int DBG = 0>
__global__ __launch_bounds__(UC, 2) void sparse_gather_kernel(PlanView<T> pv, const T *__restrict__ X,
                                                              int64_t Ttot, int64_t ldx, int64_t G,
                                                              T *__restrict__ out, int64_t ldo) {
    using L = SparseLds<T, TB>;
    constexpr int TPW = TB / NWAVE;                                // timesteps gathered per wave
    typedef T vec4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw + L::xs);              // [UC][TB+1]
    T *red = reinterpret_cast<T *>(smem_raw + L::red);            // [NWAVE][TB]  (giant groups only)
    T *sm_w = reinterpret_cast<T *>(smem_raw + L::seg_w);         // chunk's segment weights
    int32_t *sm_er = reinterpret_cast<int32_t *>(smem_raw + L::ent_r);     // chunk's entry -> region id
    uint16_t *sm_u = reinterpret_cast<uint16_t *>(smem_raw + L::seg_u);    // chunk's segment -> local cell
    uint16_t *sm_es = reinterpret_cast<uint16_t *>(smem_raw + L::ent_s);   // chunk's entry -> first segment

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // blocks b and b+8 share an XCD under round-robin placement (speed only): give every XCD a
    // contiguous run of logical ids so that neighbouring groups (shared border lines, same DRAM
    // pages) meet in one L2.  Bijective for any grid size.
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const unsigned lid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int g = (int)(lid % (unsigned)pv.n_groups);
    const int64_t t0 = (int64_t)(lid / (unsigned)pv.n_groups) * TB;
    const int nt = (int)((Ttot - t0) < TB ? (Ttot - t0) : TB);
    const bool giant = pv.grp_giant[g] != 0;
    const int c0 = pv.grp_chunk_begin[g], c1 = pv.grp_chunk_begin[g + 1];
    const bool lane_live = lane < nt && lane < TB;

    T giant_acc = T(0);
    for (int c = c0; c < c1; ++c) {
        const int u0 = pv.chunk_u_begin[c];
        const int nu = pv.chunk_u_begin[c + 1] - u0;
        const int e0 = pv.chunk_e_begin[c], ne = pv.chunk_e_begin[c + 1] - e0;
        const int sb = pv.ent_seg_begin[e0], ns = pv.ent_seg_begin[e0 + ne] - sb;
        // ---- all of the chunk's global traffic is issued here, together: the segment table and
        // entry table (a few KB, L2-resident) and TB rows of every unique cell (HBM) ----
        int mu[(SEG_MAX + UC - 1) / UC];
        T mw[(SEG_MAX + UC - 1) / UC];
#pragma unroll
        for (int i = 0; i < (SEG_MAX + UC - 1) / UC; ++i) {
            const int k = tid + UC * i;
            const bool ok = k < ns;
            mu[i] = ok ? pv.seg_u[sb + k] : 0;
            mw[i] = ok ? pv.seg_w[sb + k] : T(0);
        }
        const int er = tid < ne ? pv.ent_region[e0 + tid] : 0;
        const int es = tid <= ne ? pv.ent_seg_begin[e0 + tid] - sb : 0;
        // LDS image xs[t][u] (t-major).  TG: lane = quad, wave w takes timesteps [w*TPW, (w+1)*TPW):
        // one 16-byte (fp32) load per (quad, timestep) -- dword-per-lane gathers top out near
        // 2.4 TB/s on MI355X however contiguous the cells are -- and one aligned vector LDS store.
        // Rows past a ragged last block are clamped to its last row (their lanes never store a
        // result), so every block keeps TPW independent loads per lane in flight.
        if constexpr (LAYOUT == WAGG_LAYOUT_TG) {
            if (lane < nu) {
                const int64_t cell0 = pv.ucell[u0 + lane];
                const int tw0 = wave * TPW;
                vec4 v[TPW];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int tc = (tw0 + i < nt) ? tw0 + i : nt - 1;
                    const T *p = X + (t0 + tc) * ldx + cell0;
                    if (DBG & 1) v[i] = vec4{T(i), T(i), T(i), T(i)};
                    else if (VEC) v[i] = *reinterpret_cast<const vec4 *>(p);
                    else {                                           // unaligned rows / ragged grid end
                        const int64_t lim = G - 1 - cell0;           // >= 0
                        v[i] = vec4{p[0], p[lim < 1 ? lim : 1], p[lim < 2 ? lim : 2], p[lim < 3 ? lim : 3]};
                    }
                }
                if (!(DBG & 2)) {
#pragma unroll
                    for (int i = 0; i < TPW; ++i)
                        *reinterpret_cast<vec4 *>(&xs[(tw0 + i) * UROW + 4 * lane]) = v[i];
                } else {
                    vec4 sum = vec4{T(0), T(0), T(0), T(0)};
#pragma unroll
                    for (int i = 0; i < TPW; ++i) sum += v[i];
                    *reinterpret_cast<vec4 *>(&xs[tw0 * UROW + 4 * lane]) = sum;
                }
            }
        } else {
            for (int u = wave; u < 4 * nu; u += NWAVE) {
                int64_t cell = (int64_t)pv.ucell[u0 + (u >> 2)] + (u & 3);
                cell = cell < G ? cell : G - 1;
                if (lane < TB) xs[lane * UROW + u] = lane_live ? X[cell * ldx + t0 + lane] : T(0);
            }
        }
#pragma unroll
        for (int i = 0; i < (SEG_MAX + UC - 1) / UC; ++i) {
            const int k = tid + UC * i;
            if (k < ns) { sm_u[k] = (uint16_t)mu[i]; sm_w[k] = mw[i]; }
        }
        if (tid < ne) sm_er[tid] = er;
        if (tid <= ne) sm_es[tid] = (uint16_t)es;
        __syncthreads();
        // ---- weighted group sums: one wave per region, lane = timestep; the (cell, weight) list
        // is read from LDS at wave-uniform addresses (broadcast) ----
        for (int el = wave; el < ne; el += NWAVE) {
            const int s0 = __builtin_amdgcn_readfirstlane((int)sm_es[el]);
            const int s1 = __builtin_amdgcn_readfirstlane((int)sm_es[el + 1]);
            const int r = __builtin_amdgcn_readfirstlane(sm_er[el]);
            T den = T(1);
            if (!giant) den = pv.den[r];                         // in flight during the segment loop
            T acc = T(0);
            if (lane < TB && !(DBG & 4)) {
#pragma unroll 8
                for (int s = s0; s < s1; ++s) {
                    const int u = sm_u[s];
                    const T w = sm_w[s];
                    const T p = xs[lane * UROW + u] * w;        // aggregations.py:78 product
                    acc += (p == p) ? p : T(0);                 // skipna: NaN product counts 0 (S6)
                }
            }
            if (giant) {
                giant_acc += acc;
            } else if (lane_live) {
                const T q = acc / den;                          // aggregations.py:77-80, S7
                if constexpr (OUT_LAYOUT == WAGG_OUT_TR) out[(t0 + lane) * ldo + r] = q;
                else out[(int64_t)r * ldo + t0 + lane] = q;
            }
        }
        __syncthreads();
    }
    if (giant) {
        if (lane < TB) red[wave * TB + lane] = giant_acc;
        __syncthreads();
        if (wave == 0 && lane_live) {
            T s = red[lane];
#pragma unroll
            for (int w = 1; w < NWAVE; ++w) s += red[w * TB + lane];
            const int r = pv.ent_region[pv.chunk_e_begin[c0]];
            const T q = s / pv.den[r];
            if constexpr (OUT_LAYOUT == WAGG_OUT_TR) out[(t0 + lane) * ldo + r] = q;
            else out[(int64_t)r * ldo + t0 + lane] = q;
        }
    }
}

// (R x T) -> (T x R) through a padded 64x64 LDS tile: both sides coalesced.  The gather kernel
// stores region-major (lane = timestep: 256 contiguous bytes per region) because a (T x R) store
// from it would scatter single dwords over R-strided lines (7x write amplification measured).
template <typename T>
__global__ __launch_bounds__(256) void transpose_rt_to_tr_kernel(const T *__restrict__ in, int64_t ldi,
                                                                int64_t R, int64_t Ttot,
                                                                T *__restrict__ out, int64_t ldo) {
    __shared__ T tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        const int64_t r = r0 + ty + i, t = t0 + tx;
        if (r < R && t < Ttot) tile[ty + i][tx] = in[r * ldi + t];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        const int64_t t = t0 + ty + i, r = r0 + tx;
        if (r < R && t < Ttot) out[t * ldo + r] = tile[tx][ty + i];
    }
}

// regions without any kept segment: 0 / den (NaN when den == 0, S7)
template <typename T>
__global__ void fill_empty_kernel(const int32_t *__restrict__ regions, int n_empty,
                                  const T *__restrict__ den, int64_t Ttot, T *__restrict__ out,
                                  int64_t ldo, int out_layout) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n_empty * Ttot) return;
    const int r = regions[i / Ttot];
    const int64_t t = i % Ttot;
    const T q = T(0) / den[r];
    if (out_layout == WAGG_OUT_TR) out[t * ldo + r] = q;
    else out[(int64_t)r * ldo + t] = q;
}

template <typename T, int TB>
static int launch_sparse(const wagg_plan *plan, const T *X, int64_t Ttot, int64_t ldx, int layout,
                         T *out, int64_t ldo, int out_layout, hipStream_t stream) {
    const auto &d = plan->d;
    PlanView<T> pv;
    pv.grp_chunk_begin = d.grp_chunk_begin.p; pv.grp_giant = d.grp_giant.p;
    pv.chunk_u_begin = d.chunk_u_begin.p; pv.chunk_e_begin = d.chunk_e_begin.p;
    pv.ucell = d.ucell.p; pv.ent_region = d.ent_region.p; pv.ent_seg_begin = d.ent_seg_begin.p;
    pv.seg_u = d.seg_u.p;
    if constexpr (sizeof(T) == 4) { pv.seg_w = d.seg_w32.p; pv.den = d.den32.p; }
    else { pv.seg_w = d.seg_w64.p; pv.den = d.den64.p; }
    pv.n_groups = (int)plan->info.n_groups;
    if (Ttot == 0) return WAGG_OK;
    const int64_t n_tb = (Ttot + TB - 1) / TB;
    // (T x R) results: gather kernel writes region-major into a stream-ordered workspace, then
    // one transpose; (R x T) results go straight to the caller's buffer
    T *ws = nullptr;
    int64_t ldws = 0;
    T *kout = out;
    int64_t kldo = ldo;
    const bool via_ws = out_layout == WAGG_OUT_TR && plan->info.n_groups > 0;
    if (via_ws) {
        ldws = (Ttot + 63) / 64 * 64;
        WAGG_HIP(hipMallocAsync((void **)&ws, sizeof(T) * (size_t)(ldws * plan->info.R), stream));
        kout = ws;
        kldo = ldws;
    }
    if (plan->info.n_groups > 0) {
        const int64_t nblk = plan->info.n_groups * n_tb;
        WAGG_REQUIRE(nblk < (int64_t)0x7fffffff, "grid too large: %lld", (long long)nblk);
        const size_t shmem = SparseLds<T, TB>::total;
        dim3 grid((unsigned)nblk), block(UC);
#define WAGG_LAUNCH(L, O, V)                                                                    \
        do {                                                                                     \
            auto kern = sparse_gather_kernel<T, TB, L, O, V>;                                    \
            if (const char *dbg_ = getenv("WAGG_SPARSE_DBG")) {                                  \
                switch (atoi(dbg_)) {                                                            \
                    case 1: kern = sparse_gather_kernel<T, TB, L, O, V, 1>; break;               \
                    case 2: kern = sparse_gather_kernel<T, TB, L, O, V, 2>; break;               \
                    case 4: kern = sparse_gather_kernel<T, TB, L, O, V, 4>; break;               \
                    case 6: kern = sparse_gather_kernel<T, TB, L, O, V, 6>; break;               \
                    case 7: kern = sparse_gather_kernel<T, TB, L, O, V, 7>; break;               \
                    default: break;                                                              \
                }                                                                                \
            }                                                                                    \
            WAGG_HIP(hipFuncSetAttribute((const void *)kern,                                     \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            profile_mark(stream, true);                                                          \
            hipLaunchKernelGGL(kern, grid, block, shmem, stream, pv, X, Ttot, ldx, plan->info.G, \
                               kout, kldo);                                                      \
            profile_mark(stream, false);                                                         \
        } while (0)
        // aligned fast path: 16-byte aligned rows (fp32: ldx % 4 == 0; fp64 quads are 32 bytes but
        // are fetched as two 16-byte halves, so the same condition on bytes applies)
        const bool vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((ldx * sizeof(T)) % 16 == 0) &&
                         (sizeof(T) == 4 || true);
        if (layout == WAGG_LAYOUT_TG) { if (vec) WAGG_LAUNCH(WAGG_LAYOUT_TG, WAGG_OUT_RT, true); else WAGG_LAUNCH(WAGG_LAYOUT_TG, WAGG_OUT_RT, false); }
        else WAGG_LAUNCH(WAGG_LAYOUT_GT, WAGG_OUT_RT, false);
#undef WAGG_LAUNCH
        WAGG_HIP(hipGetLastError());
    }
    if (via_ws) {
        dim3 tg((unsigned)((plan->info.R + 63) / 64), (unsigned)((Ttot + 63) / 64));
        hipLaunchKernelGGL((transpose_rt_to_tr_kernel<T>), tg, dim3(256), 0, stream, ws, ldws,
                           (int64_t)plan->info.R, Ttot, out, ldo);
        WAGG_HIP(hipGetLastError());
        WAGG_HIP(hipFreeAsync(ws, stream));
    }
    if (plan->info.n_empty > 0) {
        const int64_t n = plan->info.n_empty * Ttot;
        const T *den;
        if constexpr (sizeof(T) == 4) den = d.den32.p; else den = d.den64.p;
        hipLaunchKernelGGL((fill_empty_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, d.empty_regions.p, (int)plan->info.n_empty, den, Ttot, out, ldo,
                           out_layout);
        WAGG_HIP(hipGetLastError());
    }
    return WAGG_OK;
}

static int check_apply_args(const wagg_plan *plan, const void *X, int64_t T, int64_t ldx, int layout,
                            const void *out, int64_t ldo, int out_layout) {
    WAGG_REQUIRE(plan != nullptr, "plan is NULL");
    WAGG_REQUIRE(T >= 0, "T < 0");
    WAGG_REQUIRE(layout == WAGG_LAYOUT_TG || layout == WAGG_LAYOUT_GT, "bad layout %d", layout);
    WAGG_REQUIRE(out_layout == WAGG_OUT_TR || out_layout == WAGG_OUT_RT, "bad out_layout %d", out_layout);
    if (T == 0) return WAGG_OK;
    WAGG_REQUIRE(X != nullptr && out != nullptr, "X/out is NULL");
    WAGG_REQUIRE(ldx >= (layout == WAGG_LAYOUT_TG ? plan->info.G : T), "ldx %lld too small", (long long)ldx);
    WAGG_REQUIRE(ldo >= (out_layout == WAGG_OUT_TR ? (int64_t)plan->info.R : T), "ldo %lld too small", (long long)ldo);
    return WAGG_OK;
}

}  // namespace wagg

// ---------------------------------------------------------------------------------------------
// host plan builder
// ---------------------------------------------------------------------------------------------
extern "C" int wagg_plan_create(const int32_t *cell_idx, const int32_t *region_code,
                                const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                int64_t row_len, int flags, wagg_plan **out) {
    using namespace wagg;
    (void)flags;
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(nseg >= 0 && G > 0 && R >= 0, "bad sizes nseg=%lld G=%lld R=%d", (long long)nseg,
                 (long long)G, R);
    WAGG_REQUIRE(G < (int64_t)0x7fffffff, "G must fit int32");
    WAGG_REQUIRE(nseg == 0 || (cell_idx && region_code && w_eff), "NULL segment arrays");
    if (row_len <= 0 || row_len > G) row_len = G;

    struct Seg { int32_t region, cell; double w; };
    std::vector<Seg> segs;
    std::vector<double> den((size_t)R, 0.0);
    try {
        segs.reserve((size_t)nseg);
        for (int64_t i = 0; i < nseg; ++i) {
            const int32_t r = region_code[i];
            if (r < 0) continue;                                  // null label (S3)
            WAGG_REQUIRE(r < R, "region_code[%lld]=%d out of range [0,%d)", (long long)i, r, R);
            WAGG_REQUIRE(cell_idx[i] >= 0 && cell_idx[i] < G, "cell_idx[%lld]=%d out of range",
                         (long long)i, cell_idx[i]);
            if (std::isnan(w_eff[i])) continue;                   // skipna on :78/:79
            den[(size_t)r] += w_eff[i];                           // aggregations.py:79
            segs.push_back({r, cell_idx[i], w_eff[i]});
        }
        std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) {
            return a.region != b.region ? a.region < b.region : a.cell < b.cell;
        });
        // coalesce duplicate (cell, region) rows (S5)
        size_t m = 0;
        for (size_t i = 0; i < segs.size(); ++i) {
            if (m && segs[m - 1].region == segs[i].region && segs[m - 1].cell == segs[i].cell)
                segs[m - 1].w += segs[i].w;
            else segs[m++] = segs[i];
        }
        segs.resize(m);
        const int64_t nnz = (int64_t)m;

        // per-region ranges + spatial key
        std::vector<int64_t> rbeg((size_t)R + 1, 0);
        for (const Seg &s : segs) rbeg[(size_t)s.region + 1]++;
        for (int32_t r = 0; r < R; ++r) rbeg[(size_t)r + 1] += rbeg[(size_t)r];
        std::vector<int32_t> order, empty;
        std::vector<double> key_col((size_t)R, 0.0);
        std::vector<int64_t> key_band((size_t)R, 0);
        for (int32_t r = 0; r < R; ++r) {
            const int64_t n = rbeg[(size_t)r + 1] - rbeg[(size_t)r];
            if (n == 0) { empty.push_back(r); continue; }
            double srow = 0, scol = 0;
            for (int64_t i = rbeg[(size_t)r]; i < rbeg[(size_t)r + 1]; ++i) {
                srow += (double)(segs[(size_t)i].cell / row_len);
                scol += (double)(segs[(size_t)i].cell % row_len);
            }
            key_band[(size_t)r] = (int64_t)(srow / (double)n) / 8;
            key_col[(size_t)r] = scol / (double)n;
            order.push_back(r);
        }
        std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
            if (key_band[(size_t)a] != key_band[(size_t)b]) return key_band[(size_t)a] < key_band[(size_t)b];
            if (key_col[(size_t)a] != key_col[(size_t)b]) return key_col[(size_t)a] < key_col[(size_t)b];
            return a < b;
        });

        // greedy grouping on aligned 4-cell QUADS (16 bytes of a fp32 row): the kernel fetches
        // a chunk as up to UQ quads per timestep with one dwordx4 per lane
        struct Group { std::vector<int32_t> regions; int64_t n_q = 0, n_seg = 0; bool giant = false; };
        std::vector<Group> groups;
        std::vector<int32_t> stamp((size_t)(G + 3) / 4, -1);
        auto quads_of = [&](int32_t r) {       // distinct quads of a region (its cells are sorted)
            int64_t n = 0;
            int32_t last = -1;
            for (int64_t i = rbeg[(size_t)r]; i < rbeg[(size_t)r + 1]; ++i) {
                const int32_t q = segs[(size_t)i].cell >> 2;
                if (q != last) { ++n; last = q; }
            }
            return n;
        };
        Group cur;
        int32_t cur_id = 0;
        auto close = [&]() {
            if (!cur.regions.empty()) { groups.push_back(std::move(cur)); cur = Group(); }
            ++cur_id;
        };
        int64_t n_giant = 0;
        for (int32_t r : order) {
            const int64_t b = rbeg[(size_t)r], e = rbeg[(size_t)r + 1];
            const int64_t nq_r = quads_of(r);
            if (nq_r > UQ || e - b > SEG_MAX) {
                close();
                Group gg; gg.regions.push_back(r); gg.n_q = nq_r; gg.giant = true;
                groups.push_back(std::move(gg));
                ++n_giant;
                continue;
            }
            int64_t fresh = 0;
            {
                int32_t last = -1;
                for (int64_t i = b; i < e; ++i) {
                    const int32_t q = segs[(size_t)i].cell >> 2;
                    if (q != last) { fresh += stamp[(size_t)q] != cur_id; last = q; }
                }
            }
            if (cur.n_q + fresh > UQ || (int64_t)cur.regions.size() + 1 > RG_MAX ||
                cur.n_seg + (e - b) > SEG_MAX) {
                close();
                fresh = nq_r;
            }
            for (int64_t i = b; i < e; ++i) stamp[(size_t)(segs[(size_t)i].cell >> 2)] = cur_id;
            cur.n_q += fresh;
            cur.n_seg += e - b;
            cur.regions.push_back(r);
        }
        close();
        // giant groups (many chunks) first, largest first; all other groups stay in band/column
        // order so that consecutive workgroups touch neighbouring cells of the same grid rows
        std::stable_sort(groups.begin(), groups.end(), [](const Group &a, const Group &b) {
            if (a.giant != b.giant) return a.giant;
            return a.giant && a.n_q > b.n_q;
        });

        // flatten
        std::vector<int32_t> grp_chunk_begin{0}, grp_giant, chunk_u_begin{0}, chunk_e_begin{0};
        std::vector<int32_t> ucell, ent_region, ent_seg_begin{0}, seg_u;   // ucell = first cell of each quad
        std::vector<double> seg_w;
        seg_u.reserve((size_t)nnz); seg_w.reserve((size_t)nnz);
        std::vector<int32_t> &pos = stamp;  // reuse as quad -> local index scratch
        std::vector<int32_t> quads;
        for (const Group &gr : groups) {
            grp_giant.push_back(gr.giant ? 1 : 0);
            if (gr.giant) {
                const int32_t r = gr.regions[0];
                const int64_t b = rbeg[(size_t)r], e = rbeg[(size_t)r + 1];
                int64_t cb = b;
                while (cb < e) {
                    // take segments until the chunk holds UQ quads
                    int64_t ce = cb, nq = 0;
                    int32_t last = -1;
                    const size_t ubase = ucell.size();
                    while (ce < e) {
                        const int32_t q = segs[(size_t)ce].cell >> 2;
                        if (q != last) {
                            if (nq == UQ) break;
                            ucell.push_back(q * 4);
                            ++nq; last = q;
                        }
                        ++ce;
                    }
                    chunk_u_begin.push_back((int32_t)ucell.size());
                    // split the chunk's segments over the waves
                    const int64_t n = ce - cb, per = (n + NWAVE - 1) / NWAVE;
                    for (int64_t sb = 0; sb < n; sb += per) {
                        const int64_t se = std::min<int64_t>(sb + per, n);
                        for (int64_t i = sb; i < se; ++i) {
                            const int32_t cell = segs[(size_t)(cb + i)].cell;
                            // local quad index: position of cell>>2 among this chunk's quads
                            const auto it = std::lower_bound(ucell.begin() + (std::ptrdiff_t)ubase, ucell.end(),
                                                             (cell >> 2) * 4);
                            seg_u.push_back((int32_t)((it - (ucell.begin() + (std::ptrdiff_t)ubase)) * 4 + (cell & 3)));
                            seg_w.push_back(segs[(size_t)(cb + i)].w);
                        }
                        ent_region.push_back(r);
                        ent_seg_begin.push_back((int32_t)seg_u.size());
                    }
                    chunk_e_begin.push_back((int32_t)ent_region.size());
                    cb = ce;
                }
            } else {
                quads.clear();
                for (int32_t r : gr.regions)
                    for (int64_t i = rbeg[(size_t)r]; i < rbeg[(size_t)r + 1]; ++i)
                        quads.push_back(segs[(size_t)i].cell >> 2);
                std::sort(quads.begin(), quads.end());
                quads.erase(std::unique(quads.begin(), quads.end()), quads.end());
                for (size_t i = 0; i < quads.size(); ++i) {
                    pos[(size_t)quads[i]] = (int32_t)i;
                    ucell.push_back(quads[i] * 4);
                }
                chunk_u_begin.push_back((int32_t)ucell.size());
                for (int32_t r : gr.regions) {
                    for (int64_t i = rbeg[(size_t)r]; i < rbeg[(size_t)r + 1]; ++i) {
                        const int32_t cell = segs[(size_t)i].cell;
                        seg_u.push_back(pos[(size_t)(cell >> 2)] * 4 + (cell & 3));
                        seg_w.push_back(segs[(size_t)i].w);
                    }
                    ent_region.push_back(r);
                    ent_seg_begin.push_back((int32_t)seg_u.size());
                }
                chunk_e_begin.push_back((int32_t)ent_region.size());
            }
            grp_chunk_begin.push_back((int32_t)(chunk_u_begin.size() - 1));
        }

        wagg_plan *plan = new wagg_plan();
        plan->den_host = den;
        plan->info.nseg_in = nseg; plan->info.nnz = nnz;
        plan->info.n_groups = (int64_t)groups.size();
        plan->info.n_chunks = (int64_t)chunk_u_begin.size() - 1;
        plan->info.n_ucells = (int64_t)ucell.size() * 4;    // cells fetched per timestep (whole quads)
        plan->info.n_giant = n_giant;
        plan->info.n_empty = (int64_t)empty.size();
        plan->info.G = G; plan->info.R = R;
        std::vector<float> seg_w32(seg_w.size()), den32(den.size());
        for (size_t i = 0; i < seg_w.size(); ++i) seg_w32[i] = (float)seg_w[i];
        for (size_t i = 0; i < den.size(); ++i) den32[i] = (float)den[i];
        (void)hipGetDevice(&plan->device);
        auto &d = plan->d;
        hipError_t he = hipSuccess;
        auto up = [&](auto &buf, const auto &h) { if (he == hipSuccess) he = buf.upload(h); };
        up(d.grp_chunk_begin, grp_chunk_begin); up(d.grp_giant, grp_giant);
        up(d.chunk_u_begin, chunk_u_begin); up(d.chunk_e_begin, chunk_e_begin);
        up(d.ucell, ucell); up(d.ent_region, ent_region); up(d.ent_seg_begin, ent_seg_begin);
        up(d.seg_u, seg_u); up(d.seg_w32, seg_w32); up(d.seg_w64, seg_w);
        up(d.den32, den32); up(d.den64, den); up(d.empty_regions, empty);
        if (he != hipSuccess) {
            set_error("plan upload failed: %s", hipGetErrorString(he));
            delete plan;
            return WAGG_EHIP;
        }
        *out = plan;
        return WAGG_OK;
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed while building the plan");
        return WAGG_ENOMEM;
    }
}

extern "C" int wagg_plan_destroy(wagg_plan *plan) {
    delete plan;
    return WAGG_OK;
}

extern "C" int wagg_plan_get_info(const wagg_plan *plan, wagg_plan_info *info) {
    WAGG_REQUIRE(plan && info, "NULL argument");
    *info = plan->info;
    return WAGG_OK;
}

extern "C" int wagg_plan_get_den(const wagg_plan *plan, double *den_host) {
    WAGG_REQUIRE(plan && (den_host || plan->info.R == 0), "NULL argument");
    if (plan->info.R) std::memcpy(den_host, plan->den_host.data(), sizeof(double) * (size_t)plan->info.R);
    return WAGG_OK;
}

extern "C" int wagg_apply_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx,
                              int layout, float *out_dev, int64_t ldo, int out_layout, void *stream) {
    int rc = wagg::check_apply_args(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    return wagg::launch_sparse<float, 64>(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout,
                                          (hipStream_t)stream);
}

extern "C" int wagg_apply_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx,
                              int layout, double *out_dev, int64_t ldo, int out_layout, void *stream) {
    int rc = wagg::check_apply_args(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    return wagg::launch_sparse<double, 32>(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout,
                                           (hipStream_t)stream);
}

namespace wagg {
template <typename T, typename F>
static int apply_host(const wagg_plan *plan, const T *X, int64_t Tn, int64_t ldx, int layout, T *out,
                      int64_t ldo, int out_layout, F fn) {
    int rc = check_apply_args(plan, X, Tn, ldx, layout, out, ldo, out_layout);
    if (rc != WAGG_OK || Tn == 0) return rc;
    const int64_t xrows = layout == WAGG_LAYOUT_TG ? Tn : plan->info.G;
    const int64_t orows = out_layout == WAGG_OUT_TR ? Tn : plan->info.R;
    DevBuf<T> dx, dout;
    WAGG_HIP(dx.alloc((size_t)(xrows * ldx)));
    WAGG_HIP(dout.alloc((size_t)(orows * ldo)));
    WAGG_HIP(hipMemcpy(dx.p, X, sizeof(T) * (size_t)(xrows * ldx), hipMemcpyHostToDevice));
    WAGG_HIP(hipMemset(dout.p, 0, sizeof(T) * (size_t)(orows * ldo)));
    rc = fn(plan, dx.p, Tn, ldx, layout, dout.p, ldo, out_layout, nullptr);
    if (rc != WAGG_OK) return rc;
    WAGG_HIP(hipDeviceSynchronize());
    WAGG_HIP(hipMemcpy(out, dout.p, sizeof(T) * (size_t)(orows * ldo), hipMemcpyDeviceToHost));
    return WAGG_OK;
}
}  // namespace wagg

extern "C" int wagg_apply_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx,
                                   int layout, float *out_host, int64_t ldo, int out_layout) {
    return wagg::apply_host<float>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, wagg_apply_f32);
}
extern "C" int wagg_apply_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx,
                                   int layout, double *out_host, int64_t ldo, int out_layout) {
    return wagg::apply_host<double>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, wagg_apply_f64);
}
