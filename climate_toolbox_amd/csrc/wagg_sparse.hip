// Sparse path: coded segment table -> region-grouped gather plan -> HIP kernels.
//
// Replaces aggregations.py:24-27 (gather) and :78-80 (grouped sums + division) of the reference
// with   out[t,r] = sum_{i in r} X[t, cell_i] * w_i / den[r]   computed WITHOUT materialising
// the (T x nseg) gathered copy.  HBM-bound; what is in this file:
//
//   * host plan builder (wagg_plan_create): coalesce duplicate (cell, region) rows, order regions
//     along 8-row latitude bands, pack neighbouring regions into GROUPS whose union of cells is at
//     most 64 aligned 4-cell quads (UC = 256 cell slots: one LDS "chunk", one 16-byte load per
//     quad and timestep); regions larger than a chunk become "giant" groups that walk several.
//   * sparse_lcv_kernel    fp32 and fp64, either layout, single-chunk groups (the production path): 8 loader waves
//                          stream whole 128-byte lines into a double-buffered, swizzled LDS image, 8 consumer waves
//                          reduce it on the vector ALU (up to four fused powers / degree-day thresholds per pass).
//   * sparse_stream_kernel fp64 (and fp32 on request): persistent, software-pipelined, reduction by
//                          v_readlane-broadcast segment lists.
//   * sparse_gather_kernel the chunk-walking form: giant groups, (gridcell, time) data, two-field
//                          degree-day transform (up to four thresholds per pass).
//   * transpose / fill kernels for the (time, region) result layout and regions without rows.
//
// (The round-1/2 kernel with MFMA consumers, sparse_lc_kernel, is in wagg_sparse_diag.hip: diagnostic build only.)
// Common to all: a t-major LDS image of the item (swizzled or padded rows), NaN products count 0
// (S6), the division by den[r] (aggregations.py:79-80) is fused, results are stored once, no
// atomics anywhere: bitwise reproducible.  docs/HISTORY.md section (d) has the measurements behind the
// shapes chosen here.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <memory>
#include <mutex>
#include <thread>
#include <numeric>
#include <utility>

#include "wagg_sparse_int.h"
#include "wagg_entry.h"

namespace wagg {

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
// LDS carve-up (bytes):  xs [TB][UROW] T | red [NWAVE][TB] T | seg_w [SEG_MAX] T | ent_den [RG_MAX+1] T |
//                        ent_r [RG_MAX+1] i32 | seg_u [SEG_MAX] u16 | ent_s [RG_MAX+1] u16
// f32/TB=64: 75.8 KB, f64/TB=32: 78.9 KB  ->  two workgroups per CU (160 KB).
template <typename T, int TB> struct SparseLds {
    static constexpr size_t xs = 0;
    static constexpr size_t red = xs + sizeof(T) * TB * urow<T>();
    static constexpr size_t seg_w = red + sizeof(T) * NWAVE * TB;
    static constexpr size_t ent_den = seg_w + sizeof(T) * SEG_MAX;
    static constexpr size_t ent_r = ent_den + sizeof(T) * (RG_MAX + 1);
    static constexpr size_t seg_u = ent_r + sizeof(int32_t) * (RG_MAX + 1);
    static constexpr size_t ent_s = seg_u + sizeof(uint16_t) * SEG_MAX;
    static constexpr size_t total = (ent_s + sizeof(uint16_t) * (RG_MAX + 1) + 15) / 16 * 16;
    static_assert(total <= 80 * 1024, "two workgroups must fit one CU's LDS");
};

// DBG: diagnostic knob (WAGG_SPARSE_DBG env, never set in production; results are wrong with
// any bit set): bit0 = no gather loads, bit1 = no LDS image stores, bit2 = no segment loop.
// VEC: rows of X are 16-byte aligned (base and ldx), so a quad is one aligned vector load.
// NTHR > 1 (Snyder degree days, several thresholds): the two fields of a chunk are loaded ONCE and
// the image / reduction phase runs once per threshold.
template <typename T, int TB, int LAYOUT, int OUT_LAYOUT, bool VEC, int DBG = 0, int NTHR = 1>
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

    T giant_acc[NTHR];
#pragma unroll
    for (int k = 0; k < NTHR; ++k) giant_acc[k] = T(0);
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
        vec4 v[TPW], hi[NTHR > 1 ? TPW : 1];
        const int tw0 = wave * TPW;
        if constexpr (LAYOUT == WAGG_LAYOUT_TG) {
            if (lane < nu) {
                const int64_t cell0 = pv.ucell[u0 + lane] & ~UCELL_UNREF;          // (whole-line chunkings flag unreferenced quads)
                const int64_t lim = G - 1 - cell0;                   // >= 0
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int tc = (tw0 + i < nt) ? tw0 + i : nt - 1;
                    const T *p = X + (t0 + tc) * ldx + cell0;
                    if (DBG & 1) v[i] = vec4{T(i), T(i), T(i), T(i)};
                    else if (VEC) v[i] = *reinterpret_cast<const vec4 *>(p);
                    else                                             // unaligned rows / ragged grid end
                        v[i] = vec4{p[0], p[lim < 1 ? lim : 1], p[lim < 2 ? lim : 2], p[lim < 3 ? lim : 3]};
                }
                if (pv.xpow > 0) {
#pragma unroll
                    for (int i = 0; i < TPW; ++i) v[i] = xform4<vec4, T>(v[i], pv.xoff, pv.xpow);
                } else if (pv.xpow == XF_EDD) {
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const int tc = (tw0 + i < nt) ? tw0 + i : nt - 1;
                        const T *p2 = pv.X2 + (t0 + tc) * ldx + cell0;
                        vec4 h;
                        if (VEC) h = *reinterpret_cast<const vec4 *>(p2);
                        else h = vec4{p2[0], p2[lim < 1 ? lim : 1], p2[lim < 2 ? lim : 2], p2[lim < 3 ? lim : 3]};
                        if constexpr (NTHR > 1) { v[i] = v[i] + pv.xoff; hi[i] = h + pv.xoff; }   // kept for every threshold
                        else {
#pragma unroll
                            for (int cc = 0; cc < 4; ++cc) v[i][cc] = snyder_edd1<T>(v[i][cc] + pv.xoff, h[cc] + pv.xoff, pv.edd_thr[0]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < NTHR; ++k) {
            if (NTHR > 1 && k >= pv.n_thr) break;
            if (k > 0) __syncthreads();                              // the previous threshold's sums are done with xs
            if constexpr (LAYOUT == WAGG_LAYOUT_TG) {
                if (lane < nu) {
                    if (!(DBG & 2)) {
#pragma unroll
                        for (int i = 0; i < TPW; ++i) {
                            vec4 val = v[i];
                            if constexpr (NTHR > 1) {
#pragma unroll
                                for (int cc = 0; cc < 4; ++cc) val[cc] = snyder_edd1<T>(v[i][cc], hi[i][cc], pv.edd_thr[k]);
                            }
                            *reinterpret_cast<vec4 *>(&xs[(tw0 + i) * urow<T>() + 4 * lane]) = val;
                        }
                    } else {
                        vec4 sum = vec4{T(0), T(0), T(0), T(0)};
#pragma unroll
                        for (int i = 0; i < TPW; ++i) sum += v[i];
                        *reinterpret_cast<vec4 *>(&xs[tw0 * urow<T>() + 4 * lane]) = sum;
                    }
                }
            } else {
                for (int u = wave; u < 4 * nu; u += NWAVE) {
                    int64_t cell = (int64_t)(pv.ucell[u0 + (u >> 2)] & ~UCELL_UNREF) + (u & 3);
                    cell = cell < G ? cell : G - 1;
                    if (lane < TB) {
                        T xv = lane_live ? X[cell * ldx + t0 + lane] : T(0);
                        if (pv.xpow > 0) xv = xform1<T>(xv, pv.xoff, pv.xpow);
                        else if (pv.xpow == XF_EDD)
                            xv = snyder_edd1<T>(xv + pv.xoff, (lane_live ? pv.X2[cell * ldx + t0 + lane] : T(0)) + pv.xoff,
                                                pv.edd_thr[k]);
                        xs[lane * urow<T>() + u] = xv;
                    }
                }
            }
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < (SEG_MAX + UC - 1) / UC; ++i) {
                    const int kk = tid + UC * i;
                    if (kk < ns) { sm_u[kk] = (uint16_t)mu[i]; sm_w[kk] = mw[i]; }
                }
                if (tid < ne) sm_er[tid] = er;
                if (tid <= ne) sm_es[tid] = (uint16_t)es;
            }
            __syncthreads();
            // ---- weighted group sums: one wave per region, lane = timestep; the (cell, weight) list
            // is read from LDS at wave-uniform addresses (broadcast) ----
            T *outk = out + (NTHR > 1 ? (int64_t)k * pv.thr_pstride : 0);
            for (int el = wave; el < ne; el += NWAVE) {
                const int s0 = __builtin_amdgcn_readfirstlane((int)sm_es[el]);
                const int s1 = __builtin_amdgcn_readfirstlane((int)sm_es[el + 1]);
                const int r = __builtin_amdgcn_readfirstlane(sm_er[el]);
                T den = T(1);
                if (!giant) den = pv.den[r];                         // in flight during the segment loop
                T acc = T(0);
                if (lane < TB && !(DBG & 4)) {
#pragma unroll 8
                    for (int sgi = s0; sgi < s1; ++sgi) {
                        const int u = sm_u[sgi] & SEG_UMASK;
                        const T w = sm_w[sgi];
                        const T p = xs[lane * urow<T>() + u] * w;        // aggregations.py:78 product
                        acc += (p == p) ? p : T(0);                 // skipna: NaN product counts 0 (S6)
                    }
                }
                if (giant) {
                    giant_acc[k] += acc;
                } else if (lane_live) {
                    const T q = acc / den;                          // aggregations.py:77-80, S7
                    if constexpr (OUT_LAYOUT == WAGG_OUT_TR) outk[(t0 + lane) * ldo + r] = q;
                    else outk[(int64_t)r * ldo + t0 + lane] = q;
                }
            }
        }
        __syncthreads();
    }
    if (giant) {
#pragma unroll
        for (int k = 0; k < NTHR; ++k) {
            if (NTHR > 1 && k >= pv.n_thr) break;
            if (k > 0) __syncthreads();
            if (lane < TB) red[wave * TB + lane] = giant_acc[k];
            __syncthreads();
            if (wave == 0 && lane_live) {
                T sred = red[lane];
#pragma unroll
                for (int w = 1; w < NWAVE; ++w) sred += red[w * TB + lane];
                const int r = pv.ent_region[pv.chunk_e_begin[c0]];
                const T q = sred / pv.den[r];
                T *outk = out + (NTHR > 1 ? (int64_t)k * pv.thr_pstride : 0);
                if constexpr (OUT_LAYOUT == WAGG_OUT_TR) outk[(t0 + lane) * ldo + r] = q;
                else outk[(int64_t)r * ldo + t0 + lane] = q;
            }
        }
    }
}

// Persistent, software-pipelined form of the gather kernel for the single-chunk ("normal") groups,
// TG layout.  NW workgroups (two per CU) each walk items it = w, w+NW, ...; item = (chunk, time
// block).  While item i is reduced out of LDS, the 64 KB of item i+1 are already in flight into
// registers (issued right after item i's registers were parked in LDS), the quad list of item
// i+2 and the descriptor of item i+3 are being fetched -- the dependent metadata chain and the
// LDS/compute phase no longer sit between two bursts of HBM requests.

// STAMP (diagnostic build only, WAGG_SPARSE_STAMP env): per-phase s_memtime sums of wave 0 go to a
// debug buffer that nothing else reads.
template <typename T, int TB, bool VEC, bool STAMP = false>
__global__ __launch_bounds__(STHREADS, 4) void sparse_stream_kernel(PlanView<T> pv, const T *__restrict__ X,
                                                              int64_t Ttot, int64_t ldx, int64_t G,
                                                              T *__restrict__ out, int64_t ldo,
                                                              int n_norm, long long n_items,
                                                              unsigned long long *__restrict__ stamps,
                                                              int knob) {
    unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int i) {
        if (STAMP) {
            __builtin_amdgcn_sched_barrier(0);
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i >= 0) ph[i] += tnow - tprev;
            tprev = tnow;
        }
    };
    using L = SparseLds<T, TB>;
    constexpr int TPW = TB / SWAVE;
    constexpr int NM = (SEG_MAX + STHREADS - 1) / STHREADS;
    typedef T vec4 __attribute__((ext_vector_type(4)));
    typedef int int4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T *xs = reinterpret_cast<T *>(smem_raw + L::xs);
    T *sm_w = reinterpret_cast<T *>(smem_raw + L::seg_w);
    T *sm_ed = reinterpret_cast<T *>(smem_raw + L::ent_den);
    int32_t *sm_er = reinterpret_cast<int32_t *>(smem_raw + L::ent_r);
    uint16_t *sm_u = reinterpret_cast<uint16_t *>(smem_raw + L::seg_u);
    uint16_t *sm_es = reinterpret_cast<uint16_t *>(smem_raw + L::ent_s);
    int *sm_flag = reinterpret_cast<int *>(smem_raw + L::red);    // the giant-group scratch is free here

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tw0 = wave * TPW;
    if (tid == 0) *sm_flag = 0;
    lds_only_barrier();
    // XCD-contiguous ids (speed only): at every step the NW resident workgroups cover a contiguous
    // run of items, and each XCD a contiguous part of it
    const unsigned NWu = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = NWu >> 3, r8 = NWu & 7u;
    const long long NW = NWu;
    const long long w0 = (long long)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot);
    if (w0 >= n_items) return;
    // item -> (chunk g, time block tb) without divisions in the loop: items advance by NW, i.e. by
    // (dg, dtb) with a carry; everything is 32-bit (n_items < 2^31 is checked on the host)
    const int dg = (int)(NW % n_norm), dtb = (int)(NW / n_norm);
    const int nst = (int)((n_items - 1 - w0) / NW) + 1;          // stages of this workgroup
    struct Item { int g, tb; };
    auto advance = [&](Item a) {
        Item b{a.g + dg, a.tb + dtb};
        if (b.g >= n_norm) { b.g -= n_norm; ++b.tb; }
        return b;
    };
    const Item i0{(int)(w0 % n_norm), (int)(w0 / n_norm)};

    // Every load below is UNCONDITIONAL (indices clamped into the chunk): no exec-skip branches, so
    // hipcc's s_waitcnt pass sees one straight path and emits counted vmcnt(N) instead of vmcnt(0).
    auto load_desc = [&](Item a, StreamDesc &d) {
        const int32_t *p = pv.chunk_desc + 8 * (int64_t)(pv.c0_normal + a.g);
        const int4v x = *reinterpret_cast<const int4v *>(p);
        const int4v y = *reinterpret_cast<const int4v *>(p + 4);
        d.u0 = x[0]; d.nq = x[1]; d.e0 = x[2]; d.ne = x[3]; d.sb = y[0]; d.ns = y[1];
        d.split = (unsigned long long)(unsigned)y[2] | ((unsigned long long)(unsigned)y[3] << 32);
    };
    auto load_cell = [&](const StreamDesc &d) {
        return pv.ucell[d.u0 + (lane < d.nq ? lane : d.nq - 1)] & ~UCELL_UNREF;     // (whole-line chunkings flag unreferenced quads)
    };
    vec4 v[TPW];
    auto issue_x = [&](int cell0, int tb) {
        const int64_t t0 = (int64_t)tb * TB;
        const int nt = (int)((Ttot - t0) < TB ? (Ttot - t0) : TB);
        // rows of this wave: tw0 .. tw0+TPW-1, clamped into the (possibly ragged) block
        const int rbase = tw0 < nt - 1 ? tw0 : nt - 1;
        int cnt = nt - tw0;
        cnt = cnt < 1 ? 1 : (cnt > TPW ? TPW : cnt);
        const T *p = X + (t0 + rbase) * ldx + cell0;
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (VEC) v[i] = *reinterpret_cast<const vec4 *>(p);
            else {
                const int64_t lim = G - 1 - cell0;
                v[i] = vec4{p[0], p[lim < 1 ? lim : 1], p[lim < 2 ? lim : 2], p[lim < 3 ? lim : 3]};
            }
            if (i + 1 < cnt) p += ldx;                                 // wave-uniform step
        }
    };
    int mu[NM]; T mw[NM]; int er, es; T ed;
    auto issue_meta = [&](const StreamDesc &d) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            int k = tid + STHREADS * i;
            k = k < d.ns ? k : d.ns - 1;
            mu[i] = pv.seg_u[d.sb + k];
            mw[i] = pv.seg_w[d.sb + k];
        }
        er = pv.ent_region[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
        ed = pv.ent_den[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
        es = pv.ent_seg_begin[d.e0 + (tid < d.ne ? tid : d.ne)];     // absolute; rebased when parked
    };

    // pipeline registers: C = current item, A = next, B = next-next, (D = the one after)
    Item iC = i0, iA = nst > 1 ? advance(iC) : iC, iB = nst > 2 ? advance(iA) : iA;
    StreamDesc dC, dA, dB;
    load_desc(iC, dC);
    load_desc(iA, dA);
    load_desc(iB, dB);
    int cellA = load_cell(dA);
    issue_meta(dC);
    issue_x(load_cell(dC), iC.tb);

    // results of the previous item: up to KQ regions per wave are kept in registers and stored at
    // the START of the next stage's traffic block (older than that block's loads, so no later
    // wait ever depends on a store acknowledgement; vmcnt retires in order and counts stores)
    constexpr int KQ = 8;
    T q_prev[KQ];
    int r_prev[KQ];
    int ec_prev = 0;
    int64_t t0_prev = 0;
    bool live_prev = false;
#pragma unroll
    for (int k = 0; k < KQ; ++k) { q_prev[k] = T(0); r_prev[k] = 0; }
    auto flush_prev = [&]() {
#pragma unroll
        for (int k = 0; k < KQ; ++k)
            if (k < ec_prev && live_prev) out[(int64_t)r_prev[k] * ldo + t0_prev + lane] = q_prev[k];
    };

    auto stage = [&](auto pf_tag) {
        constexpr bool PF = decltype(pf_tag)::value;
        const int64_t t0 = (int64_t)iC.tb * TB;
        const int nt = (int)((Ttot - t0) < TB ? (Ttot - t0) : TB);
        const bool lane_live = lane < nt && lane < TB;
        stamp(-1);
        if (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(0); }   // ph0: wait for the rows
        // ---- park the current item (registers -> LDS); note whether the whole chunk is finite ----
        bool odd = false;
        if (pv.xpow > 0) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) v[i] = xform4<vec4, T>(v[i], pv.xoff, pv.xpow);
        }
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            *reinterpret_cast<vec4 *>(&xs[(tw0 + i) * urow<T>() + 4 * lane]) = v[i];
            const vec4 z = v[i] - v[i];                         // 0 for finite values, NaN for NaN / inf
            odd |= !(z[0] == T(0) && z[1] == T(0) && z[2] == T(0) && z[3] == T(0));
        }
        if (__builtin_amdgcn_readfirstlane(__ballot(odd) != 0ull)) *sm_flag = 1;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            const int k = tid + STHREADS * i;
            if (k < dC.ns) { sm_u[k] = (uint16_t)mu[i]; sm_w[k] = mw[i]; }
        }
        if (tid < dC.ne) { sm_er[tid] = er; sm_ed[tid] = ed; }
        if (tid <= dC.ne) sm_es[tid] = (uint16_t)(es - dC.sb);
        const int ne = dC.ne;
        stamp(1);                                                                   // ph1: park
        // ---- one block of global traffic per stage, in this order (vmcnt retires in order):
        // previous item's results, next items' metadata, next item's rows ----
        if (!(STAMP && (knob & 4))) flush_prev();
        stamp(6);
        StreamDesc dD = dB;
        Item iD = iB;
        int cellB = cellA;
        if (PF) {
            iD = advance(iB);
            if (iD.tb * (long long)n_norm + iD.g >= n_items) iD = iB;     // past the end: any valid item
            load_desc(iD, dD);
            cellB = load_cell(dB);
            stamp(7);
            issue_meta(dA);
            stamp(8);
            if (!(STAMP && (knob & 2))) issue_x(cellA, iA.tb);
        }
        stamp(2);                                                                   // ph2: issue
        lds_only_barrier();
        const bool all_finite = __builtin_amdgcn_readfirstlane(*sm_flag) == 0;
        stamp(3);                                                                   // ph3: barrier 1
        // ---- weighted group sums of the current item out of LDS.  Each wave walks ONE flat run of
        // segments (a contiguous range of region entries with ~1/4 of the chunk's segments): lane j
        // of a 64-segment block holds segment j's (u, w) in registers (one LDS read per lane), the
        // loop broadcasts them with v_readlane so the 8 x-reads of a group are independent, and a
        // flag bit in u closes a region (wave-uniform branch): divide, keep the quotient in a
        // register slot for the deferred store, reset the sum.  Padding lanes hold w = 0, u = 0
        // and add exactly 0, so groups of 8 need no remainder loop. ----
        const unsigned long long split = dC.split;   // captured before any rotation
        const int ea = wave == 0 ? 0 : (int)((split >> (8 * (wave - 1))) & 0xff);
        const int eb = wave == SWAVE - 1 ? ne : (int)((split >> (8 * wave)) & 0xff);
        int ec = 0;                                            // regions closed by this wave
        if (ea < eb) {
            const int sa = __builtin_amdgcn_readfirstlane((int)sm_es[ea]);
            const int se = __builtin_amdgcn_readfirstlane((int)sm_es[eb]);
            T acc = T(0);
            for (int base = sa; base < se; base += 64) {
                const int n = se - base < 64 ? se - base : 64;
                const int ul = lane < n ? (int)sm_u[base + lane] : 0;
                const T wl = lane < n ? sm_w[base + lane] : T(0);
                for (int j0 = 0; j0 < n; j0 += 8) {
                    T xv[8], wv[8];
                    int uf[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {               // 8 independent LDS reads in flight
                        uf[j] = __builtin_amdgcn_readlane(ul, j0 + j);
                        if constexpr (sizeof(T) == 4) {
                            wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), j0 + j));
                        } else {
                            const long long wb = __builtin_bit_cast(long long, wl);
                            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(wb & 0xffffffffll), j0 + j);
                            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(wb >> 32), j0 + j);
                            wv[j] = __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
                        }
                        xv[j] = xs[lane * urow<T>() + (uf[j] & SEG_UMASK)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (all_finite) {                       // finite data: no skipna test needed;
#pragma clang fp contract(off)                                  // same roundings as the general path
                            const T p = xv[j] * wv[j];
                            acc += p;
                        } else {
                            const T p = xv[j] * wv[j];          // aggregations.py:78 product
                            acc += (p == p) ? p : T(0);         // skipna (S6)
                        }
                        if (uf[j] & SEG_LAST) {                 // wave-uniform: region complete
                            const T qv = acc / sm_ed[ea + ec];  // aggregations.py:77-80
                            const int r = __builtin_amdgcn_readfirstlane(sm_er[ea + ec]);
                            if (ec < KQ) {
#pragma unroll
                                for (int k = 0; k < KQ; ++k) {
                                    q_prev[k] = (ec == k) ? qv : q_prev[k];
                                    r_prev[k] = (ec == k) ? r : r_prev[k];
                                }
                            } else if (lane_live) {             // more than KQ regions per wave (rare)
                                out[(int64_t)r * ldo + t0 + lane] = qv;
                            }
                            acc = T(0);
                            ++ec;
                        }
                    }
                }
            }
        }
        ec_prev = ec < KQ ? ec : KQ; t0_prev = t0; live_prev = lane_live;
        stamp(4);                                                                   // ph4: compute
        lds_only_barrier();                                     // every wave has read the flag
        if (tid == 0) *sm_flag = 0;
        if (PF) { dC = dA; dA = dB; dB = dD; cellA = cellB; iC = iA; iA = iB; iB = iD; }
        lds_only_barrier();
        stamp(5);                                                                   // ph5: rotate + barrier 2
    };
    for (int st = 0; st + 1 < nst; ++st) stage(std::true_type{});
    stage(std::false_type{});
    flush_prev();
    if (STAMP && tid == 0) {
#pragma unroll
        for (int i = 0; i < 10; ++i) stamps[blockIdx.x * 10 + i] = ph[i];
    }
}

// ---------------------------------------------------------------------------------------------
// Loader/consumer form with VECTOR-ALU consumers (round 3): (time, gridcell) data, fp32 AND fp64 -- the plain aggregation,
// the fused powers and (fp32) the fused degree days.  One 1024-thread workgroup per CU, persistent over items
// (chunk, 64 timesteps); an item is 64 rows x 1 KiB of the whole-line chunking of the data type:
//   * 8 LOADER waves stream the next item into registers -- lane l fetches 16 bytes of a row, eight lanes one whole
//     128-byte line -- and park it in the other half of a double-buffered LDS image (NaN -> 0 on the way, S6; a flag
//     notes +-inf);
//   * 8 CONSUMER waves reduce the current item: each takes the chunk's next entry from a shared LDS counter (the plan
//     lists a chunk's entries longest first) and walks its segments with lane = timestep: v_readlane broadcasts
//     (cell, weight), one LDS read and one FMA per segment, eight reads in flight; results leave through a per-wave LDS
//     scratch as 16-byte stores (one store per 4 / 2 entries).  Only real (cell, region) pairs are multiplied, so
//     +-inf data needs no separate exact path, and the consumer waves never wait for each other.
// The image is SWIZZLED so that both sides are bank-conflict free: element (t, u) of row t lives at u ^ t -- the
// loaders' 16-byte pieces stay whole and a row's 64 pieces still fill its 1024 bytes (the XOR moves a piece inside the
// row and permutes the 4 / 2 elements inside it, the latter at compile time: the row is a loop constant), and the
// consumers' lanes, reading one cell u of 64 consecutive timesteps, hit 64 different banks (32 bank pairs per half-wave
// in fp64).  Rounds 1-2 read a padded image (row stride 260): 4-way conflicts, 51 % of the LDS cycles of the fp64 kernel
// (profiles/r02_pmc.csv); this one 0.03 % (profiles/r03_pmc.csv).
// ---------------------------------------------------------------------------------------------
constexpr int LV_LW = 8, LV_CW = 8, LV_THREADS = (LV_LW + LV_CW) * 64, LV_TB = 64;
constexpr int LV_ROWB = 1024;                   // bytes of an image row: 256 floats / 128 doubles = the cells of a chunk
template <typename T, int NPOW = 1> struct LvLds {
    static constexpr int E = 16 / (int)sizeof(T);                               // elements of a 16-byte piece
    static constexpr int scr_rows = NPOW > E ? NPOW : E;                        // result rows (entry, power) of one batch
    static constexpr size_t scr_wave = (size_t)scr_rows * 64 * sizeof(T);       // 1 KiB (2 KiB: fp64 with 3 or 4 powers)
    static constexpr size_t img = 0;                                            // [2][64][1024 B]
    static constexpr size_t scr = img + 2 * (size_t)LV_TB * LV_ROWB;            // [8 consumer waves][scr_wave] result scratch
    static constexpr size_t seg_w = scr + LV_CW * scr_wave;                     // [2][LC_SEGS] T
    static constexpr size_t seg_u = seg_w + 2 * sizeof(T) * LC_SEGS;            // [2][LC_SEGS] i32 (packed)
    static constexpr size_t ent_r = seg_u + 2 * sizeof(int32_t) * LC_SEGS;      // [2][LC_ENT] i32
    static constexpr size_t ent_d = ent_r + 2 * sizeof(int32_t) * LC_ENT;       // [2][LC_ENT] T
    static constexpr size_t ent_s = ent_d + 2 * sizeof(T) * LC_ENT;             // [2][LC_ENT + 2] u16
    static constexpr size_t hdr = (ent_s + 2 * sizeof(uint16_t) * (LC_ENT + 2) + 15) / 16 * 16;   // [2][16] i32
    static constexpr size_t total = hdr + 2 * 16 * sizeof(int32_t);
    static_assert(total <= 160 * 1024, "one workgroup must fit the CU's LDS");
};

// NPOW > 1 (fused tas_poly): the loaders park y = x + pv.xoff; the consumers raise every value they read to
// the powers pv.xpow .. pv.xpow + NPOW - 1 and keep NPOW sums per entry, so X is read once for NPOW powers; power i goes
// to out + i * out_pstride.  fp32 consumers take the segments two at a time through the packed instructions (v_pk_mul_f32 /
// v_pk_fma_f32: two partial sums per power, added at the end).  Only real (cell, region) pairs are multiplied, so a power
// that overflows gives +-inf as it does in the reference (transformations.py:188 then aggregations.py:78); what needs
// the general form is a NaN PRODUCT (inf times a zero weight), hence the loaders also flag |y| >= ylim, the largest
// value whose highest power is finite.
// EDD (fused Snyder degree days; NPOW = number of thresholds of the pass, <= 4): the chunks are the
// 128-cell ones in fp32 (eight 64-byte pieces per field and timestep), 64-cell ones in fp64 (four whole lines, a
// chunking of their own), an image row holds tasmin of the chunk in its first 512
// bytes and tasmax in the second (lanes 0-31 / 32-63 of the loaders fetch one field each, both shifted by pv.xoff), and
// the CONSUMERS evaluate snyder_edd1(tasmin, tasmax, thr[k]) for every (segment, timestep) and threshold -- so the
// arithmetic (about 15 vector instructions per value, wagg_common.h::snyder_edd1_finite) runs on eight waves beside the
// gather instead of on the waves that issue it (rounds 1-2: one stage per threshold on the loader waves of
// sparse_lc_kernel, 0.29 ms per threshold; now 0.07).
template <typename T, bool VEC, int NPOW = 1, bool EDD = false, bool GT = false>
__global__ __launch_bounds__(LV_THREADS) void sparse_lcv_kernel(PlanView<T> pv, const T *__restrict__ X, int64_t Ttot,
                                                                int64_t ldx, int64_t G, T *__restrict__ out, int64_t ldo,
                                                                int n_norm, long long n_items,
                                                                unsigned long long *__restrict__ stamps_arg, int knob_arg,
                                                                int64_t out_pstride = 0, T ylim = T(0)) {
#ifdef WAGG_DIAG
    const int knob = knob_arg;                       // ablation switches / phase stamps: diagnostic build only
    unsigned long long *const stamps = stamps_arg;
#else
    constexpr int knob = 0;
    constexpr unsigned long long *stamps = nullptr;
    (void)knob_arg; (void)stamps_arg;
#endif
    constexpr int E = 16 / (int)sizeof(T);           // elements per 16-byte piece: 4 / 2
    constexpr int LPQ = 4 / E;                       // lanes per 4-cell quad of the plan: 1 / 2
    // GT ((gridcell, time) data): a cell's 64 timesteps are PPC contiguous pieces; one load instruction fetches them for CPL cells
    constexpr int PPC = 64 / E, CPL = 64 / PPC;      // fp32: 16 pieces per cell, 4 cells (one quad) per load; fp64: 32, 2
    constexpr int CELLB = 64 * (int)sizeof(T);       // GT: bytes of a cell's row in the image
    typedef T vecE __attribute__((ext_vector_type(E)));
    typedef int int4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    using L = LvLds<T, NPOW>;
    char *img = smem_raw + L::img;
    T *sm_w = reinterpret_cast<T *>(smem_raw + L::seg_w);
    int32_t *sm_u = reinterpret_cast<int32_t *>(smem_raw + L::seg_u);
    int32_t *sm_er = reinterpret_cast<int32_t *>(smem_raw + L::ent_r);
    T *sm_ed = reinterpret_cast<T *>(smem_raw + L::ent_d);
    uint16_t *sm_es = reinterpret_cast<uint16_t *>(smem_raw + L::ent_s);
    int32_t *hdr = reinterpret_cast<int32_t *>(smem_raw + L::hdr);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < LV_LW;
    const bool out_vec = (ldo % E == 0) && (out_pstride % E == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    // XCD-contiguous ids (speed only)
    const unsigned NWu = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = NWu >> 3, r8 = NWu & 7u;
    const long long NW = NWu;
    const long long w0 = (long long)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot);
    if (w0 >= n_items) return;
    const int dg = (int)(NW % n_norm), dtb = (int)(NW / n_norm);
    const int nst = (int)((n_items - 1 - w0) / NW) + 1;
    struct Item { int g, tb; };
    auto advance = [&](Item a) {
        Item b{a.g + dg, a.tb + dtb};
        if (b.g >= n_norm) { b.g -= n_norm; ++b.tb; }
        return b;
    };
    unsigned long long ph[4] = {0, 0, 0, 0}, tprev = 0;
    auto stamp = [&](int i) {
        if (stamps) {
            unsigned long long tnow;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tnow)::"memory");
            if (i >= 0) ph[i] += tnow - tprev;
            tprev = tnow;
        }
    };

    if (loader) {
        // =============================== loader waves ===============================
        constexpr int TPW = LV_TB / LV_LW;                       // 8 rows per wave
        const int tw0 = wave * TPW;
        auto load_desc = [&](Item a, StreamDesc &d) {
            const int32_t *p = pv.chunk_desc + 8 * (int64_t)(pv.c0_normal + a.g);
            const int4v x = *reinterpret_cast<const int4v *>(p);
            const int4v y = *reinterpret_cast<const int4v *>(p + 4);
            // (wave-uniform: into scalar registers, the descriptors of three items are live at any time)
            d.u0 = __builtin_amdgcn_readfirstlane(x[0]); d.nq = __builtin_amdgcn_readfirstlane(x[1]);
            d.e0 = __builtin_amdgcn_readfirstlane(x[2]); d.ne = __builtin_amdgcn_readfirstlane(x[3]);
            d.sb = __builtin_amdgcn_readfirstlane(y[0]); d.ns = __builtin_amdgcn_readfirstlane(y[1]); d.split = 0;
        };
        // first cell of this lane's 16 bytes: quad lane / LPQ of the chunk (clamped), half lane % LPQ of it
        // (degree days: lanes 0-31 fetch the chunk's 32 quads of tasmin, lanes 32-63 the same quads of tasmax)
        // (GT: lane L holds the first cell of quad (first quad of this wave) + L % (quads per wave); issue() broadcasts it)
        auto load_cell = [&](const StreamDesc &d) {
            if (GT) {      // (degree days: loads 0-31 of an item are tasmin of the chunk's 32 quads, loads 32-63 tasmax of the same)
                const int q = ((EDD ? tw0 & 31 : tw0) * CPL) / 4 + lane % (TPW * CPL / 4);
                return pv.ucell[d.u0 + (q < d.nq ? q : d.nq - 1)];
            }
            const int l2 = EDD ? (lane & 31) : lane;              // (degree days: 32 lanes per field)
            const int q = l2 / LPQ;
            return pv.ucell[d.u0 + (q < d.nq ? q : d.nq - 1)] + (l2 % LPQ) * E;
        };
        const T *const Xl = EDD && lane >= 32 ? pv.X2 : X;        // this lane's field
        [[maybe_unused]] int dma_buf = 0;        // (diagnostic build, knob 0x4000: image buffer the NEXT item's rows are sent to by LDS-DMA)
        // `unref`: this lane's quad holds no referenced cell (bit 0 of its ucell entry, set by the plan builder for the whole-line
        // chunkings) -- what it loads is parked like everything else but does NOT count when the item decides between the
        // finite-data forms and the general ones below: that choice then depends on referenced data only, and is the same whether
        // the unreferenced quads of a line hold the field's values (device apply, whole lines from the host) or a dummy (the
        // quads-only rows of the host path).  (GT: one bit per load of the wave -- a load's cells belong to one quad.)
        struct Regs { vecE v[TPW]; int mu; T mw; int er, es; T ed; int unref; };
        static_assert(LC_SEGS <= LV_LW * 64, "one metadata element per loader thread");
        auto issue = [&](Regs &R, const StreamDesc &d, int cell0, int tb) {
            if constexpr (GT) {
                R.unref = 0;
#pragma unroll
                for (int i = 0; i < TPW; ++i) R.unref |= (__builtin_amdgcn_readlane(cell0, (i * CPL) / 4) & UCELL_UNREF) << i;
            } else
                R.unref = cell0 & UCELL_UNREF;
            cell0 &= ~UCELL_UNREF;
            // small metadata loads first, the rows last (vmcnt retires in order)
            {
                const int k = tid < d.ns ? tid : d.ns - 1;
                R.mu = pv.seg_u[d.sb + k];
                R.mw = pv.seg_w[d.sb + k];
                R.er = pv.ent_region[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
                R.ed = pv.ent_den[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
                R.es = pv.ent_seg_begin[d.e0 + (tid < d.ne ? tid : d.ne)];
            }
            const int64_t t0 = (int64_t)tb * LV_TB;
            const int nt = (int)((Ttot - t0) < LV_TB ? (Ttot - t0) : LV_TB);
            const int rbase = tw0 < nt - 1 ? tw0 : nt - 1;
            int cnt = nt - tw0;
            cnt = cnt < 1 ? 1 : (cnt > TPW ? TPW : cnt);
            if constexpr (GT) {
                // load i of this wave = cells CPL * (tw0 + i) .. + CPL - 1 of the chunk, 64 timesteps each: lane -> (cell of the
                // load, piece of its row).  Pieces behind the last timestep repeat the last one (their lanes are never stored);
                // cells behind G (a quad at the end of a grid that is not a whole number of quads) repeat cell G - 1
                const int piece = lane % PPC, jl = lane / PPC;
                const int np = (nt + E - 1) / E;
                const int pc = piece < np ? piece : np - 1;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    const int w = tw0 + i;
                    int64_t c = (int64_t)__builtin_amdgcn_readlane(cell0, (i * CPL) / 4) + ((w * CPL) % 4) + jl;
                    c = c < G ? c : G - 1;
                    const T *pg = (EDD && w >= 32 ? pv.X2 : X) + c * ldx + t0;
                    if (VEC) R.v[i] = *reinterpret_cast<const vecE *>(pg + E * pc);
                    else {
#pragma unroll
                        for (int e2 = 0; e2 < E; ++e2) R.v[i][e2] = pg[E * pc + e2 < nt ? E * pc + e2 : nt - 1];
                    }
                }
                return;
            }
#ifdef WAGG_DIAG
            if (VEC && !EDD && (knob & 0x4000)) {
                // TIMING ONLY (results are wrong): the rows go straight to the image by LDS-DMA (no register staging, no NaN
                // rule, no swizzle; the consumers may still be reading that buffer) -- does the CU sustain more bytes per clock
                // when the data does not return through the vector registers?
                const unsigned lbase = (unsigned)reinterpret_cast<uintptr_t>(img) + (unsigned)(dma_buf * LV_TB + tw0) * LV_ROWB;
                const T *rowp = X + (t0 + rbase) * ldx;
                const unsigned voff = (unsigned)cell0 * (unsigned)sizeof(T);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    int m0save;
                    asm volatile("s_mov_b32 %[sv], m0\n\ts_mov_b32 m0, %[l0]\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %[vo], %[src]\n\ts_mov_b32 m0, %[sv]"
                                 : [sv] "=&s"(m0save) : [l0] "s"(lbase + (unsigned)i * LV_ROWB), [vo] "v"(voff), [src] "s"(rowp) : "memory");
                    if (i + 1 < cnt) rowp += ldx;
                }
                return;
            }
#endif
            const T *p = Xl + (t0 + rbase) * ldx + cell0;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (VEC && (knob & 2)) R.v[i] = __builtin_nontemporal_load(reinterpret_cast<const vecE *>(p));   // (diagnostic build: nt loads)
                else if (VEC) R.v[i] = *reinterpret_cast<const vecE *>(p);
                else {
                    const int64_t lim = G - 1 - cell0;
#pragma unroll
                    for (int c = 0; c < E; ++c) R.v[i][c] = p[lim < c ? lim : c];
                }
                if (i + 1 < cnt) p += ldx;
            }
        };
        auto park = [&](Regs &R, const StreamDesc &d, int tb, int buf) {
            char *im = img + (size_t)buf * LV_TB * LV_ROWB;
#ifdef WAGG_DIAG
            if (VEC && !EDD && (knob & 0x4000)) {            // (timing only: the rows came by LDS-DMA; 13 younger loads may stay out)
                asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
                if (lane == 0) hdr[buf * 16 + 8 + wave] = 0;
                if (tid < d.ns) { sm_u[buf * LC_SEGS + tid] = R.mu; sm_w[buf * LC_SEGS + tid] = R.mw; }
                if (tid < d.ne) { sm_er[buf * LC_ENT + tid] = R.er; sm_ed[buf * LC_ENT + tid] = R.ed; }
                if (tid <= d.ne) sm_es[buf * (LC_ENT + 2) + tid] = (uint16_t)(R.es - d.sb);
                if (tid == 0) { hdr[buf * 16 + 0] = d.ne; hdr[buf * 16 + 1] = d.ns; hdr[buf * 16 + 2] = 0; hdr[buf * 16 + 3] = tb; }
                dma_buf = buf;          // the item issued next lands where this one was
                return;
            }
#endif
            bool odd = false;
            if (EDD) {                                            // both fields shifted (transformations.py:64-66)
#pragma unroll
                for (int i = 0; i < TPW; ++i) R.v[i] = R.v[i] + pv.xoff;
            } else if (NPOW > 1) {                                // the consumers raise y = x + xoff to the powers
#pragma unroll
                for (int i = 0; i < TPW; ++i) R.v[i] = R.v[i] + pv.xoff;
            } else if (pv.xpow > 0) {                             // (x + xoff)^xpow on the way in (transformations.py:188)
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int c = 0; c < E; ++c) R.v[i][c] = xform1<T>(R.v[i][c], pv.xoff, pv.xpow);
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                bool o = false;
#pragma unroll
                for (int c = 0; c < E; ++c) {
                    if (NPOW > 1 && !EDD) o |= !(__builtin_fabs(R.v[i][c]) < ylim);     // NaN, +-inf, or a power overflows
                    else if constexpr (sizeof(T) == 4) o |= __builtin_amdgcn_classf(R.v[i][c], 0x207);   // sNaN | qNaN | -inf | +inf
                    else o |= __builtin_amdgcn_class(R.v[i][c], 0x207);
                }
                if constexpr (GT) odd |= o && !((R.unref >> i) & 1);
                else odd |= o;
            }
            if constexpr (!GT) odd = odd && !R.unref;              // (see Regs::unref)
            bool inf_any = false;
            if (EDD) {
                // a NaN in either field must reach the formula (NaN tasmin -> NaN, skipped; NaN tasmax below the threshold -> 0:
                // the two nested xr.where of transformations.py:74-87), so nothing is replaced here: the item takes the general form
                inf_any = __builtin_amdgcn_readfirstlane(__ballot(odd) != 0ull);
            } else if (__builtin_amdgcn_readfirstlane(__ballot(odd) != 0ull)) {
                bool inf_seen = false;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    bool s_i = false;
#pragma unroll
                    for (int c = 0; c < E; ++c) {
                        const T x = R.v[i][c];
                        if (NPOW > 1 && !EDD) s_i |= __builtin_fabs(x) >= ylim;   // the consumers then take the general form
                        else s_i |= __builtin_isinf(x);
                        R.v[i][c] = (x == x) ? x : T(0);                           // NaN data counts 0 (S6)
                    }
                    if constexpr (GT) inf_seen |= s_i && !((R.unref >> i) & 1);
                    else inf_seen |= s_i;
                }
                if constexpr (!GT) inf_seen = inf_seen && !R.unref;               // (unreferenced quads do not count: Regs::unref)
                inf_any = __builtin_amdgcn_readfirstlane(__ballot(inf_seen) != 0ull);
            }
            if (lane == 0) hdr[buf * 16 + 8 + wave] = inf_any ? 1 : 0;            // every wave, every item: no reset needed
            if constexpr (GT) {
                // the image is cell-major, [cell][64 timesteps]: load i is 1 KiB of it as it comes; the consumers' lanes read
                // consecutive words of a cell's row (no conflicts, no swizzle)
#pragma unroll
                for (int i = 0; i < TPW; ++i) *reinterpret_cast<vecE *>(im + (size_t)(tw0 + i) * LV_ROWB + 16 * lane) = R.v[i];
            } else
            // row t = tw0 + i: piece `lane` goes to piece lane ^ (t / E), its elements permuted by t % E = i % E
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int m = tw0 + i;
                vecE v;
#pragma unroll
                for (int c = 0; c < E; ++c) v[c ^ (i % E)] = R.v[i][c];
                *reinterpret_cast<vecE *>(im + (size_t)(tw0 + i) * LV_ROWB + 16 * (lane ^ (m / E))) = v;
            }
            if (tid < d.ns) { sm_u[buf * LC_SEGS + tid] = R.mu; sm_w[buf * LC_SEGS + tid] = R.mw; }
            if (tid < d.ne) { sm_er[buf * LC_ENT + tid] = R.er; sm_ed[buf * LC_ENT + tid] = R.ed; }
            if (tid <= d.ne) sm_es[buf * (LC_ENT + 2) + tid] = (uint16_t)(R.es - d.sb);
            if (tid == 0) { hdr[buf * 16 + 0] = d.ne; hdr[buf * 16 + 1] = d.ns; hdr[buf * 16 + 2] = 0; hdr[buf * 16 + 3] = tb; }
        };
        // descriptors/cells run ahead: d[j] / cell[j] / it[j] describe item (parked so far) + 1 + j
        Item itq[3];
        StreamDesc dq[3];
        int cellq[2];
        itq[0] = Item{(int)(w0 % n_norm), (int)(w0 / n_norm)};
        itq[1] = nst > 1 ? advance(itq[0]) : itq[0];
        itq[2] = nst > 2 ? advance(itq[1]) : itq[1];
        load_desc(itq[0], dq[0]); load_desc(itq[1], dq[1]); load_desc(itq[2], dq[2]);
        cellq[0] = load_cell(dq[0]);
        cellq[1] = load_cell(dq[1]);
        Regs RA, RB;
        issue(RA, dq[0], cellq[0], itq[0].tb);
        dma_buf = 1;
        StreamDesc dPark = dq[0];
        int tbPark = itq[0].tb;
        struct Ahead { Item nx; StreamDesc dn; int cn; };
        auto ahead_load = [&](Ahead &a) {
            a.nx = advance(itq[2]);
            if (a.nx.tb * (long long)n_norm + a.nx.g >= n_items) a.nx = itq[2];
            load_desc(a.nx, a.dn);
            a.cn = load_cell(dq[2]);
        };
        auto ahead_commit = [&](const Ahead &a) {
            itq[0] = itq[1]; itq[1] = itq[2]; itq[2] = a.nx;
            dq[0] = dq[1]; dq[1] = dq[2]; dq[2] = a.dn;
            cellq[0] = cellq[1]; cellq[1] = a.cn;
        };
        { Ahead a; ahead_load(a); ahead_commit(a); }              // queue now describes items 1, 2, 3
        int sbuf = 0;
        auto lstage = [&](Regs &Rcur, Regs &Rnext, int st) {
            const StreamDesc dn = dq[0];
            const int tbn = itq[0].tb;
            const bool more = st + 1 < nst;
            Ahead a;
            stamp(-1);
            if (more) { ahead_load(a); issue(Rnext, dn, cellq[0], tbn); }
            stamp(0);                                             // loader ph0: issue (blocked at VMEM)
            park(Rcur, dPark, tbPark, sbuf);
            stamp(2);                                             // ph2: wait for item st + park
            if (more) { dPark = dn; tbPark = tbn; ahead_commit(a); }
            stamp(1);
            lds_only_barrier();                                   // stage st is in buffer sbuf
            stamp(3);                                             // ph3: waiting for the consumers
            sbuf ^= 1;
        };
        for (int st = 0; st < nst; st += 2) {
            lstage(RA, RB, st);
            if (st + 1 < nst) lstage(RB, RA, st + 1);
        }
        lds_only_barrier();                                       // consumers finish the last item
        if (stamps && tid == 0) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 8 + i] = ph[i];
    } else {
        // ==================== consumer waves: lane = timestep, entries from the item's shared counter ====================
        const int cw = wave - LV_LW;
        constexpr int EPB = E / NPOW > 0 ? E / NPOW : 1;          // entries per result batch
        constexpr int ROWS = EPB * NPOW;                          // its rows (entry, power): E per 16-byte-per-lane store
        constexpr int LPE = 64 / E;                               // lanes per row of that store
        T *scr = reinterpret_cast<T *>(smem_raw + L::scr + cw * L::scr_wave);
        // (LDS addresses as plain integers: the low half of a generic pointer into LDS is its LDS address; the image starts
        // at a multiple of its row size -- at 0 in fact, the kernel has no static LDS -- so the XOR stays inside the row)
        typedef const T __attribute__((address_space(3))) *lds_cptr;
        const unsigned img0 = (unsigned)reinterpret_cast<uintptr_t>(img);
        if (img0 & (GT ? 0xffffu : (unsigned)(LV_ROWB - 1))) __builtin_trap();
        // this lane's image row, with the swizzle folded in: element u sits at rowoff ^ (u * sizeof(T))
        // (GT: cell u's row starts at u * CELLB, this lane's timestep sits lane * sizeof(T) into it: the same XOR, as an add)
        const unsigned rowoff = img0 + (GT ? 0u : (unsigned)lane * LV_ROWB) + (unsigned)lane * (unsigned)sizeof(T);
        typedef T pair2 __attribute__((ext_vector_type(2)));
        constexpr bool PK = NPOW > 1 && sizeof(T) == 4 && !EDD;   // fp32 powers: two segments per packed instruction
        for (int st = 0; st < nst; ++st) {
            stamp(-1);
            lds_only_barrier();                                   // stage st has been parked
            stamp(3);                                             // consumer ph3: waiting for the loaders
            const int buf = st & 1;
            const unsigned rb = rowoff + (unsigned)buf * (LV_TB * LV_ROWB);         // (the XOR below only touches bits 0..9)
            const int ne = __builtin_amdgcn_readfirstlane(hdr[buf * 16 + 0]);
            const int64_t t0 = (int64_t)__builtin_amdgcn_readfirstlane(hdr[buf * 16 + 3]) * LV_TB;
            const int nt = (int)((Ttot - t0) < LV_TB ? (Ttot - t0) : LV_TB);
            const int fl = lane < LV_LW ? hdr[buf * 16 + 8 + lane] : 0;
            const bool odd = __builtin_amdgcn_readfirstlane(__ballot(fl != 0) != 0ull);   // +-inf (or an overflowing power) in the item
            if (knob & 1) continue;                               // (diagnostic build: consumers idle)
            // ODD (decided once per item, so the segment loop is branch-free): the general form in which a NaN product
            // counts 0 (S6: +-inf data times a zero weight)
            auto walk = [&](auto odd_tag) {
                constexpr bool ODD = decltype(odd_tag)::value;
                for (bool more = true; more;) {
                    // entries come one at a time from the item's shared counter (the waves' shares differ in length: a
                    // static deal left the first wave idle a third of the degree-day stage); EPB of them share a store
                    int es[EPB];
#pragma unroll
                    for (int kb = 0; kb < EPB; ++kb) es[kb] = ne;
#pragma unroll 1
                    for (int kb = 0; kb < EPB; ++kb) {
                        int e = 0;
                        if (lane == 0) e = __hip_atomic_fetch_add(&hdr[buf * 16 + 2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        e = __builtin_amdgcn_readfirstlane(e);
                        if (e >= ne) { more = false; break; }
#pragma unroll
                        for (int q = 0; q < EPB; ++q) if (q == kb) es[q] = e;       // (static indices: es stays in scalar registers)
                        const int s0 = __builtin_amdgcn_readfirstlane((int)sm_es[buf * (LC_ENT + 2) + e]);
                        const int s1 = __builtin_amdgcn_readfirstlane((int)sm_es[buf * (LC_ENT + 2) + e + 1]);
                        T acc[NPOW];
                        pair2 acc2[NPOW];
#pragma unroll
                        for (int pw = 0; pw < NPOW; ++pw) { acc[pw] = T(0); acc2[pw] = pair2{T(0), T(0)}; }
                        for (int base = (knob & 256) ? s1 : s0; base < s1; base += 64) {
                            // lane j holds segment base + j (padding lanes: cell 0, weight 0: they add exactly 0 to finite data)
                            const int n = s1 - base < 64 ? s1 - base : 64;
                            const int k = base + (lane < n ? lane : 0);
                            int ul = (sm_u[buf * LC_SEGS + k] & 0xff) * (GT ? CELLB : (int)sizeof(T));   // byte offset of the cell in a row (GT: of its row)
                            T wl = sm_w[buf * LC_SEGS + k];
                            if (lane >= n) { ul = 0; wl = T(0); }
                            for (int j0 = 0; j0 < n; j0 += 8) {
                                T xv[8], wv[8], xh[EDD ? 8 : 1];
#pragma unroll
                                for (int j = 0; j < 8; ++j) {     // 8 independent LDS reads in flight
                                    const unsigned u = (unsigned)__builtin_amdgcn_readlane(ul, j0 + j);
                                    if constexpr (sizeof(T) == 4) {
                                        wv[j] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wl), j0 + j));
                                    } else {
                                        const long long wb = __builtin_bit_cast(long long, wl);
                                        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(wb & 0xffffffffll), j0 + j);
                                        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(wb >> 32), j0 + j);
                                        wv[j] = __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
                                    }
                                    xv[j] = *(lds_cptr)(uintptr_t)(rb ^ u);
                                    // (degree days: tasmax of the same cell sits 512 bytes further on -- the XOR never reaches
                                    // bit 9: cells < 128, timesteps < 64 -- so the two reads fuse into one ds_read2_b32)
                                    // (GT: tasmax is the second half of the image, 128 cell rows further on)
                                    if constexpr (EDD) xh[j] = *(lds_cptr)(uintptr_t)((rb ^ u) + (GT ? 32768 : 512));
                                }
                                __builtin_amdgcn_sched_barrier(0);
                                if constexpr (PK && !ODD) {
#pragma unroll
                                    for (int j = 0; j < 8; j += 2) {
                                        const pair2 y = pair2{xv[j], xv[j + 1]}, w2 = pair2{wv[j], wv[j + 1]};
                                        pair2 yp = y;
                                        for (int i = 1; i < pv.xpow; ++i) yp *= y;                       // first power of this pass
#pragma unroll
                                        for (int pw = 0; pw < NPOW; ++pw) {
                                            if (pw) yp *= y;
                                            acc2[pw] = __builtin_elementwise_fma(yp, w2, acc2[pw]);
                                        }
                                    }
                                } else if constexpr (EDD) {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) {
#pragma unroll
                                        for (int pw = 0; pw < NPOW; ++pw) {                            // one plane per threshold
                                            // transformations.py:64-87 (finite fields: the expression that needs no selection)
                                            const T ev = ODD ? snyder_edd1<T>(xv[j], xh[j], pv.edd_thr[pw]) : snyder_edd1_finite(xv[j], xh[j], pv.edd_thr[pw]);
                                            if constexpr (ODD) {
                                                const T p = ev * wv[j];
                                                acc[pw] += (p == p) ? p : T(0);
                                            } else if constexpr (sizeof(T) == 4) {
                                                acc[pw] = __builtin_fmaf(ev, wv[j], acc[pw]);
                                            } else {
                                                acc[pw] = __builtin_fma(ev, wv[j], acc[pw]);
                                            }
                                        }
                                    }
                                } else {
#pragma unroll
                                    for (int j = 0; j < 8; ++j) {
                                        T yp = xv[j];
                                        if (NPOW > 1) for (int i = 1; i < pv.xpow; ++i) yp *= xv[j];     // first power of this pass
#pragma unroll
                                        for (int pw = 0; pw < NPOW; ++pw) {
                                            if (pw) yp *= xv[j];
                                            if constexpr (ODD) {
                                                const T p = yp * wv[j];
                                                acc[pw] += (p == p) ? p : T(0);
                                            } else if constexpr (sizeof(T) == 4) {
                                                acc[pw] = __builtin_fmaf(yp, wv[j], acc[pw]);           // aggregations.py:78
                                            } else {
                                                acc[pw] = __builtin_fma(yp, wv[j], acc[pw]);
                                            }
                                        }
                                    }
                                }
                            }
                        }
                        const T dn = sm_ed[buf * LC_ENT + e];
#pragma unroll
                        for (int pw = 0; pw < NPOW; ++pw) {
                            if constexpr (PK && !ODD) acc[pw] = acc2[pw][0] + acc2[pw][1];
                            scr[(kb * NPOW + pw) * 64 + lane] = acc[pw] / dn;                    // :77-80
                        }
                    }
                    // (LDS operations of one wave execute in order: the reads below see the writes above)
                    for (int pass = 0; pass < (ROWS + E - 1) / E; ++pass) {
                        const int row = pass * E + lane / LPE, piece = lane % LPE;
                        const int kq = row / NPOW, pw = row % NPOW;
                        int e = ne;
#pragma unroll
                        for (int q = 0; q < EPB; ++q) if (q == kq) e = es[q];
                        const int tl = E * piece;
                        if (row < ROWS && e < ne && tl < nt && !(knob & 128)) {
                            T *op = out + (int64_t)pw * out_pstride + (int64_t)sm_er[buf * LC_ENT + e] * ldo + t0 + tl;
                            const vecE qv = *reinterpret_cast<const vecE *>(&scr[row * 64 + tl]);
                            if (out_vec && tl + E - 1 < nt) {
                                *reinterpret_cast<vecE *>(op) = qv;
                            } else {
#pragma unroll
                                for (int rg = 0; rg < E; ++rg) if (tl + rg < nt) op[rg] = qv[rg];
                            }
                        }
                    }
                }
            };
            if (odd) walk(std::true_type{}); else walk(std::false_type{});
            stamp(1);                                             // ph1: the item's entries
        }
        lds_only_barrier();                                       // matches the loaders' final barrier
        if (stamps && tid == LV_LW * 64) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 8 + 4 + i] = ph[i];
    }
}

// (R x T) -> (T x R) through a padded 64x64 LDS tile: both sides coalesced.  The gather kernel
// stores region-major (lane = timestep: 256 contiguous bytes per region) because a (T x R) store
// from it would scatter single dwords over R-strided lines (7x write amplification measured).
template <typename T>
__global__ __launch_bounds__(256) void transpose_rt_to_tr_kernel(const T *__restrict__ in, int64_t ldi,
                                                                int64_t R, int64_t Ttot,
                                                                T *__restrict__ out, int64_t ldo) {
    // the staging buffer has 16-byte aligned rows of ldi = 64 k elements (zero-padded reads are harmless:
    // only t < Ttot is stored), so the region-major side is read with 16-byte loads
    constexpr int V = 16 / sizeof(T);                            // elements per load: 4 floats / 2 doubles
    typedef T vecv __attribute__((ext_vector_type(V)));
    __shared__ T tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    constexpr int TQ = 64 / V;                                   // vector columns per tile row
    const int tq = threadIdx.x % TQ, ry = threadIdx.x / TQ;
#pragma unroll
    for (int i = ry; i < 64; i += 256 / TQ) {
        const int64_t r = r0 + i;
        if (r < R) {
            const vecv v = *reinterpret_cast<const vecv *>(in + r * ldi + t0 + V * tq);
#pragma unroll
            for (int c = 0; c < V; ++c) tile[V * tq + c][i] = v[c];
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 64; i += 4) {
        const int64_t t = t0 + ty + i, r = r0 + tx;
        if (r < R && t < Ttot) out[t * ldo + r] = tile[ty + i][tx];
    }
}

// Whole-line plans: out[t][r] = (sum of region r's partial rows)[t] / den[r] (aggregations.py:77-80), the same padded
// 64 x 64 LDS tile transpose for (time, region) results; regions without any row give 0 / den (S7).  Region r's rows are
// rows part_begin[r] .. part_begin[r + 1] - 1 of the buffer (the builder numbers them region-major), added in that order
// (bitwise reproducible), four loads in flight.
constexpr int CB_THREADS = 1024;
template <typename T, bool TR>
__global__ __launch_bounds__(CB_THREADS) void combine_parts_kernel(const T *__restrict__ P, int64_t ldp, const int32_t *__restrict__ part_begin,
                                                                   const T *__restrict__ den, int64_t R, int64_t Ttot,
                                                                   T *__restrict__ out, int64_t ldo, int64_t p_pstride, int64_t out_pstride) {
    P += (int64_t)blockIdx.z * p_pstride;              // one plane (power / threshold) per blockIdx.z
    out += (int64_t)blockIdx.z * out_pstride;
    constexpr int V = 16 / sizeof(T);
    typedef T vecv __attribute__((ext_vector_type(V)));
    __shared__ T tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    constexpr int TQ = 64 / V;
    const int tq = threadIdx.x % TQ, ry = threadIdx.x / TQ;
#pragma unroll
    for (int i = ry; i < 64; i += CB_THREADS / TQ) {
        const int64_t r = r0 + i;
        if (r < R) {
            vecv s;
#pragma unroll
            for (int c = 0; c < V; ++c) s[c] = T(0);
            const int32_t kb = part_begin[r], ke = part_begin[r + 1];
            const T *p = P + t0 + V * tq;
            for (int32_t k = kb; k < ke; k += 4) {
                vecv v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const vecv *>(p + (int64_t)(k + q < ke ? k + q : ke - 1) * ldp);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (k + q < ke) s += v[q];
            }
            const T d = den[r];
            if (TR) {
#pragma unroll
                for (int c = 0; c < V; ++c) tile[V * tq + c][i] = s[c] / d;
            } else {
#pragma unroll
                for (int c = 0; c < V; ++c) { const int64_t t = t0 + V * tq + c; if (t < Ttot) out[r * ldo + t] = s[c] / d; }
            }
        }
    }
    if (!TR) return;
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 64; i += CB_THREADS / 64) {
        const int64_t t = t0 + ty + i, r = r0 + tx;
        if (r < R && t < Ttot) out[t * ldo + r] = tile[ty + i][tx];
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

#ifdef WAGG_DIAG
#define WAGG_DIAG_GATHER_VARIANTS(L, O, V)                                                       \
            else switch (diag_env("WAGG_SPARSE_DBG")) {                                          \
                case 1: kern = sparse_gather_kernel<T, TB, L, O, V, 1>; break;                   \
                case 2: kern = sparse_gather_kernel<T, TB, L, O, V, 2>; break;                   \
                case 4: kern = sparse_gather_kernel<T, TB, L, O, V, 4>; break;                   \
                case 6: kern = sparse_gather_kernel<T, TB, L, O, V, 6>; break;                   \
                case 7: kern = sparse_gather_kernel<T, TB, L, O, V, 7>; break;                   \
                default: break;                                                                  \
            }
#else
#define WAGG_DIAG_GATHER_VARIANTS(L, O, V)
#endif

// a consumer-wave barrier of an earlier sparse_lc_kernel launch on this plan timed out: its
// results are incomplete (the dead waves stopped storing)
static int check_timeout(const wagg_plan *plan) {
#ifndef WAGG_DIAG
    (void)plan;            // no kernel of the production library can time out: none of them waits on another wave's progress
    return WAGG_OK;
#endif
    if (plan->timeout_host && *(volatile int *)plan->timeout_host != 0) {
        set_error("sparse_lc_kernel: consumer-wave barrier timed out in an earlier apply on this plan; "
                  "its results are incomplete");
        return WAGG_EHIP;
    }
    return WAGG_OK;
}

// ---- which kernel runs what (one table instead of an if-ladder; docs/HISTORY.md (d) carries the same table) -----------------
// Single-chunk ("normal") groups of a plan, by (element type, data layout, transform):
//   transform            (time, gridcell) data                          (gridcell, time) data
//   none / one power     sparse_lcv_kernel<T,VEC,1>      [L32 | L64]    sparse_lcv_kernel<T,VEC,1,false,GT>    [R | L64]
//   powers p..p+n-1 <= 4 sparse_lcv_kernel<T,VEC,n>      [L32 | L64]    sparse_lcv_kernel<T,VEC,n,false,GT>    [R | L64]
//   degree days, k <= 4  sparse_lcv_kernel<T,VEC,k,EDD>  [L64 | L64e]   sparse_lcv_kernel<T,VEC,k,EDD,GT>      [L64 | L64e]
// chunking in brackets, fp32 | fp64: R = region-shaped (<= 64 quads), L32 / L64 = eight whole 128-byte lines of 32 / 16
// cells, L64e = four lines of 16 cells (both fp64 fields of a chunk in one image row).  Where the chunking named does not
// exist (WAGG_PLAN_NO_LINES, a grid without a row length, a scattered table) or a WAGG_PLAN_NO_* flag says so: fp32 plain /
// powers stay on sparse_lcv_kernel with R; everything else takes sparse_stream_kernel (no transform, one power per pass) or
// the chunk-walking sparse_gather_kernel (degree days, (gridcell, time) data), which also serves every giant group.
// More than four powers / thresholds: several passes.
template <typename T> struct LcvEntry {
    void (*kern)(PlanView<T>, const T *, int64_t, int64_t, int64_t, T *, int64_t, int, long long, unsigned long long *, int, int64_t, T);
    size_t lds;
};
template <typename T, bool VEC, bool EDD, bool GT, int N> static constexpr LcvEntry<T> lcv_entry() {
    return {sparse_lcv_kernel<T, VEC, N, EDD, GT>, LvLds<T, EDD ? 4 : N>::total};       // (degree days: one LDS size for 1..4 planes)
}
template <typename T> static const LcvEntry<T> &lcv_pick(bool vec, bool edd, bool gt, int planes) {
#define WAGG_LCV_ROW(V, E, G) {lcv_entry<T, V, E, G, 1>(), lcv_entry<T, V, E, G, 2>(), lcv_entry<T, V, E, G, 3>(), lcv_entry<T, V, E, G, 4>()}
    static const LcvEntry<T> table[2][2][2][4] = {       // [16-byte aligned rows][degree days][(gridcell, time)][planes - 1]
        {{WAGG_LCV_ROW(false, false, false), WAGG_LCV_ROW(false, false, true)}, {WAGG_LCV_ROW(false, true, false), WAGG_LCV_ROW(false, true, true)}},
        {{WAGG_LCV_ROW(true, false, false), WAGG_LCV_ROW(true, false, true)}, {WAGG_LCV_ROW(true, true, false), WAGG_LCV_ROW(true, true, true)}}};
#undef WAGG_LCV_ROW
    return table[vec ? 1 : 0][edd ? 1 : 0][gt ? 1 : 0][planes - 1];
}

template <typename T, int TB>
static int launch_sparse(const wagg_plan *plan, const T *X, int64_t Ttot, int64_t ldx, int layout,
                         T *out, int64_t ldo, int out_layout, hipStream_t stream, T xoff = T(0), int xpow = 0,
                         int nfuse = 1, int64_t pstride = 0, const T *X2 = nullptr, const double *thr = nullptr,
                         int n_thr = 0, int compact = COMPACT_NONE) {
    // compact: X holds COMPACT rows of the lines-only host path -- COMPACT_LINES: ldx >= Gc cells, the distinct quads of the
    // whole-line chunking side by side, read through ucell_c; COMPACT_QUADS: ldx >= Gq cells, only the quads a segment reads,
    // through ucell_q -- (time, gridcell) data on that chunking
    // nfuse > 1: powers xpow .. xpow + nfuse - 1 of (x + xoff) in one pass over X (fused tas_poly); the i-th goes to
    // out + i * pstride.  sparse_lcv_kernel fuses up to four planes (powers or degree-day thresholds) in both element types and
    // both layouts on the chunkings of the table above; everything it does not serve (giant groups, plans whose flags or
    // table shape rule its chunkings out) runs once per power with the transform applied on load.
    const bool lcv_off = (plan->flags & (WAGG_PLAN_NO_LC | WAGG_PLAN_NO_STREAM | WAGG_PLAN_LC_MFMA)) != 0;
    // degree days in sparse_lcv_kernel: fp32 (time, gridcell) fields on the 128-cell chunking (the fp64 one: 64-byte pieces)
    const bool edd_lcv = xpow == XF_EDD && (sizeof(T) == 4 ? plan->has_lines64 : plan->has_lines64e) && !lcv_off && n_thr >= 1 && n_thr <= 4;
    const auto &d_edd = sizeof(T) == 4 ? plan->dl64 : plan->dl64e;     // chunks whose two fields fill one 64 KiB image
    const bool use_lines = (edd_lcv && layout == WAGG_LAYOUT_TG) || ((sizeof(T) == 4 ? plan->has_lines : plan->has_lines64) && layout == WAGG_LAYOUT_TG &&
                                       xpow != XF_EDD && (nfuse == 1 || (nfuse <= 4 && !lcv_off)));
    // (gridcell, time) data in sparse_lcv_kernel: every cell of a chunk is fetched by itself (64 timesteps = two or four whole
    // lines), so fp32 takes the region-shaped chunks (fewest cells); fp64 the 128-cell whole-line chunking (a region-shaped
    // chunk holds 256)
    const bool gt_lcv = layout == WAGG_LAYOUT_GT && !lcv_off && (xpow != XF_EDD || edd_lcv) && nfuse <= 4 &&
                        (sizeof(T) == 4 || plan->has_lines64);
    const auto &d = edd_lcv ? d_edd
                            : (gt_lcv ? (sizeof(T) == 4 ? plan->d : plan->dl64) : (use_lines ? (sizeof(T) == 4 ? plan->dl : plan->dl64) : plan->d));
    if (int rc = check_timeout(plan)) return rc;
    const int64_t Gcomp = compact == COMPACT_QUADS ? d.Gq : d.Gc;
    WAGG_REQUIRE(!compact || (use_lines && !lcv_off && nfuse <= 4 && (xpow != XF_EDD || edd_lcv) && Gcomp > 0 && ldx >= Gcomp),
                 "compact rows need the whole-line chunking of this plan and data type");
    const int64_t Gk = compact ? Gcomp : (int64_t)plan->info.G;        // cells of a row as the kernels see it
    if (nfuse > 1) {
        // fused: fp32 in either loader/consumer kernel; fp64 in sparse_lcv_kernel on its whole-line chunking
        const bool lc_ok = ((layout == WAGG_LAYOUT_TG && (sizeof(T) == 4 || (use_lines && !lcv_off))) || gt_lcv) &&
                           !(plan->flags & (WAGG_PLAN_NO_STREAM | WAGG_PLAN_NO_LC)) && (int)d.n_groups - d.g0_normal > 0 && Ttot > 0;
        if (!lc_ok || nfuse > 4) {
            for (int i = 0; i < nfuse; ++i) {
                const int rc = launch_sparse<T, TB>(plan, X, Ttot, ldx, layout, out + (int64_t)i * pstride, ldo,
                                                    out_layout, stream, xoff, xpow + i, 1, 0, nullptr, nullptr, 0, compact);
                if (rc != WAGG_OK) return rc;
            }
            return WAGG_OK;
        }
    }
    PlanView<T> pv;
    pv.xoff = xoff; pv.xpow = xpow; pv.X2 = X2;
    pv.n_thr = n_thr > 0 ? n_thr : 1;
    for (int k = 0; k < 4; ++k) pv.edd_thr[k] = (T)(thr && k < n_thr ? thr[k] : 0.0);
    // planes of output: powers of the fused tas_poly, or thresholds of one degree-day pass
    const int nplanes = xpow == XF_EDD ? pv.n_thr : nfuse;
    pv.grp_chunk_begin = d.grp_chunk_begin.p; pv.grp_giant = d.grp_giant.p;
    pv.chunk_u_begin = d.chunk_u_begin.p; pv.chunk_e_begin = d.chunk_e_begin.p;
    pv.ucell = compact == COMPACT_QUADS ? d.ucell_q.p : (compact ? d.ucell_c.p : d.ucell.p); pv.ent_region = d.ent_region.p; pv.ent_seg_begin = d.ent_seg_begin.p;
    pv.seg_u = d.seg_u.p;
    if constexpr (sizeof(T) == 4) { pv.seg_w = d.seg_w32.p; pv.den = d.den32.p; pv.ent_den = d.ent_den32.p; }
    else { pv.seg_w = d.seg_w64.p; pv.den = d.den64.p; pv.ent_den = d.ent_den64.p; }
    // whole-line plan: an entry's "region" is its partial row and nothing is divided before the rows are combined
    // (ent_den holds one 1.0 per entry = per partial row)
    if (d.n_part > 0) pv.den = pv.ent_den;
    pv.n_groups = (int)d.n_groups;
    if (Ttot == 0) return WAGG_OK;
    const int64_t n_tb = (Ttot + TB - 1) / TB;
    // (T x R) results: gather kernel writes region-major into a stream-ordered workspace, then
    // one transpose; (R x T) results go straight to the caller's buffer
    T *ws = nullptr;
    int64_t ldws = 0;
    T *kout = out;
    int64_t kldo = ldo;
    // whole-line plans: the kernels write PARTIAL rows (one per (chunk, region) entry) into the workspace whatever the
    // output layout; combine_parts_kernel sums a region's rows, divides and (for (T x R)) transposes
    const bool lines = d.n_part > 0;
    const bool via_ws = (out_layout == WAGG_OUT_TR || lines) && d.n_groups > 0;
    const int64_t ws_rows = lines ? d.n_part : (int64_t)plan->info.R;
    int64_t kpstride = pstride;
    if (via_ws) {
        ldws = (Ttot + 63) / 64 * 64;
        ws = static_cast<T *>(plan->staging(stream, sizeof(T) * (size_t)(ldws * ws_rows) * (size_t)nplanes));
        if (!ws) { set_error("staging buffer of %.1f MB: allocation failed", (double)(sizeof(T) * ldws * ws_rows * nplanes) * 1e-6); return WAGG_ENOMEM; }
        kout = ws;
        kldo = ldws;
        kpstride = ldws * ws_rows;
    }
    pv.chunk_desc = d.chunk_desc.p; pv.g0_normal = d.g0_normal; pv.c0_normal = d.c0_normal;
    // aligned fast path: 16-byte aligned rows
    const bool vec = ((reinterpret_cast<uintptr_t>(X) & 15) == 0) && ((ldx * sizeof(T)) % 16 == 0) &&
                     (xpow != XF_EDD || (reinterpret_cast<uintptr_t>(X2) & 15) == 0);
    // degree days (two fields): fp32 (time, gridcell) data in the loader/consumer kernel, everything else in the
    // chunk-walking kernel
    const bool edd = xpow == XF_EDD;
    const bool stream_path = (layout == WAGG_LAYOUT_TG || gt_lcv) && !(plan->flags & WAGG_PLAN_NO_STREAM) && (!edd || sizeof(T) == 4 || edd_lcv);
    const int n_norm = (int)d.n_groups - d.g0_normal;
    bool lc_done = false;
    // plain aggregation: loaders + vector-ALU consumers (sparse_lcv_kernel).  fp32 on either chunking, fp64 on its
    // whole-line chunking only (a region-shaped chunk of 64 quads is 2 KiB of a fp64 row: twice the image row)
    if (!lc_done && stream_path && n_norm > 0 && !(plan->flags & (WAGG_PLAN_NO_LC | WAGG_PLAN_LC_MFMA)) && (!edd || edd_lcv) && nfuse <= 4 &&
        (sizeof(T) == 4 || lines)) {
        const int ncu = plan->ncu;
        const long long n_items = (long long)n_norm * ((Ttot + LV_TB - 1) / LV_TB);
        long long nw = n_items < ncu ? n_items : ncu;
        if (const int v = diag_env("WAGG_LCV_NW")) { if (v >= 1 && v < nw) nw = v; }      // (diagnostic build: fewer workgroups = CUs)
        const LcvEntry<T> &ke = lcv_pick<T>(vec, edd_lcv, gt_lcv, edd_lcv ? pv.n_thr : nfuse);
        const auto kern = ke.kern;
        const size_t lds_bytes = ke.lds;
        // |y| below this can be raised to the highest power of the pass inside T
        const T ylim = nfuse > 1 ? (T)std::pow((double)std::numeric_limits<T>::max() / 1024.0, 1.0 / (double)(xpow + nfuse - 1)) : T(0);
        WAGG_HIP(allow_dynamic_lds((const void *)kern, lds_bytes));
        unsigned long long *lc_stamps = nullptr;
        if (diag_set("WAGG_SPARSE_STAMP")) WAGG_HIP(hipMalloc((void **)&lc_stamps, sizeof(unsigned long long) * 8 * (size_t)nw));
        launch_timed(true, kern, dim3((unsigned)nw), dim3(LV_THREADS), lds_bytes, stream, pv, X, Ttot, ldx,
                     Gk, kout, kldo, n_norm, n_items, lc_stamps, diag_env("WAGG_LC_KNOB"), kpstride, ylim);
        WAGG_HIP(hipGetLastError());
#ifdef WAGG_DIAG
        if (lc_stamps) { if (int rc = report_lc_stamps(lc_stamps, nw, n_items, stream)) return rc; }
#endif
        pv.n_groups = d.g0_normal;
        lc_done = true;
    }
#ifdef WAGG_DIAG
    if constexpr (sizeof(T) == 4) {
        // (diagnostic build, wagg_sparse_diag.hip) plans created with WAGG_PLAN_LC_MFMA: the MFMA-consumer kernel
        if (stream_path && n_norm > 0 && !lc_done && (plan->flags & WAGG_PLAN_LC_MFMA) && !(plan->flags & WAGG_PLAN_NO_LC)) {
            if (int rc = launch_lc_mfma(plan, pv, X, Ttot, ldx, kout, kldo, n_norm, vec, edd, xpow, nfuse, kpstride, stream)) return rc;
            pv.n_groups = d.g0_normal;
            lc_done = true;
        }
    }
#endif
    if (stream_path && n_norm > 0 && !lc_done && !edd) {
        // persistent pipelined kernel over the single-chunk groups; two workgroups per CU
        const int ncu = plan->ncu;
        const long long n_items = (long long)n_norm * n_tb;
        const long long nw = n_items < 2LL * ncu ? n_items : 2LL * ncu;
        const size_t shmem = SparseLds<T, TB>::total;
        auto kern = vec ? sparse_stream_kernel<T, TB, true> : sparse_stream_kernel<T, TB, false>;
        unsigned long long *stamps = nullptr;
        const bool do_stamp = diag_set("WAGG_SPARSE_STAMP");
#ifdef WAGG_DIAG
        if (do_stamp) {
            kern = sparse_stream_kernel<T, TB, true, true>;
            WAGG_HIP(hipMalloc((void **)&stamps, sizeof(unsigned long long) * 10 * (size_t)nw));
        }
#endif
        WAGG_HIP(allow_dynamic_lds((const void *)kern, shmem));
        launch_timed(true, kern, dim3((unsigned)nw), dim3(STHREADS), shmem, stream, pv, X, Ttot, ldx, (int64_t)plan->info.G,
                     kout, kldo, n_norm, n_items, stamps, diag_env("WAGG_SPARSE_STAMP"));
        WAGG_HIP(hipGetLastError());
        if (do_stamp) {           // diagnostic: print mean cycles per stage and phase
            std::vector<unsigned long long> h(10 * (size_t)nw);
            WAGG_HIP(hipStreamSynchronize(stream));
            WAGG_HIP(staged_d2h(h.data(), stamps, sizeof(unsigned long long) * h.size()));
            WAGG_HIP(hipFree(stamps));
            double sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = 0; i < h.size(); ++i) sum[i % 10] += (double)h[i];
            const double stages = (double)n_items;
            fprintf(stderr, "[wagg stamp] items=%lld nw=%lld cycles/stage: wait_rows=%.0f park=%.0f [flush=%.0f desc+cell=%.0f meta=%.0f rows=%.0f] bar1=%.0f compute=%.0f (first entry %.0f) rot+bar2=%.0f\n",
                    n_items, nw, sum[0] / stages, sum[1] / stages, sum[6] / stages, sum[7] / stages, sum[8] / stages, sum[2] / stages,
                    sum[3] / stages, (sum[4] + sum[9]) / stages, sum[9] / stages, sum[5] / stages);
        }
        pv.n_groups = d.g0_normal;          // what is left for the chunk-walking kernel: giant groups
        lc_done = true;                     // (a dominant kernel has been launched and timed)
    }
    const bool main_done = lc_done;
    WAGG_REQUIRE(!compact || (lc_done && pv.n_groups == 0), "compact rows: the whole-line kernel did not take the whole plan");
    pv.thr_pstride = kpstride;
    for (int pz = 0; pz < nfuse; ++pz) {      // per power: the groups the kernels above left over (giant ones)
    if (nfuse > 1) pv.xpow = xpow + pz;
    if (pv.n_groups > 0) {
        const int64_t nblk = (int64_t)pv.n_groups * n_tb;
        WAGG_REQUIRE(nblk < (int64_t)0x7fffffff, "grid too large: %lld", (long long)nblk);
        const size_t shmem = SparseLds<T, TB>::total;
        dim3 grid((unsigned)nblk), block(UC);
#define WAGG_LAUNCH(L, O, V)                                                                    \
        do {                                                                                     \
            auto kern = sparse_gather_kernel<T, TB, L, O, V>;                                    \
            if (xpow == XF_EDD && pv.n_thr > 1) kern = sparse_gather_kernel<T, TB, L, O, V, 0, 4>; \
            WAGG_DIAG_GATHER_VARIANTS(L, O, V)                                                   \
            WAGG_HIP(allow_dynamic_lds((const void *)kern, shmem));                              \
            launch_timed(!main_done, kern, grid, block, shmem, stream, pv, X, Ttot, ldx,         \
                         (int64_t)plan->info.G, kout + (int64_t)pz * kpstride, kldo);            \
        } while (0)
        if (layout == WAGG_LAYOUT_TG) { if (vec) WAGG_LAUNCH(WAGG_LAYOUT_TG, WAGG_OUT_RT, true); else WAGG_LAUNCH(WAGG_LAYOUT_TG, WAGG_OUT_RT, false); }
        else WAGG_LAUNCH(WAGG_LAYOUT_GT, WAGG_OUT_RT, false);
#undef WAGG_LAUNCH
        WAGG_HIP(hipGetLastError());
    }
    }
    if (via_ws && lines) {                    // all planes in one launch (regions without rows come out as 0 / den there)
        dim3 tg((unsigned)((plan->info.R + 63) / 64), (unsigned)((Ttot + 63) / 64), (unsigned)nplanes);
        const T *den;
        if constexpr (sizeof(T) == 4) den = d.den32.p; else den = d.den64.p;
        if (out_layout == WAGG_OUT_TR)
            hipLaunchKernelGGL((combine_parts_kernel<T, true>), tg, dim3(CB_THREADS), 0, stream, (const T *)ws, ldws,
                               (const int32_t *)d.part_begin.p, den, (int64_t)plan->info.R, Ttot, out, ldo, kpstride, pstride);
        else
            hipLaunchKernelGGL((combine_parts_kernel<T, false>), tg, dim3(CB_THREADS), 0, stream, (const T *)ws, ldws,
                               (const int32_t *)d.part_begin.p, den, (int64_t)plan->info.R, Ttot, out, ldo, kpstride, pstride);
        WAGG_HIP(hipGetLastError());
        return WAGG_OK;
    }
    for (int pz = 0; pz < nplanes; ++pz) {    // per plane: transpose, regions without any kept row
    if (via_ws) {
        dim3 tg((unsigned)((plan->info.R + 63) / 64), (unsigned)((Ttot + 63) / 64));
        hipLaunchKernelGGL((transpose_rt_to_tr_kernel<T>), tg, dim3(256), 0, stream, ws + (int64_t)pz * kpstride, ldws,
                           (int64_t)plan->info.R, Ttot, out + (int64_t)pz * pstride, ldo);
        WAGG_HIP(hipGetLastError());
    }
    if (d.n_empty > 0) {
        const int64_t n = d.n_empty * Ttot;
        const T *den;
        if constexpr (sizeof(T) == 4) den = d.den32.p; else den = d.den64.p;
        hipLaunchKernelGGL((fill_empty_kernel<T>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, d.empty_regions.p, (int)d.n_empty, den, Ttot, out + (int64_t)pz * pstride, ldo,
                           out_layout);
        WAGG_HIP(hipGetLastError());
    }
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
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE((flags & ~(WAGG_PLAN_NO_LC | WAGG_PLAN_NO_STREAM | WAGG_PLAN_NO_LINES | WAGG_PLAN_LC_MFMA | WAGG_PLAN_SERIAL_BUILD)) == 0, "unknown plan flags 0x%x", flags);
#ifndef WAGG_DIAG
    if (flags & WAGG_PLAN_LC_MFMA) {
        set_error("WAGG_PLAN_LC_MFMA: the MFMA-consumer kernel lives in the diagnostic build (libwagg_diag.so) only");
        return WAGG_EUNSUPPORTED;
    }
#endif
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
        // by (region, cell), rows of one pair in table order (S5: they are added in that order below).  std::stable_sort on
        // the 16-byte rows was half of a c2-real plan build (42 of ~80 ms for 4e5 rows); stable counting passes -- by cell,
        // then by region -- do it in a third of that, and a table that already comes by cell (rows = grid order, the usual
        // export) or by (region, cell) skips the passes it does not need
        {
            bool by_cell = true, by_region_cell = true;
            for (size_t i = 1; i < segs.size(); ++i) {
                by_cell = by_cell && segs[i - 1].cell <= segs[i].cell;
                by_region_cell = by_region_cell && (segs[i - 1].region < segs[i].region ||
                                                    (segs[i - 1].region == segs[i].region && segs[i - 1].cell <= segs[i].cell));
            }
            auto counting_pass = [&](int64_t n_keys, auto key_of) {
                std::vector<int64_t> first((size_t)n_keys + 1, 0);
                for (const Seg &sg : segs) ++first[(size_t)key_of(sg) + 1];
                for (int64_t k = 0; k < n_keys; ++k) first[(size_t)k + 1] += first[(size_t)k];
                std::vector<Seg> sorted(segs.size());
                for (const Seg &sg : segs) sorted[(size_t)first[(size_t)key_of(sg)]++] = sg;
                segs.swap(sorted);
            };
            if (by_region_cell) {
                // nothing to do
            } else if (G > 8 * (int64_t)segs.size() + (1 << 22)) {          // a counter per cell would dwarf the table
                std::stable_sort(segs.begin(), segs.end(), [](const Seg &a, const Seg &b) {
                    return a.region != b.region ? a.region < b.region : a.cell < b.cell;
                });
            } else {
                if (!by_cell) counting_pass(G, [](const Seg &sg) { return (int64_t)sg.cell; });
                counting_pass(R, [](const Seg &sg) { return (int64_t)sg.region; });
            }
        }
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
        wagg_plan *plan = new wagg_plan();
        std::unique_ptr<wagg_plan> plan_guard(plan);
        plan->den_host = den;
        plan->info.nseg_in = nseg; plan->info.nnz = nnz;
        plan->info.G = G; plan->info.R = R;
        plan->flags = flags;
        std::vector<float> den32(den.size());
        for (size_t i = 0; i < den.size(); ++i) den32[i] = (float)den[i];
        hipError_t he = hipGetDevice(&plan->device);
        if (he == hipSuccess) he = hipDeviceGetAttribute(&plan->ncu, hipDeviceAttributeMultiprocessorCount, plan->device);
#ifdef WAGG_DIAG
        if (he == hipSuccess) he = hipHostMalloc((void **)&plan->timeout_host, sizeof(int), hipHostMallocMapped);
        if (he == hipSuccess) {
            *plan->timeout_host = 0;
            he = hipHostGetDevicePointer((void **)&plan->timeout_dev, plan->timeout_host, 0);
        }
#endif
        // one chunking of the table -> device arrays `d`; returns false when the whole-line chunking does not apply
        // (line_cells: 0 = region-shaped chunks; 32 / 16 = whole lines of that many cells, 128 bytes of a fp32 / fp64 row)
        // (kind: 0 region-shaped, 1 whole lines fp32, 2 whole lines fp64, 3 the 64-cell chunks of fp64 degree days)
        // (`he`: the caller's error word -- the whole-line chunkings are built on threads of their own, each with its own)
        auto build = [&](int line_cells, int lines_per_chunk, int kind, SparsePlanDev &d, hipError_t &he) -> bool {
        const bool for_f64 = kind >= 2;
        const bool want_lines = line_cells > 0;
        std::vector<int32_t> grp_chunk_begin{0}, grp_giant, chunk_u_begin{0}, chunk_e_begin{0};
        std::vector<int32_t> ucell, ent_region, ent_seg_begin{0}, seg_u;   // ucell = first cell of each quad
        std::vector<double> seg_w;
        std::vector<int32_t> part_begin;                                   // whole-line plan only
        int64_t n_part_rows = 0;
        std::vector<int32_t> empty;
        int64_t n_giant = 0;
        int band_rows = 8;
        seg_u.reserve((size_t)nnz); seg_w.reserve((size_t)nnz);

        // ---- whole-line plan (round 3) --------------------------------------------------------------------------------
        // A chunk = up to 8 whole 32-cell LINES (128 bytes of a fp32 row, 256 of a fp64 row) of ONE column strip: the
        // land lines of the strip in row order, eight at a time (ocean rows in between are skipped).  Every line belongs
        // to exactly one chunk and is fetched whole: 8 line requests per timestep and chunk where the region-shaped
        // chunks below need ~23 (their quads straddle lines, and border lines are fetched by both neighbours), at
        // ~1.4x the bytes (the ocean cells of a coastal line come along).  The price: a region cut by a strip or by a
        // group of eight lines becomes several (chunk, region) ENTRIES, each a partial sum with a row of its own in the
        // partial buffer; combine_parts_kernel adds them up (1.6 rows per region on the 0.25-degree impact regions).
        // Taken when the grid's row length is known (whole rows of whole quads) and the table is compact enough.
        const int LINE = want_lines ? line_cells : 32;                     // cells per line
        const int LPC = lines_per_chunk;                                   // lines per chunk
        bool lines_plan = want_lines;
        if (lines_plan) {
            struct LSeg { int64_t line; int32_t region, col; double w; };
            // segments by (line, region, column), line = strip * n_rows + row: `segs` is sorted by (region, cell), so a STABLE
            // counting sort on the line alone gives that order (cells of one line and region ascend with their column)
            const int64_t n_rows_ = G / row_len, n_strips_ = (row_len + LINE - 1) / LINE;
            std::vector<LSeg> ls((size_t)nnz);
            {
                std::vector<int64_t> first((size_t)(n_rows_ * n_strips_) + 1, 0);
                std::vector<int64_t> line_of((size_t)nnz);
                for (int64_t i = 0; i < nnz; ++i) {
                    const int64_t row = segs[(size_t)i].cell / row_len, col = segs[(size_t)i].cell % row_len;
                    line_of[(size_t)i] = (col / LINE) * n_rows_ + row;
                    ++first[(size_t)line_of[(size_t)i] + 1];
                }
                for (size_t l = 1; l < first.size(); ++l) first[l] += first[l - 1];
                for (int64_t i = 0; i < nnz; ++i) {
                    const int64_t col = segs[(size_t)i].cell % row_len;
                    ls[(size_t)first[(size_t)line_of[(size_t)i]]++] = {line_of[(size_t)i], segs[(size_t)i].region, (int32_t)(col % LINE), segs[(size_t)i].w};
                }
            }
            const int64_t n_rows = G / row_len;
            std::vector<std::vector<int32_t>> parts((size_t)R);
            std::vector<int32_t> region_mark((size_t)R, -1);
            struct CSeg { int32_t region, ulocal; double w; };
            // pass 1: form the chunks strip by strip
            struct LChunk { int64_t key; std::vector<int32_t> quads; std::vector<CSeg> cs; };
            std::vector<LChunk> chunks;
            const int64_t n_strips = (row_len + LINE - 1) / LINE;
            size_t i = 0;
            while (i < ls.size() && lines_plan) {
                // one chunk: lines of this strip while it holds < LPC lines, <= SEG_MAX segments, <= RG_MAX regions
                const int64_t strip = ls[i].line / n_rows;
                const int32_t chunk_id = (int32_t)chunks.size();
                int n_lines = 0, n_regions = 0;
                LChunk ch;
                ch.key = ((ls[i].line % n_rows) / LPC) * n_strips + strip;       // band of LPC grid rows, then strip
                while (i < ls.size() && ls[i].line / n_rows == strip && n_lines < LPC) {
                    size_t j = i;                                          // the segments of the next line
                    int fresh = 0;
                    while (j < ls.size() && ls[j].line == ls[i].line) {
                        if (region_mark[(size_t)ls[j].region] != chunk_id) { region_mark[(size_t)ls[j].region] = chunk_id; ++fresh; }
                        ++j;
                    }
                    if (n_lines > 0 && ((int64_t)ch.cs.size() + (int64_t)(j - i) > SEG_MAX || n_regions + fresh > RG_MAX)) {
                        for (size_t k = i; k < j; ++k) region_mark[(size_t)ls[k].region] = -1;   // (marks of the line not taken)
                        for (const CSeg &c : ch.cs) region_mark[(size_t)c.region] = chunk_id;
                        break;
                    }
                    if ((int64_t)(j - i) > SEG_MAX || fresh > RG_MAX) { lines_plan = false; break; }   // one line alone is too much
                    n_regions += fresh;
                    const int64_t row = ls[i].line % n_rows;
                    for (int q = 0; q < LINE / 4; ++q) {                   // the line's quads; those behind the row end repeat its last
                        int64_t c0 = strip * LINE + 4 * q;
                        if (c0 + 4 > row_len) c0 = row_len - 4;
                        ch.quads.push_back((int32_t)(row * row_len + c0));
                    }
                    for (size_t k = i; k < j; ++k) ch.cs.push_back({ls[k].region, n_lines * LINE + ls[k].col, ls[k].w});
                    ++n_lines;
                    i = j;
                }
                if (!lines_plan) break;
                std::sort(ch.cs.begin(), ch.cs.end(), [](const CSeg &a, const CSeg &b) {
                    return a.region != b.region ? a.region < b.region : a.ulocal < b.ulocal; });
                chunks.push_back(std::move(ch));
            }
            // pass 2: flatten, in the order formed (strip-major: down one column strip, then the next).  Measured against
            // band-major order (all strips of eight grid rows, then the next eight rows: neighbouring lines of the same
            // DRAM pages in flight together) on c2-real / c3-real: 0.236 / 0.331 ms strip-major, 0.237 / 0.352 ms
            // band-major (gpurun r3r) -- the diagnostic build keeps the switch
            if (diag_env("WAGG_LINES_ORDER") == 2)
                std::stable_sort(chunks.begin(), chunks.end(), [](const LChunk &a, const LChunk &b) { return a.key < b.key; });
            int64_t n_part = 0;
            for (const LChunk &ch : chunks) {
                if (!lines_plan) break;
                ucell.insert(ucell.end(), ch.quads.begin(), ch.quads.end());
                chunk_u_begin.push_back((int32_t)ucell.size());
                const std::vector<CSeg> &cs = ch.cs;
                // the chunk's entries, longest first: the consumer waves of sparse_lcv_kernel take them from a shared
                // counter, so the long ones start early and the short ones fill the end (longest-processing-time order)
                std::vector<std::pair<size_t, size_t>> runs;
                for (size_t k = 0; k < cs.size();) {
                    size_t m2 = k;
                    while (m2 < cs.size() && cs[m2].region == cs[k].region) ++m2;
                    runs.push_back({k, m2});
                    k = m2;
                }
                std::stable_sort(runs.begin(), runs.end(), [](const std::pair<size_t, size_t> &a, const std::pair<size_t, size_t> &b) {
                    return a.second - a.first > b.second - b.first; });
                for (const auto &run : runs) {
                    for (size_t m2 = run.first; m2 < run.second; ++m2) { seg_u.push_back(cs[m2].ulocal); seg_w.push_back(cs[m2].w); }
                    parts[(size_t)cs[run.first].region].push_back((int32_t)n_part);
                    ent_region.push_back((int32_t)n_part++);              // the entry's row in the partial buffer
                    ent_seg_begin.push_back((int32_t)seg_u.size());
                }
                chunk_e_begin.push_back((int32_t)ent_region.size());
                grp_giant.push_back(0);
                grp_chunk_begin.push_back((int32_t)(chunk_u_begin.size() - 1));
            }
            // scattered regions (a c5-like table) would need more partial rows than the table has regions many times
            // over: such tables keep the region-shaped chunks (and usually take a dense-family form anyway)
            if (lines_plan && n_part > 4 * (int64_t)R + 1024) lines_plan = false;
            if (lines_plan) {
                // partial rows are numbered region-major: region r owns rows part_begin[r] .. part_begin[r + 1] - 1 of the
                // partial buffer, in chunk order, so the combine kernel streams the buffer front to back
                part_begin.assign((size_t)R + 1, 0);
                std::vector<int32_t> new_id((size_t)n_part, 0);
                for (int32_t r = 0; r < R; ++r) {
                    part_begin[(size_t)r + 1] = part_begin[(size_t)r] + (int32_t)parts[(size_t)r].size();
                    for (size_t j = 0; j < parts[(size_t)r].size(); ++j)
                        new_id[(size_t)parts[(size_t)r][j]] = part_begin[(size_t)r] + (int32_t)j;
                    if (parts[(size_t)r].empty()) empty.push_back(r);
                }
                for (int32_t &e : ent_region) e = new_id[(size_t)e];
                n_part_rows = n_part;
            } else {                                                        // start over with the region-shaped chunks
                grp_chunk_begin.assign(1, 0); grp_giant.clear(); chunk_u_begin.assign(1, 0); chunk_e_begin.assign(1, 0);
                ucell.clear(); ent_region.clear(); ent_seg_begin.assign(1, 0); seg_u.clear(); seg_w.clear();
            }
        }
        if (!lines_plan) {
        // regions are ordered along latitude bands of `band_rows` grid rows (column-major inside a
        // band): a chunk then covers ~band_rows rows x a run of columns
        if (const int v = diag_env("WAGG_BAND_ROWS")) { if (v >= 1 && v <= 1024) band_rows = v; }
        std::vector<int32_t> order;
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
            key_band[(size_t)r] = (int64_t)(srow / (double)n) / band_rows;
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
        }   // region-shaped chunks

        // bit 15 of a segment's local cell index marks the LAST segment of its region entry
        // (local indices are < 256); entries without segments cannot exist (every entry has >= 1)
        for (size_t e = 0; e + 1 < ent_seg_begin.size(); ++e)
            if (ent_seg_begin[e + 1] > ent_seg_begin[e]) seg_u[(size_t)ent_seg_begin[e + 1] - 1] |= SEG_LAST;
        // bits 16..23: index of the segment's entry inside its chunk (read by the MFMA kernel)
        for (size_t c = 0; c + 1 < chunk_e_begin.size(); ++c)
            for (int32_t e = chunk_e_begin[c]; e < chunk_e_begin[c + 1]; ++e)
                for (int32_t q = ent_seg_begin[(size_t)e]; q < ent_seg_begin[(size_t)e + 1]; ++q)
                    seg_u[(size_t)q] |= (e - chunk_e_begin[c]) << 16;
        // per-chunk descriptors for the persistent kernel; giant groups come first.  dd[6] packs the
        // entry split points of the four waves (contiguous entry ranges with ~equal segment counts)
        std::vector<int32_t> chunk_desc((size_t)(chunk_u_begin.size() - 1) * 8, 0);
        for (size_t c = 0; c + 1 < chunk_u_begin.size(); ++c) {
            int32_t *dd = &chunk_desc[c * 8];
            dd[0] = chunk_u_begin[c]; dd[1] = chunk_u_begin[c + 1] - chunk_u_begin[c];
            dd[2] = chunk_e_begin[c]; dd[3] = chunk_e_begin[c + 1] - chunk_e_begin[c];
            dd[4] = ent_seg_begin[(size_t)chunk_e_begin[c]];
            dd[5] = ent_seg_begin[(size_t)chunk_e_begin[c + 1]] - dd[4];
            int32_t split[SWAVE + 1];
            for (int i = 0; i <= SWAVE; ++i) split[i] = i == SWAVE ? dd[3] : 0;
            int w = 1;
            for (int32_t e = 0; e < dd[3] && w < SWAVE; ++e) {
                const int64_t done = ent_seg_begin[(size_t)(dd[2] + e + 1)] - dd[4];
                while (w < SWAVE && done * SWAVE >= (int64_t)dd[5] * w) split[w++] = e + 1;
            }
            while (w < SWAVE) split[w++] = dd[3];
            uint64_t packed = 0;
            for (int i = 1; i < SWAVE; ++i) packed |= (uint64_t)(split[i] & 0xff) << (8 * (i - 1));
            dd[6] = (int32_t)(uint32_t)(packed & 0xffffffffu);
            dd[7] = (int32_t)(uint32_t)(packed >> 32);
        }
        int g0_normal = 0;
        while (g0_normal < (int)grp_giant.size() && grp_giant[(size_t)g0_normal]) ++g0_normal;
        const int c0_normal = grp_chunk_begin[(size_t)g0_normal];

        if (lines_plan != want_lines) return false;
        if (!want_lines) {
            plan->info.n_groups = (int64_t)grp_giant.size();
            plan->info.n_chunks = (int64_t)chunk_u_begin.size() - 1;
            plan->info.n_ucells = (int64_t)ucell.size() * 4;    // cells fetched per timestep (whole quads)
            plan->info.n_giant = n_giant;
            plan->info.n_empty = (int64_t)empty.size();
        } else if (kind == 3) {
            // (info.lines: set by the caller once the builder threads have joined)
        } else if (!for_f64) {
            plan->info.n_partial_rows = n_part_rows;
            plan->info.lines_chunks = (int64_t)chunk_u_begin.size() - 1;
            plan->info.lines_ucells = (int64_t)ucell.size() * 4;
        } else {
            plan->info.n_partial_rows64 = n_part_rows;
            plan->info.lines64_chunks = (int64_t)chunk_u_begin.size() - 1;
            plan->info.lines64_ucells = (int64_t)ucell.size() * 4;
        }
        int64_t l128s = 0, s64s = 0;
        for (size_t c = 0; c + 1 < chunk_u_begin.size(); ++c) {      // locality statistics of the gather
            int32_t l128 = -1, s64 = -1;
            std::vector<int32_t> qs(ucell.begin() + chunk_u_begin[c], ucell.begin() + chunk_u_begin[c + 1]);
            std::sort(qs.begin(), qs.end());
            for (int32_t q : qs) {
                if ((q >> 5) != l128) { l128 = q >> 5; ++l128s; }
                if ((q >> 4) != s64) { s64 = q >> 4; ++s64s; }
            }
        }
        if (!want_lines) { plan->info.n_lines128 = l128s; plan->info.n_sectors64 = s64s; }
        else if (kind == 1) plan->info.lines_lines128 = l128s;
#ifdef WAGG_DIAG
        if (diag_set("WAGG_PLAN_STATS"))      // plan statistics without a device (host experiments on the chunk builder)
            fprintf(stderr, "[wagg plan] line_cells=%d band_rows=%d chunks=%lld groups=%lld giant=%lld ucells=%lld lines128=%lld sectors64=%lld nnz=%lld partial_rows=%lld\n",
                    lines_plan ? line_cells : 0, band_rows, (long long)chunk_u_begin.size() - 1, (long long)grp_giant.size(), (long long)n_giant,
                    (long long)ucell.size() * 4, (long long)l128s, (long long)s64s, (long long)nnz, (long long)n_part_rows);
#endif
        // Whole-line chunkings: which quads of a chunk does a segment really read?  A chunk fetches every quad of its lines; the
        // others are parked in the image and never read -- except quad 0, whose first cell the padding lanes of the consumers
        // read with weight 0 (so it counts as referenced: its data must be sane for 0 * x = 0).  Unreferenced quads get bit 0 of
        // their ucell entry set (UCELL_UNREF; quads are 4-cell aligned, the low bits are free): the kernels mask it off the address,
        // and sparse_lcv_kernel leaves what such a quad loads out of its finite / general decision (Regs::unref).
        // (a segment's local cell index is the low byte of seg_u -- bit 15 and bits 16..23 carry flags by now, see above)
        std::vector<char> used;
        bool map_ok = lines_plan && kind >= 1;
        if (map_ok) {
            used.assign(ucell.size(), 0);
            for (size_t c = 0; c + 1 < chunk_u_begin.size() && map_ok; ++c) {
                const int32_t nq = chunk_u_begin[c + 1] - chunk_u_begin[c];
                if (nq > 0) used[(size_t)chunk_u_begin[c]] = 1;
                for (int32_t e = chunk_e_begin[c]; e < chunk_e_begin[c + 1] && map_ok; ++e)
                    for (int32_t sg = ent_seg_begin[(size_t)e]; sg < ent_seg_begin[(size_t)e + 1]; ++sg) {
                        const int32_t q = (seg_u[(size_t)sg] & 0xff) >> 2;
                        if (q >= nq) { map_ok = false; break; }                  // (cannot happen; then: no flags, no quad map)
                        used[(size_t)chunk_u_begin[c] + (size_t)q] = 1;
                    }
            }
            if (map_ok)
                for (size_t i = 0; i < ucell.size(); ++i) if (!used[i]) ucell[i] |= UCELL_UNREF;
        }
        std::vector<float> seg_w32(seg_w.size());
        for (size_t i = 0; i < seg_w.size(); ++i) seg_w32[i] = (float)seg_w[i];
        auto up = [&](auto &buf, const auto &h) { if (he == hipSuccess) he = buf.upload(h); };
        up(d.grp_chunk_begin, grp_chunk_begin); up(d.grp_giant, grp_giant);
        up(d.chunk_u_begin, chunk_u_begin); up(d.chunk_e_begin, chunk_e_begin);
        up(d.ucell, ucell); up(d.ent_region, ent_region); up(d.ent_seg_begin, ent_seg_begin);
        up(d.seg_u, seg_u); up(d.seg_w32, seg_w32); up(d.seg_w64, seg_w);
        up(d.den32, den32); up(d.den64, den); up(d.empty_regions, empty); up(d.chunk_desc, chunk_desc);
        {
            std::vector<double> ed64(ent_region.size() + 1, 1.0);   // +1: the clamped ent_seg_begin twin
            std::vector<float> ed32(ent_region.size() + 1, 1.0f);
            for (size_t e = 0; e < ent_region.size() && !lines_plan; ++e) {       // (partial rows are divided when combined)
                ed64[e] = den[(size_t)ent_region[e]];
                ed32[e] = den32[(size_t)ent_region[e]];
            }
            up(d.ent_den64, ed64); up(d.ent_den32, ed32);
        }
        if (lines_plan) { up(d.part_begin, part_begin); d.n_part = n_part_rows; }
        if (lines_plan && kind >= 1) {
            // the compact row of the lines-only host path: the distinct quads of this chunking in grid order, side by side
            // (ucell entries carry UCELL_UNREF in bit 0: compared without it, handed on with it)
            std::vector<int32_t> uq(ucell.size());
            for (size_t i = 0; i < ucell.size(); ++i) uq[i] = ucell[i] & ~UCELL_UNREF;
            std::sort(uq.begin(), uq.end());
            uq.erase(std::unique(uq.begin(), uq.end()), uq.end());
            std::vector<int32_t> ucell_c(ucell.size());
            for (size_t i = 0; i < ucell.size(); ++i)
                ucell_c[i] = (int32_t)(4 * (std::lower_bound(uq.begin(), uq.end(), ucell[i] & ~UCELL_UNREF) - uq.begin())) | (ucell[i] & UCELL_UNREF);
            d.run_src.clear(); d.run_len.clear();
            for (size_t i = 0; i < uq.size(); ++i) {
                if (i > 0 && uq[i] == uq[i - 1] + 4) d.run_len.back() += 4;
                else { d.run_src.push_back((int64_t)uq[i]); d.run_len.push_back(4); }
            }
            up(d.ucell_c, ucell_c);
            d.Gc = 4 * (int64_t)uq.size();
            // ... and the quads that count as referenced only (wagg_sparse_int.h: ucell_q); the others point at position 0 of
            // the row (any valid address: what they load is neither read nor counted)
            if (map_ok) {
                std::vector<int32_t> uqq;
                for (size_t i = 0; i < ucell.size(); ++i) if (used[i]) uqq.push_back(ucell[i]);
                std::sort(uqq.begin(), uqq.end());
                uqq.erase(std::unique(uqq.begin(), uqq.end()), uqq.end());
                std::vector<int32_t> ucell_q(ucell.size(), UCELL_UNREF);
                for (size_t i = 0; i < ucell.size(); ++i)
                    if (used[i]) ucell_q[i] = (int32_t)(4 * (std::lower_bound(uqq.begin(), uqq.end(), ucell[i]) - uqq.begin()));
                d.run_src_q.clear(); d.run_len_q.clear();
                for (size_t i = 0; i < uqq.size(); ++i) {
                    if (i > 0 && uqq[i] == uqq[i - 1] + 4) d.run_len_q.back() += 4;
                    else { d.run_src_q.push_back((int64_t)uqq[i]); d.run_len_q.push_back(4); }
                }
                up(d.ucell_q, ucell_q);
                d.Gq = 4 * (int64_t)uqq.size();
            }
        }
        d.g0_normal = g0_normal; d.c0_normal = c0_normal;
        d.n_groups = (int64_t)grp_giant.size(); d.n_empty = (int64_t)empty.size();
        return true;
        };   // build

        // The chunkings are independent of each other: the region-shaped one is built here, the whole-line ones (for the
        // kernel that is bound by line requests; their extra bytes -- ocean cells of coastal lines -- cost the other kernels
        // more than the aligned lines save them: c3, fp64: 0.47 -> 0.52 ms) meanwhile on threads of their own.  c2-real:
        // 87 ms one after the other (39 + 3 x ~16), ~45 ms this way.
        const bool want_line_plans = !(flags & (WAGG_PLAN_NO_LINES | WAGG_PLAN_NO_LC | WAGG_PLAN_NO_STREAM)) && row_len < G && G % row_len == 0 &&
                                     row_len % 4 == 0 && nnz > 0 && he == hipSuccess;
        // (diagnostic build: WAGG_LINE_MULT = 2, 4, 8 makes the lines that many times longer and the chunk that many
        // times flatter -- 8 lines x 128 bytes by default, 1 x 1 KiB at the other end; 16: half lines (64 bytes), 16 per chunk)
        int mult = diag_env("WAGG_LINE_MULT");
        const bool half_lines = mult == 16;
        if (mult != 2 && mult != 4 && mult != 8) mult = 1;
        struct LineJob { int line_cells, lines_per_chunk, kind; SparsePlanDev *d; bool ok = false; hipError_t he = hipSuccess; bool oom = false, failed = false; };
        LineJob jobs[3] = {{half_lines ? 16 : 32 * mult, half_lines ? 16 : 8 / mult, 1, &plan->dl},
                           {half_lines ? 8 : 16 * mult, half_lines ? 16 : 8 / mult, 2, &plan->dl64},
                           {16, 4, 3, &plan->dl64e}};
        // A job never lets an exception out of its thread (that would be std::terminate): bad_alloc and anything else it
        // meets are noted in the job.  A thread that cannot be started (std::system_error: EAGAIN under a thread or process
        // limit -- every plan build asks for three, concurrent drop-in calls multiply that) means the job runs here, on the
        // calling thread, like all of them do under WAGG_PLAN_SERIAL_BUILD.  The joiner runs before anything leaves this
        // scope, so no joinable thread is ever destroyed and `build` / `jobs` outlive every worker.
        auto run_job = [&build](LineJob &j, int dev, bool set_device) noexcept {
            try {
                if (set_device) j.he = hipSetDevice(dev);            // a new thread starts on device 0
                if (j.he == hipSuccess) j.ok = build(j.line_cells, j.lines_per_chunk, j.kind, *j.d, j.he);
            } catch (const std::bad_alloc &) { j.oom = true; }
            catch (...) { j.failed = true; }
        };
        struct Joiner {
            std::vector<std::thread> t;
            ~Joiner() { for (std::thread &w : t) if (w.joinable()) w.join(); }
        } workers;
        bool oom = false, failed = false;
        if (want_line_plans) {
            try { workers.t.reserve(3); } catch (...) { flags |= WAGG_PLAN_SERIAL_BUILD; }
            for (LineJob &j : jobs) {
                bool started = false;
                if (!(flags & WAGG_PLAN_SERIAL_BUILD)) {
                    try {
                        workers.t.emplace_back(run_job, std::ref(j), plan->device, true);
                        started = true;
                    } catch (...) {}                                  // no thread to be had: the job runs below
                }
                if (!started) run_job(j, plan->device, false);
            }
        }
        try { build(0, 0, 0, plan->d, he); } catch (const std::bad_alloc &) { oom = true; } catch (...) { failed = true; }
        for (std::thread &w : workers.t) w.join();
        for (const LineJob &j : jobs) { if (he == hipSuccess) he = j.he; oom |= j.oom; failed |= j.failed; }
        if (failed) {
            set_error("plan build failed: unexpected exception in a chunking builder");
            return WAGG_EINTERNAL;                                    // (plan_guard deletes the plan)
        }
        if (oom) throw std::bad_alloc();
        if (want_line_plans && he == hipSuccess) {
            plan->has_lines = jobs[0].ok; plan->has_lines64 = jobs[1].ok; plan->has_lines64e = jobs[2].ok;
            plan->info.lines = (jobs[0].ok ? 1 : 0) | (jobs[1].ok ? 2 : 0) | (jobs[2].ok ? 4 : 0);
        }
        if (he != hipSuccess) {
            set_error("plan upload failed: %s", hipGetErrorString(he));
            return WAGG_EHIP;                                         // (plan_guard deletes the plan)
        }
        plan_guard.release();
        *out = plan;
        return WAGG_OK;
    } catch (const std::bad_alloc &) {
        set_error("host allocation failed while building the plan");
        return WAGG_ENOMEM;
    } catch (const std::exception &e) {                               // nothing may cross the C boundary
        set_error("plan build failed: %s", e.what());
        return WAGG_EINTERNAL;
    } catch (...) {
        set_error("plan build failed: unknown exception");
        return WAGG_EINTERNAL;
    }
}

extern "C" int wagg_plan_destroy(wagg_plan *plan) {
    delete plan;
    return WAGG_OK;
}

extern "C" int wagg_plan_get_info_sized(const wagg_plan *plan, void *info, uint64_t size) {
    WAGG_REQUIRE(plan && info, "NULL argument");
    wagg::copy_sized(info, size, &plan->info, sizeof(plan->info));
    return WAGG_OK;
}
extern "C" int wagg_plan_get_info(const wagg_plan *plan, wagg_plan_info *info) {
    return wagg_plan_get_info_sized(plan, info, sizeof(wagg_plan_info));
}

extern "C" int wagg_plan_status(const wagg_plan *plan, void *stream) {
    WAGG_REQUIRE(plan != nullptr, "plan is NULL");
    WAGG_HIP(hipStreamSynchronize((hipStream_t)stream));
    return wagg::check_timeout(plan);
}

extern "C" int wagg_plan_get_den(const wagg_plan *plan, double *den_host) {
    WAGG_REQUIRE(plan && (den_host || plan->info.R == 0), "NULL argument");
    if (plan->info.R) std::memcpy(den_host, plan->den_host.data(), sizeof(double) * (size_t)plan->info.R);
    return WAGG_OK;
}

int wagg::entry::apply_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx,
                              int layout, float *out_dev, int64_t ldo, int out_layout, void *stream) {
    int rc = wagg::check_apply_args(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    return wagg::launch_sparse<float, 64>(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout,
                                          (hipStream_t)stream);
}

int wagg::entry::apply_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx,
                              int layout, double *out_dev, int64_t ldo, int out_layout, void *stream) {
    int rc = wagg::check_apply_args(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    return wagg::launch_sparse<double, 32>(plan, X_dev, T, ldx, layout, out_dev, ldo, out_layout,
                                           (hipStream_t)stream);
}

namespace wagg {
// (x + offset)^p for p = 1..n_pow, each aggregated like wagg_apply.  Power p lands at
// out + (p - 1) * out_pstride.
template <typename T, int TB>
static int apply_poly(const wagg_plan *plan, const T *X, int64_t Tn, int64_t ldx, int layout, double offset,
                      int pow_first, int n_pow, T *out, int64_t ldo, int64_t out_pstride, int out_layout,
                      hipStream_t st) {
    int rc = check_apply_args(plan, X, Tn, ldx, layout, out, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    WAGG_REQUIRE(pow_first >= 1 && n_pow >= 1 && pow_first + n_pow - 1 <= 16,
                 "powers must lie in [1, 16], got %d..%d", pow_first, pow_first + n_pow - 1);
    const int64_t orows = out_layout == WAGG_OUT_TR ? Tn : (int64_t)plan->info.R;
    WAGG_REQUIRE(n_pow == 1 || out_pstride >= orows * ldo, "out_pstride %lld overlaps the previous power",
                 (long long)out_pstride);
    for (int i = 0; i < n_pow && rc == WAGG_OK; i += 4) {     // one pass over X per four powers
        const int n = n_pow - i < 4 ? n_pow - i : 4;
        rc = launch_sparse<T, TB>(plan, X, Tn, ldx, layout, out + (int64_t)i * out_pstride, ldo, out_layout, st,
                                  (T)offset, pow_first + i, n, out_pstride);
    }
    return rc;
}

// the device a plan lives on must be the current one (single-device forms) / the slot's device (multi-device form)
static int check_plan_device(const wagg_plan *plan) {
    int cur = 0;
    WAGG_HIP(hipGetDevice(&cur));
    WAGG_REQUIRE(cur == plan->device, "the plan was created on device %d, the current device is %d", plan->device, cur);
    return WAGG_OK;
}

// The row-block pipeline of a host-resident (time, gridcell) field with a (time, region) result (wagg_host.h), for the plain
// aggregation and for the fused powers: `launch(xd, rows, ldx_dev, od, st, compact)` queues the device work of one block
// -- rows x ldx_dev cells in, n_planes planes of rows x ldo out (plane k at od + k * rows * ldo) -- on `st`.
// `dc`: the whole-line chunking the kernel of this call reads (nullptr: none -- no lines-only form); `X2`: a second field.
template <typename T, typename LaunchFn>
static int host_rows_pipeline(const wagg_plan *plan, const T *X, int64_t Tn, int64_t ldx, T *out, int64_t ldo, int flags,
                              int n_planes, int64_t opstride, LaunchFn launch, const SparsePlanDev *dc_ = nullptr, bool dc_given = false,
                              const T *X2 = nullptr) {
    HostRowsArgs a;
    a.X_host = reinterpret_cast<const char *>(X); a.out_host = reinterpret_cast<char *>(out);
    a.X2_host = reinterpret_cast<const char *>(X2);
    a.Tn = Tn;
    a.ldx_bytes = ldx * (int64_t)sizeof(T); a.xrow_bytes = plan->info.G * (int64_t)sizeof(T);
    a.ldo_bytes = ldo * (int64_t)sizeof(T); a.orow_bytes = (int64_t)plan->info.R * (int64_t)sizeof(T);
    a.quantum = 64; a.flags = flags; a.n_dev = 1; a.devices = nullptr;
    a.n_planes = n_planes; a.opstride_bytes = opstride * (int64_t)sizeof(T);
    a.release = [&](int, hipStream_t st) { plan->drop_staging(st); };
    // lines only: host threads pack the quads the whole-line chunking fetches into page-locked pieces and only those
    // cross PCIe (c2-real: 64 % of a fp32 row, 47 % of a fp64 row).  Taken when asked for, when the plan has that
    // chunking for T, when the row shrinks to <= 80 % and the field is large enough to be worth a thread team; a team
    // that cannot start (ring in use by a concurrent call, too few usable CPUs) means the plain pipeline below
    const SparsePlanDev &dc = dc_given ? (dc_ ? *dc_ : plan->d) : (sizeof(T) == 4 ? plan->dl : plan->dl64);
    const bool has_c = (dc_given ? dc_ != nullptr : (sizeof(T) == 4 ? plan->has_lines : plan->has_lines64)) && dc.Gc > 0 && !dc.run_len.empty();
    // ... and of those lines only the QUADS (16 bytes) that hold a referenced cell ("quads only", round 6; c2-real: 33.5 % of a
    // fp32 row instead of 63.6 %, 10.2 instead of 18.1 ms for packing + copy in tools/host_granule_gonogo.sh) when the packing
    // team is large enough for the shorter runs (116 instead of 455 bytes on average: twelve threads stay ahead of PCIe,
    // eight do not -- profiles/r06_host_granule.txt) and the caller has not asked for whole lines (WAGG_HOST_LINES_WHOLE)
    const bool quads = has_c && dc.Gq > 0 && !dc.run_len_q.empty() && !(flags & WAGG_HOST_LINES_WHOLE) && gather_team_threads() >= 10;
    const int64_t Gc = quads ? dc.Gq : dc.Gc;
    if ((flags & WAGG_HOST_LINES) && has_c && 5 * Gc <= 4 * (int64_t)plan->info.G &&
        Tn * (int64_t)plan->info.G * (int64_t)sizeof(T) >= ((int64_t)64 << 20)) {
        const std::vector<int64_t> &rs = quads ? dc.run_src_q : dc.run_src;
        const std::vector<int32_t> &rl = quads ? dc.run_len_q : dc.run_len;
        std::vector<int64_t> src(rs.size());
        std::vector<int32_t> len(rl.size());
        for (size_t k = 0; k < src.size(); ++k) { src[k] = rs[k] * (int64_t)sizeof(T); len[k] = rl[k] * (int32_t)sizeof(T); }
        HostRowsArgs c = a;
        c.run_src = src.data(); c.run_len = len.data(); c.n_runs = (int64_t)src.size(); c.crow_bytes = Gc * (int64_t)sizeof(T);
        const int mode = quads ? COMPACT_QUADS : COMPACT_LINES;
        c.apply = [&](int, const void *xd, int64_t rows, void *od, hipStream_t st) {
            return launch(static_cast<const T *>(xd), rows, Gc, static_cast<T *>(od), st, mode);
        };
        const int rc = stream_host_rows_any(c);
        if (rc != WAGG_EUNSUPPORTED) return rc;
        clear_error();
    }
    a.apply = [&](int, const void *xd, int64_t rows, void *od, hipStream_t st) {
        return launch(static_cast<const T *>(xd), rows, ldx, static_cast<T *>(od), st, COMPACT_NONE);
    };
    return stream_host_rows_any(a);
}

template <typename T, typename F>
static int apply_host(const wagg_plan *plan, const T *X, int64_t Tn, int64_t ldx, int layout, T *out,
                      int64_t ldo, int out_layout, int flags, F fn) {
    clear_error();
    int rc = check_apply_args(plan, X, Tn, ldx, layout, out, ldo, out_layout);
    if (rc != WAGG_OK || Tn == 0) return rc;
    WAGG_REQUIRE((flags & ~(WAGG_HOST_PIN | WAGG_HOST_WHOLE | WAGG_HOST_LINES | WAGG_HOST_LINES_WHOLE)) == 0, "unknown host flags 0x%x", flags);
    if ((rc = check_plan_device(plan)) != WAGG_OK) return rc;
    if (layout == WAGG_LAYOUT_TG && out_layout == WAGG_OUT_TR && !(flags & WAGG_HOST_WHOLE)) {
        rc = host_rows_pipeline<T>(plan, X, Tn, ldx, out, ldo, flags, 1, 0,
                                   [&](const T *xd, int64_t rows, int64_t ldx_dev, T *od, hipStream_t st, int compact) {
                                       if (!compact) return fn(plan, xd, rows, ldx_dev, WAGG_LAYOUT_TG, od, ldo, WAGG_OUT_TR, (void *)st);
                                       return launch_sparse<T, (sizeof(T) == 4 ? 64 : 32)>(plan, xd, rows, ldx_dev, WAGG_LAYOUT_TG, od, ldo, WAGG_OUT_TR, st,
                                                                                           T(0), 0, 1, 0, nullptr, nullptr, 0, compact);
                                   });
        if (rc != WAGG_OK) return rc;
        return check_timeout(plan);
    }
    const int64_t xrows = layout == WAGG_LAYOUT_TG ? Tn : plan->info.G;
    const int64_t orows = out_layout == WAGG_OUT_TR ? Tn : plan->info.R;
    ScratchBuf<T> dx, dout;                      // (call-lifetime blocks: from the pool, wagg_scratch.hip; the device is drained below)
    WAGG_HIP(dx.alloc((size_t)(xrows * ldx)));
    WAGG_HIP(dout.alloc((size_t)(orows * ldo)));
    const int64_t xcols = layout == WAGG_LAYOUT_TG ? plan->info.G : Tn, ocols = out_layout == WAGG_OUT_TR ? plan->info.R : Tn;
    const bool pin = (flags & WAGG_HOST_PIN) != 0;
    if ((rc = copy_to_device(dx.p, X, sizeof(T) * (size_t)((xrows - 1) * ldx + xcols), pin)) != WAGG_OK) return rc;
    WAGG_HIP(hipMemset(dout.p, 0, sizeof(T) * (size_t)(orows * ldo)));
    rc = fn(plan, dx.p, Tn, ldx, layout, dout.p, ldo, out_layout, nullptr);
    if (rc != WAGG_OK) return rc;
    WAGG_HIP(hipDeviceSynchronize());
    if (int rc2 = check_timeout(plan)) return rc2;
    return copy_rows_to_host(out, dout.p, orows, sizeof(T) * (size_t)ldo, sizeof(T) * (size_t)ocols, pin);
}

// The fused powers of a host-resident field (wagg_apply_poly_host_*): the same pipeline, n_pow result planes per block.
template <typename T>
static int apply_poly_host(const wagg_plan *plan, const T *X, int64_t Tn, int64_t ldx, double offset, int pow_first, int n_pow,
                           T *out, int64_t ldo, int64_t out_pstride, int flags) {
    clear_error();
    int rc = check_apply_args(plan, X, Tn, ldx, WAGG_LAYOUT_TG, out, ldo, WAGG_OUT_TR);
    if (rc != WAGG_OK) return rc;
    WAGG_REQUIRE(pow_first >= 1 && n_pow >= 1 && pow_first + n_pow - 1 <= 16, "powers must lie in [1, 16], got %d..%d", pow_first,
                 pow_first + n_pow - 1);
    WAGG_REQUIRE(n_pow == 1 || out_pstride >= Tn * ldo, "out_pstride %lld overlaps the previous power", (long long)out_pstride);
    WAGG_REQUIRE((flags & ~(WAGG_HOST_PIN | WAGG_HOST_LINES | WAGG_HOST_LINES_WHOLE)) == 0, "unknown host flags 0x%x", flags);
    if (Tn == 0) return WAGG_OK;
    if ((rc = check_plan_device(plan)) != WAGG_OK) return rc;
    constexpr int TB = sizeof(T) == 4 ? 64 : 32;
    rc = host_rows_pipeline<T>(plan, X, Tn, ldx, out, ldo, flags, n_pow, out_pstride,
                               [&](const T *xd, int64_t rows, int64_t ldx_dev, T *od, hipStream_t st, int compact) {
                                   int r2 = WAGG_OK;
                                   const int64_t ps = rows * ldo;                 // planes of one block lie side by side
                                   for (int i = 0; i < n_pow && r2 == WAGG_OK; i += 4)      // one pass over the block per four powers
                                       r2 = launch_sparse<T, TB>(plan, xd, rows, ldx_dev, WAGG_LAYOUT_TG, od + (int64_t)i * ps, ldo, WAGG_OUT_TR, st,
                                                                 (T)offset, pow_first + i, n_pow - i < 4 ? n_pow - i : 4, ps, nullptr, nullptr, 0, compact);
                                   return r2;
                               });
    if (rc != WAGG_OK) return rc;
    return check_timeout(plan);
}

// Snyder degree days of two host-resident fields (wagg_apply_edd_host_*): both fields go through the pipeline together,
// n_thr result planes per block.  Lines only: the chunking the degree-day kernel reads (128-cell chunks of 16-cell lines for
// fp32, the 64-cell chunks for fp64), one packed row = tasmin's lines then tasmax's.
template <typename T>
static int apply_edd_host(const wagg_plan *plan, const T *tmin, const T *tmax, int64_t Tn, int64_t ldx, double offset,
                          const double *thr, int n_thr, T *out, int64_t ldo, int64_t out_pstride, int flags) {
    clear_error();
    int rc = check_apply_args(plan, tmin, Tn, ldx, WAGG_LAYOUT_TG, out, ldo, WAGG_OUT_TR);
    if (rc != WAGG_OK) return rc;
    WAGG_REQUIRE(Tn == 0 || tmax != nullptr, "tasmax is NULL");
    WAGG_REQUIRE(n_thr >= 1 && n_thr <= 64 && thr != nullptr, "need 1..64 thresholds");
    WAGG_REQUIRE(n_thr == 1 || out_pstride >= Tn * ldo, "out_pstride %lld overlaps the previous threshold", (long long)out_pstride);
    WAGG_REQUIRE((flags & ~(WAGG_HOST_PIN | WAGG_HOST_LINES | WAGG_HOST_LINES_WHOLE)) == 0, "unknown host flags 0x%x", flags);
    if (Tn == 0) return WAGG_OK;
    if ((rc = check_plan_device(plan)) != WAGG_OK) return rc;
    constexpr int TB = sizeof(T) == 4 ? 64 : 32;
    const bool lcv_off = (plan->flags & (WAGG_PLAN_NO_LC | WAGG_PLAN_NO_STREAM | WAGG_PLAN_LC_MFMA)) != 0;
    const bool edd_lcv = (sizeof(T) == 4 ? plan->has_lines64 : plan->has_lines64e) && !lcv_off;
    const SparsePlanDev *dc = edd_lcv ? (sizeof(T) == 4 ? &plan->dl64 : &plan->dl64e) : nullptr;
    rc = host_rows_pipeline<T>(plan, tmin, Tn, ldx, out, ldo, flags, n_thr, out_pstride,
                               [&](const T *xd, int64_t rows, int64_t ldx_dev, T *od, hipStream_t st, int compact) {
                                   // whole rows: tasmax's block behind tasmin's; packed rows: one row = tasmin's lines, then tasmax's
                                   // (packed rows: ldx_dev = the cells of ONE field's compact row)
                                   const T *x2 = compact ? xd + ldx_dev : xd + rows * ldx_dev;
                                   const int64_t ld = compact ? 2 * ldx_dev : ldx_dev;
                                   const int64_t ps = rows * ldo;
                                   int r2 = WAGG_OK;
                                   for (int i = 0; i < n_thr && r2 == WAGG_OK; i += 4)
                                       r2 = launch_sparse<T, TB>(plan, xd, rows, ld, WAGG_LAYOUT_TG, od + (int64_t)i * ps, ldo, WAGG_OUT_TR, st,
                                                                 (T)offset, XF_EDD, 1, ps, x2, thr + i, n_thr - i < 4 ? n_thr - i : 4, compact);
                                   return r2;
                               },
                               dc, true, tmax);
    if (rc != WAGG_OK) return rc;
    return check_timeout(plan);
}

// Multi-device form (SURVEY 8b `n_devices`, 8e "one process driving all devices"): plan replica s lives on device
// devices[s]; the row blocks of the host field are dealt round-robin to the devices, each device runs its own H2D /
// kernels / D2H pipeline on its own PCIe link from its own host thread, and every block's result lands directly in
// the caller's rows -- no exchange between devices, no collective.  The same device may appear more than once (two
// replicas on one GPU: how the path is exercised on a one-GPU box).
template <typename T, typename F>
static int apply_host_multi(const wagg_plan *const *plans, const int *devices, int n, const T *X, int64_t Tn, int64_t ldx,
                            T *out, int64_t ldo, int flags, F fn) {
    clear_error();
    WAGG_REQUIRE(plans && devices && n >= 1 && n <= 64, "need 1..64 plan replicas and their devices");
    WAGG_REQUIRE((flags & ~WAGG_HOST_PIN) == 0, "unknown host flags 0x%x", flags);
    for (int s = 0; s < n; ++s) {
        WAGG_REQUIRE(plans[s] != nullptr, "plan replica %d is NULL", s);
        WAGG_REQUIRE(plans[s]->device == devices[s], "plan replica %d lives on device %d, not %d", s, plans[s]->device, devices[s]);
        WAGG_REQUIRE(plans[s]->info.G == plans[0]->info.G && plans[s]->info.R == plans[0]->info.R,
                     "plan replica %d has another shape", s);
        int rc = check_apply_args(plans[s], X, Tn, ldx, WAGG_LAYOUT_TG, out, ldo, WAGG_OUT_TR);
        if (rc != WAGG_OK) return rc;
    }
    if (Tn == 0) return WAGG_OK;
    int rc = stream_host_rows<T>(X, Tn, ldx, plans[0]->info.G, out, ldo, plans[0]->info.R, flags, 64, n, devices,
                                 [&](int s, const T *xd, int64_t rows, T *od, hipStream_t st) {
                                     return fn(plans[s], xd, rows, ldx, WAGG_LAYOUT_TG, od, ldo, WAGG_OUT_TR, (void *)st);
                                 },
                                 [&](int s, hipStream_t st) { plans[s]->drop_staging(st); });
    for (int s = 0; s < n && rc == WAGG_OK; ++s) rc = check_timeout(plans[s]);
    return rc;
}
}  // namespace wagg

int wagg::entry::apply_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx,
                                   int layout, float *out_host, int64_t ldo, int out_layout) {
    return wagg::apply_host<float>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, WAGG_HOST_PIN | WAGG_HOST_LINES, entry::apply_f32);
}
int wagg::entry::apply_host_ex_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx,
                                      int layout, float *out_host, int64_t ldo, int out_layout, int flags) {
    return wagg::apply_host<float>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, flags, entry::apply_f32);
}
int wagg::entry::apply_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx,
                                   int layout, double *out_host, int64_t ldo, int out_layout) {
    return wagg::apply_host<double>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, WAGG_HOST_PIN | WAGG_HOST_LINES, entry::apply_f64);
}
int wagg::entry::apply_host_ex_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx,
                                      int layout, double *out_host, int64_t ldo, int out_layout, int flags) {
    return wagg::apply_host<double>(plan, X_host, T, ldx, layout, out_host, ldo, out_layout, flags, entry::apply_f64);
}
int wagg::entry::apply_host_multi_f32(const wagg_plan *const *plans, const int *devices, int n_devices, const float *X_host,
                                         int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags) {
    return wagg::apply_host_multi<float>(plans, devices, n_devices, X_host, T, ldx, out_host, ldo, flags, entry::apply_f32);
}
int wagg::entry::apply_host_multi_f64(const wagg_plan *const *plans, const int *devices, int n_devices, const double *X_host,
                                         int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags) {
    return wagg::apply_host_multi<double>(plans, devices, n_devices, X_host, T, ldx, out_host, ldo, flags, entry::apply_f64);
}

int wagg::entry::apply_poly_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, double offset, int pow_first,
                                        int n_pow, float *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    return wagg::apply_poly_host<float>(plan, X_host, T, ldx, offset, pow_first, n_pow, out_host, ldo, out_pstride, flags);
}
int wagg::entry::apply_poly_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, double offset, int pow_first,
                                        int n_pow, double *out_host, int64_t ldo, int64_t out_pstride, int flags) {
    return wagg::apply_poly_host<double>(plan, X_host, T, ldx, offset, pow_first, n_pow, out_host, ldo, out_pstride, flags);
}

int wagg::entry::apply_edd_host_f32(const wagg_plan *plan, const float *tasmin_host, const float *tasmax_host, int64_t T, int64_t ldx,
                                       double offset, const double *thresholds, int n_thr, float *out_host, int64_t ldo,
                                       int64_t out_pstride, int flags) {
    return wagg::apply_edd_host<float>(plan, tasmin_host, tasmax_host, T, ldx, offset, thresholds, n_thr, out_host, ldo, out_pstride, flags);
}
int wagg::entry::apply_edd_host_f64(const wagg_plan *plan, const double *tasmin_host, const double *tasmax_host, int64_t T, int64_t ldx,
                                       double offset, const double *thresholds, int n_thr, double *out_host, int64_t ldo,
                                       int64_t out_pstride, int flags) {
    return wagg::apply_edd_host<double>(plan, tasmin_host, tasmax_host, T, ldx, offset, thresholds, n_thr, out_host, ldo, out_pstride, flags);
}

int wagg::entry::apply_poly_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout,
                                   double offset, int pow_first, int n_pow, float *out_dev, int64_t ldo,
                                   int64_t out_pstride, int out_layout, void *stream) {
    return wagg::apply_poly<float, 64>(plan, X_dev, T, ldx, layout, offset, pow_first, n_pow, out_dev, ldo, out_pstride,
                                       out_layout, (hipStream_t)stream);
}
int wagg::entry::apply_poly_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout,
                                   double offset, int pow_first, int n_pow, double *out_dev, int64_t ldo,
                                   int64_t out_pstride, int out_layout, void *stream) {
    return wagg::apply_poly<double, 32>(plan, X_dev, T, ldx, layout, offset, pow_first, n_pow, out_dev, ldo, out_pstride,
                                        out_layout, (hipStream_t)stream);
}

namespace wagg {
// Snyder exceedance degree days (transformations.py:7-93) of (tasmin, tasmax) + offset at each of
// n_thr thresholds, aggregated like wagg_apply; threshold i lands at out + i * out_pstride.
template <typename T, int TB>
static int apply_edd(const wagg_plan *plan, const T *tmin, const T *tmax, int64_t Tn, int64_t ldx, int layout,
                     double offset, const double *thr, int n_thr, T *out, int64_t ldo, int64_t out_pstride,
                     int out_layout, hipStream_t st) {
    int rc = check_apply_args(plan, tmin, Tn, ldx, layout, out, ldo, out_layout);
    if (rc != WAGG_OK) return rc;
    WAGG_REQUIRE(Tn == 0 || tmax != nullptr, "tasmax is NULL");
    WAGG_REQUIRE(n_thr >= 1 && thr != nullptr, "need at least one threshold");
    const int64_t orows = out_layout == WAGG_OUT_TR ? Tn : (int64_t)plan->info.R;
    WAGG_REQUIRE(n_thr == 1 || out_pstride >= orows * ldo, "out_pstride %lld overlaps the previous threshold",
                 (long long)out_pstride);
    for (int i = 0; i < n_thr && rc == WAGG_OK; i += 4)         // both fields are read once per four thresholds
        rc = launch_sparse<T, TB>(plan, tmin, Tn, ldx, layout, out + (int64_t)i * out_pstride, ldo, out_layout, st,
                                  (T)offset, XF_EDD, 1, out_pstride, tmax, thr + i, n_thr - i < 4 ? n_thr - i : 4);
    return rc;
}
}  // namespace wagg

int wagg::entry::apply_edd_f32(const wagg_plan *plan, const float *tasmin_dev, const float *tasmax_dev, int64_t T,
                                  int64_t ldx, int layout, double offset, const double *thresholds, int n_thr,
                                  float *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream) {
    return wagg::apply_edd<float, 64>(plan, tasmin_dev, tasmax_dev, T, ldx, layout, offset, thresholds, n_thr, out_dev,
                                      ldo, out_pstride, out_layout, (hipStream_t)stream);
}
int wagg::entry::apply_edd_f64(const wagg_plan *plan, const double *tasmin_dev, const double *tasmax_dev, int64_t T,
                                  int64_t ldx, int layout, double offset, const double *thresholds, int n_thr,
                                  double *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream) {
    return wagg::apply_edd<double, 32>(plan, tasmin_dev, tasmax_dev, T, ldx, layout, offset, thresholds, n_thr, out_dev,
                                       ldo, out_pstride, out_layout, (hipStream_t)stream);
}
