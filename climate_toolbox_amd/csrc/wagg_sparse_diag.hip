// Diagnostic build only (`make diag` -> libwagg_diag.so; never part of libwagg.so): the round-1/2 loader/consumer kernel
// with MFMA consumers -- the one kernel of this code base whose waves wait on each other through a bounded spin with a
// sticky, host-mapped timeout word.  It serves no default plan since round 3 (sparse_lcv_kernel took the plain
// aggregation, the fused powers and the degree days) and is kept as the reference point of the consumer comparison in
// docs/HISTORY.md (d) and for the forced-timeout test (plans created with WAGG_PLAN_LC_MFMA; the production library refuses
// that flag with WAGG_EUNSUPPORTED).  The LDS-DMA loader experiment of round 4 (sparse_lcd_kernel) was deleted in round
// 5: its numbers live in profiles/r04_lds_dma_loader.txt and docs/HISTORY.md (d).
#ifndef WAGG_DIAG
#error "wagg_sparse_diag.hip belongs to the diagnostic build (make diag)"
#endif
#include <cmath>
#include <cstdlib>

#include "wagg_sparse_int.h"

namespace wagg {

// ---------------------------------------------------------------------------------------------
// Loader/consumer form (fp32, TG layout, single-chunk groups): ONE 768-thread workgroup per CU.
//   * 8 LOADER waves stream the next item's 64 rows x 64 quads into registers and park them in the
//     other half of a double-buffered LDS image (NaN -> 0 on the way, S6; a flag notes +-inf);
//     they block at VMEM issue for as long as the transfer takes -- which is why they do nothing
//     else (microbenchmarks: a wave that issues its own loads cannot overlap them with work).
//   * 4 CONSUMER waves reduce the current item with the matrix cores: the chunk's segment list is
//     scattered into a dense LDS tile Aw[16 regions][256 cells] (zero elsewhere) and
//     out[e][t] = sum_u Aw[e][u] * img[t][u] runs as 64 v_mfma_f32_16x16x4_f32 per wave (wave c
//     owns timesteps 16c..16c+15); more than 16 regions take further passes.  ~3k cycles per
//     pass against ~6k cycles of load time per item: the kernel is HBM-bound.
//   * one workgroup barrier per item swaps the image halves; the consumer waves synchronise among
//     themselves through a monotonic LDS counter (bounded spin).
// Chunks whose data contain +-inf fall back to the exact per-segment VALU reduction.
// ---------------------------------------------------------------------------------------------
constexpr int LC_LW = 8, LC_CW = 4, LC_THREADS = (LC_LW + LC_CW) * 64;      // the MFMA-consumer form: 8 loader + 4 consumer waves
constexpr int LC_TB = 64;
constexpr int LC_AROW = UC + 4;                 // Aw row stride (elements)
struct LcLds {
    static constexpr size_t img = 0;                                            // [2][64][UROW] f32
    static constexpr size_t aw = img + 2 * sizeof(float) * LC_TB * UROW;        // [16][LC_AROW] f32
    static constexpr size_t seg_w = aw + sizeof(float) * 16 * LC_AROW;          // [2][LC_SEGS] f32
    static constexpr size_t seg_u = seg_w + 2 * sizeof(float) * LC_SEGS;        // [2][LC_SEGS] i32 (packed)
    static constexpr size_t ent_r = seg_u + 2 * sizeof(int32_t) * LC_SEGS;      // [2][LC_ENT] i32
    static constexpr size_t ent_d = ent_r + 2 * sizeof(int32_t) * LC_ENT;       // [2][LC_ENT] f32
    static constexpr size_t ent_s = ent_d + 2 * sizeof(float) * LC_ENT;         // [2][LC_ENT + 2] u16
    static constexpr size_t hdr = ent_s + 2 * sizeof(uint16_t) * (LC_ENT + 2);  // [2][16] i32: ne, ns, -, tb, ..., 8 per-loader-wave inf flags
    static constexpr size_t cnt = hdr + 2 * 16 * sizeof(int32_t);               // consumer barrier counter
    static constexpr size_t total = (cnt + 16 + 15) / 16 * 16;
    static_assert(total <= 160 * 1024, "one workgroup must fit the CU's LDS");
};

// NPOW > 1 (fused tas_poly): the loaders park y = x + pv.xoff; the consumers raise each
// fragment to the powers pv.xpow .. pv.xpow + NPOW - 1 in registers and keep NPOW accumulator sets,
// so X is read from HBM once for NPOW powers; the i-th of them is stored at out + i * out_pstride.  A chunk whose |y| could
// overflow fp32 at the highest power (or holds +-inf) takes the exact path, like +-inf data does.
// EDD (fused Snyder degree days): the loaders stream BOTH fields (tasmin = X, tasmax = pv.X2)
// and keep them in registers; one stage per threshold: they park snyder_edd1(tasmin + xoff, tasmax + xoff,
// thr[k]) (transformations.py:64-87) into the image buffer and the consumers reduce it into output plane
// k, so the two fields are read from HBM once for up to four thresholds.
// Round 3: the plain aggregation left this kernel for sparse_lcv_kernel (vector-ALU consumers, below); what runs here
// are the fused powers, the degree days, and plans created with WAGG_PLAN_LC_MFMA.
template <bool VEC, int NPOW = 1, bool EDD = false>
__global__ __launch_bounds__(LC_THREADS, 3) void sparse_lc_kernel(PlanView<float> pv, const float *__restrict__ X,
                                                                  int64_t Ttot, int64_t ldx, int64_t G,
                                                                  float *__restrict__ out, int64_t ldo,
                                                                  int n_norm, long long n_items,
                                                                  int *__restrict__ timeout_word,
                                                                  unsigned long long *__restrict__ stamps_arg, int knob_arg,
                                                                  int64_t out_pstride = 0, float ylim = 0.f) {
#ifdef WAGG_DIAG
    const int knob = knob_arg;                       // ablation switches / phase stamps: diagnostic build only
    unsigned long long *const stamps = stamps_arg;
#else
    constexpr int knob = 0;
    constexpr unsigned long long *stamps = nullptr;
    (void)knob_arg; (void)stamps_arg;
#endif
    typedef float vec4 __attribute__((ext_vector_type(4)));
    typedef int int4v __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float *img = reinterpret_cast<float *>(smem_raw + LcLds::img);
    float *aw = reinterpret_cast<float *>(smem_raw + LcLds::aw);
    float *sm_w = reinterpret_cast<float *>(smem_raw + LcLds::seg_w);
    int32_t *sm_u = reinterpret_cast<int32_t *>(smem_raw + LcLds::seg_u);
    int32_t *sm_er = reinterpret_cast<int32_t *>(smem_raw + LcLds::ent_r);
    float *sm_ed = reinterpret_cast<float *>(smem_raw + LcLds::ent_d);
    uint16_t *sm_es = reinterpret_cast<uint16_t *>(smem_raw + LcLds::ent_s);
    int32_t *hdr = reinterpret_cast<int32_t *>(smem_raw + LcLds::hdr);
    int *ccnt = reinterpret_cast<int *>(smem_raw + LcLds::cnt);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave < LC_LW;
    const bool out_vec = (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
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
    if (tid == 0) *ccnt = 0;
    lds_only_barrier();
    // diagnostic phase stamps (only when a stamp buffer is passed; nothing else reads it)
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
        constexpr int TPW = LC_TB / LC_LW;                       // 8 rows per wave
        const int tw0 = wave * TPW;
        auto load_desc = [&](Item a, StreamDesc &d) {
            const int32_t *p = pv.chunk_desc + 8 * (int64_t)(pv.c0_normal + a.g);
            const int4v x = *reinterpret_cast<const int4v *>(p);
            const int4v y = *reinterpret_cast<const int4v *>(p + 4);
            d.u0 = x[0]; d.nq = x[1]; d.e0 = x[2]; d.ne = x[3]; d.sb = y[0]; d.ns = y[1]; d.split = 0;
        };
        auto load_cell = [&](const StreamDesc &d) {
            const int c = pv.ucell[d.u0 + (lane < d.nq ? lane : d.nq - 1)] & ~UCELL_UNREF;
#ifdef WAGG_DIAG      // timing-only address patterns on the 720 x 1440 grid (results are wrong)
            if (knob & 16) {                    // aligned 8 x 128-B patch instead of the chunk's quads
                const int rowlen = 1440, pc = (d.u0 >> 6) % 45, pr = ((d.u0 >> 6) / 45) % 90;
                return (pr * 8 + (lane >> 3)) * rowlen + pc * 32 + (lane & 7) * 4;
            }
            if (knob & 32) return ((d.u0 >> 6) % 4050) * 256 + lane * 4;      // 1 KB contiguous
#endif
            return c;
        };
        struct Regs { vec4 v[TPW]; vec4 h[EDD ? TPW : 1]; int mu; float mw; int er, es; float ed; };
        static_assert(LC_SEGS <= LC_LW * 64, "one metadata element per loader thread");
        auto issue = [&](Regs &R, const StreamDesc &d, int cell0, int tb) {
            // small metadata loads first, the rows last (vmcnt retires in order)
            if (!(knob & 2)) {
                const int k = tid < d.ns ? tid : d.ns - 1;
                R.mu = pv.seg_u[d.sb + k];
                R.mw = pv.seg_w[d.sb + k];
                R.er = pv.ent_region[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
                R.ed = pv.ent_den[d.e0 + (tid < d.ne ? tid : d.ne - 1)];
                R.es = pv.ent_seg_begin[d.e0 + (tid < d.ne ? tid : d.ne)];
            }
            const int64_t t0 = (int64_t)tb * LC_TB;
            const int nt = (int)((Ttot - t0) < LC_TB ? (Ttot - t0) : LC_TB);
            const int rbase = tw0 < nt - 1 ? tw0 : nt - 1;
            int cnt = nt - tw0;
            cnt = cnt < 1 ? 1 : (cnt > TPW ? TPW : cnt);
            const float *p = X + (t0 + rbase) * ldx + cell0;
            if constexpr (!EDD) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (VEC) R.v[i] = *reinterpret_cast<const vec4 *>(p);
                    else {
                        const int64_t lim = G - 1 - cell0;
                        R.v[i] = vec4{p[0], p[lim < 1 ? lim : 1], p[lim < 2 ? lim : 2], p[lim < 3 ? lim : 3]};
                    }
                    if (i + 1 < cnt) p += ldx;
                }
            } else {
                // degree days: the two fields row by row (tasmin row i, tasmax row i, ...): loads retire in order, so the
                // arithmetic of row i can start while the rows behind it are still in flight
                const float *p2 = pv.X2 + (t0 + rbase) * ldx + cell0;
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    if (VEC) { R.v[i] = *reinterpret_cast<const vec4 *>(p); R.h[i] = *reinterpret_cast<const vec4 *>(p2); }
                    else {
                        const int64_t lim = G - 1 - cell0;
                        R.v[i] = vec4{p[0], p[lim < 1 ? lim : 1], p[lim < 2 ? lim : 2], p[lim < 3 ? lim : 3]};
                        R.h[i] = vec4{p2[0], p2[lim < 1 ? lim : 1], p2[lim < 2 ? lim : 2], p2[lim < 3 ? lim : 3]};
                    }
                    if (i + 1 < cnt) { p += ldx; p2 += ldx; }
                }
            }
        };
        auto park_meta = [&](const Regs &R, const StreamDesc &d, int tb, int buf, int plane) {
            if (!(knob & 2)) {
            if (tid < d.ns) { sm_u[buf * LC_SEGS + tid] = R.mu; sm_w[buf * LC_SEGS + tid] = R.mw; }
            if (tid < d.ne) { sm_er[buf * LC_ENT + tid] = R.er; sm_ed[buf * LC_ENT + tid] = R.ed; }
            if (tid <= d.ne) sm_es[buf * (LC_ENT + 2) + tid] = (uint16_t)(R.es - d.sb);
            }
            if (tid == 0) { hdr[buf * 16 + 0] = d.ne; hdr[buf * 16 + 1] = d.ns; hdr[buf * 16 + 3] = tb; hdr[buf * 16 + 4] = plane; }
        };
        // degree days of threshold k from the two fields held in registers (they stay untouched for the
        // next threshold): NaN values count 0 (S6), +-inf values send the chunk to the exact path
        auto park_edd = [&](const Regs &R, const StreamDesc &d, int tb, int buf, int k) {
            float *im = img + buf * LC_TB * UROW;
            const float e = k == 0 ? pv.edd_thr[0] : (k == 1 ? pv.edd_thr[1] : (k == 2 ? pv.edd_thr[2] : pv.edd_thr[3]));
            bool inf_seen = false;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                vec4 val;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float y = snyder_edd1<float>(R.v[i][c] + pv.xoff, R.h[EDD ? i : 0][c] + pv.xoff, e);
                    inf_seen |= __builtin_amdgcn_classf(y, 0x204);
                    val[c] = (y == y) ? y : 0.0f;
                }
                *reinterpret_cast<vec4 *>(&im[(tw0 + i) * UROW + 4 * lane]) = val;
            }
            const bool inf_any = __builtin_amdgcn_readfirstlane(__ballot(inf_seen) != 0ull);
            if (lane == 0) hdr[buf * 16 + 8 + wave] = inf_any ? 1 : 0;
            park_meta(R, d, tb, buf, k);
        };
        auto park = [&](Regs &R, const StreamDesc &d, int tb, int buf) {
            float *im = img + buf * LC_TB * UROW;
            // one v_cmp_class per element finds NaN / +-inf; the select runs only if the wave saw any
            bool odd = false;
            if (NPOW > 1) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) R.v[i] = R.v[i] + pv.xoff;
            } else if (pv.xpow > 0) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) R.v[i] = xform4<vec4, float>(R.v[i], pv.xoff, pv.xpow);
            }
            if (!(knob & 4)) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (NPOW > 1) odd |= !(__builtin_fabsf(R.v[i][c]) < ylim);      // NaN, +-inf, or too large to raise
                    else odd |= __builtin_amdgcn_classf(R.v[i][c], 0x207);          // sNaN|qNaN|-inf|+inf
                }
            }
            bool inf_any = false;
            if (__builtin_amdgcn_readfirstlane(__ballot(odd) != 0ull)) {
                bool inf_seen = false;
#pragma unroll
                for (int i = 0; i < TPW; ++i)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float x = R.v[i][c];
                        if (NPOW > 1) inf_seen |= __builtin_fabsf(x) >= ylim;     // exact path
                        else inf_seen |= __builtin_amdgcn_classf(x, 0x204);       // -inf | +inf: exact path
                        R.v[i][c] = (x == x) ? x : 0.0f;                           // NaN data counts 0 (S6)
                    }
                inf_any = __builtin_amdgcn_readfirstlane(__ballot(inf_seen) != 0ull);
            }
            if (lane == 0) hdr[buf * 16 + 8 + wave] = inf_any ? 1 : 0;      // every wave, every item: no reset needed
#pragma unroll
            for (int i = 0; i < TPW; ++i) *reinterpret_cast<vec4 *>(&im[(tw0 + i) * UROW + 4 * lane]) = R.v[i];
            park_meta(R, d, tb, buf, 0);
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
        // item 0 -> RA -> buffer 0; item 1 (if any) already in flight in RB while item 0 is parked
        issue(RA, dq[0], cellq[0], itq[0].tb);
        StreamDesc dPark = dq[0];
        int tbPark = itq[0].tb;
        // look-ahead queue: itq/dq/cellq[0] = next item to issue.  vmcnt retires in order, so the
        // queue's own loads (descriptor of the item after next-next, quad list of next-next) are
        // issued BEFORE a stage's row loads and consumed after them with a counted wait.
        struct Ahead { Item nx; StreamDesc dn; int cn; };
        auto ahead_load = [&](Ahead &a) {
            a.nx = advance(itq[2]);
            if (a.nx.tb * (long long)n_norm + a.nx.g >= n_items) a.nx = itq[2];
            if (knob & 8) { a.dn = dq[2]; a.cn = cellq[1]; return; }   // diagnostic: no look-ahead loads
            load_desc(a.nx, a.dn);
            a.cn = load_cell(dq[2]);
        };
        auto ahead_commit = [&](const Ahead &a) {
            itq[0] = itq[1]; itq[1] = itq[2]; itq[2] = a.nx;
            dq[0] = dq[1]; dq[1] = dq[2]; dq[2] = a.dn;
            cellq[0] = cellq[1]; cellq[1] = a.cn;
        };
        { Ahead a; ahead_load(a); ahead_commit(a); }              // queue now describes items 1, 2, 3
        // two register sets alternate: while one item is parked, the next one's loads are in flight
        const int K = EDD ? pv.n_thr : 1;                         // stages per item (one per degree-day threshold)
        int sbuf = 0;                                             // image buffer of the next stage
        auto lstage = [&](Regs &Rcur, Regs &Rnext, int st) {
            // Rcur holds item st (in flight since the previous call); item st+1 goes to Rnext
            const StreamDesc dn = dq[0];
            const int tbn = itq[0].tb;
            const bool more = st + 1 < nst;
            Ahead a;
            stamp(-1);
            if (more) { ahead_load(a); issue(Rnext, dn, cellq[0], tbn); }
            stamp(0);                                             // loader ph0: issue (blocked at VMEM)
            for (int k = 0; k < K; ++k) {
                park(Rcur, dPark, tbPark, sbuf);
                stamp(2);                                         // ph2: wait for item st + park
                if (k + 1 == K && more) { dPark = dn; tbPark = tbn; ahead_commit(a); }
                stamp(1);                                         // ph1: queue rotation (must not wait for rows)
                lds_only_barrier();                               // stage (st, k) is in buffer sbuf
                stamp(3);                                         // ph3: waiting for the consumers
                sbuf ^= 1;
            }
        };
        // Degree days: ONE register set (the two fields of an item are 64 registers per lane; a second set does not
        // fit the 168-register budget of three waves per SIMD -- tried: 400-500 spills).  The next item's loads are
        // issued right behind the last threshold's park, so they overlap that stage's barrier and reduction only.
        auto lstage_edd = [&](Regs &R, int st) {
            const bool more = st + 1 < nst;
            for (int k = 0; k < K; ++k) {
                stamp(-1);
                park_edd(R, dPark, tbPark, sbuf, k);
                stamp(2);                                         // ph2: wait for the item's rows + degree-day arithmetic + park
                if (k + 1 == K && more) {
                    const StreamDesc dn = dq[0];
                    const int tbn = itq[0].tb;
                    Ahead a;
                    ahead_load(a);
                    issue(R, dn, cellq[0], tbn);
                    dPark = dn; tbPark = tbn;
                    ahead_commit(a);
                }
                stamp(0);                                         // ph0: issue
                lds_only_barrier();
                stamp(3);                                         // ph3: waiting for the consumers
                sbuf ^= 1;
            }
        };
        // Stage s = st * K + k is parked into buffer s & 1; the consumers reduce it after the barrier.
        // The buffer is free: its previous tenant (stage s - 2) was reduced before barrier s - 1.
        if constexpr (EDD) {
            for (int st = 0; st < nst; ++st) lstage_edd(RA, st);
        } else {
            for (int st = 0; st < nst; st += 2) {
                lstage(RA, RB, st);
                if (st + 1 < nst) lstage(RB, RA, st + 1);
            }
        }
        lds_only_barrier();                                       // consumers finish the last item
        if (stamps && tid == 0) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 8 + i] = ph[i];
    } else {
        // =============================== consumer waves, matrix cores ===============================
        const int cw = wave - LC_LW;                              // 0..3: owns timesteps 16cw .. 16cw+15
        const int ctid = tid - LC_LW * 64;                        // 0..255
        const int lr = lane & 15, kq = lane >> 4;
        int epoch = 0;
        // A consumer wave whose barrier spin runs out marks the launch as failed (the host turns the
        // word into WAGG_EHIP), stops computing and storing, and only keeps the workgroup barriers
        // going so that the loaders drain; its partners then time out at their next arrival too.
        bool dead = false;
        auto cbarrier = [&]() {                                   // the 4 consumer waves only (bounded spin)
            if (dead) return;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            epoch += LC_CW;
            int gave_up = 0;
            if (lane == 0) {
#ifdef WAGG_DIAG
                if ((knob & 64) && cw == 3 && epoch > LC_CW) gave_up = 1;       // test hook: wave 3 stops arriving
                else
#endif
                {
                    __hip_atomic_fetch_add(ccnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    int spins = 0;
                    while (__hip_atomic_load(ccnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) {
                        if (++spins > (1 << 24)) { gave_up = 1; break; }
                    }
                }
                if (gave_up && timeout_word) __hip_atomic_store(timeout_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            dead = __builtin_amdgcn_readfirstlane(gave_up) != 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        };
        // the dense weight tile starts all-zero and is returned to all-zero after every pass
        for (int i = ctid * 4; i < 16 * LC_AROW; i += 256 * 4)
            *reinterpret_cast<vec4 *>(&aw[i]) = vec4{0.f, 0.f, 0.f, 0.f};
        const int nstages = nst * (EDD ? pv.n_thr : 1);
        for (int st = 0; st < nstages; ++st) {
            stamp(-1);
            lds_only_barrier();                                   // stage st has been parked
            stamp(3);                                             // consumer ph3: waiting for the loaders
            const int buf = st & 1;
            // degree days: the stage's threshold selects the output plane
            const int64_t plane_off = EDD ? (int64_t)__builtin_amdgcn_readfirstlane(hdr[buf * 16 + 4]) * out_pstride : 0;
            const float *im = img + buf * LC_TB * UROW;
            const int ne = __builtin_amdgcn_readfirstlane(hdr[buf * 16 + 0]);
            const int ns = __builtin_amdgcn_readfirstlane(hdr[buf * 16 + 1]);
            const int64_t t0 = (int64_t)__builtin_amdgcn_readfirstlane(hdr[buf * 16 + 3]) * LC_TB;
            const int nt = (int)((Ttot - t0) < LC_TB ? (Ttot - t0) : LC_TB);
            const int fl = lane < LC_LW ? hdr[buf * 16 + 8 + lane] : 0;
            const bool exact = __builtin_amdgcn_readfirstlane(__ballot(fl != 0) != 0ull);
            if ((knob & 1) || dead) continue;                     // (knob: diagnostic build, consumers idle)
            if (!exact) {
                for (int e0 = 0; e0 < ne; e0 += 16) {
                    // ---- dense weight tile of regions e0..e0+15: scatter the segments ----
                    for (int k = ctid; k < ns; k += 256) {
                        const int pu = sm_u[buf * LC_SEGS + k];
                        const int e = (pu >> 16) & 0xff;
                        if (e >= e0 && e < e0 + 16) aw[(e - e0) * LC_AROW + (pu & 0xff)] = sm_w[buf * LC_SEGS + k];
                    }
                    cbarrier();
                    if (dead) break;
                    stamp(0);                                     // consumer ph0: build the weight tile
                    // ---- out[t][e] = sum_u img[t][u] * Aw[e][u].  A = img (i = timestep), B = Aw^T
                    // (j = region); lane group kq = lane >> 4 walks cells 64 kq .. 64 kq + 63, so one
                    // ds_read_b128 per operand feeds four MFMA k-steps (any 4 distinct cells per step
                    // work as long as A and B agree); conflict-free with the 260-element row stride.
                    // With timesteps on the rows a lane ends up with FOUR CONSECUTIVE timesteps of one
                    // region: one 16-byte store per lane instead of four scattered dwords (consumer
                    // stores queue behind the loaders' row loads, so their count matters) ----
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    f32x4 accp[NPOW][2];
#pragma unroll
                    for (int pp = 0; pp < NPOW; ++pp) accp[pp][0] = accp[pp][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                    const float *ap = im + (16 * cw + lr) * UROW + 64 * kq;
                    const float *bp = aw + lr * LC_AROW + 64 * kq;
                    f32x4 af[2], bf[2];
                    af[0] = *reinterpret_cast<const f32x4 *>(ap);
                    bf[0] = *reinterpret_cast<const f32x4 *>(bp);
#pragma unroll
                    for (int g4 = 0; g4 < 16; ++g4) {
                        if (g4 + 1 < 16) {
                            af[(g4 + 1) & 1] = *reinterpret_cast<const f32x4 *>(ap + 4 * (g4 + 1));
                            bf[(g4 + 1) & 1] = *reinterpret_cast<const f32x4 *>(bp + 4 * (g4 + 1));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 pw = af[g4 & 1];
                        if (NPOW > 1) for (int i = 1; i < pv.xpow; ++i) pw = pw * af[g4 & 1];   // first power of this pass
#pragma unroll
                        for (int pp = 0; pp < NPOW; ++pp) {
                            if (pp > 0) pw = pw * af[g4 & 1];          // y^(pp+1), transformations.py:188
#pragma unroll
                            for (int j = 0; j < 4; ++j)                 // two chains: 40-cycle dependent latency vs 32 issue
                                accp[pp][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(pw[j], bf[g4 & 1][j], accp[pp][j & 1], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // C/D map: column (region) = lane & 15, row (timestep) = 4 * (lane >> 4) + reg
                    const int e = e0 + lr;
                    const int tl = 16 * cw + 4 * kq;
                    if (e < ne && tl < nt) {
                        const float den = sm_ed[buf * LC_ENT + e];
#pragma unroll
                        for (int pp = 0; pp < NPOW; ++pp) {
                            const f32x4 acc = accp[pp][0] + accp[pp][1];
                            float *op = out + plane_off + (int64_t)pp * out_pstride + (int64_t)sm_er[buf * LC_ENT + e] * ldo + t0 + tl;
                            const f32x4 qv = {acc[0] / den, acc[1] / den, acc[2] / den, acc[3] / den};   // :77-80
                            if (out_vec && tl + 3 < nt) {
                                *reinterpret_cast<f32x4 *>(op) = qv;
                            } else {
#pragma unroll
                                for (int rg = 0; rg < 4; ++rg) if (tl + rg < nt) op[rg] = qv[rg];
                            }
                        }
                    }
                    stamp(1);                                     // ph1: MFMAs + stores
                    cbarrier();                                   // every wave is done reading the tile
                    if (dead) break;
                    for (int k = ctid; k < ns; k += 256) {        // return the tile to all-zero
                        const int pu = sm_u[buf * LC_SEGS + k];
                        const int e = (pu >> 16) & 0xff;
                        if (e >= e0 && e < e0 + 16) aw[(e - e0) * LC_AROW + (pu & 0xff)] = 0.f;
                    }
                    if (e0 + 16 < ne) cbarrier();                 // next pass scatters into a clean tile
                    if (dead) break;
                    stamp(2);                                     // ph2: un-scatter + consumer barrier
                }
            } else {
                // ---- exact path (+-inf in the data): per-segment products with the skipna test ----
                for (int e = cw; e < ne; e += LC_CW) {
                    const int s0 = sm_es[buf * (LC_ENT + 2) + e], s1 = sm_es[buf * (LC_ENT + 2) + e + 1];
                    float accx[NPOW];
#pragma unroll
                    for (int pp = 0; pp < NPOW; ++pp) accx[pp] = 0.f;
                    for (int q = s0; q < s1; ++q) {
                        const float y = im[lane * UROW + (sm_u[buf * LC_SEGS + q] & 0xff)], w = sm_w[buf * LC_SEGS + q];
                        float yp = y;
                        if (NPOW > 1) for (int i = 1; i < pv.xpow; ++i) yp *= y;
#pragma unroll
                        for (int pp = 0; pp < NPOW; ++pp) {
                            if (pp > 0) yp *= y;
                            const float p = yp * w;
                            accx[pp] += (p == p) ? p : 0.f;
                        }
                    }
                    if (lane < nt) {
#pragma unroll
                        for (int pp = 0; pp < NPOW; ++pp)
                            out[plane_off + (int64_t)pp * out_pstride + (int64_t)sm_er[buf * LC_ENT + e] * ldo + t0 + lane] =
                                accx[pp] / sm_ed[buf * LC_ENT + e];
                    }
                }
            }
        }
        lds_only_barrier();                                       // matches the loaders' final barrier
        if (stamps && ctid == 0) for (int i = 0; i < 4; ++i) stamps[blockIdx.x * 8 + 4 + i] = ph[i];
    }
}

// (diagnostic build, WAGG_SPARSE_STAMP) phase stamps of a loader/consumer launch: mean cycles per stage and phase, to stderr
int report_lc_stamps(unsigned long long *lc_stamps, long long nw, long long n_items, hipStream_t stream) {
    std::vector<unsigned long long> h(8 * (size_t)nw);
    WAGG_HIP(hipStreamSynchronize(stream));
    WAGG_HIP(staged_d2h(h.data(), lc_stamps, sizeof(unsigned long long) * h.size()));
    WAGG_HIP(hipFree(lc_stamps));
    double sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t i = 0; i < h.size(); ++i) sm[i % 8] += (double)h[i];
    const double stg = (double)n_items;
    fprintf(stderr, "[wagg lc stamp] items=%lld nw=%lld cycles/stage  loader: issue=%.0f rotate=%.0f wait+park=%.0f barrier=%.0f | consumer: tile=%.0f mfma+store=%.0f unscatter=%.0f barrier=%.0f\n",
            n_items, nw, sm[0] / stg, sm[1] / stg, sm[2] / stg, sm[3] / stg, sm[4] / stg, sm[5] / stg, sm[6] / stg, sm[7] / stg);
    return WAGG_OK;
}

int launch_lc_mfma(const wagg_plan *plan, const PlanView<float> &pv, const float *X, int64_t Ttot, int64_t ldx, float *kout,
                   int64_t kldo, int n_norm, bool vec, bool edd, int xpow, int nfuse, int64_t kpstride, hipStream_t stream) {
    // one workgroup per CU: 8 loader + 4 consumer waves
    const int ncu = plan->ncu;
    const long long n_items = (long long)n_norm * ((Ttot + LC_TB - 1) / LC_TB);
    const long long nw = n_items < ncu ? n_items : ncu;
    auto kern = vec ? sparse_lc_kernel<true> : sparse_lc_kernel<false>;
    if (edd) kern = vec ? sparse_lc_kernel<true, 1, true> : sparse_lc_kernel<false, 1, true>;
    if (nfuse == 2) kern = vec ? sparse_lc_kernel<true, 2> : sparse_lc_kernel<false, 2>;
    if (nfuse == 3) kern = vec ? sparse_lc_kernel<true, 3> : sparse_lc_kernel<false, 3>;
    if (nfuse == 4) kern = vec ? sparse_lc_kernel<true, 4> : sparse_lc_kernel<false, 4>;
    // |y| below this can be raised to the nfuse-th power (and summed 512 times) inside fp32
    const float ylim = nfuse > 1 ? std::pow(3.0e38f / 1024.f, 1.0f / (float)(xpow + nfuse - 1)) : 0.f;
    WAGG_HIP(allow_dynamic_lds((const void *)kern, LcLds::total));
    unsigned long long *lc_stamps = nullptr;
    if (diag_set("WAGG_SPARSE_STAMP")) WAGG_HIP(hipMalloc((void **)&lc_stamps, sizeof(unsigned long long) * 8 * (size_t)nw));
    launch_timed(true, kern, dim3((unsigned)nw), dim3(LC_THREADS), LcLds::total, stream, pv, X, Ttot, ldx, (int64_t)plan->info.G, kout, kldo,
                 n_norm, n_items, plan->timeout_dev, lc_stamps, diag_env("WAGG_LC_KNOB"), kpstride, ylim);
    WAGG_HIP(hipGetLastError());
    if (lc_stamps) { if (int rc = report_lc_stamps(lc_stamps, nw, n_items, stream)) return rc; }
    return WAGG_OK;
}

}  // namespace wagg
