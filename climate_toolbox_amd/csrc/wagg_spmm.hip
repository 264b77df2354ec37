// Entry-list ("SpMM") form of the dense-family plan: weights that are scattered all over the grid
// but sparse -- BASELINE configs[4] "CSR <= 1 % nnz" with uniformly random columns.  Neither of the
// other forms fits: the segment-table gather (wagg_sparse.hip) would move ~9 TB for one c5 rank
// shard (every region is spread over the whole grid), and the MFMA forms (wagg_dense.hip) multiply
// every (cell, region) pair of a tile, i.e. 100x the algorithmic 2*T*nnz flops.  Even with K
// compacted per 16-region block a 16x16x4 MFMA tile is only ~6.7 % full, so the matrix pipe cannot
// beat ~10 TF here.  This kernel does exactly 2*T*nnz flops on the vector ALU instead:
//
//   * lanes = timesteps: two fp32 per lane, one fp64.  A workgroup (16 waves, one per CU) owns 512 bytes of
//     timesteps (128 fp32 / 64 fp64) x (16 * rw) regions; every wave keeps rw <= 43 regions as ACCUMULATOR
//     REGISTER PAIRS v[40:125] for the whole k loop (90k accumulators per CU: each byte of X that reaches the
//     CU is used ~7 times).
//   * X is packed once per apply as Xp[time block][cell][128 timesteps] (transform, NaN -> 0 and
//     zero padding fused, like the MFMA forms' pack); a chunk of 128 cells is one contiguous 64 KiB
//     run that goes HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), double buffered, one
//     workgroup barrier per chunk.
//   * W is never a matrix: per (region block, chunk, wave) a list of entries (cell_in_chunk << 9 |
//     accumulator register: 16 bits; weight: 32 / 64 bits), padded to 8-entry groups stored as
//     [4 x lo16 pairs][8 x weight] -- 48 bytes in fp32, 80 in fp64.  While a wave works through this chunk's
//     list, the NEXT chunk's list is already on its way: the lo16 halves two per dword by coalesced vector
//     loads into a second register set (the scalar cache was tried first: ~870 cycles per 64-byte line, 104 ms
//     per c5 rank shard), the WEIGHTS straight into the wave's own LDS slots by LDS-DMA (round 3: a v_readlane
//     per weight word cost more vector-ALU issue time than the FMA it fed, profiles/r03_spmm_ablation_a.txt).
//     Per entry: half a v_readlane (one brings the lo16 of TWO entries; the odd one is a scalar shift), v_bfi
//     (LDS address of the cell), ds_read_b64 (2 x 64 fp32 timesteps of the cell, or 64 doubles: conflict-free),
//     a quarter of a broadcast ds_read_b128 (four weights; two in fp64) and ONE v_pk_fma_f32 / v_fma_f64 whose
//     accumulator pair is picked by the entry itself through the VGPR index mode (s_set_gpr_idx_idx: dst / src2
//     = v[40 + M0[7:0] ...]) and whose weight operand is a VGPR selected by op_sel -- no dynamic-indexing
//     moves, no scratch.  fp64 is the reference's own arithmetic type (aggregations.py:73-80): one timestep
//     per lane, the same 512-byte cell rows.  The loop is generated (tools/gen_spmm_asm.py ->
//     wagg_spmm_asm.inc); tools/check_spmm_codegen.py verifies in the Makefile that the compiler's glue code
//     between the inline-asm statements leaves the live list registers v[3:35] alone.
//   * bound (docs/HISTORY.md (d), profiles/r04_pmc.csv): 2.85 vector + 1.42 LDS instructions per entry; three
//     resources of a CU are each within 1.6x of the kernel's time -- the LDS array (the 512-byte cell row per
//     entry plus the DMA writes), the LDS-DMA ingest of the X stream (36 region blocks re-stream X) and vector
//     issue; the formulation tops out near 25 % of the fp32 vector peak, measured 15.8 % (fp64 13.9 %).  This is
//     the worst-case structure (1 % non-zeros at random positions); frozen since round 4.
//   * k is split into S slices so that every CU gets the same number of items; partial sums go to
//     slabT[slice][region][time] (256-byte coalesced stores straight from the accumulators) and one
//     reduce kernel adds the slices, divides by den[r] (aggregations.py:77-80) and transposes to
//     (time, region).  No atomics: bitwise reproducible.
//   * +-inf data stays exact (S6): only real (cell, region) pairs are multiplied; padding entries
//     carry w = 0 into a trash accumulator.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "wagg_dense_int.h"

namespace wagg {

typedef float f32x32 __attribute__((ext_vector_type(32)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
constexpr int SP_BUF_BYTES = SP_KC * SP_ROW;              // 65,536: one X chunk in LDS
// Every wave keeps the WEIGHTS of its list in LDS (two buffers of 128 weights per wave, behind the X buffers): they get
// there by one LDS-DMA and come back four (fp64: two) at a time by broadcast ds_read_b128 -- a v_readlane per weight word
// cost more vector-ALU time than the FMA it feeds (profiles/r03_spmm_ablation.txt).  fp64: 128 + 32 KiB = all of the LDS.
template <typename T> constexpr int sp_lds_bytes() { return 2 * SP_BUF_BYTES + 2 * SP_WAVES * SpT<T>::WSLOT; }
static_assert(sp_lds_bytes<double>() <= 160 * 1024, "LDS budget");
static_assert(SP_BUF_BYTES == 0x10000, "the buffer bit of the LDS address is bit 16");

#ifndef WAGG_SPMM_ASM_INC          // tools/spmm_ablate.sh builds variants of the generated loop
#define WAGG_SPMM_ASM_INC "wagg_spmm_asm.inc"
#endif
#include WAGG_SPMM_ASM_INC
static_assert(SPMM_W_LDS0 == 2 * SP_BUF_BYTES, "weight slots start behind the X buffers (tools/gen_spmm_asm.py W_LDS0)");

template <typename T>
__global__ __launch_bounds__(SP_THREADS) void spmm_kernel(
    const T *__restrict__ Xp, const uint32_t *__restrict__ ent, const int32_t *__restrict__ grp_off,
    T *__restrict__ slabT, int n_tb, int n_rb, int n_chunks, int cps, int rw, int64_t Gpad,
    int64_t Tpad, int64_t Rpad, int n_items, int n_groups, int knob) {
    constexpr bool F64 = sizeof(T) == 8;
    constexpr int GW = SpT<T>::GW, TB = SpT<T>::TB;
#ifdef WAGG_DIAG     // knob bit 0 (WAGG_SPMM_KNOB, diagnostic build only): no end-of-chunk barrier -- WRONG results,
                     // timing only: the upper bound of what removing the per-chunk synchronisation could give
#define SPMM_CHUNK_SYNC asm volatile("s_waitcnt vmcnt(0)\n\ts_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 Lnb%=\n\ts_barrier\nLnb%=:" : : "s"(knob) : "memory", "scc")
#else
#define SPMM_CHUNK_SYNC asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory")
    (void)knob;
#endif
    extern __shared__ __attribute__((aligned(1024))) char lds[];   // [2][64 KiB] X chunks | [2][16 waves][128 weights]; filled by LDS-DMA only
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) char *)lds;   // 0: the only LDS object
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-contiguous logical ids (speed only): the 16 region blocks of one (slice, time block) run on
    // one XCD at about the same time, so its L2 serves their common X stream
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const unsigned q8 = nblk >> 3, r8 = nblk & 7u;
    const int lid = (int)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot);
    const int voff16 = lane * 16;                    // LDS-DMA: 16 bytes per lane

    for (int item = lid; item < n_items; item += (int)nblk) {
        const int rb = item % n_rb;
        const int tb = (item / n_rb) % n_tb;
        const int ks = item / (n_rb * n_tb);
        const int c0 = ks * cps;
        const int c1 = c0 + cps < n_chunks ? c0 + cps : n_chunks;
        f32x8 b0;                                    // accumulator registers 0-7   v[40:47]   (region j = registers 2j, 2j+1)
        f32x16 b1;                                   //                       8-23  v[48:63]
        f32x32 a1, a2;                               //                       24-55 v[64:95], 56-87 v[96:127]
#pragma unroll
        for (int j = 0; j < 32; ++j) { a1[j] = 0.f; a2[j] = 0.f; if (j < 8) b0[j] = 0.f; if (j < 16) b1[j] = 0.f; }
        // cell g of this time block starts at xbase + g * 512 bytes; this wave moves bytes
        // [wave * 4096, wave * 4096 + 4096) of every 64 KiB chunk
        const char *xbase = reinterpret_cast<const char *>(Xp) + ((int64_t)tb * Gpad) * SP_ROW + wave * 4096;      // (SP_ROW bytes per cell and time block in both types)
        const int32_t *goff = grp_off + ((int64_t)rb * n_chunks) * SP_WAVES + wave;

        if (c0 < c1) {
            {   // chunk c0 -> LDS buffer 0 (4 x 1 KiB pieces per wave)
                const char *src = xbase + (int64_t)c0 * SP_BUF_BYTES;
                const int l0 = lds0 + wave * 4096;
                int m0save, v1;
                asm volatile(
                    "s_mov_b32 %[sv], m0\n\t"
                    "s_mov_b32 m0, %[l0]\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %[vo], %[src]\n\t"
                    "v_add_u32 %[v1], 0x400, %[vo]\n\t"
                    "s_add_u32 m0, %[l0], 0x400\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %[v1], %[src]\n\t"
                    "v_add_u32 %[v1], 0x800, %[vo]\n\t"
                    "s_add_u32 m0, %[l0], 0x800\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %[v1], %[src]\n\t"
                    "v_add_u32 %[v1], 0xc00, %[vo]\n\t"
                    "s_add_u32 m0, %[l0], 0xc00\n\t"
                    "s_nop 0\n\t"
                    "global_load_lds_dwordx4 %[v1], %[src]\n\t"
                    "s_mov_b32 m0, %[sv]\n\t"
                    : [sv] "=&s"(m0save), [v1] "=&v"(v1)
                    : [l0] "s"(l0), [vo] "v"(voff16), [src] "s"(src)
                    : "memory", "scc");
            }
            // ... and its entry list -> register set A.  From here to the end of the chunk loop the list
            // registers live ACROSS statements: nothing but scalar code may sit between two of them
            // (tools/check_spmm_codegen.py checks the compiled kernel).
            const uint64_t p0 = reinterpret_cast<uint64_t>(ent + (int64_t)goff[(int64_t)c0 * SP_WAVES] * GW);
            const int wbase = lds0 + SPMM_W_LDS0 + wave * SpT<T>::WSLOT;      // this wave's weight slots, buffer 0
            if constexpr (F64)
                asm volatile(SPMM_LOAD_LIST_ASM_F64 : : [nplo] "s"((uint32_t)p0), [nphi] "s"((uint32_t)(p0 >> 32)), [bufbit] "s"(lds0),
                             [wl0] "s"(wbase), [wbase] "s"(wbase)
                             : "memory", SPMM_CHUNK_CLOBBERS);
            else
                asm volatile(SPMM_LOAD_LIST_ASM_F32 : : [nplo] "s"((uint32_t)p0), [nphi] "s"((uint32_t)(p0 >> 32)), [bufbit] "s"(lds0),
                             [wl0] "s"(wbase), [wbase] "s"(wbase)
                             : "memory", SPMM_CHUNK_CLOBBERS);
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // Chunks go in PAIRS, statement A then statement B, straight-line (an if/else between the two
        // statements makes the compiler shuffle the pinned accumulators through scratch); an odd count is
        // padded with an empty chunk (n = 0: its statement only re-issues harmless loads).
        // The group offsets of a chunk are scalar loads (several hundred cycles).  They are ISSUED before the
        // end-of-chunk synchronisation of the previous statement and first touched behind it (the empty asm
        // pins that order), so their latency passes while the wave waits at the barrier anyway.
        struct Raw { int g0, g1, gn, cn, real; };
        auto chunk_loads = [&](int c, Raw &r) {
            r.real = c < c1;
            const int cc = r.real ? c : c1 - 1;
            r.cn = cc + 1 < c1 ? cc + 1 : cc;                     // next chunk (the last one re-loads itself: harmless)
            r.g0 = goff[(int64_t)cc * SP_WAVES]; r.g1 = goff[(int64_t)cc * SP_WAVES + 1];
            r.gn = goff[(int64_t)r.cn * SP_WAVES];
        };
        auto chunk_args = [&](Raw &r, int par, int &n, uint64_t &pc, uint64_t &pn, const char *&src, int &l0, int &bufbit, int &wl0) {
            wl0 = lds0 + SPMM_W_LDS0 + ((par ^ 1) * SP_WAVES + wave) * SpT<T>::WSLOT;      // LDS slots of the NEXT chunk's weights
            asm volatile("" : "+s"(r.g0), "+s"(r.g1), "+s"(r.gn));
            n = r.real ? r.g1 - r.g0 : 0;
            pc = reinterpret_cast<uint64_t>(ent + (int64_t)r.g0 * GW);
            pn = reinterpret_cast<uint64_t>(ent + (int64_t)r.gn * GW);
            src = xbase + (int64_t)r.cn * SP_BUF_BYTES;
            l0 = lds0 + (par ^ 1) * SP_BUF_BYTES + wave * 4096;
            bufbit = lds0 + par * SP_BUF_BYTES;
        };
#define SPMM_CHUNK_STMT_(ASM)                                                                                 \
        asm volatile(ASM                                                                                      \
                     : [n] "+s"(n), "+{v[40:47]}"(b0), "+{v[48:63]}"(b1), "+{v[64:95]}"(a1), "+{v[96:127]}"(a2) \
                     : [cplo] "s"((uint32_t)pc), [cphi] "s"((uint32_t)(pc >> 32)), [nplo] "s"((uint32_t)pn),   \
                       [nphi] "s"((uint32_t)(pn >> 32)), [bufbit] "s"(bufbit), [l0] "s"(l0), [src] "s"(src),  \
                       [wl0] "s"(wl0)                                                                         \
                     : "memory", "scc", SPMM_CHUNK_CLOBBERS)
#define SPMM_CHUNK_STMT(AB)                                                                                   \
        do { if constexpr (F64) SPMM_CHUNK_STMT_(SPMM_CHUNK_ASM_##AB##_F64); else SPMM_CHUNK_STMT_(SPMM_CHUNK_ASM_##AB##_F32); } while (0)
        Raw ra, rb_;
        chunk_loads(c0, ra);
        for (int c = c0; c < c1; c += 2) {
            int n, l0, bufbit, wl0;
            uint64_t pc, pn;
            const char *src;
            chunk_args(ra, 0, n, pc, pn, src, l0, bufbit, wl0);
            SPMM_CHUNK_STMT(A);               // this chunk's list in set A, the next one's -> B
            chunk_loads(c + 1, rb_);
            // this wave's pieces of the next chunk and the next list have landed; it is done reading this chunk
            SPMM_CHUNK_SYNC;
            chunk_args(rb_, 1, n, pc, pn, src, l0, bufbit, wl0);
            SPMM_CHUNK_STMT(B);
            chunk_loads(c + 2, ra);
            SPMM_CHUNK_SYNC;
        }
#undef SPMM_CHUNK_STMT
#undef SPMM_CHUNK_STMT_
        // partial sums of this k slice: region j of the wave, one time block per store (fp32: two timesteps per
        // lane; fp64: the register pair is one double)
        T *dst = slabT + (((int64_t)ks * Rpad + ((int64_t)rb * SP_WAVES + wave) * rw) * Tpad) + (int64_t)tb * TB +
                 (F64 ? 1 : 2) * lane;
#pragma unroll
        for (int j = 0; j < SP_RW_MAX; ++j) {
            const int q = 2 * j;
            const float lo = q < 8 ? b0[q & 7] : (q < 24 ? b1[(q - 8) & 15] : (q < 56 ? a1[(q - 24) & 31] : a2[(q - 56) & 31]));
            const float hi = q + 1 < 8 ? b0[(q + 1) & 7] : (q + 1 < 24 ? b1[(q - 7) & 15] : (q + 1 < 56 ? a1[(q - 23) & 31] : a2[(q - 55) & 31]));
            if (j < rw) *reinterpret_cast<float2 *>(dst) = make_float2(lo, hi);     // (fp64: the two halves of the double)
            dst += Tpad;
            asm volatile("" : "+v"(dst));            // one running pointer, not 43 hoisted offsets
        }
    }
}

// X (T x G, row stride ldx) -> Xp[time block][cell][128 timesteps]: transform (tas_poly / snyder_edd),
// NaN -> 0 (S6), zeros for rows >= T and cells >= G.  64 x 64 tiles through LDS, both sides coalesced.  Round 4: 16-byte global
// loads (four / two cells of a row) and 16-byte stores (four / two timesteps of a cell) where the rows allow it -- the pass is
// pure HBM traffic (c5-uniform: 9.5 GB in, 9.5 GB out per apply) and ran at 4.75 TB/s with 4-byte accesses.  The tile is kept
// TRANSPOSED in LDS, [cell][65]: the scalar writes of a loaded piece land in different banks ((4 l + c + t) mod 64 over the 16
// lanes of a row and the four rows of a wave), the reads of a cell's consecutive timesteps are consecutive words.
template <typename T>
__global__ __launch_bounds__(256) void spmm_pack_x_kernel(const T *__restrict__ X, int64_t Tn, int64_t ldx, int64_t G,
                                                          int64_t Gpad, PackXfT<T> xf, T *__restrict__ Xp, int vec_ok) {
    constexpr int TB = SpT<T>::TB;
    constexpr int V = 16 / (int)sizeof(T);                        // elements per 16-byte piece: 4 / 2
    constexpr int PPR = 64 / V;                                   // pieces per 64-element tile row: 16 / 32
    typedef T vecv __attribute__((ext_vector_type(V)));
    __shared__ T tile[64][65];                                    // [cell][timestep]
    const int64_t g0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;      // blockIdx.y counts 64-timestep runs
    const int px = threadIdx.x % PPR, py = threadIdx.x / PPR;     // piece of the row, row of the pass
    bool inf_seen = false;
    const bool edd = xf.mode == XF_EDD;
#pragma unroll 2
    for (int i = py; i < 64; i += 256 / PPR) {
        const int64_t t = t0 + i, g = g0 + V * px;
        vecv v, v2;
#pragma unroll
        for (int c = 0; c < V; ++c) { v[c] = T(0); v2[c] = T(0); }
        if (t < Tn && g < G) {
            if (vec_ok && g + V <= G) {
                v = *reinterpret_cast<const vecv *>(X + t * ldx + g);
                if (edd) v2 = *reinterpret_cast<const vecv *>(xf.X2 + t * ldx + g);
            } else {
#pragma unroll
                for (int c = 0; c < V; ++c)
                    if (g + c < G) { v[c] = X[t * ldx + g + c]; if (edd) v2[c] = xf.X2[t * ldx + g + c]; }
            }
#pragma unroll
            for (int c = 0; c < V; ++c) v[c] = g + c < G ? pack_xf<T>(xf, v[c], v2[c], inf_seen) : T(0);
        }
#pragma unroll
        for (int c = 0; c < V; ++c) tile[V * px + c][i] = v[c];
    }
    __syncthreads();
    const int64_t tb = t0 / TB, toff = t0 % TB;
#pragma unroll 2
    for (int i = py; i < 64; i += 256 / PPR) {                    // cell g0 + i, timesteps t0 + V px .. + V - 1
        vecv v;
#pragma unroll
        for (int c = 0; c < V; ++c) v[c] = tile[i][V * px + c];
        *reinterpret_cast<vecv *>(Xp + ((tb * Gpad + g0 + i) * TB) + toff + V * px) = v;      // (rows of 512 bytes: always aligned)
    }
}

// out[t, r] = sum_s slabT[s][r][t] / den[r]   (aggregations.py:77-80), 64 x 64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void spmm_reduce_kernel(const T *__restrict__ slabT, int S, int64_t Rpad, int64_t Tpad,
                                                          int64_t Tn, int32_t R, const T *__restrict__ den,
                                                          T *__restrict__ out, int64_t ldo) {
    __shared__ T tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, t0 = (int64_t)blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int64_t r = r0 + i;
        T s = T(0);
        if (r < Rpad) {
            const T *p = slabT + r * Tpad + t0 + tx;
            for (int k = 0; k < S; ++k) s += p[(int64_t)k * Rpad * Tpad];
        }
        tile[i][tx] = s;
    }
    __syncthreads();
#pragma unroll 4
    for (int i = ty; i < 64; i += 4) {
        const int64_t t = t0 + i, r = r0 + tx;
        if (t < Tn && r < R) out[t * ldo + r] = tile[tx][i] / den[r];
    }
}

// ---------------------------------------------------------------------------------------------
// builders
// ---------------------------------------------------------------------------------------------
// entry `pos` (counted from the start of the entry buffer, 8 per group) <- (lo16, weight)
template <typename T>
__host__ __device__ inline void sp_store_entry(uint32_t *ent, int64_t pos, unsigned lo16, T w) {
    reinterpret_cast<uint16_t *>(ent)[sp_lo16_index<T>(pos)] = (uint16_t)lo16;
    if constexpr (sizeof(T) == 4) {
        ent[sp_w_index<T>(pos)] = __builtin_bit_cast(uint32_t, w);
    } else {
        const uint64_t b = __builtin_bit_cast(uint64_t, w);
        ent[sp_w_index<T>(pos)] = (uint32_t)b;
        ent[sp_w_index<T>(pos) + 1] = (uint32_t)(b >> 32);
    }
}

// synthetic W[g][r] = hash_u01(g R + r, seed) where hash_u01(g R + r, seed ^ 0x9e3779b9) < fill
// (wagg_dense_create_synth_sparse).  One workgroup per (region block, chunk), one wave per list:
// candidates are visited cell-major, kept ones are appended in that order (ballot + prefix count),
// so the list -- and with it every fp32 sum -- is the same on every build.
template <typename T, bool FILL>
__global__ __launch_bounds__(SP_THREADS) void spmm_synth_kernel(int64_t G, int32_t R, uint32_t seed, float fill, int rw,
                                                                int n_chunks, int32_t *__restrict__ counts,
                                                                const int32_t *__restrict__ grp_off, uint32_t *__restrict__ ent) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x % n_chunks, rb = blockIdx.x / n_chunks;
    const int64_t bucket = ((int64_t)rb * n_chunks + c) * SP_WAVES + wave;
    const int64_t r_first = ((int64_t)rb * SP_WAVES + wave) * rw;
    const int ncand = SP_KC * rw;
    int64_t base = FILL ? (int64_t)grp_off[bucket] * SP_GROUP : 0;      // entry position of this list's first entry
    int kept = 0;
    for (int i0 = 0; i0 < ncand; i0 += 64) {
        const int i = i0 + lane;
        const int gl = i / rw, j = i - gl * rw;
        const int64_t g = (int64_t)c * SP_KC + gl, r = r_first + j;
        bool keep = false;
        uint64_t id = 0;
        if (i < ncand && g < G && r < R) {
            id = (uint64_t)g * (uint64_t)R + (uint64_t)r;
            keep = hash_u01(id, seed ^ 0x9e3779b9u) < fill;
        }
        const unsigned long long m = __ballot(keep);
        if (FILL && keep) {
            const int64_t pos = base + kept + __popcll(m & ((1ull << lane) - 1ull));
            sp_store_entry<T>(ent, pos, sp_entry_lo(gl, j), (T)hash_u01(id, seed));
        }
        kept += __popcll(m);
    }
    if (FILL) {                                   // pad the last group: w = 0 into the trash accumulator
        const int padded = (kept + SP_GROUP - 1) / SP_GROUP * SP_GROUP;
        if (kept + lane < padded) sp_store_entry<T>(ent, base + kept + lane, (unsigned)SP_TRASH, T(0));
    } else if (lane == 0) {
        counts[bucket] = kept;
    }
}

// den[r] = sum_g W[g][r] in fp64, one wave per region, fixed summation order
__global__ __launch_bounds__(256) void spmm_synth_den_kernel(int64_t G, int32_t R, uint32_t seed, float fill,
                                                             double *__restrict__ den) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    double s = 0.0;
    for (int64_t g = lane; g < G; g += 64) {
        const uint64_t id = (uint64_t)g * (uint64_t)R + (uint64_t)r;
        if (hash_u01(id, seed ^ 0x9e3779b9u) < fill) s += (double)hash_u01(id, seed);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) den[r] = s;
}

__global__ void spmm_den32_kernel(const double *__restrict__ den64, float *__restrict__ den32, int32_t R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) den32[r] = (float)den64[r];
}

void spmm_geometry(int64_t G, int32_t R, SpmmPlan &sp) {
    sp.n_rb = (int)(((int64_t)R + SP_WAVES * SP_RW_MAX - 1) / (SP_WAVES * SP_RW_MAX));
    sp.rw = (int)(((int64_t)R + (int64_t)sp.n_rb * SP_WAVES - 1) / ((int64_t)sp.n_rb * SP_WAVES));   // balanced, <= SP_RW_MAX
    sp.n_chunks = (int)((G + SP_KC - 1) / SP_KC);
}

// counts per (region block, chunk, wave) -> first 8-entry group of each list (+ total at the end)
template <typename T>
static int spmm_offsets(wagg_dense *d, const std::vector<int32_t> &counts, hipStream_t st = nullptr) {
    constexpr int GW = SpT<T>::GW;
    SpmmPlan &sp = d->sp;
    std::vector<int32_t> off(counts.size() + 1, 0);
    int64_t groups = 0, nnz = 0;
    for (size_t i = 0; i < counts.size(); ++i) {
        off[i] = (int32_t)groups;
        groups += (counts[i] + SP_GROUP - 1) / SP_GROUP;
        nnz += counts[i];
        WAGG_REQUIRE(groups < (int64_t)0x7fffffff, "entry list too long");
    }
    off[counts.size()] = (int32_t)groups;
    sp.n_groups = groups;
    sp.nnz = nnz;
    WAGG_HIP(sp.grp_off.upload(off, st));
    // padding groups at the end: a wave always loads 16 groups from its list start
    WAGG_HIP(sp.ent.alloc((size_t)(groups + SP_PAD_GROUPS) * GW));
    WAGG_HIP(hipMemsetAsync(sp.ent.p + (size_t)groups * GW, 0, sizeof(uint32_t) * GW * SP_PAD_GROUPS, st));
    return WAGG_OK;
}

template <typename T>
int spmm_build_synth(wagg_dense *d, uint32_t seed, double fill) {
    spmm_geometry(d->G, d->R, d->sp);
    SpmmPlan &sp = d->sp;
    const int64_t n_buckets = (int64_t)sp.n_rb * sp.n_chunks * SP_WAVES;
    WAGG_REQUIRE((int64_t)sp.n_rb * sp.n_chunks < (int64_t)0x7fffffff, "grid too large");
    DevBuf<int32_t> dcounts;
    WAGG_HIP(dcounts.alloc((size_t)n_buckets));
    const dim3 grid((unsigned)((int64_t)sp.n_rb * sp.n_chunks));
    hipLaunchKernelGGL((spmm_synth_kernel<T, false>), grid, dim3(SP_THREADS), 0, nullptr, d->G, d->R, seed, (float)fill, sp.rw,
                       sp.n_chunks, dcounts.p, (const int32_t *)nullptr, (uint32_t *)nullptr);
    WAGG_HIP(hipGetLastError());
    std::vector<int32_t> counts((size_t)n_buckets);
    WAGG_HIP(hipDeviceSynchronize());
    WAGG_HIP(staged_d2h(counts.data(), dcounts.p, sizeof(int32_t) * counts.size()));
    if (int rc = spmm_offsets<T>(d, counts)) return rc;
    hipLaunchKernelGGL((spmm_synth_kernel<T, true>), grid, dim3(SP_THREADS), 0, nullptr, d->G, d->R, seed, (float)fill, sp.rw,
                       sp.n_chunks, (int32_t *)nullptr, (const int32_t *)sp.grp_off.p, (uint32_t *)sp.ent.p);
    WAGG_HIP(hipGetLastError());
    hipLaunchKernelGGL(spmm_synth_den_kernel, dim3((unsigned)((d->R + 3) / 4)), dim3(256), 0, nullptr, d->G, d->R, seed,
                       (float)fill, d->den64.p);
    hipLaunchKernelGGL(spmm_den32_kernel, dim3((unsigned)((d->R + 255) / 256)), dim3(256), 0, nullptr, d->den64.p,
                       d->den32.p, d->R);
    WAGG_HIP(hipGetLastError());
    d->den_host.resize((size_t)d->R);
    WAGG_HIP(hipDeviceSynchronize());
    WAGG_HIP(staged_d2h(d->den_host.data(), d->den64.p, sizeof(double) * (size_t)d->R));
    return WAGG_OK;
}
template int spmm_build_synth<float>(wagg_dense *, uint32_t, double);
template int spmm_build_synth<double>(wagg_dense *, uint32_t, double);

// ---- entry lists from a caller's table, coalesced and sorted on the device (wagg_build.hip) ----------------------------
// The distinct pairs arrive in key order = (bucket, cell in chunk, region in wave): every list is one run of them, already
// in the order the synthetic builder produces (cell-major, regions ascending), so a table that holds the synthetic weights
// gives the same lists -- and with them the same fp32 sums -- bit for bit.
// first[b] = place of bucket b's first pair = lower bound of the bucket's smallest possible key: one binary search per
// bucket (+ one for the end).  (Round 4 looked at every pair and its left neighbour: two 64-bit divisions per pair, 23 ms for
// c5's 2.5e8 pairs -- the largest single kernel of that build; this one reads ~28 keys per bucket, 4.7e6 buckets.)
__global__ __launch_bounds__(256) void spmm_bounds_kernel(const uint64_t *__restrict__ key, int64_t n, EntryKeyGeom geom,
                                                          int64_t n_buckets, int32_t *__restrict__ first) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b > n_buckets) return;
    const uint64_t k0 = (uint64_t)b * 128u * (uint64_t)geom.rw;      // keys of bucket b lie in [k0, k0 + 128 rw)
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (key[mid] < k0) lo = mid + 1; else hi = mid;
    }
    first[b] = (int32_t)lo;
}
// counts[b] = pairs of list b, groups[b] = its 8-entry groups (groups[n_buckets] = 0: the exclusive scan over n_buckets + 1
// values then leaves the total in the last place -- the table of first groups the kernel reads, built without the host)
__global__ __launch_bounds__(256) void spmm_counts_kernel(const int32_t *__restrict__ first, int64_t n_buckets, int32_t *__restrict__ counts,
                                                          uint32_t *__restrict__ groups) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < n_buckets) {
        const int32_t c = first[b + 1] - first[b];
        counts[b] = c;
        groups[b] = (uint32_t)((c + SP_GROUP - 1) / SP_GROUP);
    } else if (b == n_buckets) {
        groups[b] = 0u;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void spmm_fill_kernel(const uint64_t *__restrict__ key, const double *__restrict__ w, int64_t n,
                                                        EntryKeyGeom geom, const int32_t *__restrict__ first,
                                                        const int32_t *__restrict__ grp_off, uint32_t *__restrict__ ent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    int64_t cell;
    int32_t region;
    int cic, j;
    geom.decode(k, cell, region, cic, j);
    const int64_t b = geom.bucket_of(k);
    const int64_t pos = (int64_t)grp_off[b] * SP_GROUP + (i - first[b]);
    sp_store_entry<T>(ent, pos, sp_entry_lo(cic, j), (T)w[i]);
}

// the last group of every list is filled up with w = 0 entries aimed at the trash accumulator
template <typename T>
__global__ __launch_bounds__(256) void spmm_pad_kernel(const int32_t *__restrict__ counts, const int32_t *__restrict__ grp_off,
                                                       int64_t n_buckets, uint32_t *__restrict__ ent) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_buckets) return;
    const int c = counts[b], padded = (c + SP_GROUP - 1) / SP_GROUP * SP_GROUP;
    for (int e = c; e < padded; ++e) sp_store_entry<T>(ent, (int64_t)grp_off[b] * SP_GROUP + e, (unsigned)SP_TRASH, T(0));
}

// What the entry-list kernel would have to walk for this table: an item = (region block, chunk) is done when its LONGEST of
// the 16 per-wave lists is, so the work is 16 x sum over items of the longest list (in whole 8-entry groups) -- equal to the
// pair count for evenly spread weights, several times it when a chunk's regions sit in a few waves (block-local tables).
// Integer sums only (order-free): the same table gives the same figure.
__global__ __launch_bounds__(256) void spmm_cost_kernel(const int32_t *__restrict__ first, int64_t n_items,
                                                        unsigned long long *__restrict__ total) {
    const int64_t it = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long m = 0;
    if (it < n_items) {
        const int32_t *f = first + it * SP_WAVES;
        for (int w = 0; w < SP_WAVES; ++w) {
            const unsigned long long c = (unsigned long long)((f[w + 1] - f[w] + SP_GROUP - 1) / SP_GROUP * SP_GROUP);
            m = c > m ? c : m;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(total, m);
}

int spmm_list_cost(BuildCtx &ctx, const SortedEntries &se, int64_t *walked_entries) {
    const int64_t n_items = (int64_t)se.geom.n_rb * se.geom.n_chunks, n_buckets = n_items * SP_WAVES;
    WAGG_REQUIRE(n_buckets < (int64_t)0x7fffffff, "grid too large");
    const size_t mk = ctx.mark();
    int32_t *first;
    unsigned long long *total;
    WAGG_TAKE(first, ctx, int32_t, n_buckets + 1);
    WAGG_TAKE(total, ctx, unsigned long long, 1);
    WAGG_HIP(hipMemsetAsync(total, 0, sizeof(unsigned long long), ctx.st));
    hipLaunchKernelGGL(spmm_bounds_kernel, dim3((unsigned)((n_buckets + 256) / 256)), dim3(256), 0, ctx.st, (const uint64_t *)se.key.p, se.n_u,
                       se.geom, n_buckets, first);
    hipLaunchKernelGGL(spmm_cost_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, ctx.st, (const int32_t *)first, n_items, total);
    WAGG_HIP(hipGetLastError());
    unsigned long long h = 0;
    WAGG_HIP(staged_d2h(&h, total, sizeof(h), ctx.st));
    *walked_entries = (int64_t)h * SP_WAVES;
    ctx.release_to(mk);
    return WAGG_OK;
}

template <typename T>
int spmm_build_from_sorted(BuildCtx &ctx, wagg_dense *d, const SortedEntries &se) {
    spmm_geometry(d->G, d->R, d->sp);
    SpmmPlan &sp = d->sp;
    WAGG_REQUIRE(se.geom.rw == sp.rw && se.geom.n_rb == sp.n_rb && se.geom.n_chunks == sp.n_chunks, "sort key of another geometry");
    const int64_t n_buckets = (int64_t)sp.n_rb * sp.n_chunks * SP_WAVES;
    WAGG_REQUIRE(n_buckets < (int64_t)0x7fffffff, "grid too large");
    const size_t mk = ctx.mark();
    int32_t *first, *dcounts;
    WAGG_TAKE(first, ctx, int32_t, n_buckets + 1);
    WAGG_TAKE(dcounts, ctx, int32_t, n_buckets);
    const unsigned bblk = (unsigned)((n_buckets + 256) / 256), nblk = (unsigned)((se.n_u + 255) / 256);
    hipLaunchKernelGGL(spmm_bounds_kernel, dim3(bblk), dim3(256), 0, ctx.st, (const uint64_t *)se.key.p, se.n_u, se.geom, n_buckets, first);
    // the lists are laid out on the device: first group of every list = exclusive scan of the group counts, straight into
    // the plan's table; the host only learns the total (it sizes the entry array).  (Round 4 and the first half of round 5
    // brought 19 MB of counts to the host, added them up there and sent 19 MB of offsets back: 6 ms of a c5 build.)
    WAGG_HIP(sp.grp_off.alloc((size_t)n_buckets + 1));
    uint32_t *goff = reinterpret_cast<uint32_t *>(sp.grp_off.p), *gtotal;
    WAGG_TAKE(gtotal, ctx, uint32_t, 1);
    hipLaunchKernelGGL(spmm_counts_kernel, dim3(bblk), dim3(256), 0, ctx.st, (const int32_t *)first, n_buckets, dcounts, goff);
    WAGG_HIP(hipGetLastError());
    if (int rc = scan_u32_exclusive(ctx, goff, n_buckets + 1, gtotal)) return rc;
    uint32_t groups32 = 0;
    WAGG_HIP(staged_d2h(&groups32, gtotal, sizeof(groups32), ctx.st));
    // (a 32-bit sum cannot wrap: at most 2^31 - 1 pairs and as many lists, a list of c pairs has at most c groups or one)
    WAGG_REQUIRE(groups32 < 0x7fffffffu, "entry list too long");
    {
        constexpr int GW = SpT<T>::GW;
        sp.n_groups = (int64_t)groups32;
        sp.nnz = se.n_u;
        WAGG_HIP(sp.ent.alloc((size_t)(sp.n_groups + SP_PAD_GROUPS) * GW));
        // padding groups at the end: a wave always loads 16 groups from its list start
        WAGG_HIP(hipMemsetAsync(sp.ent.p + (size_t)sp.n_groups * GW, 0, sizeof(uint32_t) * GW * SP_PAD_GROUPS, ctx.st));
    }
    if (se.n_u > 0) {
        hipLaunchKernelGGL((spmm_fill_kernel<T>), dim3(nblk), dim3(256), 0, ctx.st, (const uint64_t *)se.key.p, (const double *)se.w.p,
                           se.n_u, se.geom, (const int32_t *)first, (const int32_t *)sp.grp_off.p, sp.ent.p);
        WAGG_HIP(hipGetLastError());
    }
    hipLaunchKernelGGL((spmm_pad_kernel<T>), dim3((unsigned)((n_buckets + 255) / 256)), dim3(256), 0, ctx.st, (const int32_t *)dcounts,
                       (const int32_t *)sp.grp_off.p, n_buckets, sp.ent.p);
    WAGG_HIP(hipGetLastError());
    ctx.release_to(mk);                          // (the caller synchronises ctx.st before the arena goes)
    return WAGG_OK;
}
template int spmm_build_from_sorted<float>(BuildCtx &, wagg_dense *, const SortedEntries &);
template int spmm_build_from_sorted<double>(BuildCtx &, wagg_dense *, const SortedEntries &);

template <typename T>
int spmm_apply(wagg_dense *d, const T *X, int64_t Tn, int64_t ldx, const PackXfT<T> &xf, T *out, int64_t ldo,
               hipStream_t st) {
    constexpr int TB = SpT<T>::TB;
    const SpmmPlan &sp = d->sp;
    const int n_tb = (int)((Tn + TB - 1) / TB);
    const int64_t Tpad = (int64_t)n_tb * TB, Gpad = (int64_t)sp.n_chunks * SP_KC;
    const int64_t Rpad = (int64_t)sp.n_rb * SP_WAVES * sp.rw;
    // k slices: enough items for every CU to get the same number (a multiple of the CU count where
    // possible), at least ~16 chunks per slice
    const int64_t base_items = (int64_t)n_tb * sp.n_rb;
    int S = 1;
    {
        double best = -1.0;
        for (int s = 1; s <= 64 && sp.n_chunks / s >= 16; ++s) {
            const double waves = (double)(base_items * s) / d->ncu;
            const double eff = waves / std::ceil(waves);
            if (eff > best + 0.02) { best = eff; S = s; }
            if (eff > 0.97 && base_items * s >= 4LL * d->ncu) break;
        }
    }
    const int cps = (sp.n_chunks + S - 1) / S;
    const int64_t n_items = base_items * S;
    WAGG_REQUIRE(n_items < (int64_t)0x7fffffff, "grid too large");
    // (the plan's buffers are sized in 4-byte units for both element types)
    const size_t need_x = (size_t)n_tb * (size_t)Gpad * TB * (sizeof(T) / 4), need_s = (size_t)S * (size_t)Rpad * (size_t)Tpad * (sizeof(T) / 4);
    if (d->xp.n < need_x) WAGG_HIP(d->xp.alloc(need_x));
    if (d->slabs.n < need_s) WAGG_HIP(d->slabs.alloc(need_s));
    T *xp = reinterpret_cast<T *>(d->xp.p), *slabs = reinterpret_cast<T *>(d->slabs.p);
    const int vec_ok = ((ldx * sizeof(T)) % 16 == 0) && ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                       (xf.mode != XF_EDD || (reinterpret_cast<uintptr_t>(xf.X2) & 15) == 0);
    hipLaunchKernelGGL((spmm_pack_x_kernel<T>), dim3((unsigned)(Gpad / 64), (unsigned)(n_tb * (TB / 64))), dim3(256), 0, st, X, Tn,
                       ldx, d->G, Gpad, xf, xp, vec_ok);
    WAGG_HIP(hipGetLastError());
    WAGG_HIP(allow_dynamic_lds((const void *)spmm_kernel<T>, sp_lds_bytes<T>()));
    const int nwg = (int)(n_items < d->ncu ? n_items : d->ncu);
    int knob = 0;
#ifdef WAGG_DIAG
    if (const char *k = getenv("WAGG_SPMM_KNOB")) knob = atoi(k);
#endif
    launch_timed(true, spmm_kernel<T>, dim3((unsigned)nwg), dim3(SP_THREADS), sp_lds_bytes<T>(), st, (const T *)xp,
                 (const uint32_t *)sp.ent.p, (const int32_t *)sp.grp_off.p, slabs, n_tb, sp.n_rb, sp.n_chunks, cps,
                 sp.rw, Gpad, Tpad, Rpad, (int)n_items, (int)sp.n_groups, knob);
    WAGG_HIP(hipGetLastError());
    const T *den;
    if constexpr (sizeof(T) == 4) den = d->den32.p; else den = d->den64.p;
    hipLaunchKernelGGL((spmm_reduce_kernel<T>), dim3((unsigned)((Rpad + 63) / 64), (unsigned)(Tpad / 64)), dim3(256), 0, st,
                       (const T *)slabs, S, Rpad, Tpad, Tn, d->R, den, out, ldo);
    WAGG_HIP(hipGetLastError());
    return WAGG_OK;
}
template int spmm_apply<float>(wagg_dense *, const float *, int64_t, int64_t, const PackXfT<float> &, float *, int64_t, hipStream_t);
template int spmm_apply<double>(wagg_dense *, const double *, int64_t, int64_t, const PackXfT<double> &, double *, int64_t, hipStream_t);

}  // namespace wagg
