// Internal layout of a dense-family plan (wagg_dense) shared by wagg_dense.hip (full / tile-sparse
// MFMA forms) and wagg_spmm.hip (entry-list form for scattered weights).
#pragma once
#include <memory>

#include "wagg_build.h"
#include "wagg_host.h"

namespace wagg {

// Element transform applied while X is packed for a dense-family apply: the packed
// copy is the only place every element is touched, so tas_poly / snyder_edd cost no extra pass.
template <typename T> struct PackXfT {
    int mode = 0;            // 0: identity; p > 0: (x + off)^p; XF_EDD: snyder_edd1(x + off, x2 + off, thr)
    T off = T(0), thr = T(0);
    const T *X2 = nullptr;   // tasmax (XF_EDD only), same shape and row stride as X
};
typedef PackXfT<float> PackXf;

// transformed value with NaN -> 0 (S6); *inf_seen is set when the result is +-inf (the MFMA forms
// multiply every pair of a stored tile, so the caller must redo such a field in an exact form)
template <typename T>
__device__ __forceinline__ T pack_xf(const PackXfT<T> &xf, T x, T x2, bool &inf_seen) {
    T y = x;
    if (xf.mode > 0) y = xform1<T>(x, xf.off, xf.mode);
    else if (xf.mode == XF_EDD) y = snyder_edd1<T>(x + xf.off, x2 + xf.off, xf.thr);
    inf_seen |= __builtin_isinf(y);
    return y == y ? y : T(0);
}

// ---- entry-list ("SpMM") form: geometry shared by the builders and the kernel -------------------
constexpr int SP_WAVES = 16;                 // waves per workgroup (1024 threads, one workgroup per CU)
constexpr int SP_THREADS = SP_WAVES * 64;
constexpr int SP_ROW = 512;                  // bytes of one cell's row in LDS: one time block of that cell
constexpr int SP_KC = 128;                   // grid cells per LDS chunk (128 rows x 512 B = 64 KiB)
constexpr int SP_ACC = 88;                   // accumulator registers per lane (v[40:127]) = 44 pairs
constexpr int SP_RW_MAX = SP_ACC / 2 - 1;    // regions per wave (43); the last pair swallows the padding entries
constexpr int SP_TRASH = 2 * SP_RW_MAX;      // accumulator index (register offset) of the trash pair
constexpr int SP_GROUP = 8;                  // entries per group
constexpr int SP_PAD_GROUPS = 48;            // zero groups behind the last list (a wave loads 16 groups from its list start)
// a time block = what one wave covers: 128 fp32 timesteps (two per lane) or 64 fp64 timesteps (one per lane);
// either way an accumulator is a register PAIR and a cell's LDS row is 512 bytes
template <typename T> struct SpT {
    static constexpr int TB = SP_ROW / (int)sizeof(T);            // timesteps per block: 128 / 64
    // 32-bit words per 8-entry group: [4 x lo16 pairs][8 x weight (float / double)]
    static constexpr int GW = 4 + 8 * (int)(sizeof(T) / 4);       // 12 / 20
    static constexpr int WSLOT = 128 * (int)sizeof(T);            // bytes of a wave's weights in LDS (one buffer): 128 entries
};
// lo16 of an entry = cell_in_chunk << 9 | accumulator register offset (2 j for the wave's region j)
__host__ __device__ inline unsigned sp_entry_lo(int cell_in_chunk, int j) { return (unsigned)(cell_in_chunk << 9 | 2 * j); }
// word / half-word position of entry `pos` of a list that starts at group 0 (pos counts entries)
template <typename T> __host__ __device__ inline int64_t sp_lo16_index(int64_t pos) {      // in uint16 units
    return (pos >> 3) * (2 * SpT<T>::GW) + (pos & 7);
}
template <typename T> __host__ __device__ inline int64_t sp_w_index(int64_t pos) {         // in uint32 units (first word)
    return (pos >> 3) * SpT<T>::GW + 4 + (pos & 7) * (int)(sizeof(T) / 4);
}

struct SpmmPlan {
    int rw = 0;                              // regions per wave (<= SP_RW_MAX), region r = (rb * 16 + wave) * rw + j
    int n_rb = 0;                            // region blocks of 16 * rw regions
    int n_chunks = 0;                        // ceil(G / SP_KC)
    int64_t nnz = 0, n_groups = 0;           // kept (cell, region) pairs; 8-entry groups incl. padding
    DevBuf<uint32_t> ent;                    // [(n_groups + pad) * GW] words, see SpT<T>::GW
    DevBuf<int32_t> grp_off;                 // [n_rb * n_chunks * 16 + 1]: first group of (rb, chunk, wave)
};

}  // namespace wagg

struct wagg_dense {
    int64_t G = 0;
    int32_t R = 0;
    int n_kt = 0, n_nt = 0;            // k tiles (32 cells in fp32, 16 in fp64) and column tiles (256 regions)
    bool f64 = false;                  // element type of W / X / slabs: fp32 (default) or fp64 (the *_f64 constructors)
    wagg::DevBuf<float> W, den32, slabs, xp;     // W and xp in packed tile order (sized in 4-byte units for both types)
    wagg::DevBuf<double> den64;
    std::vector<double> den_host;
    // tile-sparse form: only the non-empty (32-cell x 256-region) tiles of W are stored, grouped by
    // column tile; tile_kt[i] = k tile of stored tile i (+2 padding entries)
    bool tiled = false;
    int64_t n_tiles = 0;                // stored tiles (n_nt * n_kt when dense)
    wagg::DevBuf<int32_t> tile_kt;
    std::vector<int32_t> nt_first;               // host: first stored tile of every column tile (+ the total): [n_nt + 1]
    // piece table of a tile-sparse launch with n_mb row blocks (wagg_dense.hip: tile_pieces_for): the stored tiles of all
    // (row block, column tile) pairs, end to end, cut into one equal share per workgroup.  One table per row-block count the
    // plan has met, kept until the plan goes (an apply queued on a stream may still be reading one).
    struct TilePieces {
        int n_mb = 0, n_wg = 0, n_slabs = 0;
        wagg::DevBuf<int32_t> tab;               // [n_wg + 1] first piece of a workgroup | [n_pieces][8] | [n_mb * n_nt][2] first slab and slab count of a pair
        int64_t slab_first_at = 0;               // offset of the last part inside tab
    };
    std::vector<std::unique_ptr<TilePieces>> pieces;
    int64_t w_slots() const { return n_tiles * (8192 / 4); }   // 16-byte slots (one tile = 256 x 32 floats)
    // entry-list form (scattered weights, e.g. <= 1 % non-zeros at random positions; fp32 or fp64): no W matrix at all
    bool spmm = false;
    wagg::SpmmPlan sp;
    int ncu = 256;
    int device = 0;                    // the device the plan (W, lists, workspaces) lives on
    // +-inf seen in the (transformed) data of an apply in one of the MFMA forms: host-mapped word
    // ([1]: a pack-free first pass met NaN / +-inf: later applies of this plan go straight to the packed pass)
    int *inf_host = nullptr, *inf_dev = nullptr;
    // pack-free tile-sparse apply: "a numerator of the first pass was not finite" (device word, gates the exact second pass)
    wagg::DevBuf<int> nonfinite;
    int64_t nnz_table = -1;            // distinct (cell, region) pairs of the caller's table (constructors that take one)
    wagg::BuildTimes build;            // constructors that take a caller's table: where the seconds went
    double est_row_s[3] = {0, 0, 0};   // ... and what the form choice went by (seconds per row of X: full, tiles, entries)
    int64_t walked_entries = 0;
    bool one_pass_sort = false;              // the table was put in key order by the one-pass chunk partition
    ~wagg_dense() { if (inf_host) wagg::note_cleanup(hipHostFree(inf_host), "hipHostFree(inf note)"); }
};

namespace wagg {
// wagg_spmm.hip
template <typename T> int spmm_build_synth(wagg_dense *d, uint32_t seed, double fill);
void spmm_geometry(int64_t G, int32_t R, SpmmPlan &sp);
template <typename T> int spmm_build_from_sorted(BuildCtx &ctx, wagg_dense *d, const SortedEntries &se);
int spmm_list_cost(BuildCtx &ctx, const SortedEntries &se, int64_t *walked_entries);
template <typename T> int spmm_apply(wagg_dense *d, const T *X, int64_t Tn, int64_t ldx, const PackXfT<T> &xf, T *out,
                                     int64_t ldo, hipStream_t stream);
// wagg_dense.hip
int dense_alloc_common(wagg_dense *d);
}  // namespace wagg
