// Device-side plan building for caller-supplied weight tables (wagg_build.hip): the coded segment table -- COO triples
// (cell, region, weight) or CSR (rowptr over cells, region columns, weights) -- is sorted, coalesced and summed ON THE
// DEVICE, so that a table of BASELINE configs[4] size (2.5e8 entries, 3 GB) costs the host nothing but the upload.
// The dense-family constructors (wagg_dense.hip) turn the result into one of their three forms.
//
// Everything here is deterministic: a stable LSD radix sort (no atomics on data, integer histograms only), duplicate
// (cell, region) rows added in their input order (aggregations.py:78 -- S5), denominators (aggregations.py:79) added in a
// fixed order.
#pragma once
#include "wagg_common.h"

namespace wagg {

// Sort key of an entry = its place in the entry-list form (wagg_spmm.hip): one list per bucket = (region block rb, chunk
// of 128 cells, wave), numbered as the plan numbers them, cell-major inside a list:
//   key = (bucket * 128 + cell_in_chunk) * rw + j,  bucket = (rb * n_chunks + chunk) * 16 + wave,
//   region = (rb * 16 + wave) * rw + j.
// The other forms only need duplicates to be neighbours, which any total order gives.
struct EntryKeyGeom {
    int rw = 1, n_rb = 1, n_chunks = 1;
    __host__ __device__ uint64_t range() const { return (uint64_t)n_rb * (uint64_t)n_chunks * 16u * 128u * (uint64_t)rw; }
    __host__ __device__ uint64_t key(int64_t cell, int32_t region) const {
        const uint64_t wv = (uint64_t)(region / rw), j = (uint64_t)(region % rw);
        const uint64_t bucket = ((wv >> 4) * (uint64_t)n_chunks + (uint64_t)(cell >> 7)) * 16u + (wv & 15u);
        return (bucket * 128u + (uint64_t)(cell & 127)) * (uint64_t)rw + j;
    }
    __host__ __device__ int64_t bucket_of(uint64_t k) const { return (int64_t)((k / (uint64_t)rw) >> 7); }
    __host__ __device__ void decode(uint64_t k, int64_t &cell, int32_t &region, int &cic, int &j) const {
        j = (int)(k % (uint64_t)rw);
        k /= (uint64_t)rw;
        cic = (int)(k & 127u);
        k >>= 7;
        const uint64_t wave = k & 15u;
        k >>= 4;
        const uint64_t chunk = k % (uint64_t)n_chunks, rb = k / (uint64_t)n_chunks;
        cell = (int64_t)(chunk * 128u) + cic;
        region = (int32_t)((rb * 16u + wave) * (uint64_t)rw) + j;
    }
};

struct BuildTimes { double upload_s = 0, device_s = 0, total_s = 0; };

// One plan build = one BuildCtx: a stream of its own and ONE device arena for all of the build's scratch.
//   * The stream is non-blocking: nothing of a build is ordered against the null stream or against the streams the process
//     applies other plans on.  (Rounds 4 ran every build kernel on the null stream and called hipDeviceSynchronize() after
//     every scan and sort pass -- each one a stop for every stream of the process.)  The host waits exactly where it reads
//     something back (hipStreamSynchronize on this stream): the dropped-row note, the number of distinct pairs, the tile
//     bitmap, the list lengths.
//   * The arena is sized from the table (build_arena_bytes) and taken once, from the scratch pool (wagg_scratch.hip: the
//     arena and stream of the build before, when it fits); scratch is taken from it with stack
//     discipline (mark / release_to).  Reuse is safe without any wait because every kernel of the build runs on the one
//     stream, in order.  Per-pass hipMalloc / hipFree pairs are gone: hipFree waits for the whole device.
struct BuildCtx {
    hipStream_t st = nullptr;
    char *base = nullptr;
    size_t cap = 0, top = 0, high = 0, peak = 0;      // scratch grows up from 0 to `top`; inputs sit at the far end, from `high` to `cap`
    BuildCtx() = default;
    BuildCtx(const BuildCtx &) = delete;
    BuildCtx &operator=(const BuildCtx &) = delete;
    hipError_t init(size_t arena_bytes);
    void *take_bytes(size_t bytes);                   // 256-byte aligned; NULL when the arena is exhausted
    template <typename T> T *take(size_t count) { return static_cast<T *>(take_bytes(count * sizeof(T))); }
    size_t mark() const { return top; }
    void release_to(size_t m) { top = m; }
    // the far end: a build's INPUT (the uploaded table).  build_sorted_entries drops it as soon as the kernel that reads it
    // is queued -- what is taken there afterwards runs behind that kernel on the one stream
    void *take_input_bytes(size_t bytes);
    template <typename T> T *take_input(size_t count) { return static_cast<T *>(take_input_bytes(count * sizeof(T))); }
    void drop_inputs() { high = cap; }
    hipError_t sync() const { return hipStreamSynchronize(st); }
    ~BuildCtx();
};
// diagnostic build only (WAGG_BUILD_TRACE=1): wall time between milestones of a build, the stream drained at each
#ifdef WAGG_DIAG
void build_stamp(const BuildCtx &ctx, const char *what);
#define WAGG_BUILD_STAMP(ctx, what) wagg::build_stamp((ctx), (what))
#else
#define WAGG_BUILD_STAMP(ctx, what) ((void)0)
#endif
#define WAGG_TAKE(ptr, ctx, T, count)                                                                        \
    do {                                                                                                     \
        (ptr) = (ctx).template take<T>((size_t)(count));                                                     \
        if (!(ptr)) {                                                                                        \
            wagg::set_error("%s:%d: build arena exhausted (%zu of %zu bytes in use)", __FILE__, __LINE__,    \
                            (ctx).top, (ctx).cap);                                                           \
            return WAGG_ENOMEM;                                                                              \
        }                                                                                                    \
    } while (0)

// device scratch of build_sorted_entries + the uploaded table for a table of n rows (the coalesced output -- 16 bytes per
// distinct pair -- and the plan itself are separate allocations): 32 n for the two key / value pairs of the sort, on top of
// them the table (12 n + 8 (G + 1) as CSR, 16 n as COO), later replaced by the histograms and the run ranks
size_t build_arena_bytes(int64_t n, int64_t G, int32_t R, bool csr);

// the coalesced table on the device: n_u distinct (cell, region) pairs in key order
struct SortedEntries {
    EntryKeyGeom geom;
    int64_t n_in = 0, n_valid = 0, n_u = 0;           // rows handed in; with a label and a weight that is not NaN; distinct pairs
    bool chunkwise = false;                           // sorted by the one-pass chunk partition (CSR with ascending columns)
    DevBuf<uint64_t> key;                             // [n_u] ascending
    DevBuf<double> w;                                 // [n_u] fp64 sum of the pair's rows, in input order
    DevBuf<double> den;                               // [R]   sum of the weights of a region's pairs (aggregations.py:79)
};

// Inputs are DEVICE arrays (valid on ctx.st).  Exactly one of cell_dev (COO) / rowptr_dev (CSR, G + 1 offsets) is given.
// region < 0 and NaN weights drop the row (S3, S4); an index outside the grid / the regions fails with WAGG_EINVAL (first bad
// row in the text).  Everything runs on ctx.st; on return the outputs are complete on that stream (not necessarily on the
// host's clock: the caller keeps working on ctx.st or synchronises it).  Scratch comes from ctx's arena above its current mark
// and is released before returning; the arena's input side (where the caller put the table) is dropped once the keys exist.
int build_sorted_entries(BuildCtx &ctx, const int32_t *cell_dev, const int64_t *rowptr_dev, const int32_t *region_dev,
                         const double *w_dev, int64_t n, int64_t G, int32_t R, const EntryKeyGeom &geom, SortedEntries *out,
                         bool general_sort = false);
// what the one-pass partition of a CSR table adds to build_arena_bytes (the chunk bounds and the bucket counts)
size_t chunk_sort_scratch_bytes(const EntryKeyGeom &geom);

// primitives (also used by the synthetic-table generator); scratch from the arena, released before returning
int scan_u32_exclusive(BuildCtx &ctx, uint32_t *data_dev, int64_t n, uint32_t *total_dev /* may be NULL */);
// stable LSD radix sort of (key, value) pairs on the low 8 * passes bits; the sorted pairs end in (keys, vals) -- the
// alternates are scratch of the same size
int radix_sort_pairs(BuildCtx &ctx, uint64_t *keys, uint64_t *vals, uint64_t *keys_alt, uint64_t *vals_alt, int64_t n, int passes);

}  // namespace wagg
