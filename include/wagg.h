/* wagg.h -- C-ABI of the MI355X (gfx950) weighted grid->region aggregation engine (libwagg.so).
 *
 * This is the drop-in boundary for ONE path of ClimateImpactLab/climate_toolbox:
 *
 *   climate_toolbox/aggregations/aggregations.py
 *     :8-32    _reindex_spatial_data_to_regions      (pointwise gather, live branch :24-27)
 *     :35-84   _aggregate_reindexed_data_to_regions  (backup fill :73, grouped sums :78-79, divide)
 *     :87-124  weighted_aggregate_grid_to_regions    (:121-122 = the two above)
 *
 * The reference has no FFI; these entry points are what a ctypes binding placed at
 * aggregations.py:121-122 would call (INTEGRATION.md shows that binding).  Everything the
 * reference does with float LABELS (exact lat/lon match :27, backup fill :73, sorted unique
 * region labels :78) is resolved on the host in fp64 -- by the caller or by the host-side entry
 * points wagg_resolve_cells / wagg_backup_fill / wagg_factorize_* below; what reaches the
 * device-side entry points is the coded segment table (cell_idx, region_code, w_eff) and plain
 * device/host pointers.  The helpers either side of the path that have been folded in
 * (tas_poly, snyder_edd: transformations.py) are evaluated while the data is loaded
 * (wagg_apply_poly_*, wagg_apply_edd_*).  Since round 6 every apply is ONE entry point, wagg_apply(const
 * wagg_apply_desc *) at the end of this header; the per-combination wagg_*apply* functions below are wrappers
 * around it and stay for existing bindings.
 *
 * Conventions: every function returns 0 (WAGG_OK) or a negative wagg_status; nothing throws,
 * nothing calls exit(); wagg_last_error() gives the thread-local message of the last failure.
 * Plans are immutable after creation: concurrent wagg_apply_* on distinct streams is allowed,
 * create/destroy must not race with apply.  "_dev" pointers are HIP device pointers on the
 * current device; stream is a hipStream_t passed as void* (NULL = the null stream).
 * wagg_apply_* on device pointers is asynchronous on `stream`; the *_host_* forms block.
 */
#ifndef WAGG_H
#define WAGG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum wagg_status {
    WAGG_OK = 0,
    WAGG_EINVAL = -1,  /* bad argument (shape, index out of range, null pointer) */
    WAGG_EHIP = -2,    /* a HIP runtime call failed; see wagg_last_error() */
    WAGG_ENOMEM = -3,  /* host or device allocation failed */
    WAGG_ENODEV = -4,  /* no gfx950 device visible */
    WAGG_EUNSUPPORTED = -5,
    WAGG_EKEY = -6,    /* a segment label is absent from the grid (the reference's KeyError, S1) */
    WAGG_EINTERNAL = -7 /* an exception other than bad_alloc inside the library (never crosses the C boundary) */
} wagg_status;

/* memory layout of the flattened data matrix X and of the result */
#define WAGG_LAYOUT_TG 0 /* X[t*ldx + g]  -- (time, lat, lon) files: gridcell axis contiguous   */
#define WAGG_LAYOUT_GT 1 /* X[g*ldx + t]  -- (lat, lon, time) test fixture: time axis contiguous */
#define WAGG_OUT_TR 0    /* out[t*ldo + r]  (S10: (time, region)) */
#define WAGG_OUT_RT 1    /* out[r*ldo + t]  (S10 for the fixture layout: (region, time)) */

typedef struct wagg_plan wagg_plan;   /* sparse: coded segment table, region-grouped gather form */
typedef struct wagg_dense wagg_dense; /* dense (gridcell x region) fp32 weight matrix in HBM      */

typedef struct wagg_plan_info {
    int64_t nseg_in;     /* rows handed to wagg_plan_create                                      */
    int64_t nnz;         /* coalesced (cell, region) pairs kept (null label / NaN weight dropped) */
    int64_t n_groups;    /* region groups (one workgroup walks one group per time block)        */
    int64_t n_chunks;    /* LDS gather chunks (<= 256 unique cells each)                         */
    int64_t n_ucells;    /* sum over chunks of unique cells gathered per timestep                */
    int64_t n_giant;     /* regions with more unique cells than one chunk                       */
    int64_t n_empty;     /* regions with no kept segment (result 0/den)                          */
    int64_t G;
    int32_t R;
    int32_t lines;       /* bit 0: the plan also holds the whole-line chunking for fp32 (time, gridcell) data (lines of 32 cells);
                            bit 1: the one for fp64 data (lines of 16 cells); bit 2: the 64-cell one of fp64 degree days (four
                            lines of 16 cells: both fields of a chunk in one image).  All serve the loader/consumer kernel.       */
    int64_t n_lines128;  /* sum over chunks of distinct 128-byte lines (32 fp32 cells) their quads touch */
    int64_t n_sectors64; /* ... of distinct 64-byte sectors */
    /* whole-line chunkings (else 0): chunks of eight whole 128-byte lines (32 fp32 / 16 fp64 cells) of one column strip */
    int64_t n_partial_rows; /* (chunk, region) partial sums = rows of the partial buffer that combine_parts_kernel adds up */
    int64_t lines_chunks;   /* its chunks                                                                               */
    int64_t lines_ucells;   /* cells it fetches per timestep (every line whole, each exactly once)                      */
    int64_t lines_lines128; /* distinct 128-byte lines per timestep (= lines_ucells / 32 on grids of whole lines)       */
    int64_t n_partial_rows64; /* the same three for the fp64 chunking (16-cell lines)                                   */
    int64_t lines64_chunks;
    int64_t lines64_ucells;
} wagg_plan_info;

/* ---- process / device ------------------------------------------------------------------- */
int wagg_version(void);                 /* 10000*major + 100*minor + patch; 0.4.0: sized struct getters, wagg_apply_desc,
                                           wagg_host_stats has 18 fields */
int wagg_device_count(void);            /* number of visible HIP devices (0 if none), never <0  */
/* Time-axis sharding rule of the multi-GPU form (one process per GPU; climate_toolbox_amd/timeshard.py):
 * rank `rank` of `world` owns rows [*start, *stop) of T; the first T mod world ranks hold one row more. */
int wagg_shard_rows(int64_t T, int world, int rank, int64_t *start, int64_t *stop);
const char *wagg_last_error(void);      /* thread-local, never NULL                              */

/* ---- in-library kernel timing (HIP events on the stream the kernel is launched on) ----------- */
/* While enabled, every apply hands an event pair to the launch of its DOMINANT kernel (sparse: the
 * gather kernel; dense-family: the MFMA / entry-list kernel) -- hipExtLaunchKernel's startEvent / stopEvent, which
 * stamp the dispatch itself, so host time between two API calls is never booked as kernel time -- from a
 * ring of WAGG_PROFILE_SLOTS pairs, without any synchronisation.  wagg_profile_read blocks until the recorded
 * kernels have finished and returns their durations in milliseconds, oldest first.  Process-global and not
 * thread-safe: enable/read from the thread that issues the applies.  Not for use under stream capture.   */
#define WAGG_PROFILE_SLOTS 1024
int wagg_profile_enable(int on);        /* also resets the ring */
int wagg_profile_read(float *ms_out, int max_out, int *n_out);
/* The floor of that clock: median (and minimum) of what `n` (1..256) event pairs read for a kernel that does nothing,
 * launched on `stream` the same way.  rocprofv3's dispatch durations (profiles/) do not contain it; sub-millisecond
 * kernels therefore read a few percent longer through wagg_profile_read than there.  Blocks until the launches finished. */
int wagg_profile_event_overhead(void *stream, int n, float *median_ms, float *min_ms /* may be NULL */);

/* ---- label work ahead of the plan (host only) ------------------------------------- */
/* Exact-equality join of the segment table's lat/lon labels to the grid's (Dataset.sel without
 * method=, aggregations.py:27): cell_idx[i] = ilat*nlon + ilon (or ilon*nlat + ilat when lon_major).
 * WAGG_EKEY with *bad_row = first row whose label is absent; grid labels must be unique.          */
int wagg_resolve_cells(const double *lat, int64_t nlat, const double *lon, int64_t nlon,
                       const double *seg_lat, const double *seg_lon, int64_t nseg, int lon_major,
                       int32_t *cell_idx, int64_t *bad_row);
/* w_eff[i] = w[i] > 0 ? w[i] : backup[i]   (aggregations.py:73: NaN, 0 and negative take the row's backup) */
int wagg_backup_fill(const double *w, const double *backup, int64_t n, double *w_eff);
/* values[i] == from -> to   (aggregations.py:144: pix_cent_x 180.125 -> -179.875) */
int wagg_relabel(double *values, int64_t n, double from, double to);
/* sorted unique labels and per-row codes (-1 for null rows), aggregations.py:78 group keys (S3) */
int wagg_factorize_i64(const int64_t *labels, const uint8_t *isnull, int64_t n, int32_t *codes,
                       int64_t *uniq, int64_t *n_uniq);
int wagg_factorize_bytes(const char *buf, int64_t width, const uint8_t *isnull, int64_t n,
                         int32_t *codes, int64_t *uniq_rows, int64_t *n_uniq);

/* ---- sparse plan: replaces aggregations.py:24-27 + :64-71 (gather index, labels, weights) --- */
/* cell_idx[i]    flat grid cell of segment row i  = ilat*nlon + ilon            (0 <= . < G)
 * region_code[i] rank of the row's label among the sorted unique labels, -1 = null label (S3)
 * w_eff[i]       fp64 weight after the per-row backup fill of :73; NaN rows leave both sums
 * row_len        cells per grid row (nlon), 0 = unknown.  Known (and a whole number of 4-cell quads, G a whole number of
 *                rows), compact tables get a second, whole-line chunking next to the region-shaped one: chunks of eight
 *                whole 128-byte lines of one 32-cell column strip, every line fetched exactly once, regions cut by chunk
 *                borders summed from partial rows -- what fp32 (time, gridcell) applies use (their kernel is bound by
 *                line requests); everything else keeps the region-shaped chunks (fewer bytes)
 * flags          0, or WAGG_PLAN_* bits that pin the kernel form for this plan (tests, ablations);
 *                resolved here, once -- the library reads no environment variable
 * Host pointers; copied.  Duplicate (cell, region) rows add (S5).                              */
#define WAGG_PLAN_NO_LC 1     /* (time, gridcell) data: the persistent one-role kernel instead of the loader/consumer kernels */
#define WAGG_PLAN_NO_STREAM 2 /* every group through the chunk-walking kernel */
#define WAGG_PLAN_NO_LINES 4  /* region-shaped chunks only (rounds 1-2) even where the whole-line chunkings apply */
#define WAGG_PLAN_LC_MFMA 8   /* DIAGNOSTIC BUILD ONLY (libwagg_diag.so): fp32 loader/consumer kernel with dense-tile MFMA consumers
                                 (rounds 1-2) -- the one kernel with a bounded-spin barrier; libwagg.so refuses the flag with
                                 WAGG_EUNSUPPORTED */
#define WAGG_PLAN_SERIAL_BUILD 16 /* build the chunkings one after the other on the calling thread (the library starts three
                                 worker threads otherwise and falls back to exactly this when a thread cannot be started);
                                 the plan is the same bit for bit */
int wagg_plan_create(const int32_t *cell_idx, const int32_t *region_code, const double *w_eff,
                     int64_t nseg, int64_t G, int32_t R, int64_t row_len, int flags,
                     wagg_plan **out);
int wagg_plan_destroy(wagg_plan *plan);
int wagg_plan_get_info(const wagg_plan *plan, wagg_plan_info *info);
int wagg_plan_get_den(const wagg_plan *plan, double *den_host /* R values */); /* :79 */
/* Blocks until `stream` is idle, then reports whether any apply on this plan failed on the device.  No kernel of
 * libwagg.so can: none waits on another wave's progress (the diagnostic build's MFMA-consumer kernel can time out at its
 * consumer barrier: WAGG_EHIP, results incomplete -- there the same check also runs at the start of every later apply on the
 * plan and inside the *_host_ forms, so a device-side failure surfaces as a status, never as silent garbage).          */
int wagg_plan_status(const wagg_plan *plan, void *stream);

/* ---- apply: replaces aggregations.py:78-80 (and the gather of :27, fused) ------------------ */
/* out[t, r] = sum_i X[t, cell_i] * w_i / den[r]; NaN products count as 0 (skipna, S6); den == 0
 * gives IEEE NaN/inf (S7).  f32 accumulates in fp32 (declared deviation from S8, 1e-4 rel),
 * f64 in fp64 (1e-6 rel).  T may be any size >= 0.                                             */
int wagg_apply_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout,
                   float *out_dev, int64_t ldo, int out_layout, void *stream);
int wagg_apply_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout,
                   double *out_dev, int64_t ldo, int out_layout, void *stream);
/* Blocking forms on host buffers.  (time, gridcell) data with a (time, region) result is
 * streamed through the device in row blocks of ~256 MiB: the H2D copy of block i+1, the kernels of block i
 * and the return of block i-1's result run on three streams at once, and the device holds two blocks at a
 * time (host arrays of c4 / c5 size, 45-76 GB, need no device copy of the whole field).  flags:
 *   WAGG_HOST_PIN    page-lock the caller's arrays in place for the call (hipHostRegister / hipHostUnregister,
 *                    both statuses checked) so that the copy engines read and write them directly
 *   WAGG_HOST_WHOLE  one copy of the whole field, one apply, one copy back (the other layouts always do)
 * A caller array that is page-locked already (hipHostMalloc, the caller's own hipHostRegister, a framework's pinned
 * allocator) is used as it is, whatever its size -- the drop-in hands in result arrays from its pool of such blocks.
 * What is not page-locked -- arrays below 32 MiB (registration locks whole pages, which a small heap array
 * shares with unrelated objects), arrays whose registration the runtime refuses, and everything without
 * WAGG_HOST_PIN -- passes through the library's own page-locked staging pieces by CPU copy.  A pageable caller
 * pointer is never handed to a runtime copy: the runtime would page-lock the range on the fly and keep that
 * pin, keyed by address, beyond the call (docs/HISTORY.md (f)).  Every HIP status on this path is checked; failures
 * while resources are released are counted (wagg_host_stats) and fail the call.
 * Pitched arrays (ldx > G, ldo > R) are honoured: nothing behind the used cells of the last row is read and the
 * padding between result rows is not written.  The plan must live on the current device.
 * wagg_apply_host_* = the _ex form with WAGG_HOST_PIN | WAGG_HOST_LINES.  Measured on one MI355X (tools/host_path_timing.py):
 * the segment-table form is PCIe-bound either way (1.5 GB field: 51 ms whole from pageable memory, 29.8 ms pinned blocks =
 * 51 GB/s of X, 21.3 ms lines only; the 3 GB fp64 field: 59.2 / 33.5 ms);
 * a dense 1,369-row shard takes 602 ms whole and 523 ms in pinned blocks (the copies hide behind the 490 ms of
 * MFMA work).                                                                                         */
#define WAGG_HOST_PIN 1
#define WAGG_HOST_WHOLE 2
/*   WAGG_HOST_LINES  "lines only" (round 5; segment-table plans, (time, gridcell) data, (time, region) result, one device):
 *                    the table references a fraction of the grid -- the whole 128-byte lines its cells lie in are 64 % of a
 *                    c2-real fp32 row, 47 % of a fp64 row -- so up to twelve host threads (all but four of the CPUs the calling
 *                    thread is granted -- affinity mask and cgroup quota --, at least half of them) pack exactly those runs of every row side by side into a ring of
 *                    page-locked pieces, only the packed rows cross PCIe, and the kernel reads them through a second cell
 *                    table of the same plan.  X is read by the CPU and never page-locked; the result returns as WAGG_HOST_PIN
 *                    says.  Taken when the packed row is <= 80 % of the row, the field >= 64 MiB and the calling thread's
 *                    affinity mask and quota leave enough threads to pack faster than PCIe would move the whole rows (6 x the
 *                    packed fraction, e.g. 4 for c2-real fp32: >= 8 usable CPUs); otherwise (and when the ring is busy with
 *                    a concurrent call, or the threads cannot be started) the call runs as without the flag:
 *                    the flag permits, it never fails a call.  Same bits as every other form (the kernel sees the same cells
 *                    in the same order).  wagg_host_stats.lines_h2d_bytes counts the packed bytes sent.
 *                    Measured (tools/host_path_timing.py, c2-real T = 365): fp32 29.8 -> 21.3 ms per call (the C call alone
 *                    18.5 ms, of which 16.7 ms are the copy engine's), fp64 59.2 -> 33.5 ms; docs/HISTORY.md (f).              */
#define WAGG_HOST_LINES 4
/*   WAGG_HOST_LINES_WHOLE  with WAGG_HOST_LINES: pack the whole 128-byte lines as in round 5.  Without it (round 6, "quads
 *                    only") the packed row holds only the 16-byte QUADS that contain a referenced cell -- a whole-line chunk
 *                    fetches every quad of its lines, but the kernel reads only those; the others are pointed at a valid
 *                    dummy position -- whenever at least ten packing threads can be had (the runs are shorter: 116 instead of
 *                    455 bytes on c2-real; twelve threads stay ahead of PCIe, eight do not): 33.5 % of a c2-real fp32 row
 *                    instead of 63.6 %, c2-real T = 365 packing + copy 18.1 -> 10.2 ms (tools/host_granule_gonogo.sh,
 *                    profiles/r06_host_granule.txt).  Same kernel, same cells in the same order, and the kernel's choice between
 *                    its finite-data and general accumulation forms looks at REFERENCED quads only: the same bits as the device
 *                    apply for the plain, power and degree-day forms alike.                                              */
#define WAGG_HOST_LINES_WHOLE 8
int wagg_apply_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx,
                        int layout, float *out_host, int64_t ldo, int out_layout);
int wagg_apply_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx,
                        int layout, double *out_host, int64_t ldo, int out_layout);
int wagg_apply_host_ex_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx,
                           int layout, float *out_host, int64_t ldo, int out_layout, int flags);
int wagg_apply_host_ex_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx,
                           int layout, double *out_host, int64_t ldo, int out_layout, int flags);
/* Multi-device form of the above (SURVEY 8b `n_devices`, 8e "one process driving all devices"): plans[s] is a
 * replica of the same table created while device devices[s] was current.  The row blocks of the host field are
 * dealt round-robin to the n_devices pipelines (block i -> slot i mod n_devices, wagg_host_block_plan gives the
 * block size), each driven by its own host thread over its own PCIe link; every block's result lands directly in
 * the caller's rows, so there is no exchange between devices and no collective.  (time, gridcell) data and
 * (time, region) results only; flags: 0 or WAGG_HOST_PIN.  The same device may be listed more than once.       */
int wagg_apply_host_multi_f32(const wagg_plan *const *plans, const int *devices, int n_devices, const float *X_host,
                              int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags);
int wagg_apply_host_multi_f64(const wagg_plan *const *plans, const int *devices, int n_devices, const double *X_host,
                              int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags);
/* rows per block and number of blocks of a host field of T rows of row_bytes (= ldx * element size) bytes when a launch
 * handles `quantum` rows well (64: segment-table kernels; 368 / 176: fp32 / fp64 MFMA forms; 128 / 64: entry lists) */
int wagg_host_block_plan(int64_t T, int64_t row_bytes, int64_t quantum, int n_devices, int64_t *block_rows, int64_t *n_blocks);
/* what the host paths did since the last reset (process-wide counters): page-locks taken / refused / released /
 * whose release failed, HIP failures met while releasing resources, bytes moved through the staging pieces and
 * directly from / to page-locked caller memory                                                                      */
typedef struct wagg_host_stats {
    int64_t calls, blocks, registered, register_failed, unregistered, unregister_failed, cleanup_failed;
    int64_t staged_h2d_bytes, staged_d2h_bytes, direct_h2d_bytes, direct_d2h_bytes;
    int64_t lines_h2d_bytes;      /* packed rows of the lines-only path (WAGG_HOST_LINES) ...                              */
    int64_t lines_wait_pack_us;   /* ... time its pipeline thread waited for the packing threads (they are the bottleneck) */
    int64_t lines_wait_copy_us;   /* ... and for the copy engine to hand a ring piece back (PCIe is the bottleneck)        */
    int64_t blocks_retired;       /* calls that ran < 60 % of the best rate (bytes over PCIe / wall time) seen for calls of their
                                     shape: their own device blocks went back to the driver instead of the pool               */
    int64_t found_page_locked;    /* caller arrays that were page-locked already (used as they are, no registration)          */
    int64_t watched_calls;        /* calls the copy-rate watch judged (>= 256 MiB moved from page-locked memory)               */
    int64_t last_rate_permille;   /* the LAST call's copy rate over the best seen for its shape, x 1000 (1000 = it set or met the
                                     record; below 600 the call retired its blocks); 0 = the last call was not judged.  A state,
                                     not a counter: `reset` leaves it                                                          */
} wagg_host_stats;
int wagg_host_stats_read(wagg_host_stats *out, int reset);     /* = the _sized form with sizeof(wagg_host_stats) of THIS header */

/* ---- structs that cross the boundary by layout --------------------------------------------------------------------- */
/* wagg_plan_info, wagg_dense_info, wagg_host_stats and wagg_apply_desc only ever GROW AT THE END.  A binding that mirrors
 * them by hand (ctypes, cffi, cgo) asks the library for its sizes and hands its OWN size to the *_sized getters: the library
 * fills min(size, its own size) bytes and zeroes what the caller has beyond that, so a binding built against an older or a
 * newer header is memory-safe without a version check (fields it does not know are not written; fields the library does
 * not know read 0).  wagg_struct_ordinals writes, into every scalar field in declaration order, its 1-based ordinal (array
 * elements count one each): a binding's self-test that its field ORDER and TYPES are the library's
 * (tests/test_abi.py does exactly that for climate_toolbox_amd/_lib.py).                                                 */
#define WAGG_STRUCT_PLAN_INFO 0
#define WAGG_STRUCT_DENSE_INFO 1
#define WAGG_STRUCT_HOST_STATS 2
#define WAGG_STRUCT_APPLY_DESC 3
int wagg_struct_size(int which);                                   /* sizeof in this library; WAGG_EINVAL for an unknown id */
int wagg_struct_ordinals(int which, void *out, uint64_t size);     /* fills min(size, sizeof) bytes as described above      */
int wagg_plan_get_info_sized(const wagg_plan *plan, void *info, uint64_t size);
int wagg_host_stats_read_sized(void *out, uint64_t size, int reset);

/* ---- fused grid-level transform ---------------------------------------------- */
/* Replaces  tas_poly  (climate_toolbox/transformations/transformations.py:160-208, the arithmetic
 * at :188  ``(ds.tas - 273.15) ** power``) followed by the aggregation above, for the powers
 * p = pow_first .. pow_first + n_pow - 1 in ONE call:
 *   out_p[t, r] = sum_i (X[t, cell_i] + offset)^p * w_eff[i] / den[r].
 * The transform is evaluated in the data type while X is loaded (NaN stays NaN and is skipped
 * like any NaN product, S6), so the transformed grids are never written to memory;
 * (time, gridcell) data, fp32 and fp64, is read from HBM once per four consecutive powers.  The i-th
 * power is stored at out_dev + i * out_pstride (elements) with leading dimension ldo.  Powers must
 * lie in [1, 16]. */
int wagg_apply_poly_f32(const wagg_plan *plan, const float *X_dev, int64_t T, int64_t ldx, int layout,
                        double offset, int pow_first, int n_pow, float *out_dev, int64_t ldo,
                        int64_t out_pstride, int out_layout, void *stream);
int wagg_apply_poly_f64(const wagg_plan *plan, const double *X_dev, int64_t T, int64_t ldx, int layout,
                        double offset, int pow_first, int n_pow, double *out_dev, int64_t ldo,
                        int64_t out_pstride, int out_layout, void *stream);
/* The same for a HOST-resident (time, gridcell) field, (time, region) results in host memory (round 5): the field goes
 * through the row-block pipeline of wagg_apply_host_ex_* once -- every block is raised to its powers on the device, four
 * per pass -- and plane i of the result (power pow_first + i) lands at out_host + i * out_pstride (elements, >= T * ldo)
 * with leading dimension ldo.  flags: WAGG_HOST_PIN, WAGG_HOST_LINES (as there; WAGG_HOST_LINES sends only the 128-byte
 * lines the table references).  This is the reference's tas_poly-then-aggregate on the arrays its callers hold
 * (transformations.py:188 + aggregations.py:87): c2-real, four powers, 1.5 GB field: docs/HISTORY.md (f).  Blocking. */
int wagg_apply_poly_host_f32(const wagg_plan *plan, const float *X_host, int64_t T, int64_t ldx, double offset, int pow_first,
                             int n_pow, float *out_host, int64_t ldo, int64_t out_pstride, int flags);
int wagg_apply_poly_host_f64(const wagg_plan *plan, const double *X_host, int64_t T, int64_t ldx, double offset, int pow_first,
                             int n_pow, double *out_host, int64_t ldo, int64_t out_pstride, int flags);

/* Replaces  snyder_edd  (transformations.py:7-93: Snyder exceedance degree days of the daily
 * (tasmin, tasmax) pair at a threshold e, the nested xr.where of :75-87) followed by the
 * aggregation, for n_thr thresholds in one call.  Both fields have the layout/ldx of X above and
 * are shifted by `offset` first (-273.15 for Kelvin files; the thresholds are in the shifted
 * unit).  Threshold i is stored at out_dev + i * out_pstride.  snyder_gdd (:96-144) is the
 * difference of two of these outputs (the aggregation is linear).                               */
int wagg_apply_edd_f32(const wagg_plan *plan, const float *tasmin_dev, const float *tasmax_dev, int64_t T,
                       int64_t ldx, int layout, double offset, const double *thresholds, int n_thr,
                       float *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
int wagg_apply_edd_f64(const wagg_plan *plan, const double *tasmin_dev, const double *tasmax_dev, int64_t T,
                       int64_t ldx, int layout, double offset, const double *thresholds, int n_thr,
                       double *out_dev, int64_t ldo, int64_t out_pstride, int out_layout, void *stream);
/* The same for two HOST-resident (time, gridcell) fields, (time, region) results in host memory (round 5): tasmin and tasmax go
 * through the row-block pipeline of wagg_apply_host_ex_* together, every block's degree days are evaluated on the device (four
 * thresholds per pass over the block) and plane i (threshold i) lands at out_host + i * out_pstride (elements, >= T * ldo).
 * flags: WAGG_HOST_PIN, WAGG_HOST_LINES (only the lines of both fields the table references cross PCIe).  Blocking.  This is
 * snyder_edd-then-aggregate (transformations.py:7-93 + aggregations.py:87) on the arrays the reference's callers hold.      */
int wagg_apply_edd_host_f32(const wagg_plan *plan, const float *tasmin_host, const float *tasmax_host, int64_t T, int64_t ldx,
                            double offset, const double *thresholds, int n_thr, float *out_host, int64_t ldo,
                            int64_t out_pstride, int flags);
int wagg_apply_edd_host_f64(const wagg_plan *plan, const double *tasmin_host, const double *tasmax_host, int64_t T, int64_t ldx,
                            double offset, const double *thresholds, int n_thr, double *out_host, int64_t ldo,
                            int64_t out_pstride, int flags);

/* ---- materialised grid-level transforms (device buffers, elementwise) ------------------------ */
/* What ``.values`` of a lazily transformed variable returns: out[i] = (X[i] + offset)^power
 * (transformations.py:188), out[i] = sum_k coefs[k] * snyder_edd(tasmin[i] + offset, tasmax[i] + offset,
 * thresholds[k]) (transformations.py:64-87; two terms with coefs +1/-1 = snyder_gdd, :138-140), with the
 * device functions the fused aggregation kernels use.  n_terms <= 8.  wagg_any_less_* is the check of
 * transformations.py:62 (``assert not (tasmax < tasmin).any()``): *result = 1 iff some a[i] < b[i]; blocks. */
int wagg_transform_poly_f32(const float *X_dev, int64_t n, double offset, int power, float *out_dev, void *stream);
int wagg_transform_poly_f64(const double *X_dev, int64_t n, double offset, int power, double *out_dev, void *stream);
int wagg_transform_edd_f32(const float *tasmin_dev, const float *tasmax_dev, int64_t n, double offset,
                           const double *coefs, const double *thresholds, int n_terms, float *out_dev, void *stream);
int wagg_transform_edd_f64(const double *tasmin_dev, const double *tasmax_dev, int64_t n, double offset,
                           const double *coefs, const double *thresholds, int n_terms, double *out_dev, void *stream);
/* out[i] = sum_k coefs[k] * planes[k * plane_stride + i], n_planes <= 8: several aggregated planes combined into one
 * (snyder_gdd = EDD(threshold_low) - EDD(threshold_high), transformations.py:138-140, after the aggregation: it is linear) */
int wagg_combine_planes_f32(const float *planes_dev, int n_planes, int64_t plane_stride, const double *coefs, int64_t n,
                            float *out_dev, void *stream);
int wagg_combine_planes_f64(const double *planes_dev, int n_planes, int64_t plane_stride, const double *coefs, int64_t n,
                            double *out_dev, void *stream);
/* Data movement of device-resident fields in front of the path.  wagg_take_axis: dst[o][i][:] = src[o][idx[i]][:] for a
 * contiguous array seen as (outer, n_src, inner_bytes) -- the 29-February drop along time (utils.py:60-74) and the lon
 * re-ordering (utils.py:33-40); idx_dev is an int64 device array, inner_bytes a multiple of 4.  wagg_relayout_*: dst
 * (contiguous, shape[0..ndim)) <- src with element strides src_strides (any permutation of up to 6 dims: the transpose of a
 * field whose lat/lon axes are not adjacent).                                                                            */
int wagg_take_axis(const void *src_dev, int64_t outer, int64_t n_src, int64_t inner_bytes, const int64_t *idx_dev,
                   int64_t n_idx, void *dst_dev, void *stream);
int wagg_relayout_f32(const float *src_dev, int ndim, const int64_t *shape, const int64_t *src_strides, float *dst_dev, void *stream);
int wagg_relayout_f64(const double *src_dev, int ndim, const int64_t *shape, const int64_t *src_strides, double *dst_dev, void *stream);
/* a device field of another element type as a contiguous fp64 array (the reference's own promotion: any dtype times float64
 * weights is float64, S8), strided source as for wagg_relayout_* with strides in elements of the source type */
#define WAGG_T_F16 0
#define WAGG_T_BF16 1
#define WAGG_T_I8 2
#define WAGG_T_U8 3
#define WAGG_T_I16 4
#define WAGG_T_I32 5
#define WAGG_T_I64 6
#define WAGG_T_F32 7
#define WAGG_T_F64 8
int wagg_relayout_to_f64(const void *src_dev, int src_type, int ndim, const int64_t *shape, const int64_t *src_strides,
                         double *dst_dev, void *stream);
/* One blocking copy of a contiguous host buffer to the device, the way the library uploads its own operands: buffers of
 * >= 32 MiB are page-locked in place for the copy (one DMA; the lock goes before the call returns), smaller ones pass through
 * the library's page-locked staging pieces -- a pageable pointer is never handed to a runtime copy, which pins the range on
 * the fly and keeps that pin, keyed by address, beyond the call.  Measured (tools/host_edd_timing.py, 1.5 GB): 32.4 ms =
 * 47 GB/s lock + copy + unlock, the same as a runtime copy of an array the runtime has pinned in an earlier call (32.8 ms;
 * its first copy of a new array is slower).  The drop-in uploads the host-resident fields the row-block pipelines do not
 * serve (degree days, (gridcell, time) data, dense-family plans with transforms) through it.                                */
int wagg_upload(void *dst_dev, const void *src_host, int64_t bytes);
int wagg_any_less_f32(const float *a_dev, const float *b_dev, int64_t n, int *result, void *stream);
int wagg_any_less_f64(const double *a_dev, const double *b_dev, int64_t n, int *result, void *stream);

/* ---- materialised gather: what _reindex_spatial_data_to_regions returns (:27) -------------- */
/* out[t, i] = X[t, cell_idx[i]] in out_layout (WAGG_OUT_TR: out[t*ldo+i], RT: out[i*ldo+t]).   */
int wagg_gather_f32(const float *X_dev, int64_t T, int64_t ldx, int layout,
                    const int32_t *cell_idx_dev, int64_t nseg, float *out_dev, int64_t ldo,
                    int out_layout, void *stream);
int wagg_gather_f64(const double *X_dev, int64_t T, int64_t ldx, int layout,
                    const int32_t *cell_idx_dev, int64_t nseg, double *out_dev, int64_t ldo,
                    int out_layout, void *stream);

/* ---- dense weights: W[g, r] as an fp32 matrix resident in HBM, MFMA contraction ------------- */
/* synth: W[g,r] = hash_u01(g*R + r, seed) in [0,1), generated on device (never on host); this is
 * the c2-dense benchmark operand (1,036,800 x 24,378 = 101 GB).                                */
int wagg_dense_create_synth(int64_t G, int32_t R, uint32_t seed, wagg_dense **out);
/* the same with only a fraction `fill` of the entries non-zero, at uniformly random positions
 * (kept where hash_u01(g*R + r, seed ^ 0x9e3779b9) < fill): c5's "uniform-random columns"
 * structure at fill = 0.01 -- no tile of W is empty, yet 99 % of every tile is.                 */
int wagg_dense_create_synth_sparse(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out);
/* (fill < 10 %: built directly as entry lists, WAGG_FORM_ENTRIES -- c5 at fill = 0.01; above that the
 * full matrix is generated; the same holds for wagg_dense_create_synth_f64)                        */
/* from a host row-major (G, R) fp32 matrix (small cases / tests) */
int wagg_dense_create_host(const float *W_host, int64_t G, int32_t R, wagg_dense **out);
/* from a sparse plan's coded table (weights whose regions are scattered over the grid).  The caller brings ONE kind of
 * table, like the reference (aggregations.py:64-73); which of the three device forms it takes is the library's choice,
 * made from the structure it finds (a census of the occupied (32-cell x 256-region) tiles and the number of distinct
 * pairs) by the estimated time per row of X of each form at its measured rate (tools/form_crossover.py, docs/HISTORY.md (b)):
 * only the non-empty tiles on the MFMA kernel (WAGG_FORM_TILES: c5's block-local weights), entry lists on the vector ALU
 * (WAGG_FORM_ENTRIES: c5's uniformly random 1 %), or the full matrix (WAGG_FORM_FULL).
 * flags: WAGG_DENSE_FORM_AUTO (0), or WAGG_DENSE_FORCE_* to pin the form -- for measurements and tests; results agree
 * between the forms to rounding.  Up to 2^31 - 1 rows.                                              */
#define WAGG_DENSE_FORM_AUTO 0
#define WAGG_DENSE_FORCE_FULL 1
#define WAGG_DENSE_FORCE_TILES 2
#define WAGG_DENSE_FORCE_ENTRIES 3
/* ... or-ed with WAGG_DENSE_GENERAL_SORT: sort the table with the general radix sort even when it is a CSR table with
 * ascending columns, which the builder otherwise puts in order with one stable pass per chunk of 128 cells (the plans
 * are bit-identical; for tests and timings) */
#define WAGG_DENSE_GENERAL_SORT 16
int wagg_dense_create_from_segments(const int32_t *cell_idx, const int32_t *region_code,
                                    const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                    int flags, wagg_dense **out);
/* The same table in CSR form -- BASELINE configs[4] "sparse CSR weights (<= 1 % nnz)": rows = grid cells (rowptr[G + 1],
 * rowptr[0] = 0, non-decreasing), col = region codes (-1 = null label, dropped), val = fp64 weights (NaN: dropped);
 * columns of a row in any order, repeated (cell, region) pairs add in the order they are stored (S5).
 * Both constructors only UPLOAD the host arrays (page-locked in place for the copy when they are large, through the
 * library's staging pieces otherwise: no host copy of the table is made); sorting into the kernels' order, coalescing,
 * the denominators (aggregations.py:79), the tile census that picks the form, and the packing run on the device
 * (stable radix sort, fixed summation orders: the same table gives the same plan, bit for bit; csrc/wagg_build.hip) --
 * on a stream of the build's own, out of one scratch arena: a build never waits for, nor stops, the streams other plans
 * are applied on.  Up to 2^31 - 1 entries; a configs[4]-sized table (2.53e8 entries, 3 GB) needs ~12 GB of device scratch
 * while it is built (48 bytes per entry) plus 16 bytes per distinct pair.  wagg_dense_info.build_s / build_upload_s
 * report where the time went.                                                                                         */
int wagg_dense_create_from_csr(const int64_t *rowptr, const int32_t *col, const double *val, int64_t G, int32_t R,
                               int flags, wagg_dense **out);
/* synthetic block-local weights (SURVEY 8d, c5 "each 64-cell run touches <= 256 regions"): run j of
 * 64 cells touches the 256 regions of column tile (97 j) mod ceil(R/256); inside, W[g,r] =
 * hash_u01(g*R + r, seed) where hash_u01(g*R + r, seed ^ 0x9e3779b9) < fill, else 0.  Generated on
 * the device in tile-sparse form (benchmark operand; the oracle regenerates it from the hashes).  */
int wagg_dense_create_synth_blocklocal(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out);
#define WAGG_FORM_FULL 0     /* every (32-cell x 256-region) tile of W stored, fp32 MFMA contraction            */
#define WAGG_FORM_TILES 1    /* only the non-empty tiles ("tile-sparse": block-local weights), same MFMA kernel */
#define WAGG_FORM_ENTRIES 2  /* no matrix: per-wave (cell, region, weight) lists, vector-ALU kernel (scattered,
                                sparse weights: a few per cent of non-zeros spread over (almost) every tile), fp32 or fp64 */
/* The *_f64 constructors build the same forms with fp64 weights for fp64 data (the reference's own
 * arithmetic type, aggregations.py:73-80): tiles are (16-cell x 256-region), the contraction runs on
 * v_mfma_f64_16x16x4_f64 (wagg_dense_apply_f64); the entry-list form keeps 64-bit weights (80 bytes per
 * 8 entries instead of 48) and accumulates with v_fma_f64, one timestep per lane.  A plan serves one
 * element type.                                                                                      */
int wagg_dense_create_synth_f64(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out);
int wagg_dense_create_host_f64(const double *W_host, int64_t G, int32_t R, wagg_dense **out);
int wagg_dense_create_from_segments_f64(const int32_t *cell_idx, const int32_t *region_code,
                                        const double *w_eff, int64_t nseg, int64_t G, int32_t R,
                                        int flags, wagg_dense **out);
int wagg_dense_create_from_csr_f64(const int64_t *rowptr, const int32_t *col, const double *val, int64_t G, int32_t R,
                                   int flags, wagg_dense **out);
int wagg_dense_create_synth_blocklocal_f64(int64_t G, int32_t R, uint32_t seed, double fill, wagg_dense **out);
typedef struct wagg_dense_info {
    int64_t G, n_tiles, w_bytes;   /* stored (32 x 256) tiles; bytes of W (or of the entry lists) in HBM */
    int32_t R, n_kt, n_nt, tiled;  /* k tiles, column tiles, 1 = tile-sparse form */
    int32_t form, elem_bytes;      /* WAGG_FORM_*; 4 = fp32 plan, 8 = fp64 plan */
    int64_t nnz;                   /* kept distinct (cell, region) pairs: entry-list form and every plan built from a
                                      caller's table (from_segments / from_csr); else -1 */
    double build_s, build_upload_s; /* plans built from a caller's table: wall seconds of the constructor, and the part of
                                      them spent moving the table to the device (else 0) */
    double est_row_s[3];           /* plans built from a caller's table: what the form choice went by -- estimated seconds per
                                      row of X in the forms WAGG_FORM_FULL / _TILES / _ENTRIES (else 0) */
    int64_t walked_entries;        /* ... and the entries the entry-list kernel would walk (16 x the longest per-wave list of
                                      every (region block, chunk) item): = nnz for evenly spread weights */
    int32_t one_pass_sort, reserved0; /* 1 = the table was put in key order by the one-pass chunk partition (CSR, ascending
                                      columns, nothing dropped), 0 = by the general radix sort */
} wagg_dense_info;
int wagg_dense_get_info(const wagg_dense *d, wagg_dense_info *info);
int wagg_dense_get_info_sized(const wagg_dense *d, void *info, uint64_t size);   /* see "structs that cross the boundary by layout" */
/* Device blocks and streams the library needs for the length of one call -- the arena of a table -> plan build (11 GB for
 * a 2.5e8-row table), the block buffers and streams of a host-resident apply, a plan's per-stream staging -- come from a
 * per-device pool and return to it (csrc/wagg_scratch.hip: giving 11 GB back to the driver and asking again costs a wait
 * of seconds every dozen builds).  At most 1/16 of a device's memory is kept.  This frees it (the package's
 * clear_caches() calls it); an allocation of the library that runs out of device memory does so by itself before it
 * tries once more. */
int wagg_release_scratch(void);
int64_t wagg_scratch_bytes(void);  /* device bytes kept right now, all devices */
/* A second plan with the same weights on `device` (may be the device `src` lives on): the finished plan's device arrays --
 * packed W with its tile tables, or the entry lists, and the denominators -- are copied device to device (hipMemcpyPeer:
 * xGMI between two GPUs), nothing is uploaded or sorted again and no host copy of the caller's table is needed.  The
 * clone owns its workspaces, so the two plans apply independently (wagg_apply_host_multi_* wants one plan per pipeline).
 * `src` is only read: applies on it may be in flight.  The calling thread's current device is left as it was. */
int wagg_dense_clone(const wagg_dense *src, int device, wagg_dense **out);
int wagg_dense_destroy(wagg_dense *d);
int wagg_dense_get_den(const wagg_dense *d, double *den_host /* R values */);
/* out[t, r] = sum_g nan0(X[t,g]) * W[g,r] / den[r], fp32 MFMA (v_mfma_f32_16x16x4_f32).
 * ksplit = 0 picks the k-slice count; otherwise a multiple of 8.  The plan owns the packed copy of
 * X and the partial-sum slabs of an apply (hence the non-const handle): applies on ONE dense plan
 * must be ordered on one stream; different plans are independent.
 * Tile-sparse plans (and full ones with two or more tall row blocks) read X where it lies when its rows
 * are 16-byte aligned and G is a whole number of k tiles (32 floats / 16 doubles): no packed copy is written; a field with NaN
 * or +-inf in it is noticed on the device (non-finite numerators) and redone through the packed
 * pass on the same stream, so the result and wagg_dense_saw_inf are the same either way; after that
 * the plan keeps to the packed pass.                                                              */
int wagg_dense_apply_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx,
                         float *out_dev, int64_t ldo, int ksplit, void *stream);
/* fp64 data on an fp64 plan: out = sum_g nan0(X[t,g]) * W[g,r] / den[r] on v_mfma_f64_16x16x4_f64 */
int wagg_dense_apply_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx,
                         double *out_dev, int64_t ldo, int ksplit, void *stream);
/* the same contraction of (X + offset)^power (tas_poly, transformations.py:188) and of
 * snyder_edd(tasmin + offset, tasmax + offset, threshold) (transformations.py:64-87): the transform is
 * evaluated while X is packed, the transformed grid is never written anywhere                    */
int wagg_dense_apply_poly_f32(wagg_dense *d, const float *X_dev, int64_t T, int64_t ldx, double offset,
                              int power, float *out_dev, int64_t ldo, int ksplit, void *stream);
int wagg_dense_apply_edd_f32(wagg_dense *d, const float *tasmin_dev, const float *tasmax_dev, int64_t T,
                             int64_t ldx, double offset, double threshold, float *out_dev, int64_t ldo,
                             int ksplit, void *stream);
int wagg_dense_apply_poly_f64(wagg_dense *d, const double *X_dev, int64_t T, int64_t ldx, double offset,
                              int power, double *out_dev, int64_t ldo, int ksplit, void *stream);
int wagg_dense_apply_edd_f64(wagg_dense *d, const double *tasmin_dev, const double *tasmax_dev, int64_t T,
                             int64_t ldx, double offset, double threshold, double *out_dev, int64_t ldo,
                             int ksplit, void *stream);
/* host-resident (time, gridcell) data through a dense-family plan, row-block pipeline as wagg_apply_host_ex_* */
int wagg_dense_apply_host_f32(wagg_dense *d, const float *X_host, int64_t T, int64_t ldx,
                              float *out_host, int64_t ldo, int flags);
int wagg_dense_apply_host_f64(wagg_dense *d, const double *X_host, int64_t T, int64_t ldx,
                              double *out_host, int64_t ldo, int flags);
/* ... and over several devices, one replica of the plan per device (as wagg_apply_host_multi_*; a dense-family plan
 * owns its workspaces, so every slot needs a replica of its own).  The +-inf notes of all replicas (wagg_dense_saw_inf)
 * are collected in plans[0] before the call returns and cleared in the others.                                       */
int wagg_dense_apply_host_multi_f32(wagg_dense *const *plans, const int *devices, int n_devices, const float *X_host,
                                    int64_t T, int64_t ldx, float *out_host, int64_t ldo, int flags);
int wagg_dense_apply_host_multi_f64(wagg_dense *const *plans, const int *devices, int n_devices, const double *X_host,
                                    int64_t T, int64_t ldx, double *out_host, int64_t ldo, int flags);
/* The MFMA forms multiply every (cell, region) pair of a stored tile, so +-inf in the (transformed)
 * data turns the zero weights of regions that do not own the cell into NaN, where the reference and
 * the segment-table form confine it to the owning regions (S6).  The pack stage notes such data:
 * this call synchronises `stream`, returns in *saw whether any apply since the last call met +-inf
 * and clears the note; the caller then repeats that field through a wagg_plan.  (The entry-list
 * form multiplies real pairs only and never sets it.)                                            */
int wagg_dense_saw_inf(wagg_dense *d, void *stream, int *saw);

/* ---- synthetic data generators (device side, shared with the CPU oracle bit for bit) ------- */
/* The c5 weight tables of wagg_dense_create_synth_sparse (blocklocal = 0) / _synth_blocklocal (1) as a caller would hand
 * them in: CSR on the HOST (rows = cells, columns ascending, fp64 values = the fp32 hashes).  Generated on the device and
 * copied out; call with col_host = val_host = NULL first to get rowptr and *nnz_out, then with arrays of that capacity.  */
int wagg_synth_table_csr(int64_t G, int32_t R, uint32_t seed, double fill, int blocklocal, int64_t *rowptr_host,
                         int32_t *col_host, double *val_host, int64_t capacity, int64_t *nnz_out);
/* X[t*ldx + g] = base + amp * (hash_u01(t*G + g, seed) - 0.5)                                   */
int wagg_synth_field_f32(float *X_dev, int64_t T, int64_t G, int64_t ldx, uint32_t seed,
                         float base, float amp, void *stream);
int wagg_synth_field_f64(double *X_dev, int64_t T, int64_t G, int64_t ldx, uint32_t seed,
                         double base, double amp, void *stream);

/* ---- time-axis shards on several devices, ONE process, device-resident data (SURVEY 8b n_devices, 8e) ----------------
 * Output row t depends on input row t only: shard i (rows_i rows of the job, resident on devices[i]) goes through plans[i]
 * on devices[i]; its (rows_i x R) block lands directly in rows [sum_{j<i} rows_j, ...) of out_root on devices[root].  No
 * exchange during compute; one at the end, by `transport`:
 *   WAGG_GATHER_RCCL  grouped ncclSend / ncclRecv between the communicators of ncclCommInitAll: every shard -> root on its
 *                     own xGMI link (a direct gather, not a ring).  librccl is looked up at run time, not linked: the copy
 *                     the process already holds (torch's) if there is one.  Every device once; out_root rows contiguous.
 *   WAGG_GATHER_PEER  peer copies (hipMemcpyPeerAsync / hipMemcpy2DAsync) on the shard's own stream: needs no library, takes
 *                     pitched results, and allows a device to be listed twice (several shards on one GPU)
 *   WAGG_GATHER_AUTO  RCCL when it can be loaded, the devices are distinct and n > 1; else peer copies
 * A group owns one stream per shard and the shards' staging blocks; the apply calls are BLOCKING (the result is complete on
 * return) and restore the caller's current device.  The plans are replicas of one table, one per shard (a dense-family plan
 * owns its workspaces; a segment-table plan may serve several shards of its own device).  The process-per-GPU form of the
 * same job is torch.distributed + RCCL (climate_toolbox_amd/timeshard.py); for host-resident fields see
 * wagg_*_apply_host_multi_*.  Validated on hardware with several shards on ONE device (peer copies) and with RCCL at n = 1;
 * unmeasured at n > 1 devices (gpurun exposes one GPU).                                                                  */
typedef struct wagg_shard_group wagg_shard_group;
#define WAGG_GATHER_AUTO 0
#define WAGG_GATHER_RCCL 1
#define WAGG_GATHER_PEER 2
int wagg_shard_group_create(const int *devices, int n, int transport, wagg_shard_group **out);
int wagg_shard_group_destroy(wagg_shard_group *g);
int wagg_shard_group_info(const wagg_shard_group *g, int *n_shards, int *transport /* RCCL or PEER: what the group uses */);
int wagg_apply_sharded_f32(wagg_shard_group *g, const wagg_plan *const *plans, const float *const *X_dev, const int64_t *rows,
                           int64_t ldx, float *out_root, int64_t ldo, int root);
int wagg_apply_sharded_f64(wagg_shard_group *g, const wagg_plan *const *plans, const double *const *X_dev, const int64_t *rows,
                           int64_t ldx, double *out_root, int64_t ldo, int root);
int wagg_dense_apply_sharded_f32(wagg_shard_group *g, wagg_dense *const *plans, const float *const *X_dev, const int64_t *rows,
                                 int64_t ldx, float *out_root, int64_t ldo, int root);
int wagg_dense_apply_sharded_f64(wagg_shard_group *g, wagg_dense *const *plans, const double *const *X_dev, const int64_t *rows,
                                 int64_t ldx, double *out_root, int64_t ldo, int root);

/* ---- one entry point for every apply (round 6) ------------------------------------------------------------------- */
/* Every wagg_*apply* function above is a three-line wrapper that fills this descriptor and calls wagg_apply(): a new
 * transform or a new kind of source is a new VALUE of a field here, not eight new symbols.  New bindings should call
 * wagg_apply() only (INTEGRATION.md section 2 shows both forms; climate_toolbox_amd/engine.py uses nothing else).
 *
 *   struct_size  sizeof(wagg_apply_desc) of the header the caller was built against (the struct only grows at the end;
 *                fields beyond the caller's size read 0 = "not given")
 *   plan_kind    WAGG_PLAN_SEGMENT: `plan` is a wagg_plan*;  WAGG_PLAN_DENSE: a wagg_dense*
 *   elem         WAGG_T_F32 / WAGG_T_F64: element type of x, x2 and out (a dense-family plan must have been built for it)
 *   source       where the field lies and who moves it:
 *                  WAGG_SRC_DEVICE      device pointers, asynchronous on `stream`             (wagg_apply_f32 ...)
 *                  WAGG_SRC_HOST        host arrays through the row-block pipeline, blocking; `flags` = WAGG_HOST_*
 *                  WAGG_SRC_HOST_MULTI  the same over `n_plans` plan replicas on `devices[]`; `plan` is an ARRAY of handles
 *                  WAGG_SRC_SHARDED     `plan` = array of n_plans handles, `x` = array of n_plans device pointers (shard i on
 *                                       the device of plan i), rows[i] rows each; results gathered into `out` on shard `root`
 *                                       of `group` (a wagg_shard_group*); blocking
 *   transform    WAGG_XF_NONE; WAGG_XF_POLY: (x + offset)^p for p = pow_first .. pow_first + n_pow - 1, plane i at
 *                out + i * out_pstride; WAGG_XF_EDD: Snyder degree days of (x = tasmin, x2 = tasmax) + offset at
 *                thresholds[0 .. n_thr), plane i at out + i * out_pstride.  (Dense-family plans: one power / one threshold
 *                per call.)
 *   T, ldx, layout / out, ldo, out_layout   as in wagg_apply_f32 (dense-family plans: WAGG_LAYOUT_TG / WAGG_OUT_TR only)
 *   ksplit       dense-family plans: k-slice count, 0 = the library's choice
 * A combination the library has no kernel path for returns WAGG_EUNSUPPORTED with a message that names it; nothing is
 * emulated.                                                                                                            */
#define WAGG_PLAN_SEGMENT 0
#define WAGG_PLAN_DENSE 1
#define WAGG_SRC_DEVICE 0
#define WAGG_SRC_HOST 1
#define WAGG_SRC_HOST_MULTI 2
#define WAGG_SRC_SHARDED 3
#define WAGG_XF_NONE 0
#define WAGG_XF_POLY 1
#define WAGG_XF_EDD 2
typedef struct wagg_apply_desc {
    uint64_t struct_size;
    int32_t plan_kind, elem, source, transform;
    const void *plan;
    const void *x, *x2;
    void *out;
    int64_t T, ldx, ldo, out_pstride;
    int32_t layout, out_layout;
    double offset;
    int32_t pow_first, n_pow;
    const double *thresholds;
    int32_t n_thr, flags;
    int32_t ksplit, n_plans;
    const int32_t *devices;
    const int64_t *rows;
    void *group;
    int32_t root, reserved0;
    void *stream;
} wagg_apply_desc;
int wagg_apply(const wagg_apply_desc *desc);

#ifdef __cplusplus
}
#endif
#endif /* WAGG_H */
