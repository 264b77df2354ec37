// Host-resident (time, gridcell) data: the row-block pipeline behind wagg_apply_host_* / wagg_dense_apply_host_* and
// their multi-device forms (SURVEY 8f-4, 8b `n_devices`, 8e "one process driving all devices").
//
// What the caller's memory meets, and nothing else:
//   * the CPU (memcpy into / out of the library's own page-locked staging buffers), or
//   * an EXPLICIT page-lock of the array for the duration of the call (HostPin: hipHostRegister / hipHostUnregister,
//     every status kept), after which the copy engines read and write it in place.
// A pageable caller pointer is never handed to a runtime copy (hipMemcpy / hipMemcpyAsync): the runtime would pin the
// range on the fly and keep that pin, keyed by address, beyond the call.
//
// Every HIP status on this path is checked.  Failures inside destructors (stream / event / buffer release, a failed
// unregistration) cannot be returned from there: they are counted (wagg_host_stats.cleanup_failed /
// .unregister_failed), leave their text in wagg_last_error() if the call had none, and the entry points turn a non-zero
// count of their own call into WAGG_EHIP.
#pragma once
#include <atomic>
#include <functional>
#include <thread>

#include "wagg_common.h"

namespace wagg {

struct HostStats {
    std::atomic<int64_t> calls{0}, registered{0}, register_failed{0}, unregistered{0}, unregister_failed{0},
        cleanup_failed{0}, staged_h2d_bytes{0}, staged_d2h_bytes{0}, direct_h2d_bytes{0}, direct_d2h_bytes{0},
        blocks{0}, lines_h2d_bytes{0}, lines_wait_pack_us{0}, lines_wait_copy_us{0}, blocks_retired{0}, found_page_locked{0},
        watched_calls{0}, last_rate_permille{0};
};
extern HostStats g_host_stats;

// a HIP status that cannot be returned (destructors): count it, keep its text
void note_cleanup(hipError_t e, const char *what);

// rows x row_bytes of X per block, the pipeline's unit (also exported: wagg_host_block_plan)
void host_block_plan(int64_t Tn, int64_t row_bytes, int64_t quantum, int n_devices, int64_t *block_rows, int64_t *n_blocks);

// Page-lock of a caller array for the duration of one call.  Arrays below PIN_MIN are not registered (registration
// works on whole pages, which a small heap array shares with unrelated objects; they are cheap to stage).
class HostPin {
  public:
    static constexpr size_t PIN_MIN = (size_t)32 << 20;
    HostPin() = default;
    HostPin(const HostPin &) = delete;
    HostPin &operator=(const HostPin &) = delete;
    ~HostPin() { (void)release(); }
    // tries to register; returns whether the array is now page-locked (an array that already is -- hipHostMalloc'ed, registered
    // by its owner, a framework's pinned block -- is taken as it is, whatever its size).  A refusal is not an error (the copies are
    // staged instead) but it is counted and its reason kept (why()).
    bool acquire(const void *p, size_t bytes, bool portable);
    bool pinned() const { return ptr_ != nullptr || foreign_; }
    hipError_t why() const { return why_; }
    // explicit release: the unregistration's status (also run by the destructor, which can only count a failure)
    hipError_t release();

  private:
    void *ptr_ = nullptr;
    bool foreign_ = false;          // the array was page-locked already (by its owner): used as it is, nothing to release
    hipError_t why_ = hipSuccess;
};

// The library's own page-locked staging: two pieces per direction and device, process lifetime.
constexpr size_t STAGE_PIECE = (size_t)8 << 20;
// host (pageable) -> device, through the staging pieces; returns when the user memory is no longer needed (the last
// piece may still be in flight on `st`: the pieces themselves are guarded by events)
hipError_t staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st);
// device block (rows x ld_bytes contiguous) -> pitched host rows (only row_bytes of each row are written); blocks until
// the host memory is completely written
hipError_t staged_d2h_rows(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, hipStream_t st);

// everything one device needs for its share of the blocks; every member released (and its status noted) on destruction
struct DevicePipe {
    int device = -1;
    hipStream_t sc = nullptr, sk = nullptr, sd = nullptr;      // H2D, kernels, D2H
    hipEvent_t ready[2] = {nullptr, nullptr}, kdone[2] = {nullptr, nullptr}, ddone[2] = {nullptr, nullptr};
    void *dx[2] = {nullptr, nullptr}, *dout[2] = {nullptr, nullptr};
    bool retire = false;                                        // the device blocks go back to the driver instead of the pool
    hipError_t init(int dev, bool set_device, size_t x_bytes, size_t o_bytes, int nbuf);
    hipError_t drain();                                         // synchronise the three streams
    ~DevicePipe();
};

// apply(slot, X_dev, rows, out_dev, stream): launch one block on device slot `slot` (0 .. n_dev - 1)
// release(slot, stream): the pipeline is about to destroy `stream` (per-stream plan state keyed by it must go)
struct HostRowsArgs {
    const char *X_host; char *out_host;
    int64_t Tn, ldx_bytes, xrow_bytes, ldo_bytes, orow_bytes, quantum;
    int flags, n_dev;
    const int *devices;                                         // nullptr: the current device, n_dev == 1
    std::function<int(int, const void *, int64_t, void *, hipStream_t)> apply;
    std::function<void(int, hipStream_t)> release;
    // "lines only" (WAGG_HOST_LINES; one device): n_runs > 0 makes the copy-in stage a GATHER -- host threads pack runs
    // [run_src[k], run_src[k] + run_len[k]) (bytes of a row of X) of every row side by side into page-locked ring pieces,
    // crow_bytes per row, and only those cross PCIe; `apply` then receives blocks of rows x crow_bytes (its own business
    // to read them with a matching cell table).  The caller's X is read by the CPU only and is never page-locked.
    const int64_t *run_src = nullptr;
    const int32_t *run_len = nullptr;
    int64_t n_runs = 0, crow_bytes = 0;
    // a second field of the same shape and pitch (the degree days' tasmax beside tasmin in X_host): whole rows: block f of a
    // device block at X_dev + f * rows * ldx_bytes; gathered rows: a compact row holds field 0 then field 1 (2 * crow_bytes)
    const char *X2_host = nullptr;
    // several result planes per block (the fused powers): `apply` writes plane k of a block of `rows` rows at
    // out_dev + k * rows * ldo_bytes; plane k of the whole result starts at out_host + k * opstride_bytes
    int n_planes = 1;
    int64_t opstride_bytes = 0;
};
int stream_host_rows_any(const HostRowsArgs &a);

template <typename T, typename ApplyFn, typename ReleaseFn>
int stream_host_rows(const T *X_host, int64_t Tn, int64_t ldx, int64_t G, T *out_host, int64_t ldo, int64_t R, int flags,
                     int64_t quantum, int n_dev, const int *devices, ApplyFn apply, ReleaseFn release) {
    HostRowsArgs a;
    a.X_host = reinterpret_cast<const char *>(X_host);
    a.out_host = reinterpret_cast<char *>(out_host);
    a.Tn = Tn;
    a.ldx_bytes = ldx * (int64_t)sizeof(T); a.xrow_bytes = G * (int64_t)sizeof(T);
    a.ldo_bytes = ldo * (int64_t)sizeof(T); a.orow_bytes = R * (int64_t)sizeof(T);
    a.quantum = quantum; a.flags = flags; a.n_dev = n_dev; a.devices = devices;
    a.apply = [&](int slot, const void *xd, int64_t rows, void *od, hipStream_t st) {
        return apply(slot, static_cast<const T *>(xd), rows, static_cast<T *>(od), st);
    };
    a.release = [&](int slot, hipStream_t st) { release(slot, st); };
    return stream_host_rows_any(a);
}

// CPUs this process may really use: the affinity mask, cut down to the cgroup's CPU quota where there is one
int granted_cpus();
// host threads the lines-only gather would start for a call made from this thread right now
int gather_team_threads();
// the page-locked ring of the gather (process lifetime, one call at a time; a second concurrent call does without and
// takes the plain path): nullptr when it is in use or cannot be had.  release_host_ring(): free it (wagg_release_scratch)
char *acquire_host_ring(size_t bytes);
void return_host_ring(char *p);
void release_host_ring();

// blocking host -> device copy of a whole buffer (plan uploads, the whole-field forms): staged, or in place under a
// page-lock of its own for large buffers.  `st`: the stream the copy is queued on and waited for (the null stream by default;
// a plan build passes its own so that the copy is not ordered against the null stream's work)
int copy_to_device(void *dst_dev, const void *src_host, size_t bytes, bool pin, hipStream_t st = nullptr);
// device (rows x ld, same pitch) -> pitched host array: only the `cols` used elements of every row are written
int copy_rows_to_host(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, bool pin);

}  // namespace wagg
