// Host-resident data: page-locks, staging and the row-block pipeline (wagg_host.h).  Host code only.
#include <sched.h>
#if defined(__x86_64__)
#include <emmintrin.h>
#endif
#include <chrono>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <system_error>

#include "wagg_host.h"

namespace wagg {

HostStats g_host_stats;
// failures while resources were released, counted per THREAD as well: a call compares its own threads' counts, so a
// concurrent call on another thread that hits a cleanup failure does not fail this one
static thread_local int64_t tl_release_failures = 0;

void note_cleanup(hipError_t e, const char *what) {
    if (e == hipSuccess) return;
    g_host_stats.cleanup_failed++;
    tl_release_failures++;
    if (wagg_last_error()[0] == '\0') set_error("%s -> %s", what, hipGetErrorString(e));
}

void host_block_plan(int64_t Tn, int64_t row_bytes, int64_t quantum, int n_devices, int64_t *block_rows, int64_t *n_blocks) {
    // ~256 MiB of X per block in whole multiples of `quantum` rows (the row count one launch handles well: 64 for the
    // segment-table kernels, a full 368 / 176-row block for the MFMA forms, whose W is streamed once per launch), at
    // least two blocks per device when there are that many quanta of rows (so that copies and kernels overlap)
    if (quantum < 1) quantum = 1;
    if (n_devices < 1) n_devices = 1;
    if (row_bytes < 1) row_bytes = 1;
    int64_t B = ((int64_t)256 << 20) / row_bytes;
    B = B < quantum ? quantum : B / quantum * quantum;
    const int64_t want = 2 * (int64_t)n_devices;
    if (Tn >= want * quantum && B > (Tn + want - 1) / want) B = (Tn + want - 1) / want / quantum * quantum;     // (>= quantum)
    if (B > Tn) B = Tn;
    if (B < 1) B = 1;
    *block_rows = B;
    *n_blocks = Tn > 0 ? (Tn + B - 1) / B : 0;
}

// ---- page-lock of a caller array ---------------------------------------------------------------------------------
// the whole range [p, p + bytes) lies in memory the runtime already knows as page-locked host memory (hipHostMalloc, an
// earlier hipHostRegister, a framework's pinned allocator): the copy paths can use it as it is
static bool already_page_locked(const void *p, size_t bytes) {
    if (!p || !bytes) return false;
    const char *ends[2] = {static_cast<const char *>(p), static_cast<const char *>(p) + bytes - 1};
    for (const char *q : ends) {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (at.type != hipMemoryTypeHost) return false;
    }
    return true;
}

bool HostPin::acquire(const void *p, size_t bytes, bool portable) {
    if (ptr_ || foreign_ || p == nullptr) return pinned();
    if (already_page_locked(p, bytes)) { foreign_ = true; g_host_stats.found_page_locked++; return true; }     // (any size: nothing to lock)
    if (bytes < PIN_MIN) return pinned();
    const hipError_t e = hipHostRegister(const_cast<void *>(p), bytes, portable ? hipHostRegisterPortable : hipHostRegisterDefault);
    if (e == hipSuccess) {
        ptr_ = const_cast<void *>(p);
        g_host_stats.registered++;
    } else {
        why_ = e;                               // e.g. the range is registered already, or page-locked memory is short
        g_host_stats.register_failed++;
        (void)hipGetLastError();                // the refusal is handled here (staged copies): it must not taint later calls
    }
    return pinned();
}

hipError_t HostPin::release() {
    foreign_ = false;
    if (!ptr_) return hipSuccess;
    const hipError_t e = hipHostUnregister(ptr_);
    ptr_ = nullptr;
    if (e == hipSuccess) {
        g_host_stats.unregistered++;
    } else {
        g_host_stats.unregister_failed++;
        tl_release_failures++;
        if (wagg_last_error()[0] == '\0') set_error("hipHostUnregister -> %s", hipGetErrorString(e));
    }
    return e;
}

// ---- the library's own page-locked staging ------------------------------------------------------------------------
namespace {
struct HostStage {
    std::mutex mu;
    char *buf[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t init() {
        for (int b = 0; b < 2; ++b) {
            if (!buf[b]) { hipError_t e = hipHostMalloc((void **)&buf[b], STAGE_PIECE, hipHostMallocDefault); if (e != hipSuccess) return e; }
            if (!ev[b]) { hipError_t e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming); if (e != hipSuccess) return e; }
        }
        return hipSuccess;
    }
};
constexpr int MAX_DEV = 64;
// per device and direction: a few MiB of page-locked memory, process lifetime (a copy up and a copy down of one call
// overlap; two calls on one device take turns)
HostStage g_up[MAX_DEV], g_down[MAX_DEV];

HostStage *stage_for(HostStage *set, hipError_t *e) {
    int dev = 0;
    *e = hipGetDevice(&dev);
    if (*e != hipSuccess) return nullptr;
    if (dev < 0 || dev >= MAX_DEV) { *e = hipErrorInvalidDevice; return nullptr; }
    return &set[dev];
}
}  // namespace

hipError_t staged_h2d(void *dst_dev, const void *src_host, size_t bytes, hipStream_t st) {
    if (bytes == 0) return hipSuccess;
    hipError_t e;
    HostStage *S = stage_for(g_up, &e);
    if (!S) return e;
    std::lock_guard<std::mutex> lock(S->mu);
    if ((e = S->init()) != hipSuccess) return e;
    size_t off = 0;
    int n_used = 0;
    for (int p = 0; off < bytes; ++p) {
        const int b = p & 1;
        const size_t n = bytes - off < STAGE_PIECE ? bytes - off : STAGE_PIECE;
        if (p >= 2 && (e = hipEventSynchronize(S->ev[b])) != hipSuccess) return e;      // piece p-2 has left the buffer
        std::memcpy(S->buf[b], static_cast<const char *>(src_host) + off, n);
        if ((e = hipMemcpyAsync(static_cast<char *>(dst_dev) + off, S->buf[b], n, hipMemcpyHostToDevice, st)) != hipSuccess) return e;
        if ((e = hipEventRecord(S->ev[b], st)) != hipSuccess) return e;
        off += n;
        n_used = p + 1 < 2 ? p + 1 : 2;
    }
    for (int b = 0; b < n_used; ++b)
        if ((e = hipEventSynchronize(S->ev[b])) != hipSuccess) return e;                  // buffers free for the next caller
    g_host_stats.staged_h2d_bytes += (int64_t)bytes;
    return hipSuccess;
}

// the device block is (rows x ld_bytes) contiguous; of every row the first row_bytes go to the same offsets on the host
hipError_t staged_d2h_rows(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, hipStream_t st) {
    if (rows <= 0 || row_bytes == 0) return hipSuccess;
    hipError_t e;
    HostStage *S = stage_for(g_down, &e);
    if (!S) return e;
    std::lock_guard<std::mutex> lock(S->mu);
    if ((e = S->init()) != hipSuccess) return e;
    const size_t bytes = (size_t)(rows - 1) * ld_bytes + row_bytes;
    auto scatter = [&](int b, size_t off, size_t n) {         // staged bytes [off, off + n) -> the used part of each row
        if (ld_bytes == row_bytes) { std::memcpy(static_cast<char *>(dst_host) + off, S->buf[b], n); return; }
        for (size_t r = off / ld_bytes; r < (size_t)rows && r * ld_bytes < off + n; ++r) {
            const size_t lo = r * ld_bytes > off ? r * ld_bytes : off;
            const size_t hi_row = r * ld_bytes + row_bytes, hi = hi_row < off + n ? hi_row : off + n;
            if (hi > lo) std::memcpy(static_cast<char *>(dst_host) + lo, S->buf[b] + (lo - off), hi - lo);
        }
    };
    size_t off = 0, prev_off = 0, prev_n = 0;
    int p = 0;
    for (; off < bytes; ++p) {
        const int b = p & 1;
        const size_t n = bytes - off < STAGE_PIECE ? bytes - off : STAGE_PIECE;
        if ((e = hipMemcpyAsync(S->buf[b], static_cast<const char *>(src_dev) + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
        if ((e = hipEventRecord(S->ev[b], st)) != hipSuccess) return e;
        if (p >= 1) {                                          // drain the previous piece while this one flies
            if ((e = hipEventSynchronize(S->ev[b ^ 1])) != hipSuccess) return e;
            scatter(b ^ 1, prev_off, prev_n);
        }
        prev_off = off; prev_n = n;
        off += n;
    }
    const int last = (p - 1) & 1;
    if ((e = hipEventSynchronize(S->ev[last])) != hipSuccess) return e;
    scatter(last, prev_off, prev_n);
    g_host_stats.staged_d2h_bytes += (int64_t)((size_t)rows * row_bytes);
    return hipSuccess;
}

int copy_to_device(void *dst_dev, const void *src_host, size_t bytes, bool pin, hipStream_t st) {
    if (bytes == 0) return WAGG_OK;
    HostPin lock;
    if (pin && lock.acquire(src_host, bytes, false)) {
        // from page-locked memory: one DMA, waited for before the lock goes
        hipError_t e = st ? hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, st)
                          : hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess && st) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            (void)lock.release();
            set_error("%s:%d: host -> device copy of %zu bytes -> %s", __FILE__, __LINE__, bytes, hipGetErrorString(e));
            return WAGG_EHIP;
        }
        g_host_stats.direct_h2d_bytes += (int64_t)bytes;
        WAGG_HIP(lock.release());
        return WAGG_OK;
    }
    WAGG_HIP(staged_h2d(dst_dev, src_host, bytes, st));
    return WAGG_OK;
}

int copy_rows_to_host(void *dst_host, const void *src_dev, int64_t rows, size_t ld_bytes, size_t row_bytes, bool pin) {
    if (rows <= 0 || row_bytes == 0) return WAGG_OK;
    const size_t span = (size_t)(rows - 1) * ld_bytes + row_bytes;
    HostPin lock;
    if (pin && lock.acquire(dst_host, span, false)) {
        if (ld_bytes == row_bytes) WAGG_HIP(hipMemcpy(dst_host, src_dev, span, hipMemcpyDeviceToHost));
        else WAGG_HIP(hipMemcpy2D(dst_host, ld_bytes, src_dev, ld_bytes, row_bytes, (size_t)rows, hipMemcpyDeviceToHost));
        g_host_stats.direct_d2h_bytes += (int64_t)((size_t)rows * row_bytes);
        WAGG_HIP(lock.release());
        return WAGG_OK;
    }
    WAGG_HIP(staged_d2h_rows(dst_host, src_dev, rows, ld_bytes, row_bytes, nullptr));
    return WAGG_OK;
}

// ---- "lines only": the gather of the referenced runs of every row (WAGG_HOST_LINES) -----------------------------------
int granted_cpus() {
    // the calling thread's affinity mask, read per call: threads started from here inherit it, and it can be narrower than
    // the process's (an OpenMP runtime with OMP_PROC_BIND binds the thread that entered its first parallel region)
    int cpus = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
    if (cpus < 1) cpus = (int)std::thread::hardware_concurrency();
    if (cpus < 1) cpus = 1;
    // cgroup v2: "<quota> <period>" or "max <period>"; v1: cfs_quota_us (-1 = none) / cfs_period_us -- read once
    static const int by_quota = []() {
        long long quota = -1, period = 0;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE *f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(f2, "%lld", &period) != 1) period = 0;
                fclose(f2);
            }
        }
        return quota > 0 && period > 0 ? (int)((quota + period - 1) / period) : 0;
    }();
    if (by_quota >= 1 && by_quota < cpus) cpus = by_quota;
    return cpus;
}

int gather_team_threads() {
    // all but four of the CPUs this thread may use (at least half of them), twelve at most
    const int granted = granted_cpus();
    int want = granted / 2;
    if (granted - 4 > want) want = granted - 4;
    return want > 12 ? 12 : want;
}

namespace {
struct HostRing { std::mutex mu; char *p = nullptr; size_t cap = 0; bool busy = false; } g_ring;
}  // namespace

char *acquire_host_ring(size_t bytes) {
    std::lock_guard<std::mutex> lock(g_ring.mu);
    if (g_ring.busy) return nullptr;
    if (g_ring.cap < bytes) {
        if (g_ring.p) { note_cleanup(hipHostFree(g_ring.p), "hipHostFree(gather ring)"); g_ring.p = nullptr; g_ring.cap = 0; }
        void *q = nullptr;
        if (hipHostMalloc(&q, bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        g_ring.p = static_cast<char *>(q); g_ring.cap = bytes;
    }
    g_ring.busy = true;
    return g_ring.p;
}

void return_host_ring(char *p) {
    std::lock_guard<std::mutex> lock(g_ring.mu);
    if (p && p == g_ring.p) g_ring.busy = false;
}

void release_host_ring() {
    std::lock_guard<std::mutex> lock(g_ring.mu);
    if (g_ring.busy || !g_ring.p) return;
    note_cleanup(hipHostFree(g_ring.p), "hipHostFree(gather ring)");
    g_ring.p = nullptr; g_ring.cap = 0;
}

namespace {
// One run of a row into the ring: streaming stores (the packed bytes are read next by the copy engine, not by a CPU: no
// write-allocate read of the destination, no dirty lines for the DMA to snoop out of the caches -- tools/micro/host_gather.cpp:
// 20.1 -> 18.0 ms for the c2-real field).  dst and n are multiples of 16 (whole quads of a compact row in page-locked memory).
static inline void copy_run(char *dst, const char *src, size_t n) {
#if defined(__x86_64__)
    for (size_t i = 0; i < n; i += 16)
        _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i)));
#else
    std::memcpy(dst, src, n);
#endif
}
static inline void runs_done() {
#if defined(__x86_64__)
    _mm_sfence();
#endif
}
static inline int64_t now_us() {
    return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// Host threads that pack the rows of the field, in order, into ring pieces of RPP rows; the pipeline's thread sends a
// piece off as soon as all its rows are there and hands its slot back once the copy engine has read it.  Rows are
// claimed in ascending order from one counter, so the piece everybody may be waiting for is always being worked on.
struct GatherTeam {
    static constexpr int RPP = 8, SLOTS = 4;
    const HostRowsArgs &a;
    char *ring = nullptr;
    size_t piece_bytes = 0;
    int64_t n_pieces = 0;
    std::vector<int64_t> doff;                           // byte position of run k in the compact row
    std::unique_ptr<std::atomic<int>[]> done;            // rows packed, per piece
    std::atomic<int64_t> next_row{0}, free_upto{SLOTS};  // pieces below free_upto own a free slot
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    hipEvent_t pev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int64_t sent_bytes = 0, copy_wait_us = 0;            // of this call: packed bytes queued, time spent waiting for their copies
    int64_t crow_all = 0;                                // bytes of a packed row: every field's compact row side by side

    explicit GatherTeam(const HostRowsArgs &args) : a(args) {}
    GatherTeam(const GatherTeam &) = delete;
    GatherTeam &operator=(const GatherTeam &) = delete;
    // false: the team cannot work (too few usable CPUs, the ring is busy, threads cannot be had) -- the caller takes the plain
    // path; no copy has been queued (threads that did start are joined by the destructor)
    bool start() {
        try {
            // all but four of the CPUs this thread may use (at least half of them), twelve at most -- and not fewer than it
            // takes to pack faster than the copy path would move the WHOLE rows (one thread packs ~10 GB/s, PCIe moves ~56:
            // 6 x the packed fraction of a row; measured, tools/micro/host_gather.cpp: 4 threads 24-26 ms, 8 threads 18 ms,
            // whole rows 26.3 ms).  Twelve instead of eight of 16 CPUs: nothing for the 455-byte runs of the fp32 lines
            // (8 threads already hide behind PCIe), but the 189-byte runs of the degree-day chunking and the fp64 lines
            // stop being a co-bottleneck (degree days: 14-18 ms waiting for the packers -> 1.7 ms, 30.3-34.3 -> 29.9-30.3 ms)
            const int want = gather_team_threads();
            const int need = (int)((6 * a.crow_bytes + a.xrow_bytes - 1) / a.xrow_bytes);       // (per field: two fields, twice the bytes on both sides)
            if (want < 1 || want < need || (int64_t)want > a.Tn) return false;
            crow_all = a.crow_bytes * (a.X2_host ? 2 : 1);
            piece_bytes = (size_t)(RPP * crow_all);
            n_pieces = (a.Tn + RPP - 1) / RPP;
            doff.resize((size_t)a.n_runs);
            int64_t off = 0;
            for (int64_t k = 0; k < a.n_runs; ++k) { doff[(size_t)k] = off; off += a.run_len[k]; }
            if (off != a.crow_bytes) return false;
            done.reset(new std::atomic<int>[(size_t)n_pieces]);
            for (int64_t p = 0; p < n_pieces; ++p) done[(size_t)p].store(0, std::memory_order_relaxed);
            for (int s = 0; s < SLOTS; ++s)
                if (hipEventCreateWithFlags(&pev[s], hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return false; }
            ring = acquire_host_ring(piece_bytes * SLOTS);
            if (!ring) return false;
            th.reserve((size_t)want);
            for (int w = 0; w < want; ++w) {
                try { th.emplace_back([this]() { work(); }); } catch (const std::system_error &) { break; }
            }
            return (int)th.size() >= need;               // (fewer than that could be started: the destructor joins them)
        } catch (const std::bad_alloc &) { return false; }
    }
    void work() {
        for (;;) {
            const int64_t r = next_row.fetch_add(1, std::memory_order_relaxed);
            if (r >= a.Tn) return;
            const int64_t p = r / RPP;
            for (int spins = 0; p >= free_upto.load(std::memory_order_acquire); ++spins) {      // (the ring is full: PCIe sets the pace)
                if (stop.load(std::memory_order_relaxed)) return;
                if (spins < 64) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            char *dst = ring + (size_t)(p % SLOTS) * piece_bytes + (size_t)((r % RPP) * crow_all);
            for (int f = 0; f < (a.X2_host ? 2 : 1); ++f) {
                const char *src = (f ? a.X2_host : a.X_host) + r * a.ldx_bytes;
                char *df = dst + (size_t)f * (size_t)a.crow_bytes;
                for (int64_t k = 0; k < a.n_runs; ++k) copy_run(df + doff[(size_t)k], src + a.run_src[k], (size_t)a.run_len[k]);
            }
            runs_done();
            done[(size_t)p].fetch_add(1, std::memory_order_release);
        }
    }
    // rows [r0, r0 + rows) of the field(s) -> dst_dev (rows x crow_all), queued on `sc` piece by piece
    hipError_t send_block(char *dst_dev, int64_t r0, int64_t rows, hipStream_t sc) {
        hipError_t e;
        for (int64_t p = r0 / RPP; p * RPP < r0 + rows; ++p) {
            const int64_t pr0 = p * RPP, prow = (pr0 + RPP <= a.Tn ? RPP : a.Tn - pr0);
            const int64_t w0 = now_us();
            while (done[(size_t)p].load(std::memory_order_acquire) < (int)prow) std::this_thread::yield();
            g_host_stats.lines_wait_pack_us += now_us() - w0;
            if ((e = hipMemcpyAsync(dst_dev + (size_t)((pr0 - r0) * crow_all), ring + (size_t)(p % SLOTS) * piece_bytes,
                                    (size_t)(prow * crow_all), hipMemcpyHostToDevice, sc)) != hipSuccess) return e;
            if ((e = hipEventRecord(pev[p % SLOTS], sc)) != hipSuccess) return e;
            g_host_stats.lines_h2d_bytes += prow * crow_all;
            sent_bytes += prow * crow_all;
            if (p >= 1) {       // the copy of piece p - 1 is over (p stays queued behind it): its slot goes to piece p - 1 + SLOTS
                const int64_t w1 = now_us();
                if ((e = hipEventSynchronize(pev[(p - 1) % SLOTS])) != hipSuccess) return e;
                const int64_t dw = now_us() - w1;
                g_host_stats.lines_wait_copy_us += dw;
                copy_wait_us += dw;
                free_upto.store(p + SLOTS, std::memory_order_release);
            }
        }
        return hipSuccess;
    }
    ~GatherTeam() {
        stop.store(true);
        next_row.store(a.Tn);
        for (std::thread &t : th) if (t.joinable()) t.join();
        // (the ring returns only once nothing can still be reading it: run_device_ has drained its streams, or is about to
        //  fail the call -- the events below are waited for either way)
        for (int s = 0; s < SLOTS; ++s)
            if (pev[s]) { note_cleanup(hipEventSynchronize(pev[s]), "hipEventSynchronize(gather piece)"); note_cleanup(hipEventDestroy(pev[s]), "hipEventDestroy"); }
        if (ring) return_host_ring(ring);
    }
};
}  // namespace

// ---- one device's streams, events and block buffers ---------------------------------------------------------------
hipError_t DevicePipe::init(int dev, bool set_device, size_t x_bytes, size_t o_bytes, int nbuf) {
    // streams and block buffers from the scratch pool (wagg_scratch.hip): 0.6 GB of hipMalloc / hipFree and three stream
    // creations per call were 1.4 ms of every 31 ms host-resident apply and most of its tail (tools/host_apply_churn.py)
    hipError_t e = hipSuccess;
    if (set_device) { if ((e = hipSetDevice(dev)) != hipSuccess) return e; }
    device = dev;
    if ((e = scratch_stream(&sc, 1)) != hipSuccess) return e;
    if ((e = scratch_stream(&sk)) != hipSuccess) return e;
    if ((e = scratch_stream(&sd, 2)) != hipSuccess) return e;
    for (int b = 0; b < nbuf && b < 2; ++b) {
        if ((e = hipEventCreateWithFlags(&ready[b], hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&kdone[b], hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = hipEventCreateWithFlags(&ddone[b], hipEventDisableTiming)) != hipSuccess) return e;
        if ((e = scratch_alloc(&dx[b], x_bytes)) != hipSuccess) return e;
        if ((e = scratch_alloc(&dout[b], o_bytes)) != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t DevicePipe::drain() {
    hipError_t e;
    if (sc && (e = hipStreamSynchronize(sc)) != hipSuccess) return e;
    if (sk && (e = hipStreamSynchronize(sk)) != hipSuccess) return e;
    if (sd && (e = hipStreamSynchronize(sd)) != hipSuccess) return e;
    return hipSuccess;
}

DevicePipe::~DevicePipe() {
    // nothing may still be in flight on the buffers, whatever path led here
    if (sc) note_cleanup(hipStreamSynchronize(sc), "hipStreamSynchronize(copy-in stream)");
    if (sk) note_cleanup(hipStreamSynchronize(sk), "hipStreamSynchronize(kernel stream)");
    if (sd) note_cleanup(hipStreamSynchronize(sd), "hipStreamSynchronize(copy-out stream)");
    for (int b = 0; b < 2; ++b) {
        if (ready[b]) note_cleanup(hipEventDestroy(ready[b]), "hipEventDestroy");
        if (kdone[b]) note_cleanup(hipEventDestroy(kdone[b]), "hipEventDestroy");
        if (ddone[b]) note_cleanup(hipEventDestroy(ddone[b]), "hipEventDestroy");
        scratch_free(dx[b], !retire);
        scratch_free(dout[b], !retire);
    }
    scratch_stream_done(sc, 1);
    scratch_stream_done(sk, 0);
    scratch_stream_done(sd, 2);
}

// ---- the pipeline of one device: blocks slot, slot + n_dev, ... ---------------------------------------------------------
static int run_device_(const HostRowsArgs &a, int slot, bool set_device, int64_t B, int64_t nb, bool pin_x, bool pin_o);
// *release_failures: what this thread's share of the call failed to release (DevicePipe's destructor has run by then)
static int run_device(const HostRowsArgs &a, int slot, bool set_device, int64_t B, int64_t nb, bool pin_x, bool pin_o,
                      int64_t *release_failures) {
    const int64_t before = tl_release_failures;
    const int rc = run_device_(a, slot, set_device, B, nb, pin_x, pin_o);
    *release_failures = tl_release_failures - before;
    return rc;
}
static int run_device_(const HostRowsArgs &a, int slot, bool set_device, int64_t B, int64_t nb, bool pin_x, bool pin_o) {
    const int64_t my_blocks = nb > slot ? (nb - slot + a.n_dev - 1) / a.n_dev : 0;
    if (my_blocks == 0) return WAGG_OK;
    // (declared before the pipe: the pipe's destructor drains the streams, then the team waits for its pieces and goes)
    std::unique_ptr<GatherTeam> team;
    const bool gather = a.n_runs > 0;
    DevicePipe P;
    const int dev = a.devices ? a.devices[slot] : -1;
    int cur = 0;
    if (!set_device) WAGG_HIP(hipGetDevice(&cur));
    WAGG_HIP(P.init(set_device ? dev : cur, set_device, (size_t)(B * (gather ? a.crow_bytes : a.ldx_bytes) * (a.X2_host ? 2 : 1)), (size_t)(B * a.ldo_bytes * a.n_planes), my_blocks >= 2 ? 2 : 1));
    if (gather) {
        team.reset(new (std::nothrow) GatherTeam(a));
        if (!team || !team->start()) return WAGG_EUNSUPPORTED;       // (nothing queued yet: the caller takes the plain path)
    }
    struct Releaser {                                   // per-stream plan state keyed by P.sk goes before the stream does --
        const HostRowsArgs &a; int slot; hipStream_t s; // and, on every path (also the early returns), only once nothing on
        ~Releaser() {                                   // that stream can still be using it (this runs BEFORE ~DevicePipe)
            note_cleanup(hipStreamSynchronize(s), "hipStreamSynchronize(kernel stream, before its plan state is dropped)");
            if (a.release) a.release(slot, s);
        }
    } rel{a, slot, P.sk};
    auto span = [](int64_t rows, int64_t ld, int64_t row) { return (size_t)((rows - 1) * ld + row); };
    const int64_t t_begin_us = now_us();
    int64_t moved_plain = 0;
    int64_t prev_r0 = -1, prev_rows = 0;
    int prev_b = 0;
    auto drain_prev = [&]() -> int {                    // staged return of the previous block (blocks the host)
        if (prev_r0 < 0) return WAGG_OK;
        WAGG_HIP(hipStreamWaitEvent(P.sd, P.kdone[prev_b], 0));
        for (int k = 0; k < a.n_planes; ++k)
            WAGG_HIP(staged_d2h_rows(a.out_host + k * a.opstride_bytes + prev_r0 * a.ldo_bytes,
                                     static_cast<const char *>(P.dout[prev_b]) + (size_t)(k * prev_rows * a.ldo_bytes), prev_rows,
                                     (size_t)a.ldo_bytes, (size_t)a.orow_bytes, P.sd));
        WAGG_HIP(hipEventRecord(P.ddone[prev_b], P.sd));
        prev_r0 = -1;
        return WAGG_OK;
    };
    int64_t j = 0;
    for (int64_t i = slot; i < nb; i += a.n_dev, ++j) {
        const int b = (int)(j & 1);
        const int64_t r0 = i * B, rows = a.Tn - r0 < B ? a.Tn - r0 : B;
        const char *src = a.X_host + r0 * a.ldx_bytes;
        const size_t xspan = span(rows, a.ldx_bytes, a.xrow_bytes);
        if (j >= 2) WAGG_HIP(hipStreamWaitEvent(P.sc, P.kdone[b], 0));            // the kernels of block j-2 have read dx[b]
        if (gather) {
            WAGG_HIP(team->send_block(static_cast<char *>(P.dx[b]), r0, rows, P.sc));
        } else {
            for (int f = 0; f < (a.X2_host ? 2 : 1); ++f) {
                const char *sf = f ? a.X2_host + r0 * a.ldx_bytes : src;
                char *df = static_cast<char *>(P.dx[b]) + (size_t)f * (size_t)(rows * a.ldx_bytes);
                if (pin_x) {
                    WAGG_HIP(hipMemcpyAsync(df, sf, xspan, hipMemcpyHostToDevice, P.sc));        // page-locked source: truly asynchronous
                    g_host_stats.direct_h2d_bytes += (int64_t)xspan;
                    moved_plain += (int64_t)xspan;
                } else {
                    WAGG_HIP(staged_h2d(df, sf, xspan, P.sc));
                }
            }
        }
        WAGG_HIP(hipEventRecord(P.ready[b], P.sc));
        WAGG_HIP(hipStreamWaitEvent(P.sk, P.ready[b], 0));
        if (j >= 2) WAGG_HIP(hipStreamWaitEvent(P.sk, P.ddone[b], 0));            // the result of block j-2 has left dout[b]
        const int rc = a.apply(slot, P.dx[b], rows, P.dout[b], P.sk);
        if (rc != WAGG_OK) return rc;
        WAGG_HIP(hipEventRecord(P.kdone[b], P.sk));
        g_host_stats.blocks++;
        if (pin_o) {                                                               // page-locked destination: asynchronous
            WAGG_HIP(hipStreamWaitEvent(P.sd, P.kdone[b], 0));
            for (int k = 0; k < a.n_planes; ++k) {
                char *dst = a.out_host + k * a.opstride_bytes + r0 * a.ldo_bytes;
                const char *srcd = static_cast<const char *>(P.dout[b]) + (size_t)(k * rows * a.ldo_bytes);
                if (a.ldo_bytes == a.orow_bytes)
                    WAGG_HIP(hipMemcpyAsync(dst, srcd, (size_t)(rows * a.orow_bytes), hipMemcpyDeviceToHost, P.sd));
                else
                    WAGG_HIP(hipMemcpy2DAsync(dst, (size_t)a.ldo_bytes, srcd, (size_t)a.ldo_bytes, (size_t)a.orow_bytes, (size_t)rows,
                                              hipMemcpyDeviceToHost, P.sd));
            }
            WAGG_HIP(hipEventRecord(P.ddone[b], P.sd));
            g_host_stats.direct_d2h_bytes += (int64_t)(rows * a.orow_bytes) * a.n_planes;
        } else {
            // the previous block's result returns through the staging pieces WHILE this block's kernels run (they are
            // queued already); this block's own result follows in the next round (or behind the loop)
            if (int rc2 = drain_prev()) return rc2;
            prev_r0 = r0; prev_rows = rows; prev_b = b;
        }
    }
    if (int rc2 = drain_prev()) return rc2;
    WAGG_HIP(P.drain());
    // Copy-rate watch.  The pipeline can get into a state in which the copies into its pooled device blocks run at half the
    // rate (seen on MI355X / ROCm 7.2 in one call sequence of round 5 -- plain lines-only calls after wagg_apply_poly_host_* calls
    // with page-locked whole rows: 21.3 -> 47 ms per call, wait for the copy engine 16.6 -> 31 ms; it stayed until the device
    // blocks went back to the driver -- fresh streams or a fresh ring did not help; not reproduced in five fresh processes in
    // round 6, with the SDMA engines on or off: DESIGN.md section 6).  What the watch compares, per device and per SHAPE of call
    // (lines only or whole rows, one or two fields, the number of result planes, the packed fraction of a row, where the result
    // goes -- a call with four result planes or two fields is legitimately slower than a plain one and never shares a record
    // with it): the bytes that crossed PCIe over the call's wall time.  (Round 6 first judged lines-only calls on the time spent
    // waiting for their copies; with the quads-only rows the packing threads and the copies take turns as the bottleneck, a call
    // that waited little for its copies set a record the next one could not meet, and the watch misfired four times in nine
    // calls.)  A call that moved >= 256 MiB at < 60 % of the best rate seen for its shape hands ITS OWN four device blocks back
    // to the driver (the next call allocates fresh ones, ~1 ms); what idles in the pool -- the build arena among it -- is left
    // alone.  A merely busy host or PCIe link can trip it too: the cost is that one millisecond.  The verdict is visible:
    // wagg_host_stats.watched_calls / .last_rate_permille / .blocks_retired.
    {
        const int64_t moved = gather ? team->sent_bytes : moved_plain;
        const int64_t us = now_us() - t_begin_us;
        if (moved >= ((int64_t)256 << 20) && us > 0 && (gather || pin_x)) {
            struct Best { uint64_t key; double rate; };
            static std::mutex mu;
            static std::vector<Best> best;
            const uint64_t frac16 = gather ? (uint64_t)((16 * a.crow_bytes + a.xrow_bytes - 1) / a.xrow_bytes) : 16u;
            const uint64_t key = ((uint64_t)(P.device & 0xff) << 32) | ((uint64_t)(gather ? 1 : 0) << 31) | ((uint64_t)(a.X2_host ? 1 : 0) << 30) |
                                 ((uint64_t)(pin_o ? 1 : 0) << 29) | ((uint64_t)(a.n_planes & 0xff) << 8) | frac16;
            const double rate = (double)moved / (double)us;
            try {
                std::lock_guard<std::mutex> lock(mu);
                Best *b = nullptr;
                for (Best &k : best) if (k.key == key) { b = &k; break; }
                if (!b) { best.push_back(Best{key, 0.0}); b = &best.back(); }
                g_host_stats.watched_calls++;
                if (rate >= b->rate) { b->rate = rate; g_host_stats.last_rate_permille = 1000; }
                else {
                    g_host_stats.last_rate_permille = (int64_t)(1000.0 * rate / b->rate);
                    if (rate < 0.6 * b->rate) { P.retire = true; g_host_stats.blocks_retired++; }
                }
            } catch (const std::bad_alloc &) {}           // (no record: nothing watched)
        } else {
            g_host_stats.last_rate_permille = 0;          // the last call was not one the watch judges
        }
    }
    return WAGG_OK;
}

int stream_host_rows_any(const HostRowsArgs &a) {
    if (a.Tn == 0) return WAGG_OK;
    WAGG_REQUIRE(a.n_dev >= 1 && a.n_dev <= 64, "n_devices must lie in [1, 64], got %d", a.n_dev);
    g_host_stats.calls++;
    int64_t release_failures = 0;                        // of THIS call: its own threads' counts
    const int64_t fail0 = tl_release_failures;
    int64_t B, nb;
    host_block_plan(a.Tn, a.ldx_bytes, a.quantum, a.n_dev, &B, &nb);
    if (a.n_runs > 0) {
        WAGG_REQUIRE(a.n_dev == 1 && a.devices == nullptr, "the lines-only host path drives one device");
        WAGG_REQUIRE(nb == 1 || B % GatherTeam::RPP == 0, "row blocks of %lld rows do not hold whole gather pieces", (long long)B);
    }
    WAGG_REQUIRE(a.n_planes >= 1 && (a.n_planes == 1 || a.opstride_bytes >= a.Tn * a.ldo_bytes), "bad result planes");
    const size_t xbytes = (size_t)((a.Tn - 1) * a.ldx_bytes + a.xrow_bytes),
                 obytes = (size_t)((a.n_planes - 1) * a.opstride_bytes + (a.Tn - 1) * a.ldo_bytes + a.orow_bytes);
    int rc = WAGG_OK;
    {
        HostPin px, px2, po;
        const bool want = (a.flags & WAGG_HOST_PIN) != 0;
        bool pin_x = want && a.n_runs == 0 && px.acquire(a.X_host, xbytes, a.n_dev > 1);            // (gathered rows: the CPU reads X)
        if (pin_x && a.X2_host && !px2.acquire(a.X2_host, xbytes, a.n_dev > 1)) pin_x = false;      // (both fields, or both staged)
        const bool pin_o = want && po.acquire(a.out_host, obytes, a.n_dev > 1);
        if (a.n_dev == 1 && a.devices == nullptr) {
            int64_t f = 0;
            rc = run_device(a, 0, false, B, nb, pin_x, pin_o, &f);      // (this thread: counted through fail0 below)
        } else {
            // one host thread per device: each drives its own copy engines and kernels, results land straight in the
            // caller's rows (no exchange between devices)
            std::vector<int> rcs((size_t)a.n_dev, WAGG_OK);
            std::vector<int64_t> fails((size_t)a.n_dev, 0);
            std::vector<std::string> msgs((size_t)a.n_dev);
            std::vector<std::thread> th;
            int home = 0;
            WAGG_HIP(hipGetDevice(&home));
            for (int s = 0; s < a.n_dev; ++s)
                th.emplace_back([&, s]() {
                    rcs[(size_t)s] = run_device(a, s, true, B, nb, pin_x, pin_o, &fails[(size_t)s]);
                    if (rcs[(size_t)s] != WAGG_OK || fails[(size_t)s]) msgs[(size_t)s] = wagg_last_error();
                });
            for (auto &t : th) t.join();
            for (int s = 0; s < a.n_dev; ++s) {
                release_failures += fails[(size_t)s];
                if (fails[(size_t)s] && wagg_last_error()[0] == '\0' && !msgs[(size_t)s].empty()) set_error("device slot %d: %s", s, msgs[(size_t)s].c_str());
            }
            for (int s = 0; s < a.n_dev && rc == WAGG_OK; ++s)
                if (rcs[(size_t)s] != WAGG_OK) { rc = rcs[(size_t)s]; set_error("device slot %d: %s", s, msgs[(size_t)s].c_str()); }
            (void)home;                                  // (the calling thread's current device was never changed)
        }
        // the page-locks go only now: every stream that touched the arrays has been drained and destroyed
        const hipError_t ex1 = px.release(), ex2 = px2.release(), eo = po.release();
        const hipError_t ex = ex1 != hipSuccess ? ex1 : ex2;
        if (rc == WAGG_OK && (ex != hipSuccess || eo != hipSuccess)) {
            set_error("hipHostUnregister -> %s", hipGetErrorString(ex != hipSuccess ? ex : eo));
            rc = WAGG_EHIP;
        }
    }
    release_failures += tl_release_failures - fail0;
    if (rc == WAGG_OK && release_failures != 0) {
        if (wagg_last_error()[0] == '\0') set_error("a HIP call failed while the host pipeline released its resources");
        rc = WAGG_EHIP;
    }
    return rc;
}

}  // namespace wagg

extern "C" int wagg_host_block_plan(int64_t T, int64_t row_bytes, int64_t quantum, int n_devices, int64_t *block_rows,
                                    int64_t *n_blocks) {
    using namespace wagg;
    WAGG_REQUIRE(T >= 0 && row_bytes >= 1 && quantum >= 1 && n_devices >= 1, "bad block plan request");
    WAGG_REQUIRE(block_rows && n_blocks, "NULL argument");
    host_block_plan(T, row_bytes, quantum, n_devices, block_rows, n_blocks);
    return WAGG_OK;
}

extern "C" int wagg_host_stats_read_sized(void *out_buf, uint64_t size, int reset) {
    using namespace wagg;
    WAGG_REQUIRE(out_buf != nullptr, "NULL argument");
    HostStats &s = g_host_stats;
    wagg_host_stats st;
    std::memset(&st, 0, sizeof(st));
    wagg_host_stats *out = &st;
    out->calls = s.calls; out->blocks = s.blocks;
    out->registered = s.registered; out->register_failed = s.register_failed;
    out->unregistered = s.unregistered; out->unregister_failed = s.unregister_failed;
    out->cleanup_failed = s.cleanup_failed;
    out->staged_h2d_bytes = s.staged_h2d_bytes; out->staged_d2h_bytes = s.staged_d2h_bytes;
    out->direct_h2d_bytes = s.direct_h2d_bytes; out->direct_d2h_bytes = s.direct_d2h_bytes;
    out->blocks_retired = s.blocks_retired; out->found_page_locked = s.found_page_locked;
    out->lines_h2d_bytes = s.lines_h2d_bytes; out->lines_wait_pack_us = s.lines_wait_pack_us; out->lines_wait_copy_us = s.lines_wait_copy_us;
    out->watched_calls = s.watched_calls; out->last_rate_permille = s.last_rate_permille;
    if (reset) {
        s.watched_calls = 0;                       // (last_rate_permille is a state, not a counter: it stays)
        s.calls = 0; s.blocks = 0; s.registered = 0; s.register_failed = 0; s.unregistered = 0; s.unregister_failed = 0;
        s.cleanup_failed = 0; s.staged_h2d_bytes = 0; s.staged_d2h_bytes = 0; s.direct_h2d_bytes = 0; s.direct_d2h_bytes = 0;
        s.lines_h2d_bytes = 0; s.lines_wait_pack_us = 0; s.lines_wait_copy_us = 0; s.blocks_retired = 0; s.found_page_locked = 0;
    }
    copy_sized(out_buf, size, &st, sizeof(st));
    return WAGG_OK;
}
extern "C" int wagg_host_stats_read(wagg_host_stats *out, int reset) {
    return wagg_host_stats_read_sized(out, sizeof(wagg_host_stats), reset);
}
