// Device scratch that outlives the call that asked for it: blocks and streams the library needs for the length of ONE call
// (the arena of a table -> plan build, the block buffers and the three streams of a host-resident apply, a plan's
// per-stream staging) are taken from and returned to a small per-device pool instead of hipMalloc / hipFree and
// hipStreamCreate / hipStreamDestroy.
//
// Why: the driver reclaims freed VRAM lazily.  After some 90-130 GB of LARGE blocks have gone through hipMalloc / hipFree --
// twelve c5 plan builds with their 11 GB arena -- the next hipMalloc waits for the reclaim: 0.7 to 4 s, seen in the
// builder's own timings and reproduced with nothing but hipMalloc / hipFree (tools/diag/malloc_stall.cpp on MI355X /
// ROCm 7.2: 11 GiB blocks stall for 3.8 s at the twelfth cycle, 0.5 GiB blocks for 2.1 s at cycles 173, 344, 518, 694).
// The 256 MiB block buffers of a host-resident apply did not reach it in 400 calls (tools/host_apply_churn.py), but they
// and the call's three streams (1.4 ms to create, 1.2-2.4 ms to destroy, each) cost 1.4 ms of every 31 ms call and a
// tail of 44-51 ms; from the pool: 29.8 ms median, 32.8 ms worst.  Frameworks with a caching allocator see neither.
//
// Policy: a freed block is kept unless it is larger than 1/16 of the device's memory; when the blocks kept for a device
// exceed that same budget the least recently returned go back to the driver.  A request takes the smallest kept block that
// is large enough and at most a quarter (at least 32 MiB) larger than asked.  Streams: up to four idle ones per device and role (kernels, copies in, copies out).
// wagg_release_scratch() frees everything kept (the package's clear_caches() calls it), and an allocation of the library
// that runs out of device memory does so before it tries once more.
//
// Contract for callers: a block or stream is returned only when nothing on the device can still be using it (the caller
// has synchronised the streams it used it on) -- the next taker uses it without any wait.
#include <mutex>
#include <new>
#include <vector>

#include "wagg_common.h"
#include "wagg_host.h"

namespace wagg {
namespace {
struct Block { int device; void *p; size_t cap; uint64_t tick; };
struct IdleStream { int device; hipStream_t st; int role; };
std::mutex g_mu;
std::vector<Block> g_free, g_live;               // kept blocks; blocks handed out (their size is needed when they return)
std::vector<IdleStream> g_streams;
uint64_t g_tick = 0;
constexpr size_t ROUND = (size_t)2 << 20;
constexpr size_t MAX_IDLE_STREAMS = 8;

size_t device_budget(int device) {
    static std::mutex mu;
    static std::vector<std::pair<int, size_t>> known;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &k : known) if (k.first == device) return k.second;
    size_t total = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) total = prop.totalGlobalMem;
    try { known.emplace_back(device, total / 16); } catch (const std::bad_alloc &) {}
    return total / 16;
}

void give_back(const std::vector<Block> &gone) {
    for (const Block &b : gone) note_cleanup(hipFree(b.p), "hipFree(kept scratch block)");
}
}  // namespace

void release_scratch(int what) {                 // 1 device blocks, 2 idle streams, 4 the host ring
    std::vector<Block> blocks;
    std::vector<IdleStream> streams;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        if (what & 1) blocks.swap(g_free);
        if (what & 2) streams.swap(g_streams);
    }
    give_back(blocks);
    for (const IdleStream &s : streams) note_cleanup(hipStreamDestroy(s.st), "hipStreamDestroy(kept stream)");
    if (what & 4) release_host_ring();           // (the page-locked ring of the lines-only host path, wagg_host.hip)
}

int64_t scratch_bytes_kept() {
    std::lock_guard<std::mutex> lock(g_mu);
    int64_t total = 0;
    for (const Block &b : g_free) total += (int64_t)b.cap;
    return total;
}

hipError_t scratch_alloc(void **out, size_t bytes) {
    *out = nullptr;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    const size_t need = (bytes + ROUND - 1) / ROUND * ROUND + (bytes == 0 ? ROUND : 0);
    const size_t slack = need / 4 > ((size_t)32 << 20) ? need / 4 : ((size_t)32 << 20);
    try {
        std::lock_guard<std::mutex> lock(g_mu);
        g_live.reserve(g_live.size() + 1);       // (so that nothing below can throw once a block has changed hands)
        size_t best = g_free.size();
        for (size_t i = 0; i < g_free.size(); ++i)
            if (g_free[i].device == device && g_free[i].cap >= need && g_free[i].cap <= need + slack &&
                (best == g_free.size() || g_free[i].cap < g_free[best].cap))
                best = i;
        if (best != g_free.size()) {
            Block b = g_free[best];
            g_free.erase(g_free.begin() + (long)best);
            g_live.push_back(b);
            *out = b.p;
            return hipSuccess;
        }
    } catch (const std::bad_alloc &) {
        return hipErrorOutOfMemory;
    }
    void *p = nullptr;
    e = hipMalloc(&p, need);
    if (e == hipErrorOutOfMemory) {              // what is kept for later must not be what this request lacks
        (void)hipGetLastError();
        release_scratch();
        e = hipMalloc(&p, need);
    }
    if (e != hipSuccess) return e;
    try {
        std::lock_guard<std::mutex> lock(g_mu);
        g_live.push_back(Block{device, p, need, 0});
    } catch (const std::bad_alloc &) {
        (void)hipFree(p);
        return hipErrorOutOfMemory;
    }
    *out = p;
    return hipSuccess;
}

void scratch_free(void *p, bool keep) {
    if (!p) return;
    std::vector<Block> gone;
    bool found = false;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (size_t i = 0; i < g_live.size(); ++i) {
            if (g_live[i].p != p) continue;
            Block b = g_live[i];
            g_live.erase(g_live.begin() + (long)i);
            found = true;
            const size_t budget = device_budget(b.device);
            if (!keep || b.cap > budget) { gone.push_back(b); break; }
            b.tick = ++g_tick;
            try { g_free.push_back(b); } catch (const std::bad_alloc &) { gone.push_back(b); break; }
            size_t kept = 0;
            for (const Block &f : g_free) if (f.device == b.device) kept += f.cap;
            while (kept > budget) {              // least recently returned first (never the one that just came: it is the newest)
                size_t oldest = g_free.size();
                for (size_t k = 0; k < g_free.size(); ++k)
                    if (g_free[k].device == b.device && (oldest == g_free.size() || g_free[k].tick < g_free[oldest].tick)) oldest = k;
                if (oldest == g_free.size()) break;
                kept -= g_free[oldest].cap;
                try { gone.push_back(g_free[oldest]); } catch (const std::bad_alloc &) { break; }
                g_free.erase(g_free.begin() + (long)oldest);
            }
            break;
        }
    }
    if (!found) { note_cleanup(hipFree(p), "hipFree(block the scratch pool does not know)"); return; }
    give_back(gone);
}

hipError_t scratch_stream(hipStream_t *st, int role) {
    *st = nullptr;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        for (size_t i = 0; i < g_streams.size(); ++i)
            if (g_streams[i].device == device && g_streams[i].role == role) {
                *st = g_streams[i].st;
                g_streams.erase(g_streams.begin() + (long)i);
                return hipSuccess;
            }
    }
    return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
}

void scratch_stream_done(hipStream_t st, int role, bool keep) {
    if (!st) return;
    int device = -1;
    keep = keep && hipStreamGetDevice(st, &device) == hipSuccess;
    if (keep) {
        std::lock_guard<std::mutex> lock(g_mu);
        size_t idle = 0;
        for (const IdleStream &s : g_streams) if (s.device == device && s.role == role) ++idle;
        keep = idle < MAX_IDLE_STREAMS / 2;
        if (keep) { try { g_streams.push_back(IdleStream{device, st, role}); } catch (const std::bad_alloc &) { keep = false; } }
    }
    if (!keep) note_cleanup(hipStreamDestroy(st), "hipStreamDestroy(scratch stream)");
}

}  // namespace wagg

extern "C" int64_t wagg_scratch_bytes(void) { return wagg::scratch_bytes_kept(); }

extern "C" int wagg_release_scratch(void) {
    wagg::clear_error();
    wagg::release_scratch();
    return WAGG_OK;
}

