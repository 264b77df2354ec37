// Time-axis shards on several devices driven by ONE process, device-resident data (SURVEY 8b `wagg_apply_*(..., n_devices)`,
// 8e "prefer one process driving all devices"): shard i -- rows [off_i, off_i + rows_i) of the job, already resident on
// devices[i] -- goes through plans[i] on devices[i], and its (rows_i x R) block lands DIRECTLY in its rows of the result on
// the root device.  No exchange during compute (output row t depends on input row t only); the one exchange at the end is
//   * RCCL over xGMI: grouped ncclSend / ncclRecv between the communicators of ncclCommInitAll (every rank -> root on its
//     own link, the direct gather SURVEY 8e asks for; no ring), or
//   * peer copies (hipMemcpyPeerAsync / hipMemcpy2DAsync on the shard's own stream): what a single process can always do,
//     and the transport when RCCL cannot be loaded or a device is listed twice (several shards on one GPU: how the
//     orchestration -- streams, offsets, ragged blocks -- is tested on a one-GPU box).
// RCCL is not linked: librccl is looked up at run time (the copy the process already holds -- torch's -- if there is one, so
// the process keeps ONE RCCL and one HIP runtime); without it the group uses peer copies.
//
// The process-per-GPU form of the same thing is climate_toolbox_amd/timeshard.py (torch.distributed, backend "nccl" = RCCL);
// the host-resident form is wagg_*_apply_host_multi_* (no exchange at all: every block's result lands in the caller's rows).
#include <dlfcn.h>

#include <rccl/rccl.h>

#include "wagg_dense_int.h"
#include "wagg_sparse_int.h"
#include "wagg_entry.h"

namespace wagg {

struct RcclApi {
    void *so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && GroupStart && GroupEnd && Send && Recv && GetErrorString; }
};

static const RcclApi &rccl_api() {
    static RcclApi api = [] {
        RcclApi a;
        // a copy the process has loaded already first (torch ships one under the soname "librccl.so")
        for (const char *name : {"librccl.so", "librccl.so.1"}) {
            a.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (a.so) break;
        }
        if (!a.so) {
            for (const char *name : {"librccl.so.1", "librccl.so"}) {
                a.so = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (a.so) break;
            }
        }
        if (!a.so) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.so, "ncclCommInitAll");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.so, "ncclCommDestroy");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.so, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.so, "ncclGroupEnd");
        a.Send = (decltype(a.Send))dlsym(a.so, "ncclSend");
        a.Recv = (decltype(a.Recv))dlsym(a.so, "ncclRecv");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.so, "ncclGetErrorString");
        return a;
    }();
    return api;
}

#define WAGG_NCCL(api, expr)                                                                      \
    do {                                                                                          \
        ncclResult_t r__ = (expr);                                                                \
        if (r__ != ncclSuccess) {                                                                 \
            wagg::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (api).GetErrorString(r__)); \
            return WAGG_EHIP;                                                                     \
        }                                                                                         \
    } while (0)

struct DeviceGuard {              // the calling thread's current device is restored on every path
    int home = 0;
    bool ok = false;
    DeviceGuard() { ok = hipGetDevice(&home) == hipSuccess; }
    ~DeviceGuard() { if (ok) (void)hipSetDevice(home); }
};

}  // namespace wagg

struct wagg_shard_group {
    int n = 0;
    int transport = WAGG_GATHER_PEER;                 // what the group uses (never AUTO)
    std::vector<int> devices;
    std::vector<hipStream_t> streams;                 // one per shard, on its device
    std::vector<ncclComm_t> comms;                    // RCCL transport: communicator of rank i = shard i
    std::vector<void *> local;                        // per shard: its block before it travels (grown on demand)
    std::vector<size_t> local_bytes;
    ~wagg_shard_group() {
        wagg::DeviceGuard g;
        for (int i = 0; i < n; ++i) {
            if (hipSetDevice(devices[(size_t)i]) != hipSuccess) continue;
            if (i < (int)streams.size() && streams[(size_t)i]) {
                wagg::note_cleanup(hipStreamSynchronize(streams[(size_t)i]), "hipStreamSynchronize(shard stream)");
                wagg::note_cleanup(hipStreamDestroy(streams[(size_t)i]), "hipStreamDestroy(shard stream)");
            }
            if (i < (int)local.size() && local[(size_t)i]) wagg::note_cleanup(hipFree(local[(size_t)i]), "hipFree(shard block)");
        }
        if (!comms.empty()) {
            const wagg::RcclApi &api = wagg::rccl_api();
            for (ncclComm_t c : comms) if (c && api.CommDestroy) (void)api.CommDestroy(c);
        }
    }
};

extern "C" int wagg_shard_group_create(const int *devices, int n, int transport, wagg_shard_group **out) {
    using namespace wagg;
    clear_error();
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    *out = nullptr;
    WAGG_REQUIRE(devices != nullptr && n >= 1 && n <= 64, "need 1..64 devices");
    WAGG_REQUIRE(transport == WAGG_GATHER_AUTO || transport == WAGG_GATHER_RCCL || transport == WAGG_GATHER_PEER, "unknown transport %d", transport);
    int n_dev = 0;
    WAGG_HIP(hipGetDeviceCount(&n_dev));
    bool distinct = true;
    for (int i = 0; i < n; ++i) {
        WAGG_REQUIRE(devices[i] >= 0 && devices[i] < n_dev, "device %d: this process sees %d device(s)", devices[i], n_dev);
        for (int j = 0; j < i; ++j) distinct &= devices[j] != devices[i];
    }
    const RcclApi &api = rccl_api();
    if (transport == WAGG_GATHER_RCCL) {
        WAGG_REQUIRE(api.ok(), "RCCL transport asked for, but librccl cannot be loaded");
        WAGG_REQUIRE(distinct, "RCCL takes every device once (a communicator per device); list distinct devices or use peer copies");
    }
    std::unique_ptr<wagg_shard_group> g(new (std::nothrow) wagg_shard_group());
    if (!g) { set_error("host allocation failed"); return WAGG_ENOMEM; }
    g->n = n;
    g->devices.assign(devices, devices + n);
    g->streams.assign((size_t)n, nullptr);
    g->local.assign((size_t)n, nullptr);
    g->local_bytes.assign((size_t)n, 0);
    g->transport = (transport == WAGG_GATHER_RCCL || (transport == WAGG_GATHER_AUTO && api.ok() && distinct && n > 1)) ? WAGG_GATHER_RCCL
                                                                                                                   : WAGG_GATHER_PEER;
    DeviceGuard guard;
    for (int i = 0; i < n; ++i) {
        WAGG_HIP(hipSetDevice(devices[i]));
        WAGG_HIP(hipStreamCreateWithFlags(&g->streams[(size_t)i], hipStreamNonBlocking));
    }
    if (g->transport == WAGG_GATHER_RCCL) {
        g->comms.assign((size_t)n, nullptr);
        WAGG_NCCL(api, api.CommInitAll(g->comms.data(), n, devices));
    } else {
        // peer copies: let every shard's device write into every other's memory where the hardware allows it (xGMI); a copy
        // between devices without peer access still works, staged by the runtime
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                if (devices[i] == devices[j]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) continue;
                if (hipSetDevice(devices[i]) != hipSuccess) continue;
                const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
                if (e != hipSuccess) (void)hipGetLastError();          // (already enabled, or refused: the staged copy serves)
            }
    }
    *out = g.release();
    return WAGG_OK;
}

extern "C" int wagg_shard_group_destroy(wagg_shard_group *g) {
    delete g;
    return WAGG_OK;
}

extern "C" int wagg_shard_group_info(const wagg_shard_group *g, int *n_shards, int *transport) {
    using namespace wagg;
    WAGG_REQUIRE(g && n_shards && transport, "NULL argument");
    *n_shards = g->n;
    *transport = g->transport;
    return WAGG_OK;
}

namespace wagg {

// Leaves apply_sharded in a defined state on EVERY exit: an RCCL group that was opened is closed -- a thread that
// returns between ncclGroupStart and ncclGroupEnd would have every later RCCL call it makes (torch.distributed's included)
// silently batched into the open group -- and after a failure every shard stream is drained before the caller gets the status:
// the function is documented as blocking, so the caller may free the result, the fields or the plans as soon as it returns, and
// work queued by the shards that did not fail must not still be using them.  (Declared AFTER the DeviceGuard: it runs first, the
// guard then restores the caller's device.  Statuses are ignored here: the error that is being reported stays the message.)
struct ShardExit {
    wagg_shard_group *g;
    const RcclApi *api = nullptr;
    bool group_open = false;
    bool ok = false;
    explicit ShardExit(wagg_shard_group *g_) : g(g_) {}
    ~ShardExit() {
        if (group_open && api) (void)api->GroupEnd();
        if (ok) return;
        for (int i = 0; i < g->n; ++i) {
            if (hipSetDevice(g->devices[(size_t)i]) != hipSuccess) continue;
            (void)hipStreamSynchronize(g->streams[(size_t)i]);
        }
        (void)hipGetLastError();
    }
};

// apply(i, X_i, rows_i, out_i, ld_i, stream_i) enqueues shard i's kernels on ITS device (current when called)
template <typename T, typename Apply>
static int apply_sharded(wagg_shard_group *g, const int *plan_devices, int32_t R, const T *const *X_dev, const int64_t *rows, T *out_root,
                         int64_t ldo, int root, Apply apply) {
    clear_error();
    WAGG_REQUIRE(g != nullptr, "shard group is NULL");
    WAGG_REQUIRE(X_dev && rows && plan_devices, "NULL argument");
    WAGG_REQUIRE(root >= 0 && root < g->n, "root %d outside the group of %d", root, g->n);
    WAGG_REQUIRE(ldo >= R, "ldo too small");
    const int n = g->n;
    int64_t total = 0;
    std::vector<int64_t> off((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        WAGG_REQUIRE(rows[i] >= 0, "rows[%d] < 0", i);
        WAGG_REQUIRE(plan_devices[i] == g->devices[(size_t)i], "plan %d lives on device %d, the group's shard %d on device %d", i, plan_devices[i],
                     i, g->devices[(size_t)i]);
        WAGG_REQUIRE(rows[i] == 0 || X_dev[i] != nullptr, "X_dev[%d] is NULL", i);
        off[(size_t)i] = total;
        total += rows[i];
    }
    if (total == 0) return WAGG_OK;
    WAGG_REQUIRE(out_root != nullptr, "out_root is NULL");
    const bool rccl = g->transport == WAGG_GATHER_RCCL;
    WAGG_REQUIRE(!rccl || ldo == R || n == 1, "the RCCL transport moves whole blocks: out_root must have contiguous rows (ldo == R)");
    DeviceGuard guard;
    ShardExit leave(g);
    // 1. every shard computes on its own device and stream: the root's block straight into its rows of the result, the others
    //    into a block of their own
    for (int i = 0; i < n; ++i) {
        if (rows[i] == 0) continue;
        WAGG_HIP(hipSetDevice(g->devices[(size_t)i]));
        T *dst = out_root + off[(size_t)i] * ldo;
        int64_t ld = ldo;
        if (i != root) {
            const size_t need = sizeof(T) * (size_t)rows[i] * (size_t)R;
            if (g->local_bytes[(size_t)i] < need) {
                if (g->local[(size_t)i]) {
                    WAGG_HIP(hipStreamSynchronize(g->streams[(size_t)i]));
                    WAGG_HIP(hipFree(g->local[(size_t)i]));
                    g->local[(size_t)i] = nullptr; g->local_bytes[(size_t)i] = 0;
                }
                WAGG_HIP(hipMalloc(&g->local[(size_t)i], need));
                g->local_bytes[(size_t)i] = need;
            }
            dst = static_cast<T *>(g->local[(size_t)i]);
            ld = R;
        }
        if (int rc = apply(i, X_dev[i], rows[i], dst, ld, g->streams[(size_t)i])) return rc;
    }
    // 2. the blocks travel to their rows on the root
    if (rccl && n > 1) {
        const RcclApi &api = rccl_api();
        const ncclDataType_t dt = sizeof(T) == 8 ? ncclFloat64 : ncclFloat32;
        WAGG_NCCL(api, api.GroupStart());
        leave.api = &api;
        leave.group_open = true;
        for (int i = 0; i < n; ++i) {
            if (i == root || rows[i] == 0) continue;
            const size_t count = (size_t)rows[i] * (size_t)R;
            // (ordered behind shard i's kernels on its stream; the receive is ordered behind the root's own kernels, which write
            //  other rows)
            WAGG_NCCL(api, api.Send(g->local[(size_t)i], count, dt, root, g->comms[(size_t)i], g->streams[(size_t)i]));
            WAGG_NCCL(api, api.Recv(out_root + off[(size_t)i] * ldo, count, dt, i, g->comms[(size_t)root], g->streams[(size_t)root]));
        }
        leave.group_open = false;
        WAGG_NCCL(api, api.GroupEnd());
    } else {
        for (int i = 0; i < n; ++i) {
            if (i == root || rows[i] == 0) continue;
            WAGG_HIP(hipSetDevice(g->devices[(size_t)i]));
            T *dst = out_root + off[(size_t)i] * ldo;
            if (ldo == R)
                WAGG_HIP(hipMemcpyPeerAsync(dst, g->devices[(size_t)root], g->local[(size_t)i], g->devices[(size_t)i],
                                            sizeof(T) * (size_t)rows[i] * (size_t)R, g->streams[(size_t)i]));
            else
                WAGG_HIP(hipMemcpy2DAsync(dst, sizeof(T) * (size_t)ldo, g->local[(size_t)i], sizeof(T) * (size_t)R, sizeof(T) * (size_t)R,
                                          (size_t)rows[i], hipMemcpyDeviceToDevice, g->streams[(size_t)i]));
        }
    }
    // 3. blocking, like every multi-device entry point: the result is complete on return
    for (int i = 0; i < n; ++i) {
        WAGG_HIP(hipSetDevice(g->devices[(size_t)i]));
        WAGG_HIP(hipStreamSynchronize(g->streams[(size_t)i]));
    }
    leave.ok = true;
    return WAGG_OK;
}

template <typename T>
static int sparse_sharded(wagg_shard_group *g, const wagg_plan *const *plans, const T *const *X_dev, const int64_t *rows, int64_t ldx,
                          T *out_root, int64_t ldo, int root) {
    WAGG_REQUIRE(g != nullptr && plans != nullptr, "NULL argument");
    std::vector<int> devs((size_t)g->n);
    for (int i = 0; i < g->n; ++i) {
        WAGG_REQUIRE(plans[i] != nullptr, "plan %d is NULL", i);
        WAGG_REQUIRE(plans[i]->info.G == plans[0]->info.G && plans[i]->info.R == plans[0]->info.R, "plan %d has another shape", i);
        devs[(size_t)i] = plans[i]->device;
    }
    WAGG_REQUIRE(ldx >= plans[0]->info.G, "ldx too small");
    return apply_sharded<T>(g, devs.data(), plans[0]->info.R, X_dev, rows, out_root, ldo, root,
                            [&](int i, const T *x, int64_t r, T *o, int64_t ld, hipStream_t st) {
                                if constexpr (sizeof(T) == 4) return entry::apply_f32(plans[i], x, r, ldx, WAGG_LAYOUT_TG, o, ld, WAGG_OUT_TR, st);
                                else return entry::apply_f64(plans[i], x, r, ldx, WAGG_LAYOUT_TG, o, ld, WAGG_OUT_TR, st);
                            });
}

template <typename T>
static int dense_sharded(wagg_shard_group *g, wagg_dense *const *plans, const T *const *X_dev, const int64_t *rows, int64_t ldx, T *out_root,
                         int64_t ldo, int root) {
    WAGG_REQUIRE(g != nullptr && plans != nullptr, "NULL argument");
    std::vector<int> devs((size_t)g->n);
    for (int i = 0; i < g->n; ++i) {
        WAGG_REQUIRE(plans[i] != nullptr, "plan %d is NULL", i);
        WAGG_REQUIRE(plans[i]->G == plans[0]->G && plans[i]->R == plans[0]->R && plans[i]->f64 == (sizeof(T) == 8), "plan %d has another shape or element type", i);
        for (int q = 0; q < i; ++q) WAGG_REQUIRE(plans[q] != plans[i], "shards %d and %d share a plan (a dense-family plan owns its workspaces: one replica per shard)", q, i);
        devs[(size_t)i] = plans[i]->device;
    }
    WAGG_REQUIRE(ldx >= plans[0]->G, "ldx too small");
    return apply_sharded<T>(g, devs.data(), plans[0]->R, X_dev, rows, out_root, ldo, root,
                            [&](int i, const T *x, int64_t r, T *o, int64_t ld, hipStream_t st) {
                                if constexpr (sizeof(T) == 4) return entry::dense_apply_f32(plans[i], x, r, ldx, o, ld, 0, st);
                                else return entry::dense_apply_f64(plans[i], x, r, ldx, o, ld, 0, st);
                            });
}

}  // namespace wagg

int wagg::entry::apply_sharded_f32(wagg_shard_group *g, const wagg_plan *const *plans, const float *const *X_dev, const int64_t *rows,
                                      int64_t ldx, float *out_root, int64_t ldo, int root) {
    return wagg::sparse_sharded<float>(g, plans, X_dev, rows, ldx, out_root, ldo, root);
}
int wagg::entry::apply_sharded_f64(wagg_shard_group *g, const wagg_plan *const *plans, const double *const *X_dev, const int64_t *rows,
                                      int64_t ldx, double *out_root, int64_t ldo, int root) {
    return wagg::sparse_sharded<double>(g, plans, X_dev, rows, ldx, out_root, ldo, root);
}
int wagg::entry::dense_apply_sharded_f32(wagg_shard_group *g, wagg_dense *const *plans, const float *const *X_dev, const int64_t *rows,
                                            int64_t ldx, float *out_root, int64_t ldo, int root) {
    return wagg::dense_sharded<float>(g, plans, X_dev, rows, ldx, out_root, ldo, root);
}
int wagg::entry::dense_apply_sharded_f64(wagg_shard_group *g, wagg_dense *const *plans, const double *const *X_dev, const int64_t *rows,
                                            int64_t ldx, double *out_root, int64_t ldo, int root) {
    return wagg::dense_sharded<double>(g, plans, X_dev, rows, ldx, out_root, ldo, root);
}
