// Device-side plan building for caller-supplied weight tables (see wagg_build.h): scan, stable radix sort, key
// generation from COO / CSR, coalescing of duplicate (cell, region) rows, denominators -- and the synthetic c5 tables in
// CSR form (wagg_synth_table_csr), the benchmark's stand-in for a caller's file.
//
// The reference's only weights type is a caller-supplied table (aggregations.py:64-73, :128-152); BASELINE configs[4]
// calls it "sparse CSR weights (<= 1 % nnz)": 2.53e8 entries.  Rounds 1-3 sorted such a table on the host with two
// single-threaded std::stable_sort passes; here the host only uploads it.  Round 5: a build runs on a stream of its
// own out of one arena (BuildCtx, wagg_build.h) and waits only where the host reads a count back.
#include <utility>

#include "wagg_build.h"
#include <chrono>
#include <cstdlib>
#include <mutex>
#include <new>
#include <vector>
#include "wagg_host.h"

namespace wagg {

// The arena and the stream come from the scratch pool (wagg_scratch.hip): a build of a configs[4]-sized table takes 11 GB,
// and handing that back to the driver after every build made every twelfth build wait seconds in hipMalloc.
hipError_t BuildCtx::init(size_t arena_bytes) {
    cap = (arena_bytes + 255) / 256 * 256;
    if (cap == 0) cap = 256;
    top = peak = 0;
    high = cap;
    hipError_t e = scratch_stream(&st);
    if (e != hipSuccess) return e;
    return scratch_alloc(reinterpret_cast<void **>(&base), cap);
}

void *BuildCtx::take_bytes(size_t bytes) {
    const size_t need = (bytes + 255) / 256 * 256;
    if (!base || need > high - top) return nullptr;
    void *p = base + top;
    top += need;
    if (top + (cap - high) > peak) peak = top + (cap - high);
    return p;
}

void *BuildCtx::take_input_bytes(size_t bytes) {
    const size_t need = (bytes + 255) / 256 * 256;
    if (!base || need > high - top) return nullptr;
    high -= need;
    if (top + (cap - high) > peak) peak = top + (cap - high);
    return base + high;
}

BuildCtx::~BuildCtx() {
    // nothing of the build may still be running on the arena (error paths return without having waited)
    if (st) note_cleanup(hipStreamSynchronize(st), "hipStreamSynchronize(build stream)");
    scratch_free(base);
    scratch_stream_done(st);
}

constexpr int RS_TILE_ = 512 * 16;
static size_t scan_scratch_bytes(int64_t n) { return ((size_t)((n + 2047) / 2048) * sizeof(uint32_t) + 255) / 256 * 256 + 256; }

size_t build_arena_bytes(int64_t n, int64_t G, int32_t R, bool csr) {
    const size_t a256 = 256;
    auto up = [&](size_t b) { return (b + a256 - 1) / a256 * a256; };
    const size_t pairs = 4 * up(sizeof(uint64_t) * (size_t)n);
    const size_t table = csr ? up(sizeof(int64_t) * ((size_t)G + 1)) + up(sizeof(int32_t) * (size_t)n) + up(sizeof(double) * (size_t)n)
                             : 2 * up(sizeof(int32_t) * (size_t)n) + up(sizeof(double) * (size_t)n);
    const int64_t NB = (n + RS_TILE_ - 1) / RS_TILE_;
    const size_t hist = up(sizeof(uint32_t) * 256 * (size_t)NB) + scan_scratch_bytes(256 * NB);
    const size_t rank = up(sizeof(uint32_t) * (size_t)n) + scan_scratch_bytes(n) + 2 * a256;
    const size_t after_keys = hist > rank ? hist : rank;
    (void)R;
    return pairs + (table > after_keys ? table : after_keys) + 16 * a256;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of uint32 (three kernels: tile sums, one block over the sums, tiles again)
// ---------------------------------------------------------------------------------------------
constexpr int SC_THREADS = 256, SC_PER = 8, SC_TILE = SC_THREADS * SC_PER;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}

// exclusive prefix of v over the workgroup (any whole number of waves <= 16) and the workgroup's total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *lds /* [16] */, uint32_t &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const uint32_t inc = wave_incl_scan(v, lane);
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    total = 0;
    for (int w = 0; w < nw; ++w) {
        const uint32_t s = lds[w];
        if (w < wave) base += s;
        total += s;
    }
    __syncthreads();
    return base + inc - v;
}

__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(const uint32_t *__restrict__ data, int64_t n,
                                                               uint32_t *__restrict__ sums) {
    __shared__ uint32_t lds[16];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_PER;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < SC_PER; ++k)
        if (base + k < n) s += data[base + k];
    uint32_t total;
    (void)block_excl_scan(s, lds, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void scan_top_kernel(uint32_t *__restrict__ sums, int64_t m, uint32_t *__restrict__ total_out) {
    __shared__ uint32_t lds[16];
    uint32_t carry = 0;
    for (int64_t b = 0; b < m; b += 1024) {
        const int64_t i = b + threadIdx.x;
        const uint32_t v = i < m ? sums[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan(v, lds, tot);
        if (i < m) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(SC_THREADS) void scan_apply_kernel(uint32_t *__restrict__ data, int64_t n,
                                                                const uint32_t *__restrict__ sums) {
    __shared__ uint32_t lds[16];
    const int64_t base = (int64_t)blockIdx.x * SC_TILE + (int64_t)threadIdx.x * SC_PER;
    uint32_t v[SC_PER], s = 0;
#pragma unroll
    for (int k = 0; k < SC_PER; ++k) {
        v[k] = base + k < n ? data[base + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan(s, lds, total) + sums[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SC_PER; ++k) {
        if (base + k < n) data[base + k] = ex;
        ex += v[k];
    }
}

int scan_u32_exclusive(BuildCtx &ctx, uint32_t *data, int64_t n, uint32_t *total_dev) {
    if (n <= 0) {
        if (total_dev) WAGG_HIP(hipMemsetAsync(total_dev, 0, sizeof(uint32_t), ctx.st));
        return WAGG_OK;
    }
    const int64_t m = (n + SC_TILE - 1) / SC_TILE;
    WAGG_REQUIRE(m < (int64_t)0x7fffffff, "scan too long");
    const size_t mk = ctx.mark();
    uint32_t *sums;
    WAGG_TAKE(sums, ctx, uint32_t, m);
    hipLaunchKernelGGL(scan_sums_kernel, dim3((unsigned)m), dim3(SC_THREADS), 0, ctx.st, (const uint32_t *)data, n, sums);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(1024), 0, ctx.st, sums, m, total_dev);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)m), dim3(SC_THREADS), 0, ctx.st, data, n, (const uint32_t *)sums);
    WAGG_HIP(hipGetLastError());
    ctx.release_to(mk);                          // (whatever takes this place next runs behind these kernels on ctx.st)
    return WAGG_OK;
}

// ---------------------------------------------------------------------------------------------
// stable LSD radix sort of (uint64 key, uint64 value) pairs, 8 bits per pass
// ---------------------------------------------------------------------------------------------
// A workgroup owns a tile of 8,192 consecutive pairs, wave w the 1,024 at [1024 w, 1024 w + 1024): round r of a wave is
// the 64 consecutive pairs at 64 r (coalesced).  Pass = histogram per (digit, tile) -> exclusive scan over the digit-major
// table -> scatter, where a pair's place is  scanned[digit][tile] + (same digit in lower waves of the tile) + (same digit
// in earlier rounds of its wave) + (same digit in lower lanes of its round): input order is kept among equal digits.
constexpr int RS_THREADS = 512, RS_WAVES = RS_THREADS / 64, RS_ROUNDS = 16, RS_TILE = RS_THREADS * RS_ROUNDS;
static_assert(RS_TILE == RS_TILE_, "build_arena_bytes sizes the histograms by the sort tile");

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t *__restrict__ keys, int64_t n, int shift,
                                                             uint32_t *__restrict__ hist, int64_t NB) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
    for (int k = 0; k < RS_ROUNDS; ++k) {
        const int64_t i = base + (int64_t)k * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(unsigned)(keys[i] >> shift) & 255u], 1u);      // integer counts: order-free
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(int64_t)threadIdx.x * NB + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                                int64_t n, int shift, const uint32_t *__restrict__ offs, int64_t NB,
                                                                uint64_t *__restrict__ keys_out, uint64_t *__restrict__ vals_out) {
    __shared__ uint32_t cnt[RS_WAVES][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < RS_WAVES * 256; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();
    const int64_t wbase = (int64_t)blockIdx.x * RS_TILE + (int64_t)wave * (64 * RS_ROUNDS) + lane;
    uint64_t k[RS_ROUNDS], v[RS_ROUNDS];
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const int64_t i = wbase + 64 * r;
        const bool ok = i < n;
        k[r] = ok ? keys[i] : 0ull;
        v[r] = ok ? vals[i] : 0ull;
        if (ok) atomicAdd(&cnt[wave][(unsigned)(k[r] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) {                       // counts -> first place of (wave, digit)
        uint32_t run = offs[(int64_t)threadIdx.x * NB + blockIdx.x];
#pragma unroll
        for (int w = 0; w < RS_WAVES; ++w) {
            const uint32_t c = cnt[w][threadIdx.x];
            cnt[w][threadIdx.x] = run;
            run += c;
        }
    }
    __syncthreads();
    const uint64_t below_mask = (1ull << lane) - 1ull;
    // Rounds of a wave hand the running place of (wave, digit) on through cnt[wave][d]: every lane of round r reads it, the
    // lowest peer of each digit then stores the new place, round r + 1 reads that.  The lanes are different threads of the
    // memory model, so the hand-over is spelled out: relaxed wavefront-scope atomics for the accesses (nothing may be cached
    // in a register or forwarded across rounds), a wave barrier between the reads and the store, and release / acquire
    // fences at wavefront scope around the barrier that separates the store from the next round's reads.  The stable
    // order of the sort -- and with it the bit-for-bit plan -- no longer rests on how a compiler schedules plain LDS code.
#pragma unroll
    for (int r = 0; r < RS_ROUNDS; ++r) {
        const bool ok = wbase + 64 * r < n;
        const unsigned d = (unsigned)(k[r] >> shift) & 255u;
        uint64_t peers = __ballot(ok);             // lanes of this round with my digit
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        uint32_t first = 0;
        if (ok) first = __hip_atomic_load(&cnt[wave][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();           // every lane of the round holds its count before any lane replaces one
        if (ok) {
            const uint32_t at = first + (uint32_t)__popcll(peers & below_mask);
            keys_out[at] = k[r];
            vals_out[at] = v[r];
            if ((peers & below_mask) == 0)
                __hip_atomic_store(&cnt[wave][d], first + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

#ifdef WAGG_DIAG
void build_stamp(const BuildCtx &ctx, const char *what) {
    static const bool on = getenv("WAGG_BUILD_TRACE") != nullptr;
    static thread_local double last = 0.0;
    if (!on) return;
    (void)ctx.sync();
    const double now = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (what[0] == '^') last = now;
    std::fprintf(stderr, "[build] %-28s %8.3f ms\n", what, (now - last) * 1e3);
    last = now;
}
#endif

int radix_sort_pairs(BuildCtx &ctx, uint64_t *keys, uint64_t *vals, uint64_t *keys_alt, uint64_t *vals_alt, int64_t n, int passes) {
    if (n <= 1 || passes <= 0) return WAGG_OK;
    WAGG_REQUIRE(n < (int64_t)0x7fffffff, "too many entries to sort (%lld)", (long long)n);
    const int64_t NB = (n + RS_TILE - 1) / RS_TILE;
    const size_t mk = ctx.mark();
    uint32_t *hist;
    WAGG_TAKE(hist, ctx, uint32_t, 256 * NB);
    uint64_t *ks = keys, *vs = vals, *kd = keys_alt, *vd = vals_alt;
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(rs_hist_kernel, dim3((unsigned)NB), dim3(RS_THREADS), 0, ctx.st, (const uint64_t *)ks, n, 8 * p, hist, NB);
        WAGG_HIP(hipGetLastError());
        if (int rc = scan_u32_exclusive(ctx, hist, 256 * NB, nullptr)) return rc;
        hipLaunchKernelGGL(rs_scatter_kernel, dim3((unsigned)NB), dim3(RS_THREADS), 0, ctx.st, (const uint64_t *)ks, (const uint64_t *)vs, n,
                           8 * p, (const uint32_t *)hist, NB, kd, vd);
        WAGG_HIP(hipGetLastError());
        std::swap(ks, kd);
        std::swap(vs, vd);
    }
    if (ks != keys) {
        WAGG_HIP(hipMemcpyAsync(keys, ks, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx.st));
        WAGG_HIP(hipMemcpyAsync(vals, vs, sizeof(uint64_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx.st));
    }
    ctx.release_to(mk);
    return WAGG_OK;
}

static int passes_for(uint64_t range) {          // 8-bit digits covering [0, range] (range itself = the sentinel's floor)
    int bits = 0;
    while (bits < 64 && (range >> bits) != 0) ++bits;
    return bits == 0 ? 1 : (bits + 7) / 8;
}

// ---------------------------------------------------------------------------------------------
// table -> sorted keys
// ---------------------------------------------------------------------------------------------
constexpr uint64_t KEY_DROPPED = ~0ull;           // null label / NaN weight: sorts behind every real key

__global__ __launch_bounds__(256) void keygen_kernel(const int32_t *__restrict__ cell, const int64_t *__restrict__ rowptr,
                                                     const int32_t *__restrict__ region, const double *__restrict__ w, int64_t n,
                                                     int64_t G, int32_t R, EntryKeyGeom geom, uint64_t *__restrict__ keys,
                                                     uint64_t *__restrict__ vals, unsigned long long *__restrict__ note) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool dropped = false;
    // CSR: the row that holds entry i = the last g with rowptr[g] <= i.  The block's 256 consecutive entries lie in the rows
    // between those of its first and its last entry: two threads search all of rowptr for these, the rest search between them
    // (a step or two for tables with hundreds of entries per row, where a search of its own per entry was most of this kernel)
    __shared__ int64_t row_span[2];
    auto row_of = [&](int64_t e, int64_t lo, int64_t hi) {      // rowptr[lo] <= e < rowptr[hi]
        while (hi - lo > 1) {
            const int64_t mid = (lo + hi) >> 1;
            if (rowptr[mid] <= e) lo = mid; else hi = mid;
        }
        return lo;
    };
    if (rowptr) {
        if (threadIdx.x < 2) {
            const int64_t first = (int64_t)blockIdx.x * blockDim.x, last = first + blockDim.x - 1 < n ? first + blockDim.x - 1 : n - 1;
            row_span[threadIdx.x] = row_of(threadIdx.x ? last : first, 0, G);
        }
        __syncthreads();
    }
    if (i < n) {
        int64_t c;
        if (rowptr) {
            c = row_of(i, row_span[0], row_span[1] + 1);
        } else {
            c = cell[i];
        }
        const int32_t r = region[i];
        const double wv = w[i];
        // CSR rows whose columns ascend are already in (cell, region) order: the one-pass partition below relies on it
        if (rowptr && i > rowptr[c] && region[i - 1] > r) note[2] = 1ull;
        uint64_t key = KEY_DROPPED;
        if (r >= R || c < 0 || c >= G) atomicMin(&note[0], (unsigned long long)i);           // first bad row
        else if (r >= 0 && wv == wv) key = geom.key(c, r);
        dropped = key == KEY_DROPPED;
        keys[i] = key;
        vals[i] = __builtin_bit_cast(uint64_t, wv);
    }
    const uint64_t m = __ballot(dropped);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(&note[1], (unsigned long long)__popcll(m));
}

__global__ __launch_bounds__(256) void heads_kernel(const uint64_t *__restrict__ key, int64_t n, uint32_t *__restrict__ flag) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = (i == 0 || key[i] != key[i - 1]) ? 1u : 0u;
}

// rows of one (cell, region) pair are neighbours now, in input order: add them in that order (S5)
__global__ __launch_bounds__(256) void coalesce_kernel(const uint64_t *__restrict__ key, const uint64_t *__restrict__ val, int64_t n,
                                                       const uint32_t *__restrict__ rank, uint64_t *__restrict__ ukey,
                                                       double *__restrict__ uw, uint64_t *__restrict__ rkey, EntryKeyGeom geom) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    if (i > 0 && key[i - 1] == k) return;
    double s = 0.0;
    for (int64_t j = i; j < n && key[j] == k; ++j) s += __builtin_bit_cast(double, val[j]);
    const uint32_t u = rank[i];
    ukey[u] = k;
    uw[u] = s;
    int64_t cellv;
    int32_t regionv;
    int cic, jj;
    geom.decode(k, cellv, regionv, cic, jj);
    rkey[u] = (uint64_t)regionv;
}

// ---------------------------------------------------------------------------------------------
// CSR tables with ascending columns: the sort in ONE stable pass
// ---------------------------------------------------------------------------------------------
// Key order is (region block, chunk, wave, cell in chunk, region in wave); such a table arrives in (chunk, cell in chunk,
// region block, wave, region in wave) order.  Inside one chunk the pairs of one (region block, wave) bin are therefore
// already in key order, and the whole sort is a stable partition of every chunk's run into its n_rb x 16 bins whose places
// come from one exclusive scan over the counts laid out [region block][chunk][wave] -- the buckets of the key.  One
// workgroup per chunk (its pairs are the contiguous run rowptr[128 c] .. rowptr[128 (c + 1)]), tiles of 8,192 pairs walked
// in order with the running place of every bin in LDS; inside a tile the rounds of rs_scatter_kernel.
constexpr int CP_BINS_MAX = 1024;

__global__ __launch_bounds__(256) void chunk_first_kernel(const int64_t *__restrict__ rowptr, int64_t G, int n_chunks,
                                                          uint32_t *__restrict__ cf, unsigned long long *__restrict__ note) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c > n_chunks) return;
    const int64_t g0 = c * 128 < G ? c * 128 : G, g1 = (c + 1) * 128 < G ? (c + 1) * 128 : G;
    cf[c] = (uint32_t)rowptr[g0];
    if (c < n_chunks) atomicMax(&note[3], (unsigned long long)(rowptr[g1] - rowptr[g0]));
}

__device__ __forceinline__ unsigned chunk_bin(uint64_t key, const EntryKeyGeom &geom, int64_t chunk) {
    const uint64_t b = (uint64_t)geom.bucket_of(key);            // (rb n_chunks + chunk) 16 + wave
    return (unsigned)((((b >> 4) - (uint64_t)chunk) / (uint64_t)geom.n_chunks) * 16u + (b & 15u));
}

__global__ __launch_bounds__(RS_THREADS) void chunk_hist_kernel(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ cf,
                                                                EntryKeyGeom geom, uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[CP_BINS_MAX];
    const int nbins = geom.n_rb * 16;
    const int64_t c = blockIdx.x;
    for (int b = threadIdx.x; b < nbins; b += RS_THREADS) h[b] = 0;
    __syncthreads();
    const int64_t end = cf[c + 1];
    for (int64_t i = (int64_t)cf[c] + threadIdx.x; i < end; i += RS_THREADS) atomicAdd(&h[chunk_bin(keys[i], geom, c)], 1u);
    __syncthreads();
    for (int b = threadIdx.x; b < nbins; b += RS_THREADS)
        hist[((int64_t)(b >> 4) * geom.n_chunks + c) * 16 + (b & 15)] = h[b];
}

__global__ __launch_bounds__(RS_THREADS) void chunk_scatter_kernel(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ vals,
                                                                   const uint32_t *__restrict__ cf, EntryKeyGeom geom,
                                                                   const uint32_t *__restrict__ offs, uint64_t *__restrict__ keys_out,
                                                                   uint64_t *__restrict__ vals_out) {
    __shared__ uint32_t cnt[RS_WAVES][CP_BINS_MAX];
    __shared__ uint32_t run[CP_BINS_MAX];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nbins = geom.n_rb * 16;
    const int64_t c = blockIdx.x, end = cf[c + 1];
    for (int b = threadIdx.x; b < nbins; b += RS_THREADS) run[b] = offs[((int64_t)(b >> 4) * geom.n_chunks + c) * 16 + (b & 15)];
    const uint64_t below_mask = (1ull << lane) - 1ull;
    for (int64_t tile = cf[c]; tile < end; tile += RS_TILE) {
        for (int w = 0; w < RS_WAVES; ++w)
            for (int b = threadIdx.x; b < nbins; b += RS_THREADS) cnt[w][b] = 0;
        __syncthreads();                           // (also: run[] of the tile before is complete)
        const int64_t wbase = tile + (int64_t)wave * (64 * RS_ROUNDS) + lane;
        uint64_t k[RS_ROUNDS], v[RS_ROUNDS];
        uint16_t d[RS_ROUNDS];
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const int64_t i = wbase + 64 * r;
            const bool ok = i < end;
            k[r] = ok ? keys[i] : 0ull;
            v[r] = ok ? vals[i] : 0ull;
            d[r] = ok ? (uint16_t)chunk_bin(k[r], geom, c) : (uint16_t)0;
            if (ok) atomicAdd(&cnt[wave][d[r]], 1u);
        }
        __syncthreads();
        for (int b = threadIdx.x; b < nbins; b += RS_THREADS) {      // counts -> first place of (wave, bin) in this tile
            uint32_t at = run[b];
#pragma unroll
            for (int w = 0; w < RS_WAVES; ++w) {
                const uint32_t n_w = cnt[w][b];
                cnt[w][b] = at;
                at += n_w;
            }
            run[b] = at;
        }
        __syncthreads();
        // the rounds of rs_scatter_kernel (see there for the hand-over of cnt[wave][bin] between rounds), ten bin bits
#pragma unroll
        for (int r = 0; r < RS_ROUNDS; ++r) {
            const bool ok = wbase + 64 * r < end;
            const unsigned bin = d[r];
            uint64_t peers = __ballot(ok);
#pragma unroll
            for (int b = 0; b < 10; ++b) {
                const bool bit = (bin >> b) & 1u;
                const uint64_t m = __ballot(bit);
                peers &= bit ? m : ~m;
            }
            uint32_t first = 0;
            if (ok) first = __hip_atomic_load(&cnt[wave][bin], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            if (ok) {
                const uint32_t at = first + (uint32_t)__popcll(peers & below_mask);
                keys_out[at] = k[r];
                vals_out[at] = v[r];
                if ((peers & below_mask) == 0)
                    __hip_atomic_store(&cnt[wave][bin], first + (uint32_t)__popcll(peers), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        __syncthreads();                           // every wave is through with cnt[] before the next tile clears it
    }
}

// den[r] = sum of the weights of region r's pairs (aggregations.py:79), pairs in (region, key) order: one wave per
// region, lane-strided partial sums, then the fixed shuffle tree
__global__ __launch_bounds__(256) void den_kernel(const uint64_t *__restrict__ rkey, const uint64_t *__restrict__ w, int64_t n,
                                                  int32_t R, double *__restrict__ den) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    auto lower = [&](uint64_t v) {
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (rkey[mid] < v) lo = mid + 1; else hi = mid;
        }
        return lo;
    };
    const int64_t b = lower((uint64_t)r), e = lower((uint64_t)r + 1);
    double s = 0.0;
    for (int64_t i = b + lane; i < e; i += 64) s += __builtin_bit_cast(double, w[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if (lane == 0) den[r] = s;
}

size_t chunk_sort_scratch_bytes(const EntryKeyGeom &geom) {
    const size_t nb = (size_t)geom.n_rb * 16 * (size_t)geom.n_chunks;
    return sizeof(uint32_t) * (nb + (size_t)geom.n_chunks + 1) + scan_scratch_bytes((int64_t)nb) + 2048;
}

int build_sorted_entries(BuildCtx &ctx, const int32_t *cell_dev, const int64_t *rowptr_dev, const int32_t *region_dev,
                         const double *w_dev, int64_t n, int64_t G, int32_t R, const EntryKeyGeom &geom, SortedEntries *out,
                         bool general_sort) {
    WAGG_REQUIRE(out != nullptr, "out is NULL");
    WAGG_REQUIRE(n >= 0 && n < (int64_t)0x7fffffff, "table of %lld rows: at most 2^31 - 1", (long long)n);
    WAGG_REQUIRE((cell_dev != nullptr) != (rowptr_dev != nullptr) || n == 0, "exactly one of cell / rowptr");
    WAGG_REQUIRE(geom.range() < (1ull << 62), "grid x regions too large for the sort key");
    out->geom = geom;
    out->n_in = n;
    out->n_valid = out->n_u = 0;
    WAGG_HIP(out->den.alloc((size_t)R));
    WAGG_HIP(hipMemsetAsync(out->den.p, 0, sizeof(double) * (size_t)R, ctx.st));
    if (n == 0) return WAGG_OK;
    const unsigned nblk = (unsigned)((n + 255) / 256);
    const size_t mk = ctx.mark();
    uint64_t *ka, *va, *kb, *vb;
    unsigned long long *note;
    WAGG_TAKE(ka, ctx, uint64_t, n);
    WAGG_TAKE(va, ctx, uint64_t, n);
    WAGG_TAKE(kb, ctx, uint64_t, n);
    WAGG_TAKE(vb, ctx, uint64_t, n);
    WAGG_TAKE(note, ctx, unsigned long long, 4);
    {
        // first row out of range | rows dropped | CSR columns out of order somewhere | longest chunk of a CSR table
        const unsigned long long init[4] = {~0ull, 0ull, 0ull, 0ull};
        WAGG_HIP(staged_h2d(note, init, sizeof(init), ctx.st));
    }
    const int nbins = geom.n_rb * 16;
    const int64_t n_buckets = (int64_t)nbins * geom.n_chunks;
    const bool chunkwise_possible = rowptr_dev != nullptr && nbins <= CP_BINS_MAX && !general_sort;
    uint32_t *cf = nullptr, *chist = nullptr;
    if (chunkwise_possible) {
        WAGG_TAKE(cf, ctx, uint32_t, geom.n_chunks + 1);
        WAGG_TAKE(chist, ctx, uint32_t, n_buckets);
        hipLaunchKernelGGL(chunk_first_kernel, dim3((unsigned)(geom.n_chunks / 256 + 1)), dim3(256), 0, ctx.st, rowptr_dev, G,
                           geom.n_chunks, cf, note);
    }
    hipLaunchKernelGGL(keygen_kernel, dim3(nblk), dim3(256), 0, ctx.st, cell_dev, rowptr_dev, region_dev, w_dev, n, G, R, geom,
                       ka, va, note);
    WAGG_HIP(hipGetLastError());
    ctx.drop_inputs();                           // the table has been read once this kernel is through (stream order)
    unsigned long long noted[4];
    WAGG_HIP(staged_d2h(noted, note, sizeof(noted), ctx.st));      // (waits for the copy, hence for the kernel: the host reads here)
    WAGG_REQUIRE(noted[0] == ~0ull, "segment %llu out of range", noted[0]);
    const int64_t n_valid = n - (int64_t)noted[1];
    out->n_valid = n_valid;
    WAGG_BUILD_STAMP(ctx, "keygen");
    // one pass when the table is CSR with ascending columns, drops nothing and no chunk holds so much of it that its one
    // workgroup would be the build (a chunk walks its tiles in order); the general sort otherwise
    const int64_t longest_ok = std::max<int64_t>(64 * (int64_t)RS_TILE, 8 * (n / geom.n_chunks + 1));
    out->chunkwise = chunkwise_possible && noted[1] == 0 && noted[2] == 0 && (int64_t)noted[3] <= longest_ok;
    if (out->chunkwise) {
        hipLaunchKernelGGL(chunk_hist_kernel, dim3((unsigned)geom.n_chunks), dim3(RS_THREADS), 0, ctx.st, (const uint64_t *)ka,
                           (const uint32_t *)cf, geom, chist);
        WAGG_HIP(hipGetLastError());
        if (int rc = scan_u32_exclusive(ctx, chist, n_buckets, nullptr)) return rc;
        hipLaunchKernelGGL(chunk_scatter_kernel, dim3((unsigned)geom.n_chunks), dim3(RS_THREADS), 0, ctx.st, (const uint64_t *)ka,
                           (const uint64_t *)va, (const uint32_t *)cf, geom, (const uint32_t *)chist, kb, vb);
        WAGG_HIP(hipGetLastError());
        std::swap(ka, kb);
        std::swap(va, vb);
    } else if (int rc = radix_sort_pairs(ctx, ka, va, kb, vb, n, passes_for(geom.range()))) {
        return rc;
    }
    WAGG_BUILD_STAMP(ctx, "sort by key");
    if (n_valid == 0) { ctx.release_to(mk); return WAGG_OK; }
    // distinct pairs: rank of every run's head, then one sum per run
    uint32_t *rank, *total;
    WAGG_TAKE(rank, ctx, uint32_t, n_valid);
    WAGG_TAKE(total, ctx, uint32_t, 1);
    const unsigned vblk = (unsigned)((n_valid + 255) / 256);
    hipLaunchKernelGGL(heads_kernel, dim3(vblk), dim3(256), 0, ctx.st, (const uint64_t *)ka, n_valid, rank);
    WAGG_HIP(hipGetLastError());
    if (int rc = scan_u32_exclusive(ctx, rank, n_valid, total)) return rc;
    uint32_t n_u32 = 0;
    WAGG_HIP(staged_d2h(&n_u32, total, sizeof(n_u32), ctx.st));    // (the host sizes the output from it)
    const int64_t n_u = n_u32;
    out->n_u = n_u;
    WAGG_BUILD_STAMP(ctx, "heads + scan");
    WAGG_HIP(out->key.alloc((size_t)n_u));
    WAGG_HIP(out->w.alloc((size_t)n_u));
    // kb <- region of every distinct pair (the key of the denominator sort), vb <- a copy of the sums to sort along
    hipLaunchKernelGGL(coalesce_kernel, dim3(vblk), dim3(256), 0, ctx.st, (const uint64_t *)ka, (const uint64_t *)va, n_valid,
                       (const uint32_t *)rank, out->key.p, out->w.p, kb, geom);
    WAGG_HIP(hipGetLastError());
    WAGG_HIP(hipMemcpyAsync(vb, out->w.p, sizeof(double) * (size_t)n_u, hipMemcpyDeviceToDevice, ctx.st));
    WAGG_BUILD_STAMP(ctx, "alloc + coalesce");
    if (int rc = radix_sort_pairs(ctx, kb, vb, ka, va, n_u, passes_for((uint64_t)R))) return rc;
    WAGG_BUILD_STAMP(ctx, "sort by region");
    hipLaunchKernelGGL(den_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, ctx.st, (const uint64_t *)kb, (const uint64_t *)vb,
                       n_u, R, out->den.p);
    WAGG_HIP(hipGetLastError());
    ctx.release_to(mk);                          // (reused only by later work on ctx.st, i.e. behind den_kernel)
    return WAGG_OK;
}

// ---------------------------------------------------------------------------------------------
// synthetic c5 tables in CSR form (device-generated, copied to the caller's host arrays)
// ---------------------------------------------------------------------------------------------
// W[g][r] = hash_u01(g R + r, seed) where hash_u01(g R + r, seed ^ 0x9e3779b9) < fill; block-local: only the 256 regions of
// column tile (97 (g / 64)) mod ceil(R / 256) are candidates of cell g.  Exactly the tables of
// wagg_dense_create_synth_sparse / _synth_blocklocal, as a caller would hand them in: rows = cells, columns ascending.
template <bool FILL>
__global__ __launch_bounds__(256) void synth_csr_kernel(int64_t G, int32_t R, uint32_t seed, float fill, int blocklocal,
                                                        uint32_t *__restrict__ counts, const uint32_t *__restrict__ first,
                                                        int32_t *__restrict__ col, double *__restrict__ val) {
    const int lane = threadIdx.x & 63;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int n_nt = (int)(((int64_t)R + 255) / 256);
    for (int64_t g = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); g < G; g += nwaves) {
        int64_t r0 = 0, r1 = R;
        if (blocklocal) {
            r0 = (int64_t)((97 * (g / 64)) % n_nt) * 256;
            r1 = r0 + 256 < R ? r0 + 256 : R;
        }
        uint32_t kept = 0;
        const uint32_t base = FILL ? first[g] : 0u;
        for (int64_t rb = r0; rb < r1; rb += 64) {
            const int64_t r = rb + lane;
            const uint64_t id = (uint64_t)g * (uint64_t)R + (uint64_t)r;
            const bool keep = r < r1 && hash_u01(id, seed ^ 0x9e3779b9u) < fill;
            const uint64_t m = __ballot(keep);
            if (FILL && keep) {
                const uint32_t at = base + kept + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                col[at] = (int32_t)r;
                val[at] = (double)hash_u01(id, seed);
            }
            kept += (uint32_t)__popcll(m);
        }
        if (!FILL && lane == 0) counts[g] = kept;
    }
}

__global__ void widen_rowptr_kernel(const uint32_t *__restrict__ first, int64_t G, uint32_t total, int64_t *__restrict__ rowptr) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) rowptr[g] = first[g];
    if (g == G) rowptr[g] = total;
}

}  // namespace wagg

using namespace wagg;

extern "C" int wagg_synth_table_csr(int64_t G, int32_t R, uint32_t seed, double fill, int blocklocal, int64_t *rowptr_host,
                                    int32_t *col_host, double *val_host, int64_t capacity, int64_t *nnz_out) {
    clear_error();
    WAGG_REQUIRE(G > 0 && R > 0 && fill > 0.0 && fill <= 1.0, "bad arguments");
    WAGG_REQUIRE(rowptr_host != nullptr && nnz_out != nullptr, "NULL pointer");
    WAGG_REQUIRE((col_host == nullptr) == (val_host == nullptr), "col and val go together");
    BuildCtx ctx;
    WAGG_HIP(ctx.init(sizeof(uint32_t) * (size_t)G + sizeof(int64_t) * ((size_t)G + 1) + scan_scratch_bytes(G) + 4096));
    uint32_t *first, *total;
    WAGG_TAKE(first, ctx, uint32_t, G);
    WAGG_TAKE(total, ctx, uint32_t, 1);
    hipLaunchKernelGGL((synth_csr_kernel<false>), dim3(256 * 32), dim3(256), 0, ctx.st, G, R, seed, (float)fill, blocklocal, first,
                       (const uint32_t *)nullptr, (int32_t *)nullptr, (double *)nullptr);
    WAGG_HIP(hipGetLastError());
    if (int rc = scan_u32_exclusive(ctx, first, G, total)) return rc;
    uint32_t nnz = 0;
    WAGG_HIP(staged_d2h(&nnz, total, sizeof(nnz), ctx.st));
    *nnz_out = nnz;
    int64_t *rowptr;
    WAGG_TAKE(rowptr, ctx, int64_t, G + 1);
    hipLaunchKernelGGL(widen_rowptr_kernel, dim3((unsigned)((G + 256) / 256)), dim3(256), 0, ctx.st, (const uint32_t *)first, G, nnz,
                       rowptr);
    WAGG_HIP(hipGetLastError());
    WAGG_HIP(staged_d2h(rowptr_host, rowptr, sizeof(int64_t) * ((size_t)G + 1), ctx.st));
    if (!col_host) return WAGG_OK;
    WAGG_REQUIRE(capacity >= (int64_t)nnz, "capacity %lld below the table's %u entries", (long long)capacity, nnz);
    if (nnz == 0) return WAGG_OK;
    DevBuf<int32_t> col;
    DevBuf<double> val;
    WAGG_HIP(col.alloc(nnz));
    WAGG_HIP(val.alloc(nnz));
    hipLaunchKernelGGL((synth_csr_kernel<true>), dim3(256 * 32), dim3(256), 0, ctx.st, G, R, seed, (float)fill, blocklocal,
                       (uint32_t *)nullptr, (const uint32_t *)first, col.p, val.p);
    WAGG_HIP(hipGetLastError());
    WAGG_HIP(ctx.sync());
    if (int rc = copy_rows_to_host(col_host, col.p, 1, sizeof(int32_t) * (size_t)nnz, sizeof(int32_t) * (size_t)nnz, true)) return rc;
    return copy_rows_to_host(val_host, val.p, 1, sizeof(double) * (size_t)nnz, sizeof(double) * (size_t)nnz, true);
}
