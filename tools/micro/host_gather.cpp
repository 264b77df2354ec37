// Go / no-go for a "lines only" host path: how fast can N host threads compact the referenced runs of every
// row of a pageable (T, G) field into page-locked staging, and does the DMA of the compact rows keep the PCIe rate?
//   runs.bin: int64 n_runs, then n_runs x (int64 first_cell, int64 n_cells) -- written by tools/host_gather_probe.py
// build: hipcc -O2 -mavx2 -pthread --offload-arch=gfx950 -o /tmp/host_gather tools/micro/host_gather.cpp
// usage: host_gather runs.bin T G elem_bytes threads [nt: 0 = memcpy, 1 = 16-byte loads + streaming stores, 2 = 32-byte AVX2 loads + streaming stores, 3 / 4 = 1 / 2 with software prefetch of the source six runs ahead] [slots: ring pieces in (c), 0 = whole field]
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char **argv) {
    if (argc < 6) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 3;
    long long n_runs = 0;
    if (fread(&n_runs, 8, 1, f) != 1) return 3;
    std::vector<long long> rr((size_t)n_runs * 2);
    if (fread(rr.data(), 16, (size_t)n_runs, f) != (size_t)n_runs) return 3;
    fclose(f);
    const long long T = atoll(argv[2]), G = atoll(argv[3]), eb = atoll(argv[4]);
    const int nth = atoi(argv[5]);
    const int nt = argc > 6 ? atoi(argv[6]) : 0;
    const long long slots = argc > 7 ? atoll(argv[7]) : 0;
    std::vector<long long> soff((size_t)n_runs), doff((size_t)n_runs), len((size_t)n_runs);
    long long Gc = 0;
    for (long long k = 0; k < n_runs; ++k) { soff[(size_t)k] = rr[(size_t)k * 2] * eb; len[(size_t)k] = rr[(size_t)k * 2 + 1] * eb; doff[(size_t)k] = Gc * eb; Gc += rr[(size_t)k * 2 + 1]; }
    const size_t xbytes = (size_t)(T * G * eb), cbytes = (size_t)(T * Gc * eb);
    char *X = (char *)malloc(xbytes);                      // pageable, touched
    for (size_t i = 0; i < xbytes; i += 4096) X[i] = (char)i;
    memset(X, 1, xbytes);
    char *P = nullptr, *D = nullptr;
    CK(hipHostMalloc((void **)&P, cbytes, hipHostMallocDefault));
    CK(hipMalloc((void **)&D, xbytes));
    memset(P, 0, cbytes);
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    printf("runs %lld  compact %.3f of the row  avg run %.0f B  compact field %.3f GB  threads %d\n", n_runs, (double)Gc / (double)G,
           (double)(Gc * eb) / (double)n_runs, (double)cbytes * 1e-9, nth);
    auto copy_nt = [](char *d, const char *s, size_t n) {       // d 16-byte aligned, n a multiple of 16
        for (size_t i = 0; i < n; i += 16) _mm_stream_si128((__m128i *)(d + i), _mm_loadu_si128((const __m128i *)(s + i)));
    };
    auto copy_nt2 = [](char *d, const char *s, size_t n) {      // 32-byte loads, two 16-byte streaming stores (nt = 2)
        size_t i = 0;
        for (; i + 32 <= n; i += 32) {
            const __m256i v = _mm256_loadu_si256((const __m256i *)(s + i));
            _mm_stream_si128((__m128i *)(d + i), _mm256_castsi256_si128(v));
            _mm_stream_si128((__m128i *)(d + i + 16), _mm256_extracti128_si256(v, 1));
        }
        if (i < n) _mm_stream_si128((__m128i *)(d + i), _mm_loadu_si128((const __m128i *)(s + i)));
    };
    auto gather_to = [&](char *d, long long r) {
        const char *s = X + (size_t)(r * G * eb);
        if (nt == 3 || nt == 4) {                   // nt 1 / nt 2 with the source of the run PF runs ahead prefetched (two cache lines)
            constexpr long long PF = 6;
            for (long long k = 0; k < n_runs; ++k) {
                if (k + PF < n_runs) { _mm_prefetch(s + soff[(size_t)(k + PF)], _MM_HINT_NTA); _mm_prefetch(s + soff[(size_t)(k + PF)] + 64, _MM_HINT_NTA); }
                if (nt == 3) copy_nt(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]);
                else copy_nt2(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]);
            }
            _mm_sfence();
        } else if (nt == 2) { for (long long k = 0; k < n_runs; ++k) copy_nt2(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]); _mm_sfence(); }
        else if (nt) { for (long long k = 0; k < n_runs; ++k) copy_nt(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]); _mm_sfence(); }
        else for (long long k = 0; k < n_runs; ++k) memcpy(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]);
    };
    auto gather_rows = [&](long long r0, long long r1) {
        for (long long r = r0; r < r1; ++r) gather_to(P + (size_t)(r * Gc * eb), r);
    };
    for (int rep = 0; rep < 3; ++rep) {
        // (a) gather alone
        double t0 = now();
        {
            std::atomic<long long> next{0};
            std::vector<std::thread> th;
            for (int w = 0; w < nth; ++w) th.emplace_back([&]() { for (;;) { long long r = next.fetch_add(1); if (r >= T) break; gather_rows(r, r + 1); } });
            for (auto &t : th) t.join();
        }
        double t1 = now();
        // (b) DMA of the compact field alone
        CK(hipMemcpyAsync(D, P, cbytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        // (c) both, pipelined in pieces of 8 rows
        const long long RPP = 8, npieces = (T + RPP - 1) / RPP;
        std::vector<std::atomic<int>> done((size_t)npieces);
        for (auto &d : done) d.store(0);
        std::atomic<long long> next{0};
        std::vector<std::thread> th;
        double t3 = now();
        std::atomic<long long> free_upto{slots > 0 ? slots : npieces};
        std::vector<hipEvent_t> ev((size_t)(slots > 0 ? slots : 1));
        for (auto &e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        const size_t pb = (size_t)(RPP * Gc * eb);
        for (int w = 0; w < nth; ++w) th.emplace_back([&]() { for (;;) { long long r = next.fetch_add(1); if (r >= T) break;
            const long long p = r / RPP;
            while (p >= free_upto.load()) std::this_thread::yield();
            gather_to(P + (size_t)((slots > 0 ? p % slots : p)) * pb + (size_t)((r % RPP) * Gc * eb), r);
            done[(size_t)p].fetch_add(1); } });
        for (long long p = 0; p < npieces; ++p) {
            const long long rows = (p + 1) * RPP <= T ? RPP : T - p * RPP;
            while (done[(size_t)p].load() < rows) std::this_thread::yield();
            CK(hipMemcpyAsync(D + (size_t)(p * RPP * Gc * eb), P + (size_t)((slots > 0 ? p % slots : p)) * pb, (size_t)(rows * Gc * eb), hipMemcpyHostToDevice, st));
            if (slots > 0) {
                CK(hipEventRecord(ev[(size_t)(p % slots)], st));
                if (p >= 1) { CK(hipEventSynchronize(ev[(size_t)((p - 1) % slots)])); free_upto.store(p + slots); }
            }
        }
        CK(hipStreamSynchronize(st));
        double t4 = now();
        for (auto &t : th) t.join();
        // (d) reference: DMA of the whole field from registered memory
        CK(hipHostRegister(X, xbytes, hipHostRegisterDefault));
        double t5 = now();
        CK(hipMemcpyAsync(D, X, xbytes, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t6 = now();
        CK(hipHostUnregister(X));
        printf("gather %.2f ms (%.1f GB/s out)  dma compact %.2f ms (%.1f GB/s)  pipelined %.2f ms  |  whole field dma %.2f ms (%.1f GB/s)\n",
               (t1 - t0) * 1e3, (double)cbytes / (t1 - t0) * 1e-9, (t2 - t1) * 1e3, (double)cbytes / (t2 - t1) * 1e-9, (t4 - t3) * 1e3,
               (t6 - t5) * 1e3, (double)xbytes / (t6 - t5) * 1e-9);
        fflush(stdout);
    }
    return 0;
}
