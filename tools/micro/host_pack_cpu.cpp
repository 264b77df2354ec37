// CPU-only: how fast can N host threads pack the referenced runs of every row of a pageable (T, G) field (the "lines only" /
// "quads only" host path, csrc/wagg_host.hip GatherTeam::work) -- variants of the copy loop.  No GPU call.
//   runs.bin: int64 n_runs, then n_runs x (int64 first_cell, int64 n_cells)   (tools/host_gather_probe.py)
// build: g++ -O2 -mavx2 -pthread -o /tmp/host_pack_cpu tools/micro/host_pack_cpu.cpp
// usage: host_pack_cpu runs.bin T G elem_bytes threads
// variants: 0 = 16-byte loads + streaming stores (what the library does), 1 = the same + software prefetch of the source of
// the run PF runs ahead, 2 = 32-byte AVX2 loads + two 16-byte streaming stores, 3 = memcpy (regular stores), 4 = variant 1
// with rows dealt in pairs (two rows per claim: the run table is walked once for both)
#include <immintrin.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
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
    std::vector<long long> soff((size_t)n_runs), doff((size_t)n_runs);
    std::vector<int> len((size_t)n_runs);
    long long Gc = 0;
    for (long long k = 0; k < n_runs; ++k) { soff[(size_t)k] = rr[(size_t)k * 2] * eb; len[(size_t)k] = (int)(rr[(size_t)k * 2 + 1] * eb); doff[(size_t)k] = Gc * eb; Gc += rr[(size_t)k * 2 + 1]; }
    const size_t xbytes = (size_t)(T * G * eb), crow = (size_t)(Gc * eb);
    char *X = (char *)malloc(xbytes);
    memset(X, 1, xbytes);
    const long long RPP = 8, SLOTS = 4;
    char *ring = (char *)aligned_alloc(4096, (size_t)(RPP * SLOTS) * crow + 4096);
    memset(ring, 0, (size_t)(RPP * SLOTS) * crow);
    printf("runs %lld  compact %.3f of the row  avg run %.0f B  packed field %.3f GB  threads %d\n", n_runs, (double)Gc / (double)G, (double)crow / (double)n_runs,
           (double)(crow * (size_t)T) * 1e-9, nth);
    constexpr int PF = 6;
    auto row_v0 = [&](char *d, const char *s) {
        for (long long k = 0; k < n_runs; ++k) {
            char *dd = d + doff[(size_t)k]; const char *ss = s + soff[(size_t)k];
            for (int i = 0; i < len[(size_t)k]; i += 16) _mm_stream_si128((__m128i *)(dd + i), _mm_loadu_si128((const __m128i *)(ss + i)));
        }
        _mm_sfence();
    };
    auto row_v1 = [&](char *d, const char *s) {
        for (long long k = 0; k < n_runs; ++k) {
            if (k + PF < n_runs) { _mm_prefetch(s + soff[(size_t)(k + PF)], _MM_HINT_NTA); _mm_prefetch(s + soff[(size_t)(k + PF)] + 64, _MM_HINT_NTA); }
            char *dd = d + doff[(size_t)k]; const char *ss = s + soff[(size_t)k];
            for (int i = 0; i < len[(size_t)k]; i += 16) _mm_stream_si128((__m128i *)(dd + i), _mm_loadu_si128((const __m128i *)(ss + i)));
        }
        _mm_sfence();
    };
    auto row_v2 = [&](char *d, const char *s) {
        for (long long k = 0; k < n_runs; ++k) {
            char *dd = d + doff[(size_t)k]; const char *ss = s + soff[(size_t)k];
            const int n = len[(size_t)k];
            int i = 0;
            for (; i + 32 <= n; i += 32) {
                const __m256i v = _mm256_loadu_si256((const __m256i *)(ss + i));
                _mm_stream_si128((__m128i *)(dd + i), _mm256_castsi256_si128(v));
                _mm_stream_si128((__m128i *)(dd + i + 16), _mm256_extracti128_si256(v, 1));
            }
            if (i < n) _mm_stream_si128((__m128i *)(dd + i), _mm_loadu_si128((const __m128i *)(ss + i)));
        }
        _mm_sfence();
    };
    auto row_v3 = [&](char *d, const char *s) {
        for (long long k = 0; k < n_runs; ++k) memcpy(d + doff[(size_t)k], s + soff[(size_t)k], (size_t)len[(size_t)k]);
    };
    for (int variant = 0; variant < 5; ++variant) {
        double best = 1e9;
        for (int rep = 0; rep < 4; ++rep) {
            std::atomic<long long> next{0};
            std::vector<std::thread> th;
            const long long step = variant == 4 ? 2 : 1;
            const double t0 = now();
            for (int w = 0; w < nth; ++w) th.emplace_back([&]() {
                for (;;) {
                    const long long r = next.fetch_add(step);
                    if (r >= T) break;
                    for (long long q = r; q < r + step && q < T; ++q) {
                        char *d = ring + (size_t)(q % (RPP * SLOTS)) * crow;
                        const char *s = X + (size_t)q * (size_t)(G * eb);
                        switch (variant) { case 0: row_v0(d, s); break; case 1: case 4: row_v1(d, s); break; case 2: row_v2(d, s); break; default: row_v3(d, s); }
                    }
                }
            });
            for (auto &t : th) t.join();
            const double dt = now() - t0;
            if (dt < best) best = dt;
        }
        printf("variant %d: %.2f ms  (%.1f GB/s packed)\n", variant, best * 1e3, (double)(crow * (size_t)T) / best * 1e-9);
        fflush(stdout);
    }
    return 0;
}
