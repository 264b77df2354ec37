// Diagnostic: how long do a large hipMalloc / hipFree / stream create take when repeated?  (A plan build allocates one
// arena of ~11 GB for a c5 table and frees it at the end; one build in several was seen to wait 0.7-4 s in that step.)
// build: hipcc -O2 --offload-arch=gfx950 -o /tmp/malloc_stall tools/diag/malloc_stall.cpp
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const size_t bytes = (size_t)((argc > 1 ? atof(argv[1]) : 11.0) * (double)(1ull << 30));      // GiB, fractions allowed
    const int reps = argc > 2 ? atoi(argv[2]) : 12;
    const int touch = argc > 3 ? atoi(argv[3]) : 1;
    (void)hipFree(nullptr);
    for (int i = 0; i < reps; ++i) {
        hipStream_t st;
        double t0 = now();
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        double t1 = now();
        void *p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        double t2 = now();
        if (touch) { hipMemsetAsync(p, 1, bytes, st); hipStreamSynchronize(st); }
        double t3 = now();
        hipFree(p);
        double t4 = now();
        hipStreamDestroy(st);
        double t5 = now();
        if ((t2 - t1) < 0.05 && i > 0 && i + 1 < reps && reps > 40) continue;      // long runs: the stalls only
        printf("%2d  stream %.2f ms  malloc %.2f ms (%s)  memset %.2f ms  free %.2f ms  destroy %.2f ms\n", i, (t1 - t0) * 1e3,
               (t2 - t1) * 1e3, hipGetErrorString(e), (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3);
        fflush(stdout);
    }
    return 0;
}
