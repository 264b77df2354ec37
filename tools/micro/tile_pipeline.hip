// Microbenchmark (diagnostic): how much non-load work per stage can a 2-workgroups-per-CU tile
// gather afford before HBM bandwidth drops?  Persistent workgroups stream tiles of ROWS x 1 KB
// (row stride = grid row) through LDS with optional busy-work per stage.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: load -> sum (no LDS).  1: load -> LDS park -> barrier -> busy(work) -> barrier (no overlap)
//      2: software pipelined: issue next tile's loads, then barrier + busy(work) on the parked tile
template <int MODE, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k(const float* __restrict__ X, long T, long G, long nchunk,
                                                long ntile, int work, float* out) {
    constexpr int RPW = 64 / WAVES;                       // rows per wave (64-row tiles)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0, 0, 0, 0};
    f4 v[RPW];
    auto issue = [&](long b) {
        const long c = b % nchunk, tb = b / nchunk;
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            long t = tb * 64 + wave * RPW + i;
            t = t < T ? t : T - 1;
            v[i] = *reinterpret_cast<const f4*>(X + t * G + c * 256 + lane * 4);
        }
    };
    auto busy = [&](int n) {                               // LDS-read + FMA chain standing in for the reduction
        float a = acc[0];
        for (int i = 0; i < n; ++i) a = fmaf(a, 1.0001f, lds[((i * 67 + lane) & 16383)]);
        acc[0] = a;
    };
    long b = blockIdx.x;
    if (MODE == 2 && b < ntile) issue(b);
    for (; b < ntile; b += gridDim.x) {
        if (MODE != 2) issue(b);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < RPW; ++i) acc += v[i];
            continue;
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) *reinterpret_cast<f4*>(&lds[(wave * RPW + i) * 260 + lane * 4]) = v[i];
        if (MODE == 2 && b + gridDim.x < ntile) issue(b + gridDim.x);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        busy(work);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

// MODE 3: wave-specialised: LW loader waves fill image buffer (i+1)&1 (global -> regs -> LDS) while CW
// consumer waves run busy(work) on buffer i&1; one workgroup barrier per tile.
template <int LW, int CW, int EX = 0, int PAT = 0>
__global__ __launch_bounds__((LW + CW) * 64) void k3(const float* __restrict__ X, long T, long G, long nchunk,
                                                     long ntile, int work, float* out) {
    constexpr int RPW = 64 / LW;
    extern __shared__ __attribute__((aligned(16))) float lds[];     // 2 x [64][260]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc = 0.f;
    long b = blockIdx.x;
    int buf = 0;
    auto load_tile = [&](long bb, int bf) {
        const long c = bb % nchunk, tb = bb / nchunk;
        f4 v[RPW];
        float ex[EX > 0 ? EX : 1];
#pragma unroll
        for (int i = 0; i < EX; ++i) ex[i] = X[(c * 977 + i * 131 + threadIdx.x) & 0xfffff];   // small L2-resident loads first
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            long t = tb * 64 + wave * RPW + i;
            t = t < T ? t : T - 1;
            long cell = c * 256 + lane * 4;
            if (PAT == 1) {                                   // 8 grid rows x 32 cells: 8 pieces of 128 B per instruction
                const long pr = c / 45, pc = c % 45;          // 1440 / 32 = 45 patches per band, 90 bands of 8 rows
                cell = (pr * 8 + (lane >> 3)) * 1440 + pc * 32 + (lane & 7) * 4;
            }
            if (PAT == 2) {                                   // same patches, runs shifted off the 128-B lines
                const long pr = c / 45, pc = c % 45;
                const long r = lane >> 3;
                long col = pc * 32 + (lane & 7) * 4 + ((r * 12 + 4) & 28);
                col = col < 1436 ? col : 1436;
                cell = (pr * 8 + r) * 1440 + col;
            }
            v[i] = *reinterpret_cast<const f4*>(X + t * G + cell);
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i) *reinterpret_cast<f4*>(&lds[bf * 64 * 260 + (wave * RPW + i) * 260 + lane * 4]) = v[i];
#pragma unroll
        for (int i = 0; i < EX; ++i) lds[2 * 64 * 260 - 1 - ((threadIdx.x + i * 768) & 1023)] += ex[i] * 0.f;
    };
    if (wave < LW && b < ntile) load_tile(b, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (; b < ntile; b += gridDim.x) {
        if (wave < LW) {
            if (b + gridDim.x < ntile) load_tile(b + gridDim.x, buf ^ 1);
        } else {
            const float* im = lds + buf * 64 * 260;
            float a = acc;
            for (int i = 0; i < work; ++i) a = fmaf(a, 1.0001f, im[((i * 67 + lane) & 16383)]);
            acc = a;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        buf ^= 1;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const long T = 365, G = 1036800;
    float *X, *out;
    CK(hipMalloc(&X, sizeof(float) * T * G));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 0, sizeof(float) * T * G));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const long nchunk = G / 256, ntile = nchunk * ((T + 63) / 64);
    const size_t shmem = 75 * 1024;
    auto run = [&](const char* name, auto kern, int threads, int grid, int work) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shmem, 0, X, T, G, nchunk, ntile, work, out);
        CK(hipEventRecord(a));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), shmem, 0, X, T, G, nchunk, ntile, work, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-34s thr %4d grid %5d work %5d : %.3f ms  %.2f TB/s\n", name, threads, grid, work, ms, sizeof(float) * T * G / ms * 1e-9);
    };
    run("mode0 stream (75KB LDS -> 2 WG/CU)", k<0, 4>, 256, 512, 0);
    for (int w : {0, 100, 300, 1000})
        run("mode1 load->park->work, 4 waves", k<1, 4>, 256, 512, w);
    for (int w : {0, 100, 300, 1000, 3000})
        run("mode2 pipelined, 4 waves", k<2, 4>, 256, 512, w);
    for (int w : {0, 100, 300, 1000, 3000})
        run("mode2 pipelined, 8 waves", k<2, 8>, 512, 512, w);
    auto run3 = [&](const char* name, auto kern, int threads, int grid, int work) {
        const size_t sh = 2 * 64 * 260 * 4;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh));
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), sh, 0, X, T, G, nchunk, ntile, work, out);
        CK(hipEventRecord(a));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), sh, 0, X, T, G, nchunk, ntile, work, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-34s thr %4d grid %5d work %5d : %.3f ms  %.2f TB/s\n", name, threads, grid, work, ms, sizeof(float) * T * G / ms * 1e-9);
    };
    run3("mode3 8L+4C, 8x128B patches", k3<8, 4, 0, 1>, 768, 256, 0);
    run3("mode3 8L+4C, misaligned runs", k3<8, 4, 0, 2>, 768, 256, 0);
    run3("mode3 8L+4C, 0 extra loads", k3<8, 4, 0>, 768, 256, 0);
    run3("mode3 8L+4C, 4 extra loads", k3<8, 4, 4>, 768, 256, 0);
    run3("mode3 8L+4C, 8 extra loads", k3<8, 4, 8>, 768, 256, 0);
    run3("mode3 8L+4C, 16 extra loads", k3<8, 4, 16>, 768, 256, 0);
    for (int w : {0, 300, 1000}) run3("mode3 4 loaders + 4 consumers", k3<4, 4>, 512, 256, w);
    for (int w : {0, 300, 1000}) run3("mode3 8 loaders + 4 consumers", k3<8, 4>, 768, 256, w);
    for (int w : {0, 300, 1000}) run3("mode3 8 loaders + 8 consumers", k3<8, 8>, 1024, 256, w);
    return 0;
}
