// Microbenchmark (diagnostic): read bandwidth of X[T][G] fp32 under the sparse kernel's access shape:
// a workgroup reads TBLK rows x (64 lanes x 16 B) with the row stride of the grid, vs a flat stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// mode 0: flat stream: block b reads a contiguous 64 KB
// mode 1: tile: block (chunk c, tblock tb): wave w reads rows tb*64 + w*16 + i, columns c*256 .. +255 (1 KB)
template <int MODE, int ROWS_PER_WAVE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ X, long T, long G, long nchunk, float* out, long nblk_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc = {0, 0, 0, 0};
    for (long b = blockIdx.x; b < nblk_total; b += gridDim.x) {
        f4 v[ROWS_PER_WAVE];
        if (MODE == 0) {
            const f4* p = reinterpret_cast<const f4*>(X) + b * (256L * ROWS_PER_WAVE) ;
#pragma unroll
            for (int i = 0; i < ROWS_PER_WAVE; ++i) v[i] = p[(wave * ROWS_PER_WAVE + i) * 64 + lane];
        } else {
            const long c = b % nchunk, tb = b / nchunk;
#pragma unroll
            for (int i = 0; i < ROWS_PER_WAVE; ++i) {
                long t = tb * (4 * ROWS_PER_WAVE) + wave * ROWS_PER_WAVE + i;
                t = t < T ? t : T - 1;
                v[i] = *reinterpret_cast<const f4*>(X + t * G + c * 256 + lane * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < ROWS_PER_WAVE; ++i) acc += v[i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}

int main() {
    const long T = 365, G = 1036800;
    float *X, *out;
    CK(hipMalloc(&X, sizeof(float) * T * G));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(X, 0, sizeof(float) * T * G));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const long nchunk = G / 256;
    auto run = [&](const char* name, auto kern, long nblk_total, int grid) {
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, X, T, G, nchunk, out, nblk_total);
        CK(hipEventRecord(a));
        const int reps = 10;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, X, T, G, nchunk, out, nblk_total);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
        printf("%-44s grid %6d  %.3f ms  %.2f TB/s\n", name, grid, ms, sizeof(float) * T * G / ms * 1e-9);
    };
    const long nb16 = T * G / (256L * 16 * 4);           // 64 KB blocks
    for (int grid : {512, 1024, 2048, 4096, (int)nb16})
        run("flat stream, 64 KB/WG, 16 x 16 B per lane", k<0, 16>, nb16, grid);
    const long ntb16 = (T + 63) / 64;
    for (int grid : {512, 1024, 2048, 4096, (int)(nchunk * ntb16)})
        run("tile 64 rows x 1 KB (stride 4 MB), 16/lane", k<1, 16>, nchunk * ntb16, grid);
    const long ntb8 = (T + 31) / 32;
    for (int grid : {1024, 2048, 4096, (int)(nchunk * ntb8)})
        run("tile 32 rows x 1 KB, 8 loads/lane", k<1, 8>, nchunk * ntb8, grid);
    const long ntb4 = (T + 15) / 16;
    for (int grid : {2048, 4096, 8192, (int)(nchunk * ntb4)})
        run("tile 16 rows x 1 KB, 4 loads/lane", k<1, 4>, nchunk * ntb4, grid);
    return 0;
}
