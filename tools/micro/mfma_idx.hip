// Go / no-go microbenchmark for VERDICT r3 item 2b (diagnostic, not part of the library):
//   "evaluate v_mfma_f32_4x4x1_16B_f32 with lanes = timesteps for the entry-list form: can the accumulator group be selected
//    per instruction (VGPR index mode on an MFMA's vdst / src2)?"
// Part 1: correctness -- one v_mfma_f32_4x4x1_16b_f32 under s_set_gpr_idx_on (SRC2 | DST), index 4: does the result land in
//         the accumulator group v[20:23] instead of the encoded v[16:19]?
// Part 2: rate -- the per-slot skeleton such a kernel would run: A = the slot's four weights spread to lanes l % 4
//         (ds_bpermute_b32 from a register that holds 16 slots), B = X[t = lane][cell] (one ds_read_b32 per ~3.7 slots),
//         the MFMA into an accumulator group picked by M0.  Cycles per slot at one and two waves per SIMD, against the same
//         loop without the LDS instructions (MFMA issue alone).
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_idx mfma_idx.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void idx_test(float *out) {
    const int lane = threadIdx.x;
    float a = (float)(lane % 4 + 1), b = 1.0f;
    float r[16];
    asm volatile(
        "v_mov_b32 v16, 0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\t"
        "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, 0\n\tv_mov_b32 v23, 0\n\t"
        "v_mov_b32 v24, 0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, 0\n\tv_mov_b32 v27, 0\n\t"
        "v_mov_b32 v28, 0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, 0\n\tv_mov_b32 v31, 0\n\t"
        "s_mov_b32 s20, 4\n\t"
        "s_set_gpr_idx_on s20, 0xc\n\t"                                   // index SRC2 and DST by M0[7:0] = 4
        "v_mfma_f32_4x4x1_16b_f32 v[16:19], %16, %17, v[16:19]\n\t"
        "s_set_gpr_idx_off\n\t"
        "s_nop 7\n\ts_nop 7\n\t"
        "v_mov_b32 %0, v16\n\tv_mov_b32 %1, v17\n\tv_mov_b32 %2, v18\n\tv_mov_b32 %3, v19\n\t"
        "v_mov_b32 %4, v20\n\tv_mov_b32 %5, v21\n\tv_mov_b32 %6, v22\n\tv_mov_b32 %7, v23\n\t"
        "v_mov_b32 %8, v24\n\tv_mov_b32 %9, v25\n\tv_mov_b32 %10, v26\n\tv_mov_b32 %11, v27\n\t"
        "v_mov_b32 %12, v28\n\tv_mov_b32 %13, v29\n\tv_mov_b32 %14, v30\n\tv_mov_b32 %15, v31\n\t"
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), "=v"(r[8]), "=v"(r[9]),
          "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
        : "v"(a), "v"(b)
        : "memory", "s20", "m0", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29",
          "v30", "v31");
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = r[i];
}

// MODE 0: MFMA + index switch only; 1: + ds_bpermute (A operand); 2: + ds_bpermute + one ds_read_b32 of X per 4 slots
template <int MODE>
__global__ __launch_bounds__(512) void rate_test(float *out, int iters, long long *cycles) {
    extern __shared__ float lds[];                        // 96 KiB requested at launch: one workgroup per CU
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128 * 64; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    int vaddr = lds0 + lane * 4;                         // X row of a cell: 256 B, lane = timestep
    int vperm = (lane & 3) * 4;                          // bpermute source lane (slot 0 of the weight register) * 4
    float w = 1.0f + lane, x = 1.0f, a = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        // 8 slots per iteration, accumulator groups 0, 3, 5, 9, 2, 11, 7, 1 of v[64:111] (12 groups)
        asm volatile(
            ".macro SLOT_M g, poff, xoff, dox, mode\n\t"
            "  .if \\mode >= 1\n\t"
            "    ds_bpermute_b32 %[a], %[vperm], %[w] offset:\\poff\n\t"
            "  .endif\n\t"
            "  .if (\\mode >= 2) && \\dox\n\t"
            "    ds_read_b32 %[x], %[vaddr] offset:\\xoff\n\t"
            "  .endif\n\t"
            "  s_mov_b32 s20, \\g * 4\n\t"
            "  .if \\mode >= 1\n\t"
            "    s_waitcnt lgkmcnt(0)\n\t"
            "  .endif\n\t"
            "  s_set_gpr_idx_on s20, 0xc\n\t"
            "  v_mfma_f32_4x4x1_16b_f32 v[64:67], %[a], %[x], v[64:67]\n\t"
            "  s_set_gpr_idx_off\n\t"
            ".endm\n\t"
            "SLOT_M 0, 0, 0, 1, %[mode]\n\t"
            "SLOT_M 3, 16, 256, 0, %[mode]\n\t"
            "SLOT_M 5, 32, 512, 0, %[mode]\n\t"
            "SLOT_M 9, 48, 768, 0, %[mode]\n\t"
            "SLOT_M 2, 64, 1024, 1, %[mode]\n\t"
            "SLOT_M 11, 80, 1280, 0, %[mode]\n\t"
            "SLOT_M 7, 96, 1536, 0, %[mode]\n\t"
            "SLOT_M 1, 112, 1792, 0, %[mode]\n\t"
            ".purgem SLOT_M\n\t"
            : [a] "+v"(a), [x] "+v"(x)
            : [vperm] "v"(vperm), [w] "v"(w), [vaddr] "v"(vaddr), [mode] "n"(MODE)
            : "memory", "s20", "m0", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77",
              "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94",
              "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109",
              "v110", "v111");
    }
    long long t1 = __builtin_readcyclecounter();
    float s;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_add_f32 %0, v64, v68" : "=v"(s) : : "v64", "v68");
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (s == 12345.678f) out[0] = s + a + x;
}

template <int MODE>
static void run_rate(const char *what, int waves_per_simd) {
    float *out; long long *cyc;
    const int nblk = 256, iters = 20000;
    CK(hipFuncSetAttribute((const void *)rate_test<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    CK(hipMalloc(&out, 4096));
    CK(hipMalloc(&cyc, sizeof(long long) * nblk));
    hipLaunchKernelGGL((rate_test<MODE>), dim3(nblk), dim3(256 * waves_per_simd), 96 * 1024, nullptr, out, iters, cyc);
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((rate_test<MODE>), dim3(nblk), dim3(256 * waves_per_simd), 96 * 1024, nullptr, out, iters, cyc);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    std::vector<long long> h(nblk);
    CK(hipMemcpy(h.data(), cyc, sizeof(long long) * nblk, hipMemcpyDeviceToHost));
    // wall clock -> cycles at 2.4 GHz per slot per SIMD: every SIMD runs waves_per_simd waves x iters x 8 slots
    const double slots_per_simd = (double)waves_per_simd * iters * 8;
    printf("%-58s %d wave(s)/SIMD: %.3f ms -> %.1f cycles per slot and SIMD (2.4 GHz)  [s_memtime ticks wave 0: %lld]\n", what,
           waves_per_simd, ms, ms * 1e-3 * 2.4e9 / slots_per_simd, h[0]);
    CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
    float *out;
    CK(hipMalloc(&out, 64 * 16 * 4));
    hipLaunchKernelGGL(idx_test, dim3(1), dim3(64), 0, nullptr, out);
    CK(hipDeviceSynchronize());
    std::vector<float> h(64 * 16);
    CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    int nz[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 16; ++i) if (h[l * 16 + i] != 0.f) nz[i / 4]++;
    printf("part 1: non-zero results per accumulator group v[16:19] v[20:23] v[24:27] v[28:31]: %d %d %d %d\n", nz[0], nz[1], nz[2], nz[3]);
    printf("        lane 5, group 1 registers: %g %g %g %g (expected a[i] * b = 1 2 3 4 if the index applies to vdst/src2)\n", h[5 * 16 + 4],
           h[5 * 16 + 5], h[5 * 16 + 6], h[5 * 16 + 7]);
    printf("        => VGPR index mode on v_mfma_f32_4x4x1_16b_f32 vdst/src2: %s\n", nz[1] == 256 && nz[0] == 0 ? "WORKS" : "DOES NOT APPLY");
    for (int w = 1; w <= 2; ++w) {
        run_rate<0>("part 2: MFMA + index switch only", w);
        run_rate<1>("part 2: + ds_bpermute_b32 (A = the slot's 4 weights)", w);
        run_rate<2>("part 2: + ds_bpermute_b32 + ds_read_b32 of X per 4 slots", w);
    }
    return 0;
}
