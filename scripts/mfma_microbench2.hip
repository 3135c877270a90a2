// Does VALU work of a SECOND wave on the same SIMD overlap with v_mfma_f32_16x16x4_f32 of the first?
// 512-thread blocks: waves 0-3 (one per SIMD) run an MFMA chain, waves 4-7 a VALU fma chain.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: MFMA waves only work, 1: VALU waves only, 2: both, 3: both kinds of waves run MFMA, 4: both run VALU
__global__ __launch_bounds__(512) void kern(float *out, long long *cyc, int iters, float seed)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool second = wv >= 4;
    f4 acc0 = (f4){seed, seed, seed, seed}, acc1 = acc0;
    float a = seed + threadIdx.x, b = seed * 2;
    float v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, v4 = seed + 4, v5 = seed + 5, v6 = seed + 6, v7 = seed + 7;
    const bool do_mfma = (MODE == 0 && !second) || (MODE == 2 && !second) || MODE == 3;
    const bool do_valu = (MODE == 1 && second) || (MODE == 2 && second) || MODE == 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 32; ++u) { acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0); }
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {   // 8 independent fma chains x 64 = 512 VALU per iteration
                v0 = __builtin_fmaf(v0, 1.0001f, 0.5f); v1 = __builtin_fmaf(v1, 1.0001f, 0.5f); v2 = __builtin_fmaf(v2, 1.0001f, 0.5f); v3 = __builtin_fmaf(v3, 1.0001f, 0.5f);
                v4 = __builtin_fmaf(v4, 1.0001f, 0.5f); v5 = __builtin_fmaf(v5, 1.0001f, 0.5f); v6 = __builtin_fmaf(v6, 1.0001f, 0.5f); v7 = __builtin_fmaf(v7, 1.0001f, 0.5f);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int MODE> void run(const char *name)
{
    float *out; long long *cyc; const int blocks = 256, iters = 1000;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
    hipLaunchKernelGGL((kern<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, 10, 1.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<MODE>), dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-44s wall %.3f ms   ticks wave0 %lld  wave4 %lld   (64 MFMA or 512 VALU per iter, %d iters)\n", name, ms, c[0], c[4], iters);
}
int main()
{
    run<0>("MFMA waves alone (1/SIMD)");
    run<1>("VALU waves alone (1/SIMD)");
    run<2>("MFMA wave + VALU wave on each SIMD");
    run<3>("two MFMA waves per SIMD");
    run<4>("two VALU waves per SIMD");
    return 0;
}
