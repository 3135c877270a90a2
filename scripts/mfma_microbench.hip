// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 on one wave per SIMD, as a function of the number of independent
// accumulators and of VALU instructions interleaved per MFMA.  hipcc --offload-arch=gfx950 -O3 -o mfma_mb mfma_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, int NVALU, bool BSRC_VARY>
__global__ __launch_bounds__(256) void kern(float *out, long long *cyc, int iters, float seed)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f4){seed, seed, seed, seed};
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 48 / NACC; ++u) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, BSRC_VARY ? v[(u + i) & 7] : b, acc[i], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < NVALU; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC, int NVALU, bool V>
void run(const char *name, int blocks)
{
    float *out; long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((kern<NACC, NVALU, V>), dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.0f);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<NACC, NVALU, V>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0f);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 48;
    printf("%-34s blocks=%4d  s_memtime ticks/MFMA = %6.2f   wall ns/MFMA = %6.2f  (=> %.1f cyc @2.4GHz)\n", name, blocks, c / nm, ms * 1e6 / nm, ms * 1e6 / nm * 2.4);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {1, 256}) {
        run<1, 0, false>("1 acc", blocks);
        run<2, 0, false>("2 acc", blocks);
        run<3, 0, false>("3 acc", blocks);
        run<4, 0, false>("4 acc", blocks);
        run<6, 0, false>("6 acc", blocks);
        run<2, 0, true>("2 acc, B operand varies", blocks);
        run<6, 0, true>("6 acc, B operand varies", blocks);
        run<2, 2, false>("2 acc + 2 VALU/MFMA", blocks);
        run<2, 4, false>("2 acc + 4 VALU/MFMA", blocks);
        run<2, 6, false>("2 acc + 6 VALU/MFMA", blocks);
        run<2, 8, false>("2 acc + 8 VALU/MFMA", blocks);
        run<6, 4, false>("6 acc + 4 VALU/MFMA", blocks);
        run<6, 6, false>("6 acc + 6 VALU/MFMA", blocks);
    }
    return 0;
}
