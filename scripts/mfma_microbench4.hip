// Feasibility of computing the rollout's fp32 layers on the bf16 matrix pipe with an exact 3-way split (a = a0 + a1 + a2, 8 mantissa
// bits each; the 6 products with i + j <= 2 keep every term down to 2^-24 relative — fp32 grade): one "hidden stage" per iteration,
// dependent through relu -> split like the real kernel, weights streamed from L2.
//   fp32 : 64 v_mfma_f32_16x16x4_f32                      (2 out blocks x 8 k blocks x 4)
//   x6/x9: 48 / 72 v_mfma_f32_16x16x32_bf16               (2 out blocks x 4 K-chunks of 32 x 6 or 9 products)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(const f4 x0, const f4 x1, u4 (&p)[3])     // 8 values -> 3 pieces of 8 bf16 (truncation split: exact)
{
    float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
    unsigned a0[8], a1[8], a2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const unsigned u = __float_as_uint(v[i]);
        a0[i] = u & 0xFFFF0000u;
        const float r1 = v[i] - __uint_as_float(a0[i]);
        a1[i] = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(a1[i]);
        a2[i] = __float_as_uint(r2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        p[0][i] = __builtin_amdgcn_perm(a0[2 * i + 1], a0[2 * i], 0x07060302u);     // hi halves of two floats -> one dword
        p[1][i] = __builtin_amdgcn_perm(a1[2 * i + 1], a1[2 * i], 0x07060302u);
        p[2][i] = __builtin_amdgcn_perm(a2[2 * i + 1], a2[2 * i], 0x07060302u);
    }
}

template <int MODE>   // 0 fp32, 6 / 9 products
__global__ __launch_bounds__(768) void kern(const u4 *w, float *out, long long *cyc, int iters)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<u4 *>(w + (size_t)(blockIdx.x % 5) * 65536 + (size_t)(wv & 3) * 8192), 0, 8192 * 16, 0x00020000);
    f4 acc0 = {0.1f * lane, 0.2f, 0.3f, 0.4f}, acc1 = {0.5f, 0.6f, 0.7f, 0.01f * lane};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f4 h0, h1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { h0[r] = fmaxf(acc0[r], 0.f) * 0.5f + 0.25f; h1[r] = fmaxf(acc1[r], 0.f) * 0.5f + 0.125f; }
        if (MODE == 0) {
            acc0 = (f4){0.01f, 0.02f, 0.03f, 0.04f}; acc1 = acc0;
#pragma unroll
            for (int F = 0; F < 8; ++F) {
                const f4 a = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, F * 2048, 0));
                const f4 b = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16 + 1024, F * 2048, 0));
                const f4 hb = (F & 1) ? h1 : h0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], hb[r], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[r], hb[r], acc1, 0, 0, 0);
                }
            }
        } else {
            u4 p[3];
            split3(h0, h1, p);
            acc0 = (f4){0.01f, 0.02f, 0.03f, 0.04f}; acc1 = acc0;
#pragma unroll
            for (int C = 0; C < 4; ++C) {
                u4 wa[3], wb[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    wa[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (C * 6 + i) * 1024, 0);
                    wb[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (C * 6 + 3 + i) * 1024, 0);
                }
#pragma unroll
                for (int i = 2; i >= 0; --i)
#pragma unroll
                    for (int j = 2; j >= 0; --j) {
                        if (MODE == 6 && i + j > 2) continue;
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wa[i]), __builtin_bit_cast(bf8, p[j]), acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, wb[i]), __builtin_bit_cast(bf8, p[j]), acc1, 0, 0, 0);
                    }
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 768 + threadIdx.x] = acc0[0] + acc1[1];
    if (lane == 0) cyc[blockIdx.x * 16 + wv] = t1 - t0;
}

template <int M> void run(const char *name, int threads)
{
    u4 *w; float *out; long long *cyc; const int blocks = 256, iters = 2000;
    const size_t wb = 5 * 65536 * 16 + (1 << 20);
    hipMalloc(&w, wb);
    unsigned *hw = (unsigned *)malloc(wb); srand(1);
    for (size_t i = 0; i < wb / 4; ++i) {        // small random values (as bf16 pairs or floats: both finite, magnitude ~1e-2)
        const unsigned e = 0x3C00u + (rand() & 0xFF);
        hw[i] = M == 0 ? ((e << 16) | (rand() & 0xFFFF)) : ((e << 16) | (0x3C00u + (rand() & 0xFF)));
    }
    hipMemcpy(w, hw, wb, hipMemcpyHostToDevice); free(hw);
    hipMalloc(&out, blocks * 768 * 4); hipMalloc(&cyc, blocks * 128);
    hipLaunchKernelGGL((kern<M>), dim3(blocks), dim3(threads), 0, 0, w, out, cyc, 10);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<M>), dim3(blocks), dim3(threads), 0, 0, w, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[16]; hipMemcpy(c, cyc, 128, hipMemcpyDeviceToHost);
    const int waves_per_simd = threads / 256;
    printf("%-28s %d wave(s)/SIMD: %7.0f ticks per stage per wave, wall %.3f ms = %.3f us per stage per SIMD-slot-set (x%d waves)\n", name, waves_per_simd,
           (double)c[0] / iters, ms, ms * 1e3 / iters, waves_per_simd);
    hipFree(w); hipFree(out); hipFree(cyc);
}
int main()
{
    for (int t = 256; t <= 768; t += 256) { run<0>("fp32 16x16x4 (64 MFMA)", t); run<6>("bf16x6 16x16x32 (48 MFMA)", t); run<9>("bf16x9 16x16x32 (72 MFMA)", t); }
    return 0;
}
