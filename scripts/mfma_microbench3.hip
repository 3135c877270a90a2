// Why does the rollout's MFMA chain run at ~42 cycles/MFMA instead of 32?  Replicates cem_mfma_stage: per group, two
// 16-B/lane loads of A operands (L2 resident stream) + 24 v_mfma_f32_16x16x4_f32 on 6 accumulators, 4-slot ring.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
struct AGroup { f4 a, b; };

template <int VARIANT>   // 0: no loads (A constant)  1: ring loads, pinned  2: ring loads, SALU-free addressing (pointer bump)  3: like 1 but 2 waves/SIMD
__global__ __launch_bounds__(512) void kern(const f4 *w, float *out, long long *cyc, int iters, int ngroups)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const f4 *base = w + (size_t)(blockIdx.x % 5) * 40000 + (size_t)(wv & 3) * 36 * 128 + lane;
    f4 acc0[3], acc1[3], hB[8][3];
    for (int c = 0; c < 3; ++c) { acc0[c] = (f4){0, 0, 0, 0}; acc1[c] = acc0[c]; for (int F = 0; F < 8; ++F) hB[F][c] = (f4){1.f + c, 2.f + F, 3.f, 4.f}; }
    AGroup slot[4];
    slot[0].a = base[0]; slot[0].b = base[64]; slot[1].a = base[128]; slot[1].b = base[192]; slot[2].a = base[256]; slot[2].b = base[320]; slot[3] = slot[2];
    int pos = 3;
    const f4 *ptr = base + 3 * 128;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int F = 0; F < 8; ++F) {
            if (VARIANT == 1 || VARIANT == 3) {
                slot[(F + 3) & 3].a = base[pos * 128]; slot[(F + 3) & 3].b = base[pos * 128 + 64];
                pos = (pos + 1 == ngroups) ? 0 : pos + 1;
                __builtin_amdgcn_sched_barrier(0);
            } else if (VARIANT == 2) {
                slot[(F + 3) & 3].a = ptr[0]; slot[(F + 3) & 3].b = ptr[64];
                ptr += 128;
                __builtin_amdgcn_sched_barrier(0);
            }
            const AGroup g = slot[F & 3];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    acc0[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.a[r], hB[F][c][r], acc0[c], 0, 0, 0);
                    acc1[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(g.b[r], hB[F][c][r], acc1[c], 0, 0, 0);
                }
        }
        if (VARIANT == 2 && (it & 3) == 3) ptr = base;     // wrap every 32 groups
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < 3; ++c) s += acc0[c][0] + acc1[c][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wv] = t1 - t0;
}

template <int V> void run(const char *name, int threads)
{
    f4 *w; float *out; long long *cyc; const int blocks = 210, iters = 200;
    hipMalloc(&w, 5 * 40000 * 16 + (1 << 20)); hipMemset(w, 0, 5 * 40000 * 16 + (1 << 20));
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 64);
    hipLaunchKernelGGL((kern<V>), dim3(blocks), dim3(threads), 0, 0, w, out, cyc, 10, 36);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<V>), dim3(blocks), dim3(threads), 0, 0, w, out, cyc, iters, 36);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c[8]; hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
    printf("%-52s ticks/MFMA = %6.2f   wall: %.3f ms\n", name, (double)c[0] / (iters * 192.0), ms);
}
int main()
{
    run<0>("no loads (A in registers)", 256);
    run<1>("ring loads + modulo position (as in the kernel)", 256);
    run<2>("ring loads + pointer bump", 256);
    run<3>("as kernel, 2 waves/SIMD (512 threads)", 512);
    return 0;
}
