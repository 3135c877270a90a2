// Microbenchmark (round 3): what ONE instruction of each class costs beside v_mfma_f32_16x16x4_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_mb5 mfma_microbench5.hip && ./mfma_mb5
// Part A: one wave per SIMD, K fillers of one class issued after every MFMA (2 independent accumulators): cycles per MFMA gap.
// Part B: two waves per SIMD, one issuing only MFMAs and one issuing only fillers: does the filler wave's work overlap with the
//         other wave's fp32 MFMAs (time = max) or serialise with it (time = sum)?
// The rollout epilogue's instruction classes: plain fp32 VALU, packed fp32 (what -O3 SLP-packs adjacent scalar ops into),
// 64-bit integer multiply-add (Philox), transcendentals (Box-Muller, softplus), integer logic, v_mov, cross-lane swaps, LDS reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
// -DMB_BF16: the same measurements beside v_mfma_f32_16x16x32_bf16 (the split-product rollout's MFMA; 16 cycles bare)
#ifdef MB_BF16
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define MB_AB_DECL const bf8 av = __builtin_bit_cast(bf8, (u4){(unsigned)threadIdx.x | 0x3c003c00u, 0x3c013c02u, 0x3c033c04u, 0x3c053c06u}), bv = __builtin_bit_cast(bf8, (u4){0x3c073c08u, (unsigned)threadIdx.x | 0x3c003c00u, 0x3c093c0au, 0x3c0b3c0cu});
#define MB_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, ACC, 0, 0, 0)
#else
#define MB_AB_DECL
#define MB_MFMA(ACC) ACC = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, ACC, 0, 0, 0)
#endif

enum Op { NONE, FMA, MUL, XOR, MOV, PKMUL, PKFMA, MAD64, EXP, SQRT, SIN, MAX, CNDMASK, PERMSWAP, DSREAD, SNOP, CVT, MULLO };
static const char *op_name[] = {"none", "v_fma_f32", "v_mul_f32", "v_xor_b32", "v_mov_b32", "v_pk_mul_f32", "v_pk_fma_f32", "v_mad_u64_u32",
                                "v_exp_f32", "v_sqrt_f32", "v_sin_f32", "v_max_f32", "v_cndmask_b32", "v_permlane16_swap", "ds_read_b128", "s_nop 0",
                                "v_cvt_f32_u32", "v_mul_lo_u32"};

template <int OP>
__device__ __forceinline__ void filler(float &x, float &y, f2 &p, f2 &q, unsigned long long &u, unsigned &m, f4 &l, const char *lds, float c)
{
    if (OP == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(c));
    if (OP == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(m) : "v"(c));
    if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(y));
    if (OP == PKMUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    if (OP == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p) : "v"(q));
    if (OP == MAD64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(p) : "v"(m), "v"(c) : "vcc");
    if (OP == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (OP == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));
    if (OP == SIN) asm volatile("v_sin_f32 %0, %0" : "+v"(x));
    if (OP == MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c));
    if (OP == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c) : );
    if (OP == PERMSWAP) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(x), "+v"(y));
    if (OP == DSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(size_t)lds));
    if (OP == SNOP) asm volatile("s_nop 0");
    if (OP == CVT) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(x) : "v"(m));
    if (OP == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(m) : "v"(c));
}

// ROLE 0: MFMAs with K fillers each.  ROLE 1 (part B): even waves (w < 4) MFMA only, odd waves (w >= 4) fillers only
template <int OP, int K, int ROLE>
__global__ __launch_bounds__(512) void kern(float *out, long long *cyc, int iters, float seed)
{
    __shared__ __attribute__((aligned(16))) char lds[4096];
    const int w = threadIdx.x >> 6;
    f4 acc0 = (f4){seed, seed, seed, seed}, acc1 = acc0;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x; (void)a; (void)b;
    MB_AB_DECL
    float x[4] = {seed, seed + 1, seed + 2, seed + 3}, y = 0.f;
    f2 p[4] = {{seed, 1.f}, {seed, 2.f}, {seed, 3.f}, {seed, 4.f}}, q = {1.0001f, 0.999f};
    unsigned long long u = 0; unsigned m[4] = {threadIdx.x, 2, 3, 4}; f4 l = {0, 0, 0, 0};
    const bool do_mfma = ROLE == 0 || w < 4, do_fill = ROLE == 0 || w >= 4;
    if (ROLE == 2 && w >= 4) __builtin_amdgcn_s_setprio(3);
    if (ROLE == 3 && w < 4) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int uu = 0; uu < 32; ++uu) {
            if (do_mfma) {
                if (uu & 1) MB_MFMA(acc1);
                else MB_MFMA(acc0);
            }
            if (do_fill) {
#pragma unroll
                for (int k = 0; k < K; ++k) filler<OP>(x[k & 3], y, p[k & 3], q, u, m[k & 3], l, lds + (threadIdx.x & 63) * 16, 1.0001f);
            }
        }
        if (OP == DSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = acc0[0] + acc1[1] + x[0] + x[1] + x[2] + x[3] + y + p[0][0] + p[1][1] + p[2][0] + p[3][1] + (float)u + (float)(m[0] ^ m[1] ^ m[2] ^ m[3]) + l[0];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

// Part C: W waves per SIMD (W workgroup-like groups of 4 waves in one 256*W-thread block), each wave alternating a phase of M
// back-to-back MFMAs with a phase of V plain VALU instructions, the groups started out of phase (group g skips g*M/W MFMAs of its
// first phase).  PRIO 0: all equal; 1: s_setprio 2 during the VALU phase; 2: s_setprio 2 during the MFMA phase.
template <int W, int M, int V, int PRIO>
__global__ __launch_bounds__(256 * W) void kern_c(float *out, long long *cyc, int iters, float seed)
{
    const int w = threadIdx.x >> 6, g = w >> 2;
    f4 acc0 = (f4){seed, seed, seed, seed}, acc1 = acc0;
    float a = seed + threadIdx.x, b = seed * 2 + threadIdx.x; (void)a; (void)b;
    MB_AB_DECL
    float x[4] = {seed, seed + 1, seed + 2, seed + 3};
    // stagger: group g first runs a partial MFMA phase
    for (int i = 0; i < g * (M / W); i += 2) { MB_MFMA(acc0); MB_MFMA(acc1); }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (PRIO == 2) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int i = 0; i < M; i += 2) { MB_MFMA(acc0); MB_MFMA(acc1); }
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (PRIO == 1) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int k = 0; k < V; ++k) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[k & 3]) : "v"(1.0001f));
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 * W + threadIdx.x] = acc0[0] + acc1[1] + x[0] + x[1] + x[2] + x[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + w] = t1 - t0;
}

static float *g_out; static long long *g_cyc;

template <int W, int M, int V, int PRIO> double run_c()
{
    const int iters = 200;
    hipLaunchKernelGGL((kern_c<W, M, V, PRIO>), dim3(256), dim3(256 * W), 0, 0, g_out, g_cyc, 4, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kern_c<W, M, V, PRIO>), dim3(256), dim3(256 * W), 0, 0, g_out, g_cyc, iters, 1.0f);
    hipDeviceSynchronize();
    long long c[16]; hipMemcpy(c, g_cyc, 128, hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < 4 * W; ++i) mx = c[i] > mx ? c[i] : mx;
    return (double)mx / iters / W;            // SIMD cycles per (M MFMAs + V VALU) unit of work
}
template <int M, int V> void partC()
{
    printf("C phases of %3d MFMA + %3d VALU: ideal %6d | 1 wave/SIMD %7.0f | 2 waves: %7.0f (VALU prio %7.0f, MFMA prio %7.0f) | 3 waves: %7.0f (VALU prio %7.0f, MFMA prio %7.0f)\n",
           M, V, M * 32 + V * 4, run_c<1, M, V, 0>(), run_c<2, M, V, 0>(), run_c<2, M, V, 1>(), run_c<2, M, V, 2>(), run_c<3, M, V, 0>(), run_c<3, M, V, 1>(), run_c<3, M, V, 2>());
}

static double g_wave_a, g_wave_b;   // part B: cycles per slot of the MFMA wave / of the filler wave
template <int OP, int K, int ROLE> double run(int threads, int blocks = 256)
{
    const int iters = 400;
    hipLaunchKernelGGL((kern<OP, K, ROLE>), dim3(blocks), dim3(threads), 0, 0, g_out, g_cyc, 4, 1.0f);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((kern<OP, K, ROLE>), dim3(blocks), dim3(threads), 0, 0, g_out, g_cyc, iters, 1.0f);
    hipDeviceSynchronize();
    long long c[8]; hipMemcpy(c, g_cyc, 64, hipMemcpyDeviceToHost);
    long long mx = 0; for (int i = 0; i < threads / 64; ++i) mx = c[i] > mx ? c[i] : mx;
    g_wave_a = (double)c[0] / (iters * 32.0); g_wave_b = (double)c[threads / 64 - 1] / (iters * 32.0);
    return (double)mx / (iters * 32.0);
}

template <int OP> void partA()
{
    const double k0 = run<NONE, 0, 0>(256), k1 = run<OP, 1, 0>(256), k2 = run<OP, 2, 0>(256), k4 = run<OP, 4, 0>(256), k8 = run<OP, 8, 0>(256);
    printf("A %-18s cycles per MFMA gap with 0/1/2/4/8 fillers: %6.1f %6.1f %6.1f %6.1f %6.1f   => per filler (8): %5.2f\n", op_name[OP], k0, k1, k2, k4, k8, (k8 - k0) / 8);
}
template <int OP, int K> void partB()
{
    // K fillers per slot on the filler wave, one MFMA per slot on the MFMA wave; equal / filler wave raised / MFMA wave raised priority
    run<OP, K, 1>(512); const double a1 = g_wave_a, b1 = g_wave_b;
    run<OP, K, 2>(512); const double a2 = g_wave_a, b2 = g_wave_b;
    run<OP, K, 3>(512); const double a3 = g_wave_a, b3 = g_wave_b;
    const double one = run<OP, K, 0>(256);
    printf("B %-14s x%d  MFMA wave / filler wave cycles per slot: equal prio %6.1f / %6.1f   filler prio 3: %6.1f / %6.1f   MFMA prio 3: %6.1f / %6.1f   (one wave doing both: %6.1f)\n",
           op_name[OP], K, a1, b1, a2, b2, a3, b3, one);
}

int main()
{
    hipMalloc(&g_out, 256 * 512 * 4); hipMalloc(&g_cyc, 256 * 128);
    partA<FMA>(); partA<MUL>(); partA<MAX>(); partA<XOR>(); partA<MOV>(); partA<CNDMASK>(); partA<CVT>();
    partA<PKMUL>(); partA<PKFMA>(); partA<MAD64>(); partA<MULLO>(); partA<EXP>(); partA<SQRT>(); partA<SIN>();
    partA<PERMSWAP>(); partA<DSREAD>(); partA<SNOP>();
    partB<NONE, 0>();
    partB<FMA, 1>(); partB<FMA, 2>(); partB<FMA, 4>(); partB<FMA, 8>(); partB<FMA, 16>();
    partB<XOR, 4>(); partB<PKMUL, 4>(); partB<EXP, 4>(); partB<DSREAD, 2>(); partB<SNOP, 4>();
    partC<64, 16>(); partC<64, 64>(); partC<64, 200>(); partC<288, 330>();
    return 0;
}
