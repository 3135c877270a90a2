// cem_train.h — ensemble training step on the device: MlpEnsemble.training_step / validation_step,
// simba/models/mlp_ensemble.py:134-155, with negative_log_likelihood (:64-67) and
// tf.keras.optimizers.Adam(lr, clipvalue=1.0, epsilon=1e-5) (:113-117).
//
// One workgroup per ensemble member (the members are independent: their own minibatch, weights and Adam moments).
// A training step is ~28 MFLOP per member in 17 small dependent GEMMs (batch <= 64): latency-bound by construction, so this
// is a plain LDS-tiled fp32 FMA GEMM (the fp32 MFMA has the same peak rate as the packed vector FMA on gfx950), everything
// L2 resident.  The workgroup is 512 threads (two waves per SIMD: the partner hides LDS / L2 latency) and its output tile
// 64 x 128, so every layer is ONE pass over its K dimension.  Weights stay in the Keras layout ([in][out]) the planner's
// set_weights() consumes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CEM_TB 64            // max minibatch rows per member (config/models.yaml:4 batch_size: 64)

struct TrainParams {
    float *W, *Mo, *Vo;          // [E][nat] weights, Adam first / second moments (natural blob layout of cem_mpc.h)
    float *grad;                 // [E][nat]
    float *scratch;              // [E][scratch_per_member]
    const float *x, *y;          // [n][D] scaled inputs, [n][O] targets (next_obs - obs)
    const int32_t *perm;         // [E][nperm] bootstrap shuffles (mlp_ensemble.py:172-173) or nullptr (rows offset.. directly)
    int32_t nperm, offset, Bt;
    int32_t D, O, U, L, E;
    uint32_t nat, scratch_per_member;
    float lr_t, beta1, beta2, eps, clip;
    float *loss_out;             // train: [E] loss share of each member; eval: [E][2] raw sums (log term, squared term)
    int32_t train;
    long long *stamps;           // [32] phase stamps (member 0), written by -DCEM_STAMPS diagnostic builds only
};

#ifdef CEM_STAMPS
#define CEM_TR_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) p.stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CEM_TR_STAMP(i) do { } while (0)
#endif

// C(m,n) = sum_k A(m,k) B(k,n) with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; epi(m, n, value).
// 64 x 128 output tile per pass (512 threads x 4x4 micro-tiles), k in slabs of CEM_TK staged through a double-buffered LDS
// tile; the next slab's global loads are issued into registers before the current slab's FMAs, so the L2 latency hides
// behind the arithmetic and a slab costs one barrier.
#ifndef CEM_TNT
#define CEM_TNT 512                                     // threads per workgroup (8 waves: 2 per SIMD; 1024 measured slower: 262 vs 250 us, spills under the 128-VGPR cap)
#endif
#define CEM_NCB (CEM_TN / 16 / (CEM_TNT / 256))         // 16-column blocks per wave: waves = 4 row blocks x (CEM_TNT/256) column groups
#define CEM_TK 32
#define CEM_TM 64
#define CEM_TN 128
#define CEM_TPAD 16                                     // row stride = 16 (mod 64) words: the 4 k-rows of an MFMA operand read hit disjoint banks
#define CEM_TRAIN_LDS_FLOATS (2 * CEM_TK * (CEM_TM + CEM_TPAD) + 2 * CEM_TK * (CEM_TN + CEM_TPAD))
// the workgroup's GEMM staging tiles (also the scratch of the small reductions between GEMMs); file scope so that the
// non-inlined GEMM addresses it as LDS
__shared__ __attribute__((aligned(16))) float g_train_lds[CEM_TRAIN_LDS_FLOATS];

typedef const __attribute__((address_space(1))) float *gcptr;       // global memory, said explicitly: inside a non-inlined
typedef __attribute__((address_space(1))) float *gptr;              // function a generic pointer would become flat accesses

// what happens to C(m,n) = sum_k A(m,k) B(k,n):  v = C + bias[n];  v = max(v, 0) (relu);  v = gate[m][n] > 0 ? v : 0;
// out[m][n] = v.   One body for all 17 GEMMs of a step: inlined per call site the kernel was
// ~70 KB of straight-line code, more than the instruction cache, and every step streamed its instructions from L2.
struct GemmEpi {
    gptr out; int ldo;
    gcptr bias;            // [N] or null
    gcptr gate; int ldg;   // [M][ldg] or null
    int relu;
    // optional column split (the mu | variance head pair as ONE GEMM): columns n >= nsplit go to out1 / bias1 at n - nsplit
    gptr out1; gcptr bias1;
    long long *st;         // -DCEM_STAMPS builds: accumulates [8] prologue (first slab in LDS), [9] k loop, [10] epilogue cycles of member 0
};

// operand split of the fused head GEMMs: B(k, n) comes from B1 at (k - ksplit, n) for k >= ksplit or at (k, n - nsplit) for
// n >= nsplit; A(m, k) from A1 at (m, k - ksplit).  Unused splits are INT_MAX.
struct GemmSplit { gcptr A1, B1; int ksplit, nsplit; };
#define CEM_NOSPLIT GemmSplit{nullptr, nullptr, 0x7fffffff, 0x7fffffff}

__device__ __attribute__((noinline)) void wg_gemm(const int M, const int N, const int K, const gcptr Ag, const int sam, const int sak,
                                                  const gcptr Bg, const int sbk, const int sbn, const GemmEpi e, const GemmSplit sp)
{
    float *lds = g_train_lds;
    typedef float TileA[CEM_TK][CEM_TM + CEM_TPAD];
    typedef float TileB[CEM_TK][CEM_TN + CEM_TPAD];
    TileA *As = reinterpret_cast<TileA *>(lds);                                    // As[buf][k][m]
    TileB *Bs = reinterpret_cast<TileB *>(lds + 2 * CEM_TK * (CEM_TM + CEM_TPAD));   // Bs[buf][k][n]
    constexpr int NEA = CEM_TM * CEM_TK / CEM_TNT, NEB = CEM_TN * CEM_TK / CEM_TNT;   // elements per thread per operand slab
    // MFMA 16x16x4 core: wave w owns rows [16 rb, +16) x columns [64 ch, +64) of the tile as four 16x16 blocks; lane (kq, i)
    // feeds A[16 rb + i][4P + kq] and B[4P + kq][64 ch + 16 cb + i]; the block's D has rows 4 kq + r, column i on the lane.
    // Each loaded operand word serves 16 FMAs (4x4 micro-tiles: 2), which takes the kernel off the LDS-bandwidth bound; the
    // hardware accumulates k in ascending order, the same chain the FMA version ran.
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, rb = wv & 3, ch = wv >> 2, kq = lane >> 4, li = lane & 15;
    constexpr int CW = 16 * CEM_NCB;                                               // columns per wave
    const int nk = (K + CEM_TK - 1) / CEM_TK;
    for (int m0 = 0; m0 < M; m0 += CEM_TM) {
        for (int n0 = 0; n0 < N; n0 += CEM_TN) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v acc[CEM_NCB];                             // acc[cb][r] = C[16 rb + 4 kq + r][CW ch + 16 cb + li]
#pragma unroll
            for (int cb = 0; cb < CEM_NCB; ++cb) acc[cb] = (f4v){0.f, 0.f, 0.f, 0.f};
            float ra[NEA], rbuf[NEB];
            auto a_idx = [&](const int q, int &mm, int &kk) {
                const int el = tid + CEM_TNT * q;
                if (sak == 1) { kk = el % CEM_TK; mm = el / CEM_TK; } else { mm = el % CEM_TM; kk = el / CEM_TM; }   // coalesce along the unit stride
            };
            auto b_idx = [&](const int q, int &nn, int &kb) {
                const int el = tid + CEM_TNT * q;
                if (sbn == 1) { nn = el % CEM_TN; kb = el / CEM_TN; } else { kb = el % CEM_TK; nn = el / CEM_TK; }
            };
            // loads are unconditional on clamped indices and zeroed afterwards: a guarded load `ok ? A[i] : 0` compiles to a
            // branch around the load with a wait behind it, i.e. the slab's loads go out one L2 round trip at a time
            auto fetch = [&](const int k0) {
#pragma unroll
                for (int q = 0; q < NEA; ++q) {
                    int mm, kk; a_idx(q, mm, kk);
                    const int gm = m0 + mm, gk = k0 + kk;
                    const int cm = gm < M ? gm : M - 1, ck = gk < K ? gk : K - 1;
                    const bool a0 = ck < sp.ksplit;                      // pointer and index are selected, then ONE unconditional load
                    const gcptr ap = a0 ? Ag : sp.A1;
                    ra[q] = ap[cm * sam + (a0 ? ck : ck - sp.ksplit) * sak];           // 32-bit offsets: every operand is < 2^31 floats
                }
#pragma unroll
                for (int q = 0; q < NEB; ++q) {
                    int nn, kb; b_idx(q, nn, kb);
                    const int gn = n0 + nn, gkb = k0 + kb;
                    const int cn = gn < N ? gn : N - 1, ck = gkb < K ? gkb : K - 1;
                    const gcptr bp = (ck < sp.ksplit && cn < sp.nsplit) ? Bg : sp.B1;
                    rbuf[q] = bp[(ck < sp.ksplit ? ck : ck - sp.ksplit) * sbk + (cn < sp.nsplit ? cn : cn - sp.nsplit) * sbn];
                }
#pragma unroll
                for (int q = 0; q < NEA; ++q) { int mm, kk; a_idx(q, mm, kk); if (m0 + mm >= M || k0 + kk >= K) ra[q] = 0.f; }
#pragma unroll
                for (int q = 0; q < NEB; ++q) { int nn, kb; b_idx(q, nn, kb); if (n0 + nn >= N || k0 + kb >= K) rbuf[q] = 0.f; }
            };
            auto stash = [&](const int buf) {
#pragma unroll
                for (int q = 0; q < NEA; ++q) { int mm, kk; a_idx(q, mm, kk); As[buf][kk][mm] = ra[q]; }
#pragma unroll
                for (int q = 0; q < NEB; ++q) { int nn, kb; b_idx(q, nn, kb); Bs[buf][kb][nn] = rbuf[q]; }
            };
#ifdef CEM_STAMPS
            long long t0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
            fetch(0);
            stash(0);
            __syncthreads();
#ifdef CEM_STAMPS
            long long t1_ = (long long)__builtin_amdgcn_s_memtime();
#endif
            // epilogue operands, requested now (batched, clamped indices) so that their latency hides behind the k loop
            float bia[CEM_NCB], gat[4][CEM_NCB];
#pragma unroll
            for (int jn = 0; jn < CEM_NCB; ++jn) {
                const int n = n0 + CW * ch + 16 * jn + li, cn = n < N ? n : N - 1;
                const gcptr bp = cn < sp.nsplit ? e.bias : e.bias1;
                bia[jn] = e.bias ? bp[cn < sp.nsplit ? cn : cn - sp.nsplit] : 0.f;
            }
            if (e.gate) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < CEM_NCB; ++jn) {
                        const int mI = m0 + 16 * rb + 4 * kq + i, n = n0 + CW * ch + 16 * jn + li;
                        gat[i][jn] = e.gate[(mI < M ? mI : M - 1) * e.ldg + (n < N ? n : N - 1)];
                    }
            }
            for (int kt = 0; kt < nk; ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nk) fetch((kt + 1) * CEM_TK);
#pragma unroll
                for (int P = 0; P < CEM_TK / 4; ++P) {
                    const float a = As[buf][4 * P + kq][16 * rb + li];
                    float b[CEM_NCB];
#pragma unroll
                    for (int cb = 0; cb < CEM_NCB; ++cb) b[cb] = Bs[buf][4 * P + kq][CW * ch + 16 * cb + li];
#pragma unroll
                    for (int cb = 0; cb < CEM_NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cb], acc[cb], 0, 0, 0);
                }
                if (kt + 1 < nk) stash(buf ^ 1);
                __syncthreads();
            }
#ifdef CEM_STAMPS
            long long t2_ = (long long)__builtin_amdgcn_s_memtime();
#endif
            // epilogue (its bias / gate operands were requested before the k loop)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < CEM_NCB; ++jn) {
                    const int mI = m0 + 16 * rb + 4 * kq + i, n = n0 + CW * ch + 16 * jn + li;
                    float v = acc[jn][i];
                    if (e.bias) v = v + bia[jn];
                    if (e.relu) v = fmaxf(v, 0.f);
                    if (e.gate) v = gat[i][jn] > 0.f ? v : 0.f;
                    if (mI < M && n < N) (n < sp.nsplit ? e.out : e.out1)[(mI < M ? mI : M - 1) * e.ldo + (n < sp.nsplit ? n : n - sp.nsplit)] = v;
                }
#ifdef CEM_STAMPS
            if (e.st && blockIdx.x == 0 && tid == 0) { const long long t3_ = (long long)__builtin_amdgcn_s_memtime(); e.st[8] += t1_ - t0_; e.st[9] += t2_ - t1_; e.st[10] += t3_ - t2_; e.st[11] += 1; }
#endif
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float train_softplus(float x)         // Eigen's three branches, precise
{
    const float thr = -13.942383766174316f;
    if (x > -thr) return x;
    const float ex = expf(x);
    if (x < thr) return ex;
    return log1pf(ex);
}

// Elementwise pass over n items with the loads of CEM_UNR items in flight at once: with one or two waves per SIMD a plain
// `for (e = tid; ...)` loop pays most of the L2 latency every iteration.
#define CEM_UNR 4
template <class T, class Ld, class St>
__device__ __forceinline__ void wg_map(const int n, Ld ld, St st)
{
    for (int base = 0; base < n; base += CEM_TNT * CEM_UNR) {
        T v[CEM_UNR];
#pragma unroll
        for (int q = 0; q < CEM_UNR; ++q) { const int e = base + q * CEM_TNT + (int)threadIdx.x; if (e < n) v[q] = ld(e); }
#pragma unroll
        for (int q = 0; q < CEM_UNR; ++q) { const int e = base + q * CEM_TNT + (int)threadIdx.x; if (e < n) st(e, v[q]); }
    }
}

__device__ __forceinline__ float block_sum(float v, float *red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < CEM_TNT / 64; ++i) t = t + red[i];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(CEM_TNT) void cem_train_step_kernel(const TrainParams p)
{
    float *lds = g_train_lds;
    __shared__ float red[CEM_TNT / 64];
    const int m = blockIdx.x, tid = threadIdx.x;
    const int D = p.D, O = p.O, U = p.U, L = p.L, Bt = p.Bt;
    float *W = p.W + (size_t)m * p.nat, *G = p.grad + (size_t)m * p.nat;
    float *sc = p.scratch + (size_t)m * p.scratch_per_member;
    // scratch carve (row stride U for every activation matrix; D, O <= U)
    float *xs = sc;                              // [TB][U]   h_0
    float *hs = xs + CEM_TB * U;                 // [L][TB][U] h_1..h_L
    float *mu = hs + (size_t)L * CEM_TB * U;     // [TB][U]
    float *vp = mu + CEM_TB * U;
    float *ys = vp + CEM_TB * U;
    float *dmu = ys + CEM_TB * U;
    float *dv = dmu + CEM_TB * U;
    float *dha = dv + CEM_TB * U;
    float *dhb = dha + CEM_TB * U;
    // natural-blob offsets (cem_mpc.h): W_0,b_0,...,W_mu,b_mu,W_var,b_var
    auto offW = [&](int l) { return l == 0 ? (size_t)0 : (size_t)D * U + U + (size_t)(l - 1) * ((size_t)U * U + U); };
    auto offb = [&](int l) { return offW(l) + (size_t)(l == 0 ? D : U) * U; };
    const size_t oWmu = (size_t)D * U + U + (size_t)(L - 1) * ((size_t)U * U + U), obmu = oWmu + (size_t)U * O;
    const size_t oWv = obmu + O, obv = oWv + (size_t)U * O;

    CEM_TR_STAMP(0);
#ifdef CEM_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { p.stamps[8] = p.stamps[9] = p.stamps[10] = p.stamps[11] = 0; }
#endif
    // ---- gather the minibatch ---------------------------------------------------------------------------------
    {
        int32_t *rows = reinterpret_cast<int32_t *>(lds);
        if (tid < Bt) rows[tid] = p.perm ? p.perm[(size_t)m * p.nperm + p.offset + tid] : p.offset + tid;
        __syncthreads();
        wg_map<float2>(Bt * U,
            [&](int e) { const int r = e / U, c = e % U; const int row = rows[r];
                         return make_float2(c < D ? p.x[(size_t)row * D + c] : 0.f, c < O ? p.y[(size_t)row * O + c] : 0.f); },
            [&](int e, float2 v) { xs[e] = v.x; ys[e] = v.y; });
    }
    __syncthreads();

    CEM_TR_STAMP(1);
    // ---- forward (mlp_ensemble.py:18-22,33-34,59-61) -----------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TB * U;
        float *hout = hs + (size_t)l * CEM_TB * U;
        const float *Wl = W + offW(l), *bl = W + offb(l);
        wg_gemm(Bt, U, l == 0 ? D : U, (gcptr)hin, U, 1, (gcptr)Wl, U, 1, GemmEpi{(gptr)hout, U, (gcptr)bl, nullptr, 0, 1, nullptr, nullptr, p.stamps}, CEM_NOSPLIT);     // relu(h W + b)
    }
    CEM_TR_STAMP(2);
    const float *hL = hs + (size_t)(L - 1) * CEM_TB * U;
    // both heads as ONE GEMM: columns [0, O) = mu head, [O, 2O) = variance head (2O <= 128 fills the tile two N = O GEMMs half use)
    wg_gemm(Bt, 2 * O, U, (gcptr)hL, U, 1, (gcptr)(W + oWmu), O, 1,
            GemmEpi{(gptr)mu, U, (gcptr)(W + obmu), nullptr, 0, 0, (gptr)vp, (gcptr)(W + obv), p.stamps}, GemmSplit{nullptr, (gcptr)(W + oWv), 0x7fffffff, O});
    CEM_TR_STAMP(3);
    // ---- negative_log_likelihood (:64-67) and its gradient w.r.t. mu and the pre-softplus variance -----------------
    float s_log = 0.f, s_sq = 0.f;
    const float ninv = 1.0f / ((float)Bt * (float)O * (float)p.E);
    wg_map<float3>(Bt * O,
        [&](int e) { const int r = e / O, c = e % O; return make_float3(vp[r * U + c], mu[r * U + c], ys[r * U + c]); },
        [&](int e, float3 in) {
            const int r = e / O, c = e % O;
            const float v = in.x, var = train_softplus(v) + 1e-4f;
            const float diff = in.y - in.z;
            s_log += logf(6.283185307179586f * var);
            s_sq += diff * diff / var;
            if (p.train) {
                dmu[r * U + c] = diff / var * ninv;
                const float dvar = (0.5f / var - 0.5f * diff * diff / (var * var)) * ninv;
                dv[r * U + c] = dvar / (1.0f + expf(-v));               // d softplus(v)/dv = sigmoid(v)
            }
        });
    s_log = block_sum(s_log, red);
    s_sq = block_sum(s_sq, red);
    CEM_TR_STAMP(4);
    if (!p.train) {
        if (tid == 0) { p.loss_out[2 * m] = s_log; p.loss_out[2 * m + 1] = s_sq; }
        return;
    }
    if (tid == 0) p.loss_out[m] = (0.5f * s_log / ((float)Bt * (float)O) + 0.5f * s_sq / ((float)Bt * (float)O)) / (float)p.E;
    __syncthreads();

    CEM_TR_STAMP(5);
    // ---- backward ------------------------------------------------------------------------------------------------
    // column sums (bias gradients) of a [Bt][U]-strided matrix: CEM_TNT/128 partial sums over interleaved rows, added in order
    float *colred = lds;
    auto col_sums = [&](const float *src, const int ncol, float *dst) {
        constexpr int NP = CEM_TNT / 128;
        for (int c0 = 0; c0 < ncol; c0 += 128) {
            const int c = c0 + (tid & 127), part = tid >> 7;
            float a = 0.f;
            if (c < ncol) {
#pragma unroll 8
                for (int r = part; r < Bt; r += NP) a += src[r * U + c];
            }
            colred[tid] = a;
            __syncthreads();
            if (tid < 128 && c < ncol) {
                float t = colred[tid];
#pragma unroll
                for (int q = 1; q < NP; ++q) t = t + colred[tid + 128 * q];
                dst[c] = t;
            }
            __syncthreads();
        }
    };
    // [dW_mu | dW_var] = h_L^T [dmu | dv] as one GEMM
    wg_gemm(U, 2 * O, Bt, (gcptr)hL, 1, U, (gcptr)dmu, U, 1,
            GemmEpi{(gptr)(G + oWmu), O, nullptr, nullptr, 0, 0, (gptr)(G + oWv), nullptr, p.stamps}, GemmSplit{nullptr, (gcptr)dv, 0x7fffffff, O});
    col_sums(dmu, O, G + obmu);
    col_sums(dv, O, G + obv);
    // dh_L = (dmu Wmu^T + dv Wvar^T) * relu'(h_L): the relu mask rides in the epilogue of the GEMM that completes dh
    // dh_L = ([dmu | dv] [W_mu | W_var]^T) * relu'(h_L): one GEMM over K = 2O; the relu mask rides in its epilogue
    wg_gemm(Bt, U, 2 * O, (gcptr)dmu, U, 1, (gcptr)(W + oWmu), 1, O,
            GemmEpi{(gptr)dha, U, nullptr, (gcptr)hL, U, 0, nullptr, nullptr, p.stamps}, GemmSplit{(gcptr)dv, (gcptr)(W + oWv), O, 0x7fffffff});
    CEM_TR_STAMP(6);
    float *dcur = dha, *dnext = dhb;
    for (int l = L - 1; l >= 0; --l) {
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TB * U;
        const int in = l == 0 ? D : U;
        wg_gemm(in, U, Bt, (gcptr)hin, 1, U, (gcptr)dcur, U, 1, GemmEpi{(gptr)(G + offW(l)), U, nullptr, nullptr, 0, 0, nullptr, nullptr, p.stamps}, CEM_NOSPLIT);        // dW_l = h_{l-1}^T dh_l
        col_sums(dcur, U, G + offb(l));
        if (l > 0) {
            wg_gemm(Bt, U, U, (gcptr)dcur, U, 1, (gcptr)(W + offW(l)), 1, U, GemmEpi{(gptr)dnext, U, nullptr, (gcptr)hin, U, 0, nullptr, nullptr, p.stamps}, CEM_NOSPLIT);   // dh_{l-1} = (dh_l W_l^T) relu'
            float *t = dcur; dcur = dnext; dnext = t;
        }
    }
    CEM_TR_STAMP(7);
}

// ---- Adam with clipvalue (mlp_ensemble.py:113-117,143-144), every member's parameters in one grid ---------------------
__global__ __launch_bounds__(256) void cem_adam_kernel(const TrainParams p)
{
    const size_t n = (size_t)p.E * p.nat, n4 = n / 4;
    const float ob1 = 1.0f - p.beta1, ob2 = 1.0f - p.beta2;
    auto upd = [&](float g, float &mo, float &vo, float &w) {
        g = fminf(fmaxf(g, -p.clip), p.clip);
        mo = mo + (g - mo) * ob1;
        vo = vo + (g * g - vo) * ob2;
        w = w - p.lr_t * mo / (sqrtf(vo) + p.eps);
    };
    float4 *W4 = reinterpret_cast<float4 *>(p.W), *M4 = reinterpret_cast<float4 *>(p.Mo), *V4 = reinterpret_cast<float4 *>(p.Vo);
    const float4 *G4 = reinterpret_cast<const float4 *>(p.grad);
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256) {
        const float4 g = G4[e];
        float4 mo = M4[e], vo = V4[e], w = W4[e];
        upd(g.x, mo.x, vo.x, w.x); upd(g.y, mo.y, vo.y, w.y); upd(g.z, mo.z, vo.z, w.z); upd(g.w, mo.w, vo.w, w.w);
        M4[e] = mo; V4[e] = vo; W4[e] = w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t e = n4 * 4 + threadIdx.x;
        upd(p.grad[e], p.Mo[e], p.Vo[e], p.W[e]);
    }
}
