// cem_train.h — ensemble training step on the device: MlpEnsemble.training_step / validation_step,
// simba/models/mlp_ensemble.py:134-155, with negative_log_likelihood (:64-67) and
// tf.keras.optimizers.Adam(lr, clipvalue=1.0, epsilon=1e-5) (:113-117).
//
// One workgroup per ensemble member (the members are independent: their own minibatch, weights and Adam moments).
// A training step is ~28 MFLOP per member in 17 small GEMMs (batch <= 64): latency-bound by construction, so this is a
// plain LDS-tiled fp32 FMA GEMM (the fp32 MFMA has the same peak rate as the vector FMA on gfx950), everything L2
// resident.  Weights stay in the Keras layout ([in][out]) the planner's set_weights() consumes.
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
};

// C(m,n) = sum_k A(m,k) B(k,n) with A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; epi(m, n, value)
template <class Epi>
__device__ __forceinline__ void wg_gemm(const int M, const int N, const int K, const float *A, const int sam, const int sak,
                                        const float *B, const int sbk, const int sbn, Epi epi, float *lds)
{
    float (*As)[68] = reinterpret_cast<float (*)[68]>(lds);
    float (*Bs)[68] = reinterpret_cast<float (*)[68]>(lds + 16 * 68);
    const int tid = threadIdx.x, tm = tid >> 4, tn = tid & 15;
    for (int m0 = 0; m0 < M; m0 += 64) {
        for (int n0 = 0; n0 < N; n0 += 64) {
            float acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) acc[i][jn] = 0.f;
            for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = tid + 256 * e4;
                    int mm, kk;
                    if (sak == 1) { kk = e & 15; mm = e >> 4; } else { mm = e & 63; kk = e >> 6; }      // coalesce along the unit stride
                    const int gm = m0 + mm, gk = k0 + kk;
                    As[kk][mm] = (gm < M && gk < K) ? A[(size_t)gm * sam + (size_t)gk * sak] : 0.f;
                    int nn, kb;
                    if (sbn == 1) { nn = e & 63; kb = e >> 6; } else { kb = e & 15; nn = e >> 4; }
                    const int gn = n0 + nn, gkb = k0 + kb;
                    Bs[kb][nn] = (gn < N && gkb < K) ? B[(size_t)gkb * sbk + (size_t)gn * sbn] : 0.f;
                }
                __syncthreads();
#pragma unroll
                for (int kk = 0; kk < 16; ++kk) {
                    const float4 a = *reinterpret_cast<const float4 *>(&As[kk][tm * 4]);
                    const float4 b = *reinterpret_cast<const float4 *>(&Bs[kk][tn * 4]);
                    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < 4; ++jn) acc[i][jn] = __builtin_fmaf(av[i], bv[jn], acc[i][jn]);
                }
                __syncthreads();
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int jn = 0; jn < 4; ++jn) {
                    const int m = m0 + tm * 4 + i, n = n0 + tn * 4 + jn;
                    if (m < M && n < N) epi(m, n, acc[i][jn]);
                }
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

__device__ __forceinline__ float block_sum(float v, float *red)
{
    const int tid = threadIdx.x;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void cem_train_step_kernel(const TrainParams p)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * 16 * 68];
    __shared__ float red[4];
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

    // ---- gather the minibatch ---------------------------------------------------------------------------------
    for (int e = tid; e < Bt * U; e += 256) {
        const int r = e / U, c = e % U;
        const int row = p.perm ? p.perm[(size_t)m * p.nperm + p.offset + r] : p.offset + r;
        xs[e] = c < D ? p.x[(size_t)row * D + c] : 0.f;
        ys[e] = c < O ? p.y[(size_t)row * O + c] : 0.f;
    }
    __syncthreads();

    // ---- forward (mlp_ensemble.py:18-22,33-34,59-61) -----------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TB * U;
        float *hout = hs + (size_t)l * CEM_TB * U;
        const float *Wl = W + offW(l), *bl = W + offb(l);
        wg_gemm(Bt, U, l == 0 ? D : U, hin, U, 1, Wl, U, 1, [&](int r, int n, float v) { hout[r * U + n] = fmaxf(v + bl[n], 0.f); }, lds);
    }
    const float *hL = hs + (size_t)(L - 1) * CEM_TB * U;
    wg_gemm(Bt, O, U, hL, U, 1, W + oWmu, O, 1, [&](int r, int n, float v) { mu[r * U + n] = v + W[obmu + n]; }, lds);
    wg_gemm(Bt, O, U, hL, U, 1, W + oWv, O, 1, [&](int r, int n, float v) { vp[r * U + n] = v + W[obv + n]; }, lds);

    // ---- negative_log_likelihood (:64-67) and its gradient w.r.t. mu and the pre-softplus variance -----------------
    float s_log = 0.f, s_sq = 0.f;
    const float ninv = 1.0f / ((float)Bt * (float)O * (float)p.E);
    for (int e = tid; e < Bt * O; e += 256) {
        const int r = e / O, c = e % O;
        const float v = vp[r * U + c], var = train_softplus(v) + 1e-4f;
        const float diff = mu[r * U + c] - ys[r * U + c];
        s_log += logf(6.283185307179586f * var);
        s_sq += diff * diff / var;
        if (p.train) {
            dmu[r * U + c] = diff / var * ninv;
            const float dvar = (0.5f / var - 0.5f * diff * diff / (var * var)) * ninv;
            dv[r * U + c] = dvar / (1.0f + expf(-v));               // d softplus(v)/dv = sigmoid(v)
        }
    }
    s_log = block_sum(s_log, red);
    s_sq = block_sum(s_sq, red);
    if (!p.train) {
        if (tid == 0) { p.loss_out[2 * m] = s_log; p.loss_out[2 * m + 1] = s_sq; }
        return;
    }
    if (tid == 0) p.loss_out[m] = (0.5f * s_log / ((float)Bt * (float)O) + 0.5f * s_sq / ((float)Bt * (float)O)) / (float)p.E;
    __syncthreads();

    // ---- backward ------------------------------------------------------------------------------------------------
    wg_gemm(U, O, Bt, hL, 1, U, dmu, U, 1, [&](int u, int n, float v) { G[oWmu + (size_t)u * O + n] = v; }, lds);
    wg_gemm(U, O, Bt, hL, 1, U, dv, U, 1, [&](int u, int n, float v) { G[oWv + (size_t)u * O + n] = v; }, lds);
    for (int c = tid; c < O; c += 256) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < Bt; ++r) { a += dmu[r * U + c]; b += dv[r * U + c]; }
        G[obmu + c] = a; G[obv + c] = b;
    }
    // dh_L = dmu Wmu^T + dv Wvar^T
    wg_gemm(Bt, U, O, dmu, U, 1, W + oWmu, 1, O, [&](int r, int n, float v) { dha[r * U + n] = v; }, lds);
    wg_gemm(Bt, U, O, dv, U, 1, W + oWv, 1, O, [&](int r, int n, float v) { dha[r * U + n] += v; }, lds);
    float *dcur = dha, *dnext = dhb;
    for (int l = L - 1; l >= 0; --l) {
        const float *hout = hs + (size_t)l * CEM_TB * U;
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TB * U;
        const int in = l == 0 ? D : U;
        for (int e = tid; e < Bt * U; e += 256) dcur[e] = hout[e] > 0.f ? dcur[e] : 0.f;     // relu'
        __syncthreads();
        wg_gemm(in, U, Bt, hin, 1, U, dcur, U, 1, [&](int i, int n, float v) { G[offW(l) + (size_t)i * U + n] = v; }, lds);
        for (int c = tid; c < U; c += 256) {
            float a = 0.f;
            for (int r = 0; r < Bt; ++r) a += dcur[r * U + c];
            G[offb(l) + c] = a;
        }
        if (l > 0) {
            wg_gemm(Bt, U, U, dcur, U, 1, W + offW(l), 1, U, [&](int r, int n, float v) { dnext[r * U + n] = v; }, lds);
            float *t = dcur; dcur = dnext; dnext = t;
        }
    }
    __syncthreads();

    // ---- Adam with clipvalue (mlp_ensemble.py:113-117,143-144) ---------------------------------------------------------
    float *Mo = p.Mo + (size_t)m * p.nat, *Vo = p.Vo + (size_t)m * p.nat;
    for (uint32_t e = tid; e < p.nat; e += 256) {
        const float g = fminf(fmaxf(G[e], -p.clip), p.clip);
        const float mo = Mo[e] + (g - Mo[e]) * (1.0f - p.beta1);
        const float vo = Vo[e] + (g * g - Vo[e]) * (1.0f - p.beta2);
        Mo[e] = mo; Vo[e] = vo;
        W[e] = W[e] - p.lr_t * mo / (sqrtf(vo) + p.eps);
    }
}
