// cem_train.h — ensemble training step on the device: shared definitions (TrainParams, the Adam kernel) and the
// GEMM-by-GEMM step kernel — since round 2 the FALLBACK (more than 6 layers, or CEM_TRAIN_GEMM_KERNEL=1); the default step is the
// rollout-style kernel of cem_train_tile.h.  MlpEnsemble.training_step / validation_step,
// simba/models/mlp_ensemble.py:134-155, with negative_log_likelihood (:64-67) and
// tf.keras.optimizers.Adam(lr, clipvalue=1.0, epsilon=1e-5) (:113-117).
//
// A training step is ~28 MFLOP per member in 14 small dependent GEMMs (batch <= 64): latency-bound by construction.  The
// members are independent (their own minibatch, weights and Adam moments), and so are the ROWS of a member's minibatch in
// everything except the weight gradients, which sum over rows.  So a member's step runs on CEM_TPARTS workgroups, each
// taking CEM_TROWS = 16 of the rows through the forward pass, the loss, and the backward pass; each writes its PARTIAL weight
// gradients, and the Adam kernel adds the partials in a fixed order before the update (deterministic: no atomics).  15
// members x 4 parts = 60 workgroups instead of 15, and each GEMM's row dimension is one MFMA block.
// The GEMM is LDS-tiled on v_mfma_f32_16x16x4_f32, everything L2 resident; the workgroup is 512 threads (two waves per SIMD:
// the partner hides LDS / L2 latency).  Weights stay in the Keras layout ([in][out]) the planner's set_weights() consumes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define CEM_TB 64            // max minibatch rows per member (config/models.yaml:4 batch_size: 64)
#define CEM_TROWS 16         // minibatch rows per workgroup
#define CEM_TS 128           // row stride of every activation matrix in the scratch for units <= 128 (TrainParams::ts: 256 above)
#define CEM_TWIDE 256         // widest hidden layer the GEMM-by-GEMM kernel is laid out for (inputs_dim, outputs_dim <= 128)
#define CEM_TPARTS (CEM_TB / CEM_TROWS)

struct TrainParams {
    float *W, *Mo, *Vo;          // [E][nat] weights, Adam first / second moments (natural blob layout of cem_mpc.h)
    float *grad;                 // [CEM_TPARTS][gpart] partial gradients of the row parts, each [E][nat] (gpart = E * nat rounded up to
                                 // a multiple of 4 floats: the Adam kernel reads every part with 16-byte loads)
    float *loss_part;            // [E][CEM_TPARTS][2] partial sums of the loss (log term, squared term)
    float *scratch;              // [E * CEM_TPARTS][scratch_per_member]
    const float *x, *y;          // [n][D] scaled inputs, [n][O] targets (next_obs - obs)
    const int32_t *perm;         // [E][nperm] bootstrap shuffles (mlp_ensemble.py:172-173) or nullptr (rows offset.. directly)
    int32_t nperm, offset, Bt;   // Bt rows from `offset` on; the tile kernel takes them in chunks of `chunk` rows along blockIdx.y (training: one
    int32_t chunk;               // chunk = the minibatch; validation: every 64-row slice of the set in ONE launch, loss_part per chunk)
    int32_t D, O, U, L, E;
    uint32_t nat, scratch_per_member, gpart;
    int32_t ts;                  // row stride of the GEMM kernel's activation matrices in the scratch: CEM_TS, or CEM_TWIDE for units > 128
    float lr_t, beta1, beta2, eps, clip;
    float *loss_out;             // train: [E] loss share of each member; eval: [E][2] raw sums (log term, squared term)
    int32_t train;
    int32_t act;                 // enum cem_activation of the hidden layers (the tile kernel is relu only: cem_capi.hip routes the others here)
    uint32_t drop_thresh, drop_step, drop_k0, drop_k1;   // training-time Dropout (GemmEpi): rate * 2^32 (0 = none), this step's index, the key
    float drop_scale, drop_keep;
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
#define CEM_TKMAX 128                                   // the deepest K a 16-row GEMM stages in ONE shot (units, 2 * outputs_dim <= 128)
#define CEM_TRAIN_LDS_FLOATS (CEM_TKMAX * (CEM_TM + CEM_TPAD) + CEM_TKMAX * (CEM_TN + CEM_TPAD))   // the one-shot 16-row form is the largest (112 KB)
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
    gcptr gate; int ldg;   // [M][ldg] or null: v = v * f'(z) with gate = f(z), the gated layer's output — or, for swish / gelu (CEM_ACT_NEEDS_Z),
                           // the layer's kept PRE-activation z (GemmEpi::outz of its forward GEMM; a unit Dropout dropped holds CEM_Z_DROPPED)
    int relu;              // 0: none; 1 + enum cem_activation: the hidden layers' nonlinearity (forward) / the one whose derivative gates (backward)
    // optional column split (the mu | variance head pair as ONE GEMM): columns n >= nsplit go to out1 / bias1 at n - nsplit
    gptr out1; gcptr bias1;
    long long *st;         // -DCEM_STAMPS builds: accumulates [8] prologue (first slab in LDS), [9] k loop, [10] epilogue cycles of member 0
    // optional: column sums of B over its K rows (K <= one slab), i.e. the bias gradient sum_r dh[r][n] next to dW = h^T dh
    gptr colsum, colsum1;  // columns n >= nsplit go to colsum1[n - nsplit]
    // optional training-time Dropout of the layer this GEMM produces (BaseLayer.call, mlp_ensemble.py:15,21; Keras semantics: keep
    // with probability 1 - rate, kept values scaled by 1 / (1 - rate)): element (row, n) is kept iff word (n & 3) of Philox4x32-7 at
    // counter (drop_row0 + row, step, (n >> 2) | layer << 8 | 3 << 16, member), key (seed_lo, seed_hi), is >= drop_thresh = rate * 2^32.
    // The backward gate needs no mask: a dropped unit's stored output is exactly 0, a kept one's is f(z) / (1 - rate).
    uint32_t drop_thresh;  // 0: no dropout
    float drop_scale, drop_keep;           // 1 / (1 - rate), 1 - rate
    uint32_t drop_step, drop_c2, drop_member, drop_k0, drop_k1;
    int drop_row0;
    // optional (forward GEMMs of swish / gelu layers): the pre-activation z of every output element, same indexing as `out` — what the
    // backward gate of a non-monotone activation needs; an element Dropout drops is stored as CEM_Z_DROPPED (a NaN bit pattern, compared as bits)
    gptr outz;
};

__device__ __forceinline__ float cem_dropout_fwd(const GemmEpi &e, const int row, const int n, const float v)
{
    uint32_t c0 = (uint32_t)(e.drop_row0 + row), c1 = e.drop_step, c2 = (uint32_t)(n >> 2) | e.drop_c2, c3 = e.drop_member;
    philox4x32_7(c0, c1, c2, c3, e.drop_k0, e.drop_k1);
    const uint32_t wsel = (n & 3) == 0 ? c0 : ((n & 3) == 1 ? c1 : ((n & 3) == 2 ? c2 : c3));
    return wsel >= e.drop_thresh ? v * e.drop_scale : 0.f;
}
#define CEM_Z_DROPPED 0xFFC00001u
// d * (d output / d pre-activation) of a hidden layer given its STORED output h (after activation and dropout) — or its kept pre-activation
__device__ __forceinline__ float cem_layer_gate(const GemmEpi &e, const float d, const float h)
{
    if (e.relu > 1 && CEM_ACT_NEEDS_Z(e.relu - 1)) {                  // swish / gelu: `h` is z (or the dropped marker)
        if (__float_as_uint(h) == CEM_Z_DROPPED) return 0.f;
        const float g = cem_activation_gate_z(e.relu - 1, d, h);
        return e.drop_thresh ? g * e.drop_scale : g;
    }
    if (e.drop_thresh == 0u) return e.relu <= 1 ? (h > 0.f ? d : 0.f) : cem_activation_gate(e.relu - 1, d, h);
    if (h == 0.f) return 0.f;                                  // dropped (or a kink / zero of f: a set of measure zero)
    return cem_activation_gate(e.relu - 1, d, h * e.drop_keep) * e.drop_scale;
}

// operand split of the fused head GEMMs: B(k, n) comes from B1 at (k - ksplit, n) for k >= ksplit or at (k, n - nsplit) for
// n >= nsplit; A(m, k) from A1 at (m, k - ksplit).  Unused splits are INT_MAX.
struct GemmSplit { gcptr A1, B1; int ksplit, nsplit; };
#define CEM_NOSPLIT GemmSplit{nullptr, nullptr, 0x7fffffff, 0x7fffffff}

// Tile forms (TMODE).  0: 64 x 128 output tile, waves = 4 row blocks x 2 groups of 64 columns.  1 (R16): the output has at most
// 16 rows (a row part's activations): the eight waves take eight 16-column groups of ONE row block.  2 (M128): 128 x 128 output
// tile, every wave two row blocks x four column blocks — the weight-gradient GEMMs (K = the part's 16 rows) in one pass.
template <int TMODE>
__device__ __attribute__((noinline)) void wg_gemm_t(const int M, const int N, const int K, const gcptr Ag, const int sam, const int sak,
                                                    const gcptr Bg, const int sbk, const int sbn, const GemmEpi e, const GemmSplit sp)
{
    constexpr bool R16 = TMODE == 1;
    constexpr int NCB = R16 ? 1 : CEM_NCB;
    constexpr int NRB = TMODE == 2 ? 2 : 1;                                        // row blocks per wave
    constexpr int TM = TMODE == 2 ? 2 * CEM_TM : CEM_TM;
    float *lds = g_train_lds;
    typedef float TileA[CEM_TK][TM + CEM_TPAD];
    typedef float TileB[CEM_TK][CEM_TN + CEM_TPAD];
    TileA *As = reinterpret_cast<TileA *>(lds);                                    // As[buf][k][m]
    TileB *Bs = reinterpret_cast<TileB *>(lds + 2 * CEM_TK * (TM + CEM_TPAD));       // Bs[buf][k][n]
    constexpr int NEA = TM * CEM_TK / CEM_TNT, NEB = CEM_TN * CEM_TK / CEM_TNT;       // elements per thread per operand slab
    // MFMA 16x16x4 core: wave w owns rows [16 rb, +16) (+64 for its second row block) x columns [CW ch, +CW) of the tile as 16x16
    // blocks; lane (kq, i) feeds A[16 rb + i][4P + kq] and B[4P + kq][CW ch + 16 cb + i]; the block's D has rows 4 kq + r, column i
    // on the lane.  Each loaded operand word serves 16 FMAs; the hardware accumulates k in ascending order.
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, rb = R16 ? 0 : (wv & 3), ch = R16 ? wv : (wv >> 2), kq = lane >> 4, li = lane & 15;
    constexpr int CW = 16 * NCB;                                                   // columns per wave
    const int nk = (K + CEM_TK - 1) / CEM_TK;
    for (int m0 = 0; m0 < M; m0 += TM) {
        for (int n0 = 0; n0 < N; n0 += CEM_TN) {
            typedef float f4v __attribute__((ext_vector_type(4)));
            f4v acc[NRB][NCB];                        // acc[rk][cb][r] = C[16 rb + 64 rk + 4 kq + r][CW ch + 16 cb + li]
#pragma unroll
            for (int rk = 0; rk < NRB; ++rk)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) acc[rk][cb] = (f4v){0.f, 0.f, 0.f, 0.f};
            float ra[NEA], rbuf[NEB];
            auto a_idx = [&](const int q, int &mm, int &kk) {
                const int el = tid + CEM_TNT * q;
                if (sak == 1) { kk = el % CEM_TK; mm = el / CEM_TK; } else { mm = el % TM; kk = el / TM; }   // coalesce along the unit stride
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
            // bias gradient riding on the weight-gradient GEMM: the whole K (<= one slab) of B sits in LDS buffer 0 (rows >= K are
            // zero); one thread per column adds its K values in row order
            if (e.colsum && m0 == 0 && tid < CEM_TN && n0 + tid < N) {
                float t = 0.f;
#pragma unroll 8
                for (int k = 0; k < CEM_TK; ++k) t = t + Bs[0][k][tid];
                const int n = n0 + tid;
                (n < sp.nsplit ? e.colsum : e.colsum1)[n < sp.nsplit ? n : n - sp.nsplit] = t;
            }
            // epilogue operands, requested now (batched, clamped indices) so that their latency hides behind the k loop
            float bia[NCB], gat[NRB][4][NCB];
#pragma unroll
            for (int jn = 0; jn < NCB; ++jn) {
                const int n = n0 + CW * ch + 16 * jn + li, cn = n < N ? n : N - 1;
                const gcptr bp = cn < sp.nsplit ? e.bias : e.bias1;
                bia[jn] = e.bias ? bp[cn < sp.nsplit ? cn : cn - sp.nsplit] : 0.f;
            }
            if (e.gate) {
#pragma unroll
                for (int rk = 0; rk < NRB; ++rk)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jn = 0; jn < NCB; ++jn) {
                            const int mI = m0 + 16 * rb + 64 * rk + 4 * kq + i, n = n0 + CW * ch + 16 * jn + li;
                            gat[rk][i][jn] = e.gate[(mI < M ? mI : M - 1) * e.ldg + (n < N ? n : N - 1)];
                        }
            }
            for (int kt = 0; kt < nk; ++kt) {
                const int buf = kt & 1;
                if (kt + 1 < nk) fetch((kt + 1) * CEM_TK);
#pragma unroll
                for (int P = 0; P < CEM_TK / 4; ++P) {
                    float a[NRB], b[NCB];
#pragma unroll
                    for (int rk = 0; rk < NRB; ++rk) a[rk] = As[buf][4 * P + kq][16 * rb + 64 * rk + li];
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) b[cb] = Bs[buf][4 * P + kq][CW * ch + 16 * cb + li];
#pragma unroll
                    for (int rk = 0; rk < NRB; ++rk)
#pragma unroll
                        for (int cb = 0; cb < NCB; ++cb) acc[rk][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rk], b[cb], acc[rk][cb], 0, 0, 0);
                }
                if (kt + 1 < nk) stash(buf ^ 1);
                __syncthreads();
            }
#ifdef CEM_STAMPS
            long long t2_ = (long long)__builtin_amdgcn_s_memtime();
#endif
            // epilogue (its bias / gate operands were requested before the k loop)
#pragma unroll
            for (int rk = 0; rk < NRB; ++rk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < NCB; ++jn) {
                        const int mI = m0 + 16 * rb + 64 * rk + 4 * kq + i, n = n0 + CW * ch + 16 * jn + li;
                        float v = acc[rk][jn][i];
                        if (e.bias) v = v + bia[jn];
                        if (e.gate) v = cem_layer_gate(e, v, gat[rk][i][jn]);
                        else if (e.relu) {
                            const float zpre = v;
                            v = e.relu == 1 ? fmaxf(v, 0.f) : cem_activation_fwd(e.relu - 1, v);
                            bool dropped = false;
                            if (e.drop_thresh) { const float kept = cem_dropout_fwd(e, mI, n, 1.0f); dropped = kept == 0.f; v = dropped ? 0.f : v * e.drop_scale; }
                            if (e.outz && mI < M && n < N) e.outz[mI * e.ldo + n] = dropped ? __uint_as_float(CEM_Z_DROPPED) : zpre;
                        }
                        if (mI < M && n < N) (n < sp.nsplit ? e.out : e.out1)[(mI < M ? mI : M - 1) * e.ldo + (n < sp.nsplit ? n : n - sp.nsplit)] = v;
                    }
#ifdef CEM_STAMPS
            if (e.st && blockIdx.x == 0 && tid == 0) { const long long t3_ = (long long)__builtin_amdgcn_s_memtime(); e.st[8] += t1_ - t0_; e.st[9] += t2_ - t1_; e.st[10] += t3_ - t2_; e.st[11] += 1; }
#endif
        }
    }
    __syncthreads();
}

// One-shot form of the 16-row GEMMs (forward layers, heads, dh: M <= 16, N <= 128, K <= 128).  With 16 rows a k slab is eight
// MFMAs per wave — nothing to hide an L2 round trip behind — so the slab pipeline above degenerates into K/32 serial round
// trips.  Here every thread issues ALL its operand loads at once (36 words), the whole A [K][16] and B [K][128] go to LDS
// behind one barrier, and the k loop runs uninterrupted.
__device__ __attribute__((noinline)) void wg_gemm_r16_deep(const int M, const int N, const int K, const gcptr Ag, const int sam, const int sak,
                                                         const gcptr Bg, const int sbk, const int sbn, const GemmEpi e, const GemmSplit sp)
{
    float *lds = g_train_lds;
    typedef float RowA[CEM_TM + CEM_TPAD];
    typedef float RowB[CEM_TN + CEM_TPAD];
    RowA *As = reinterpret_cast<RowA *>(lds);                                      // As[k][m], m < 16 used
    RowB *Bs = reinterpret_cast<RowB *>(lds + CEM_TKMAX * (CEM_TM + CEM_TPAD));      // Bs[k][n]
    constexpr int NEA = 16 * CEM_TKMAX / CEM_TNT, NEB = CEM_TN * CEM_TKMAX / CEM_TNT;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, kq = lane >> 4, li = lane & 15;
    typedef float f4v __attribute__((ext_vector_type(4)));
    float ra[NEA], rbuf[NEB];
    auto a_idx = [&](const int q, int &mm, int &kk) {
        const int el = tid + CEM_TNT * q;
        if (sak == 1) { kk = el % CEM_TKMAX; mm = el / CEM_TKMAX; } else { mm = el % 16; kk = el / 16; }
    };
    auto b_idx = [&](const int q, int &nn, int &kb) {
        const int el = tid + CEM_TNT * q;
        if (sbn == 1) { nn = el % CEM_TN; kb = el / CEM_TN; } else { kb = el % CEM_TKMAX; nn = el / CEM_TKMAX; }
    };
#ifdef CEM_STAMPS
    long long t0_ = (long long)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int q = 0; q < NEA; ++q) {
        int mm, kk; a_idx(q, mm, kk);
        const int cm = mm < M ? mm : M - 1, ck = kk < K ? kk : K - 1;
        const bool a0 = ck < sp.ksplit;
        const gcptr ap = a0 ? Ag : sp.A1;
        ra[q] = ap[cm * sam + (a0 ? ck : ck - sp.ksplit) * sak];
    }
    // a row-major B (the forward GEMMs: W[k][n]) in 16-byte loads and 16-byte LDS stores where N, the row stride and the split
    // point are multiples of 4 words and the operands 16-byte aligned: a quarter of the load / store instructions
    typedef float f4v_ __attribute__((ext_vector_type(4)));
#ifdef CEM_TRAIN_NOQUADS       // diagnostic A/B build
    const bool quads = false;
#else
    const bool quads = sbn == 1 && sbk % 4 == 0 && N % 4 == 0 && (sp.nsplit >= N || sp.nsplit % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(Bg) | reinterpret_cast<uintptr_t>(sp.B1 ? sp.B1 : Bg)) & 15) == 0;
#endif
    auto bq_idx = [&](const int q, int &nn, int &kb) {     // first element of quad q of this thread
        const int el = tid + CEM_TNT * q;
        nn = (el % (CEM_TN / 4)) * 4; kb = el / (CEM_TN / 4);
    };
    if (quads) {
#pragma unroll
        for (int q = 0; q < NEB / 4; ++q) {
            int nn, kb; bq_idx(q, nn, kb);
            // clamp the quad as a whole (N, K and the split points are multiples of 4 here)
            const int cn = nn < N ? nn : N - 4, ck = kb < K ? kb : K - 1;
            const gcptr bp = (ck < sp.ksplit && cn < sp.nsplit) ? Bg : sp.B1;
            const f4v_ v = *reinterpret_cast<const __attribute__((address_space(1))) f4v_ *>(
                bp + (ck < sp.ksplit ? ck : ck - sp.ksplit) * sbk + (cn < sp.nsplit ? cn : cn - sp.nsplit) * sbn);
            rbuf[4 * q] = v[0]; rbuf[4 * q + 1] = v[1]; rbuf[4 * q + 2] = v[2]; rbuf[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < NEB; ++q) {
            int nn, kb; b_idx(q, nn, kb);
            const int cn = nn < N ? nn : N - 1, ck = kb < K ? kb : K - 1;
            const gcptr bp = (ck < sp.ksplit && cn < sp.nsplit) ? Bg : sp.B1;
            rbuf[q] = bp[(ck < sp.ksplit ? ck : ck - sp.ksplit) * sbk + (cn < sp.nsplit ? cn : cn - sp.nsplit) * sbn];
        }
    }
    // epilogue operands ride in the same round trip
    const int n = 16 * wv + li, cn = n < N ? n : N - 1;
    float bia = 0.f, gat[4];
    if (e.bias) { const gcptr bp = cn < sp.nsplit ? e.bias : e.bias1; bia = bp[cn < sp.nsplit ? cn : cn - sp.nsplit]; }
    if (e.gate) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { const int mI = 4 * kq + i; gat[i] = e.gate[(mI < M ? mI : M - 1) * e.ldg + cn]; }
    }
#pragma unroll
    for (int q = 0; q < NEA; ++q) { int mm, kk; a_idx(q, mm, kk); As[kk][mm] = (mm < M && kk < K) ? ra[q] : 0.f; }
    if (quads) {
#pragma unroll
        for (int q = 0; q < NEB / 4; ++q) {
            int nn, kb; bq_idx(q, nn, kb);
            const bool ok = nn < N && kb < K;           // N is a multiple of 4: a quad is inside or outside as a whole
            *reinterpret_cast<f4v_ *>(&Bs[kb][nn]) = ok ? (f4v_){rbuf[4 * q], rbuf[4 * q + 1], rbuf[4 * q + 2], rbuf[4 * q + 3]} : (f4v_){0.f, 0.f, 0.f, 0.f};
        }
    } else {
#pragma unroll
        for (int q = 0; q < NEB; ++q) { int nn, kb; b_idx(q, nn, kb); Bs[kb][nn] = (nn < N && kb < K) ? rbuf[q] : 0.f; }
    }
    __syncthreads();
#ifdef CEM_STAMPS
    long long t1_ = (long long)__builtin_amdgcn_s_memtime();
#endif
    if (e.colsum && tid < N) {                             // bias gradient next to a weight gradient with few rows (inputs_dim <= 16)
        float t = 0.f;
        for (int k = 0; k < K; ++k) t = t + Bs[k][tid];
        (tid < sp.nsplit ? e.colsum : e.colsum1)[tid < sp.nsplit ? tid : tid - sp.nsplit] = t;
    }
    f4v acc = (f4v){0.f, 0.f, 0.f, 0.f};
    const int nP = (K + 3) / 4;
#pragma unroll 8
    for (int P = 0; P < nP; ++P)                           // ascending k: the same accumulation chain as the slab form
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(As[4 * P + kq][li], Bs[4 * P + kq][16 * wv + li], acc, 0, 0, 0);
#ifdef CEM_STAMPS
    long long t2_ = (long long)__builtin_amdgcn_s_memtime();
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mI = 4 * kq + i;
        float v = acc[i];
        if (e.bias) v = v + bia;
        if (e.gate) v = cem_layer_gate(e, v, gat[i]);
        else if (e.relu) {
            const float zpre = v;
            v = e.relu == 1 ? fmaxf(v, 0.f) : cem_activation_fwd(e.relu - 1, v);
            bool dropped = false;
            if (e.drop_thresh) { const float kept = cem_dropout_fwd(e, mI, n, 1.0f); dropped = kept == 0.f; v = dropped ? 0.f : v * e.drop_scale; }
            if (e.outz && mI < M && n < N) e.outz[mI * e.ldo + n] = dropped ? __uint_as_float(CEM_Z_DROPPED) : zpre;
        }
        if (mI < M && n < N) (n < sp.nsplit ? e.out : e.out1)[mI * e.ldo + (n < sp.nsplit ? n : n - sp.nsplit)] = v;
    }
#ifdef CEM_STAMPS
    if (e.st && blockIdx.x == 0 && tid == 0) { const long long t3_ = (long long)__builtin_amdgcn_s_memtime(); e.st[8] += t1_ - t0_; e.st[9] += t2_ - t1_; e.st[10] += t3_ - t2_; e.st[11] += 1; }
#endif
    __syncthreads();
}

__device__ __forceinline__ void wg_gemm(const int M, const int N, const int K, const gcptr Ag, const int sam, const int sak,
                                        const gcptr Bg, const int sbk, const int sbn, const GemmEpi e, const GemmSplit sp)
{
    if (M <= 16 && N <= CEM_TN && K <= CEM_TKMAX) wg_gemm_r16_deep(M, N, K, Ag, sam, sak, Bg, sbk, sbn, e, sp);
    else if (M <= 16) wg_gemm_t<1>(M, N, K, Ag, sam, sak, Bg, sbk, sbn, e, sp);
    else if (M > CEM_TM && K <= CEM_TK) wg_gemm_t<2>(M, N, K, Ag, sam, sak, Bg, sbk, sbn, e, sp);
    else wg_gemm_t<0>(M, N, K, Ag, sam, sak, Bg, sbk, sbn, e, sp);
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
    // workgroup = (member m, row part): rows [part * CEM_TROWS, +Bt) of the member's minibatch of p.Bt rows
    const int m = blockIdx.x / CEM_TPARTS, part = blockIdx.x % CEM_TPARTS, tid = threadIdx.x;
    const int D = p.D, O = p.O, U = p.U, L = p.L;
    const int row0 = part * CEM_TROWS;
    const int Bt = p.Bt - row0 < CEM_TROWS ? p.Bt - row0 : CEM_TROWS;
    if (Bt <= 0) return;                          // a short minibatch: the Adam kernel only adds the parts that exist
    float *W = p.W + (size_t)m * p.nat, *G = p.grad + (size_t)part * p.gpart + (size_t)m * p.nat;
    float *sc = p.scratch + (size_t)blockIdx.x * p.scratch_per_member;
    // scratch carve: every activation matrix has row stride S (128: D, O, U <= 128, narrower units leave columns unused; 256 for wider units)
    const int S = p.ts;
    float *xs = sc;                              // [TROWS][S]   h_0
    float *hs = xs + CEM_TROWS * S;              // [L][TROWS][S] h_1..h_L
    float *mu = hs + (size_t)L * CEM_TROWS * S;  // [TROWS][S]
    float *vp = mu + CEM_TROWS * S;
    float *ys = vp + CEM_TROWS * S;
    float *dmu = ys + CEM_TROWS * S;
    float *dv = dmu + CEM_TROWS * S;
    float *dha = dv + CEM_TROWS * S;
    float *dhb = dha + CEM_TROWS * S;
    float *zs = CEM_ACT_NEEDS_Z(p.act) ? dhb + CEM_TROWS * S : nullptr;      // [L][TROWS][S] pre-activations of swish / gelu layers (the host sizes the scratch for them)
    // natural-blob offsets (cem_mpc.h): W_0,b_0,...,W_mu,b_mu,W_var,b_var
    auto offW = [&](int l) { return l == 0 ? (size_t)0 : (size_t)D * U + U + (size_t)(l - 1) * ((size_t)U * U + U); };
    auto offb = [&](int l) { return offW(l) + (size_t)(l == 0 ? D : U) * U; };
    const size_t oWmu = (size_t)D * U + U + (size_t)(L - 1) * ((size_t)U * U + U), obmu = oWmu + (size_t)U * O;
    const size_t oWv = obmu + O, obv = oWv + (size_t)U * O;

    CEM_TR_STAMP(0);
#ifdef CEM_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) { p.stamps[8] = p.stamps[9] = p.stamps[10] = p.stamps[11] = 0; }
#endif
    // ---- gather this part's rows of the minibatch ---------------------------------------------------------------
    {
        int32_t *rows = reinterpret_cast<int32_t *>(lds);
        if (tid < Bt) rows[tid] = p.perm ? p.perm[(size_t)m * p.nperm + p.offset + row0 + tid] : p.offset + row0 + tid;
        __syncthreads();
        wg_map<float2>(Bt * S,
            [&](int e) { const int r = e / S, c = e % S; const int row = rows[r];
                         return make_float2(c < D ? p.x[(size_t)row * D + c] : 0.f, c < O ? p.y[(size_t)row * O + c] : 0.f); },
            [&](int e, float2 v) { xs[e] = v.x; ys[e] = v.y; });
    }
    __syncthreads();

    CEM_TR_STAMP(1);
    // ---- forward (mlp_ensemble.py:18-22,33-34,59-61) -----------------------------------------------------------
    for (int l = 0; l < L; ++l) {
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TROWS * S;
        float *hout = hs + (size_t)l * CEM_TROWS * S;
        const float *Wl = W + offW(l), *bl = W + offb(l);
        GemmEpi fe{(gptr)hout, S, (gcptr)bl, nullptr, 0, 1 + p.act, nullptr, nullptr, p.stamps, nullptr, nullptr};      // f(h W + b), f = relu unless configured otherwise
        if (p.train && p.drop_thresh) {                                       // Dropout(training=True), mlp_ensemble.py:21,138
            fe.drop_thresh = p.drop_thresh; fe.drop_scale = p.drop_scale; fe.drop_keep = p.drop_keep; fe.drop_step = p.drop_step;
            fe.drop_c2 = ((uint32_t)l << 8) | (3u << 16); fe.drop_member = (uint32_t)m; fe.drop_k0 = p.drop_k0; fe.drop_k1 = p.drop_k1; fe.drop_row0 = row0;
        }
        if (zs && p.train) fe.outz = (gptr)(zs + (size_t)l * CEM_TROWS * S);
        wg_gemm(Bt, U, l == 0 ? D : U, (gcptr)hin, S, 1, (gcptr)Wl, U, 1, fe, CEM_NOSPLIT);
    }
    CEM_TR_STAMP(2);
    const float *hL = hs + (size_t)(L - 1) * CEM_TROWS * S;
    // both heads as ONE GEMM: columns [0, O) = mu head, [O, 2O) = variance head (2O <= 128 fills the tile two N = O GEMMs half use)
    wg_gemm(Bt, 2 * O, U, (gcptr)hL, S, 1, (gcptr)(W + oWmu), O, 1,
            GemmEpi{(gptr)mu, S, (gcptr)(W + obmu), nullptr, 0, 0, (gptr)vp, (gcptr)(W + obv), p.stamps, nullptr, nullptr}, GemmSplit{nullptr, (gcptr)(W + oWv), 0x7fffffff, O});
    CEM_TR_STAMP(3);
    // ---- negative_log_likelihood (:64-67) and its gradient w.r.t. mu and the pre-softplus variance -----------------
    float s_log = 0.f, s_sq = 0.f;
    const float ninv = 1.0f / ((float)p.Bt * (float)O * (float)p.E);          // the mean runs over the WHOLE minibatch (mlp_ensemble.py:64-67)
    wg_map<float3>(Bt * O,
        [&](int e) { const int r = e / O, c = e % O; return make_float3(vp[r * S + c], mu[r * S + c], ys[r * S + c]); },
        [&](int e, float3 in) {
            const int r = e / O, c = e % O;
            const float v = in.x, var = train_softplus(v) + 1e-4f;
            const float diff = in.y - in.z;
            s_log += logf(6.283185307179586f * var);
            s_sq += diff * diff / var;
            if (p.train) {
                dmu[r * S + c] = diff / var * ninv;
                const float dvar = (0.5f / var - 0.5f * diff * diff / (var * var)) * ninv;
                dv[r * S + c] = dvar / (1.0f + expf(-v));               // d softplus(v)/dv = sigmoid(v)
            }
        });
    s_log = block_sum(s_log, red);
    s_sq = block_sum(s_sq, red);
    CEM_TR_STAMP(4);
    // this part's share of the two sums of the loss; the Adam kernel (training) or the host (validation) adds the parts in order
    if (tid == 0) { p.loss_part[((size_t)m * CEM_TPARTS + part) * 2] = s_log; p.loss_part[((size_t)m * CEM_TPARTS + part) * 2 + 1] = s_sq; }
    if (!p.train) return;
    __syncthreads();

    CEM_TR_STAMP(5);
    // ---- backward ------------------------------------------------------------------------------------------------
    // [dW_mu | dW_var] = h_L^T [dmu | dv] as one GEMM
    wg_gemm(U, 2 * O, Bt, (gcptr)hL, 1, S, (gcptr)dmu, S, 1,
            GemmEpi{(gptr)(G + oWmu), O, nullptr, nullptr, 0, 0, (gptr)(G + oWv), nullptr, p.stamps, (gptr)(G + obmu), (gptr)(G + obv)},
            GemmSplit{nullptr, (gcptr)dv, 0x7fffffff, O});                    // + [db_mu | db_var] = column sums of [dmu | dv]
    // dh_L = (dmu Wmu^T + dv Wvar^T) * relu'(h_L): the relu mask rides in the epilogue of the GEMM that completes dh
    // dh_L = ([dmu | dv] [W_mu | W_var]^T) * relu'(h_L): one GEMM over K = 2O; the relu mask rides in its epilogue
    wg_gemm(Bt, U, 2 * O, (gcptr)dmu, S, 1, (gcptr)(W + oWmu), 1, O,
            GemmEpi{(gptr)dha, S, nullptr, (gcptr)(zs ? zs + (size_t)(L - 1) * CEM_TROWS * S : hL), S, 1 + p.act, nullptr, nullptr, p.stamps, nullptr, nullptr, p.drop_thresh, p.drop_scale, p.drop_keep}, GemmSplit{(gcptr)dv, (gcptr)(W + oWv), O, 0x7fffffff});
    CEM_TR_STAMP(6);
    float *dcur = dha, *dnext = dhb;
    for (int l = L - 1; l >= 0; --l) {
        const float *hin = l == 0 ? xs : hs + (size_t)(l - 1) * CEM_TROWS * S;
        const int in = l == 0 ? D : U;
        wg_gemm(in, U, Bt, (gcptr)hin, 1, S, (gcptr)dcur, S, 1, GemmEpi{(gptr)(G + offW(l)), U, nullptr, nullptr, 0, 0, nullptr, nullptr, p.stamps, (gptr)(G + offb(l)), nullptr}, CEM_NOSPLIT);   // dW_l = h_{l-1}^T dh_l, db_l = column sums of dh_l
        if (l > 0) {
            wg_gemm(Bt, U, U, (gcptr)dcur, S, 1, (gcptr)(W + offW(l)), 1, U, GemmEpi{(gptr)dnext, S, nullptr, (gcptr)(zs ? zs + (size_t)(l - 1) * CEM_TROWS * S : hin), S, 1 + p.act, nullptr, nullptr, p.stamps, nullptr, nullptr, p.drop_thresh, p.drop_scale, p.drop_keep}, CEM_NOSPLIT);   // dh_{l-1} = (dh_l W_l^T) f'(z_{l-1})
            float *t = dcur; dcur = dnext; dnext = t;
        }
    }
    CEM_TR_STAMP(7);
}

// ---- Adam with clipvalue (mlp_ensemble.py:113-117,143-144), every member's parameters in one grid ---------------------
__global__ __launch_bounds__(256) void cem_adam_kernel(const TrainParams p)
{
    const size_t n = (size_t)p.E * p.nat, n4 = n / 4;
    const int nparts = (p.Bt + CEM_TROWS - 1) / CEM_TROWS;          // row parts that ran this step (a short last minibatch has fewer)
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
        float4 gp[CEM_TPARTS];
#pragma unroll
        for (int q = 0; q < CEM_TPARTS; ++q) gp[q] = G4[(q < nparts ? (size_t)q : 0) * (p.gpart / 4) + e];      // all parts' loads in flight at once
        float4 g = gp[0];                                            // dW = sum over the row parts, part 0 first: a fixed order
#pragma unroll
        for (int q = 1; q < CEM_TPARTS; ++q)
            if (q < nparts) { g.x = g.x + gp[q].x; g.y = g.y + gp[q].y; g.z = g.z + gp[q].z; g.w = g.w + gp[q].w; }
        float4 mo = M4[e], vo = V4[e], w = W4[e];
        upd(g.x, mo.x, vo.x, w.x); upd(g.y, mo.y, vo.y, w.y); upd(g.z, mo.z, vo.z, w.z); upd(g.w, mo.w, vo.w, w.w);
        M4[e] = mo; V4[e] = vo; W4[e] = w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t e = n4 * 4 + threadIdx.x;
        float g = p.grad[e];
        for (int q = 1; q < nparts; ++q) g = g + p.grad[(size_t)q * p.gpart + e];
        upd(g, p.Mo[e], p.Vo[e], p.W[e]);
    }
    // training_step's return value, per member: negative_log_likelihood / ensemble_size (mlp_ensemble.py:64-67,139-141)
    if (blockIdx.x == 0) for (int m = threadIdx.x; m < p.E; m += blockDim.x) {       // any ensemble size (the reference takes any)
        float s_log = 0.f, s_sq = 0.f;
        for (int q = 0; q < nparts; ++q) { s_log = s_log + p.loss_part[((size_t)m * CEM_TPARTS + q) * 2]; s_sq = s_sq + p.loss_part[((size_t)m * CEM_TPARTS + q) * 2 + 1]; }
        const float cnt = (float)p.Bt * (float)p.O;
        p.loss_out[m] = (0.5f * s_log / cnt + 0.5f * s_sq / cnt) / (float)p.E;
    }
}
