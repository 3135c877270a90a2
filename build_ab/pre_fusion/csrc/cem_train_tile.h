// cem_train_tile.h — the ensemble training step in the style of the rollout kernel:
// MlpEnsemble.training_step / validation_step, simba/models/mlp_ensemble.py:134-155, loss negative_log_likelihood (:64-67).
//
// A workgroup of 8 waves (two per SIMD: one's MFMA chain covers the other's LDS / barrier / load stalls) takes 16 rows of ONE member's minibatch through the forward pass, the loss and the whole backward
// pass without leaving the CU: every product is a chain of v_mfma_f32_16x16x4_f32 on 16-feature blocks, the activations of
// all layers stay in LDS in the accumulator layout (lane = (feature quad q, row j), register r = feature 4q + r of the block:
// what one layer's MFMA writes IS the next layer's B operand, as in cem_rollout_kernel), and the weights are read straight
// from their natural Keras layout ([in][out]) with per-lane addressing, so the Adam kernel keeps one copy of the weights and
// nothing is re-packed per step.
//   forward   h_l^T [U x rows]   = relu(W_l^T h_{l-1}^T + b_l)      A = W_l[k][out], B = h_{l-1} block
//   heads     mu, v              = W_mu^T h_L^T + b, W_var^T h_L^T + b
//   loss      dmu, dv, partial sums of the NLL                       elementwise on the accumulators
//   backward  dh_{l-1}^T         = (W_l dh_l^T) * relu'(h_{l-1})      A = W_l[in][k], B = dh_l block
//   weights   dW_l [in x out]    = h_{l-1}^T dh_l                     A, B gathered from LDS with the row as the k index
//   biases    db_l               = sum over rows of dh_l               16-lane reductions of the accumulators
// Each workgroup writes PARTIAL gradients (its 16 rows); the Adam kernel adds a member's parts in a fixed order.
//
// Weight traffic.  A stage (one layer for one wave: 1 output block x 8 k blocks; 2 for the heads) needs 32 words per lane and 32
// MFMAs that take 1 K cycles — less than one L2 round trip with nothing else in flight.  The weights do not depend on the activations, so every
// stage's loads are issued a whole stage AHEAD into a second register buffer (the layer count is a template parameter: the
// stage sequence, and with it every register index, is fixed at compile time).  The loads are raw buffer loads: the k offset
// lives in an SGPR, the lane offset is one VGPR per accumulator, and an out-of-range row or column block reads as zero.
#pragma once
#include "cem_train.h"

#define CEM_TT_BLK 1152                      // bytes of one 16-feature x 16-row block in LDS: [4 feature quads][16 rows][4 words], each
                                             // quad's 256 B followed by 32 B of padding: the dW products read a block TRANSPOSED (lane =
                                             // (row quad, feature)), and without the skew the four feature quads of a row share a bank
#define CEM_TT_LANE(c) ((c).lane * 16 + (c).q * 32)      // byte offset of lane (q, j)'s four words inside a block
#define CEM_TT_NB 8                          // blocks per activation matrix (128 features)
#define CEM_TT_WAVES 8                       // waves per workgroup: wave w owns 16-feature block w of every activation matrix
#define CEM_TT_MAXL 6                        // layer counts with their own instantiation (the reference ships 4)

struct TtCtx { int lane, q, j, w, cnt; };

__device__ __forceinline__ float tt_row_sum(float v)          // sum over the 16 rows (lanes j) of a feature: four DPP row rotations
{                                                             // (fixed order; every lane of the row ends with the sum; no LDS traffic)
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
    return v;
}

// One weight operand of a stage: matrix W (a buffer resource over exactly its words), A(F) at MFMA step r on lane (q, j) =
// W[(16F + 4q + r) * sk + (mb + j) * sm]: forward sk = row stride, sm = 1 (k runs down the rows); backward sk = 1, sm = row stride.
struct TtOp {
    __amdgpu_buffer_rsrc_t rsrc;
    int lane_off;                            // bytes: (4q * sk + (mb + j) * sm) * 4, or an out-of-range offset when mb + j is past the matrix
    int sk4;                                 // bytes per k
};

__device__ __forceinline__ TtOp tt_op(const gcptr W, const int words, const int sk, const int sm, const int mb, const int Mdim, const TtCtx &c)
{
    TtOp o;
    o.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((const float *)W), 0, words * 4, 0x00020000);
    o.lane_off = (mb + c.j < Mdim) ? (4 * c.q * sk + (mb + c.j) * sm) * 4 : 0x7fffff00;      // past the end: the load returns 0
    o.sk4 = sk * 4;
    return o;
}

// issue the loads of a stage: wv[F][a][r] for ALL eight k blocks, branch-free (a k block past the matrix reads zeros in the
// forward form — its offsets are past the end of the buffer; in the backward form it may alias the next row, and the matching B
// values are exact zeros: every LDS block past a matrix's width is kept zero)
template <int NACC>
__device__ __forceinline__ void tt_load(float (&wv)[CEM_TT_NB][2][4], const TtOp (&op)[NACC])
{
#pragma unroll
    for (int F = 0; F < CEM_TT_NB; ++F)
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                wv[F][a][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(op[a].rsrc, op[a].lane_off, (16 * F + r) * op[a].sk4, 0));
    // pin the stage's loads HERE: left alone the scheduler hoists later stages' loads as well (renaming their registers: 512 VGPRs
    // and spills) or sinks these to their uses (one L2 round trip per k block)
    __builtin_amdgcn_sched_barrier(0);
}

// the backward form of a stage's loads when the matrix's row length is a multiple of 4 words: the four MFMA steps r of a k block
// are four CONSECUTIVE words of a row (sk = 1), so one 16-byte load brings what four 4-byte loads did — and those touched 16 rows
// x 4 separate quads per instruction.  A quad past the row's end is whole (row length % 4 == 0) and only meets zero B values.
template <int NACC>
__device__ __forceinline__ void tt_load_rows(float (&wv)[CEM_TT_NB][2][4], const TtOp (&op)[NACC])
{
#pragma unroll
    for (int F = 0; F < CEM_TT_NB; ++F)
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(op[a].rsrc, op[a].lane_off, 64 * F, 0));
#pragma unroll
            for (int r = 0; r < 4; ++r) wv[F][a][r] = v[r];
        }
    __builtin_amdgcn_sched_barrier(0);
}

// acc[a] += sum over the eight k blocks of A_a(F) . B(F), B(F) = LDS block F of `bsrc`
template <int NACC>
__device__ __forceinline__ void tt_mfma(f4 (&acc)[NACC], const float (&wv)[CEM_TT_NB][2][4], const char *bsrc, const TtCtx &c)
{
#pragma unroll
    for (int F = 0; F < CEM_TT_NB; ++F) {
        const f4 hb = *reinterpret_cast<const f4 *>(bsrc + F * CEM_TT_BLK + CEM_TT_LANE(c));
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[F][a][r], hb[r], acc[a], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
}

// element (feature j of the block, row 4P + q) of an LDS block in the accumulator layout: the operands of the dW products,
// where the ROW is the contraction index
__device__ __forceinline__ float tt_gather(const char *blk, const int P, const TtCtx &c)
{
    return *reinterpret_cast<const float *>(blk + (c.j >> 2) * 288 + ((4 * P + c.q) * 4 + (c.j & 3)) * 4);
}

// dW[in][out] partial of one layer: this wave owns the in-feature block Gi = w (< nIn) and the out blocks F < NF (8, or 4 where the
// matrix is at most 64 wide — the heads of a 60-dimensional observation: half the MFMAs):
// dW[16Gi + 4q + r][16F + j] = sum over rows.  hsrc / dsrc: LDS activations of the layer's input / the gradient of its output.
template <int NF>
__device__ __forceinline__ void tt_dw_n(const char *hsrc, const char *dsrc, const int nIn, const int inDim, const int outDim,
                                        float *Gw, const int ldw, const TtCtx &c)
{
#ifdef CEM_TT_DBG_NODW
    return;
#endif
    const int Gi = c.w;
    if (Gi < nIn) {                                                           // wave-uniform
        f4 acc[NF];
#pragma unroll
        for (int F = 0; F < NF; ++F) acc[F] = (f4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int P = 0; P < 4; ++P) {
            const float a = tt_gather(hsrc + Gi * CEM_TT_BLK, P, c);
#pragma unroll
            for (int F = 0; F < NF; ++F)                                      // out blocks past the width hold zeros
                acc[F] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, tt_gather(dsrc + F * CEM_TT_BLK, P, c), acc[F], 0, 0, 0);
        }
#pragma unroll
        for (int F = 0; F < NF; ++F) {
            const int n = 16 * F + c.j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mI = 16 * Gi + 4 * c.q + r;
#ifdef CEM_TT_DBG_NOSTORE
                if (mI < inDim && n < outDim && acc[F][r] == 123.456f) Gw[(size_t)mI * ldw + n] = acc[F][r];
#else
                if (mI < inDim && n < outDim) Gw[(size_t)mI * ldw + n] = acc[F][r];
#endif
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void tt_dw(const char *hsrc, const char *dsrc, const int nIn, const int inDim, const int outDim,
                                      float *Gw, const int ldw, const TtCtx &c)
{
    if (outDim <= 64) tt_dw_n<4>(hsrc, dsrc, nIn, inDim, outDim, Gw, ldw, c);
    else tt_dw_n<8>(hsrc, dsrc, nIn, inDim, outDim, Gw, ldw, c);
}

template <int L>
__global__ __launch_bounds__(64 * CEM_TT_WAVES) void cem_train_tile_kernel(const TrainParams p)
{
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    __shared__ float red[2][CEM_TT_WAVES];
    __shared__ int32_t rows_s[CEM_TROWS];
    const int m = blockIdx.x / CEM_TPARTS, part = blockIdx.x % CEM_TPARTS, tid = threadIdx.x;
    const int D = p.D, O = p.O, U = p.U;
    // rows of this workgroup: chunk blockIdx.y of the launch (training launches have one chunk = the minibatch), part `part` of it
    const int chunk0 = (int)blockIdx.y * p.chunk;
    const int Bt = p.Bt - chunk0 < p.chunk ? p.Bt - chunk0 : p.chunk;
    const int row0 = part * CEM_TROWS;
    const int cnt = Bt - row0 < CEM_TROWS ? Bt - row0 : CEM_TROWS;
    if (cnt <= 0) return;                          // a short minibatch: the Adam kernel only adds the parts that exist
    CEM_TR_STAMP(0);
    TtCtx c; c.lane = tid & 63; c.q = c.lane >> 4; c.j = c.lane & 15; c.w = __builtin_amdgcn_readfirstlane(tid >> 6); c.cnt = cnt;
    const gcptr W = (gcptr)(p.W + (size_t)m * p.nat);
    float *G = p.grad + (size_t)part * p.gpart + (size_t)m * p.nat;
    // natural-blob offsets (cem_mpc.h): W_0,b_0,...,W_mu,b_mu,W_var,b_var
    auto offW = [&](int l) { return l == 0 ? (size_t)0 : (size_t)D * U + U + (size_t)(l - 1) * ((size_t)U * U + U); };
    auto offb = [&](int l) { return offW(l) + (size_t)(l == 0 ? D : U) * U; };
    const size_t oWmu = (size_t)D * U + U + (size_t)(L - 1) * ((size_t)U * U + U), obmu = oWmu + (size_t)U * O;
    const size_t oWv = obmu + O, obv = oWv + (size_t)U * O;
    const int nbD = (D + 15) >> 4, nbU = (U + 15) >> 4, nbO = (O + 15) >> 4;
    const bool own = c.w < nbU;                                    // this wave has hidden-unit block w (wave-uniform)
    const bool ownO = c.w < nbO;                                   // ... and head block w
    const int mb = 16 * c.w;
    // LDS: act[0..L] (layer inputs / outputs), dh ping-pong, dmu | dv
    char *act = tsm;                                               // [(L + 1)][8 blocks]
    char *dbuf = tsm + (size_t)(L + 1) * CEM_TT_NB * CEM_TT_BLK;   // [2][8 blocks]
    char *dhd = dbuf + 2 * CEM_TT_NB * CEM_TT_BLK;                 // [16 blocks]: dmu blocks 0.., dv blocks 8..

    // Stage sequence (compile time): 0..L-1 the hidden layers, L the heads (mu and var of block w), L+1 / L+2 the W_mu / W_var parts
    // of dh_L, L+3+i the dh of layer L-1-i.  Stage s's weights sit in wb[s & 1] and are requested during stage s - 1.
    float wb[2][CEM_TT_NB][2][4];
    auto fwd_op = [&](const int l) { return tt_op(W + offW(l), (l == 0 ? D : U) * U, U, 1, mb, U, c); };
    auto bwd_op = [&](const gcptr Wm, const int words, const int ld) { return tt_op(Wm, words, 1, ld, mb, U, c); };
    auto load_bwd = [&](float (&dst)[CEM_TT_NB][2][4], const gcptr Wm, const int words, const int ld) {
        TtOp op[1] = {bwd_op(Wm, words, ld)};
        if (own) { if (ld & 3) tt_load<1>(dst, op); else tt_load_rows<1>(dst, op); }
    };
    {   // the first layer's weights go out before anything else
        TtOp op[1] = {fwd_op(0)};
        if (own) tt_load<1>(wb[0], op);
    }

    // every bias this wave will start an accumulator from, requested now: a load issued right before its stage would expose a full
    // L2 round trip at each of the L + 1 forward stages
    f4 bias[L], bias_mu4, bias_v4;
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int o = mb + 4 * c.q + r; bias[l][r] = W[offb(l) + (o < U ? o : 0)]; }
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int o = mb + 4 * c.q + r; bias_mu4[r] = W[obmu + (o < O ? o : 0)]; bias_v4[r] = W[obv + (o < O ? o : 0)]; }

    if (tid < CEM_TROWS) {
        const int rr = tid < cnt ? tid : cnt - 1;                  // rows past the end repeat the last one; their gradients are masked to zero
        rows_s[tid] = p.perm ? p.perm[(size_t)m * p.nperm + p.offset + chunk0 + row0 + rr] : p.offset + chunk0 + row0 + rr;
    }
    __syncthreads();
    const int myrow = rows_s[c.j];
    // targets of this lane's row for the head block this wave owns (requested now, needed after the forward pass)
    f4 yt;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int o = mb + 4 * c.q + r; yt[r] = p.y[(size_t)myrow * O + (o < O ? o : O - 1)]; }
    // ---- h_0 = the gathered, already scaled inputs: wave w brings block w (a block past the input width: zeros) ----------------
    {
        f4 x;
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int f = mb + 4 * c.q + r; x[r] = f < D ? p.x[(size_t)myrow * D + f] : 0.f; }
        *reinterpret_cast<f4 *>(act + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = x;
        // the head-gradient blocks this wave would own: zero until (unless) the loss writes them
        *reinterpret_cast<f4 *>(dhd + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = (f4){0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f4 *>(dhd + (CEM_TT_NB + c.w) * CEM_TT_BLK + CEM_TT_LANE(c)) = (f4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    CEM_TR_STAMP(1);

    // ---- forward (mlp_ensemble.py:18-22,59-61): wave w computes output block w of every layer ----------------------------------
#pragma unroll
    for (int l = 0; l < L; ++l) {
        // next stage's weights: the next layer, or this wave's head block (mu and var)
        if (l + 1 < L) { TtOp op[1] = {fwd_op(l + 1)}; if (own) tt_load<1>(wb[(l + 1) & 1], op); }
        else {
            TtOp op[2] = {tt_op(W + oWmu, U * O, O, 1, mb, O, c), tt_op(W + oWv, U * O, O, 1, mb, O, c)};
            if (ownO) tt_load<2>(wb[(l + 1) & 1], op);
        }
        f4 acc[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int o = mb + 4 * c.q + r; acc[0][r] = o < U ? bias[l][r] : 0.f; }
        if (own) tt_mfma<1>(acc, wb[l & 1], act + (size_t)l * CEM_TT_NB * CEM_TT_BLK, c);
        f4 h = acc[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = (mb + 4 * c.q + r < U) ? fmaxf(h[r], 0.f) : 0.f;     // units past U stay exactly zero
        *reinterpret_cast<f4 *>(act + (size_t)(l + 1) * CEM_TT_NB * CEM_TT_BLK + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = h;
        __syncthreads();
        CEM_TR_STAMP(2 + l);
    }
    const char *hL = act + (size_t)L * CEM_TT_NB * CEM_TT_BLK;

    // ---- heads (mlp_ensemble.py:33-34) + negative_log_likelihood (:64-67) and its gradients: stage L ----------------------------
    float s_log = 0.f, s_sq = 0.f;
    const float ninv = 1.0f / ((float)Bt * (float)O * (float)p.E);         // the mean runs over the WHOLE minibatch
    if (p.train) load_bwd(wb[(L + 1) & 1], W + oWmu, U * O, O);            // stage L + 1: the W_mu part of dh_L
    if (ownO) {                                                            // wave-uniform
        f4 acc[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int o = mb + 4 * c.q + r; acc[0][r] = o < O ? bias_mu4[r] : 0.f; acc[1][r] = o < O ? bias_v4[r] : 0.f; }
        tt_mfma<2>(acc, wb[L & 1], hL, c);
        f4 dmu = (f4){0.f, 0.f, 0.f, 0.f}, dv = dmu;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = mb + 4 * c.q + r;
            const bool live = o < O && c.j < cnt;
            const float v = acc[1][r], var = train_softplus(v) + 1e-4f;
            const float diff = acc[0][r] - yt[r];
            if (live) {
                s_log += logf(6.283185307179586f * var);
                s_sq += diff * diff / var;
                dmu[r] = diff / var * ninv;
                const float dvar = (0.5f / var - 0.5f * diff * diff / (var * var)) * ninv;
                dv[r] = dvar / (1.0f + expf(-v));                           // d softplus(v)/dv = sigmoid(v)
            }
        }
        if (p.train) {
            *reinterpret_cast<f4 *>(dhd + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = dmu;
            *reinterpret_cast<f4 *>(dhd + (CEM_TT_NB + c.w) * CEM_TT_BLK + CEM_TT_LANE(c)) = dv;
            // bias gradients of the heads: sums over the rows
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = mb + 4 * c.q + r;
                const float a = tt_row_sum(dmu[r]), b = tt_row_sum(dv[r]);
                if (c.j == 0 && o < O) { G[obmu + o] = a; G[obv + o] = b; }
            }
        }
    }
    // this part's share of the two sums of the loss; the Adam kernel (training) or the host (validation) adds the parts in order
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s_log += __shfl_xor(s_log, d); s_sq += __shfl_xor(s_sq, d); }
    if (c.lane == 0) { red[0][c.w] = s_log; red[1][c.w] = s_sq; }
    __syncthreads();
    if (tid == 0) {
        float a = red[0][0], b = red[1][0];
#pragma unroll
        for (int w = 1; w < CEM_TT_WAVES; ++w) { a += red[0][w]; b += red[1][w]; }
        float *lp = p.loss_part + (((size_t)blockIdx.y * p.E + m) * CEM_TPARTS + part) * 2;
        lp[0] = a; lp[1] = b;
    }
    CEM_TR_STAMP(2 + L);
    if (!p.train) return;

    // ---- backward ---------------------------------------------------------------------------------------------------------
    load_bwd(wb[(L + 2) & 1], W + oWv, U * O, O);                           // stage L + 2: the W_var part of dh_L
    // dh_L = (W_mu dmu^T + W_var dv^T) * relu'(h_L)
    {
        f4 acc[1] = {(f4){0.f, 0.f, 0.f, 0.f}};
        if (own) tt_mfma<1>(acc, wb[(L + 1) & 1], dhd, c);
        if (L > 1) load_bwd(wb[(L + 3) & 1], W + offW(L - 1), U * U, U);    // stage L + 3: dh_{L-1}
        if (own) tt_mfma<1>(acc, wb[(L + 2) & 1], dhd + CEM_TT_NB * CEM_TT_BLK, c);
        const f4 h = *reinterpret_cast<const f4 *>(hL + c.w * CEM_TT_BLK + CEM_TT_LANE(c));
        f4 d = acc[0];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = h[r] > 0.f ? d[r] : 0.f;
        *reinterpret_cast<f4 *>(dbuf + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = d;
        // db_{L-1}: sums over the rows
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int o = mb + 4 * c.q + r;
            const float sum = tt_row_sum(d[r]);
            if (c.j == 0 && o < U) G[offb(L - 1) + o] = sum;
        }
    }
    // [dW_mu | dW_var] = h_L^T [dmu | dv]: off the dh chain's critical path (the other wave of the SIMD runs ahead meanwhile)
    tt_dw(hL, dhd, nbU, U, O, G + oWmu, O, c);
    tt_dw(hL, dhd + CEM_TT_NB * CEM_TT_BLK, nbU, U, O, G + oWv, O, c);
    __syncthreads();
    CEM_TR_STAMP(3 + L);
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        const int st = L + 3 + (L - 1 - l);                                 // the stage that consumes W_l
        const int cur = (L - 1 - l) & 1;
        const int in = l == 0 ? D : U, nIn = l == 0 ? nbD : nbU;
        const char *hin = act + (size_t)l * CEM_TT_NB * CEM_TT_BLK;
        const char *dcur = dbuf + (size_t)cur * CEM_TT_NB * CEM_TT_BLK;
        if (l > 1) load_bwd(wb[(st + 1) & 1], W + offW(l - 1), U * U, U);
        if (l > 0) {
            // dh_{l-1} = (W_l dh_l^T) * relu'(h_{l-1});  db_{l-1} = its row sums
            f4 acc[1] = {(f4){0.f, 0.f, 0.f, 0.f}};
            if (own) tt_mfma<1>(acc, wb[st & 1], dcur, c);
            char *dnext = dbuf + (size_t)(cur ^ 1) * CEM_TT_NB * CEM_TT_BLK;
            const f4 h = *reinterpret_cast<const f4 *>(hin + c.w * CEM_TT_BLK + CEM_TT_LANE(c));
            f4 d = acc[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) d[r] = h[r] > 0.f ? d[r] : 0.f;
            *reinterpret_cast<f4 *>(dnext + c.w * CEM_TT_BLK + CEM_TT_LANE(c)) = d;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = mb + 4 * c.q + r;
                const float sum = tt_row_sum(d[r]);
                if (c.j == 0 && o < U) G[offb(l - 1) + o] = sum;
            }
        }
        tt_dw(hin, dcur, nIn, in, U, G + offW(l), U, c);                    // dW_l = h_{l-1}^T dh_l
        if (l > 0) __syncthreads();
        CEM_TR_STAMP(4 + L + (L - 1 - l));
    }
}
