// cem_rollout_split.h — the rollout with every fp32 product formed on the bf16 matrix pipe from EXACT three-way splits
// (config field `precision` = CEM_PRECISION_SPLIT_BF16X3; opt-in, never the default).
//
// An fp32 number is the exact sum of three bf16 numbers, x = x0 + x1 + x2 (8 significand bits each: x0 = RN(x), x1 = RN(x - x0),
// x2 = x - x0 - x1, round to nearest even).  A product w * x is then the sum of nine bf16 x bf16 products, each exact in the fp32
// accumulator of v_mfma_f32_16x16x32_bf16; the six with i + j <= 2 carry everything down to 2^-23 of the product — the scale of
// one fp32 rounding — and the other three are dropped.  Six bf16 MFMAs of K = 32 replace eight fp32
// MFMAs of K = 4 per (two input blocks x one output block): 96 matrix-pipe cycles instead of 256.  (A wave's VALU work still adds
// to its MFMA time — interleaving the publish with the next row-chunk's MFMAs gained nothing, profiles/r03_split_tile_sizes.txt —
// so the split itself, 44 VALU per 8 values, is part of the price.)  Not bit-identical to the fp32 kernels (the products are summed
// inside the MFMA, 32 at a time), same error scale; parity is measured against the same oracle at the same tolerances
// (tests/test_gpu_split.py).
//
// Structure: cem_rollout_tile's — 4 waves per tile of 16 RC rows of one member for the whole horizon, wave w computes output blocks
// 2w, 2w + 1 of a hidden layer and keeps them, SPLIT, as its own K = 32 input chunk of the next stage; the other chunks travel through
// LDS as three bf16 planes; weights stream from a per-wave image of 6 KB groups (chunk x {a, b} output block x 3 planes) in visiting
// order.  The epilogue, scorer terms and bookkeeping are cem_rollout_tile's own macros.  Whole-horizon tiles only (no floating
// segments); MODE 1 also serves cem_unfold_sequences (trajectory / head-moment outputs).
#pragma once
#include "cem_device.h"

typedef __bf16 cem_bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int cem_u2 __attribute__((ext_vector_type(2)));
#define CEM_MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(cem_bf8, (a)), __builtin_bit_cast(cem_bf8, (b)), (c), 0, 0, 0)
#define CEM_SPLIT_CHUNKS 4                   // K = 32 chunks of a 128-feature layer

// x -> (x0, x1, x2) as bit patterns whose upper halves are the bf16 pieces: x0 = RN(x), x1 = RN(x - x0), x2 = x - x0 - x1 with RN =
// round to nearest even at 8 significand bits (what v_cvt_pk_bf16_f32 does).  Both subtractions are exact and x2 has at most 8
// significant bits, so x = x0 + x1 + x2 exactly (|x| below the largest bf16, 3.39e38) with |x1| <= 2^-8 |x|, |x2| <= 2^-16 |x|:
// the three dropped products are below 2^-23 of w x.  (A truncation split is exact too, but its pieces only shrink by 2^-7 each:
// the dropped terms reach 2^-21; tests/test_split_cpu.py.)  Host version (weights); the device splits pairs, below.
__host__ __device__ inline unsigned cem_rn_bf16_bits(const float x)
{
    union { float f; unsigned u; } v; v.f = x;
    return (v.u + 0x7FFFu + ((v.u >> 16) & 1u)) & 0xFFFF0000u;
}
__host__ __device__ inline void cem_split3_bits(const float x, unsigned &a0, unsigned &a1, unsigned &a2)
{
    union { float f; unsigned u; } t, r1, r2;
    a0 = cem_rn_bf16_bits(x);
    t.u = a0; r1.f = x - t.f; a1 = cem_rn_bf16_bits(r1.f);
    t.u = a1; r2.f = r1.f - t.f; a2 = r2.u;                 // exactly a bf16 value already
}

typedef __bf16 cem_bf2 __attribute__((ext_vector_type(2)));
// two values -> their three pieces, one packed word per plane: 11 VALU instructions (3 v_cvt_pk_bf16_f32, 4 shifts / masks, 4 subtractions)
__device__ __forceinline__ void cem_split_pair(const float x0, const float x1, unsigned &p0, unsigned &p1, unsigned &p2)
{
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){x0, x1}, cem_bf2));
    const float r1l = x0 - __uint_as_float(p0 << 16), r1h = x1 - __uint_as_float(p0 & 0xFFFF0000u);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r1l, r1h}, cem_bf2));
    const float r2l = r1l - __uint_as_float(p1 << 16), r2h = r1h - __uint_as_float(p1 & 0xFFFF0000u);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector((f2){r2l, r2h}, cem_bf2));
}
// 8 values (two accumulator quads: features 4q..4q+3 of two 16-feature blocks) -> three planes of 8 bf16
__device__ __forceinline__ void cem_split8(const f4 x0, const f4 x1, cem_u4 (&p)[3])
{
    unsigned a[4], b[4], c[4];
    cem_split_pair(x0[0], x0[1], a[0], b[0], c[0]);
    cem_split_pair(x0[2], x0[3], a[1], b[1], c[1]);
    cem_split_pair(x1[0], x1[1], a[2], b[2], c[2]);
    cem_split_pair(x1[2], x1[3], a[3], b[3], c[3]);
    p[0] = (cem_u4){a[0], a[1], a[2], a[3]}; p[1] = (cem_u4){b[0], b[1], b[2], b[3]}; p[2] = (cem_u4){c[0], c[1], c[2], c[3]};
}
// 4 values (one feature quad of one block) -> three planes of 4 bf16
__device__ __forceinline__ void cem_split4(const f4 x, cem_u2 (&p)[3])
{
    unsigned a[2], b[2], c[2];
    cem_split_pair(x[0], x[1], a[0], b[0], c[0]);
    cem_split_pair(x[2], x[3], a[1], b[1], c[1]);
    p[0] = (cem_u2){a[0], a[1]}; p[1] = (cem_u2){b[0], b[1]}; p[2] = (cem_u2){c[0], c[1]};
}

// chunk visited at position phi of a hidden / heads stage by wave w: its own chunk (blocks 2w, 2w + 1) first, the others ascending
__host__ __device__ inline int cem_split_perm(int w, int phi) { return phi == 0 ? w : (phi - 1 < w ? phi - 1 : phi); }

// LDS: [2 buffers][RC][4 chunks][3 planes][64 lanes][16 B]; a lane's 16 bytes = 4 bf16 of block 2c, 4 bf16 of block 2c + 1
#define CEM_SPLIT_XB(RC_) ((RC_) * CEM_SPLIT_CHUNKS * 3 * 1024)
__device__ __forceinline__ int cem_split_off(const int c_rc, const int chunk, const int plane, const int lane)
{
    return (((c_rc * CEM_SPLIT_CHUNKS + chunk) * 3 + plane) * 64 + lane) * 16;
}

// Weight ring over the wave's stream of 6 KB groups: [a planes 0..2][b planes 0..2], a plane = [64 lanes][8 bf16].  CEM_SPLIT_RING
// slots, RING - 1 groups ahead of the MFMAs; every stage is a multiple of RING chunks long, so the slot of a stage's chunk phi is
// the compile-time constant phi % RING (with RING 4 the two-chunk layer 0 of the obs+act <= 64 family is padded with two
// zero-weight chunks: cem_split_l0_chunks, host and device).
#ifndef CEM_SPLIT_RING
#define CEM_SPLIT_RING 2                       // measured 2 vs 4 (same box): B1 0.123 vs 0.133 ms, B2 0.358 vs 0.510, B3 2.93 vs 3.12, B4 1.60 vs 1.86 (48 more VGPRs cost a resident workgroup)
#endif
__host__ __device__ constexpr int cem_split_l0_chunks(int nfw) { return 2 * nfw < CEM_SPLIT_RING ? CEM_SPLIT_RING : 2 * nfw; }
struct SRing {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff, n, pos;
    cem_u4 slot[CEM_SPLIT_RING][6];
    __device__ __forceinline__ void ld(cem_u4 (&s)[6], const int g) const
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) s[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * 6144 + i * 1024, 0);
    }
    __device__ __forceinline__ void init(const f4 *b, const int lane_, const int n_)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(b), 0, n_ * 6144, 0x00020000);
        voff = lane_ * 16; n = n_;
#pragma unroll
        for (int i = 0; i < CEM_SPLIT_RING - 1; ++i) ld(slot[i], i % n_);
        pos = (CEM_SPLIT_RING - 1) % n_;
    }
};

// One dense stage: acc{0,1}[c] += sum over the stage's chunks of the six split products.  OWN: chunk 0 of the visiting order is the
// wave's own (planes in registers) and the barrier that publishes the other waves' chunks comes after its MFMAs.  XMODE as in
// cem_mfma_stage (exchange: barrier inside; re-read: the input was published and waited for by an earlier stage).
template <int RC, int NCH, bool OWN, int XMODE>
__device__ __forceinline__ void cem_split_stage(f4 (&acc0)[RC], f4 (&acc1)[RC], const cem_u4 (&own)[RC][3], SRing &wq, const char *smem, const int xr,
                                                const int lane, const int w)
{
    static_assert(NCH % CEM_SPLIT_RING == 0, "stage lengths must keep the ring phase");
    constexpr int FL = OWN ? 1 : 0;                        // visiting position of the first chunk that comes from LDS
    cem_u4 bp[2][RC][3];                                   // B planes of the chunk in use and of the next one
    // the six products of one chunk for both output blocks, smallest terms first: (weight plane, activation plane)
    // (2,0) (1,1) (0,2) | (1,0) (0,1) (0,0); HALF 0 / 1 = the first / last three (the own chunk straddles the barrier)
#define CEM_SPLIT_PRODUCTS(G_, B0_, B1_, B2_, C_, HALF_) do { \
        if ((HALF_) != 1) { \
            acc0[C_] = CEM_MFMA_BF((G_)[2], B0_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[5], B0_, acc1[C_]); \
            acc0[C_] = CEM_MFMA_BF((G_)[1], B1_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[4], B1_, acc1[C_]); \
            acc0[C_] = CEM_MFMA_BF((G_)[0], B2_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[3], B2_, acc1[C_]); } \
        if ((HALF_) != 0) { \
            acc0[C_] = CEM_MFMA_BF((G_)[1], B0_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[4], B0_, acc1[C_]); \
            acc0[C_] = CEM_MFMA_BF((G_)[0], B1_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[3], B1_, acc1[C_]); \
            acc0[C_] = CEM_MFMA_BF((G_)[0], B0_, acc0[C_]); acc1[C_] = CEM_MFMA_BF((G_)[3], B0_, acc1[C_]); } } while (0)
#define CEM_SPLIT_READ(Q_) do { const int F_ = OWN ? cem_split_perm(w, (Q_)) : (Q_); \
        _Pragma("unroll") for (int c = 0; c < RC; ++c) \
            _Pragma("unroll") for (int j = 0; j < 3; ++j) \
                bp[(Q_) & 1][c][j] = *reinterpret_cast<const cem_u4 *>(smem + xr + cem_split_off(c, F_, j, lane)); } while (0)
#pragma unroll
    for (int P = 0; P < NCH; ++P) {
        wq.ld(wq.slot[(P + CEM_SPLIT_RING - 1) % CEM_SPLIT_RING], wq.pos);     // the group RING - 1 chunks ahead (possibly the next stage's)
        wq.pos = (wq.pos + 1 == wq.n) ? 0 : wq.pos + 1;
        __builtin_amdgcn_sched_barrier(0);
        const cem_u4 (&g)[6] = wq.slot[P % CEM_SPLIT_RING];
        if (OWN && P == 0) {
            // the wave's own chunk: half of its MFMAs cover the wait at the barrier, the other half the LDS round trip of chunk 1
#pragma unroll
            for (int c = 0; c < RC; ++c) CEM_SPLIT_PRODUCTS(g, own[c][0], own[c][1], own[c][2], c, 0);
            if (XMODE == CEM_X_EXCHANGE) __syncthreads();
            if (NCH > 1) CEM_SPLIT_READ(1);
#pragma unroll
            for (int c = 0; c < RC; ++c) CEM_SPLIT_PRODUCTS(g, own[c][0], own[c][1], own[c][2], c, 1);
            continue;
        }
        if (!OWN && P == 0) {
            if (XMODE == CEM_X_EXCHANGE) __syncthreads();
            CEM_SPLIT_READ(0);
        }
        if (P + 1 < NCH && P + 1 > FL) CEM_SPLIT_READ(P + 1);       // a chunk ahead of its MFMAs
#pragma unroll
        for (int c = 0; c < RC; ++c) CEM_SPLIT_PRODUCTS(g, bp[P & 1][c][0], bp[P & 1][c][1], bp[P & 1][c][2], c, 2);
    }
#undef CEM_SPLIT_PRODUCTS
#undef CEM_SPLIT_READ
}

// One tile for the whole horizon (cem_rollout_tile with the split stages; MODE as there)
template <int RC, int NFW, int MODE>
__device__ __forceinline__ void cem_rollout_tile_split(const RolloutParams &p, char *smem, const int tile_idx)
{
    const int tid = (int)((threadIdx.x + 64u * (unsigned)((tile_idx + (tile_idx >> 8)) & 3)) & 255u);   // wave roles rotate with the tile
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const TileDesc td = p.tiles[tile_idx];
    const int wbk = 0;
    const int O = p.O, A = p.A, H = p.H;
    constexpr int XB = CEM_SPLIT_XB(RC);
    constexpr int NCH0 = cem_split_l0_chunks(NFW);        // K = 32 chunks of the layer-0 input (padded to the ring length)
    float *part = reinterpret_cast<float *>(smem + 2 * XB);
    int xw = 0;
    if (NCH0 > 2 * NFW) {                                 // padded layer-0 chunks meet zero weights: what LDS holds there must be finite
        for (int o = (int)threadIdx.x * 16; o < 2 * XB; o += 256 * 16) *reinterpret_cast<cem_u4 *>(smem + o) = (cem_u4){0u, 0u, 0u, 0u};
        __syncthreads();
    }
    const PhiloxKey key = cem_key(p.ctrl);
    const float rscale = p.sampling ? CEM_BM_RSCALE : 0.0f;
    const int member_u = __builtin_amdgcn_readfirstlane(td.member);
    SRing wq;
    wq.init(p.wpack + (size_t)member_u * p.member_stride_f4 + p.wave_off_f4[w], lane, (int)p.wave_groups[w]);
    const __amdgpu_buffer_rsrc_t et_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.etab + (size_t)member_u * (CEM_ET_ROWS + p.L) * CEM_U), 0, (CEM_ET_ROWS + p.L) * CEM_U * 4, 0x00020000);
    const int tab_v = 64 * w + 16 * q;
    const int bias_v = 128 * w + 16 * q;

    f4 s[NFW][RC];
    int slotc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) { const int sl = 16 * c + j; slotc[c] = sl < td.cnt ? sl : td.cnt - 1; }
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int f0 = 16 * (w + 4 * i) + 4 * q;
#pragma unroll
        for (int c = 0; c < RC; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + r;
                float v = 0.f;
                if (f < O) v = td.s0_base < 0 ? p.ctrl->state[f] : p.s0[(size_t)(td.s0_base + slotc[c]) * O + f];
                s[i][c][r] = v;
            }
    }
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(p.act_pad), 0, MODE == 0 ? p.act_pad_bytes : 0u, 0x00020000);
    int actv[NFW][RC];
    const float *actrow[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        actrow[c] = p.actions + (size_t)(td.act_base + slotc[c]) * H * A;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            int qi = 4 * (w + 4 * i) + q - p.act_q0;
            qi = qi < 0 ? 0 : (qi >= p.act_nq ? p.act_nq - 1 : qi);
            actv[i][c] = ((td.act_base + slotc[c]) * H * p.act_nq + qi) * 16;
        }
    }
#define CEM_LOAD_ACT(DST, I_, C_, TN_) do { \
        if (MODE == 0) DST = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(act_rs, actv[I_][C_], (TN_) * p.act_nq * 16, 0)); \
        else { _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
            int af = 16 * (w + 4 * (I_)) + 4 * q + r - O; af = af < 0 ? 0 : (af >= A ? A - 1 : af); \
            DST[r] = actrow[C_][(TN_) * A + af]; } } } while (0)
    // the scaled input block Fo of chunk c_rc goes to LDS as three planes of 4 bf16 per lane (its half of the lane's 16 bytes)
#define CEM_PUBLISH_X(X_, FO_, C_) do { cem_u2 px_[3]; cem_split4((X_), px_); \
        _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) \
            *reinterpret_cast<cem_u2 *>(smem + xw + cem_split_off((C_), (FO_) >> 1, j_, lane) + 8 * ((FO_) & 1)) = px_[j_]; } while (0)

    float d_prev = 0.f, c_prev = 0.f, cum = 0.f;
    bool done = false;
    const int nk = 1 + p.sc.n_cost;
    const float csz[4] = {p.sc.cost_size[0], p.sc.cost_size[1], p.sc.cost_size[2], p.sc.cost_size[3]};
    const float ind_cap = p.sc.indicator ? 1.0f : __builtin_inff(), clipv = p.sc.reward_clip > 0.f ? p.sc.reward_clip : __builtin_inff();
    const __amdgpu_buffer_rsrc_t cost_rs = __builtin_amdgcn_make_buffer_rsrc(p.costs, 0, p.costs ? (uint32_t)(H * p.Bloc) : 0u, 0x00020000);

    // ---- prologue: x_0 = scale(concat[s_0, a_0]) and the scorer terms of s_0 --------------------------------------------------
    {
        float pm[2][RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) { pm[0][c] = __builtin_inff(); pm[1][c] = __builtin_inff(); }
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int tv = tab_v + 256 * i;
            const f4 mn4 = cem_ld_tab(et_rs, tv, CEM_ET_NMIN * 512), rd4 = cem_ld_tab(et_rs, tv, CEM_ET_RDELTA * 512);
            const f4 isact4 = cem_ld_tab(et_rs, tv, CEM_ET_ACT * 512);
            const f4 sel0 = cem_ld_tab(et_rs, tv, CEM_ET_SEL0 * 512), sel1 = cem_ld_tab(et_rs, tv, CEM_ET_SEL1 * 512);
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                f4 act4; CEM_LOAD_ACT(act4, i, c, 0);
                const f4 sn = s[i][c];
                if (MODE == 1) {
                    const int slot = 16 * c + j, f0 = 16 * (w + 4 * i) + 4 * q;
                    if (p.traj && slot < td.cnt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (f0 + r < O) p.traj[((size_t)(td.row_base + slot) * (H + 1)) * O + f0 + r] = sn[r];
                    }
                }
                cem_scorer_terms(sn, p.sc.D, sel0, sel1, pm[0][c], pm[1][c]);
                const f4 x = cem_sub4(__builtin_elementwise_fma(isact4, act4, sn), mn4) * rd4;
                CEM_PUBLISH_X(x, w + 4 * i, c);
            }
        }
        CEM_RARE_KINDS_AND_STORE();
        xw = XB;
    }
    f4 nb0 = cem_ld_tab(et_rs, bias_v, CEM_ET_ROWS * 512);
    f4 nb1 = cem_ld_tab(et_rs, bias_v + 64, CEM_ET_ROWS * 512);
    cem_u4 own[RC][3];                                    // the wave's own chunk (its two output blocks of the last hidden stage), split
#pragma unroll
    for (int c = 0; c < RC; ++c)
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) own[c][jj] = (cem_u4){0u, 0u, 0u, 0u};

#ifdef CEM_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = (long long)__builtin_amdgcn_s_memtime();
    st_[7] = tprev_;
#endif
    const int prio_r0 = (tile_idx >> 8) % 3;
    f4 bm0, bv0;
    for (int t = 0; t < H; ++t) {
        {
            const int lvl = (t + prio_r0) % 3;
            if (lvl == 0) __builtin_amdgcn_s_setprio(0); else if (lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
        }
#define CEM_SPLIT_PUBLISH() do { \
            _Pragma("unroll") for (int c = 0; c < RC; ++c) { \
                f4 h0 = acc0[c], h1 = acc1[c]; \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) { h0[r] = fmaxf(h0[r], 0.f); h1[r] = fmaxf(h1[r], 0.f); } \
                cem_split8(h0, h1, own[c]); \
                _Pragma("unroll") for (int j_ = 0; j_ < 3; ++j_) *reinterpret_cast<cem_u4 *>(smem + xw + cem_split_off(c, w, j_, lane)) = own[c][j_]; \
            } \
            xw ^= XB; } while (0)
#define CEM_NEXT_BIAS(LN) do { \
            nb0 = cem_ld_tab(et_rs, bias_v, (CEM_ET_ROWS + (LN)) * 512); \
            nb1 = cem_ld_tab(et_rs, bias_v + 64, (CEM_ET_ROWS + (LN)) * 512); } while (0)
        {
            f4 acc0[RC], acc1[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) { acc0[c] = nb0; acc1[c] = nb1; }
            CEM_NEXT_BIAS(p.L > 1 ? 1 : 0);
            // layer 0: every chunk of the scaled input comes from LDS (a wave's input blocks w, w + 4 are halves of two chunks)
            cem_split_stage<RC, NCH0, false, CEM_X_EXCHANGE>(acc0, acc1, own, wq, smem, xw ^ XB, lane, w);
            // the head biases of the wave's first observation block start the heads' accumulators: requested here, a step's worth of
            // hidden stages ahead (requested next to their use they cost the heads stage an L2 round trip: 2.26 K vs 1.8 K cycles)
            bm0 = cem_ld_tab(et_rs, tab_v, CEM_ET_BMU * 512); bv0 = cem_ld_tab(et_rs, tab_v, CEM_ET_BVAR * 512);
            CEM_STAMP(0);
            CEM_BOOKKEEP(t - 1);
            CEM_STAMP(6);
            CEM_SPLIT_PUBLISH();
            CEM_STAMP(5);
        }
        for (int l = 1; l < p.L; ++l) {
            f4 acc0[RC], acc1[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) { acc0[c] = nb0; acc1[c] = nb1; }
            CEM_NEXT_BIAS(l + 1 < p.L ? l + 1 : 0);
            cem_split_stage<RC, CEM_SPLIT_CHUNKS, true, CEM_X_EXCHANGE>(acc0, acc1, own, wq, smem, xw ^ XB, lane, w);
            CEM_STAMP(1);
            CEM_SPLIT_PUBLISH();
            CEM_STAMP(5);
        }
#undef CEM_SPLIT_PUBLISH
#undef CEM_NEXT_BIAS

        // ---- heads, state update, scorer terms, next scaled input: cem_rollout_tile's epilogue ------------------------------------
        float pm[2][RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) { pm[0][c] = __builtin_inff(); pm[1][c] = __builtin_inff(); }
        const int tn = (t + 1 < H) ? t + 1 : H - 1;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int Fo = w + 4 * i;
            const int tv = tab_v + 256 * i;
            const f4 mn4 = cem_ld_tab(et_rs, tv, CEM_ET_NMIN * 512), rd4 = cem_ld_tab(et_rs, tv, CEM_ET_RDELTA * 512);
            const f4 bm = i == 0 ? bm0 : cem_ld_tab(et_rs, tv, CEM_ET_BMU * 512), bv = i == 0 ? bv0 : cem_ld_tab(et_rs, tv, CEM_ET_BVAR * 512);
            const f4 om4 = cem_ld_tab(et_rs, tv, CEM_ET_OBS * 512), isact4 = cem_ld_tab(et_rs, tv, CEM_ET_ACT * 512);
            const f4 sel0 = cem_ld_tab(et_rs, tv, CEM_ET_SEL0 * 512), sel1 = cem_ld_tab(et_rs, tv, CEM_ET_SEL1 * 512);
            f4 accm[RC], accv[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) { accm[c] = bm; accv[c] = bv; }
            CEM_STAMP(2);
            if (Fo < p.KB_obs) {                                                           // wave-uniform
                if (i == 0) cem_split_stage<RC, CEM_SPLIT_CHUNKS, true, CEM_X_EXCHANGE>(accm, accv, own, wq, smem, xw ^ XB, lane, w);
                else cem_split_stage<RC, CEM_SPLIT_CHUNKS, true, CEM_X_REREAD>(accm, accv, own, wq, smem, xw ^ XB, lane, w);
            } else if (i == 0) {
                __syncthreads();                      // keep the barrier count of waves without observation features
            }
            CEM_STAMP(3);
            // the step's model noise and actions: independent of the heads' MFMAs, so drawn AFTER they are issued — VALU work behind
            // queued bf16 MFMAs runs in their shadow (scripts/mfma_microbench5.hip -DMB_BF16: 288 MFMAs + 330 VALU take 1.08 x the MFMAs alone)
            f4 act4[RC], eps4[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                CEM_LOAD_ACT(act4[c], i, c, tn);
                if (MODE == 1 && p.eps_model) {
                    const int f0 = 16 * Fo + 4 * q;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fc = (f0 + r < O) ? f0 + r : O - 1;
                        eps4[c][r] = p.eps_model[((size_t)t * p.Btot + td.noise_row_base + slotc[c]) * O + fc];
                    }
                    eps4[c] = eps4[c] * (p.sampling ? 1.0f : 0.0f);
                } else {
                    eps4[c] = cem_normal4((uint32_t)(td.noise_row_base + slotc[c]), (uint32_t)t, (uint32_t)p.it,
                                          (uint32_t)(4 * Fo + q), CEM_STREAM_MODEL, key, rscale);
                }
            }
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const f4 mu = accm[c];
                const f4 var = cem_softplus4(accv[c]) + 1e-4f;
                f4 sd;
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = __builtin_amdgcn_sqrtf(var[r]);
                const f4 d = mu + sd * eps4[c];
                const f4 sn = s[i][c] + d * om4;
                if (MODE == 1) {
                    const int slot = 16 * c + j, f0 = 16 * Fo + 4 * q;
                    if (slot < td.cnt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (f0 + r < O) {
                                const size_t o = ((size_t)(td.row_base + slot) * H + t) * O + f0 + r;
                                if (p.mu_out) p.mu_out[o] = mu[r];
                                if (p.sd_out) p.sd_out[o] = sd[r];
                                if (p.traj) p.traj[((size_t)(td.row_base + slot) * (H + 1) + (t + 1)) * O + f0 + r] = sn[r];
                            }
                    }
                }
                s[i][c] = sn;
                cem_scorer_terms(sn, p.sc.D, sel0, sel1, pm[0][c], pm[1][c]);
                const f4 x = cem_sub4(__builtin_elementwise_fma(isact4, act4[c], sn), mn4) * rd4;
                CEM_PUBLISH_X(x, Fo, c);
            }
        }
        CEM_RARE_KINDS_AND_STORE();
        xw ^= XB;
        CEM_STAMP(4);
    }
    __syncthreads();
    CEM_BOOKKEEP(H - 1);
    if (w == wbk && lane < td.cnt) p.ret[td.row_base + lane] = cum;
#ifdef CEM_STAMPS
    if (p.stamps && lane == 0) for (int i = 0; i < 8; ++i) p.stamps[((size_t)tile_idx * 4 + w) * 8 + i] = st_[i];
#endif
#undef CEM_LOAD_ACT
#undef CEM_PUBLISH_X
}

template <int RC, int NFW, int MODE>
__global__ __launch_bounds__(256) void cem_rollout_split_kernel(const RolloutParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.check_done && p.ctrl->done) return;
    cem_tile_sample_actions(p, (int)blockIdx.x, 0, p.H, true, MODE == 1);
    cem_rollout_tile_split<RC, NFW, MODE>(p, smem, (int)blockIdx.x);
}
