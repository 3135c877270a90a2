// cem_rollout_wide.h — the GENERIC rollout kernel: hidden layers wider than the fast kernel's 128 units (units <= 256; obs+act <= 128)
// and hidden activations other than relu at any width.
//
// The reference takes any `units` and any activation from config/models.yaml:11-12 (128 / relu ship).  cem_rollout_tile keeps a
// layer's 8 feature blocks in registers and streams pre-packed weights through a ring sized for exactly that; this kernel trades
// some of that speed for generality: the same tile (16 rows of one member for the whole horizon, 4 waves), the same arithmetic per
// element, the same Philox keys, the same epilogue / scorer terms / bookkeeping (cem_rollout_tile's own macros on the same
// per-member feature table) — but runtime loops over 16-feature blocks, activations exchanged through LDS at every stage, and
// weights streamed from a per-member image packed in A-operand order: one 1 KB group per (k block, output block), so a lane's four
// MFMA steps of a group are ONE 16-byte load.  A layer's products are summed over k blocks in ascending order (the fast kernel
// visits a wave's own blocks first): the two kernels agree to fp32 rounding, not bit for bit — which kernel runs depends on
// (units, activation) alone, so shard / tile-plan invariance holds within either.
//
// What makes it run (round 3; profiles/r03_ab_generic_kernel.txt: 10.3 -> 7.1 ms per B2-shaped plan at 256 units, 0.49 -> 0.67 of the
// fp32 MFMA peak, 0.82 with four tiles per CU; 160 units 7.1 -> 3.8 ms):
//  * every load of the steady state is UNCONDITIONAL.  s_waitcnt counts loads in issue order, so a prefetch that sits under a
//    wave-uniform branch (`if (kb + 1 < nbK) load next`) makes the compiler assume it may not have been issued and wait with
//    vmcnt(0) before the MFMAs — i.e. for the prefetch itself, every k block.  Indices are clamped into the image instead; a load
//    past the end of a stage re-reads its last group and is never used.
//  * the first weight groups AND the biases of the next stage are requested before the barrier that publishes its input, so an
//    L2 round trip overlaps the barrier wait and the activation epilogue instead of following them.
//  * the epilogue addresses one buffer resource per table with a single lane-offset VGPR (no 64-bit address pairs held across
//    the loop): 213 -> 126 VGPRs, four workgroups per CU instead of two.
//  * MODE 0 (planning) and MODE 1 (caller-supplied tensors, trajectory / head-moment outputs) are separate instantiations.
#pragma once
#include "cem_device.h"

#define CEM_WIDE_U 256                       // widest hidden layer
#define CEM_WIDE_NB (CEM_WIDE_U / 16)        // 16 feature blocks
#define CEM_WIDE_OB 4                        // hidden output blocks per wave (w, w + 4, w + 8, w + 12)
#define CEM_WIDE_SMEM (2 * CEM_WIDE_NB * 1024 + CEM_PART_FLOATS * 4)

struct WideParams {
    RolloutParams r;                         // tiles, tables, actions, noise, outputs, scorer: as for the fast kernel (wpack / bias_* unused;
                                             // etab has TWO rows per hidden layer's biases: [E][CEM_ET_ROWS + 2 L][128])
    const f4 *wimg;                          // [E][img_f4] packed weight images: groups [64 lanes][f4], cem_wide_group_* order
    uint32_t img_f4;                         // f4 per member image
    int32_t U;
    int32_t act;                             // enum cem_activation of the hidden layers
};

// Packed image of one member (host: pack_member_wide; device: here).  Group of (layer l, k block kb, output block ob):
//   hidden layers   g = base(l) + kb * nbU + ob,  base(0) = 0, base(l) = nbIn * nbU + (l - 1) * nbU * nbU
//   mean head       g = baseH + kb * nbO + ob,    baseH = nbIn * nbU + (L - 1) * nbU * nbU
//   variance head   g = baseH + nbU * nbO + kb * nbO + ob
// and inside a group lane (q, j) holds W[16 kb + 4 q + r][16 ob + j] for r = 0..3 (zero past the matrix).
__host__ __device__ inline int cem_wide_base(int l, int nbIn, int nbU) { return l == 0 ? 0 : nbIn * nbU + (l - 1) * nbU * nbU; }
__host__ __device__ inline int cem_wide_groups(int L, int nbIn, int nbU, int nbO) { return nbIn * nbU + (L - 1) * nbU * nbU + 2 * nbU * nbO; }

__device__ __forceinline__ f4 cem_wide_a(const __amdgpu_buffer_rsrc_t rsrc, const int lane16, const int g)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane16, g * 1024, 0));
}

// The k loop of one stage for a wave with NOWN operand groups per k block (a hidden layer: output blocks w, w + 4, ...; the heads:
// mean and variance of one observation block); group (kb, i) = g0 + kb * gsk + i * gsi.  The weights of k blocks kb + 1 ..
// kb + RING - 1 are in flight during the MFMAs of k block kb, and so is the next activation block.  On entry ring slots
// 0 .. RING - 2 hold k blocks 0 .. RING - 2 (cem_wide_prime, issued BEFORE the barrier that publishes this stage's input: an L2
// hit takes about as long as two k blocks' MFMAs, and with two workgroups on a CU nothing else covers it); on exit the ring is free.
#ifndef CEM_WIDE_RING
#define CEM_WIDE_RING 2                      // measured 2 / 3 / 4: 7.06 / 7.24 / 8.70 ms per B2-shaped plan at 256 units (126 / 159 / 190 VGPRs: 4 / 3 / 2 workgroups per CU)
#endif
typedef f4 WideRing[CEM_WIDE_RING][CEM_WIDE_OB];

// Every load here is unconditional (indices clamped into the image instead of branched around): s_waitcnt counts loads in issue
// order, and a load that is only sometimes issued between a request and its use makes the compiler wait for everything.
__device__ __forceinline__ void cem_wide_prime(WideRing &ring, const __amdgpu_buffer_rsrc_t img, const int lane16, const int g0, const int gsk,
                                               const int gsi, const int glast)
{
#pragma unroll
    for (int s = 0; s < CEM_WIDE_RING - 1; ++s)
#pragma unroll
        for (int i = 0; i < CEM_WIDE_OB; ++i) ring[s][i] = cem_wide_a(img, lane16, min(g0 + s * gsk + i * gsi, glast));
}

template <int NOWN>
__device__ __forceinline__ void cem_wide_kloop(f4 (&acc)[CEM_WIDE_OB], WideRing &ring, const __amdgpu_buffer_rsrc_t img, const int lane16, const int g0,
                                               const int gsk, const int gsi, const int nbK, const char *xin, const int lane)
{
    constexpr int RD = CEM_WIDE_RING;
    f4 hb = *reinterpret_cast<const f4 *>(xin + lane * 16);
    for (int kb = 0; kb < nbK; kb += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int kn = min(kb + u + RD - 1, nbK - 1);                             // past the stage: its last k block again, never used
#pragma unroll
            for (int i = 0; i < NOWN; ++i) ring[(u + RD - 1) % RD][i] = cem_wide_a(img, lane16, g0 + kn * gsk + i * gsi);
            if (kb + u < nbK) {                                                       // wave-uniform
                const f4 hc = hb;
                hb = *reinterpret_cast<const f4 *>(xin + (min(kb + u + 1, nbK - 1) * 64 + lane) * 16);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int i = 0; i < NOWN; ++i) acc[i] = CEM_MFMA(ring[u][i][r], hc[r], acc[i]);
            }
        }
    }
}

// MODE 0: planning (actions from the padded quad layout the tile's own prologue writes, cem_tile_sample_actions; Philox noise).  MODE 1: caller-supplied action /
// noise tensors and the trajectory / head-moment outputs of cem_unfold_sequences.  The epilogue, the scorer terms and the
// bookkeeping are cem_rollout_tile's (its macros, with RC = 1), on the same per-member feature table; the hidden layers' biases
// are two table rows per layer here (256 features).
template <int MODE>
__global__ __launch_bounds__(256, 2) void cem_rollout_wide_kernel(const WideParams wp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RolloutParams &p = wp.r;
    if (p.check_done && p.ctrl->done) return;
    cem_tile_sample_actions(p, (int)blockIdx.x, 0, p.H, true, MODE == 1);
    constexpr int RC = 1, NFW = 2;                         // 16-row tiles; obs + act <= 128: a wave owns input blocks w and w + 4
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const int wbk = 0;
    const TileDesc td = p.tiles[blockIdx.x];
    const int O = p.O, A = p.A, H = p.H, U = wp.U, L = p.L;
    const int nbU = (U + 15) >> 4, nbIn = p.KB_in, nbO = p.KB_obs;
    constexpr int XB = CEM_WIDE_NB * 1024;
    float *part = reinterpret_cast<float *>(smem + 2 * XB);
    int xw = 0;                                            // LDS buffer the current stage's outputs go to
    const PhiloxKey key = cem_key(p.ctrl);
    const float rscale = p.sampling ? CEM_BM_RSCALE : 0.0f;
    const int member_u = __builtin_amdgcn_readfirstlane(td.member);
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(wp.wimg + (size_t)member_u * wp.img_f4), 0, wp.img_f4 * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t et_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.etab + (size_t)member_u * (CEM_ET_ROWS + 2 * L) * CEM_U), 0, (CEM_ET_ROWS + 2 * L) * CEM_U * 4, 0x00020000);
    const int tab_v = 64 * w + 16 * q;                     // + 256 i: this lane's feature quad of input block w + 4 i
    const int lane16 = lane * 16;
    const int baseH = cem_wide_base(L, nbIn, nbU);         // = nbIn * nbU + (L - 1) * nbU * nbU
    const int glast = cem_wide_groups(L, nbIn, nbU, nbO) - 1;
    const int nown = max(0, (nbU - w + 3) >> 2);           // hidden output blocks this wave owns: w, w + 4, ...
    WideRing ring;                                         // weight groups in flight (cem_wide_kloop)
    f4 accn[CEM_WIDE_OB];                                  // the next hidden layer's biases, requested one stage ahead like its first weights
    // hidden-layer biases: rows CEM_ET_ROWS + 2 l, + 1 of the table (256 features, zero padded); output block w + 4 i at byte 64 (w + 4 i) + 16 q
#define CEM_WIDE_NEXT_BIAS(LN) do { _Pragma("unroll") for (int i_ = 0; i_ < CEM_WIDE_OB; ++i_) \
        accn[i_] = cem_ld_tab(et_rs, tab_v + 256 * i_, (CEM_ET_ROWS + 2 * (LN)) * 512); } while (0)

    // ---- state registers ----------------------------------------------------------------------------------------------
    const int slot0 = j < td.cnt ? j : td.cnt - 1;
    f4 s[NFW][RC];
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int f0 = 16 * (w + 4 * i) + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + r;
            float v = 0.f;
            if (f < O) v = td.s0_base < 0 ? p.ctrl->state[f] : p.s0[(size_t)(td.s0_base + slot0) * O + f];
            s[i][0][r] = v;
        }
    }
    // this lane's actions: as in cem_rollout_tile
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(p.act_pad), 0, MODE == 0 ? p.act_pad_bytes : 0u, 0x00020000);
    int actv[NFW][RC];
    const float *actrow[RC] = {p.actions + (size_t)(td.act_base + slot0) * H * A};
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        int qi = 4 * (w + 4 * i) + q - p.act_q0;
        qi = qi < 0 ? 0 : (qi >= p.act_nq ? p.act_nq - 1 : qi);
        actv[i][0] = ((td.act_base + slot0) * H * p.act_nq + qi) * 16;
    }

    float d_prev = 0.f, c_prev = 0.f, cum = 0.f;
    bool done = false;
    const int nk = 1 + p.sc.n_cost;
    const float csz[4] = {p.sc.cost_size[0], p.sc.cost_size[1], p.sc.cost_size[2], p.sc.cost_size[3]};
    const float ind_cap = p.sc.indicator ? 1.0f : __builtin_inff(), clipv = p.sc.reward_clip > 0.f ? p.sc.reward_clip : __builtin_inff();
    const __amdgpu_buffer_rsrc_t cost_rs = __builtin_amdgcn_make_buffer_rsrc(p.costs, 0, p.costs ? (uint32_t)(H * p.Bloc) : 0u, 0x00020000);

    // t = -1: the scaled input of step 0 and the scorer terms of s_0 (no network evaluation: the update is masked off)
    for (int t = -1; t < H; ++t) {
        {                                                  // issue priority rotates step by step, offset by tile (cem_rollout_tile)
            const int lvl = (t + 1 + (int)(blockIdx.x >> 8) % 3) % 3;
            if (lvl == 0) __builtin_amdgcn_s_setprio(0); else if (lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
        }
        if (t >= 0) {
            __syncthreads();                               // the previous step's next-input blocks and scorer terms are in LDS
            CEM_BOOKKEEP(t - 1);
            // ---- dense layers: h = act(h W + b)  (mlp_ensemble.py:18-22): wave w computes output blocks w, w + 4, w + 8, w + 12 ----
            for (int l = 0; l < L; ++l) {
                const int nbK = l == 0 ? nbIn : nbU;
                const int gl = cem_wide_base(l, nbIn, nbU);
                f4 acc[CEM_WIDE_OB];
#pragma unroll
                for (int i = 0; i < CEM_WIDE_OB; ++i) acc[i] = accn[i];
                const char *xin = smem + (xw ^ XB);
                switch (nown) {                                                       // wave-uniform
                case 0: break;
                case 1: cem_wide_kloop<1>(acc, ring, img, lane16, gl + w, nbU, 4, nbK, xin, lane); break;
                case 2: cem_wide_kloop<2>(acc, ring, img, lane16, gl + w, nbU, 4, nbK, xin, lane); break;
                case 3: cem_wide_kloop<3>(acc, ring, img, lane16, gl + w, nbU, 4, nbK, xin, lane); break;
                default: cem_wide_kloop<4>(acc, ring, img, lane16, gl + w, nbU, 4, nbK, xin, lane); break;
                }
                // the next stage's first weights (and biases): in flight across the barrier
                if (l + 1 < L) { cem_wide_prime(ring, img, lane16, cem_wide_base(l + 1, nbIn, nbU) + w, nbU, 4, glast); CEM_WIDE_NEXT_BIAS(l + 1); }
                else cem_wide_prime(ring, img, lane16, baseH + w, nbO, nbU * nbO, glast);   // a wave without an observation block: loaded, not used
#pragma unroll
                for (int i = 0; i < CEM_WIDE_OB; ++i) {
                    const int ob = w + 4 * i;
                    if (ob < nbU) {                                                   // wave-uniform
                        f4 h = acc[i];                                                // (output features past `units`: zero weights and bias, act(0) never read as nonzero input — their next-layer weights are zero)
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = wp.act == 0 ? fmaxf(h[r], 0.f) : cem_activation_fwd(wp.act, h[r]);
                        *reinterpret_cast<f4 *>(smem + xw + (ob * 64 + lane) * 16) = h;
                    }
                }
                xw ^= XB;
                __syncthreads();
            }
        }
        // ---- heads (mlp_ensemble.py:33-34,189-193), state update (transition_model.py:75), scorer terms (safety_gym.py:188-192)
        //      and the next scaled input (transition_model.py:70-72,79-87): cem_rollout_tile's epilogue, block by block ----
        float pm[2][RC] = {{__builtin_inff()}, {__builtin_inff()}};
        const int tn = (t + 1 < H) ? t + 1 : H - 1;
        const float live = (t >= 0) ? 1.0f : 0.0f;
        const char *hL = smem + (xw ^ XB);                 // the last hidden layer's output
        bool primed0 = false;                              // layer 0 of the next step requested
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int Fo = w + 4 * i;
            if (Fo < nbIn) {                               // wave-uniform
                const int tv = tab_v + 256 * i;
                const f4 bm = cem_ld_tab(et_rs, tv, CEM_ET_BMU * 512), bv = cem_ld_tab(et_rs, tv, CEM_ET_BVAR * 512);
                const f4 mn4 = cem_ld_tab(et_rs, tv, CEM_ET_NMIN * 512), rd4 = cem_ld_tab(et_rs, tv, CEM_ET_RDELTA * 512);
                const f4 om4 = cem_ld_tab(et_rs, tv, CEM_ET_OBS * 512) * live, isact4 = cem_ld_tab(et_rs, tv, CEM_ET_ACT * 512);
                const f4 sel0 = cem_ld_tab(et_rs, tv, CEM_ET_SEL0 * 512), sel1 = cem_ld_tab(et_rs, tv, CEM_ET_SEL1 * 512);
                f4 act4, eps4;
                if (MODE == 0) act4 = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(act_rs, actv[i][0], tn * p.act_nq * 16, 0));
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        int af = 16 * Fo + 4 * q + r - O; af = af < 0 ? 0 : (af >= A ? A - 1 : af);
                        act4[r] = actrow[0][tn * A + af];
                    }
                }
                if (MODE == 1 && p.eps_model) {
                    const int f0 = 16 * Fo + 4 * q, tc = t < 0 ? 0 : t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int fc = (f0 + r < O) ? f0 + r : O - 1;
                        eps4[r] = p.eps_model[((size_t)tc * p.Btot + td.noise_row_base + slot0) * O + fc];
                    }
                    eps4 = eps4 * (p.sampling ? 1.0f : 0.0f);
                } else {
                    eps4 = cem_normal4((uint32_t)(td.noise_row_base + slot0), (uint32_t)t, (uint32_t)p.it, (uint32_t)(4 * Fo + q), CEM_STREAM_MODEL, key, rscale);
                }
                f4 ah[CEM_WIDE_OB] = {bm, bv, bm, bm};
                if (t >= 0 && Fo < nbO) {                  // wave-uniform: mean and variance heads of observation block Fo
                    if (i == 1) cem_wide_prime(ring, img, lane16, baseH + Fo, nbO, nbU * nbO, glast);
                    cem_wide_kloop<2>(ah, ring, img, lane16, baseH + Fo, nbO, nbU * nbO, nbU, hL, lane);
                }
                if (!primed0 && (i == 1 || !(t >= 0 && w + 4 < nbO))) {               // this wave's last k loop of the step is behind it
                    cem_wide_prime(ring, img, lane16, w, nbU, 4, glast);
                    primed0 = true;
                }
                const f4 mu = ah[0];
                const f4 var = cem_softplus4(ah[1]) + 1e-4f;
                f4 sd;
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = __builtin_amdgcn_sqrtf(var[r]);
                const f4 d = mu + sd * eps4;                                          // Normal.sample = loc + scale * eps
                const f4 sn = s[i][0] + d * om4;                                      // s_t += d_s_t on observation features
                if (MODE == 1) {
                    const int f0 = 16 * Fo + 4 * q;
                    if (j < td.cnt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (f0 + r < O) {
                                const size_t o = ((size_t)(td.row_base + j) * H + t) * O + f0 + r;
                                if (t >= 0 && p.mu_out) p.mu_out[o] = mu[r];
                                if (t >= 0 && p.sd_out) p.sd_out[o] = sd[r];
                                if (p.traj) p.traj[((size_t)(td.row_base + j) * (H + 1) + (t + 1)) * O + f0 + r] = sn[r];
                            }
                    }
                }
                s[i][0] = sn;
                cem_scorer_terms(sn, p.sc.D, sel0, sel1, pm[0][0], pm[1][0]);
                const f4 x = cem_sub4(__builtin_elementwise_fma(isact4, act4, sn), mn4) * rd4;   // s is 0 off the observation features
                *reinterpret_cast<f4 *>(smem + xw + (Fo * 64 + lane) * 16) = x;
            }
        }
        if (!primed0) cem_wide_prime(ring, img, lane16, w, nbU, 4, glast);             // a wave without an input block
        CEM_WIDE_NEXT_BIAS(0);
        CEM_RARE_KINDS_AND_STORE();
        xw ^= XB;
    }
    __syncthreads();
    CEM_BOOKKEEP(H - 1);
    if (w == wbk && lane < td.cnt) p.ret[td.row_base + lane] = cum;
}
#undef CEM_WIDE_NEXT_BIAS
#undef CEM_BOOKKEEP
#undef CEM_PART_MIN4
#undef CEM_PAIR_MIN_STORE
#undef CEM_RARE_KINDS_AND_STORE
