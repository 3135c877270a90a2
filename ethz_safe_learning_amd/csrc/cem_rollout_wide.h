// cem_rollout_wide.h — the rollout for hidden layers wider than the fast kernel's 128 units (128 < units <= 256; obs+act <= 128).
//
// The reference takes any `units` from config/models.yaml:11 (only 128 ships).  cem_rollout_tile keeps a layer's 8 feature blocks
// in registers and streams pre-packed weights through a ring sized for exactly that; this kernel trades that speed for width:
// the same tile (16 rows of one member for the whole horizon, 4 waves), the same arithmetic per element, the same Philox keys,
// the same epilogue and scorer terms — but runtime loops over 16-feature blocks, activations exchanged through LDS at every
// stage, and weights streamed from a per-member image packed in A-operand order — one 1 KB group per (k block, output block),
// so a lane's four MFMA steps of a group are ONE 16-byte load (read from the natural layout they were four 4-byte loads of 64-byte
// rows: 63 -> 77 TFLOP/s at 256 units; the biases still come from the natural blob, which travels with the image).  A layer's products are summed over k blocks in ascending order (the fast kernel visits a wave's
// own blocks first): the two kernels agree to fp32 rounding, not bit for bit — which kernel runs depends on `units` alone, so
// shard / tile-plan invariance holds within either.  One instantiation serves planning, explicit noise tensors and the
// trajectory / head-moment outputs of cem_unfold_sequences (null pointers switch them off).
#pragma once
#include "cem_device.h"

#define CEM_WIDE_U 256                       // widest hidden layer
#define CEM_WIDE_NB (CEM_WIDE_U / 16)        // 16 feature blocks
#define CEM_WIDE_OB 4                        // hidden output blocks per wave (w, w + 4, w + 8, w + 12)
#define CEM_WIDE_SMEM (2 * CEM_WIDE_NB * 1024 + CEM_PART_FLOATS * 4)

struct WideParams {
    RolloutParams r;                         // tiles, tables, actions, noise, outputs, scorer: as for the fast kernel (wpack / bias_* unused)
    const float *wnat;                       // [E][nat] natural weight blobs (the biases are read from here)
    const f4 *wimg;                          // [E][img_f4] packed weight images: groups [64 lanes][f4], cem_wide_group_* order
    uint32_t nat;                            // floats per member
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

// The k loop of one layer for a wave with NOWN output blocks (w, w + 4, ...): weights of k block kb + 1 are requested before the
// MFMAs of k block kb.  NOWN is a template parameter so that a layer of, say, 10 blocks costs its waves 3 / 3 / 2 / 2 blocks, not 4 each.
template <int NOWN>
__device__ __forceinline__ void cem_wide_layer(f4 (&acc)[CEM_WIDE_OB], const __amdgpu_buffer_rsrc_t img, const int lane16, const int gl, const int nbK,
                                               const int nbOut, const int w, const char *xin, const int lane)
{
    f4 a_nxt[NOWN];
#pragma unroll
    for (int i = 0; i < NOWN; ++i) a_nxt[i] = cem_wide_a(img, lane16, gl + w + 4 * i);
    for (int kb = 0; kb < nbK; ++kb) {
        f4 a_cur[NOWN];
#pragma unroll
        for (int i = 0; i < NOWN; ++i) a_cur[i] = a_nxt[i];
        if (kb + 1 < nbK) {
#pragma unroll
            for (int i = 0; i < NOWN; ++i) a_nxt[i] = cem_wide_a(img, lane16, gl + (kb + 1) * nbOut + w + 4 * i);
        }
        const f4 hb = *reinterpret_cast<const f4 *>(xin + (kb * 64 + lane) * 16);
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NOWN; ++i) acc[i] = CEM_MFMA(a_cur[i][r], hb[r], acc[i]);
    }
}

__global__ __launch_bounds__(256) void cem_rollout_wide_kernel(const WideParams wp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RolloutParams &p = wp.r;
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const TileDesc td = p.tiles[blockIdx.x];
    const int O = p.O, A = p.A, H = p.H, U = wp.U, L = p.L, D = O + A;
    const int nbU = (U + 15) >> 4, nbIn = p.KB_in, nbO = p.KB_obs;
    constexpr int XB = CEM_WIDE_NB * 1024;
    float *part = reinterpret_cast<float *>(smem + 2 * XB);
    int xw = 0;                                            // LDS buffer the current stage's outputs go to
    const PhiloxKey key = cem_key(p.ctrl);
    const float *Wm = wp.wnat + (size_t)__builtin_amdgcn_readfirstlane(td.member) * wp.nat;
    // natural-blob offsets (cem_mpc.h): W_0,b_0,...,W_mu,b_mu,W_var,b_var
    auto offW = [&](int l) { return l == 0 ? (size_t)0 : (size_t)D * U + U + (size_t)(l - 1) * ((size_t)U * U + U); };
    auto offb = [&](int l) { return offW(l) + (size_t)(l == 0 ? D : U) * U; };
    const size_t oWmu = (size_t)D * U + U + (size_t)(L - 1) * ((size_t)U * U + U), obmu = oWmu + (size_t)U * O;
    const size_t oWv = obmu + O, obv = oWv + (size_t)U * O;
    const __amdgpu_buffer_rsrc_t img = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<f4 *>(wp.wimg + (size_t)__builtin_amdgcn_readfirstlane(td.member) * wp.img_f4), 0, wp.img_f4 * 16, 0x00020000);
    const int lane16 = lane * 16;
    const int baseH = cem_wide_base(L, nbIn, nbU);         // = nbIn * nbU + (L - 1) * nbU * nbU

    // ---- state registers: wave w owns input feature blocks Fo = w + 4 i (i < 2: obs+act <= 128) ---------------------
    const int slot0 = j < td.cnt ? j : td.cnt - 1;
    f4 s[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int f0 = 16 * (w + 4 * i) + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + r;
            float v = 0.f;
            if (f < O) v = td.s0_base < 0 ? p.ctrl->state[f] : p.s0[(size_t)(td.s0_base + slot0) * O + f];
            s[i][r] = v;
        }
    }
    const float *actrow = p.actions + (size_t)(td.act_base + slot0) * H * A;

    float d_prev = 0.f, c_prev = 0.f, cum = 0.f;
    bool done = false;
    const int nk = 1 + p.sc.n_cost;
    // reward / cost / done bookkeeping of step T_ (wave 0): the same statements as cem_rollout_tile's CEM_BOOKKEEP
#define CEM_WIDE_BOOKKEEP(T_) do { if (w == 0) { \
        float dn = fminf(fminf(part[0 * 64 + lane], part[1 * 64 + lane]), fminf(part[2 * 64 + lane], part[3 * 64 + lane])); \
        float cn = 0.f; \
        for (int k = 1; k < nk; ++k) { \
            const float dk = fminf(fminf(part[(k * 4 + 0) * 64 + lane], part[(k * 4 + 1) * 64 + lane]), \
                                   fminf(part[(k * 4 + 2) * 64 + lane], part[(k * 4 + 3) * 64 + lane])); \
            cn = cn + ((dk <= p.sc.cost_size[k - 1]) ? 1.0f : 0.0f); } \
        if (p.sc.indicator) cn = cn > 0.f ? 1.0f : 0.0f; \
        if ((T_) >= 0) { \
            const bool ga = d_prev <= p.sc.goal_thresh; \
            float r = (d_prev - dn) * p.sc.reward_distance + (ga ? 1.0f : 0.0f) * p.sc.reward_goal; \
            if (p.sc.reward_clip > 0.f) r = fminf(fmaxf(r, -p.sc.reward_clip), p.sc.reward_clip); \
            if (p.variant == 1) { \
                done = done || ga; \
                const float nd = done ? 0.0f : 1.0f; \
                const float cst = c_prev * nd; \
                if (p.costs && lane < td.cnt) p.costs[(size_t)(T_) * p.Bloc + td.row_base + lane] = (uint8_t)cst; \
                cum = cum + r * nd; \
            } else { \
                const float nd = done ? 0.0f : 1.0f; \
                cum = cum + r * nd; \
                done = done || ga; \
            } } \
        d_prev = dn; c_prev = cn; } } while (0)

    for (int t = -1; t < H; ++t) {
        if (t >= 0) {
            __syncthreads();                               // the previous step's next-input blocks and scorer terms are in LDS
            CEM_WIDE_BOOKKEEP(t - 1);
            // ---- dense layers: h = relu(h W + b)  (mlp_ensemble.py:18-22): wave w computes output blocks w, w + 4, w + 8, w + 12 ----
            for (int l = 0; l < L; ++l) {
                const int nbK = l == 0 ? nbIn : nbU;
                const int gl = cem_wide_base(l, nbIn, nbU);
                const float *bl = Wm + offb(l);
                f4 acc[CEM_WIDE_OB];
#pragma unroll
                for (int i = 0; i < CEM_WIDE_OB; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int o = 16 * (w + 4 * i) + 4 * q + r; acc[i][r] = o < U ? bl[o] : 0.f; }
                const char *xin = smem + (xw ^ XB);
                switch ((nbU - w + 3) >> 2) {                                         // output blocks this wave owns (wave-uniform)
                case 1: cem_wide_layer<1>(acc, img, lane16, gl, nbK, nbU, w, xin, lane); break;
                case 2: cem_wide_layer<2>(acc, img, lane16, gl, nbK, nbU, w, xin, lane); break;
                case 3: cem_wide_layer<3>(acc, img, lane16, gl, nbK, nbU, w, xin, lane); break;
                default: cem_wide_layer<4>(acc, img, lane16, gl, nbK, nbU, w, xin, lane); break;
                }
#pragma unroll
                for (int i = 0; i < CEM_WIDE_OB; ++i) {
                    const int ob = w + 4 * i;
                    if (ob < nbU) {                                                   // wave-uniform
                        f4 h = acc[i];
#pragma unroll
                        for (int r = 0; r < 4; ++r) h[r] = (16 * ob + 4 * q + r < U) ? (wp.act == 0 ? fmaxf(h[r], 0.f) : cem_activation_fwd(wp.act, h[r])) : 0.f;
                        *reinterpret_cast<f4 *>(smem + xw + (ob * 64 + lane) * 16) = h;
                    }
                }
                xw ^= XB;
                __syncthreads();
            }
        }
        // ---- heads (mlp_ensemble.py:33-34,189-193), state update (transition_model.py:75), scorer terms (safety_gym.py:188-192)
        //      and the next scaled input (transition_model.py:70-72,79-87): cem_rollout_tile's epilogue, block by block ----
        float pm[CEM_NKIND];
#pragma unroll
        for (int k = 0; k < CEM_NKIND; ++k) pm[k] = __builtin_inff();
        const int tn = (t + 1 < H) ? t + 1 : H - 1;
        const float live = (t >= 0) ? 1.0f : 0.0f;
        const float sampling = p.sampling ? 1.0f : 0.0f;
        const float goalm = p.sc.goal_mode ? 1.0f : 0.0f;
        const char *hL = smem + (xw ^ XB);                 // the last hidden layer's output
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int Fo = w + 4 * i;
            if (Fo >= nbIn) continue;                      // wave-uniform: no such input block
            const int f0 = 16 * Fo + 4 * q;
            const f4 mn4 = *reinterpret_cast<const f4 *>(p.nmin + f0);
            const f4 rd4 = *reinterpret_cast<const f4 *>(p.nrdelta + f0);
            const f4 om4 = *reinterpret_cast<const f4 *>(p.omask + f0) * live;
            const f4 isact4 = *reinterpret_cast<const f4 *>(p.omask + CEM_U + f0);
            const f4 sel0 = *reinterpret_cast<const f4 *>(p.kind_sel + f0);
            const f4 sel1 = *reinterpret_cast<const f4 *>(p.kind_sel + CEM_U + f0);
            f4 act4, eps4, accm, accv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int af = f0 + r - O; af = af < 0 ? 0 : (af >= A ? A - 1 : af);
                act4[r] = actrow[tn * A + af];
                const int fc = (f0 + r < O) ? f0 + r : O - 1;
                accm[r] = (f0 + r < O) ? Wm[obmu + fc] : 0.f;
                accv[r] = (f0 + r < O) ? Wm[obv + fc] : 0.f;
            }
            if (p.eps_model) {
                const int tc = t < 0 ? 0 : t;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int fc = (f0 + r < O) ? f0 + r : O - 1;
                    eps4[r] = p.eps_model[((size_t)tc * p.Btot + td.noise_row_base + slot0) * O + fc];
                }
            } else {
                eps4 = cem_normal4((uint32_t)(td.noise_row_base + slot0), (uint32_t)t, (uint32_t)p.it, (uint32_t)(4 * Fo + q), CEM_STREAM_MODEL, key);
            }
            eps4 = eps4 * sampling;
            if (t >= 0 && Fo < nbO) {                      // wave-uniform: mean and variance heads of observation block Fo
                const int gm = baseH + Fo, gv = baseH + nbU * nbO + Fo;
                f4 am = cem_wide_a(img, lane16, gm), av = cem_wide_a(img, lane16, gv);
                for (int kb = 0; kb < nbU; ++kb) {
                    const f4 cm = am, cv = av;
                    if (kb + 1 < nbU) { am = cem_wide_a(img, lane16, gm + (kb + 1) * nbO); av = cem_wide_a(img, lane16, gv + (kb + 1) * nbO); }
                    const f4 hb = *reinterpret_cast<const f4 *>(hL + (kb * 64 + lane) * 16);
#pragma unroll
                    for (int r = 0; r < 4; ++r) { accm = CEM_MFMA(cm[r], hb[r], accm); accv = CEM_MFMA(cv[r], hb[r], accv); }
                }
            }
            f4 sn = s[i], x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float mu = accm[r];
                const float sd = __builtin_amdgcn_sqrtf(cem_softplus(accv[r]) + 1e-4f);
                const float d = mu + sd * eps4[r];                                   // Normal.sample = loc + scale * eps
                sn[r] = sn[r] + d * om4[r];                                          // s_t += d_s_t on observation features
                if (t >= 0 && j < td.cnt && f0 + r < O) {
                    const size_t o = ((size_t)(td.row_base + j) * H + t) * O + f0 + r;
                    if (p.mu_out) p.mu_out[o] = mu;
                    if (p.sd_out) p.sd_out[o] = sd;
                }
                const float lid = fminf(fmaxf(p.sc.D - p.sc.D * (1.0f - sn[r]), 0.f), p.sc.D);
                const float gv = goalm != 0.f ? fmaxf(sn[r], 0.f) : lid;
                pm[0] = fminf(pm[0], fmaxf(gv, sel0[r]));
                pm[1] = fminf(pm[1], fmaxf(lid, sel1[r]));
                const float xv = __builtin_fmaf(isact4[r], act4[r], sn[r]);          // s is 0 off the observation features
                x[r] = (xv - mn4[r]) * rd4[r];
            }
            s[i] = sn;
            if (p.traj && j < td.cnt) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (f0 + r < O) p.traj[((size_t)(td.row_base + j) * (H + 1) + (t + 1)) * O + f0 + r] = sn[r];
            }
            *reinterpret_cast<f4 *>(smem + xw + (Fo * 64 + lane) * 16) = x;
            for (int k = 2; k < nk; ++k) {                 // cost kinds beyond the first
                const f4 selk = *reinterpret_cast<const f4 *>(p.kind_sel + k * CEM_U + f0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float lid = fminf(fmaxf(p.sc.D - p.sc.D * (1.0f - sn[r]), 0.f), p.sc.D);
                    pm[k] = fminf(pm[k], fmaxf(lid, selk[r]));
                }
            }
        }
        // min over the 4 lane rows holding different features of the same batch row, then this wave's term per row
#pragma unroll
        for (int k = 0; k < CEM_NKIND; ++k) {
            if (k < 2 || k < nk) {
                const uint32_t mb = __float_as_uint(pm[k]);
                const auto r16 = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
                const float m16 = fminf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
                const uint32_t m16b = __float_as_uint(m16);
                const auto r32 = __builtin_amdgcn_permlane32_swap(m16b, m16b, false, false);
                part[(k * 4 + w) * 64 + j] = fminf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
            }
        }
        xw ^= XB;
    }
    __syncthreads();
    CEM_WIDE_BOOKKEEP(H - 1);
    if (w == 0 && lane < td.cnt) p.ret[td.row_base + lane] = cum;
}
#undef CEM_WIDE_BOOKKEEP
