// cem_rollout_w8.h — the rollout kernel for launches that leave at most ONE tile per CU (B1-sized populations, obs+act <= 64).
//
// A lone 4-wave tile runs one wave per SIMD: nothing overlaps its barrier / LDS / MFMA-drain bubbles (15.0 K cycles per step
// against 9.2 K of MFMA).  Here the SAME tile is computed by 8 waves: wave (s = w & 3, u = w >> 2) does what
// accumulator u of wave s does in cem_rollout_tile<1, 1, 0> — output block 2s + u of every hidden layer, the mean (u = 0) or the
// variance (u = 1) head of observation block s — from the same packed weight stream (the a / b half of each 2 KB group), in the
// same k order (cem_mfma_stage: the block itself, its pair, the rest ascending), with the same arithmetic.  Both waves of a slot
// sit on SIMD s, so one's MFMA chain covers the other's bubbles, and the epilogue is split: the upper wave draws the Philox
// noise and evaluates softplus / sqrt (sd * eps), the lower wave does the state update, the scorer terms and the next input.
// Results are bit-identical to the 4-wave kernel (tests/test_gpu_parity.py::test_eight_wave_rollout_is_bit_identical).
#pragma once
#include "cem_device.h"

#define CEM_W8_SMEM (2 * CEM_NG * 1024 + CEM_PART_FLOATS * 4 + 4 * 1024)      // two exchange buffers, scorer terms, sd * eps of 4 blocks

struct WRing8 {                    // the 4-slot prefetch ring of cem_rollout_tile, one accumulator's half (16 B per lane) of each group
    __amdgpu_buffer_rsrc_t rsrc;
    int voff, n, pos;
    f4 slot[4];
    __device__ __forceinline__ f4 ld(int g) const { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * 2048, 0)); }
    __device__ __forceinline__ void init(const f4 *b, int lane_, int u, int n_)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(b), 0, n_ * 2048, 0x00020000);
        voff = lane_ * 16 + u * 1024; n = n_;
        slot[0] = ld(0); slot[1] = ld(1 % n_); slot[2] = ld(2 % n_); slot[3] = slot[2];
        pos = 3 % n_;
    }
};

// One dense stage of one wave: acc += W^T-groups . h over KF input blocks in the canonical order of its output block.
// L0IN: layer 0 (input blocks s, then the others ascending; only the lower wave holds block s in registers).  Otherwise a hidden /
// heads stage: own block 2s + u (in registers), its pair 2s + (u ^ 1), then the rest ascending.  Exactly one barrier per call:
// after the own block's MFMAs are issued where there is one, before anything else where there is not.
template <int KF, bool L0IN>
__device__ __forceinline__ void cem_w8_stage(f4 &acc, const f4 own, WRing8 &wq, const char *smem, const int xr, const int lane, const int s, const int u)
{
    static_assert(KF % 4 == 0, "stage lengths must keep the ring phase");
    const bool have_own = L0IN ? (u == 0) : true;                       // wave-uniform
    f4 hb[KF];                                                          // every input block of the stage (KF x 4 registers)
    auto block_of = [&](int P) { return L0IN ? cem_perm_l0(s, KF / 4, P) : cem_perm_hidden(s, P < 2 ? (P ^ u) : P); };
    auto rd = [&](int P) { return *reinterpret_cast<const f4 *>(smem + xr + (block_of(P) * 64 + lane) * 16); };
    hb[0] = own;
#pragma unroll
    for (int P = 0; P < KF; ++P) {
        wq.slot[(P + 3) & 3] = wq.ld(wq.pos);
        wq.pos = (wq.pos + 1 == wq.n) ? 0 : wq.pos + 1;
        __builtin_amdgcn_sched_barrier(0);
        // all the other waves' blocks are requested at once behind the barrier: only the first read's latency is exposed
        if (P == 0 && !have_own) {
            __syncthreads();
#pragma unroll
            for (int Q = 0; Q < KF; ++Q) hb[Q] = rd(Q);
        }
        if (P == 1 && have_own) {
            __syncthreads();
#pragma unroll
            for (int Q = 1; Q < KF; ++Q) hb[Q] = rd(Q);
        }
        const f4 g = wq.slot[P & 3];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = CEM_MFMA(g[r], hb[P][r], acc);
    }
}

__global__ __launch_bounds__(512) void cem_rollout_w8_kernel(const RolloutParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int s = w & 3, u = w >> 2;
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const TileDesc td = p.tiles[blockIdx.x];
    const int O = p.O, A = p.A, H = p.H;
    constexpr int XB = CEM_NG * 1024;
    float *part = reinterpret_cast<float *>(smem + 2 * XB);
    f4 *sde = reinterpret_cast<f4 *>(smem + 2 * XB + CEM_PART_FLOATS * 4);      // [4 blocks][64 lanes] sd * eps of the step
    int xw = 0;
    const PhiloxKey key = cem_key(p.ctrl);

    WRing8 wq;
    {
        const int member_u = __builtin_amdgcn_readfirstlane(td.member);
        wq.init(p.wpack + (size_t)member_u * p.member_stride_f4 + p.wave_off_f4[s], lane, u, (int)p.wave_groups[s]);
    }
    const float *bias_h = p.bias_h + (size_t)td.member * p.L * CEM_U;
    const float *bias_mu = p.bias_mu + (size_t)td.member * CEM_U;
    const float *bias_var = p.bias_var + (size_t)td.member * CEM_U;
    const int ob = 16 * (2 * s + u) + 4 * q;                  // this lane's four features of the wave's hidden output block

    // state of input block s (lower wave only): features 16 s + 4 q + r of row j
    const int slot0 = j < td.cnt ? j : td.cnt - 1;
    const int f0 = 16 * s + 4 * q;
    f4 st = {0.f, 0.f, 0.f, 0.f};
    if (u == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = f0 + r;
            if (f < O) st[r] = td.s0_base < 0 ? p.ctrl->state[f] : p.s0[(size_t)(td.s0_base + slot0) * O + f];
        }
    }
    const float *actrow = p.actions + (size_t)(td.act_base + slot0) * H * A;

    float d_prev = 0.f, c_prev = 0.f, cum = 0.f;
    bool done = false;
    const int nk = 1 + p.sc.n_cost;
    // reward / cost / done bookkeeping of step T_ (wave 0): the same statements as cem_rollout_tile's CEM_BOOKKEEP
#define CEM_W8_BOOKKEEP(T_) do { if (w == 0) { \
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

    f4 own = {0.f, 0.f, 0.f, 0.f};                            // the wave's block of the current stage's INPUT, in registers
    f4 nb = *reinterpret_cast<const f4 *>(bias_h + ob);      // layer-0 bias of the wave's output block

    for (int t = -1; t < H; ++t) {
        if (t >= 0) {
            {   // layer 0
                f4 acc = nb;
                nb = *reinterpret_cast<const f4 *>(bias_h + (p.L > 1 ? 1 : 0) * CEM_U + ob);
                cem_w8_stage<4, true>(acc, own, wq, smem, xw ^ XB, lane, s, u);
                CEM_W8_BOOKKEEP(t - 1);                       // the stage's barrier published step t-1's scorer terms
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
                *reinterpret_cast<f4 *>(smem + xw + ((2 * s + u) * 64 + lane) * 16) = acc;
                own = acc;
                xw ^= XB;
            }
            for (int l = 1; l < p.L; ++l) {
                f4 acc = nb;
                nb = *reinterpret_cast<const f4 *>(bias_h + (l + 1 < p.L ? l + 1 : 0) * CEM_U + ob);
                cem_w8_stage<CEM_NG, false>(acc, own, wq, smem, xw ^ XB, lane, s, u);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
                *reinterpret_cast<f4 *>(smem + xw + ((2 * s + u) * 64 + lane) * 16) = acc;
                own = acc;
                xw ^= XB;
            }
        }
        // ---- heads + epilogue of input block s (cem_rollout_tile's block for i = 0, c = 0), split between the slot's two waves ----
        const int tn = (t + 1 < H) ? t + 1 : H - 1;
        const float live = (t >= 0) ? 1.0f : 0.0f;
        const float sampling = p.sampling ? 1.0f : 0.0f;
        const float goalm = p.sc.goal_mode ? 1.0f : 0.0f;
        // everything the epilogue needs from memory is requested before the MFMA stage
        f4 mn4, rd4, om4, isact4, sel0, sel1, act4, eps4, hbias;
        if (u == 0) {
            mn4 = *reinterpret_cast<const f4 *>(p.nmin + f0);
            rd4 = *reinterpret_cast<const f4 *>(p.nrdelta + f0);
            hbias = *reinterpret_cast<const f4 *>(bias_mu + f0);
            om4 = *reinterpret_cast<const f4 *>(p.omask + f0) * live;
            isact4 = *reinterpret_cast<const f4 *>(p.omask + CEM_U + f0);
            sel0 = *reinterpret_cast<const f4 *>(p.kind_sel + f0);
            sel1 = *reinterpret_cast<const f4 *>(p.kind_sel + CEM_U + f0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int af = f0 + r - O; af = af < 0 ? 0 : (af >= A ? A - 1 : af);
                act4[r] = actrow[tn * A + af];
            }
        } else {
            hbias = *reinterpret_cast<const f4 *>(bias_var + f0);
            eps4 = cem_normal4((uint32_t)(td.noise_row_base + slot0), (uint32_t)t, (uint32_t)p.it, (uint32_t)(4 * s + q), CEM_STREAM_MODEL, key);
            eps4 = eps4 * sampling;
        }
        f4 hacc = hbias;
        if (t >= 0) {
            if (s < p.KB_obs) cem_w8_stage<CEM_NG, false>(hacc, own, wq, smem, xw ^ XB, lane, s, u);     // wave-uniform
            else __syncthreads();                             // keep the barrier count of slots without observation features
            if (u == 1) {
                f4 t4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float sd = __builtin_amdgcn_sqrtf(cem_softplus(hacc[r]) + 1e-4f);
                    t4[r] = sd * eps4[r];                     // the product of Normal.sample = loc + scale * eps; the add is the lower wave's
                }
                sde[s * 64 + lane] = t4;
            }
            __syncthreads();
        }
        if (u == 0) {
            f4 t4 = {0.f, 0.f, 0.f, 0.f};                     // prologue (t = -1): no heads; the "update" adds exactly 0 (om4 = 0)
            if (t >= 0) t4 = sde[s * 64 + lane];
            float pm[CEM_NKIND];
#pragma unroll
            for (int k = 0; k < CEM_NKIND; ++k) pm[k] = __builtin_inff();
            f4 sn = st, x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = hacc[r] + t4[r];
                sn[r] = sn[r] + d * om4[r];
                const float lid = fminf(fmaxf(p.sc.D - p.sc.D * (1.0f - sn[r]), 0.f), p.sc.D);
                const float gv = goalm != 0.f ? fmaxf(sn[r], 0.f) : lid;
                pm[0] = fminf(pm[0], fmaxf(gv, sel0[r]));
                pm[1] = fminf(pm[1], fmaxf(lid, sel1[r]));
                const float xv = __builtin_fmaf(isact4[r], act4[r], sn[r]);
                x[r] = (xv - mn4[r]) * rd4[r];
            }
            st = sn;
            *reinterpret_cast<f4 *>(smem + xw + (s * 64 + lane) * 16) = x;
            own = x;                                          // the lower wave's own input block of the next layer-0 stage
            if (nk > 2) {
                for (int k = 2; k < nk; ++k) {
                    const f4 selk = *reinterpret_cast<const f4 *>(p.kind_sel + k * CEM_U + f0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float lid = fminf(fmaxf(p.sc.D - p.sc.D * (1.0f - st[r]), 0.f), p.sc.D);
                        pm[k] = fminf(pm[k], fmaxf(lid, selk[r]));
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < CEM_NKIND; ++k) {
                if (k < 2 || k < nk) {
                    const uint32_t mb = __float_as_uint(pm[k]);
                    const auto r16 = __builtin_amdgcn_permlane16_swap(mb, mb, false, false);
                    const float m16 = fminf(__uint_as_float(r16[0]), __uint_as_float(r16[1]));
                    const uint32_t m16b = __float_as_uint(m16);
                    const auto r32 = __builtin_amdgcn_permlane32_swap(m16b, m16b, false, false);
                    part[(k * 4 + s) * 64 + j] = fminf(__uint_as_float(r32[0]), __uint_as_float(r32[1]));
                }
            }
        }
        xw ^= XB;
    }
    __syncthreads();
    CEM_W8_BOOKKEEP(H - 1);
    if (w == 0 && lane < td.cnt) p.ret[td.row_base + lane] = cum;
}
#undef CEM_W8_BOOKKEEP
