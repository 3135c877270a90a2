// cem_device.h — gfx950 (CDNA4) device code of the CEM-MPC planner.
//
// Kernels (one CEM iteration = rollout [-> reduce] -> select; the sampler is the rollout tiles' prologue, the particle mean of the
// CemMpc objective is folded into the select's key staging on a single-rank whole plan):
//   cem_tile_sample_actions  cem_mpc.py:44-48   clip(eps*sigma+mu, lb, ub) -> actions[N][H][A], per tile, inside the rollout launch
//   cem_rollout_kernel  cem_mpc.py:49-55        tile x P, unfold_sequences (transition_model.py:64-77) through
//                                               the ensemble MLP (mlp_ensemble.py:59-61,122-132,189-193) and
//                                               the reward/cost scorer (safety_gym.py:110-192) with the done
//                                               masking of mpc_policy.py:26-37 / safe_cem_mpc.py:82-93, fused:
//                                               traj[B][H+1][O] is never materialised.
//   cem_reduce_kernel   mpc_policy.py:38-39, safe_cem_mpc.py:94-96,110-120   particle mean, Beta safety filter
//   cem_select_kernel   cem_mpc.py:56-67        top_k, best-so-far, moments, smoothing, early stop
//
// Rollout kernel design: a workgroup of 4 waves owns a tile of 16*RC rows of ONE ensemble
// member for the whole horizon.  Every dense layer is computed transposed, out^T[U x rows] = W^T . h^T, on
// v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain): A = a 16-feature block of W^T, B = 16 rows of h^T.  In
// that orientation the D layout of one layer (column = row of the batch on the lane, 4 consecutive output
// features in the 4 accumulator registers) IS the B-operand layout of the next layer's k-step, so
// activations move between layers as whole accumulator registers: each wave computes 32 of the 128 output
// features, publishes them with 16-B LDS stores, and after one barrier every wave re-reads all 128 with
// 16-B LDS loads, conflict-free, no shuffles, no transposes.  Weights are pre-packed on the host in
// A-operand order per (member, wave) as one linear stream and prefetched L2 -> VGPR three groups ahead.
// The state s_t lives in registers of the wave that owns its 16-feature block for all H steps.
// fp32 MFMA and VALU instructions share one issue pipe, so everything beside the MFMAs is written to be few instructions: the
// epilogue (heads -> Normal sample -> state update -> scorer terms -> next scaled input) works on a lane's four features at a
// time in packed v_pk_{fma,mul,add}_f32 (one issue slot for two elements), per-feature constants come from one per-member table
// through a single buffer resource, the sampled actions from a padded quad layout (one 16-byte load).  Tiles that share a CU
// take turns at issue priority (rotating with the step) and at the heavier wave roles (rotating with the tile).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));

#define CEM_U 128            // hidden units (config/models.yaml:11)
// rows of the per-member table RolloutParams::etab (each 128 floats, zero / identity padded)
#define CEM_ET_NMIN 0        // scale(): inputs_min                              (transition_model.py:79-87)
#define CEM_ET_RDELTA 1      // scale(): 1 / delta, the 1.01 rule applied
#define CEM_ET_BMU 2         // bias of the mean head                            (mlp_ensemble.py:33)
#define CEM_ET_BVAR 3        // bias of the variance head                        (mlp_ensemble.py:34)
#define CEM_ET_OBS 4         // 1.0 on observation features
#define CEM_ET_ACT 5         // 1.0 on action features
#define CEM_ET_SEL0 6        // -inf on the features of scorer kind 0 (goal), +inf elsewhere
#define CEM_ET_SEL1 7        // ... of the first cost kind
#define CEM_ET_ROWS 8        // then L rows: the hidden layers' biases
#define CEM_NG 8             // 16-feature blocks in U
#define CEM_NKIND 5          // goal + up to 4 cost kinds
#define CEM_PART_FLOATS (CEM_NKIND * 4 * 64)
// dynamic LDS of the tuned rollout kernels: two activation-exchange buffers, the scorer-term scratch, and the member's CEM_ET_ROWS
// per-feature table rows (4 KB) the epilogue reads every step
#define CEM_TAB_LDS_BYTES (CEM_ET_ROWS * CEM_U * 4)
#define CEM_ROLLOUT_LDS_BYTES(RC_) ((size_t)2 * (RC_) * CEM_NG * 1024 + CEM_PART_FLOATS * 4 + CEM_TAB_LDS_BYTES)

struct TileDesc {
    int32_t row_base;        // local row index of slot 0 (index into returns/costs/traj)
    int32_t cnt;             // valid rows in the tile (<= 16*RC)
    int32_t member;          // ensemble member of every row of the tile (mlp_ensemble.py:123-126)
    int32_t act_base;        // action-sequence index of slot 0
    int32_t noise_row_base;  // GLOBAL row id of slot 0 (Philox counter / eps_model row): shard invariant
    int32_t s0_base;         // -1: broadcast state (cem_mpc.py:53); else row index into s0[B][O]
};

struct CtrlBlock {
    uint32_t seed_lo, seed_hi, call_lo, call_hi;
    int32_t done;            // early stop reached (cem_mpc.py:66-67)
    int32_t iters;           // iterations run
    float best_score;        // best_so_far_score (cem_mpc.py:42)
    int32_t fault;           // bit 0: a floating rollout segment never got its work-queue entry (bounded spin) -> CEM_ERR_DEVICE; bit 1: a grid
                             // barrier of the fused select expired and the iteration's select has NOT been redone yet; bit 2: it expired and
                             // cem_msel_solo_kernel redid that select (same bits as select_mode 2): the plan is valid, the host stops fusing
    float state[CEM_U];      // the observation (cem_mpc.py:32)
    float best[32];          // best_so_far (cem_mpc.py:41)
    uint32_t seq;            // the host's plan counter: the kernel that completes the plan's result echoes it into result[36] LAST, so
                             // the host can watch pinned memory for it instead of going through a stream synchronisation
    uint32_t inject;         // test hook (cem_planner_inject_fault): 1 = the first grid barrier of this plan's first fused select expires on its
                             // last workgroup as if it had been starved — exercises the recovery path without loading the GPU
};

struct ScorerDev {
    int32_t goal_mode, goal_lo, goal_hi;
    float D;                 // lidar_max_dist
    float goal_thresh;       // fl32(goal_size * 0.8)  (safety_gym.py:116)
    float reward_distance, reward_goal, reward_clip;
    int32_t indicator, n_cost;
    int32_t cost_lo[4], cost_hi[4];
    float cost_size[4];
};

struct RolloutParams {
    const TileDesc *tiles;
    const f4 *wpack;             // packed weight streams [E][member_stride_f4]
    const float *bias_h;         // [E][L][128]
    const float *bias_mu;        // [E][128] zero padded
    const float *bias_var;       // [E][128]
    const float *nmin;           // [128] scale(): inputs_min, 0 on padding
    const float *nrdelta;        // [128] scale(): 1/delta (1.01 rule applied), 1 on padding
    const float *omask;          // [2][128] 1.0 on observation features / on action features, else 0
    const float *kind_sel;       // [CEM_NKIND][128] -inf where the feature belongs to scorer kind k (goal, costs...), +inf elsewhere
    const float *etab;           // [E][CEM_ET_ROWS + L][128] everything the hot kernel's epilogue and stage prologues read per feature, one
                                 // table per member (rows: CEM_ET_*), so that a wave addresses it as ONE buffer with a lane offset
    const f4 *act_pad;           // [N][H][act_nq] the sampled actions again, as the feature quads of the network input that hold an
                                 // action (quad act_q0 + i of the 128-feature input; zeros off the action features): the hot kernel
                                 // (MODE 0) fetches a lane's four input features with one 16-byte load.  MODE 1 reads `actions`
    uint32_t act_pad_bytes;
    int32_t act_q0, act_nq;
    const float *s0;             // [O] broadcast or [B][O]
    const float *actions;        // [n_act][H][A]
    const float *eps_model;      // nullptr -> Philox; else this iteration's [H][Btot][O]
    const CtrlBlock *ctrl;
    float *ret;                  // [Bloc] done-masked return per row
    uint8_t *costs;              // [H][Bloc] masked cost per step (safe variant) or nullptr
    float *traj, *mu_out, *sd_out;   // debug outputs (MODE 1 instantiation only)
    long long *stamps;           // [n_tiles][4 waves][8] cycle stamps, written only by -DCEM_STAMPS diagnostic builds
    uint32_t member_stride_f4;
    uint32_t wave_off_f4[4];
    uint32_t wave_groups[4];
    int32_t O, A, L, H, KB_in, KB_obs;
    int32_t Bloc;
    int32_t Btot;
    int32_t it;
    int32_t variant, sampling, check_done;
    ScorerDev sc;
    // the sampler, folded into the rollout launch (cem_tile_sample_actions; null musig: the caller supplied `actions`, cem_unfold_sequences)
    const float *musig;          // [2][H][A] this iteration's mu, sigma (cem_mpc.py:39-40,64-65)
    const float *eps_act;        // this iteration's [N][H][A] standard normals, or null -> Philox
    const float *act_bounds;     // [2][32] lb, ub of tf.clip_by_value (cem_mpc.py:48)
    float *actions_w;            // = actions, written here
    float *act_pad_w;            // = act_pad as floats: action a of (n, t) at [(n*H + t)*pad_floats + pad_shift + a]
    int32_t pad_shift, pad_floats;
    int32_t N, Nloc, n_off;      // all candidates; this rank's shard [n_off, n_off + Nloc)
    // horizon-segment work queue (cem_rollout_seg_kernel): items (segment, tile) in segment-major order
    uint32_t *seg_queue;         // [3] ticket counter, FIFO tail, finished floating tiles; zero between launches (the launch's last floating
                                 // workgroup resets it; cem_init_kernel clears it at the start of every plan)
    uint32_t *seg_flags;         // [n_float * (n_seg - 1)] FIFO of ready floating items ((tile << 8 | segment) + 1; 0 = not written yet)
    f4 *seg_state;               // [n_float][2*NFW*RC*256 + 64] state a floating tile carries across a segment boundary
    int32_t seg_len, n_seg, n_tiles, n_pinned;
};

// ---------------------------------------------------------------------------------------------------------
// Philox4x32-7 counter RNG (Salmon et al., SC'11: 7 rounds is the fewest that passes BigCrush; every VALU instruction
// is paid in full next to fp32 MFMAs, and the 3 extra rounds of the -10 variant are 27 of them per draw) + Box-Muller.  Counter = (index, t | it<<16,
// sub | stream<<16, call_lo), key = (seed_lo, seed_hi ^ call_hi): a pure function of GLOBAL indices, so
// every rank of a candidate-sharded plan draws bit-identical noise for the same (candidate, particle).
// ---------------------------------------------------------------------------------------------------------
#define CEM_STREAM_MODEL 0u
#define CEM_STREAM_ACT 1u
#define CEM_STREAM_OUT 2u

__device__ __forceinline__ void philox4x32_7(uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        // one 32x32->64 multiply (v_mad_u64_u32) per product instead of a mul_hi + mul_lo pair
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c0, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        // hi ^ c ^ k as ONE three-input bit operation (v_bitop3_b32, truth table 0x96 = xor3): left alone the compiler emits two v_xor
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(hi0, c3, k1, 0x96);
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

struct PhiloxKey { uint32_t k0, k1, c3; };

__device__ __forceinline__ PhiloxKey cem_key(const CtrlBlock *ctrl)
{
    PhiloxKey k; k.k0 = ctrl->seed_lo; k.k1 = ctrl->seed_hi ^ ctrl->call_hi; k.c3 = ctrl->call_lo; return k;
}

typedef float f2 __attribute__((ext_vector_type(2)));
#define CEM_BM_RSCALE (-1.3862943611198906f)     // -2 ln 2: r = sqrt(-2 ln u) = sqrt(CEM_BM_RSCALE * log2 u)

// Four standard normals of counter (idx, t, it, sub, stream): Philox words -> uniforms -> Box-Muller.
//   u = fl32(fl32(word) * 2^-32 + 2^-33)  (v_cvt_f32_u32 + one fma, the pair of a radius / angle as one v_pk_fma_f32): in (0, 1],
//       never 0, spacing 2^-24 near 1 and finer towards 0 (largest radius sqrt(66 ln 2) = 6.76)
//   z0 = r(u0) cos(2 pi u1), z1 = r(u0) sin(2 pi u1), z2 = r(u2) cos(2 pi u3), z3 = r(u2) sin(2 pi u3),  r(u) = sqrt(rscale * log2 u)
// on v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in revolutions).  rscale = CEM_BM_RSCALE; 0 gives four zeros
// exactly (sampling_propagation False costs no extra multiply).  tests/test_gpu_rng.py restates this in numpy (known answers).
__device__ __forceinline__ f4 cem_normal4(uint32_t idx, uint32_t t, uint32_t it, uint32_t sub, uint32_t stream, const PhiloxKey key,
                                          const float rscale = CEM_BM_RSCALE)
{
    uint32_t c0 = idx, c1 = t | (it << 16), c2 = sub | (stream << 16), c3 = key.c3;
    philox4x32_7(c0, c1, c2, c3, key.k0, key.k1);
    const f2 h = {1.1641532182693481e-10f, 1.1641532182693481e-10f};       // 2^-33
    const f2 ur = __builtin_elementwise_fma((f2){(float)c0, (float)c2}, (f2){2.3283064365386963e-10f, 2.3283064365386963e-10f}, h);   // radii
    const f2 ua = __builtin_elementwise_fma((f2){(float)c1, (float)c3}, (f2){2.3283064365386963e-10f, 2.3283064365386963e-10f}, h);   // angles
    const f2 l = (f2){__builtin_amdgcn_logf(ur[0]), __builtin_amdgcn_logf(ur[1])} * rscale;
    const float ra = __builtin_amdgcn_sqrtf(l[0]), rb = __builtin_amdgcn_sqrtf(l[1]);
    const f2 za = (f2){__builtin_amdgcn_cosf(ua[0]), __builtin_amdgcn_sinf(ua[0])} * ra;
    const f2 zb = (f2){__builtin_amdgcn_cosf(ua[1]), __builtin_amdgcn_sinf(ua[1])} * rb;
    return (f4){za[0], za[1], zb[0], zb[1]};
}

// tf.math.softplus.  Eigen evaluates x (x > 13.94), exp(x) (x < -13.94), log1p(exp(x)) otherwise;
// all three branches are the one function max(x,0) + log1p(exp(-|x|)) to within 2e-6 relative, computed here
// branch-free: t = exp(-|x|) on v_exp_f32, log1p(t) = 2 atanh(z), z = t/(2+t) <= 1/3, with 2 atanh(z)/z as a degree-4 minimax
// polynomial in z^2 on [0, 1/9] (relative error 4e-9; the factor 2 is folded into the coefficients).  Written on four values
// at a time: beside fp32 MFMAs a packed v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 costs what ONE scalar VALU instruction costs
// (scripts/mfma_microbench5.hip), so the element chains of the rollout epilogue run two elements per instruction.  The scalar
// form below is the same sequence of operations (bit-identical).  Measured against the fp64 oracle in tests/test_gpu_parity.py.
#define CEM_SP_C4 0.28191542625427246f
#define CEM_SP_C3 0.27957665920257568f
#define CEM_SP_C2 0.4002511501312256f
#define CEM_SP_C1 0.66666311025619507f
#define CEM_SP_C0 2.0f
__device__ __forceinline__ f4 cem_splat4(const float v) { return (f4){v, v, v, v}; }
// a - b on four values as two v_pk_add_f32 with the second operand negated: the compiler scalarises a <2 x float> fsub into
// v_sub_f32 (only packed add / mul / fma are selected), and folds fma(b, -1, a) back into that fsub.  Same result bit for bit.
__device__ __forceinline__ f4 cem_sub4(const f4 a, const f4 b)
{
    f2 lo, hi;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"((f2){a[0], a[1]}), "v"((f2){b[0], b[1]}));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"((f2){a[2], a[3]}), "v"((f2){b[2], b[3]}));
    return (f4){lo[0], lo[1], hi[0], hi[1]};
}
__device__ __forceinline__ f4 cem_softplus4(const f4 x)
{
    f4 t, rc, mx;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = __builtin_amdgcn_exp2f(-1.4426950408889634f * __builtin_fabsf(x[r]));
    const f4 den = t + 2.0f;
#pragma unroll
    for (int r = 0; r < 4; ++r) rc[r] = __builtin_amdgcn_rcpf(den[r]);
    const f4 z = t * rc;
    const f4 z2 = z * z;
    f4 q = __builtin_elementwise_fma(cem_splat4(CEM_SP_C4), z2, cem_splat4(CEM_SP_C3));
    q = __builtin_elementwise_fma(q, z2, cem_splat4(CEM_SP_C2));
    q = __builtin_elementwise_fma(q, z2, cem_splat4(CEM_SP_C1));
    q = __builtin_elementwise_fma(q, z2, cem_splat4(CEM_SP_C0));
#pragma unroll
    for (int r = 0; r < 4; ++r) mx[r] = fmaxf(x[r], 0.f);
    return __builtin_elementwise_fma(z, q, mx);
}
__device__ __forceinline__ float cem_softplus(float x)
{
    const float t = __builtin_amdgcn_exp2f(-1.4426950408889634f * __builtin_fabsf(x));
    const float z = t * __builtin_amdgcn_rcpf(t + 2.0f);
    const float z2 = z * z;
    float q = __builtin_fmaf(CEM_SP_C4, z2, CEM_SP_C3);
    q = __builtin_fmaf(q, z2, CEM_SP_C2);
    q = __builtin_fmaf(q, z2, CEM_SP_C1);
    q = __builtin_fmaf(q, z2, CEM_SP_C0);
    return __builtin_fmaf(z, q, fmaxf(x, 0.f));
}

// mlp_params['activation'] (mlp_ensemble.py:14,20; enum cem_activation of cem_mpc.h): the hidden layers' nonlinearity on the generic
// paths (cem_rollout_wide_kernel, cem_train_step_kernel) — the tuned kernels are relu only.  tanh / exp / expm1 are the device
// library's (1-2 ulp).  The derivative is written as a function of the layer's OUTPUT h = f(z), which is what the backward pass
// holds (TensorFlow's own EluGrad / SoftplusGrad / ReluGrad use the same forms): relu [h > 0], tanh 1 - h^2, sigmoid h (1 - h),
// elu h + 1 below zero, leaky_relu 0.2 at and below zero, softplus sigma(z) = 1 - exp(-h), selu h + scale alpha below zero and scale above.
__device__ __forceinline__ float cem_activation_fwd(const int a, const float v)
{
    switch (a) {
    case 1: return tanhf(v);
    case 2: return 1.0f / (1.0f + expf(-v));
    case 3: return v > 0.f ? v : expm1f(v);
    case 4: return v > 0.f ? v : 0.2f * v;
    case 5: return cem_softplus(v);
    case 6: return v > 0.f ? 1.0507009873554805f * v : 1.7580993408473766f * expm1f(v);     // tf.nn.selu: scale * (z or alpha * (e^z - 1)); scale * alpha = 1.7580993
    case 7: return v / (1.0f + expf(-v));                                                     // tf.nn.swish = tf.nn.silu: z * sigmoid(z)
    case 8: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));                        // tf.nn.gelu (approximate=False, its default): z * Phi(z)
    default: return fmaxf(v, 0.f);
    }
}
// swish and gelu are not monotone: their derivative is NOT a function of the layer's output.  The trainer keeps these layers' PRE-activations
// (cem_train.h: GemmEpi::outz) and gates with f'(z):  swish' = s + z s (1 - s), s = sigmoid(z);  gelu' = Phi(z) + z phi(z).
#define CEM_ACT_NEEDS_Z(a) ((a) >= 7)
__device__ __forceinline__ float cem_activation_gate_z(const int a, const float d, const float z)
{
    if (a == 7) { const float sg = 1.0f / (1.0f + expf(-z)); return d * (sg + z * sg * (1.0f - sg)); }
    return d * (0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * expf(-0.5f * z * z));
}
__device__ __forceinline__ float cem_activation_gate(const int a, const float d, const float h)     // d * f'(z), given h = f(z)
{
    switch (a) {
    case 1: return d * (1.0f - h * h);
    case 2: return d * (h * (1.0f - h));
    case 3: return h < 0.f ? d * (h + 1.0f) : d;
    case 4: return h > 0.f ? d : 0.2f * d;
    case 5: return d * (1.0f - expf(-h));
    case 6: return h < 0.f ? d * (h + 1.7580993408473766f) : d * 1.0507009873554805f;       // TensorFlow's SeluGrad, on the layer's output
    default: return h > 0.f ? d : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// rollout kernel
// ---------------------------------------------------------------------------------------------------------
struct AGroup { f4 a, b; };

// Per-wave K order: "own blocks first".  The 32 output features a wave computes in one layer are, after ReLU, already
// in its registers in the B-operand layout of the next layer.  So every stage starts its MFMA chain on the wave's OWN
// input blocks with no wait at all, and the workgroup barrier + the LDS reads of the other blocks are issued underneath
// those MFMAs.  The k-blocks of a stage are therefore visited in a per-wave order (phi = 0..KF-1 -> block), and the
// host packs each wave's weight stream in exactly that order (cem_capi.hip pack_member, same two functions).
__host__ __device__ inline int cem_perm_hidden(int w, int phi)          // hidden-layer / heads input: own blocks 2w, 2w+1
{
    return phi < 2 ? 2 * w + phi : ((phi - 2 < 2 * w) ? phi - 2 : phi);
}
__host__ __device__ inline int cem_perm_l0(int w, int nfw, int phi)      // layer-0 input: own blocks w, w+4, ...
{
    if (phi < nfw) return w + 4 * phi;
    int n = phi - nfw;
    for (int F = 0; F < 4 * nfw; ++F) { if ((F & 3) == w) continue; if (n == 0) return F; --n; }
    return 0;
}

// Weight prefetch ring: 4 register slots over this wave's linear weight stream, always 3 groups ahead of the
// MFMAs.  Every stage consumes a multiple of 4 groups (layer 0 is zero-padded to 4*NFW groups on the host), so the
// slot of stage-local group phi is the compile-time constant phi & 3: no register moves, no branches.
struct WRing {
    __amdgpu_buffer_rsrc_t rsrc;   // this wave's weight stream as a buffer: group offset in an SGPR, lane offset in one constant
    int voff;                      // VGPR -> no per-group VALU address arithmetic inside the MFMA stream
    int n, pos;
    AGroup slot[4];
    __device__ __forceinline__ AGroup ld(int g) const
    {
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        AGroup r;
#ifdef CEM_DBG_NOWLOAD         // timing-only diagnostic: no weight traffic at all
        r.a = (f4){(float)g, 1.f, 2.f, 3.f}; r.b = r.a; return r;
#endif
        const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, g * 2048, 0);
        const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024, g * 2048, 0);
        r.a = __builtin_bit_cast(f4, a); r.b = __builtin_bit_cast(f4, b);
        return r;
    }
    __device__ __forceinline__ void init(const f4 *b, int lane_, int n_)
    {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(b), 0, n_ * 2048, 0x00020000);
        voff = lane_ * 16; n = n_;
        slot[0] = ld(0); slot[1] = ld(1 % n_); slot[2] = ld(2 % n_); slot[3] = slot[2];
        pos = 3 % n_;
    }
};

// Groups of lead the LDS reads of the other waves' blocks get over their MFMAs (template parameter LA of cem_mfma_stage).  A group
// is 8 RC MFMAs = 256 RC cycles.  One-chunk tiles of the obs+act <= 64 family read all six blocks right after the barrier
// (LA 6: B1 -2.6 %, B2 unchanged; 151 VGPRs, still three workgroups per CU); every other form reads one group ahead — more would
// cost the RC = 2 kernels and the one-chunk obs+act > 64 kernel their third resident workgroup (measured: -3 % at the B5 rank).
#ifndef CEM_LA_RC1_NFW2
#define CEM_LA_RC1_NFW2 1
#endif
#ifndef CEM_LA_RC2
#define CEM_LA_RC2 1
#endif
#ifndef CEM_LA_RC3
#define CEM_LA_RC3 1
#endif
#ifndef CEM_LA_RC4
#define CEM_LA_RC4 1
#endif
#define CEM_LDS_AHEAD_OF(RC_, NFW_) ((RC_) == 1 ? ((NFW_) == 1 ? 6 : CEM_LA_RC1_NFW2) : ((RC_) == 2 ? CEM_LA_RC2 : ((RC_) == 3 ? CEM_LA_RC3 : CEM_LA_RC4)))
#define CEM_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// One dense stage for this wave: acc{0,1}[c] += W^T-groups . hB.  hB[0..NOWN-1] (the wave's own blocks) are already
// in registers; if EXCHANGE, the barrier that publishes the other waves' blocks and the LDS reads of hB[NOWN..KF-1]
// are issued after the first group's MFMAs.  L0IN selects the block permutation (layer-0 input vs hidden input).
// XMODE 1: barrier after the first group, then just-in-time LDS reads of the other waves' blocks; 2: the same reads without
// the barrier (the blocks were published and waited for by an earlier stage on the same input and are re-read rather than
// kept: eight blocks held in registers across two stages cost the obs+act > 64 kernels their second resident workgroup).
#define CEM_X_EXCHANGE 1
#define CEM_X_REREAD 2
template <int RC, int KF, int NOWN, bool L0IN, int XMODE, int LA>
__device__ __forceinline__ void cem_mfma_stage(f4 (&acc0)[RC], f4 (&acc1)[RC], f4 (&hB)[CEM_NG][RC], WRing &wq,
                                               const char *smem, const int xr, const int lane, const int w)
{
    static_assert(KF % 4 == 0, "stage lengths must keep the ring phase");
#pragma unroll
    for (int P = 0; P < KF; ++P) {
        wq.slot[(P + 3) & 3] = wq.ld(wq.pos);            // group P+3 of this stage (or the next stage's first groups)
        wq.pos = (wq.pos + 1 == wq.n) ? 0 : wq.pos + 1;
        // pin the prefetch here: unpinned, the machine scheduler sinks the load to just before its use and every
        // group of MFMAs eats a full L2 round trip
        __builtin_amdgcn_sched_barrier(0);
        if (XMODE != 0 && P >= 1) {
            // the other waves' blocks are read just in time, LA groups before their MFMAs: all of them at once
            // is 24*RC live registers from group 1 on, which at RC = 3 pushes the kernel over the 256 architectural VGPRs
#ifndef CEM_DBG_NOBARRIER      // (timing-only diagnostic builds may drop the barrier / the LDS reads; never the shipped library)
            if (P == 1 && XMODE == CEM_X_EXCHANGE) __syncthreads();   // every wave's blocks of the previous stage are in LDS
#endif
#pragma unroll
            for (int Q = NOWN; Q < KF; ++Q) {
                const bool now = (P == 1) ? (Q <= 1 + LA) : (Q == P + LA);
                if (now) {
#ifdef CEM_DBG_STATICADDR      // timing-only diagnostic: block index without the wave's role in it (compile-time LDS offsets: WRONG results) — what the per-read address arithmetic costs
                    const int F = Q;
#else
                    const int F = L0IN ? cem_perm_l0(w, KF / 4, Q) : cem_perm_hidden(w, Q);
#endif
#pragma unroll
                    for (int c = 0; c < RC; ++c)
#ifdef CEM_DBG_NOLDSREAD
                        hB[Q][c] = hB[Q & 1][c];
#else
                        hB[Q][c] = *reinterpret_cast<const f4 *>(smem + xr + ((c * CEM_NG + F) * 64 + lane) * 16);
#endif
                }
            }
        }
        const AGroup g = wq.slot[P & 3];
        // Canonical k order of an OUTPUT block b of a hidden / heads stage: block b itself, block b ^ 1, then the rest ascending.
        // The wave's two accumulators are blocks 2w and 2w + 1, so the second one visits the wave's own two input blocks swapped
        // (its weights are packed to match).  Every output block thus starts on the input block of the same index, whichever wave
        // computes it: a form of the kernel that spreads the output blocks over more waves sums in the same order (the 8-wave
        // workgroup of commit 0e0e847 was bit-identical to this kernel — and 1.5 % slower at B1, hence not kept; profiles/HISTORY.md 4.1).
        const int Pb = (!L0IN && P < 2) ? (P ^ 1) : P;            // (P is a compile-time constant once the loop is unrolled)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                acc0[c] = CEM_MFMA(g.a[r], hB[P][c][r], acc0[c]);
                acc1[c] = CEM_MFMA(g.b[r], hB[Pb][c][r], acc1[c]);
            }
        }
    }
}

#ifdef CEM_STAMPS   // diagnostic build only: where a step's cycles go (never in the timed library)
#define CEM_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); const long long now_ = (long long)__builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_s_waitcnt(0xC07F); st_[i] += now_ - tprev_; tprev_ = now_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CEM_STAMP(i) do { } while (0)
#endif

// MODE 0: the planner's hot path (Philox noise, no debug outputs).  MODE 1: general path — explicit eps_model
// tensors (parity mode) and/or the trajectory / head-moment outputs of cem_unfold_sequences.
// Segment state crosses CUs — and XCDs, whose L2s are not coherent with each other for ordinary accesses.  An agent-scope
// acquire / release FENCE would make them so by invalidating / writing back the whole L2 (buffer_inv sc1 / buffer_wbl2 sc1),
// i.e. by evicting the ensemble weights every other workgroup of the XCD is streaming from it — measured 2.8x slower.  So the
// few KB of state move with sc1 loads and stores (performed at the device coherence point, no cache maintenance — what an
// agent-scope relaxed atomic access is on gfx940+), the flag likewise, and the order "state, then flag" is kept by EVERY wave
// waiting for its own stores' acknowledgements (an explicit s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the
// flag store (a barrier alone does not drain stores on gfx940+, nor does a workgroup-scope fence).
// ... as 16-byte buffer accesses with the sc1 cache-policy bit (aux bit 4 on gfx940+), the same instruction form the compiler
// emits for agent-scope relaxed atomics, four words at a time.
typedef unsigned int cem_u4 __attribute__((ext_vector_type(4)));
#define CEM_AUX_SC1 16
__device__ __forceinline__ f4 cem_ld_coherent(__amdgpu_buffer_rsrc_t rsrc, int byte_off)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, CEM_AUX_SC1));
}
__device__ __forceinline__ void cem_st_coherent(__amdgpu_buffer_rsrc_t rsrc, int byte_off, const f4 v)
{
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(cem_u4, v), rsrc, byte_off, 0, CEM_AUX_SC1);
}

__device__ __forceinline__ f4 cem_ld_tab(__amdgpu_buffer_rsrc_t rs, int voff, int row_bytes)
{
    return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, row_bytes, 0));
}

// closest_distance terms (safety_gym.py:188-192) of four features of one batch row, folded into the running minima of the goal
// kind and of the first cost kind: lid = clip(D - D (1 - x), 0, D); a feature outside a kind's slice has sel = +inf (no effect),
// one inside has sel = -inf.  (The goal kind in observe_goal_dist mode is redone by the caller: rare, kept out of the hot block.)
__device__ __forceinline__ void cem_scorer_terms(const f4 sn, const float D, const f4 sel0, const f4 sel1, float &pm0, float &pm1)
{
    const f4 lr = cem_sub4(cem_splat4(D), D * cem_sub4(cem_splat4(1.0f), sn));       // D - D (1 - x), each operation rounded
    f4 lid;
#pragma unroll
    for (int r = 0; r < 4; ++r) lid[r] = __builtin_amdgcn_fmed3f(lr[r], 0.f, D);          // = min(max(lr, 0), D) for D >= 0
    pm0 = fminf(fminf(pm0, fmaxf(lid[0], sel0[0])), fmaxf(lid[1], sel0[1]));
    pm0 = fminf(fminf(pm0, fmaxf(lid[2], sel0[2])), fmaxf(lid[3], sel0[3]));
    pm1 = fminf(fminf(pm1, fmaxf(lid[0], sel1[0])), fmaxf(lid[1], sel1[1]));
    pm1 = fminf(fminf(pm1, fmaxf(lid[2], sel1[2])), fmaxf(lid[3], sel1[3]));
}

// cem_mpc.py:44-48 inside the rollout launch: a = clip(eps * sigma + mu, lb, ub) (tf.random.normal(mean, stddev) = eps * stddev + mean as
// a separate multiply and add; tf.clip_by_value), eps keyed on the GLOBAL (candidate, step, iteration) or read from the caller's tensor.
// A workgroup samples what ITS tile is about to read — its own candidates, steps [t0, t1) — into both layouts (`actions` [N][H][A] for
// the select kernel and the explicit-tensor rollout forms, the padded quads for the hot kernel) and reads it back after one barrier:
// no workgroup ever consumes another's samples inside the launch, so nothing has to be coherent across CUs or XCDs.  The P tiles
// that share a candidate (one per particle) each draw the same values from the same counters and store the same words — a few
// hundred VALU instructions per tile and horizon, next to 30 steps of ~10 K cycles each.  On a candidate-sharded rank (world > 1) the
// select still runs over ALL N candidates, so the tiles also share out the sequences of the OTHER ranks' candidates (`actions` only):
// tile b of n_tiles takes the b-th slice of those (N - Nloc) * H * ceil(A / 4) draws.  With this the sampler costs no launch of its own
// (it was 4.8 us + a graph-node gap per iteration at B2).  The caller waits (vmcnt(0)) and barriers before the first action load.
__device__ __forceinline__ void cem_sample_store(const RolloutParams &p, const int n, const int t, const int z, const PhiloxKey key, const bool pad,
                                                 const bool natural = true)
{
    const int A = p.A, HA = p.H * A;
    f4 e;
    if (p.eps_act) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int a = 4 * z + r; e[r] = p.eps_act[((size_t)n * p.H + t) * A + (a < A ? a : A - 1)]; }
    } else e = cem_normal4((uint32_t)n, (uint32_t)t, (uint32_t)p.it, (uint32_t)z, CEM_STREAM_ACT, key);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int a = 4 * z + r;
        if (a < A) {
            float v = e[r] * p.musig[HA + t * A + a] + p.musig[t * A + a];             // tf.random.normal(mean, stddev)
            v = fminf(fmaxf(v, p.act_bounds[a]), p.act_bounds[32 + a]);                // tf.clip_by_value
            if (natural) p.actions_w[((size_t)n * p.H + t) * A + a] = v;
            if (pad) p.act_pad_w[((size_t)n * p.H + t) * p.pad_floats + p.pad_shift + a] = v;   // padding words stay 0 (zeroed at create)
        }
    }
}

__device__ __forceinline__ void cem_tile_sample_join()
{
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0): this wave's stores are acknowledged (a barrier alone does not drain them)
    __syncthreads();
}
// WAIT false: the stores are only issued; the caller joins (cem_tile_sample_join) once the rest of its tile set-up is issued too
// natural_all: every tile also stores its candidates in the [N][H][A] layout (the explicit-tensor rollout forms read that one back);
// otherwise only the tiles of particle 0 do — the select is its only reader then, and P - 1 of P copies were 2 MB of stores a launch
template <bool WAIT = true>
__device__ __forceinline__ void cem_tile_sample_actions(const RolloutParams &p, const int tile_idx, const int t0, const int t1, const bool foreign,
                                                        const bool natural_all = true)
{
    if (!p.musig) return;                                    // wave-uniform (a kernel argument)
    const TileDesc td = p.tiles[tile_idx];
    const bool natural = natural_all || td.row_base < p.Nloc;
    const int AZ = (p.A + 3) >> 2, nst = t1 - t0;
    const PhiloxKey key = cem_key(p.ctrl);
    const int own = td.cnt * nst * AZ;
    for (int idx = (int)threadIdx.x; idx < own; idx += 256) {
        const int z = idx % AZ, tt = (idx / AZ) % nst, r_ = idx / (AZ * nst);
        cem_sample_store(p, td.act_base + r_, t0 + tt, z, key, true, natural);
    }
    if (foreign && p.Nloc < p.N) {                           // the other ranks' candidates, shared out over this rank's tiles
        const long long total = (long long)(p.N - p.Nloc) * p.H * AZ;
        const long long per = (total + p.n_tiles - 1) / p.n_tiles;
        const long long lo = per * tile_idx, hi = (lo + per < total) ? lo + per : total;
        for (long long idx = lo + (long long)threadIdx.x; idx < hi; idx += 256) {
            const int z = (int)(idx % AZ), t = (int)((idx / AZ) % p.H);
            const int jn = (int)(idx / ((long long)AZ * p.H));
            cem_sample_store(p, jn < p.n_off ? jn : jn + p.Nloc, t, z, key, false);
        }
    }
    if (WAIT) cem_tile_sample_join();
}

// The same sampler as a launch of its own, over all N candidates once: what a plan uses when its rollout launch runs SEVERAL rounds of
// tiles per CU or many particles per candidate (the tiles' prologue samples a candidate once per particle and once per round of
// tiles on the critical path: measured +1.5 % on B3's launch, K = 16, against 0.2 % for this kernel; at B1 / B2 — every tile resident
// at once, five particles — the prologue costs what this launch plus its graph node cost, and saves the node).  Host rule: cem_capi.hip
// sample_in_rollout().
__global__ __launch_bounds__(256) void cem_sample_kernel(const RolloutParams p)
{
    if (p.check_done && p.ctrl->done) return;
    const int AZ = (p.A + 3) >> 2;
    const int total = p.N * p.H * AZ;
    const PhiloxKey key = cem_key(p.ctrl);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int z = idx % AZ, t = (idx / AZ) % p.H, n = idx / (AZ * p.H);
        cem_sample_store(p, n, t, z, key, true);
    }
}

// One tile for steps [t_begin, t_end) of the horizon.  SEG false: the whole horizon (t_begin = 0, t_end = H).  SEG true: one
// segment of it; what a tile carries across a segment boundary (state registers, its next layer-0 input blocks, the bookkeeping
// wave's reward / done state) goes through p.seg_state, so any workgroup on any CU can run the tile's next segment and the
// result is bit-identical to the unsegmented run.
template <int RC, int NFW, int MODE, bool SEG>
__device__ __forceinline__ void cem_rollout_tile(const RolloutParams &p, char *smem, const int tile_idx, const int t_begin, const int t_end)
{
    // The four waves' ROLES (which input / output feature blocks a wave owns, hence its weight stream; who keeps the books) rotate
    // with the tile: hardware wave v plays logical wave w = (v + tile rotation) mod 4.  The roles are not equally heavy — the
    // bookkeeping wave has about 70 VALU instructions per step more, and where the observation does not fill every wave's last
    // input block (obs 100: 7 blocks over 4 waves) one wave skips a whole heads stage (64 RC MFMAs of 704 per step) — and the tiles
    // that share a CU would otherwise all put their heavy roles on the same SIMDs.  Everything below is in terms of w.
    const int tid = (int)((threadIdx.x + 64u * (unsigned)((tile_idx + (tile_idx >> 8)) & 3)) & 255u);
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int j = lane & 15, q = lane >> 4;
    const TileDesc td = p.tiles[tile_idx];
    const bool resumed = SEG && t_begin > 0;
    const int wbk = 0;                                   // the logical wave that keeps the tile's reward / cost / done bookkeeping
    // this tile's slot of the hand-over buffer as a buffer resource: [2*NFW*RC][256 threads] f4 + [64 lanes] f4
    const __amdgpu_buffer_rsrc_t seg_rs = __builtin_amdgcn_make_buffer_rsrc(
        SEG ? const_cast<f4 *>(p.seg_state + (size_t)(tile_idx - p.n_pinned) * (2 * NFW * RC * 256 + 64)) : const_cast<f4 *>(p.wpack), 0,
        (2 * NFW * RC * 256 + 64) * 16, 0x00020000);
    const int O = p.O, A = p.A, H = p.H;
    constexpr int XB = RC * CEM_NG * 1024;
    constexpr int LA = CEM_LDS_AHEAD_OF(RC, NFW);
    float *part = reinterpret_cast<float *>(smem + 2 * XB);
    int xw = 0;                                          // LDS buffer the current stage's outputs go to
    const PhiloxKey key = cem_key(p.ctrl);
    const float rscale = p.sampling ? CEM_BM_RSCALE : 0.0f;     // sampling_propagation False: the model noise is exactly 0

    // descriptor inputs made provably wave-uniform (the tile descriptor load and the wave id are uniform in fact)
    const int member_u = __builtin_amdgcn_readfirstlane(td.member);
    WRing wq;
    wq.init(p.wpack + (size_t)member_u * p.member_stride_f4 + p.wave_off_f4[w], lane, (int)p.wave_groups[w]);

    // the member's feature tables as one buffer: row r at byte r * 512, this lane's feature quad f0 = 16 (w + 4 i) + 4 q at
    // byte 4 f0 of a row.  No 64-bit address arithmetic and no address registers besides tab_v.
    const __amdgpu_buffer_rsrc_t et_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(p.etab + (size_t)member_u * (CEM_ET_ROWS + p.L) * CEM_U), 0, (CEM_ET_ROWS + p.L) * CEM_U * 4, 0x00020000);
    const int tab_v = 64 * w + 16 * q;                   // + 256 i
    const int bias_v = 128 * w + 16 * q;                 // the wave's output blocks 2w (and 2w+1: + 64) of a hidden layer
    // The eight per-feature rows the epilogue needs every step (CEM_ET_NMIN .. CEM_ET_SEL1) are copied into LDS once per tile, one
    // 16-byte piece per thread, and read from there at the point of use (ds_read_b128, every 16 lanes the same address: a broadcast).
    // Fetched from memory they had to be requested a whole MFMA stage ahead and held in 32 registers across it — which was the
    // peak of the kernel's register use (RC 1: 153 -> 129 VGPRs, RC 2 / two input blocks: 219 -> 195).  The hidden layers' biases
    // (rows CEM_ET_ROWS + l) stay buffer loads a stage ahead: two registers' worth each.
    const char *tabl = smem + 2 * XB + CEM_PART_FLOATS * 4;
    {
        const int ht = (int)threadIdx.x;                 // (hardware thread id: any one-to-one assignment of the 256 pieces)
        *reinterpret_cast<f4 *>(const_cast<char *>(tabl) + ht * 16) = cem_ld_tab(et_rs, (ht & 31) * 16, (ht >> 5) * 512);
    }
#define CEM_TAB(ROW_, TV_) (*reinterpret_cast<const f4 *>(tabl + (ROW_) * 512 + (TV_)))
#define CEM_SEL0_ROW(TV_) CEM_TAB(CEM_ET_SEL0, TV_)       /* (what CEM_RARE_KINDS_AND_STORE reads; the other kernels that use that macro fetch it from memory) */
    // one barrier publishes the table rows AND (with the sampler in this launch: the kernel entry issued its stores without waiting)
    // the tile's own action samples; a resumed segment without a sampler has stage barriers in front of its first table read
    if (p.musig) cem_tile_sample_join();
    else if (!resumed) __syncthreads();

    // ---- state registers: wave w owns input feature blocks Fo = w + 4 i --------------------------------
    f4 s[NFW][RC];
    int slotc[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) { const int sl = 16 * c + j; slotc[c] = sl < td.cnt ? sl : td.cnt - 1; }
#pragma unroll
    for (int i = 0; i < NFW; ++i) {
        const int f0 = 16 * (w + 4 * i) + 4 * q;
#pragma unroll
        for (int c = 0; c < RC; ++c) {
            if (resumed) { s[i][c] = cem_ld_coherent(seg_rs, ((i * RC + c) * 256 + tid) * 16); continue; }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = f0 + r;
                float v = 0.f;
                if (f < O) v = td.s0_base < 0 ? p.ctrl->state[f] : p.s0[(size_t)(td.s0_base + slotc[c]) * O + f];
                s[i][c][r] = v;
            }
        }
    }

    // this lane's actions.  MODE 0: the padded quad layout (one 16-byte buffer load per unit and step, the step in the scalar
    // offset); MODE 1 (caller-supplied action tensors): the natural [n][H][A] layout, element by element.
    const __amdgpu_buffer_rsrc_t act_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(p.act_pad), 0, MODE == 0 ? p.act_pad_bytes : 0u, 0x00020000);
    int actv[NFW][RC];
    const float *actrow[RC];
#pragma unroll
    for (int c = 0; c < RC; ++c) {
        actrow[c] = p.actions + (size_t)(td.act_base + slotc[c]) * H * A;
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            int qi = 4 * (w + 4 * i) + q - p.act_q0;
            qi = qi < 0 ? 0 : (qi >= p.act_nq ? p.act_nq - 1 : qi);      // a quad without action features: any valid quad (its mask is 0)
            actv[i][c] = ((td.act_base + slotc[c]) * H * p.act_nq + qi) * 16;
        }
    }
#define CEM_LOAD_ACT(DST, I_, C_, TN_) do { \
        if (MODE == 0) DST = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(act_rs, actv[I_][C_], (TN_) * p.act_nq * 16, 0)); \
        else { _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
            int af = 16 * (w + 4 * (I_)) + 4 * q + r - O; af = af < 0 ? 0 : (af >= A ? A - 1 : af); \
            DST[r] = actrow[C_][(TN_) * A + af]; } } } while (0)

    // score owner (wave wbk, lane == row slot)
    float d_prev = 0.f, c_prev = 0.f, cum = 0.f;
    bool done = false;
    const int nk = 1 + p.sc.n_cost;
    const float csz[4] = {p.sc.cost_size[0], p.sc.cost_size[1], p.sc.cost_size[2], p.sc.cost_size[3]};
    const float ind_cap = p.sc.indicator ? 1.0f : __builtin_inff(), clipv = p.sc.reward_clip > 0.f ? p.sc.reward_clip : __builtin_inff();
    const __amdgpu_buffer_rsrc_t cost_rs = __builtin_amdgcn_make_buffer_rsrc(p.costs, 0, p.costs ? (uint32_t)(H * p.Bloc) : 0u, 0x00020000);

    // reward / cost / done bookkeeping of step T_ from the scorer terms in `part` (rows of the tile on the bookkeeping wave's
    // lanes); T_ = -1 only initialises d_prev / c_prev from s_0
#define CEM_PART_MIN4(K_) fminf(fminf(part[((K_) * 4 + 0) * 64 + lane], part[((K_) * 4 + 1) * 64 + lane]), \
                                fminf(part[((K_) * 4 + 2) * 64 + lane], part[((K_) * 4 + 3) * 64 + lane]))
#define CEM_BOOKKEEP(T_) do { if (w == wbk) { \
        const float dn = CEM_PART_MIN4(0); \
        float cn = 0.f; \
        _Pragma("unroll") for (int k = 1; k < CEM_NKIND; ++k) \
            if (k < nk) { const float dk = CEM_PART_MIN4(k); cn = cn + ((dk <= csz[k - 1]) ? 1.0f : 0.0f); } \
        cn = fminf(cn, ind_cap);                                   /* constrain_indicator: cost > 0 -> 1 (cn is a count) */ \
        if ((T_) >= 0) { \
            const bool ga = d_prev <= p.sc.goal_thresh;                                   /* safety_gym.py:116 */ \
            float r = (d_prev - dn) * p.sc.reward_distance + (ga ? 1.0f : 0.0f) * p.sc.reward_goal; \
            r = fminf(fmaxf(r, -clipv), clipv);                        /* reward_clip (safety_gym.py:141); +inf: none */ \
            if (p.variant == 1) {                                                         /* safe_cem_mpc.py:86-93 */ \
                done = done || ga; \
                const float nd = done ? 0.0f : 1.0f; \
                const float cst = c_prev * nd; \
                if (p.costs && lane < td.cnt) __builtin_amdgcn_raw_buffer_store_b8((uint8_t)cst, cost_rs, td.row_base + lane, __builtin_amdgcn_readfirstlane((T_) * p.Bloc), 0); \
                cum = cum + r * nd; \
            } else {                                                                      /* mpc_policy.py:34-37 */ \
                const float nd = done ? 0.0f : 1.0f; \
                cum = cum + r * nd; \
                done = done || ga; \
            } } \
        d_prev = dn; c_prev = cn; } } while (0)

    // min over the 4 lane rows that hold different features of the same batch row, for TWO scorer kinds at once: one row swap
    // puts kind KA's partial minima into the even lane rows and kind KA+1's into the odd ones, one half swap finishes both (two
    // VALU swaps + two v_min for a pair of kinds, no LDS).  Lane rows 0 / 2 then hold kind KA, rows 1 / 3 kind KA + 1 (PAIRED)
    // and every row stores its kind's value for its batch row (rows q and q + 2 store the same word).
#define CEM_PAIR_MIN_STORE(KA, VA, VB, PAIRED, C_) do { \
        const auto r16_ = __builtin_amdgcn_permlane16_swap(__float_as_uint(VA), __float_as_uint(VB), false, false); \
        const uint32_t m16_ = __float_as_uint(fminf(__uint_as_float(r16_[0]), __uint_as_float(r16_[1]))); \
        const auto r32_ = __builtin_amdgcn_permlane32_swap(m16_, m16_, false, false); \
        part[(((KA) + ((PAIRED) ? (q & 1) : 0)) * 4 + w) * 64 + 16 * (C_) + j] = fminf(__uint_as_float(r32_[0]), __uint_as_float(r32_[1])); \
    } while (0)

    // scorer kinds beyond (goal, first cost kind) and the observe_goal_dist form of the goal kind: rare, kept out of the hot block
#define CEM_RARE_KINDS_AND_STORE() do { \
        if (p.sc.goal_mode) {                                 /* squeeze(relu(goal_dist)), safety_gym.py:172-174 */ \
            _Pragma("unroll") for (int c = 0; c < RC; ++c) pm[0][c] = __builtin_inff(); \
            _Pragma("unroll") for (int i = 0; i < NFW; ++i) { \
                const f4 selg = CEM_SEL0_ROW(tab_v + 256 * i); \
                _Pragma("unroll") for (int c = 0; c < RC; ++c) \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) pm[0][c] = fminf(pm[0][c], fmaxf(fmaxf(s[i][c][r], 0.f), selg[r])); } } \
        _Pragma("unroll") for (int c = 0; c < RC; ++c) CEM_PAIR_MIN_STORE(0, pm[0][c], pm[1][c], true, c); \
        if (nk > 2) {                                         /* vases + hazards + pillars + gremlins all constrained */ \
            float pk[3][RC]; \
            _Pragma("unroll") for (int k = 0; k < 3; ++k) _Pragma("unroll") for (int c = 0; c < RC; ++c) pk[k][c] = __builtin_inff(); \
            _Pragma("unroll") for (int i = 0; i < NFW; ++i) { \
                const int f0 = 16 * (w + 4 * i) + 4 * q; \
                _Pragma("unroll") for (int k = 2; k < CEM_NKIND; ++k) if (k < nk) { \
                    const f4 selk = *reinterpret_cast<const f4 *>(p.kind_sel + k * CEM_U + f0); \
                    _Pragma("unroll") for (int c = 0; c < RC; ++c) \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) { \
                            const float lid = fminf(fmaxf(p.sc.D - p.sc.D * (1.0f - s[i][c][r]), 0.f), p.sc.D); \
                            pk[k - 2][c] = fminf(pk[k - 2][c], fmaxf(lid, selk[r])); } } } \
            _Pragma("unroll") for (int c = 0; c < RC; ++c) { \
                CEM_PAIR_MIN_STORE(2, pk[0][c], pk[1][c], true, c); \
                if (nk > 4) CEM_PAIR_MIN_STORE(4, pk[2][c], pk[2][c], false, c); } } } while (0)

    f4 hB[CEM_NG][RC];
    if (resumed) {
        // the tile's state as its previous segment left it: the wave's own layer-0 input blocks go back into registers and
        // into the LDS exchange buffer (the barrier inside the first stage publishes them), the bookkeeping wave takes its state back
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const f4 x = cem_ld_coherent(seg_rs, (((NFW + i) * RC + c) * 256 + tid) * 16);
                hB[i][c] = x;
                *reinterpret_cast<f4 *>(smem + ((c * CEM_NG + w + 4 * i) * 64 + lane) * 16) = x;
            }
        xw = XB;
        if (w == wbk) {
            const f4 b = cem_ld_coherent(seg_rs, (2 * NFW * RC * 256 + lane) * 16);
            d_prev = b[0]; c_prev = b[1]; cum = b[2]; done = b[3] != 0.f;
        }
    } else {
        // ---- prologue: the scaled input of step 0, x_0 = scale(concat[s_0, a_0]) (transition_model.py:70-72,79-87), and the
        //      scorer terms of s_0 (d_prev / c_prev of the first reward, safety_gym.py:62-66).  No network evaluation.
        float pm[2][RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) { pm[0][c] = __builtin_inff(); pm[1][c] = __builtin_inff(); }
#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int tv = tab_v + 256 * i;
            const f4 mn4 = CEM_TAB(CEM_ET_NMIN, tv), rd4 = CEM_TAB(CEM_ET_RDELTA, tv);
            const f4 isact4 = CEM_TAB(CEM_ET_ACT, tv);
            const f4 sel0 = CEM_TAB(CEM_ET_SEL0, tv), sel1 = CEM_TAB(CEM_ET_SEL1, tv);
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
                const f4 x = cem_sub4(__builtin_elementwise_fma(isact4, act4, sn), mn4) * rd4;      // s is 0 off the observation features
                *reinterpret_cast<f4 *>(smem + ((c * CEM_NG + w + 4 * i) * 64 + lane) * 16) = x;
                hB[i][c] = x;
            }
        }
        CEM_RARE_KINDS_AND_STORE();
        xw = XB;
    }
    f4 nb0 = cem_ld_tab(et_rs, bias_v, CEM_ET_ROWS * 512);                              // layer-0 bias, own blocks 2w, 2w+1
    f4 nb1 = cem_ld_tab(et_rs, bias_v + 64, CEM_ET_ROWS * 512);
#ifdef CEM_STAMPS
    long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long tprev_ = (long long)__builtin_amdgcn_s_memtime();
    st_[5] = (long long)(unsigned)__builtin_amdgcn_s_getreg(0xF804);      // HW_REG_HW_ID: which CU runs this tile
    st_[6] = (long long)(unsigned)__builtin_amdgcn_s_getreg(0xF814);      // HW_REG_XCC_ID
    st_[7] = tprev_;
#endif

    // Issue priority rotates with the step.  Tiles that share a CU's SIMDs are otherwise served oldest-wave-first: the oldest tile
    // runs nearly as if alone, the others advance in its stalls and reach their barriers wave by wave (three co-started tiles used to
    // finish at 754 K / 964 K / 1100 K cycles).  With each tile at level (t + dispatch round) mod 3 for step t, every tile gets
    // steps in which all four of its waves are preferred on all four SIMDs at once, and the CU's tiles advance together:
    // B2 0.382 -> 0.367 ms, B4 2.49 -> 2.45 ms, B3 / B5 rank +1 % (profiles/r03_ab_priority_rotation.txt; per-stage rotation,
    // two levels, MFMA-phase-high and epilogue-high were measured too and gain less or lose).  Floating tiles keep level 3.
    const bool prio_rot = !(SEG && tile_idx >= p.n_pinned);
    const int prio_r0 = (tile_idx >> 8) % 3;
    for (int t = t_begin; t < t_end; ++t) {
#ifndef CEM_NO_PRIO_ROTATION
        if (prio_rot) {
            const int lvl = (t + prio_r0) % 3;
            if (lvl == 0) __builtin_amdgcn_s_setprio(0); else if (lvl == 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(2);
        }
#endif
        // ---- dense layers: h = relu(h W + b)  (mlp_ensemble.py:18-22).  The accumulators start at the bias
        // (x W + b with b added first: same sum, one rounding order apart); the bias of the NEXT layer is requested a
        // whole stage ahead of its use.  Layer 0 is peeled out of the loop: with both stage shapes inside one runtime
        // loop the compiler merges their weight-ring registers at the join with moves behind an s_waitcnt vmcnt(0),
        // i.e. drains the prefetch ring once per step.
#define CEM_RELU_PUBLISH() do { \
            _Pragma("unroll") for (int c = 0; c < RC; ++c) { \
                f4 h0 = acc0[c], h1 = acc1[c]; \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) { h0[r] = fmaxf(h0[r], 0.f); h1[r] = fmaxf(h1[r], 0.f); } \
                *reinterpret_cast<f4 *>(smem + xw + ((c * CEM_NG + 2 * w) * 64 + lane) * 16) = h0; \
                *reinterpret_cast<f4 *>(smem + xw + ((c * CEM_NG + 2 * w + 1) * 64 + lane) * 16) = h1; \
                hB[0][c] = h0; hB[1][c] = h1;         /* own blocks of the next stage: no LDS round trip */ \
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
            // stage input = previous stage's output buffer = xw ^ XB (the previous stage toggled xw after writing)
            cem_mfma_stage<RC, 4 * NFW, NFW, true, CEM_X_EXCHANGE, LA>(acc0, acc1, hB, wq, smem, xw ^ XB, lane, w);
            if (!(resumed && t == t_begin)) CEM_BOOKKEEP(t - 1);   // the barrier inside the stage published step t-1's scorer terms (a resumed segment took them from seg_state)
            CEM_RELU_PUBLISH();
            CEM_STAMP(0);
        }
        for (int l = 1; l < p.L; ++l) {
            f4 acc0[RC], acc1[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) { acc0[c] = nb0; acc1[c] = nb1; }
            CEM_NEXT_BIAS(l + 1 < p.L ? l + 1 : 0);
            cem_mfma_stage<RC, CEM_NG, 2, false, CEM_X_EXCHANGE, LA>(acc0, acc1, hB, wq, smem, xw ^ XB, lane, w);
            CEM_RELU_PUBLISH();
            CEM_STAMP(1);
        }
#undef CEM_RELU_PUBLISH
#undef CEM_NEXT_BIAS

        // ---- heads (mlp_ensemble.py:33-34,189-193), state update (transition_model.py:75), scorer partials
        //      (safety_gym.py:188-192) and the next scaled input (transition_model.py:70-72,79-87).
        // A lone wave runs this between the MFMA stages, so it is written as ONE branch-free basic block of independent element
        // chains, FOUR elements (a lane's feature quad) at a time: the multiplies / adds / fmas compile to packed
        // v_pk_{mul,add,fma}_f32, which beside fp32 MFMAs cost what one scalar VALU instruction costs (2 elements each).
        // Per-lane predicates come from per-feature float tables (rows of etab), not from lane-mask SGPR pairs.
        float pm[2][RC];
#pragma unroll
        for (int c = 0; c < RC; ++c) { pm[0][c] = __builtin_inff(); pm[1][c] = __builtin_inff(); }
        const int tn = (t + 1 < H) ? t + 1 : H - 1;
        f4 xown[NFW][RC];

#pragma unroll
        for (int i = 0; i < NFW; ++i) {
            const int Fo = w + 4 * i;                  // < 4*NFW: every such block is an input block (zero padded)
            const int tv = tab_v + 256 * i;
            // from memory only the next step's action is requested BEFORE the MFMA stage; the table rows come out of LDS at the point
            // of use, and the step's model noise is drawn AFTER the stage's MFMAs are issued: Philox and Box-Muller do not depend on
            // them, so they fill the matrix pipe's drain instead of standing in front of it
            f4 act4[RC], eps4[RC];
#pragma unroll
            for (int c = 0; c < RC; ++c) CEM_LOAD_ACT(act4[c], i, c, tn);
            f4 accm[RC], accv[RC];
            {
                const f4 bm = CEM_TAB(CEM_ET_BMU, tv), bv = CEM_TAB(CEM_ET_BVAR, tv);
#pragma unroll
                for (int c = 0; c < RC; ++c) { accm[c] = bm; accv[c] = bv; }
            }
            CEM_STAMP(2);
            // the first heads stage also performs the exchange of the last hidden layer's output
            if (Fo < p.KB_obs) {                                                           // wave-uniform
                if (i == 0) cem_mfma_stage<RC, CEM_NG, 2, false, CEM_X_EXCHANGE, LA>(accm, accv, hB, wq, smem, xw ^ XB, lane, w);
                else cem_mfma_stage<RC, CEM_NG, 2, false, CEM_X_REREAD, LA>(accm, accv, hB, wq, smem, xw ^ XB, lane, w);
            } else if (i == 0) {
                __syncthreads();                      // keep the barrier count of waves without observation features
            }
            CEM_STAMP(3);
#pragma unroll
            for (int c = 0; c < RC; ++c) {
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
            const f4 mn4 = CEM_TAB(CEM_ET_NMIN, tv), rd4 = CEM_TAB(CEM_ET_RDELTA, tv);
            const f4 om4 = CEM_TAB(CEM_ET_OBS, tv), isact4 = CEM_TAB(CEM_ET_ACT, tv);
            const f4 sel0 = CEM_TAB(CEM_ET_SEL0, tv), sel1 = CEM_TAB(CEM_ET_SEL1, tv);

#pragma unroll
            for (int c = 0; c < RC; ++c) {
                const f4 mu = accm[c];
                const f4 var = cem_softplus4(accv[c]) + 1e-4f;
                f4 sd;
#pragma unroll
                for (int r = 0; r < 4; ++r) sd[r] = __builtin_amdgcn_sqrtf(var[r]);
                const f4 d = mu + sd * eps4[c];                                  // Normal.sample = loc + scale*eps
                const f4 sn = s[i][c] + d * om4;                                 // s_t += d_s_t on observation features
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
                // closest_distance terms of these features, folded into the kinds they belong to
                cem_scorer_terms(sn, p.sc.D, sel0, sel1, pm[0][c], pm[1][c]);
                // next scaled input x = (concat[s, a] - min) * (1/delta); padding features have min 0, 1/delta 1, value 0
                const f4 x = cem_sub4(__builtin_elementwise_fma(isact4, act4[c], sn), mn4) * rd4;   // s is 0 off the observation features
                *reinterpret_cast<f4 *>(smem + xw + ((c * CEM_NG + Fo) * 64 + lane) * 16) = x;
                xown[i][c] = x;
            }
        }
        // the wave's own input blocks of the next layer-0 stage stay in registers
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int c = 0; c < RC; ++c) hB[i][c] = xown[i][c];
        CEM_RARE_KINDS_AND_STORE();
        xw ^= XB;
        CEM_STAMP(4);
    }
    // the last step's scorer terms: publish, then its bookkeeping
    __syncthreads();
    CEM_BOOKKEEP(t_end - 1);
    if (!SEG || t_end == H) {
        if (w == wbk && lane < td.cnt) p.ret[td.row_base + lane] = cum;
    } else {
#pragma unroll
        for (int i = 0; i < NFW; ++i)
#pragma unroll
            for (int c = 0; c < RC; ++c) {
                cem_st_coherent(seg_rs, ((i * RC + c) * 256 + tid) * 16, s[i][c]);
                cem_st_coherent(seg_rs, (((NFW + i) * RC + c) * 256 + tid) * 16, hB[i][c]);
            }
        if (w == wbk) cem_st_coherent(seg_rs, (2 * NFW * RC * 256 + lane) * 16, (f4){d_prev, c_prev, cum, done ? 1.0f : 0.0f});
    }
#ifdef CEM_STAMPS
    if (p.stamps && lane == 0) for (int i = 0; i < 8; ++i) p.stamps[((size_t)tile_idx * 4 + w) * 8 + i] = st_[i];
#endif
}
#undef CEM_LOAD_ACT
#undef CEM_TAB
#undef CEM_SEL0_ROW
#define CEM_SEL0_ROW(TV_) cem_ld_tab(et_rs, (TV_), CEM_ET_SEL0 * 512)
// CEM_BOOKKEEP, CEM_PART_MIN4, CEM_PAIR_MIN_STORE and CEM_RARE_KINDS_AND_STORE stay defined: cem_rollout_wide.h uses them with the
// same local names (RC = 1) and undefines them.

template <int RC, int NFW, int MODE>
__global__ __launch_bounds__(256) void cem_rollout_kernel(const RolloutParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (p.check_done && p.ctrl->done) return;
    cem_tile_sample_actions<false>(p, (int)blockIdx.x, 0, p.H, true, MODE == 1);
    cem_rollout_tile<RC, NFW, MODE, false>(p, smem, (int)blockIdx.x, 0, p.H);
}

// Pinned tiles + floating segments.  A tile is 16*RC rows for the WHOLE horizon, so a launch whose tile count is not a multiple
// of the CU count leaves CUs idle while the busiest one finishes (B2: 625 tiles on 256 CUs = 3 on some, 2 on the others: 19 %).
// Here the first n_pinned tiles (a multiple of the CU count: the same number on every CU) run as before, one workgroup each
// for the whole horizon.  The remaining "floating" tiles are cut into horizon segments, one workgroup per (tile, segment)
// item: every floating workgroup draws a ticket; tickets below n_float are the floaters' first segments, a later ticket takes
// the next entry of a FIFO of floaters whose previous segment has finished (waiting for the entry to be written if need be).
// The hardware starts a floating workgroup in whatever slot is free, so a floater's segments visit the CUs whose spare slot
// has been idle longest, and every CU ends up carrying the same share of the floaters' work.  Floating workgroups run at
// raised issue priority: a floater is a 30-step dependent chain that always shares its CU with the pinned tiles, and would
// otherwise finish last.  State crosses segments through p.seg_state; results are bit-identical to the plain launch.
// Deadlock-free for any residency: pinned workgroups wait for nothing; FIFO entry e is written when the e-th non-final
// floating item completes; if every resident floating workgroup were waiting, all drawn tickets below the FIFO tail would be
// complete and the number of completed final segments would equal n_float — then the tail is the item count and nobody waits.
#define CEM_SEG_SPIN_LIMIT (1u << 23)
template <int RC, int NFW>
__global__ __launch_bounds__(256) void cem_rollout_seg_kernel(const RolloutParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ uint32_t item_s;
    if (p.check_done && p.ctrl->done) return;
    if ((int)blockIdx.x < p.n_pinned) {
        cem_tile_sample_actions<false>(p, (int)blockIdx.x, 0, p.H, true, false);
        cem_rollout_tile<RC, NFW, 0, true>(p, smem, (int)blockIdx.x, 0, p.H);
        return;
    }
#ifndef CEM_FLOAT_PRIO
#define CEM_FLOAT_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(CEM_FLOAT_PRIO);
    const uint32_t n_float = (uint32_t)(p.n_tiles - p.n_pinned);
    if (threadIdx.x == 0) {
        const uint32_t ticket = atomicAdd(p.seg_queue, 1u);
        uint32_t item = (((uint32_t)p.n_pinned + ticket) << 8);      // (tile << 8) | segment
        if (ticket >= n_float) {
            const uint32_t *slot = p.seg_flags + (ticket - n_float);
            uint32_t spins = 0, v;
            while ((v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins < CEM_SEG_SPIN_LIMIT)
                __builtin_amdgcn_s_sleep(16);
            item = v ? v - 1u : 0xffffffffu;               // never filled within the limit: give up rather than hang the device ...
            if (!v) atomicOr(const_cast<int32_t *>(&p.ctrl->fault), 1 /* CEM_FAULT_SEGMENT */);   // ... and say so: the host returns CEM_ERR_DEVICE for this plan
        }
        item_s = item;
    }
    __syncthreads();                                       // the state loads that follow are agent-scope atomic loads themselves
    const uint32_t item = item_s;
    if (item == 0xffffffffu) return;
    const int tile = (int)(item >> 8), seg = (int)(item & 255u);
    const int t0 = seg * p.seg_len, t1 = (t0 + p.seg_len < p.H) ? t0 + p.seg_len : p.H;
    // every segment samples the steps IT reads (the epilogue of step t fetches the action of step t + 1): a floating tile's segments
    // run on different CUs, and nothing sampled by one workgroup is read by another
    cem_tile_sample_actions<false>(p, tile, t0, t1 < p.H ? t1 + 1 : p.H, seg == 0, false);
    cem_rollout_tile<RC, NFW, 0, true>(p, smem, tile, t0, t1);
    if (t1 == p.H) {
        // the launch's last floating tile leaves the work queue as the next launch needs it (all items have run by then: every ticket
        // is drawn, every FIFO entry written and read), so no other kernel has to reset it between iterations
        if (threadIdx.x == 0) item_s = atomicAdd(p.seg_queue + 2, 1u);
        __syncthreads();
        if (item_s == n_float - 1u) {
            const int n_ready = (int)n_float * (p.n_seg - 1);
            for (int i = (int)threadIdx.x; i < n_ready; i += 256) __hip_atomic_store(p.seg_flags + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 3) __hip_atomic_store(p.seg_queue + threadIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (t1 < p.H) {
        // EVERY wave waits for the acknowledgements of its own sc1 state stores (vmcnt(0); the encoding leaves expcnt / lgkmcnt
        // alone) before the barrier: s_barrier does not drain stores on gfx940+, and a workgroup-scope release fence compiles
        // to no wait at all here.  tests/test_capi_cpu.py checks the ISA for this wait between the last sc1 store and the barrier.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t pos = atomicAdd(p.seg_queue + 1, 1u);
            __hip_atomic_store(p.seg_flags + pos, (((uint32_t)tile << 8) | (uint32_t)(seg + 1)) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// small kernels of the optimiser loop
// ---------------------------------------------------------------------------------------------------------
struct InitParams { CtrlBlock *ctrl; const CtrlBlock *host_ctrl; float *musig; int32_t HA, A; float mu0[32], sigma0[32]; uint32_t *seg_queue, *seg_flags; int32_t n_ready; };

__global__ void cem_init_kernel(const InitParams p)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the plan's control block (Philox key, observation, cleared best-so-far / flags) as the host staged it in PINNED memory: read
    // from there directly — a copy node of its own in front of every plan cost a 3.6 us blit kernel plus its gap
    if (p.host_ctrl && i < (int)(sizeof(CtrlBlock) / 4)) reinterpret_cast<uint32_t *>(p.ctrl)[i] = reinterpret_cast<const uint32_t *>(p.host_ctrl)[i];
    // the rollout launches' work queue starts every plan empty (a launch resets it itself; this covers a plan that ended in a fault)
    if (p.seg_queue) {
        if (i < 3) p.seg_queue[i] = 0u;
        for (int e = i; e < p.n_ready; e += gridDim.x * blockDim.x) p.seg_flags[e] = 0u;
    }
    if (i < p.HA) { p.musig[i] = p.mu0[i % p.A]; p.musig[p.HA + i] = p.sigma0[i % p.A]; }   // cem_mpc.py:39-40
    if (!p.host_ctrl) {                                       // (staged by the host along with the rest of the block otherwise: stage_ctrl)
        if (i < 32) p.ctrl->best[i] = 0.f;                                                     // cem_mpc.py:41
        if (i == 0) { p.ctrl->best_score = -__builtin_inff(); p.ctrl->done = 0; p.ctrl->iters = 0; p.ctrl->fault = 0; }
    }
}

struct ReduceParams {
    const float *ret; const uint8_t *costs; float *scores; const CtrlBlock *ctrl;
    int32_t Nloc, P, H, variant, check_done;
    float alpha, beta, thr;
    uint32_t *zero; int32_t zero_n;        // words block 0 clears for the multi-workgroup select that follows (digit histograms + barrier counter), or null
};

// One block = 64 candidates (one per lane) x 16 waves.  The kernel is a latency chain — a few hundred bytes per candidate, one dependent
// round of loads, a barrier, a store — so what matters is how many round trips to L2 a wave makes, not bandwidth (round 5, measured at
// the shipped SafeCemMpc shape, P = 45, H = 8, where ten of the sixteen waves used to idle and the others made six trips each):
//   * horizons of 16 steps and more: wave w counts the particle costs of steps t = w, w + 16, ..., TWO steps per trip, up to 8 particles each;
//   * shorter horizons: the 16 waves share out (step, particle slice) pairs — 16 / H waves per step, each counting every (16 / H)-th
//     particle, up to 16 loads per trip — and add their counts in LDS (integers: exact, order-free);
//   * wave 0's return loads (the particle mean, summed in the reference's order q = 0 .. P-1) are requested before its cost loads.
// Counts are small integers, exact in the reference's fp32 sums as well.
#define CEM_REDUCE_THREADS 1024
__global__ __launch_bounds__(CEM_REDUCE_THREADS) void cem_reduce_kernel(const ReduceParams p)
{
    __shared__ int32_t unsafe_w[16][64];
    __shared__ uint32_t cnt_s[16][64];
    if (p.check_done && p.ctrl->done) return;
    if (p.zero && blockIdx.x == 0) for (int i = threadIdx.x; i < p.zero_n; i += CEM_REDUCE_THREADS) p.zero[i] = 0u;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + lane;
    const bool live = n < p.Nloc;
    const int nn = live ? n : p.Nloc - 1;
    const int P = p.P, H = p.H;
    // wave 0: the first 16 particles' returns, in flight while the costs are counted
    float r0[16];
    if (w == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) r0[j] = p.ret[(size_t)(j < P ? j : 0) * p.Nloc + nn];
    }
    int32_t unsafe = 0;
    if (p.variant == 1) {                                              // safe_cem_mpc.py:90-96,110-120
        const float denom = (p.alpha + p.beta) + (float)P;
        const size_t Bloc = (size_t)P * p.Nloc;
        if (H >= 16) {
            for (int t = w; t < H; t += 32) {
                const int t2 = t + 16 < H ? t + 16 : t;               // (clamped: the loads are unconditional, the second count is dropped)
                const uint8_t *ca = p.costs + (size_t)t * Bloc + nn, *cb = p.costs + (size_t)t2 * Bloc + nn;
                uint32_t cnta = 0, cntb = 0;
                for (int q = 0; q < P; q += 8) {
                    uint32_t va[8], vb[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const size_t o = (size_t)(q + j < P ? q + j : q) * p.Nloc; va[j] = ca[o]; vb[j] = cb[o]; }
#pragma unroll
                    for (int j = 0; j < 8; ++j) if (q + j < P) { cnta += va[j]; cntb += vb[j]; }
                }
                unsafe |= ((p.alpha + (float)cnta) / denom <= p.thr) ? 0 : 1;
                if (t + 16 < H) unsafe |= ((p.alpha + (float)cntb) / denom <= p.thr) ? 0 : 1;
            }
        } else {
            const int wpt = 16 / H;                                    // waves per step (>= 1), H * wpt <= 16 of the waves count
            cnt_s[w][lane] = 0u;
            __syncthreads();
            if (w < H * wpt) {
                const int t = w / wpt, part = w % wpt;
                const uint8_t *c = p.costs + (size_t)t * Bloc + nn;
                uint32_t cnt = 0;
                for (int q = part; q < P; q += 16 * wpt) {
                    uint32_t v[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) { const int qq = q + j * wpt; v[j] = c[(size_t)(qq < P ? qq : q) * p.Nloc]; }
#pragma unroll
                    for (int j = 0; j < 16; ++j) if (q + j * wpt < P) cnt += v[j];
                }
                if (wpt > 1) atomicAdd(&cnt_s[t][lane], cnt); else cnt_s[t][lane] = cnt;
            }
            __syncthreads();
            if (w < H) unsafe = ((p.alpha + (float)cnt_s[w][lane]) / denom <= p.thr) ? 0 : 1;
        }
        unsafe_w[w][lane] = unsafe;
    }
    float sum = 0.f;
    if (w == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j) if (j < P) sum = sum + r0[j];
        for (int q = 16; q < P; q += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = p.ret[(size_t)(q + j < P ? q + j : q) * p.Nloc + nn];
#pragma unroll
            for (int j = 0; j < 16; ++j) if (q + j < P) sum = sum + v[j];
        }
    }
    __syncthreads();
    if (w != 0 || !live) return;
    float score = sum / (float)P;                                      // reduce_mean over particles
    if (p.variant == 1) {
        int32_t u = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) u |= unsafe_w[i][lane];
        score = score - (u ? 1.0f : 0.0f) * 100.0f;
    }
    p.scores[n] = score;
}

// ---------------------------------------------------------------------------------------------------------
// the objective and the scorer as ops of their own, on tensors the caller holds (cem_compute_objective,
// cem_scorer_reward, cem_scorer_cost).  HBM-bound: a group of 16 lanes owns one row and reads its features
// 16 at a time (64-byte segments), min over the group with 4 shuffles; the arithmetic is the rollout epilogue's.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float cem_min16(float v)
{
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) v = fminf(v, __shfl_xor(v, d, 16));
    return v;
}

// closest_distance over obs[lo:hi) (safety_gym.py:188-192): min over bins of clip(D - D*(1-x), 0, D)
__device__ __forceinline__ float cem_closest16(const float *obs, int lo, int hi, float D, int j)
{
    float m = __builtin_inff();
    for (int f = lo + j; f < hi; f += 16) m = fminf(m, fminf(fmaxf(D - D * (1.0f - obs[f]), 0.f), D));
    return cem_min16(m);
}

__device__ __forceinline__ float cem_goal_dist16(const float *obs, const ScorerDev &sc, int j)
{
    if (sc.goal_mode) return fmaxf(obs[sc.goal_lo], 0.f);                       // squeeze(relu(goal_dist)), safety_gym.py:172-174
    return cem_closest16(obs, sc.goal_lo, sc.goal_hi, sc.D, j);
}

__device__ __forceinline__ float cem_cost16(const float *obs, const ScorerDev &sc, int j)
{
    float c = 0.f;
    for (int k = 0; k < sc.n_cost; ++k)                                          // safety_gym.py:148-163
        c = c + ((cem_closest16(obs, sc.cost_lo[k], sc.cost_hi[k], sc.D, j) <= sc.cost_size[k]) ? 1.0f : 0.0f);
    if (sc.indicator) c = c > 0.f ? 1.0f : 0.0f;                                 // :164-165
    return c;
}

__device__ __forceinline__ float cem_reward_of(float d, float dn, bool ga, const ScorerDev &sc)
{
    float r = (d - dn) * sc.reward_distance + (ga ? 1.0f : 0.0f) * sc.reward_goal;   // safety_gym.py:117-119
    if (sc.reward_clip > 0.f) r = fminf(fmaxf(r, -sc.reward_clip), sc.reward_clip);  // :140-142
    return r;
}

struct ObjectiveParams {
    const float *traj;           // [B][H+1][O]
    float *ret;                  // [B]
    uint8_t *costs;              // [H][B] (safe variant) or nullptr
    int32_t B, H, O, variant;
    ScorerDev sc;
};

__global__ __launch_bounds__(256) void cem_objective_kernel(const ObjectiveParams p)
{
    const int row = (int)((blockIdx.x * 256u + threadIdx.x) >> 4), j = threadIdx.x & 15;
    if (row >= p.B) return;                                   // whole 16-lane groups leave together
    const float *tr = p.traj + (size_t)row * (p.H + 1) * p.O;
    float d_prev = cem_goal_dist16(tr, p.sc, j);
    float c_prev = p.variant == 1 ? cem_cost16(tr, p.sc, j) : 0.f;
    float cum = 0.f;
    bool done = false;
    for (int t = 0; t < p.H; ++t) {
        const float *nx = tr + (size_t)(t + 1) * p.O;
        const float dn = cem_goal_dist16(nx, p.sc, j);
        const bool ga = d_prev <= p.sc.goal_thresh;                              // safety_gym.py:116
        const float r = cem_reward_of(d_prev, dn, ga, p.sc);
        if (p.variant == 1) {                                                    // safe_cem_mpc.py:86-93
            done = done || ga;
            const float nd = done ? 0.0f : 1.0f;
            if (p.costs && j == 0) p.costs[(size_t)t * p.B + row] = (uint8_t)(c_prev * nd);
            cum = cum + r * nd;
            c_prev = cem_cost16(nx, p.sc, j);
        } else {                                                                 // mpc_policy.py:34-37
            const float nd = done ? 0.0f : 1.0f;
            cum = cum + r * nd;
            done = done || ga;
        }
        d_prev = dn;
    }
    if (j == 0) p.ret[row] = cum;
}

struct ScorerOpParams {
    const float *obs, *next_obs; float *out; uint8_t *flag;
    int32_t n, O, what;          // what 0: reward + goal_achieved, 1: cost
    ScorerDev sc;
};

__global__ __launch_bounds__(256) void cem_scorer_kernel(const ScorerOpParams p)
{
    const int row = (int)((blockIdx.x * 256u + threadIdx.x) >> 4), j = threadIdx.x & 15;
    if (row >= p.n) return;
    const float *o = p.obs + (size_t)row * p.O;
    if (p.what == 1) {
        const float c = cem_cost16(o, p.sc, j);
        if (j == 0) p.out[row] = c;
        return;
    }
    const float d = cem_goal_dist16(o, p.sc, j), dn = cem_goal_dist16(p.next_obs + (size_t)row * p.O, p.sc, j);
    const bool ga = d <= p.sc.goal_thresh;
    const float r = cem_reward_of(d, dn, ga, p.sc);
    if (j == 0) { p.out[row] = r; if (p.flag) p.flag[row] = ga ? 1 : 0; }
}

__device__ __forceinline__ float cem_out_noise(const CtrlBlock *ctrl, const float *eps_out, const int a)
{
    if (eps_out) return eps_out[a];
    const f4 e = cem_normal4((uint32_t)(a >> 2), 0u, 0u, 0u, CEM_STREAM_OUT, cem_key(ctrl));
    return e[a & 3];
}


// The plan's result as the HOST reads it from pinned memory: words [0, A) action, [32] score, [33] iterations run, [34] early-stop flag,
// [35] fault bits, [36] the host's plan counter (CtrlBlock::seq), [37] the XOR of words 0..36 and a constant.  The host may poll for word
// 36 instead of synchronising the stream, and writes to host memory from the device are NOT ordered with one another on the way (a
// marker stored after the data — even behind a barrier and the stores' acknowledgements — was seen BEFORE the data: measured), so the
// block carries a checksum and the host accepts it only when counter and checksum both match.  One wave stores the block from an LDS copy.
#define CEM_RESULT_WORDS 38
#define CEM_RESULT_MAGIC 0x5EC0DE5Au
// checksum of the block: FNV-1a over its 36 data words, seeded with the plan counter — position dependent (a plain XOR lets two stale
// words cancel: iters 5 -> 4 together with done 0 -> 1), the same function on the host (cem_capi.hip result_landed)
__host__ __device__ inline uint32_t cem_result_checksum(const volatile uint32_t *w, const uint32_t seq)
{
    uint32_t x = CEM_RESULT_MAGIC ^ seq;
    for (int i = 0; i < 36; ++i) x = (x ^ w[i]) * 0x01000193u;
    return x ^ (x >> 15);
}
// `result`: the pinned host block; `result_dev`: the same 38 words in the workspace (cem_layout_t::result) for callers that stay on the device
__device__ __forceinline__ void cem_emit_result(float *result, uint32_t *result_dev, const uint32_t *res_l /* LDS [36] */, const uint32_t seq, const int lane)
{
    const uint32_t x = cem_result_checksum(res_l, seq);
    if (lane < CEM_RESULT_WORDS) {
        const uint32_t w = lane < 36 ? res_l[lane] : (lane == 36 ? seq : x);
        if (result_dev) result_dev[lane] = w;
        __hip_atomic_store(reinterpret_cast<uint32_t *>(result) + lane, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

struct SelectParams {
    const float *scores; const float *actions; float *musig; CtrlBlock *ctrl; int32_t *elite_idx;
    // the particle mean of the CemMpc objective folded into the key staging (single-rank whole plans; null: read `scores`):
    // score[i] = (sum over q = 0..P-1, in that order, of ret[q * N + i]) / P  (mpc_policy.py:38-39), also stored to scores_w[i]
    const float *ret; float *scores_w; int32_t P;
    // the plan's result written by the select itself (whole plans on this kernel: no final kernel; see FinalParams): null = not here
    float *result; uint32_t *result_dev; const float *eps_out; float noise_stddev;
    int32_t is_last;             // this is the plan's last iteration: with `result`, the completion marker (result[36] = ctrl->seq) follows the result
    int32_t N, k, HA, A, check_done;
    float smoothing, one_minus_smoothing, threshold;   // one_minus_smoothing = fl32(1.0 - smoothing) rounded once, as cem_mpc.py:64-65 does
    long long *stamps;           // [8] section stamps of -DCEM_STAMPS diagnostic builds
};

#ifdef CEM_STAMPS
#define CEM_SEL_STAMP(i) do { if (threadIdx.x == 0 && p.stamps) p.stamps[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define CEM_SEL_STAMP(i) do { } while (0)
#endif

#define CEM_SEL_KIDX(i) ((i) + ((i) >> 5))        // LDS word of key i in the one-workgroup select's staged key list (one pad word per 32 keys)
#define CEM_SEL_KWORDS(n) ((n) + ((n) >> 5) + 1)    // words that list takes
__device__ __forceinline__ uint32_t cem_f2key(float f)
{
    const uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0u;          // NaN sorts lowest (bit test: immune to -fno-honor-nans)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Histogram add for a whole wave when few bins are hot (the first counted radix pass: sign + exponent bits): LDS atomics
// on one address serialise, so up to four rounds of "leader's digit -> one add of the group size", plain adds for the rest.
__device__ __forceinline__ void cem_hist_add_clustered(uint32_t *hist, bool match, const uint32_t digit)
{
    const int lane = threadIdx.x & 63;
    uint64_t rem = __builtin_amdgcn_ballot_w64(match);
#pragma unroll 1
    for (int round = 0; round < 4 && rem; ++round) {
        const int leader = __builtin_ctzll(rem);
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
        const uint64_t same = __builtin_amdgcn_ballot_w64(match && digit == d);
        if (lane == leader) atomicAdd(&hist[d], (uint32_t)__builtin_popcountll(same));
        rem &= ~same;
        if ((same >> lane) & 1ull) match = false;
    }
    if (match) atomicAdd(&hist[digit], 1u);
}

__device__ __forceinline__ float cem_key2f(const uint32_t key)       // inverse of cem_f2key (a NaN comes back as a NaN)
{
    return __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
}

// exclusive scans of two values per thread over a 1024-thread block (one barrier): pa, pb = sums over lower thread ids
__device__ __forceinline__ void cem_block_excl_scan2(const uint32_t a, const uint32_t b, uint32_t (*wsum)[16], uint32_t &pa, uint32_t &pb)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t ia = a, ib = b;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t oa = __shfl_up(ia, d), ob = __shfl_up(ib, d);
        if (lane >= d) { ia += oa; ib += ob; }
    }
    if (lane == 63) { wsum[0][wv] = ia; wsum[1][wv] = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0;
    for (int i = 0; i < wv; ++i) { ba += wsum[0][i]; bb += wsum[1][i]; }
    pa = ba + ia - a; pb = bb + ib - b;
}

// block-wide (1024 threads): the bin b with ge[b] >= need > ge[b + 1], ge[b] = #keys in bins >= b; returns (b, need - ge[b + 1])
// COHERENT: the histogram was written by other workgroups of the SAME kernel (atomics at the device coherence point): read it there too
template <bool COHERENT = false>
__device__ __forceinline__ void cem_ms_find(const uint32_t *h, const int nbins, const uint32_t need, uint32_t *sh /* [20] */, uint32_t &bin, uint32_t &need_next)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int b0 = 2 * tid, b1 = 2 * tid + 1;
    uint32_t h0 = 0u, h1 = 0u;
    if (COHERENT) {
        if (b0 < nbins) h0 = __hip_atomic_load(h + b0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b1 < nbins) h1 = __hip_atomic_load(h + b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { h0 = b0 < nbins ? h[b0] : 0u; h1 = b1 < nbins ? h[b1] : 0u; }
    const uint32_t tot = h0 + h1;
    uint32_t inc = tot;                                   // inclusive prefix over lower thread ids
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(inc, d); if (lane >= d) inc += o; }
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    uint32_t before = 0, all = 0;
    for (int i = 0; i < 16; ++i) { const uint32_t v = sh[i]; all += v; if (i < wv) before += v; }
    const uint32_t above = all - (before + inc);          // keys in bins of higher thread ids
    const uint32_t ge1 = h1 + above, ge0 = tot + above;
    if (ge1 >= need && above < need) { sh[16] = (uint32_t)b1; sh[17] = need - above; }
    if (ge0 >= need && ge1 < need) { sh[16] = (uint32_t)b0; sh[17] = need - ge1; }
    __syncthreads();
    bin = sh[16]; need_next = sh[17];
    __syncthreads();
}

// top_k + best-so-far + moments + smoothing + early stop, one 1024-thread workgroup  (cem_mpc.py:56-67).
// One CU, so the kernel is a latency chain: the scores are staged in LDS once (CACHE; they are read by 4 radix passes, the
// compaction and the best-of-elite), the 256-bin suffix scan of a pass is done by one wave with shuffles (2 barriers per
// pass), gathers are issued in batches, and the two serial tails run on different waves.  All sums keep a fixed order.
// CROWDED: the scores have a crowd — SafeCemMpc, where most candidates are unsafe and sit near -100 (the host picks the instantiation by the
// objective; a speed choice only: both give the same result on any scores).  It is a template parameter because the extra paths, though never
// taken on CemMpc's spread-out scores, cost that case 0.5 us through code layout and register allocation alone (round 5, scripts/stamp_select.py).
template <bool CACHE, bool CROWDED = false>
__global__ __launch_bounds__(1024) void cem_select_kernel(const SelectParams p)
{
    extern __shared__ __attribute__((aligned(16))) char sel_smem[];
    __shared__ __attribute__((aligned(16))) uint32_t hist[2][256];
    __shared__ uint32_t wsum[2][16];
    __shared__ uint32_t sh_prefix, sh_need, sh_cnt;
    __shared__ uint32_t sh20[20];
    __shared__ __attribute__((aligned(16))) float red[4096];   // per-thread partial sums (float4 in the wide moments path)
    __shared__ float bsc[16];
    __shared__ int bpos[16];
    if (p.check_done && p.ctrl->done) return;

    const int tid = threadIdx.x;
    const int N = p.N, k = p.k, HA = p.HA;
    CEM_SEL_STAMP(0);
    int32_t *elite = reinterpret_cast<int32_t *>(sel_smem);                 // [k]
    float *colmean = reinterpret_cast<float *>(sel_smem + (size_t)((k + 3) & ~3) * 4);   // [HA]
    float *newsig = colmean + HA;                                           // [HA] smoothed sigma
    // [N + N / 32 + 1] order-preserving keys of the scores (CACHE), key i at word CEM_SEL_KIDX(i) = i + i / 32: the compaction walks a
    // thread's own run of C = N / 1024 consecutive keys, i.e. the lanes of a wave read words C apart — with C = 16 (N = 16 000) two LDS banks
    // for all 64 lanes; one pad word per 32 keys spreads them over all banks (compaction 8.4 -> 2 us at N = 16 000)
    uint32_t *ckey = reinterpret_cast<uint32_t *>(newsig + HA);

    // old mu / sigma of the columns this thread will finish (first column block): requested now, needed at the very end
    float old_mu = 0.f, old_sg = 0.f;
    if (tid < HA) { old_mu = p.musig[tid]; old_sg = p.musig[HA + tid]; }

    // stage the keys; their block-wide min / max tell which leading bytes every key shares (scores of one iteration
    // usually share sign and exponent): those radix passes have nothing to count
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    if (CACHE) {
        for (int i0 = 0; i0 < N; i0 += 4096) {
            float v[4];
            if (p.ret) {
                // cem_reduce_kernel's sum for this thread's four candidates: particles in ascending order, eight (clamped, hence
                // unconditional) loads per candidate in flight; reduce_mean = sum / P
                float acc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int q0 = 0; q0 < p.P; q0 += 8) {
                    float r[8][4];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (i0 + j * 1024 < N) {                              // (block-uniform: a slice of 1024 candidates that exists)
                            const int i = i0 + j * 1024 + tid, ic = i < N ? i : N - 1;
#pragma unroll
                            for (int u = 0; u < 8; ++u) r[u][j] = p.ret[(size_t)(q0 + u < p.P ? q0 + u : p.P - 1) * N + ic];
                        }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (q0 + u < p.P) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) if (i0 + j * 1024 < N) acc[j] = acc[j] + r[u][j];
                        }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int i = i0 + j * 1024 + tid; v[j] = acc[j] / (float)p.P; if (i < N) p.scores_w[i] = v[j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int i = i0 + j * 1024 + tid; if (i < N) v[j] = p.scores[i]; }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = i0 + j * 1024 + tid;
                if (i < N) { const uint32_t key = cem_f2key(v[j]); ckey[CEM_SEL_KIDX(i)] = key; kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t a = (uint32_t)__shfl_xor((int)kmin, d), b = (uint32_t)__shfl_xor((int)kmax, d);
            kmin = a < kmin ? a : kmin; kmax = b > kmax ? b : kmax;
        }
        if ((tid & 63) == 0) { wsum[0][tid >> 6] = kmin; wsum[1][tid >> 6] = kmax; }
    }
    auto K = [&](const int i) { return CACHE ? ckey[CEM_SEL_KIDX(i)] : cem_f2key(p.scores[i]); };
    if (tid < 256) hist[1][tid] = 0;                  // pass 3 counts into hist[3 & 1]
    __syncthreads();
    uint32_t kdiff = 0xFFFFFFFFu;
    if (CACHE) {
        for (int i = 0; i < 16; ++i) { kmin = wsum[0][i] < kmin ? wsum[0][i] : kmin; kmax = wsum[1][i] > kmax ? wsum[1][i] : kmax; }
        kdiff = kmin ^ kmax;
        __syncthreads();                              // wsum is reused by the compaction scan
    }
    CEM_SEL_STAMP(1);

    // ---- the k-th largest key -------------------------------------------------------------------------------
    uint32_t prefix = 0, mask = 0, need = (uint32_t)k;
    bool solved = false;
    if (CACHE) {
        // Bucket select on the range the keys actually span (round 4).  The scores of an iteration sit in a narrow band (same sign,
        // an exponent or two), so 2048 buckets of [kmin, kmax] — bucket = (key - base) >> sh, monotone in the key — usually leave a
        // handful of keys in the k-th key's bucket; those are ranked against each other directly (key descending, candidate index
        // ascending: tf.nn.top_k's tie order).  A bucket that still holds more than 256 keys (SafeCemMpc: hundreds of scores crowd
        // around -100) is split again, 11 bits finer, until it is small or one key wide (sh = 0: all its keys are equal).  At most
        // three levels; typically ONE atomic pass + one block-wide scan + a rank over a few keys, where the byte-wide radix took
        // three passes of an atomic pass, a one-wave scan and three barriers each.  Same T, same `need`: the compaction is unchanged.
        uint32_t *h2k = reinterpret_cast<uint32_t *>(red);               // [2048] bucket counts   (the moments' scratch: not in use yet)
        uint32_t *lkey = h2k + 2048, *lidx = lkey + 1024;                // the k-th key's bucket: keys, candidate indices
        uint32_t base = kmin, window = kmax - kmin;                      // keys in play: base <= key <= base + window
        int sh = window ? max(0, 21 - (int)__clz((int)window)) : 0;      // window >> sh < 2048
#pragma unroll 1
        for (int level = 0; level < 4; ++level) {
            h2k[tid] = 0u; h2k[tid + 1024] = 0u;
            if (tid == 0) sh_cnt = 0u;
            __syncthreads();
            if (CROWDED) {
                // SafeCemMpc: most candidates are unsafe and their scores — return minus 100 — share ONE bucket.  Whole waves enter and the bucket
                // of a wave's first key is added once for all the lanes that share it (one round: -0.7 us at B2; a second round gains nothing,
                // and on CemMpc's spread-out scores the round costs 0.4 us for nothing)
                for (int i0 = 0; i0 < N; i0 += 1024) {
                    const int i = i0 + tid;
                    const uint32_t off = i < N ? ckey[CEM_SEL_KIDX(i)] - base : 0xFFFFFFFFu;
                    bool match = i < N && off <= window;
                    const uint32_t digit = off >> sh;
                    const uint64_t rem = __builtin_amdgcn_ballot_w64(match);
                    if (rem) {
                        const int leader = __builtin_ctzll(rem);
                        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)digit, leader);
                        const uint64_t same = __builtin_amdgcn_ballot_w64(match && digit == d);
                        if ((tid & 63) == leader) atomicAdd(&h2k[d], (uint32_t)__builtin_popcountll(same));
                        if (digit == d) match = false;
                    }
                    if (match) atomicAdd(&h2k[digit], 1u);
                }
            } else {
                for (int i = tid; i < N; i += 1024) {
                    const uint32_t off = ckey[CEM_SEL_KIDX(i)] - base;      // (a key below base wraps to a huge offset: out of the window)
                    if (off <= window) atomicAdd(&h2k[off >> sh], 1u);
                }
            }
            __syncthreads();
            uint32_t bucket, need1;
            cem_ms_find<false>(h2k, 2048, need, sh20, bucket, need1);   // ge[bucket] >= need > ge[bucket + 1]; need1 = need - ge[bucket + 1]
            const uint32_t m = h2k[bucket];
            if (sh == 0) { prefix = base + bucket; need = need1; solved = true; break; }     // the bucket is ONE key: `need1` of its ties are taken
            if (m <= (CROWDED ? 32u : 256u)) {                           // (workgroup-uniform) a handful of keys — the usual CemMpc case: m threads rank them, one barrier fewer
                for (int i = tid; i < N; i += 1024) {
                    const uint32_t key = ckey[CEM_SEL_KIDX(i)], off = key - base;
                    if (off <= window && (off >> sh) == bucket) { const uint32_t pos = atomicAdd(&sh_cnt, 1u); lkey[pos] = key; lidx[pos] = (uint32_t)i; }
                }
                __syncthreads();
                if ((uint32_t)tid < m) {
                    const uint32_t kj = lkey[tid], ij = lidx[tid];
                    uint32_t gt = 0, before = 0;
                    for (uint32_t l = 0; l < m; ++l) { const uint32_t kl = lkey[l]; gt += kl > kj; before += (kl > kj) || (kl == kj && lidx[l] < ij); }
                    if (before == need1 - 1u) { sh_prefix = kj; sh_need = need1 - gt; }     // the k-th largest; ties at its key still to take
                }
                __syncthreads();
                prefix = sh_prefix; need = sh_need; solved = true;
                break;
            }
            if (CROWDED && m <= 256u) {
                // Collect the bucket's keys (a wave takes its slots with ONE returning atomic, lanes place themselves behind it), then rank
                // them against each other with the whole workgroup: key j is compared with a quarter of the list by each of four threads and the
                // partial counts meet in LDS (integers: order-free).  Round 5: with `m` threads walking all m keys alone, a SafeCemMpc
                // iteration — 170 keys in the k-th key's bucket — spent 15 of the select's 27 us here (CemMpc: m is 1-5).
                uint32_t *gt_s = &hist[0][0], *bf_s = &hist[1][0];       // [256] each (the radix passes below do not run once this path solves)
                if (tid < 512) (&hist[0][0])[tid] = 0u;
                for (int i0 = 0; i0 < N; i0 += 1024) {
                    const int i = i0 + tid;
                    const uint32_t key = i < N ? ckey[CEM_SEL_KIDX(i)] : 0u, off = key - base;
                    const bool hit = i < N && off <= window && (off >> sh) == bucket;
                    const uint64_t hits = __builtin_amdgcn_ballot_w64(hit);
                    if (hits) {                                          // (wave-uniform)
                        const int lane = tid & 63, leader = __builtin_ctzll(hits);
                        uint32_t first = 0u;
                        if (lane == leader) first = atomicAdd(&sh_cnt, (uint32_t)__builtin_popcountll(hits));
                        first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
                        if (hit) { const uint32_t pos = first + (uint32_t)__builtin_popcountll(hits & ((1ull << lane) - 1ull)); lkey[pos] = key; lidx[pos] = (uint32_t)i; }
                    }
                }
                __syncthreads();
                {
                    const uint32_t j = (uint32_t)tid & 255u, qtr = (uint32_t)tid >> 8, span = (m + 3u) >> 2;
                    if (j < m) {
                        const uint32_t kj = lkey[j], ij = lidx[j];
                        const uint32_t lo = qtr * span, hi = lo + span < m ? lo + span : m;
                        uint32_t gt = 0, before = 0;
                        for (uint32_t l = lo; l < hi; l += 8) {           // eight (clamped, hence unconditional) LDS reads in flight: one latency per eight keys
                            uint32_t kl[8], il[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) { const uint32_t lu = l + u < hi ? l + u : hi - 1u; kl[u] = lkey[lu]; il[u] = lidx[lu]; }
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (l + u < hi) { gt += kl[u] > kj; before += (kl[u] > kj) || (kl[u] == kj && il[u] < ij); }
                        }
                        if (gt) atomicAdd(&gt_s[j], gt);
                        if (before) atomicAdd(&bf_s[j], before);
                    }
                }
                __syncthreads();
                if ((uint32_t)tid < m && bf_s[tid] == need1 - 1u) { sh_prefix = lkey[tid]; sh_need = need1 - gt_s[tid]; }     // the k-th largest; ties at its key still to take
                __syncthreads();
                prefix = sh_prefix; need = sh_need; solved = true;
                break;
            }
            base += bucket << sh; window = (1u << sh) - 1u; need = need1;    // split that bucket, 11 bits finer
            sh = sh > 11 ? sh - 11 : 0;
            __syncthreads();                                             // (h2k / sh20 are rewritten by the next level)
        }
    }
    bool leading = true;
    for (int pass = solved ? -1 : 3; pass >= 0; --pass) {
        const int hb = pass & 1;
        if (leading && ((kdiff >> (8 * pass)) & 255u) == 0u) {             // every key has this byte (workgroup-uniform test)
            prefix |= kmin & (0xFFu << (8 * pass)); mask |= 0xFFu << (8 * pass);
            if (tid < 256) hist[hb ^ 1][tid] = 0;
            __syncthreads();
            continue;
        }
        if (leading) {                                 // first counted pass: few hot bins; whole waves enter (wave-level add)
            for (int i0 = 0; i0 < N; i0 += 1024) {
                const int i = i0 + tid;
                const uint32_t key = i < N ? K(i) : 0u;
                cem_hist_add_clustered(hist[hb], i < N && (key & mask) == prefix, (key >> (8 * pass)) & 255u);
            }
        } else {
            for (int i = tid; i < N; i += 1024) {
                const uint32_t key = K(i);
                if ((key & mask) == prefix) atomicAdd(&hist[hb][(key >> (8 * pass)) & 255u], 1u);
            }
        }
        leading = false;
        if (tid < 256) hist[hb ^ 1][tid] = 0;         // the next pass's histogram
        __syncthreads();
        // ge[b] = #keys in bins >= b; the one bin with ge[b] >= need > ge[b+1] is the next byte of the k-th largest key.
        // Lane l of wave 0 owns bins 4l..4l+3: local suffix sums + an exclusive suffix scan of the lane totals.
        if (tid < 64) {
            const uint4 h = *reinterpret_cast<const uint4 *>(&hist[hb][4 * tid]);
            const uint32_t s3 = h.w, s2 = h.z + s3, s1 = h.y + s2, s0 = h.x + s1;
            uint32_t inc = s0;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_down(inc, d); if (tid + d < 64) inc += o; }
            const uint32_t ex = inc - s0;
            const uint32_t ge[4] = {s0 + ex, s1 + ex, s2 + ex, s3 + ex}, gt[4] = {s1 + ex, s2 + ex, s3 + ex, ex};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ge[j] >= need && gt[j] < need) { sh_prefix = prefix | ((uint32_t)(4 * tid + j) << (8 * pass)); sh_need = need - gt[j]; }
        }
        __syncthreads();
        prefix = sh_prefix; need = sh_need; mask |= 0xFFu << (8 * pass);
    }
    CEM_SEL_STAMP(2);
    const uint32_t T = prefix;          // key of the k-th largest score; `need` ties are taken, lowest index first

    // ---- compaction in ascending candidate index (tf.nn.top_k ties -> lower index) -------------------------
    const int C = (N + 1023) / 1024;
    const int beg = tid * C, end = (beg + C < N) ? beg + C : N;
    uint32_t ngt = 0, neq = 0;
    for (int i = beg; i < end; ++i) { const uint32_t key = K(i); ngt += key > T; neq += key == T; }
    uint32_t pre_gt, pre_eq;
    cem_block_excl_scan2(ngt, neq, wsum, pre_gt, pre_eq);
    // ... and, in the same walk, the best of elite: max score, first (= lowest candidate index) among exact ties (cem_mpc.py:57-60).  A thread
    // sees its taken keys in ascending candidate order, so a strict `>` keeps the lowest index; threads and waves combine on (score, index).
    // (Round 5: this was a pass of its own over the elite list behind the barrier — 0.8 of the select's 10 us.)  Per-wave candidates now;
    // the final combine and the update of best-so-far run on wave 1 at the very end.
    {
        uint32_t eqr = pre_eq;
        uint32_t pos = pre_gt + (pre_eq < need ? pre_eq : need);
        float bs = -__builtin_inff(); int bp = 0x7fffffff;               // bp: CANDIDATE index of this thread's best elite
        for (int i = beg; i < end; ++i) {
            const float scv = CACHE ? 0.f : p.scores[i];
            const uint32_t key = CACHE ? ckey[CEM_SEL_KIDX(i)] : cem_f2key(scv);
            bool take = key > T;
            if (key == T) { take = eqr < need; ++eqr; }
            if (take) {
                elite[pos] = i; p.elite_idx[pos] = i; ++pos;
                const float sc = CACHE ? cem_key2f(key) : scv;
                if (bp == 0x7fffffff || sc > bs) { bs = sc; bp = i; }
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float os = __shfl_xor(bs, d); const int op = __shfl_xor(bp, d);
            if (op != 0x7fffffff && (bp == 0x7fffffff || os > bs || (os == bs && op < bp))) { bs = os; bp = op; }
        }
        if ((tid & 63) == 0) { bsc[tid >> 6] = bs; bpos[tid >> 6] = bp; }
    }
    __syncthreads();

    CEM_SEL_STAMP(3);

    CEM_SEL_STAMP(4);
    // ---- moments over the elite set (tf.nn.moments: mean, then mean squared difference) ---------------------
    const float fk = (float)k;
    const float sm = p.smoothing, osm = p.one_minus_smoothing;
    // Wide path for large elite sets (the replicated select of a multi-GPU plan): a thread gathers whole float4s of an elite's
    // action row, i.e. a quarter of the address arithmetic and load instructions per element — this kernel is issue-bound
    // on its one CU.  Partial sums still add up in a fixed order (elite index ascending within a part, parts ascending).
    {
        int tpc1 = 1; { const int nc = HA < 1024 ? HA : 1024; while (tpc1 * 2 * nc <= 1024) tpc1 *= 2; }
        if ((HA & 3) == 0 && HA <= 4096 && k > 16 * tpc1) {
            const int ncol4 = HA >> 2;
            int tpc = 1; while (tpc * 2 * ncol4 <= 1024) tpc *= 2;
            const int part = tid / ncol4, c4 = tid % ncol4;
            const bool act = part < tpc;
            f4 *red4 = reinterpret_cast<f4 *>(red);
            const f4 *act4 = reinterpret_cast<const f4 *>(p.actions);
            const f4 zero4 = (f4){0.f, 0.f, 0.f, 0.f};
            f4 av[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = part + j * tpc;
                av[j] = (act && e < k) ? act4[(size_t)elite[e] * ncol4 + c4] : zero4;
            }
            f4 mean4 = zero4;
            for (int phase = 0; phase < 2; ++phase) {
                f4 acc = zero4;
                if (act) {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (part + j * tpc < k) { const f4 a = av[j]; acc = phase ? acc + (a - mean4) * (a - mean4) : acc + a; }
                    for (int e0 = part + 8 * tpc; e0 < k; e0 += 8 * tpc) {
                        f4 b[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) { const int e = e0 + j * tpc; b[j] = e < k ? act4[(size_t)elite[e] * ncol4 + c4] : zero4; }
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (e0 + j * tpc < k) { const f4 a = b[j]; acc = phase ? acc + (a - mean4) * (a - mean4) : acc + a; }
                    }
                }
                __syncthreads();
                red4[tid] = acc;
                __syncthreads();
                if (part == 0) {
                    f4 tot = zero4;
                    for (int pp = 0; pp < tpc; ++pp) tot = tot + red4[pp * ncol4 + c4];
                    if (!phase) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) colmean[4 * c4 + r] = tot[r] / fk;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float sd = sqrtf(tot[r] / fk);
                            const int ci = 4 * c4 + r;
                            const float nsg = sm * p.musig[HA + ci] + osm * sd;                // cem_mpc.py:65
                            p.musig[ci] = sm * p.musig[ci] + osm * colmean[ci];                // cem_mpc.py:64
                            p.musig[HA + ci] = nsg;
                            newsig[ci] = nsg;
                        }
                    }
                }
                __syncthreads();
                if (!phase && act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) mean4[r] = colmean[4 * c4 + r];
                }
            }
        } else
    for (int cb = 0; cb < HA; cb += 1024) {
        const int ncol = (HA - cb < 1024) ? HA - cb : 1024;
        int tpc = 1; while (tpc * 2 * ncol <= 1024) tpc *= 2;
        const int part = tid / ncol, col = tid % ncol;
        const bool act = part < tpc;
        // this thread's elite rows e = part, part + tpc, ...: the first 16 are gathered at once and kept for both phases
        float av[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = part + j * tpc;
            av[j] = (act && e < k) ? p.actions[(size_t)elite[e] * HA + cb + col] : 0.f;
        }
        for (int phase = 0; phase < 2; ++phase) {
            float acc = 0.f;
            const float m = phase ? colmean[cb + col] : 0.f;
            if (act) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int e = part + j * tpc;
                    if (e < k) { const float a = av[j]; acc = phase ? acc + (a - m) * (a - m) : acc + a; }
                }
                for (int e0 = part + 16 * tpc; e0 < k; e0 += 8 * tpc) {      // large k: 8 gathers in flight, same summation order
                    float b[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const int e = e0 + j * tpc; b[j] = e < k ? p.actions[(size_t)elite[e] * HA + cb + col] : 0.f; }
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (e0 + j * tpc < k) { const float a = b[j]; acc = phase ? acc + (a - m) * (a - m) : acc + a; }
                }
            }
            __syncthreads();
            red[tid] = acc;
            __syncthreads();
            if (part == 0) {
                float tot = 0.f;
                for (int pp = 0; pp < tpc; ++pp) tot = tot + red[pp * ncol + col];
                if (!phase) colmean[cb + col] = tot / fk;
                else {
                    const float sd = sqrtf(tot / fk);
                    const int ci = cb + col;
                    const float omu = cb == 0 ? old_mu : p.musig[ci], osg = cb == 0 ? old_sg : p.musig[HA + ci];
                    const float nsg = sm * osg + osm * sd;                             // cem_mpc.py:65
                    p.musig[ci] = sm * omu + osm * colmean[ci];                        // cem_mpc.py:64
                    p.musig[HA + ci] = nsg;
                    newsig[ci] = nsg;
                }
            }
            __syncthreads();
        }
    }
    }
    CEM_SEL_STAMP(5);
    uint32_t *res_l = reinterpret_cast<uint32_t *>(red);     // [36] the plan's result in the making + [36] "this select completes the plan"  (LDS: the moments are done with `red`)
    if (p.result && tid < 37) res_l[tid] = 0u;
    if (p.result) __syncthreads();
    if (tid == 0) {
        float ssum = 0.f;
        for (int i = 0; i < HA; ++i) ssum = ssum + newsig[i];
        const float mean_sigma = ssum / (float)HA;
        const int iters = p.ctrl->iters + 1;
        const bool stop = mean_sigma <= p.threshold;                                          // cem_mpc.py:66-67
        p.ctrl->iters = iters;
        if (stop) p.ctrl->done = 1;
        if (p.result) {             // (the select that ends the plan — its last iteration or the early stop — hands the result to the host)
            res_l[33] = (uint32_t)iters; res_l[34] = stop ? 1u : (uint32_t)p.ctrl->done; res_l[35] = (uint32_t)p.ctrl->fault;
            res_l[36] = (stop || p.is_last != 0) ? 1u : 0u;
        }
        CEM_SEL_STAMP(6);
    }
    if (tid == 64) {
        float bs = bsc[0]; int bp = bpos[0];
        for (int i = 1; i < 16; ++i) {
            const float os = bsc[i]; const int op = bpos[i];
            if (op != 0x7fffffff && (bp == 0x7fffffff || os > bs || (os == bs && op < bp))) { bs = os; bp = op; }
        }
        const bool better = bs > p.ctrl->best_score;                         // strict (cem_mpc.py:58)
        const int idx = bp;                                                  // (candidate index: the compaction walk recorded it)
        for (int a0 = 0; a0 < p.A; a0 += 4) {
            f4 e = (f4){0.f, 0.f, 0.f, 0.f};                                                    // the output noise of these four actions: one draw
            if (p.result && !p.eps_out) e = cem_normal4((uint32_t)(a0 >> 2), 0u, 0u, 0u, CEM_STREAM_OUT, cem_key(p.ctrl));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = a0 + r;
                if (a < p.A) {
                    const float b = better ? p.actions[(size_t)idx * HA + a] : p.ctrl->best[a];     // first step's action
                    if (better) p.ctrl->best[a] = b;
                    if (p.result) res_l[a] = __float_as_uint(b + (p.eps_out ? p.eps_out[a] : e[r]) * p.noise_stddev);   // cem_mpc.py:68
                }
            }
        }
        if (better) p.ctrl->best_score = bs;
        if (p.result) res_l[32] = __float_as_uint(better ? bs : p.ctrl->best_score);
    }
    if (p.result) {
        __syncthreads();
        if (tid < 64 && res_l[36]) cem_emit_result(p.result, p.result_dev, res_l, p.ctrl->seq, tid);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same selection for LARGE populations (the replicated select of a many-GPU plan: B5 has N = 65536, k = 6554) as a chain
// of multi-workgroup kernels.  One workgroup cannot hold 65536 keys in LDS; its global-memory radix passes took 190 us of a
// 280-us kernel (scripts/stamp_select.py).  Here: three histogram kernels (11 + 11 + 10 bit digits of the order-preserving key,
// per-workgroup LDS histograms flushed with integer atomics — exact and order-independent), a count and a compaction kernel
// (elite indices in ascending candidate order, ties lowest index first, as tf.nn.top_k), two moment kernels (partial sums
// over groups of 256 elites, combined in group order: deterministic) and a one-workgroup tail (smoothing, early stop,
// best-so-far).  Every workgroup recomputes the few scalars it needs (digit of the k-th key, ...) from the global
// histograms instead of waiting for another workgroup.
// ---------------------------------------------------------------------------------------------------------
#define CEM_MS_KEYS 4096              // keys per workgroup (1024 threads x 4) in the histogram / count / compaction kernels
#define CEM_MS_BINS 2048
#define CEM_MS_EPG 256                // elites per workgroup in the moment kernels
struct MSelParams {
    const float *scores; const float *actions; float *musig; CtrlBlock *ctrl; int32_t *elite_idx;
    uint32_t *hist;                   // [3][CEM_MS_BINS] digit histograms (zeroed before the first pass)
    uint32_t *sel;                    // [2] key of the k-th largest score, ties to take
    uint32_t *wg_counts;              // [G][2] keys > T / == T in each workgroup's slice
    float *best_sc; int32_t *best_ix; // [G] best elite of each slice (score, candidate)
    float *part;                      // [2][G2][HA] partial sums of the two moment passes
    float *colmean;                   // [HA] elite mean, later the smoothed sigma
    int32_t N, k, HA, A, check_done, G, G2;
    float smoothing, one_minus_smoothing, threshold;   // one_minus_smoothing = fl32(1.0 - smoothing) rounded once, as cem_mpc.py:64-65 does
    uint32_t *bar;                    // cem_msel_fused_kernel: arrival counter of its grid barriers (zeroed with the histograms)
};

template <int PASS>
__global__ __launch_bounds__(1024) void cem_msel_hist_kernel(const MSelParams p)
{
    __shared__ uint32_t lh[CEM_MS_BINS];
    __shared__ uint32_t sh[20];
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    uint32_t need = (uint32_t)p.k, b0 = 0, b1 = 0;
    if (PASS >= 1) cem_ms_find(p.hist, CEM_MS_BINS, need, sh, b0, need);
    if (PASS >= 2) cem_ms_find(p.hist + CEM_MS_BINS, CEM_MS_BINS, need, sh, b1, need);
    for (int b = tid; b < CEM_MS_BINS; b += 1024) lh[b] = 0u;
    __syncthreads();
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int i = blockIdx.x * CEM_MS_KEYS + j * 1024 + tid; v[j] = i < p.N ? p.scores[i] : 0.f; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * CEM_MS_KEYS + j * 1024 + tid;
        if (i < p.N) {
            const uint32_t key = cem_f2key(v[j]);
            if (PASS == 0) atomicAdd(&lh[key >> 21], 1u);
            else if (PASS == 1) { if ((key >> 21) == b0) atomicAdd(&lh[(key >> 10) & 2047u], 1u); }
            else { if ((key >> 10) == ((b0 << 11) | b1)) atomicAdd(&lh[key & 1023u], 1u); }
        }
    }
    __syncthreads();
    for (int b = tid; b < CEM_MS_BINS; b += 1024) { const uint32_t c = lh[b]; if (c) atomicAdd(&p.hist[PASS * CEM_MS_BINS + b], c); }
}

__global__ __launch_bounds__(1024) void cem_msel_count_kernel(const MSelParams p)
{
    __shared__ uint32_t sh[20];
    __shared__ uint32_t red[2][16];
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    uint32_t need = (uint32_t)p.k, b0, b1, b2;
    cem_ms_find(p.hist, CEM_MS_BINS, need, sh, b0, need);
    cem_ms_find(p.hist + CEM_MS_BINS, CEM_MS_BINS, need, sh, b1, need);
    cem_ms_find(p.hist + 2 * CEM_MS_BINS, 1024, need, sh, b2, need);
    const uint32_t T = (b0 << 21) | (b1 << 10) | b2;
    if (blockIdx.x == 0 && tid == 0) { p.sel[0] = T; p.sel[1] = need; }
    uint32_t ngt = 0, neq = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * CEM_MS_KEYS + 4 * tid + j;              // the same key -> thread map as the compaction kernel
        if (i < p.N) { const uint32_t key = cem_f2key(p.scores[i]); ngt += key > T; neq += key == T; }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { ngt += __shfl_xor(ngt, d); neq += __shfl_xor(neq, d); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = ngt; red[1][tid >> 6] = neq; }
    __syncthreads();
    if (tid == 0) {
        uint32_t a = 0, b = 0;
        for (int i = 0; i < 16; ++i) { a += red[0][i]; b += red[1][i]; }
        p.wg_counts[2 * blockIdx.x] = a; p.wg_counts[2 * blockIdx.x + 1] = b;
    }
}

__global__ __launch_bounds__(1024) void cem_msel_compact_kernel(const MSelParams p)
{
    __shared__ uint32_t wsum[2][16];
    __shared__ uint32_t base[2];
    __shared__ float bsc[16];
    __shared__ int bix[16];
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    const uint32_t T = p.sel[0], need = p.sel[1];
    // keys > T / == T in the slices before this one (ascending candidate order = ascending workgroup order)
    {
        uint32_t a = 0, b = 0;
        for (int g = tid; g < (int)blockIdx.x; g += 1024) { a += p.wg_counts[2 * g]; b += p.wg_counts[2 * g + 1]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
        if ((tid & 63) == 0) { wsum[0][tid >> 6] = a; wsum[1][tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) { uint32_t x = 0, y = 0; for (int i = 0; i < 16; ++i) { x += wsum[0][i]; y += wsum[1][i]; } base[0] = x; base[1] = y; }
        __syncthreads();
    }
    const uint32_t gt_before = base[0], eq_before = base[1];
    __syncthreads();
    uint32_t key[4]; float sc[4];
    uint32_t ngt = 0, neq = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * CEM_MS_KEYS + 4 * tid + j;
        sc[j] = i < p.N ? p.scores[i] : 0.f;
        key[j] = i < p.N ? cem_f2key(sc[j]) : 0u;
        if (i < p.N) { ngt += key[j] > T; neq += key[j] == T; }
    }
    uint32_t pre_gt, pre_eq;
    cem_block_excl_scan2(ngt, neq, wsum, pre_gt, pre_eq);
    uint32_t eqr = eq_before + pre_eq;
    uint32_t pos = gt_before + pre_gt + (eqr < need ? eqr : need);
    float bs = -__builtin_inff(); int bi = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = blockIdx.x * CEM_MS_KEYS + 4 * tid + j;
        if (i < p.N) {
            bool take = key[j] > T;
            if (key[j] == T) { take = eqr < need; ++eqr; }
            if (take) {
                p.elite_idx[pos++] = i;
                if (bi == 0x7fffffff || sc[j] > bs) { bs = sc[j]; bi = i; }      // ascending i: the first maximum is the lowest index
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float os = __shfl_xor(bs, d); const int oi = __shfl_xor(bi, d);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; }
    }
    __syncthreads();
    if ((tid & 63) == 0) { bsc[tid >> 6] = bs; bix[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int i = 1; i < 16; ++i) { const float os = bsc[i]; const int oi = bix[i]; if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; } }
        p.best_sc[blockIdx.x] = bs; p.best_ix[blockIdx.x] = bi;
    }
}

// moments over the elite set (tf.nn.moments: mean, then mean squared difference).  Workgroup g owns elites [256 g, 256 g + 256);
// thread (sub, col) adds column col over the elites e = sub, sub + 4, ... of its group, the four sub-sums are added in order
// 0..3, and the groups' partial sums are added in group order by whoever needs the total: a fixed order.
template <int PHASE>
__global__ __launch_bounds__(256) void cem_msel_moments_kernel(const MSelParams p)
{
    __shared__ float red[4][64];
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x, sub = tid >> 6, lc = tid & 63;
    const int e0 = blockIdx.x * CEM_MS_EPG, e1 = (e0 + CEM_MS_EPG < p.k) ? e0 + CEM_MS_EPG : p.k;
    const float fk = (float)p.k;
    for (int c0 = 0; c0 < p.HA; c0 += 64) {
        const int col = c0 + lc;
        const bool live = col < p.HA;
        float mean = 0.f;
        if (PHASE == 1 && live) {
            float t = 0.f;
            for (int g = 0; g < p.G2; ++g) t = t + p.part[(size_t)g * p.HA + col];
            mean = t / fk;
            if (blockIdx.x == 0 && sub == 0) p.colmean[col] = mean;
        }
        float acc = 0.f;
        if (live) {
            for (int e = e0 + sub; e < e1; e += 32) {                      // 8 gathers in flight
                float a[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int ee = e + 4 * j; a[j] = ee < e1 ? p.actions[(size_t)p.elite_idx[ee] * p.HA + col] : 0.f; }
#pragma unroll
                for (int j = 0; j < 8; ++j) if (e + 4 * j < e1) acc = PHASE ? acc + (a[j] - mean) * (a[j] - mean) : acc + a[j];
            }
        }
        red[sub][lc] = acc;
        __syncthreads();
        if (sub == 0 && live) p.part[((size_t)PHASE * p.G2 + blockIdx.x) * p.HA + col] = ((red[0][lc] + red[1][lc]) + red[2][lc]) + red[3][lc];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void cem_msel_final_kernel(const MSelParams p)
{
    if (p.check_done && p.ctrl->done) return;
    const int tid = threadIdx.x;
    const float fk = (float)p.k, sm = p.smoothing, osm = p.one_minus_smoothing;
    for (int col = tid; col < p.HA; col += 256) {
        float t = 0.f;
        for (int g = 0; g < p.G2; ++g) t = t + p.part[((size_t)p.G2 + g) * p.HA + col];
        const float sd = sqrtf(t / fk);
        const float nsg = sm * p.musig[p.HA + col] + osm * sd;                   // cem_mpc.py:65
        p.musig[col] = sm * p.musig[col] + osm * p.colmean[col];                 // cem_mpc.py:64
        p.musig[p.HA + col] = nsg;
    }
    __syncthreads();
    if (tid == 0) {
        float ssum = 0.f;
        for (int i = 0; i < p.HA; ++i) ssum = ssum + p.musig[p.HA + i];
        p.ctrl->iters = p.ctrl->iters + 1;
        if (ssum / (float)p.HA <= p.threshold) p.ctrl->done = 1;                         // cem_mpc.py:66-67
    }
    if (tid == 64) {
        float bs = p.best_sc[0]; int bi = p.best_ix[0];
        for (int g = 1; g < p.G; ++g) { const float os = p.best_sc[g]; const int oi = p.best_ix[g]; if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; } }
        if (bi != 0x7fffffff && bs > p.ctrl->best_score) {                                // strict (cem_mpc.py:58)
            for (int a = 0; a < p.A; ++a) p.ctrl->best[a] = p.actions[(size_t)bi * p.HA + a];
            p.ctrl->best_score = bs;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The chain above as ONE launch (round 3): the same phases, the same per-phase arithmetic and summation orders (elite set, best
// action, mu and sigma are bit-identical to the chain's), separated by grid barriers instead of kernel boundaries.  The chain's
// eight launches cost a fixed ~40 us, which made it lose to the one-workgroup kernel below ~30 000 keys; a grid barrier is one
// atomic and a short poll.  A thread keeps its four keys in registers across all phases (the chain re-reads the scores three
// times).  All G = ceil(N / 4096) workgroups must be resident at once: the host only takes this form when G <= what the runtime says
// the device keeps resident of this kernel, and every poll is bounded — a barrier that does not complete (CUs held by other work)
// sets CEM_FAULT_BARRIER instead of hanging the device; the launch then commits NOTHING of the optimiser's state, and
// cem_msel_solo_kernel, launched right behind it, redoes the iteration's select from the same scores (round 5; before, the plan
// failed with CEM_ERR_DEVICE).
// Data that crosses workgroups INSIDE the launch (histograms, slice counts, the elite list, moment partial sums, slice bests)
// is written and read at the device coherence point (agent-scope relaxed atomics = sc1 accesses; no cache maintenance), every
// wave drains its stores (s_waitcnt vmcnt(0)) before its workgroup arrives at a barrier: the XCDs' L2s are not coherent with
// each other for plain accesses (MI355X_MICROARCH.md, correctness boundaries).
// ---------------------------------------------------------------------------------------------------------
#define CEM_GRID_SPIN_LIMIT (1u << 22)
#define CEM_FAULT_SEGMENT 1          // CtrlBlock::fault bits
#define CEM_FAULT_BARRIER 2
#define CEM_FAULT_RECOVERED 4
// `may_fault`: whether an expired poll of THIS workgroup invalidates the select.  False only at the last barrier for the workgroups that
// have nothing left to do after it (all but workgroup 0): whether the tail commits must depend on workgroup 0's own view alone — a
// late report from a workgroup that is about to exit, after the tail has committed, would have the recovery kernel redo a select
// whose smoothing blend has already been applied.  `inject`: test hook (CtrlBlock::inject), this workgroup behaves as if starved.
__device__ __forceinline__ void cem_grid_barrier(uint32_t *ctr, const uint32_t target, CtrlBlock *ctrl, const bool may_fault = true, const bool inject = false)
{
    __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0): this wave's sc1 stores / atomics (a fault report included) are acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = inject ? CEM_GRID_SPIN_LIMIT : 0u;
        while (spins < CEM_GRID_SPIN_LIMIT && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { ++spins; __builtin_amdgcn_s_sleep(1); }
        if (spins >= CEM_GRID_SPIN_LIMIT && may_fault) atomicOr(&ctrl->fault, CEM_FAULT_BARRIER);
    }
    __syncthreads();
}
#define CEM_LDC(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CEM_STC(ptr, v) __hip_atomic_store((ptr), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// The multi-workgroup select as one body.  SOLO false: workgroup blockIdx.x of a grid of G plays slice blockIdx.x, phases separated by
// grid barriers (cem_msel_fused_kernel).  SOLO true: ONE workgroup plays all G slices of every phase in turn, phases separated by
// its own barrier (cem_msel_solo_kernel, the recovery form): the same per-slice arithmetic, the same summation orders — bit-identical
// results, no residency assumption, several times slower.  A slice's four keys per thread stay in registers across the phases when a
// workgroup owns one slice; the solo form re-reads the scores per phase (as select_mode 2's chain does).
template <bool SOLO>
__device__ __forceinline__ void cem_msel_body(const MSelParams &p, uint32_t *lh /* [CEM_MS_BINS] */, uint32_t *sh /* [20] */, uint32_t (*wsum)[16],
                                              uint32_t *base /* [2] */, float *bsc /* [16] */, int *bix /* [16] */, float (*red)[4][64])
{
    const int tid = threadIdx.x;
    const uint32_t G = (uint32_t)p.G;
    const int vb0 = SOLO ? 0 : (int)blockIdx.x, vb1 = SOLO ? (int)G : (int)blockIdx.x + 1;
    uint32_t phase = 0;
    const bool inject = !SOLO && p.ctrl->inject == 1u && p.ctrl->iters == 0 && blockIdx.x == G - 1u;
#define CEM_MS_BARRIER(LAST_) do { if (SOLO) { __builtin_amdgcn_s_waitcnt(0x0F70); __syncthreads(); } \
        else { ++phase; cem_grid_barrier(p.bar, G * phase, p.ctrl, !(LAST_) || blockIdx.x == 0, inject && phase == 1u); } } while (0)

    // a slice's four keys per thread (consecutive candidates: ascending order inside the thread, the thread order = candidate order)
    uint32_t key[4]; float sc[4];
    int i0 = 0;
#define CEM_MS_LOAD(VB_) do { i0 = (VB_) * CEM_MS_KEYS + 4 * tid; \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) { sc[j] = i0 + j < p.N ? p.scores[i0 + j] : 0.f; key[j] = i0 + j < p.N ? cem_f2key(sc[j]) : 0u; } } while (0)
    if (!SOLO) CEM_MS_LOAD(vb0);

    // ---- three digit histograms (11 + 11 + 10 bits) of the order-preserving keys: per-workgroup LDS histogram -> global atomics
    uint32_t need = (uint32_t)p.k, b0 = 0, b1 = 0, b2 = 0;
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 1) cem_ms_find<true>(p.hist, CEM_MS_BINS, need, sh, b0, need);
        if (pass == 2) cem_ms_find<true>(p.hist + CEM_MS_BINS, CEM_MS_BINS, need, sh, b1, need);
        for (int vb = vb0; vb < vb1; ++vb) {
            if (SOLO) { __syncthreads(); CEM_MS_LOAD(vb); }          // (the previous slice's flush has read lh)
            for (int b = tid; b < CEM_MS_BINS; b += 1024) lh[b] = 0u;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < p.N) {
                    if (pass == 0) atomicAdd(&lh[key[j] >> 21], 1u);
                    else if (pass == 1) { if ((key[j] >> 21) == b0) atomicAdd(&lh[(key[j] >> 10) & 2047u], 1u); }
                    else { if ((key[j] >> 10) == ((b0 << 11) | b1)) atomicAdd(&lh[key[j] & 1023u], 1u); }
                }
            __syncthreads();
            for (int b = tid; b < CEM_MS_BINS; b += 1024) { const uint32_t c = lh[b]; if (c) atomicAdd(&p.hist[pass * CEM_MS_BINS + b], c); }
        }
        CEM_MS_BARRIER(false);
    }
    cem_ms_find<true>(p.hist + 2 * CEM_MS_BINS, 1024, need, sh, b2, need);
    const uint32_t T = (b0 << 21) | (b1 << 10) | b2;      // key of the k-th largest score; `need` keys equal to T are taken, lowest index first

    // ---- keys > T / == T per slice
    uint32_t ngt = 0, neq = 0;
    for (int vb = vb0; vb < vb1; ++vb) {
        if (SOLO) { __syncthreads(); CEM_MS_LOAD(vb); }
        ngt = 0; neq = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (i0 + j < p.N) { ngt += key[j] > T; neq += key[j] == T; }
        uint32_t a = ngt, b = neq;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
        if ((tid & 63) == 0) { wsum[0][tid >> 6] = a; wsum[1][tid >> 6] = b; }
        __syncthreads();
        if (tid == 0) {
            uint32_t x = 0, y = 0;
            for (int i = 0; i < 16; ++i) { x += wsum[0][i]; y += wsum[1][i]; }
            CEM_STC(&p.wg_counts[2 * vb], x); CEM_STC(&p.wg_counts[2 * vb + 1], y);
        }
    }
    CEM_MS_BARRIER(false);

    // ---- compaction: elite indices in ascending candidate order, ties lowest index first (tf.nn.top_k); best elite of the slice
    for (int vb = vb0; vb < vb1; ++vb) {
        if (SOLO) {
            __syncthreads(); CEM_MS_LOAD(vb);
            ngt = 0; neq = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) if (i0 + j < p.N) { ngt += key[j] > T; neq += key[j] == T; }
        }
        {
            uint32_t a = 0, b = 0;
            for (int g = tid; g < vb; g += 1024) { a += CEM_LDC(&p.wg_counts[2 * g]); b += CEM_LDC(&p.wg_counts[2 * g + 1]); }
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d); b += __shfl_xor(b, d); }
            __syncthreads();
            if ((tid & 63) == 0) { wsum[0][tid >> 6] = a; wsum[1][tid >> 6] = b; }
            __syncthreads();
            if (tid == 0) { uint32_t x = 0, y = 0; for (int i = 0; i < 16; ++i) { x += wsum[0][i]; y += wsum[1][i]; } base[0] = x; base[1] = y; }
            __syncthreads();
        }
        const uint32_t gt_before = base[0], eq_before = base[1];
        __syncthreads();
        uint32_t pre_gt, pre_eq;
        cem_block_excl_scan2(ngt, neq, wsum, pre_gt, pre_eq);
        uint32_t eqr = eq_before + pre_eq;
        uint32_t pos = gt_before + pre_gt + (eqr < need ? eqr : need);
        float bs = -__builtin_inff(); int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i0 + j < p.N) {
                bool take = key[j] > T;
                if (key[j] == T) { take = eqr < need; ++eqr; }
                if (take) {
                    if (pos < (uint32_t)p.k) CEM_STC(&p.elite_idx[pos], i0 + j);              // (pos >= k only after an expired barrier: counts of another iteration)
                    ++pos;
                    if (bi == 0x7fffffff || sc[j] > bs) { bs = sc[j]; bi = i0 + j; }          // ascending i: the first maximum is the lowest index
                }
            }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const float os = __shfl_xor(bs, d); const int oi = __shfl_xor(bi, d);
            if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { bsc[tid >> 6] = bs; bix[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int i = 1; i < 16; ++i) { const float os = bsc[i]; const int oi = bix[i]; if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; } }
            CEM_STC(&p.best_sc[vb], bs); CEM_STC(&p.best_ix[vb], bi);
        }
    }
    CEM_MS_BARRIER(false);

    // ---- moments over the elite set (tf.nn.moments: mean, then mean squared difference): the chain's decomposition — groups of
    //      256 elites, four sub-sums per group added in order 0..3, groups added in group order — with each quarter of a
    //      workgroup (256 threads) playing one of the chain's moment workgroups
    const int sb = tid >> 8, t256 = tid & 255, sub = t256 >> 6, lc = t256 & 63;
    const float fk = (float)p.k;
    const int groups_per_round = SOLO ? 4 : (int)G * 4;
    const int rounds = (p.G2 + groups_per_round - 1) / groups_per_round;
#pragma unroll 1
    for (int ph = 0; ph < 2; ++ph) {
        for (int rd = 0; rd < rounds; ++rd) {
            const int g2 = SOLO ? rd * 4 + sb : (rd * (int)G + (int)blockIdx.x) * 4 + sb;
            const bool have = g2 < p.G2;
            const int e0 = g2 * CEM_MS_EPG, e1 = (e0 + CEM_MS_EPG < p.k) ? e0 + CEM_MS_EPG : p.k;
            for (int c0 = 0; c0 < p.HA; c0 += 64) {
                const int col = c0 + lc;
                const bool live = have && col < p.HA;
                float mean = 0.f;
                if (ph == 1 && live) {
                    float t = 0.f;
                    for (int g = 0; g < p.G2; ++g) t = t + CEM_LDC(&p.part[(size_t)g * p.HA + col]);
                    mean = t / fk;
                    if (g2 == 0 && sub == 0) CEM_STC(&p.colmean[col], mean);
                }
                float acc = 0.f;
                if (live) {
                    for (int e = e0 + sub; e < e1; e += 32) {                      // 8 gathers in flight
                        float a[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int ee = e + 4 * j;
                            int ei = ee < e1 ? CEM_LDC(&p.elite_idx[ee]) : 0;
                            ei = ei < 0 ? 0 : (ei >= p.N ? p.N - 1 : ei);          // (an expired barrier may leave another launch's word here: stay inside `actions`)
                            a[j] = ee < e1 ? p.actions[(size_t)ei * p.HA + col] : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) if (e + 4 * j < e1) acc = ph ? acc + (a[j] - mean) * (a[j] - mean) : acc + a[j];
                    }
                }
                red[sb][sub][lc] = acc;
                __syncthreads();
                if (sub == 0 && live) CEM_STC(&p.part[((size_t)ph * p.G2 + g2) * p.HA + col], ((red[sb][0][lc] + red[sb][1][lc]) + red[sb][2][lc]) + red[sb][3][lc]);
                __syncthreads();
            }
        }
        CEM_MS_BARRIER(ph == 1);
    }
#undef CEM_MS_BARRIER
#undef CEM_MS_LOAD

    // ---- tail (workgroup 0): smoothing, early stop, best-so-far — cem_msel_final_kernel's statements.  Nothing of the optimiser's
    //      state (mu / sigma, iteration count, best-so-far, early stop) is touched before this point, and none of it is touched if a
    //      barrier of this launch expired anywhere: the recovery kernel then redoes the whole select from the same scores.
    if (!SOLO && blockIdx.x != 0) return;
    if (!SOLO && (CEM_LDC(&p.ctrl->fault) & CEM_FAULT_BARRIER)) return;
    const float sm = p.smoothing, osm = p.one_minus_smoothing;
    for (int col = tid; col < p.HA; col += 1024) {
        float t = 0.f;
        for (int g = 0; g < p.G2; ++g) t = t + CEM_LDC(&p.part[((size_t)p.G2 + g) * p.HA + col]);
        const float sd = sqrtf(t / fk);
        const float nsg = sm * p.musig[p.HA + col] + osm * sd;                   // cem_mpc.py:65
        p.musig[col] = sm * p.musig[col] + osm * CEM_LDC(&p.colmean[col]);        // cem_mpc.py:64
        p.musig[p.HA + col] = nsg;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (tid == 0) {
        float ssum = 0.f;
        for (int i = 0; i < p.HA; ++i) ssum = ssum + p.musig[p.HA + i];
        p.ctrl->iters = p.ctrl->iters + 1;
        if (ssum / (float)p.HA <= p.threshold) p.ctrl->done = 1;                         // cem_mpc.py:66-67
    }
    if (tid == 64) {
        float bs = CEM_LDC(&p.best_sc[0]); int bi = CEM_LDC(&p.best_ix[0]);
        for (int g = 1; g < p.G; ++g) { const float os = CEM_LDC(&p.best_sc[g]); const int oi = CEM_LDC(&p.best_ix[g]); if (oi != 0x7fffffff && (bi == 0x7fffffff || os > bs || (os == bs && oi < bi))) { bs = os; bi = oi; } }
        if (bi != 0x7fffffff && bs > p.ctrl->best_score) {                                // strict (cem_mpc.py:58)
            for (int a = 0; a < p.A; ++a) p.ctrl->best[a] = p.actions[(size_t)bi * p.HA + a];
            p.ctrl->best_score = bs;
        }
    }
}

#define CEM_MSEL_SHARED() \
    __shared__ uint32_t lh[CEM_MS_BINS]; __shared__ uint32_t sh[20]; __shared__ uint32_t wsum[2][16]; __shared__ uint32_t base[2]; \
    __shared__ float bsc[16]; __shared__ int bix[16]; __shared__ float red[4][4][64]

__global__ __launch_bounds__(1024) void cem_msel_fused_kernel(const MSelParams p)
{
    CEM_MSEL_SHARED();
    if (p.check_done && p.ctrl->done) return;             // uniform over the grid: set by the previous iteration's tail
    cem_msel_body<false>(p, lh, sh, wsum, base, bsc, bix, red);
}

// Recovery: launched behind every cem_msel_fused_kernel, one workgroup, returns at once unless a grid barrier of that launch expired
// (CEM_FAULT_BARRIER: the fused kernel's workgroups were not all resident — another stream / process held CUs).  Then it redoes the
// iteration's whole select from the same scores, alone, phase by phase — no residency assumption, the bits of select_mode 2 — clears the
// fault and leaves CEM_FAULT_RECOVERED for the host, which logs it once and stops fusing on this handle (cem_capi.hip after_plan).
// In stream order and rank-local: a sharded plan's collectives are not disturbed and no rank has to agree with another about it.
__global__ __launch_bounds__(1024) void cem_msel_solo_kernel(const MSelParams p)
{
    CEM_MSEL_SHARED();
    if (p.check_done && p.ctrl->done) return;
    if (!(CEM_LDC(&p.ctrl->fault) & CEM_FAULT_BARRIER)) return;
    for (int b = threadIdx.x; b < 3 * CEM_MS_BINS; b += 1024) CEM_STC(&p.hist[b], 0u);      // the expired launch left partial counts
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    cem_msel_body<true>(p, lh, sh, wsum, base, bsc, bix, red);
    __syncthreads();
    if (threadIdx.x == 0) { atomicAnd(&p.ctrl->fault, ~CEM_FAULT_BARRIER); atomicOr(&p.ctrl->fault, CEM_FAULT_RECOVERED); }
}
#undef CEM_MSEL_SHARED
#undef CEM_LDC
#undef CEM_STC

// What a plan returns (cem_mpc.py:68: best_so_far + N(0, noise_stddev), and best_so_far_score), written where the HOST reads it:
// `result` is pinned host memory (no device -> host copy node after the plan).  Layout: [0, A) action, [32] score, [33] iterations run,
// [34] early-stop flag, [35] fault bits, [36] plan counter, [37] checksum (cem_result_checksum); the same words go to the workspace copy.
struct FinalParams { const CtrlBlock *ctrl; const float *eps_out; float *result; uint32_t *result_dev; int32_t A; float noise_stddev; };

__global__ void cem_final_kernel(const FinalParams p)
{
    __shared__ uint32_t res_l[40];
    const int a = threadIdx.x;
    if (a < 36) res_l[a] = 0u;
    __syncthreads();
    if (a < p.A) res_l[a] = __float_as_uint(p.ctrl->best[a] + cem_out_noise(p.ctrl, p.eps_out, a) * p.noise_stddev);   // cem_mpc.py:68
    if (a == 0) {
        res_l[32] = __float_as_uint(p.ctrl->best_score);
        res_l[33] = (uint32_t)p.ctrl->iters; res_l[34] = (uint32_t)p.ctrl->done; res_l[35] = (uint32_t)p.ctrl->fault;
    }
    __syncthreads();
    if (a < 64) cem_emit_result(p.result, p.result_dev, res_l, p.ctrl->seq, a);
}

// the raw Philox4x32-7 words of n counters (idx0 + i, t | it << 16, sub | stream << 16, call_lo) — what cem_normal4 turns into four
// normals — so that a test can hold the generator against an independent implementation word for word (cem_philox_words)
struct WordsParams { uint32_t *out; const CtrlBlock *ctrl; uint32_t stream, it, t, sub, idx0, n; };
__global__ __launch_bounds__(256) void cem_philox_words_kernel(const WordsParams p)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const PhiloxKey key = cem_key(p.ctrl);
    uint32_t c0 = p.idx0 + i, c1 = p.t | (p.it << 16), c2 = p.sub | (p.stream << 16), c3 = key.c3;
    philox4x32_7(c0, c1, c2, c3, key.k0, key.k1);
    p.out[4 * (size_t)i + 0] = c0; p.out[4 * (size_t)i + 1] = c1; p.out[4 * (size_t)i + 2] = c2; p.out[4 * (size_t)i + 3] = c3;
}

struct FillParams { float *eps_act, *eps_model, *eps_out; const CtrlBlock *ctrl; int32_t I, N, H, A, B, O; };

// dump the Philox streams in the explicit-tensor layouts (parity mode inputs)
__global__ __launch_bounds__(256) void cem_fill_noise_kernel(const FillParams p)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gsz = (size_t)gridDim.x * blockDim.x;
    if (p.eps_act) {
        const int AZ = (p.A + 3) >> 2;
        const size_t total = (size_t)p.I * p.N * p.H * AZ;
        for (size_t idx = gid; idx < total; idx += gsz) {
            const int z = idx % AZ, t = (idx / AZ) % p.H, n = (idx / ((size_t)AZ * p.H)) % p.N, it = idx / ((size_t)AZ * p.H * p.N);
            const f4 e = cem_normal4((uint32_t)n, (uint32_t)t, (uint32_t)it, (uint32_t)z, CEM_STREAM_ACT, cem_key(p.ctrl));
            for (int r = 0; r < 4; ++r) if (4 * z + r < p.A) p.eps_act[(((size_t)it * p.N + n) * p.H + t) * p.A + 4 * z + r] = e[r];
        }
    }
    if (p.eps_model) {
        const int OZ = (p.O + 3) >> 2;
        const size_t total = (size_t)p.I * p.H * p.B * OZ;
        for (size_t idx = gid; idx < total; idx += gsz) {
            const int fq = idx % OZ; const size_t row = (idx / OZ) % p.B; const int t = (idx / ((size_t)OZ * p.B)) % p.H, it = idx / ((size_t)OZ * p.B * p.H);
            const f4 e = cem_normal4((uint32_t)row, (uint32_t)t, (uint32_t)it, (uint32_t)fq, CEM_STREAM_MODEL, cem_key(p.ctrl));
            for (int r = 0; r < 4; ++r) if (4 * fq + r < p.O) p.eps_model[(((size_t)it * p.H + t) * p.B + row) * p.O + 4 * fq + r] = e[r];
        }
    }
    if (p.eps_out && gid < (size_t)p.A) {
        const f4 e = cem_normal4((uint32_t)(gid >> 2), 0u, 0u, 0u, CEM_STREAM_OUT, cem_key(p.ctrl));
        p.eps_out[gid] = e[gid & 3];
    }
}
