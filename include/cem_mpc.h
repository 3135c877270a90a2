/*
 * cem_mpc.h — C ABI of the MI355X-native CEM-MPC planner (libcem_mpc_gfx950.so).
 *
 * Drop-in boundary for ONE path of yardenas/ethz-safe-learning ("simba"):
 *   CemMpc.generate_action / do_generate_action      simba/policies/cem_mpc.py:31-68
 *   SafeCemMpc.compute_objective                     simba/policies/safe_cem_mpc.py:76-120
 *   MpcPolicy.compute_objective / sampling_params    simba/policies/mpc_policy.py:26-57
 *   TransitionModel.unfold_sequences / scale         simba/models/transition_model.py:64-87
 *   MlpEnsemble.forward / __call__                   simba/models/mlp_ensemble.py:122-132,189-193
 *   SafetyGymStateScorer.reward / cost ('goal' task) simba/environment_utils/safety_gym.py:110-192
 *
 * The reference is pure Python on TensorFlow; it has no FFI of its own.  These
 * entry points are what a ctypes binding inside simba/policies/cem_mpc.py would
 * call (INTEGRATION.md shows that binding).  Plain pointers and sizes only; no
 * torch types.  Device memory (the workspace, optional noise tensors) is owned
 * by the caller (torch-ROCm tensors or hipMalloc), the HIP stream is the
 * caller's.  Every function returns an int status (CEM_OK == 0); nothing
 * throws across the boundary.  A handle is not thread-safe; one plan in flight
 * per handle (the reference has one synchronous caller, simba/agents/agent.py:120).
 * A shape change (scripts/tune_cem_policy.py:109-115) = a new handle.
 *
 * Environment variables the library reads (none changes a result; all are diagnostics or deployment overrides):
 *   CEM_RCCL_LIBRARY=<file>        the RCCL to dlopen instead of librccl.so.1 (a site's build; the tests' shared-memory stand-in).  No
 *                                  fallback if it does not load; logged on stderr whenever it is honoured.
 *   CEM_FORCE_SAMPLER=tile|kernel  where cem_mpc.py:44-48 runs: as the rollout tiles' prologue or as a launch of its own (default: by the
 *                                  tile plan, see cem_planner_launches_per_iteration).
 *   CEM_ASSUME_CUS=<n>             price tile plans for n compute units (the GPU-less host helpers default to 256).  It moves the tile
 *                                  PLAN only (tile size, pinned / floating split — bit-identical results either way); residency decisions
 *                                  (the fused select's grid, where the sampler runs) always use the device's real multiProcessorCount.
 *   CEM_NO_POLL                    cem_planner_plan waits for a captured plan with hipStreamSynchronize instead of polling the result block in
 *                                  pinned memory (polling spins one host core for the duration of the plan and returns ~10 us sooner).
 *   CEM_FORCE_GENERIC_ROLLOUT      every configuration on the width-generic rollout kernel;  CEM_TRAIN_GEMM_KERNEL: the GEMM-by-GEMM trainer.
 */
#ifndef CEM_MPC_H
#define CEM_MPC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CEM_ABI_VERSION 4
#define CEM_MAX_ACT 32
#define CEM_MAX_COST_KINDS 4

enum cem_status {
    CEM_OK = 0,
    CEM_ERR_INVALID_ARG = 1,     /* NULL pointer / bad dims */
    CEM_ERR_UNSUPPORTED = 2,     /* e.g. units > 256, task 'push', obs+act > 128 */
    CEM_ERR_SPLIT = 3,           /* (particles*n_samples) % ensemble_size != 0: tf.split would raise (mlp_ensemble.py:123) */
    CEM_ERR_WORKSPACE = 4,       /* workspace too small / misaligned */
    CEM_ERR_HIP = 5,             /* a HIP runtime call failed; cem_last_hip_error() has the code */
    CEM_ERR_NO_WEIGHTS = 6,      /* plan() before set_weights() */
    CEM_ERR_STATE = 7,           /* stepwise calls out of order, or a call that would clobber the state of a plan in flight */
    CEM_ERR_COMM = 8,            /* librccl could not be opened, or an RCCL call failed (cem_last_hip_error() holds the ncclResult_t) */
    CEM_ERR_DEVICE = 9           /* a kernel reported that it could not finish its work (a floating rollout segment never received its
                                    work-queue entry within the spin bound; or a fused select's barrier expired AND its recovery did not
                                    run): the plan's result is not valid */
};

enum cem_variant { CEM_VARIANT_CEM = 0 /* CemMpc */, CEM_VARIANT_SAFE = 1 /* SafeCemMpc */ };

/* mlp_params['activation'] of config/models.yaml:12, which the reference `eval`s (mlp_ensemble.py:14): the hidden layers'
 * nonlinearity.  relu (the shipped value) runs on the tuned kernels; the others on the generic rollout kernel and the
 * GEMM-by-GEMM trainer.  tf.nn.elu: alpha 1; tf.nn.leaky_relu: alpha 0.2; tf.nn.selu: scale 1.0507..., alpha 1.6732...; tf.nn.swish (= silu):
 * z sigmoid(z); tf.nn.gelu: the exact form z Phi(z) (approximate=False) — TensorFlow's defaults.  Up to selu the derivative is a function
 * of the layer's OUTPUT, which is what the trainer keeps; swish / gelu are not monotone, so for them the trainer keeps the pre-activations too. */
enum cem_activation { CEM_ACT_RELU = 0, CEM_ACT_TANH = 1, CEM_ACT_SIGMOID = 2, CEM_ACT_ELU = 3, CEM_ACT_LEAKY_RELU = 4, CEM_ACT_SOFTPLUS = 5, CEM_ACT_SELU = 6,
                      CEM_ACT_SWISH = 7, CEM_ACT_GELU = 8 };

/* SafetyGymStateScorer fields used by the 'goal' task (safety_gym.py:104-176).
 * The constants come from safety_gym's Engine config (absent from the
 * reference tree), hence explicit. */
typedef struct cem_scorer {
    int32_t goal_mode;            /* 0: observe_goal_lidar (closest_distance over goal slice); 1: observe_goal_dist (relu of one feature) */
    int32_t goal_lo, goal_hi;     /* sensor_offset_table['goal_lidar'|'goal_dist'] */
    float lidar_max_dist;
    float goal_size;
    float goal_reached_dist;      /* the threshold of `goal_achieved = dist <= 0.8 * goal_size` (safety_gym.py:116) as the reference
                                   * rounds it: the Python-float product converted to an fp32 tensor, fl32(0.8 * goal_size) evaluated in
                                   * double — NOT fl32(goal_size) * 0.8, which is 1 ulp higher at the default goal_size 0.3 */
    float reward_distance;
    float reward_goal;
    float reward_clip;            /* <= 0: no clip (safety_gym.py:141) */
    int32_t constrain_indicator;
    int32_t n_cost_kinds;         /* constrained kinds, reference order vases,hazards,pillars,gremlins (safety_gym.py:148-163) */
    int32_t cost_lo[CEM_MAX_COST_KINDS];
    int32_t cost_hi[CEM_MAX_COST_KINDS];
    float cost_size[CEM_MAX_COST_KINDS];
} cem_scorer_t;

/* How the rollout's dense layers multiply.  Both accumulate in fp32 and keep every term of a product down to 2^-24 of it.
 * CEM_PRECISION_FP32: v_mfma_f32_16x16x4_f32 (the default; what every number in BASELINE / DESIGN is quoted on unless labelled).
 * CEM_PRECISION_SPLIT_BF16X3: weights and activations as exact three-way bf16 splits, the six leading bf16 x bf16 products per
 * fp32 product on v_mfma_f32_16x16x32_bf16 (csrc/cem_rollout_split.h).  Same oracle, same tolerances, not bit-identical to the
 * fp32 form; units <= 128 and relu only. */
enum cem_precision { CEM_PRECISION_FP32 = 0, CEM_PRECISION_SPLIT_BF16X3 = 1 };

/* Constructor kwargs of CemMpc / SafeCemMpc (cem_mpc.py:7-17, safe_cem_mpc.py:8-19)
 * + the model dims of TransitionModel/MlpEnsemble (transition_model.py:8-21,
 * config/models.yaml) + candidate sharding. */
typedef struct cem_config {
    int32_t abi_version;          /* CEM_ABI_VERSION */
    int32_t obs_dim, act_dim;
    int32_t units, n_layers;      /* mlp_params: units <= 128 run on the fast kernels (narrower layers zero-padded to the 128-wide form:
                                   * exactly the narrow network's result); 129..256 — and any activation other than relu — on the generic
                                   * kernels (same semantics; cem_rollout_wide.h) */
    int32_t activation;           /* enum cem_activation */
    int32_t ensemble_size;        /* E */
    int32_t particles;            /* P */
    int32_t n_samples;            /* N (global, over all ranks) */
    int32_t horizon;              /* H */
    int32_t n_elite;              /* k */
    int32_t iterations;           /* I */
    float smoothing;
    float one_minus_smoothing;    /* the factor `(1.0 - self.smoothing)` of cem_mpc.py:64-65 as the reference rounds it: a Python-float
                                   * difference converted ONCE to an fp32 tensor, fl32(1.0 - smoothing) evaluated in double — NOT
                                   * 1.0f - fl32(smoothing), which is one ulp off for 41 of the 99 two-decimal smoothing values
                                   * (0.09, 0.16, 0.29, 0.33 ...).  Must lie within 2e-7 of 1 - smoothing (else CEM_ERR_INVALID_ARG) */
    float stddev_threshold;
    float noise_stddev;
    int32_t variant;              /* enum cem_variant */
    float posterior_mean_threashold;   /* sic: the YAML key, config/policies.yaml:20 */
    int32_t sampling_propagation; /* config/agents.yaml:14 */
    int32_t scale_features;       /* config/agents.yaml:13 */
    /* MpcPolicy.sampling_params (mpc_policy.py:45-57), resolved by the caller */
    float act_lb[CEM_MAX_ACT], act_ub[CEM_MAX_ACT], act_mu0[CEM_MAX_ACT], act_sigma0[CEM_MAX_ACT];
    cem_scorer_t scorer;
    /* candidate sharding: this rank owns candidates [rank*N/world, (rank+1)*N/world) x all particles */
    int32_t world_size, rank;
    int32_t chunks_per_tile;      /* 0 = auto; 1..4 = 16-row chunks per workgroup tile */
    int32_t use_graph;            /* 1: capture the whole plan in a hipGraph (single-rank, Philox noise only) */
    int32_t select_mode;          /* 0 = auto; 1 = the one-workgroup select kernel (elite list + 2 H A floats must fit 140 KB of LDS, n_elite <= 24576:
                                   * else CEM_ERR_UNSUPPORTED — auto routes such shapes to 3 / 2 instead); 2 = the multi-workgroup chain of eight launches; 3 = that
                                   * chain as ONE launch with grid barriers: needs its ceil(N / 4096) workgroups resident at once (checked against the
                                   * runtime's occupancy x CU count; else 2 is taken) and is FASTEST on an otherwise idle GPU — CUs held by another
                                   * stream / handle / process or a CU-masked queue can starve a barrier, which then times out (bounded polls, never
                                   * a hang); the launch then commits nothing, a one-workgroup recovery kernel queued right behind it redoes that
                                   * iteration's select from the same scores with mode 2's bits, the plan completes normally (CEM_OK), one line goes
                                   * to stderr and the handle uses mode 2 from then on (cem_planner_select_mode reports it) — what auto picks
                                   * from 24 000 candidates on (the replicated select of a many-GPU plan; below, mode 1 is faster).  Same elite set, best action and
                                   * early stop in every mode; 2 and 3 are bit-identical; mu / sigma of 1 vs 2 / 3 agree to fp32 rounding
                                   * (the moments are summed in a different, still fixed, order) */
    int32_t rollout_segments;     /* 0 = auto; 1 = one workgroup per tile for the whole horizon; n > 1 = the rollout launch is a
                                   * work queue of (tile, horizon/n) items drawn by resident workgroups — evens out CU load when the
                                   * tile count is not a multiple of the CU count; results are bit-identical either way */
    int32_t precision;            /* enum cem_precision: how the rollout forms its fp32 products (ABI 4) */
} cem_config_t;

/* Byte offsets into the caller's workspace of the arrays a host binding needs
 * (torch views for the collective, debug outputs). */
typedef struct cem_layout {
    size_t scores_local;   /* float [N/world]   — this rank's candidate scores (input to the collective) */
    size_t scores_global;  /* float [N]         — all candidates' scores (output of the collective; == scores_local slot for world 1) */
    size_t actions;        /* float [N][H][A]   — the current iteration's clipped action sequences */
    size_t mu_sigma;       /* float [2][H][A]   — sampling mean, stddev */
    size_t elite_idx;      /* int32 [k]         — elite set of the last select, ascending index */
    size_t returns;        /* float [P*N/world] — per-row done-masked return of the last rollout */
    size_t costs;          /* uint8 [H][P*N/world] — per-step masked cost (safe variant) */
    size_t result;         /* uint32 [38]: the last completed plan's result as the final kernel left it, word for word the pinned-host block
                            * cem_planner_plan reads: [0, A) action (float), [32] best score (float), [33] iterations run, [34] early-stop
                            * flag, [35] fault bits, [36] the handle's plan counter, [37] checksum.  Valid after the handle's stream has
                            * drained (see cem_planner_plan) */
    size_t stamps;         /* int64 [tiles][4][8] — cycle stamps of the last rollout; written only by -DCEM_STAMPS diagnostic builds */
    size_t total;
} cem_layout_t;

typedef struct cem_planner cem_planner_t;

int cem_abi_version(void);
const char *cem_status_string(int status);
int cem_last_hip_error(void);

/* natural (Keras) weight blob: per member m, in order
 *   W_0[obs+act][U], b_0[U], W_1[U][U], b_1[U], ... W_{L-1}, b_{L-1},
 *   W_mu[U][obs], b_mu[obs], W_var[U][obs], b_var[obs]        (all row-major [in][out], mlp_ensemble.py:13,28-29) */
size_t cem_weight_blob_floats(const cem_config_t *cfg);
size_t cem_packed_weight_floats(const cem_config_t *cfg);
size_t cem_workspace_bytes(const cem_config_t *cfg);

/* host-only helpers (no GPU needed; exercised by the CPU test-suite) */
int cem_pack_weights_host(const cem_config_t *cfg, const float *blob, float *packed);
int cem_plan_tiles_host(const cem_config_t *cfg, int32_t *chunks_per_tile_out, int32_t *n_tiles_out,
                        int32_t *tiles_out /* [n_tiles][6]: row_base,cnt,member,act_base,noise_row_base,s0_base */, int32_t max_tiles);
/* horizon segments the rollout launch of this configuration uses (1 = unsegmented), as cem_planner_create would choose */
int cem_plan_segments_host(const cem_config_t *cfg, int32_t *segments_out, int32_t *steps_per_segment_out);

/* diagnostic: workgroups of the rollout kernels for (chunks_per_tile, obs+act <= 64 ? 1 : 2 input blocks per wave) one CU keeps
 * resident — what the tile-size choice assumes (`table_out[2]`) and what the HIP runtime reports (`runtime_out[2]`, 0 without a
 * device); element 0: one workgroup per tile, element 1: the pinned + floating-segment launch form. */
int cem_rollout_residency(int32_t chunks_per_tile, int32_t input_blocks_per_wave, int32_t *table_out, int32_t *runtime_out);

/* lifecycle.  `workspace` is device memory of >= cem_workspace_bytes(cfg), 256-B aligned; `hip_stream` a hipStream_t (NULL = default). */
int cem_planner_create(const cem_config_t *cfg, void *workspace, size_t workspace_bytes, void *hip_stream, cem_planner_t **out);
int cem_planner_destroy(cem_planner_t *h);
int cem_planner_layout(const cem_planner_t *h, cem_layout_t *out);

/* weight / normaliser sync after MlpEnsemble.fit and TransitionModel._fit_statistics
 * (mlp_ensemble.py:143-144, transition_model.py:42-50).  Host pointers. */
int cem_planner_set_weights(cem_planner_t *h, const float *blob, size_t n_floats);
int cem_planner_set_normaliser(cem_planner_t *h, const float *inputs_min, const float *inputs_max /* [obs+act] */);

/* CemMpc.generate_action (cem_mpc.py:31-33): state[obs] (host) -> action[act] (host).
 * Noise: Philox4x32-7 keyed (seed, call) when the eps pointers are NULL, otherwise explicit
 * DEVICE tensors eps_act[I][N][H][A], eps_model[I][H][P*N][obs] and HOST eps_out[A]
 * (parity mode: "identical seeds" == identical noise tensors).
 * Ordering: a captured plan (use_graph) returns as soon as its result block has landed in pinned host memory (sequence number +
 * checksum), which can be BEFORE the handle's stream has drained.  What it hands back on the host — action, score, iterations — is
 * complete; the DEVICE arrays of cem_layout_t — mu_sigma, elite_idx, scores, actions, result — are ordered only by the stream: a
 * caller that reads them synchronises the handle's stream first (the Python binding's accessors do).  cem_planner_destroy drains
 * the stream itself. */
int cem_planner_plan(cem_planner_t *h, const float *state, uint64_t seed, uint64_t call,
                     const float *eps_act_dev, const float *eps_model_dev, const float *eps_out_host,
                     float *action_out, float *best_score_out, int32_t *iters_out);

/* the same plan split at its one exchange step, for candidate-sharded ranks:
 *   begin; for it: rollout(it) -> [collective on scores_local -> scores_global] -> select(it); end */
int cem_plan_begin(cem_planner_t *h, const float *state, uint64_t seed, uint64_t call,
                   const float *eps_act_dev, const float *eps_model_dev);
int cem_plan_rollout(cem_planner_t *h, int32_t it);   /* sample actions, roll out + score this rank's candidates -> scores_local */
int cem_plan_select(cem_planner_t *h, int32_t it);    /* top-k / moments refit / best-so-far / early-stop on scores_global */
int cem_plan_end(cem_planner_t *h, const float *eps_out_host, float *action_out, float *best_score_out, int32_t *iters_out);

/* The exchange step inside the library: an RCCL communicator owned by the handle, so that a candidate-sharded
 * plan runs without the host between its kernels — cem_planner_plan() then works for world_size > 1 (rollout ->
 * ncclAllGather of the N/world local scores on the handle's stream -> select, per iteration) and, with use_graph, replays
 * it as ONE hipGraph per rank including the collectives.  librccl is opened at run time (dlopen), not linked.
 * Rank 0 calls cem_comm_unique_id() and hands the 128 bytes to the other ranks by any means (torch.distributed broadcast,
 * MPI, a file); every rank then calls cem_planner_comm_init() — collectively, like ncclCommInitRank.
 * cem_plan_exchange() is the same all-gather for the stepwise API (between cem_plan_rollout and cem_plan_select). */
#define CEM_COMM_ID_BYTES 128
int cem_comm_unique_id(void *id_out /* CEM_COMM_ID_BYTES */);
int cem_planner_comm_init(cem_planner_t *h, const void *id /* CEM_COMM_ID_BYTES */, int32_t n_ranks, int32_t rank);
int cem_planner_comm_destroy(cem_planner_t *h);
/* ranks of the handle's communicator as RCCL itself reports them (ncclCommCount); 0 without a communicator */
int cem_planner_comm_ranks(const cem_planner_t *h, int32_t *n_ranks_out);
int cem_plan_exchange(cem_planner_t *h);
/* The select form the handle's next iteration takes (1 / 2 / 3, see cem_config_t::select_mode): what automatic resolves to on this device,
 * and 2 once a fused select has had to be recovered on this handle. */
int cem_planner_select_mode(const cem_planner_t *h, int32_t *mode_out);
/* Test hook.  kind 1: in the NEXT plan's first iteration, the last workgroup of the fused select treats its first grid barrier as
 * expired (as if its peers were not resident) — the recovery path then runs without having to load the GPU.  No effect on plans
 * whose select is not fused. */
int cem_planner_inject_fault(cem_planner_t *h, int32_t kind);
/* 0: cem_planner_plan launches kernel by kernel; 1: it replays a captured hipGraph; 2: capturing was tried and is not supported
 * with this communicator / runtime (the plan then stays kernel by kernel — same results) */
int cem_planner_graph_status(const cem_planner_t *h, int32_t *status_out);
/* Kernel launches one CEM iteration of cem_planner_plan takes on this handle (the collective of a sharded plan not counted):
 * 2 = rollout (its tiles sample their own action sequences, cem_mpc.py:44-48) + select (which forms the particle mean of the CemMpc
 * objective itself, mpc_policy.py:38-39) — single-rank CemMpc plans whose tiles are all resident at once; + 1 where the sampler is a
 * launch of its own (tiles queue for slots), + 1 where the reduce kernel stays (SafeCemMpc's Beta filter, sharded plans, the
 * multi-workgroup selects), + 1 for select_mode 3's recovery kernel (returns at once unless a barrier expired), + 7 for select_mode 2's
 * chain.  The stepwise calls always launch the reduce kernel. */
int cem_planner_launches_per_iteration(const cem_planner_t *h, int32_t *launches_out);

/* TransitionModel.unfold_sequences (transition_model.py:64-77) as an API of its own:
 * s0[B][obs], actions[B][H][A] (device) -> traj[B][H+1][obs] (device); optional mu/stddev[B][H][obs].
 * Row r uses member r / (B/E).  Noise: eps_model_dev[H][B][obs] or Philox (seed, call). */
int cem_unfold_sequences(cem_planner_t *h, const float *s0_dev, const float *actions_dev, int32_t n_rows, int32_t horizon,
                         const float *eps_model_dev, uint64_t seed, uint64_t call,
                         float *traj_out_dev, float *mu_out_dev, float *sd_out_dev);

/* MpcPolicy.compute_objective (mpc_policy.py:26-39) / SafeCemMpc.compute_objective (safe_cem_mpc.py:76-96) as an op of its
 * own, on a GIVEN trajectory tensor: traj[n_rows][horizon+1][obs] (device), row r = p * (n_rows / particles) + candidate
 * (the tf.tile order of cem_mpc.py:49-51) -> scores[n_rows / particles] (device).  Uses the handle's variant, particles,
 * posterior threshold and scorer; `horizon` need not be the handle's.  The planner's own rollouts never call this (their
 * objective is the rollout kernel's epilogue and the trajectory is never materialised); it serves callers that hold a
 * trajectory tensor, e.g. from cem_unfold_sequences. */
int cem_compute_objective(cem_planner_t *h, const float *traj_dev, int32_t n_rows, int32_t horizon, float *scores_out_dev);

/* MbrlSafetyGym.get_reward / get_cost (safety_gym.py:62-66) -> SafetyGymStateScorer.reward / cost (:110-166), 'goal' task:
 * obs[n][obs], next_obs[n][obs] (device) -> reward[n] (float), goal_achieved[n] (uint8; may be NULL); obs -> cost[n] (float). */
int cem_scorer_reward(cem_planner_t *h, const float *obs_dev, const float *next_obs_dev, int32_t n, float *reward_out_dev,
                      uint8_t *goal_achieved_out_dev);
int cem_scorer_cost(cem_planner_t *h, const float *obs_dev, int32_t n, float *cost_out_dev);

/* dump the Philox streams a (seed, call) plan consumes, in the explicit-tensor layouts above (device pointers; any may be NULL) */
int cem_fill_noise(cem_planner_t *h, uint64_t seed, uint64_t call, float *eps_act_dev, float *eps_model_dev, float *eps_out_dev);

/* The generator behind those streams, word for word (test hook): the four Philox4x32-7 output words of the n counters
 *   (idx0 + i,  t | iteration << 16,  sub | stream << 16,  call & 0xffffffff),  key (seed & 0xffffffff, (seed >> 32) ^ (call >> 32)),
 * written to words_out_dev[n][4] (uint32).  stream: 0 model noise (idx = global batch row, sub = feature quad), 1 action noise
 * (idx = candidate, sub = action quad), 2 output noise (idx = action quad, t = iteration = sub = 0).  Four normals of a counter:
 *   u_k = fl32(fl32(word_k) * 2^-32 + 2^-33);  z0 = r(u0) cos(2 pi u1), z1 = r(u0) sin(2 pi u1), z2 = r(u2) cos(2 pi u3),
 *   z3 = r(u2) sin(2 pi u3),  r(u) = sqrt(-2 ln u)   (tf.random.normal draws of cem_mpc.py:44-47,68 and mlp_ensemble.py:192-193) */
int cem_philox_words(cem_planner_t *h, uint64_t seed, uint64_t call, uint32_t stream, uint32_t iteration, uint32_t t, uint32_t sub,
                     uint32_t idx0, uint32_t n, uint32_t *words_out_dev);

/* device time (ms) of the rollout kernels of the last plan, measured with HIP events on the handle's stream
 * (enabled by cem_planner_set_timing(h, 1); costs one event pair per launch). */
int cem_planner_set_timing(cem_planner_t *h, int32_t enable);
int cem_planner_last_timing(cem_planner_t *h, float *rollout_ms_total, int32_t *rollout_launches, float *select_ms_total);
/* the same plan's other launches: the particle-mean / Beta-filter kernel (where it is a launch of its own) and the sampler launch (where the
 * sampler is not the rollout tiles' prologue); 0 where the plan has no such launch */
int cem_planner_last_timing_detail(cem_planner_t *h, float *reduce_ms_total, float *sampler_ms_total);

/* ---------------------------------------------------------------------------------------------------------------
 * Ensemble training on the device: MlpEnsemble.training_step / validation_step
 * (simba/models/mlp_ensemble.py:134-155), loss negative_log_likelihood (:64-67), optimizer
 * tf.keras.optimizers.Adam(lr, clipvalue=1.0, epsilon=1e-5) (:113-117).  The shuffling / batching / learning-rate
 * schedule loop of fit() (:163-187, :70-88) is host logic above this ABI.  Weights use the natural blob layout above,
 * so the result of training feeds cem_planner_set_weights() unchanged. */
typedef struct cem_train_config {
    int32_t abi_version;
    int32_t inputs_dim, outputs_dim, units, n_layers, ensemble_size;
    int32_t batch_size;           /* rows per member per step, <= 64 (config/models.yaml:4) */
    int32_t activation;           /* enum cem_activation */
    float dropout_rate;           /* mlp_params['dropout_rate'] (config/models.yaml:13; the shipped value is 0): Dropout after every hidden layer in
                                   * training_step only (mlp_ensemble.py:15,21,138); 0 <= rate < 1.  The keep mask of training step s (0-based,
                                   * counted from cem_trainer_create — cem_trainer_set_state does not restart it, so re-staged weights do not replay masks) is a pure function of (dropout_seed, s, member,
                                   * layer, row of the minibatch, unit): cem_train.h GemmEpi */
    uint32_t dropout_seed_lo, dropout_seed_hi;
    float beta1, beta2, epsilon, clipvalue;
} cem_train_config_t;
typedef struct cem_trainer cem_trainer_t;

size_t cem_trainer_workspace_bytes(const cem_train_config_t *cfg);
size_t cem_trainer_blob_floats(const cem_train_config_t *cfg);
int cem_trainer_create(const cem_train_config_t *cfg, void *workspace, size_t workspace_bytes, void *hip_stream, cem_trainer_t **out);
int cem_trainer_destroy(cem_trainer_t *h);
/* weights + Adam moments (host blobs; moments may be NULL = zeros) */
int cem_trainer_set_state(cem_trainer_t *h, const float *weights, const float *m, const float *v);
int cem_trainer_get_state(cem_trainer_t *h, float *weights, float *m, float *v);
/* one training_step on rows perm[member][offset .. offset+bt) of x_dev[n][inputs_dim] / y_dev[n][outputs_dim];
 * lr_t = lr * sqrt(1-beta2^t)/(1-beta1^t) (Keras folds the bias correction into the step size);
 * loss_dev[ensemble_size] receives every member's share of the loss (their sum is training_step's return value) */
int cem_trainer_step(cem_trainer_t *h, const float *x_dev, const float *y_dev, const int32_t *perm_dev, int32_t nperm,
                     int32_t offset, int32_t bt, float lr_t, float *loss_dev);
/* n_steps consecutive training_steps in ONE call (an epoch of MlpEnsemble.fit's inner loop, mlp_ensemble.py:174-180): step s uses
 * rows perm[member][offsets[s] .. offsets[s] + bts[s]) with step size lr_ts[s] (host arrays) and writes its members' losses to
 * loss_dev[s * ensemble_size ..].  Same arithmetic as n_steps cem_trainer_step calls; the point is one host call per epoch. */
int cem_trainer_steps(cem_trainer_t *h, const float *x_dev, const float *y_dev, const int32_t *perm_dev, int32_t nperm,
                      int32_t n_steps, const int32_t *offsets, const int32_t *bts, const float *lr_ts, float *loss_dev);
/* validation_step on rows [0, n) of x_dev / y_dev: *loss_out = sum over members of NLL / ensemble_size (synchronises) */
int cem_trainer_eval(cem_trainer_t *h, const float *x_dev, const float *y_dev, int32_t n, float *loss_out);

#ifdef __cplusplus
}
#endif
#endif /* CEM_MPC_H */
