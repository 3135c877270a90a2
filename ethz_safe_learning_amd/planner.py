"""Thin Python host over the C ABI: device memory and streams come from
torch-ROCm (plumbing), every computation happens in libcem_mpc_gfx950.so.

``CemPlanner`` is what the simba-shaped policies (``simba/policies``) hold; it
corresponds to one compiled ``@tf.function`` graph of the reference
(simba/policies/cem_mpc.py:35) for one set of shapes.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi


@dataclass
class ScorerConfig:
    """SafetyGymStateScorer fields for the 'goal' task
    (reference simba/environment_utils/safety_gym.py:104-176)."""
    goal_slice: Tuple[int, int]
    observe_goal_lidar: bool = True
    lidar_max_dist: float = 4.0
    goal_size: float = 0.3
    reward_distance: float = 1.0
    reward_goal: float = 1.0
    reward_clip: float = 10.0
    constrain_indicator: bool = True
    cost_kinds: List[Tuple[int, int, float]] = field(default_factory=list)   # (lo, hi, size), reference order


@dataclass
class PlannerConfig:
    """CemMpc/SafeCemMpc ctor kwargs (cem_mpc.py:7-17, safe_cem_mpc.py:8-19) +
    model dims (config/models.yaml) + sharding."""
    obs_dim: int
    act_dim: int
    ensemble_size: int
    particles: int
    n_samples: int
    horizon: int
    n_elite: int
    iterations: int
    scorer: ScorerConfig
    act_low: Sequence[float]
    act_high: Sequence[float]
    units: int = 128
    n_layers: int = 4
    activation: str = 'relu'           # mlp_params['activation'] (config/models.yaml:12): relu | tanh | sigmoid | elu | leaky_relu | softplus | selu | swish | gelu (see ACTIVATIONS)
    smoothing: float = 0.0
    stddev_threshold: float = -1.0
    noise_stddev: float = 0.0
    variant: str = 'cem'
    posterior_mean_threashold: float = 0.15
    sampling_propagation: bool = True
    scale_features: bool = True
    world_size: int = 1
    rank: int = 0
    chunks_per_tile: int = 0
    use_graph: bool = False
    rollout_segments: int = 0          # 0 auto, 1 off, n > 1: horizon-segment work queue (cem_mpc.h)
    precision: str = 'fp32'            # 'fp32' | 'bf16x3' (enum cem_precision: exact three-way bf16 split products, opt-in)
    select_mode: int = 0               # 0 auto, 1 one-workgroup select, 2 multi-workgroup chain, 3 the chain fused into one launch (cem_mpc.h)


# mlp_params['activation'] is a string the reference `eval`s (mlp_ensemble.py:14): the TensorFlow names that map onto enum cem_activation
ACTIVATIONS = {'relu': 0, 'tanh': 1, 'sigmoid': 2, 'elu': 3, 'leaky_relu': 4, 'softplus': 5, 'selu': 6, 'swish': 7, 'silu': 7, 'gelu': 8}


def activation_code(name) -> int:
    """'tf.nn.relu' / 'tf.nn.tanh' / 'tf.math.tanh' / 'tf.keras.activations.elu' / 'tf.nn.swish' / 'relu' ... -> enum cem_activation.  Raises
    for anything else.  (swish / silu and gelu are not monotone — their derivative is not a function of the layer's output — so the
    device trainer keeps the pre-activations of those layers; gelu is TensorFlow's default exact form, approximate=False.)"""
    key = str(name).strip().split('.')[-1].lower()
    if key not in ACTIVATIONS:
        raise NotImplementedError("activation %r is not built (supported: %s — as bare names or with a tf.nn. / tf.math. / "
                                  "tf.keras.activations. prefix)" % (name, ', '.join(sorted(ACTIVATIONS))))
    return ACTIVATIONS[key]


def sampling_params(low, high):
    """MpcPolicy.sampling_params (reference simba/policies/mpc_policy.py:45-57)."""
    low = np.asarray(low, np.float32)
    high = np.asarray(high, np.float32)
    if np.all(np.isfinite(low)) and np.all(np.isfinite(high)):
        return low, high, (high + low) / np.float32(2.0), (high - low) / np.float32(2.0)
    a = low.shape[0]
    return (np.full(a, -100, np.float32), np.full(a, 100, np.float32), np.zeros(a, np.float32),
            np.full(a, 100, np.float32))


def to_c_config(cfg: PlannerConfig) -> _capi.CemConfig:
    c = _capi.CemConfig()
    c.abi_version = _capi.CEM_ABI_VERSION
    c.obs_dim, c.act_dim, c.units, c.n_layers = cfg.obs_dim, cfg.act_dim, cfg.units, cfg.n_layers
    c.activation = activation_code(cfg.activation)
    c.ensemble_size, c.particles, c.n_samples = cfg.ensemble_size, cfg.particles, cfg.n_samples
    c.horizon, c.n_elite, c.iterations = cfg.horizon, cfg.n_elite, cfg.iterations
    c.smoothing, c.stddev_threshold, c.noise_stddev = cfg.smoothing, cfg.stddev_threshold, cfg.noise_stddev
    # `(1.0 - self.smoothing)` is a Python-float difference that TF converts once to fp32 (cem_mpc.py:64-65)
    c.one_minus_smoothing = float(np.float32(1.0 - float(cfg.smoothing)))
    if cfg.variant not in ('cem', 'safe'):
        raise ValueError("variant must be 'cem' or 'safe'")
    c.variant = 1 if cfg.variant == 'safe' else 0
    c.posterior_mean_threashold = cfg.posterior_mean_threashold
    c.sampling_propagation = int(bool(cfg.sampling_propagation))
    c.scale_features = int(bool(cfg.scale_features))
    if cfg.act_dim > _capi.CEM_MAX_ACT:
        raise ValueError('act_dim > %d' % _capi.CEM_MAX_ACT)
    lb, ub, mu0, sg0 = sampling_params(cfg.act_low, cfg.act_high)
    if lb.shape != (cfg.act_dim,):
        raise ValueError('act_low/act_high must have shape [act_dim]')
    for a in range(cfg.act_dim):
        c.act_lb[a], c.act_ub[a], c.act_mu0[a], c.act_sigma0[a] = float(lb[a]), float(ub[a]), float(mu0[a]), float(sg0[a])
    s = cfg.scorer
    c.scorer.goal_mode = 0 if s.observe_goal_lidar else 1
    c.scorer.goal_lo, c.scorer.goal_hi = int(s.goal_slice[0]), int(s.goal_slice[1])
    c.scorer.lidar_max_dist, c.scorer.goal_size = s.lidar_max_dist, s.goal_size
    # goal_achieved = dist <= self.goal_size * 0.8 (safety_gym.py:116): a Python-float product converted to an fp32 tensor
    c.scorer.goal_reached_dist = float(np.float32(float(s.goal_size) * 0.8))
    c.scorer.reward_distance, c.scorer.reward_goal = s.reward_distance, s.reward_goal
    c.scorer.reward_clip = float(s.reward_clip) if s.reward_clip else 0.0
    c.scorer.constrain_indicator = int(bool(s.constrain_indicator))
    if len(s.cost_kinds) > _capi.CEM_MAX_COST_KINDS:
        raise ValueError('too many cost kinds')
    c.scorer.n_cost_kinds = len(s.cost_kinds)
    for i, (lo, hi, size) in enumerate(s.cost_kinds):
        c.scorer.cost_lo[i], c.scorer.cost_hi[i], c.scorer.cost_size[i] = int(lo), int(hi), float(size)
    c.world_size, c.rank, c.chunks_per_tile, c.use_graph = cfg.world_size, cfg.rank, cfg.chunks_per_tile, int(cfg.use_graph)
    c.rollout_segments = int(cfg.rollout_segments)
    c.precision = {'fp32': 0, 'bf16x3': 1}[cfg.precision]
    c.select_mode = int(cfg.select_mode)
    return c


def flatten_weights(weights) -> np.ndarray:
    """Keras-layout per-member weights -> the natural blob of cem_mpc.h:
    W_0,b_0,...,W_{L-1},b_{L-1},W_mu,b_mu,W_var,b_var per member, [in][out] row-major."""
    parts = []
    for w in weights:
        for W, b in zip(w['W'], w['b']):
            parts += [np.asarray(W, np.float32).ravel(), np.asarray(b, np.float32).ravel()]
        parts += [np.asarray(w['W_mu'], np.float32).ravel(), np.asarray(w['b_mu'], np.float32).ravel(),
                  np.asarray(w['W_var'], np.float32).ravel(), np.asarray(w['b_var'], np.float32).ravel()]
    return np.ascontiguousarray(np.concatenate(parts))


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class CemPlanner:
    """One planner handle (fixed shapes) on one GPU."""

    def __init__(self, cfg: PlannerConfig, device='cuda:0'):
        import torch
        self._torch = torch
        self.lib = _capi.load()                       # raises if the HIP extension is missing
        if not torch.cuda.is_available():
            raise RuntimeError('CemPlanner needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path')
        self.cfg = cfg
        self.ccfg = to_c_config(cfg)
        self.device = torch.device(device)
        nbytes = self.lib.cem_workspace_bytes(C.byref(self.ccfg))
        if nbytes == 0:
            # let create() report the precise status
            nbytes = 256
        with torch.cuda.device(self.device):
            self.workspace = torch.zeros(nbytes + 256, dtype=torch.uint8, device=self.device)
            off = (-self.workspace.data_ptr()) % 256
            self._ws_view = self.workspace[off:off + nbytes]
            # a stream of the planner's own: the legacy default stream cannot be captured into a hipGraph
            self.stream = torch.cuda.Stream(device=self.device)
            torch.cuda.synchronize(self.device)         # workspace zero-fill (default stream) before the library uses it
            h = C.c_void_p()
            _capi.check(self.lib.cem_planner_create(C.byref(self.ccfg), _ptr(self._ws_view), nbytes,
                                                    C.c_void_p(self.stream.cuda_stream), C.byref(h)), 'cem_planner_create')
        self.h = h
        lay = _capi.CemLayout()
        _capi.check(self.lib.cem_planner_layout(self.h, C.byref(lay)), 'cem_planner_layout')
        self.layout = lay
        self._call = 0
        self.has_comm = False
        # the generate_action hot path: staging buffers and their ctypes views are made once (a.ctypes.data_as and the two small
        # numpy allocations were 12 of the 17 us the wrapper added to a 1.9-ms plan)
        self._st_buf = np.zeros(cfg.obs_dim, np.float32)
        self._act_buf = np.zeros(cfg.act_dim, np.float32)
        self._st_ptr, self._act_ptr = _np_ptr(self._st_buf), _np_ptr(self._act_buf)
        self._score, self._iters = C.c_float(), C.c_int32()
        self._score_ref, self._iters_ref = C.byref(self._score), C.byref(self._iters)

    # ------------------------------------------------------------------ stream plumbing
    def _wait_inputs(self):
        """Order the planner's stream after whatever torch's current stream has queued (input tensors)."""
        self.stream.wait_stream(self._torch.cuda.current_stream(self.device))

    def stream_context(self):
        """Context in which torch ops (the RCCL collective on the score buffers) run on the planner's stream."""
        return self._torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    # ------------------------------------------------------------------ views
    # Views into the workspace (no copy).  plan() may return before the planner's stream has drained — it watches the pinned result
    # block, not the stream (cem_mpc.h, cem_planner_plan) — and these arrays are written by the kernels behind that result, on a
    # stream torch's current stream knows nothing about: every accessor therefore drains the planner's stream first.  sync=False is for
    # callers that enqueue work on the planner's own stream (`with planner.stream_context():`), where stream order already holds.
    def _view(self, off, count, dtype, sync=True):
        t = self._torch
        if sync:
            self.stream.synchronize()
        nb = count * t.tensor([], dtype=dtype).element_size()
        return self._ws_view[off:off + nb].view(dtype)

    @property
    def n_local(self):
        return self.cfg.n_samples // self.cfg.world_size

    def scores_local(self, sync=True):
        return self._view(self.layout.scores_local, self.n_local, self._torch.float32, sync)

    def scores_global(self, sync=True):
        return self._view(self.layout.scores_global, self.cfg.n_samples, self._torch.float32, sync)

    def actions(self, sync=True):
        c = self.cfg
        return self._view(self.layout.actions, c.n_samples * c.horizon * c.act_dim, self._torch.float32, sync).view(
            c.n_samples, c.horizon, c.act_dim)

    def mu_sigma(self, sync=True):
        c = self.cfg
        return self._view(self.layout.mu_sigma, 2 * c.horizon * c.act_dim, self._torch.float32, sync).view(2, c.horizon, c.act_dim)

    def elite_idx(self, sync=True):
        return self._view(self.layout.elite_idx, self.cfg.n_elite, self._torch.int32, sync)

    def returns(self, sync=True):
        c = self.cfg
        return self._view(self.layout.returns, c.particles * self.n_local, self._torch.float32, sync).view(c.particles, self.n_local)

    def costs(self, sync=True):
        c = self.cfg
        return self._view(self.layout.costs, c.horizon * c.particles * self.n_local, self._torch.uint8, sync).view(
            c.horizon, c.particles, self.n_local)

    def result_block(self, sync=True):
        """The last completed plan's result as the device keeps it (cem_layout_t.result): uint32 [38] — [0, A) action bits, [32] score
        bits, [33] iterations, [34] early-stop flag, [35] fault bits, [36] plan counter, [37] checksum."""
        return self._view(self.layout.result, 38, self._torch.int32, sync)

    # ------------------------------------------------------------------ sync hooks
    def set_weights(self, weights):
        blob = flatten_weights(weights)
        expect = self.lib.cem_weight_blob_floats(C.byref(self.ccfg))
        if blob.size != expect:
            raise ValueError('weight blob has %d floats, expected %d' % (blob.size, expect))
        _capi.check(self.lib.cem_planner_set_weights(self.h, _np_ptr(blob), blob.size), 'cem_planner_set_weights')

    def set_normaliser(self, inputs_min, inputs_max):
        mn = np.ascontiguousarray(np.asarray(inputs_min, np.float32))
        mx = np.ascontiguousarray(np.asarray(inputs_max, np.float32))
        if mn.shape != (self.cfg.obs_dim + self.cfg.act_dim,) or mx.shape != mn.shape:
            raise ValueError('normaliser must have shape [obs_dim + act_dim]')
        _capi.check(self.lib.cem_planner_set_normaliser(self.h, _np_ptr(mn), _np_ptr(mx)), 'cem_planner_set_normaliser')

    # ------------------------------------------------------------------ native exchange (RCCL inside the library)
    def comm_init(self, group=None):
        """Give the handle its own RCCL communicator over the ranks of ``group`` (default: the world), so that ``plan()`` runs
        the whole candidate-sharded plan — kernels and the per-iteration all-gather of the scores — inside the library, as one
        hipGraph per rank when ``use_graph`` is set.  Collective: every rank of the group calls it.  torch.distributed only
        carries the 128-byte communicator id from rank 0 to the others; with world_size 1 it is not needed at all."""
        c = self.cfg
        buf = (C.c_char * _capi.CEM_COMM_ID_BYTES)()
        if c.world_size > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError('comm_init for world_size > 1 needs an initialised torch.distributed process group (it carries the id)')
            if dist.get_world_size(group) != c.world_size or dist.get_rank(group) != c.rank:
                raise ValueError('the process group does not match the planner\'s (world_size, rank)')
            box = [None]
            if c.rank == 0:
                _capi.check(self.lib.cem_comm_unique_id(buf), 'cem_comm_unique_id')
                box[0] = bytes(buf.raw)
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            buf.raw = box[0]
        else:
            _capi.check(self.lib.cem_comm_unique_id(buf), 'cem_comm_unique_id')
        with self._torch.cuda.device(self.device):
            _capi.check(self.lib.cem_planner_comm_init(self.h, buf, c.world_size, c.rank), 'cem_planner_comm_init')
        self.has_comm = True

    def comm_ranks(self):
        """Ranks of the handle's RCCL communicator as RCCL reports them (0 without one)."""
        n = C.c_int32()
        _capi.check(self.lib.cem_planner_comm_ranks(self.h, C.byref(n)), 'cem_planner_comm_ranks')
        return n.value

    def comm_destroy(self):
        _capi.check(self.lib.cem_planner_comm_destroy(self.h), 'cem_planner_comm_destroy')
        self.has_comm = False

    def graph_status(self):
        """'eager' | 'graph' | 'graph-unsupported' (cem_planner_graph_status)."""
        st = C.c_int32()
        _capi.check(self.lib.cem_planner_graph_status(self.h, C.byref(st)), 'cem_planner_graph_status')
        return ('eager', 'graph', 'graph-unsupported')[st.value]

    def launches_per_iteration(self):
        """Kernel launches one CEM iteration of plan() takes on this handle (cem_planner_launches_per_iteration)."""
        n = C.c_int32()
        _capi.check(self.lib.cem_planner_launches_per_iteration(self.h, C.byref(n)), 'cem_planner_launches_per_iteration')
        return n.value

    def plan_exchange(self):
        _capi.check(self.lib.cem_plan_exchange(self.h), 'cem_plan_exchange')

    def select_mode(self):
        """The select form the next iteration takes on this handle: 1 one workgroup, 2 the multi-launch chain, 3 the chain fused into one
        launch (cem_planner_select_mode; 2 for good once a fused select had to be recovered)."""
        m = C.c_int32()
        _capi.check(self.lib.cem_planner_select_mode(self.h, C.byref(m)), 'cem_planner_select_mode')
        return m.value

    def inject_fault(self, kind=1):
        """Test hook (cem_planner_inject_fault): the next plan's first fused select sees one of its grid barriers expire."""
        _capi.check(self.lib.cem_planner_inject_fault(self.h, kind), 'cem_planner_inject_fault')

    # ------------------------------------------------------------------ planning
    def _noise_args(self, eps_act, eps_model):
        c = self.cfg
        t = self._torch
        if eps_act is None and eps_model is None:
            return None, None
        if eps_act is None or eps_model is None:
            raise ValueError('eps_act and eps_model must be given together')
        B = c.particles * c.n_samples
        ea = t.as_tensor(eps_act, dtype=t.float32, device=self.device).contiguous()
        em = t.as_tensor(eps_model, dtype=t.float32, device=self.device).contiguous()
        if tuple(ea.shape) != (c.iterations, c.n_samples, c.horizon, c.act_dim):
            raise ValueError('eps_act must be [I,N,H,A]')
        if tuple(em.shape) != (c.iterations, c.horizon, B, c.obs_dim):
            raise ValueError('eps_model must be [I,H,P*N,O]')
        return ea, em

    def plan(self, state, seed=0, call=None, eps_act=None, eps_model=None, eps_out=None):
        """CemMpc.generate_action (cem_mpc.py:31-33): state[O] -> (action[A], best_score, iters)."""
        if np.shape(state) != self._st_buf.shape:
            raise ValueError('state must have shape [%d]' % self.cfg.obs_dim)
        self._st_buf[:] = state                                     # (float64 observations are cast here, as cem_mpc.py:32 does)
        if call is None:
            call = self._call
            self._call += 1
        if eps_act is None and eps_model is None and eps_out is None:     # the generator path: nothing else to marshal
            st = self.lib.cem_planner_plan(self.h, self._st_ptr, seed, call, None, None, None, self._act_ptr, self._score_ref, self._iters_ref)
            if st:
                _capi.check(st, 'cem_planner_plan')
            return self._act_buf.copy(), self._score.value, self._iters.value
        ea, em = self._noise_args(eps_act, eps_model)
        eo = np.ascontiguousarray(np.asarray(eps_out, np.float32)) if eps_out is not None else None
        if ea is not None:
            self._wait_inputs()
        _capi.check(self.lib.cem_planner_plan(self.h, self._st_ptr, seed, call, _ptr(ea), _ptr(em), _np_ptr(eo),
                                              self._act_ptr, self._score_ref, self._iters_ref), 'cem_planner_plan')
        return self._act_buf.copy(), float(self._score.value), int(self._iters.value)

    def plan_begin(self, state, seed=0, call=0, eps_act=None, eps_model=None):
        st = np.ascontiguousarray(np.asarray(state, np.float32))
        ea, em = self._noise_args(eps_act, eps_model)
        self._keep = (ea, em)
        self._wait_inputs()
        _capi.check(self.lib.cem_plan_begin(self.h, _np_ptr(st), seed, call, _ptr(ea), _ptr(em)), 'cem_plan_begin')

    def plan_rollout(self, it):
        _capi.check(self.lib.cem_plan_rollout(self.h, it), 'cem_plan_rollout')

    def plan_select(self, it):
        _capi.check(self.lib.cem_plan_select(self.h, it), 'cem_plan_select')

    def plan_end(self, eps_out=None):
        c = self.cfg
        eo = np.ascontiguousarray(np.asarray(eps_out, np.float32)) if eps_out is not None else None
        action = np.zeros(c.act_dim, np.float32)
        score = C.c_float()
        iters = C.c_int32()
        _capi.check(self.lib.cem_plan_end(self.h, _np_ptr(eo), _np_ptr(action), C.byref(score), C.byref(iters)), 'cem_plan_end')
        self._keep = None
        return action, float(score.value), int(iters.value)

    # ------------------------------------------------------------------ model API
    def unfold_sequences(self, s0, actions, eps_model=None, seed=0, call=0, return_moments=False):
        """TransitionModel.unfold_sequences (transition_model.py:64-77) on device:
        s0 [B,O], actions [B,H,A] -> traj [B,H+1,O] (torch tensors on the GPU)."""
        t = self._torch
        c = self.cfg
        s0 = t.as_tensor(s0, dtype=t.float32, device=self.device).contiguous()
        actions = t.as_tensor(actions, dtype=t.float32, device=self.device).contiguous()
        B, H = actions.shape[0], actions.shape[1]
        if tuple(s0.shape) != (B, c.obs_dim) or actions.shape[2] != c.act_dim:
            raise ValueError('bad shapes for unfold_sequences')
        em = None
        if eps_model is not None:
            em = t.as_tensor(eps_model, dtype=t.float32, device=self.device).contiguous()
            if tuple(em.shape) != (H, B, c.obs_dim):
                raise ValueError('eps_model must be [H,B,O]')
        traj = t.empty((B, H + 1, c.obs_dim), dtype=t.float32, device=self.device)
        mu = t.empty((B, H, c.obs_dim), dtype=t.float32, device=self.device) if return_moments else None
        sd = t.empty((B, H, c.obs_dim), dtype=t.float32, device=self.device) if return_moments else None
        self._wait_inputs()
        _capi.check(self.lib.cem_unfold_sequences(self.h, _ptr(s0), _ptr(actions), B, H, _ptr(em), seed, call,
                                                  _ptr(traj), _ptr(mu), _ptr(sd)), 'cem_unfold_sequences')
        return (traj, mu, sd) if return_moments else traj

    def compute_objective(self, trajectories):
        """MpcPolicy.compute_objective (mpc_policy.py:26-39) / SafeCemMpc.compute_objective (safe_cem_mpc.py:76-96) on a
        given trajectory tensor [P*n, H+1, O] (rows in the tf.tile order p*n + candidate) -> scores [n] (torch, on the GPU)."""
        t = self._torch
        c = self.cfg
        traj = t.as_tensor(trajectories, dtype=t.float32, device=self.device).contiguous()
        if traj.dim() != 3 or traj.shape[2] != c.obs_dim or traj.shape[1] < 2:
            raise ValueError('trajectories must be [P*n, H+1, obs_dim]')
        B, H = traj.shape[0], traj.shape[1] - 1
        if B % c.particles != 0:
            raise ValueError('trajectory rows (%d) are not a multiple of particles (%d)' % (B, c.particles))
        scores = t.empty((B // c.particles,), dtype=t.float32, device=self.device)
        self._wait_inputs()
        _capi.check(self.lib.cem_compute_objective(self.h, _ptr(traj), B, H, _ptr(scores)), 'cem_compute_objective')
        return scores

    def scorer_reward(self, observations, next_observations):
        """SafetyGymStateScorer.reward (safety_gym.py:110-119,140-143): (reward [n] float32, goal_achieved [n] bool)."""
        t = self._torch
        obs = t.as_tensor(observations, dtype=t.float32, device=self.device).contiguous()
        nxt = t.as_tensor(next_observations, dtype=t.float32, device=self.device).contiguous()
        if obs.dim() != 2 or obs.shape[1] != self.cfg.obs_dim or nxt.shape != obs.shape:
            raise ValueError('observations / next_observations must both be [n, obs_dim]')
        r = t.empty((obs.shape[0],), dtype=t.float32, device=self.device)
        g = t.empty((obs.shape[0],), dtype=t.uint8, device=self.device)
        self._wait_inputs()
        _capi.check(self.lib.cem_scorer_reward(self.h, _ptr(obs), _ptr(nxt), obs.shape[0], _ptr(r), _ptr(g)), 'cem_scorer_reward')
        return r, g.bool()

    def scorer_cost(self, observations):
        """SafetyGymStateScorer.cost (safety_gym.py:145-166): cost [n] float32."""
        t = self._torch
        obs = t.as_tensor(observations, dtype=t.float32, device=self.device).contiguous()
        if obs.dim() != 2 or obs.shape[1] != self.cfg.obs_dim:
            raise ValueError('observations must be [n, obs_dim]')
        c = t.empty((obs.shape[0],), dtype=t.float32, device=self.device)
        self._wait_inputs()
        _capi.check(self.lib.cem_scorer_cost(self.h, _ptr(obs), obs.shape[0], _ptr(c)), 'cem_scorer_cost')
        return c

    def fill_noise(self, seed=0, call=0):
        """The Philox streams a (seed, call) plan consumes, as explicit tensors."""
        t = self._torch
        c = self.cfg
        B = c.particles * c.n_samples
        ea = t.empty((c.iterations, c.n_samples, c.horizon, c.act_dim), dtype=t.float32, device=self.device)
        em = t.empty((c.iterations, c.horizon, B, c.obs_dim), dtype=t.float32, device=self.device)
        eo = t.empty((c.act_dim,), dtype=t.float32, device=self.device)
        self._wait_inputs()
        _capi.check(self.lib.cem_fill_noise(self.h, seed, call, _ptr(ea), _ptr(em), _ptr(eo)), 'cem_fill_noise')
        return ea, em, eo

    def philox_words(self, seed, call, stream, iteration, t, sub, idx0, n):
        """The generator's raw Philox4x32-7 output words for n consecutive counters (test hook, cem_mpc.h): uint32 [n, 4]."""
        out = self._torch.empty((n, 4), dtype=self._torch.int32, device=self.device)
        self._wait_inputs()
        _capi.check(self.lib.cem_philox_words(self.h, seed, call, stream, iteration, t, sub, idx0, n, _ptr(out)), 'cem_philox_words')
        return out.cpu().numpy().view('uint32')

    def set_timing(self, enable=True):
        _capi.check(self.lib.cem_planner_set_timing(self.h, int(enable)), 'cem_planner_set_timing')

    def last_timing(self):
        r, n, s = C.c_float(), C.c_int32(), C.c_float()
        _capi.check(self.lib.cem_planner_last_timing(self.h, C.byref(r), C.byref(n), C.byref(s)), 'cem_planner_last_timing')
        rd, sa = C.c_float(), C.c_float()
        _capi.check(self.lib.cem_planner_last_timing_detail(self.h, C.byref(rd), C.byref(sa)), 'cem_planner_last_timing_detail')
        return dict(rollout_ms=float(r.value), rollout_launches=int(n.value), select_ms=float(s.value), reduce_ms=float(rd.value),
                    sampler_ms=float(sa.value))

    def tiles(self):
        """(chunks_per_tile, tiles[n,6]) of this handle's plan (host-side logic, no GPU call)."""
        return plan_tiles(self.cfg)

    def segments(self):
        return plan_segments(self.cfg)

    def close(self):
        if getattr(self, 'h', None):
            self.lib.cem_planner_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_tiles(cfg: PlannerConfig):
    lib = _capi.load()
    cc = to_c_config(cfg)
    rc, nt = C.c_int32(), C.c_int32()
    _capi.check(lib.cem_plan_tiles_host(C.byref(cc), C.byref(rc), C.byref(nt), None, 0), 'cem_plan_tiles_host')
    tiles = np.zeros((nt.value, 6), np.int32)
    _capi.check(lib.cem_plan_tiles_host(C.byref(cc), C.byref(rc), C.byref(nt), _np_ptr(tiles), nt.value), 'cem_plan_tiles_host')
    return rc.value, tiles


def plan_segments(cfg: PlannerConfig):
    """(segments, steps per segment) of the rollout launch this configuration gets (1 segment = unsegmented)."""
    lib = _capi.load()
    cc = to_c_config(cfg)
    ns, sl = C.c_int32(), C.c_int32()
    _capi.check(lib.cem_plan_segments_host(C.byref(cc), C.byref(ns), C.byref(sl)), 'cem_plan_segments_host')
    return ns.value, sl.value


def pack_weights_host(cfg: PlannerConfig, weights) -> np.ndarray:
    lib = _capi.load()
    cc = to_c_config(cfg)
    blob = flatten_weights(weights)
    out = np.zeros(lib.cem_packed_weight_floats(C.byref(cc)), np.float32)
    _capi.check(lib.cem_pack_weights_host(C.byref(cc), _np_ptr(blob), _np_ptr(out)), 'cem_pack_weights_host')
    return out


# ---------------------------------------------------------------------------------------------------------------
# Shape-keyed handle cache: scripts/tune_cem_policy.py replaces agent.policy with fresh CemMpc objects
# of different (H, I, N, k) at run time (reference scripts/tune_cem_policy.py:109-115, "to trigger tensorflow's
# retracing").  A planner handle is this build's "traced graph": one per distinct shape, reused when a shape recurs.
# ---------------------------------------------------------------------------------------------------------------
_PLANNER_CACHE = {}
_PLANNER_CACHE_MAX = 32


def _freeze(v):
    if isinstance(v, np.ndarray):
        return tuple(np.asarray(v, np.float64).ravel().tolist())
    if isinstance(v, (list, tuple)):
        return tuple(_freeze(x) for x in v)
    if hasattr(v, '__dataclass_fields__'):
        return tuple((k, _freeze(getattr(v, k))) for k in v.__dataclass_fields__)
    return v


def config_key(cfg: PlannerConfig, device='cuda:0'):
    return (str(device),) + _freeze(cfg)


def cached_planner(cfg: PlannerConfig, device='cuda:0') -> CemPlanner:
    key = config_key(cfg, device)
    pl = _PLANNER_CACHE.pop(key, None)
    if pl is None:
        pl = CemPlanner(cfg, device=device)
        pl.staged = None                         # (model.uid, model.version) whose weights/normaliser are on the device
        while len(_PLANNER_CACHE) >= _PLANNER_CACHE_MAX:
            # least recently used entry: only the cache's reference goes; a policy still holding the handle keeps it alive
            # (CemPlanner.__del__ destroys it with its last reference)
            _PLANNER_CACHE.pop(next(iter(_PLANNER_CACHE)))
    _PLANNER_CACHE[key] = pl                     # most recently used last
    return pl


def planner_cache_info():
    return dict(size=len(_PLANNER_CACHE), keys=list(_PLANNER_CACHE.keys()))
