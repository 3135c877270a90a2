"""ctypes binding of include/cem_mpc.h (libcem_mpc_gfx950.so).

There is no CPU fallback: if the HIP library is missing or a call fails, this
module raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C ethz_safe_learning_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

CEM_ABI_VERSION = 4
CEM_MAX_ACT = 32
CEM_MAX_COST_KINDS = 4
CEM_COMM_ID_BYTES = 128

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('CEM_MPC_LIB') or os.path.join(_HERE, 'lib', 'libcem_mpc_gfx950.so')   # env override: A/B builds

EXPORTED_SYMBOLS = [
    'cem_abi_version', 'cem_status_string', 'cem_last_hip_error', 'cem_weight_blob_floats',
    'cem_packed_weight_floats', 'cem_workspace_bytes', 'cem_pack_weights_host', 'cem_plan_tiles_host', 'cem_plan_segments_host', 'cem_rollout_residency',
    'cem_planner_create', 'cem_planner_destroy', 'cem_planner_layout', 'cem_planner_set_weights',
    'cem_planner_set_normaliser', 'cem_planner_plan', 'cem_plan_begin', 'cem_plan_rollout', 'cem_plan_select',
    'cem_plan_end', 'cem_comm_unique_id', 'cem_planner_comm_init', 'cem_planner_comm_destroy', 'cem_planner_comm_ranks', 'cem_plan_exchange', 'cem_planner_graph_status', 'cem_planner_launches_per_iteration', 'cem_unfold_sequences', 'cem_compute_objective', 'cem_scorer_reward', 'cem_scorer_cost', 'cem_fill_noise', 'cem_philox_words', 'cem_planner_set_timing', 'cem_planner_last_timing', 'cem_planner_last_timing_detail', 'cem_planner_select_mode', 'cem_planner_inject_fault',
    'cem_trainer_workspace_bytes', 'cem_trainer_blob_floats', 'cem_trainer_create', 'cem_trainer_destroy', 'cem_trainer_set_state',
    'cem_trainer_get_state', 'cem_trainer_step', 'cem_trainer_steps', 'cem_trainer_eval',
]


class CemScorer(C.Structure):
    _fields_ = [
        ('goal_mode', C.c_int32), ('goal_lo', C.c_int32), ('goal_hi', C.c_int32),
        ('lidar_max_dist', C.c_float), ('goal_size', C.c_float), ('goal_reached_dist', C.c_float), ('reward_distance', C.c_float),
        ('reward_goal', C.c_float), ('reward_clip', C.c_float),
        ('constrain_indicator', C.c_int32), ('n_cost_kinds', C.c_int32),
        ('cost_lo', C.c_int32 * CEM_MAX_COST_KINDS), ('cost_hi', C.c_int32 * CEM_MAX_COST_KINDS),
        ('cost_size', C.c_float * CEM_MAX_COST_KINDS),
    ]


class CemConfig(C.Structure):
    _fields_ = [
        ('abi_version', C.c_int32), ('obs_dim', C.c_int32), ('act_dim', C.c_int32),
        ('units', C.c_int32), ('n_layers', C.c_int32), ('activation', C.c_int32), ('ensemble_size', C.c_int32),
        ('particles', C.c_int32), ('n_samples', C.c_int32), ('horizon', C.c_int32),
        ('n_elite', C.c_int32), ('iterations', C.c_int32),
        ('smoothing', C.c_float), ('one_minus_smoothing', C.c_float), ('stddev_threshold', C.c_float), ('noise_stddev', C.c_float),
        ('variant', C.c_int32), ('posterior_mean_threashold', C.c_float),
        ('sampling_propagation', C.c_int32), ('scale_features', C.c_int32),
        ('act_lb', C.c_float * CEM_MAX_ACT), ('act_ub', C.c_float * CEM_MAX_ACT),
        ('act_mu0', C.c_float * CEM_MAX_ACT), ('act_sigma0', C.c_float * CEM_MAX_ACT),
        ('scorer', CemScorer),
        ('world_size', C.c_int32), ('rank', C.c_int32), ('chunks_per_tile', C.c_int32), ('use_graph', C.c_int32),
        ('select_mode', C.c_int32), ('rollout_segments', C.c_int32), ('precision', C.c_int32),
    ]


class CemTrainConfig(C.Structure):
    _fields_ = [('abi_version', C.c_int32), ('inputs_dim', C.c_int32), ('outputs_dim', C.c_int32), ('units', C.c_int32),
                ('n_layers', C.c_int32), ('ensemble_size', C.c_int32), ('batch_size', C.c_int32), ('activation', C.c_int32),
                ('dropout_rate', C.c_float), ('dropout_seed_lo', C.c_uint32), ('dropout_seed_hi', C.c_uint32),
                ('beta1', C.c_float), ('beta2', C.c_float), ('epsilon', C.c_float), ('clipvalue', C.c_float)]


class CemLayout(C.Structure):
    _fields_ = [(n, C.c_size_t) for n in
                ('scores_local', 'scores_global', 'actions', 'mu_sigma', 'elite_idx', 'returns', 'costs', 'result', 'stamps', 'total')]


class CemError(RuntimeError):
    def __init__(self, status, where, lib=None):
        msg = lib.cem_status_string(status).decode() if lib is not None else str(status)
        hip = lib.cem_last_hip_error() if lib is not None else 0
        super().__init__('%s failed: status %d (%s)%s' % (where, status, msg, (', hip error %d' % hip) if status == 5 else ''))
        self.status = status


_lib = None


def load():
    """Load libcem_mpc_gfx950.so or raise (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError('HIP extension %s is missing: run `python -c "import __graft_entry__ as g; g.build()"` '
                          'or `make -C ethz_safe_learning_amd/csrc`. There is no CPU fallback.' % LIB_PATH)
    # The planner's device memory comes from torch, whose wheel carries its own libamdhip64.so.7; the library must bind to THAT
    # runtime.  Loaded before torch it would pull in /opt/rocm's copy through its RUNPATH, torch would then load its own beside it,
    # and every pointer torch hands over would be foreign to the runtime the kernels are launched with (cem_planner_create:
    # hipErrorNoDevice).  So torch first: its runtime is then the one in the process and the soname resolves to it.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    vp, fp, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)
    cfgp = C.POINTER(CemConfig)
    lib.cem_abi_version.restype = C.c_int
    lib.cem_status_string.restype = C.c_char_p
    lib.cem_status_string.argtypes = [C.c_int]
    lib.cem_last_hip_error.restype = C.c_int
    for f in ('cem_weight_blob_floats', 'cem_packed_weight_floats', 'cem_workspace_bytes'):
        getattr(lib, f).restype = C.c_size_t
        getattr(lib, f).argtypes = [cfgp]
    lib.cem_pack_weights_host.argtypes = [cfgp, vp, vp]
    lib.cem_plan_tiles_host.argtypes = [cfgp, i32p, i32p, vp, C.c_int32]
    lib.cem_plan_segments_host.argtypes = [cfgp, i32p, i32p]
    lib.cem_rollout_residency.argtypes = [C.c_int32, C.c_int32, i32p, i32p]
    lib.cem_planner_create.argtypes = [cfgp, vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.cem_planner_destroy.argtypes = [vp]
    lib.cem_planner_layout.argtypes = [vp, C.POINTER(CemLayout)]
    lib.cem_planner_set_weights.argtypes = [vp, vp, C.c_size_t]
    lib.cem_planner_set_normaliser.argtypes = [vp, vp, vp]
    lib.cem_planner_plan.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp, vp, vp, fp, i32p]
    lib.cem_plan_begin.argtypes = [vp, vp, C.c_uint64, C.c_uint64, vp, vp]
    lib.cem_plan_rollout.argtypes = [vp, C.c_int32]
    lib.cem_plan_select.argtypes = [vp, C.c_int32]
    lib.cem_plan_end.argtypes = [vp, vp, vp, fp, i32p]
    lib.cem_unfold_sequences.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, vp, C.c_uint64, C.c_uint64, vp, vp, vp]
    lib.cem_comm_unique_id.argtypes = [vp]
    lib.cem_planner_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
    lib.cem_planner_comm_destroy.argtypes = [vp]
    lib.cem_plan_exchange.argtypes = [vp]
    lib.cem_planner_graph_status.argtypes = [vp, i32p]
    lib.cem_planner_comm_ranks.argtypes = [vp, i32p]
    lib.cem_planner_launches_per_iteration.argtypes = [vp, i32p]
    lib.cem_compute_objective.argtypes = [vp, vp, C.c_int32, C.c_int32, vp]
    lib.cem_scorer_reward.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
    lib.cem_scorer_cost.argtypes = [vp, vp, C.c_int32, vp]
    lib.cem_fill_noise.argtypes = [vp, C.c_uint64, C.c_uint64, vp, vp, vp]
    lib.cem_philox_words.argtypes = [vp, C.c_uint64, C.c_uint64] + [C.c_uint32] * 6 + [vp]
    lib.cem_planner_select_mode.argtypes = [vp, i32p]
    lib.cem_planner_inject_fault.argtypes = [vp, C.c_int32]
    lib.cem_planner_set_timing.argtypes = [vp, C.c_int32]
    lib.cem_planner_last_timing.argtypes = [vp, fp, i32p, fp]
    lib.cem_planner_last_timing_detail.argtypes = [vp, fp, fp]
    tcfgp = C.POINTER(CemTrainConfig)
    for f in ('cem_trainer_workspace_bytes', 'cem_trainer_blob_floats'):
        getattr(lib, f).restype = C.c_size_t
        getattr(lib, f).argtypes = [tcfgp]
    lib.cem_trainer_create.argtypes = [tcfgp, vp, C.c_size_t, vp, C.POINTER(vp)]
    lib.cem_trainer_destroy.argtypes = [vp]
    lib.cem_trainer_set_state.argtypes = [vp, vp, vp, vp]
    lib.cem_trainer_get_state.argtypes = [vp, vp, vp, vp]
    lib.cem_trainer_step.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_float, vp]
    lib.cem_trainer_steps.argtypes = [vp, vp, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    lib.cem_trainer_eval.argtypes = [vp, vp, vp, C.c_int32, fp]
    for name in EXPORTED_SYMBOLS:
        fn = getattr(lib, name)          # raises AttributeError if the symbol is not exported
        if name not in ('cem_status_string', 'cem_weight_blob_floats', 'cem_packed_weight_floats', 'cem_workspace_bytes',
                        'cem_trainer_workspace_bytes', 'cem_trainer_blob_floats'):
            fn.restype = C.c_int
    if lib.cem_abi_version() != CEM_ABI_VERSION:
        raise ImportError('ABI mismatch: library %d, binding %d' % (lib.cem_abi_version(), CEM_ABI_VERSION))
    _lib = lib
    return lib


def check(status, where):
    if status != 0:
        raise CemError(status, where, _lib)
