#!/usr/bin/env python3
"""The C ABI of include/cem_mpc.h driven with nothing but ctypes + a torch tensor for device memory: what a maintainer's
binding does, without this repo's planner.py / policy classes.

    python examples/capi_minimal.py            # needs an MI355X; prints one planned action
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ethz_safe_learning_amd._capi import CemConfig, load          # ctypes mirror of cem_config_t + dlopen of the .so


def main(seed=0):
    import torch
    lib = load()
    O, A, E, U, L = 60, 2, 5, 128, 4
    cfg = CemConfig()
    cfg.abi_version = lib.cem_abi_version()
    cfg.obs_dim, cfg.act_dim, cfg.units, cfg.n_layers, cfg.ensemble_size = O, A, U, L, E
    cfg.particles, cfg.n_samples, cfg.horizon, cfg.n_elite, cfg.iterations = E, 400, 12, 40, 4
    cfg.smoothing, cfg.stddev_threshold, cfg.noise_stddev = 0.1, 0.0, 0.01
    cfg.one_minus_smoothing = float(np.float32(1.0 - 0.1))                     # fl32 of the Python-float difference (cem_mpc.py:64-65)
    cfg.variant, cfg.posterior_mean_threashold = 0, 0.15                       # 0 = CemMpc objective, 1 = SafeCemMpc
    cfg.sampling_propagation, cfg.scale_features = 1, 1
    for a in range(A):
        cfg.act_lb[a], cfg.act_ub[a], cfg.act_mu0[a], cfg.act_sigma0[a] = -1.0, 1.0, 0.0, 1.0     # MpcPolicy.sampling_params
    sc = cfg.scorer                                                            # SafetyGymStateScorer, PointGoal1 layout
    sc.goal_mode, sc.goal_lo, sc.goal_hi = 0, 3, 19                            # goal lidar bins
    sc.lidar_max_dist, sc.goal_size, sc.reward_distance, sc.reward_goal, sc.reward_clip = 3.0, 0.3, 1.0, 1.0, 10.0
    sc.goal_reached_dist = float(np.float32(0.3 * 0.8))                        # fl32 of the Python-float product (safety_gym.py:116)
    sc.constrain_indicator, sc.n_cost_kinds = 1, 1
    sc.cost_lo[0], sc.cost_hi[0], sc.cost_size[0] = 22, 38, 0.2                # hazards lidar bins
    cfg.world_size, cfg.rank, cfg.chunks_per_tile, cfg.use_graph = 1, 0, 0, 1

    nbytes = lib.cem_workspace_bytes(C.byref(cfg))
    assert nbytes > 0, 'configuration rejected'
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device='cuda:0')         # the caller owns the device memory
    ws_ptr = (ws.data_ptr() + 255) & ~255
    handle = C.c_void_p()
    rc = lib.cem_planner_create(C.byref(cfg), C.c_void_p(ws_ptr), nbytes, None, C.byref(handle))
    assert rc == 0, lib.cem_status_string(rc).decode()

    # weights in Keras order per member: W_0[O+A][U], b_0[U], ..., W_mu[U][O], b_mu[O], W_var[U][O], b_var[O]
    rng = np.random.default_rng(seed)
    blob = []
    for _ in range(E):
        dims = [O + A] + [U] * L
        for i in range(L):
            lim = np.sqrt(6.0 / (dims[i] + dims[i + 1]))
            blob += [rng.uniform(-lim, lim, (dims[i], dims[i + 1])), np.zeros(dims[i + 1])]
        blob += [0.05 * rng.uniform(-0.2, 0.2, (U, O)), np.zeros(O), 0.05 * rng.uniform(-0.2, 0.2, (U, O)), np.full(O, -8.0)]
    blob = np.concatenate([b.ravel() for b in blob]).astype(np.float32)
    assert blob.size == lib.cem_weight_blob_floats(C.byref(cfg))
    assert lib.cem_planner_set_weights(handle, blob.ctypes.data_as(C.c_void_p), blob.size) == 0
    lo = np.concatenate([np.zeros(O), -np.ones(A)]).astype(np.float32)         # TransitionModel.inputs_min / inputs_max
    hi = np.ones(O + A, np.float32)
    assert lib.cem_planner_set_normaliser(handle, lo.ctypes.data_as(C.c_void_p), hi.ctypes.data_as(C.c_void_p)) == 0

    state = rng.uniform(0.2, 0.8, O).astype(np.float32)
    action = np.empty(A, np.float32)
    score, iters = C.c_float(), C.c_int32()
    for call in range(3):                                                      # generate_action(state), three control steps
        rc = lib.cem_planner_plan(handle, state.ctypes.data_as(C.c_void_p), C.c_uint64(seed), C.c_uint64(call), None, None, None,
                                  action.ctypes.data_as(C.c_void_p), C.byref(score), C.byref(iters))
        assert rc == 0, lib.cem_status_string(rc).decode()
    lib.cem_planner_destroy(handle)
    print('action', action, 'score %.4f after %d iterations' % (score.value, iters.value))
    return action.copy(), float(score.value), int(iters.value)


if __name__ == '__main__':
    main()
