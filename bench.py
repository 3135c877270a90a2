#!/usr/bin/env python3
"""bench.py — planning steps/sec of the MI355X CEM-MPC planner (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
(one rank per GPU, RCCL).  A "step" is one complete plan = CemMpc.generate_action: I=5 CEM iterations of
sample -> rollout+score -> select/refit, early stop disabled so exactly I iterations run.

Workload: BASELINE config B2 (obs=60, act=2, K=P=E=5, N=2000, H=30, I=5, k=N/10) at N=1.  At G>1 the candidate
count is weak-scaled (N = 2000*G, every rank rolls out 2000 candidates x 5 particles) and `value` is in
B2-equivalent plans/s = candidate-trajectory-steps/s / 300000, so it equals plain plans/s at G=1.  The same line carries the
unscaled rate (`weak_plans_per_s`) and, as labelled extras, the literal metric strong-scaled (`b2_strong`: N = 2000 in total, 2000/G per
rank, plain plans/s) and BASELINE config 5 (`b5`: N = 65536 over the G ranks).  A single-GPU run adds `configs`: B1 / B3 / B4 whole plans,
one rank of B5's eight, and the reference's own policies — SafeCemMpc at the headline shape (`B2_safe`) and the shapes it ships
(`shipped_safe_cem_mpc`, `shipped_cem_mpc`, each with early stop off and as shipped).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_MFMA_TFLOPS = 2516.6     # MI355X_MICROARCH.md: dense bf16 MFMA = 16 x the fp32 matrix rate (the split leg's pipe)
SHADER_CLOCK_HZ = 2.4e9            # MI355X peak engine clock (MI355X_MICROARCH.md)


def source_sha16():
    """Identity of the device code a profile was collected on: sha256 over the kernel sources and the C ABI header.  The PMC
    summaries under profiles/ carry it (scripts/summarise_profiles.py); a bench run only quotes counters whose hash matches."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'ethz_safe_learning_amd', 'csrc', '*.h')) + glob.glob(os.path.join(ROOT, 'ethz_safe_learning_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, 'include', '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def rollout_kernel_name(pl, obs, act):
    """The instantiation this plan launches, as rocprofv3 prints it (cem_capi.hip: launch_rollout / launch_rollout_seg)."""
    rc = pl.tiles()[0]
    nfw = ((obs + act + 15) // 16 + 3) // 4
    if pl.segments()[0] > 1:
        return 'void cem_rollout_seg_kernel<%d, %d>(RolloutParams)' % (rc, nfw)
    return 'void cem_rollout_kernel<%d, %d, 0>(RolloutParams)' % (rc, nfw)


def split_leg(torch, pb, dev, steps, warmup, n_per_gpu):
    """The SAME workload on the opt-in split-product rollout (PlannerConfig.precision = 'bf16x3', csrc/cem_rollout_split.h): every fp32
    product formed from exact three-way bf16 splits as six bf16 MFMAs, fp32 accumulate — scores within the fp32 kernels' own distance
    of the oracle (tests/test_gpu_split.py).  A separately labelled line: never part of `value`, which is the fp32-MFMA path."""
    from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
    obs, act, K, H, I, N = 60, 2, 5, 30, 5, n_per_gpu
    cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=N // 10, iterations=I,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=-1.0, noise_stddev=1e-3, variant='cem',
                        precision='bf16x3', use_graph=True)
    pl = CemPlanner(cfg, device=dev)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(warmup):
        pl.plan(pb['state'], seed=2026, call=i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        a, s, it = pl.plan(pb['state'], seed=2026, call=warmup + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert it == I and np.all(np.isfinite(a))
    pl.set_timing(True)
    roll_ms, roll_n = 0.0, 0
    for i in range(5):
        pl.plan(pb['state'], seed=2027, call=i)
        tm = pl.last_timing()
        roll_ms += tm['rollout_ms']; roll_n += tm['rollout_launches']
    pl.set_timing(False)
    avg_ms = roll_ms / max(roll_n, 1)
    flops_launch = synthetic.flops_per_row_step(obs, act) * K * N * H
    rc = pl.tiles()[0]
    out = dict(label='opt-in: fp32 products as six bf16 MFMAs of exact three-way splits (precision bf16x3); NOT the headline value',
               value=steps / dt, unit='plans/s', ms_per_step=1e3 * dt / steps, steps=steps, chunks_per_tile=rc, tiles=int(len(pl.tiles()[1])),
               kernel='void cem_rollout_split_kernel<%d, 1, 0>(RolloutParams)' % rc, rollout_ms_per_launch=avg_ms,
               algorithmic_tflops=flops_launch / (avg_ms * 1e-3) / 1e12,
               matrix_pipe_tflops=flops_launch * 6.0 / (avg_ms * 1e-3) / 1e12, matrix_pipe_peak_tflops=PEAK_BF16_MFMA_TFLOPS,
               matrix_pipe_frac=flops_launch * 6.0 / (avg_ms * 1e-3) / 1e12 / PEAK_BF16_MFMA_TFLOPS,
               candidate_trajectory_steps_per_s=steps / dt * I * N * H, hip_graph=pl.graph_status() == 'graph')
    pl.close()
    return out


def cpu_baseline(budget_s=25.0):
    """The repo's own CPU path as SURVEY.md 8d defines it: the oracle in its fast form (oracle/cem_oracle_fast.py, a port:
    torch-CPU fp32, one batched matmul over members per layer), timed on this box's host cores on a bounded sample of the
    same synthetic workload: 1 warm-up + >= 3 whole plans of B2 (the headline config) and of B1 (the reference-scale config),
    model noise drawn inside the plan as the reference does."""
    import torch
    from oracle import cem_oracle as o      # cpu_baseline leg only
    from oracle import cem_oracle_fast as of
    from ethz_safe_learning_amd import synthetic
    obs, act, K, I = 60, 2, 5, 5
    pb = synthetic.problem(obs, act, K)
    sp = pb['scorer']
    osp = o.ScorerParams(goal_slice=sp.goal_slice, observe_goal_lidar=sp.observe_goal_lidar, lidar_max_dist=sp.lidar_max_dist,
                         goal_size=sp.goal_size, reward_distance=sp.reward_distance, reward_goal=sp.reward_goal,
                         reward_clip=sp.reward_clip, constrain_indicator=sp.constrain_indicator, cost_kinds=list(sp.cost_kinds))
    sw = of.stack_weights(pb['weights'])
    default_threads = torch.get_num_threads()

    def run_plan(N, H, gen):
        cfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=N // 10, particles=K, ensemble_size=K,
                           stddev_threshold=-1.0, noise_stddev=1e-3)
        ea = torch.randn((I, N, H, act), generator=gen)
        eo = torch.randn((act,), generator=gen)
        t0 = time.perf_counter()
        a, s, it = of.plan(pb['state'], sw, pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], ea, None, eo, cfg, osp)
        dt = time.perf_counter() - t0
        assert it == I and np.all(np.isfinite(a))
        return dt

    # The per-member GEMMs are small ([N x 128] x [128 x 128]): on a many-core host torch's default (one thread per core)
    # is far slower than a moderate count (measured on a 256-CPU MI355X host: 19 s per B2 plan with 128 threads).  Pick the
    # fastest of a few thread counts on one B1 plan each, and say which one was used.
    gen = torch.Generator().manual_seed(2026)
    trial = {}
    for th in sorted({t for t in (4, 8, 16, 32, default_threads) if t <= max(default_threads, 4)}):
        torch.set_num_threads(th)
        run_plan(500, 25, gen)
        trial[th] = run_plan(500, 25, gen)
        if trial[th] > 3.0 * min(trial.values()):
            break                                           # far past the optimum already: do not spend the budget there
    threads = min(trial, key=trial.get)
    torch.set_num_threads(threads)
    res = {}
    for name, N, H, share in (('B1', 500, 25, 0.3), ('B2', 2000, 30, 0.7)):
        times = []
        t_start = time.perf_counter()
        while len(times) < 4 or (time.perf_counter() - t_start < budget_s * share and len(times) < 41):
            times.append(run_plan(N, H, gen))
        times = times[1:]                                   # the first plan is the warm-up
        res[name] = dict(plans_per_s=1.0 / float(np.median(times)), plans_timed=len(times), median_s=float(np.median(times)),
                         workload='obs=60 act=2 K=P=E=5 N=%d H=%d I=5 k=N/10' % (N, H))
    torch.set_num_threads(default_threads)
    torch_leg = dict(value=res['B2']['plans_per_s'], unit='plans/s', cores=int(threads), kind='port', host_cpu_count=os.cpu_count(),
                     torch_num_threads=int(threads), torch_default_threads=int(default_threads),
                     thread_trial_b1_seconds={str(k): round(v, 3) for k, v in trial.items()},
                     sample='median of %d whole B2 plans (N=2000,H=30,K=5,I=5) after 1 warm-up through oracle/cem_oracle_fast.py '
                            '(torch-CPU fp32, baddbmm over members, %d intra-op threads — the fastest of a short trial — of %s host CPUs); B1 (N=500,H=25): median of %d plans'
                            % (res['B2']['plans_timed'], threads, os.cpu_count(), res['B1']['plans_timed']),
                     b1=res['B1'], b2=res['B2'])
    # The same plan through the C + OpenMP restatement (oracle/cem_oracle_c.c, pinned to the numpy oracle by tests/test_oracle_c.py): the
    # faster of the two is the reported baseline, both are in the line.
    try:
        c_leg = cpu_baseline_openmp(pb, osp, o, budget_s)
    except Exception as e:                                  # no gcc / OpenMP on this box: the torch port stands alone
        c_leg = dict(error=str(e)[:200])
    best = dict(c_leg if c_leg.get('value', 0.0) > torch_leg['value'] else torch_leg)
    best['torch_port'] = {k: torch_leg[k] for k in ('value', 'cores', 'sample', 'thread_trial_b1_seconds', 'b1', 'b2')}
    best['c_openmp'] = c_leg
    return best


def cpu_baseline_openmp(pb, osp, o, budget_s):
    """oracle/cem_oracle_c.c on this box's host cores: noise drawn inside the plan (per-thread generator), thread count = the fastest of
    a short trial on B1, then the median of >= 3 whole plans of B1 and B2."""
    from oracle import cem_oracle_c as oc                   # cpu_baseline leg only
    I, K = 5, 5

    def run_plan(N, H, seed):
        cfg = o.PlanConfig(horizon=H, iterations=I, n_samples=N, n_elite=N // 10, particles=K, ensemble_size=K, stddev_threshold=-1.0, noise_stddev=1e-3)
        t0 = time.perf_counter()
        a, s, it = oc.do_generate_action(pb['state'], pb['weights'], pb['inputs_min'], pb['inputs_max'], pb['low'], pb['high'], None, None, None, cfg, osp, seed=seed)
        dt = time.perf_counter() - t0
        assert it == I and np.all(np.isfinite(a))
        return dt
    trial = {}
    top = max(oc.max_threads(), 1)
    for th in sorted({t for t in (4, 8, 16, 32, 64, 128, top) if t <= max(top, 4)}):
        oc.set_threads(th)
        run_plan(500, 25, 1)
        trial[th] = run_plan(500, 25, 2)
        if trial[th] > 2.0 * min(trial.values()):
            break
    threads = min(trial, key=trial.get)
    oc.set_threads(threads)
    res = {}
    for name, N, H, share in (('B1', 500, 25, 0.15), ('B2', 2000, 30, 0.35)):
        times = []
        t_start = time.perf_counter()
        while len(times) < 4 or (time.perf_counter() - t_start < budget_s * share and len(times) < 41):
            times.append(run_plan(N, H, 10 + len(times)))
        times = times[1:]
        res[name] = dict(plans_per_s=1.0 / float(np.median(times)), plans_timed=len(times), median_s=float(np.median(times)),
                         workload='obs=60 act=2 K=P=E=5 N=%d H=%d I=5 k=N/10' % (N, H))
    return dict(value=res['B2']['plans_per_s'], unit='plans/s', cores=int(threads), kind='port', host_cpu_count=os.cpu_count(),
                thread_trial_b1_seconds={str(k): round(v, 3) for k, v in trial.items()},
                sample='median of %d whole B2 plans (N=2000,H=30,K=5,I=5) after 1 warm-up through oracle/cem_oracle_c.c (plain C, fp32, OpenMP over '
                       '16-row blocks, %d threads — the fastest of a short trial — of %s host CPUs); B1 (N=500,H=25): median of %d plans'
                       % (res['B2']['plans_timed'], threads, os.cpu_count(), res['B1']['plans_timed']),
                b1=res['B1'], b2=res['B2'])


def comm_ranks_or_none(pl):
    """RCCL's own rank count of the planner's communicator (ncclCommCount), or None where the loaded librccl does not export it."""
    try:
        return pl.comm_ranks()
    except Exception as e:                            # CEM_ERR_COMM: symbol missing — say so, do not fail the run over a diagnostic
        sys.stderr.write('ncclCommCount unavailable (%s): communicator rank count not verified\n' % e)
        return None


def all_ranks_ok(torch, dist, ok, ctl_dev):
    """True iff `ok` holds on every rank (one MIN all-reduce over the control group); single process: `ok` itself."""
    if dist is None:
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=ctl_dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return int(flag.item()) == 1


class HeadlineGuard:
    """The extra legs of a multi-rank run (B5, its split-product form) contain collectives.  If a rank dies or faults INSIDE one, the
    others block in it and nothing in-process can agree about anything any more — so before the extras start, every rank arms a
    timer: if the extras have not finished within `seconds`, rank 0 prints the line it already has (headline complete, top-level
    `extras_timed_out: true`, the extras marked as timed out), every rank writes a line to stderr and leaves with os._exit, so the
    launcher sees an end instead of its own timeout and the run keeps its headline.  The exit status is `--extras-timeout-status`
    (default 0: a launcher that discards the stdout of a failed run would lose the headline the guard exists to keep; the flag and the
    stderr lines mark the run as one to investigate either way)."""

    def __init__(self, out, rank, seconds, exit_code=0):
        import threading
        self.out, self.rank, self.exit_code = out, rank, exit_code
        self.timer = threading.Timer(seconds + (0.0 if rank == 0 else 5.0), self.fire)
        self.timer.daemon = True
        self.seconds = seconds

    def fire(self):
        # never silent: the cause of the hang has to be looked for in this run's records, so every rank says so on stderr and the line
        # carries a TOP-LEVEL flag (not only an error string nested in the extras)
        sys.stderr.write('bench.py rank %d: the multi-rank extras did not finish within %d s (a rank blocked in a collective, or a GPU hang): '
                         'printing the headline as it stands with "extras_timed_out": true and leaving with status %d\n' % (self.rank, self.seconds, self.exit_code))
        sys.stderr.flush()
        if self.rank == 0:
            line = dict(self.out)
            line['extras_timed_out'] = True
            for key in ('b2_strong', 'b5', 'b5_split_bf16x3'):
                line.setdefault(key, {'error': 'did not finish within %d s (a rank blocked in a collective?); headline printed by the guard' % self.seconds})
            print(json.dumps(line), flush=True)
        os._exit(self.exit_code)

    def __enter__(self):
        self.timer.start()
        return self

    def __exit__(self, *exc):
        self.timer.cancel()
        return False


class RunGuard:
    """A multi-rank run whose HEADLINE blocks — a rank died, a collective (captured or not) never completes: nothing in-process can recover
    from that, and the launcher would only see its own timeout and no line at all.  Armed before the first collective of a distributed run: if
    the line has not been printed within `seconds`, rank 0 prints a line that SAYS so (`value` null, an `error`, what had been reached) and
    every rank leaves with a non-zero status.  Disarmed once the headline is measured (the extras have their own guard)."""

    def __init__(self, rank, gpus, seconds):
        import threading
        self.rank, self.gpus, self.seconds, self.stage = rank, gpus, seconds, 'set-up'
        self.timer = threading.Timer(seconds, self.fire)
        self.timer.daemon = True

    def fire(self):
        sys.stderr.write('bench.py rank %d: the headline did not finish within %d s (reached: %s): a rank is blocked in a collective or the GPU hangs\n'
                         % (self.rank, self.seconds, self.stage))
        sys.stderr.flush()
        if self.rank == 0:
            print(json.dumps({'metric': 'planning steps/sec (CEM-MPC, N=2000 K=5 H=30)', 'value': None, 'unit': 'plans/s', 'n_gpus': self.gpus,
                              'error': 'the headline did not finish within %d s (reached: %s)' % (self.seconds, self.stage), 'headline_timed_out': True}), flush=True)
        os._exit(4)

    def start(self):
        self.timer.start()
        return self

    def cancel(self):
        self.timer.cancel()


def config_leg(torch, name, obs, act, K, N, H, dev, steps, warmup, P=None, I=5, k=None, variant='cem', thr=-1.0, post=0.15, noise=1e-3, what=None):
    """One more configuration on the same line as the headline: whole plans timed like the headline (graph replay), then the launches
    of an iteration by HIP events on the planner's stream.  Labelled extras — `value` stays B2.  The BASELINE configs (configs[0], [2],
    [3] of BASELINE.json; [4] is b5_leg) use K = P = E, I = 5, k = N/10, the CemMpc objective and no early stop; the policy legs pass the
    reference's own ctor values (config/policies.yaml:2-20, config/models.yaml:3): E = K members, P particles, I, k, the variant, the
    early-stop threshold and the Beta threshold.  `rollout_share_of_plan` = the rollout launches' device time / the plan's wall time;
    `reduce_us_per_iteration` / `select_us_per_iteration` / `sampler_us_per_iteration`: the other launches of an iteration (0 where a
    launch does not exist: the reduce is folded into the select on single-rank CemMpc plans, the sampler into the rollout tiles)."""
    from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
    P = K if P is None else P
    k = N // 10 if k is None else k
    pb = synthetic.problem(obs, act, K)
    cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=P, n_samples=N, horizon=H, n_elite=k, iterations=I,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=thr, noise_stddev=noise, variant=variant,
                        posterior_mean_threashold=post, use_graph=True)
    pl = CemPlanner(cfg, device=dev)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    for i in range(warmup):
        pl.plan(pb['state'], seed=2029, call=i)
    torch.cuda.synchronize()
    iters_run = []
    t0 = time.perf_counter()
    for i in range(steps):
        a, s, it = pl.plan(pb['state'], seed=2029, call=warmup + i)
        iters_run.append(it)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert (it == I or thr > 0) and np.all(np.isfinite(a))
    graph = pl.graph_status() == 'graph'
    pl.set_timing(True)
    roll_ms, sel_ms, red_ms, samp_ms, roll_n = 0.0, 0.0, 0.0, 0.0, 0
    for i in range(3):
        pl.plan(pb['state'], seed=2029, call=warmup + i)            # (plans timed above: the same iteration counts under early stop)
        tm = pl.last_timing()
        roll_ms += tm['rollout_ms']; sel_ms += tm['select_ms']; red_ms += tm['reduce_ms']; samp_ms += tm['sampler_ms']; roll_n += tm['rollout_launches']
    pl.set_timing(False)
    avg_ms = roll_ms / max(roll_n, 1)
    mean_iters = float(np.mean(iters_run))
    flops_launch = synthetic.flops_per_row_step(obs, act) * P * N * H
    label = what or ('%s: obs=%d act=%d K=P=E=%d N=%d H=%d I=%d k=%d units=128 layers=4, %s objective, early stop %s'
                     % (name, obs, act, K, N, H, I, k, 'SafeCemMpc' if variant == 'safe' else 'CemMpc', 'off' if thr <= 0 else 'at mean(sigma) <= %g' % thr))
    out = dict(workload=label,
               plans_per_s=steps / dt, ms_per_plan=1e3 * dt / steps, steps=steps, warmup=warmup, iterations_run_mean=mean_iters,
               candidate_trajectory_steps_per_s=steps / dt * mean_iters * N * H,
               rollout_ms_per_launch=avg_ms, rollout_launches_timed=roll_n, algorithmic_flops_per_launch=flops_launch,
               rollout_tflops=flops_launch / (avg_ms * 1e-3) / 1e12,
               frac_of_fp32_mfma_peak=flops_launch / (avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
               plan_frac_of_fp32_mfma_peak=mean_iters * flops_launch / (dt / steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
               rollout_share_of_plan=mean_iters * avg_ms / (1e3 * dt / steps),
               select_us_per_iteration=1e3 * sel_ms / max(roll_n, 1), reduce_us_per_iteration=1e3 * red_ms / max(roll_n, 1),
               sampler_us_per_iteration=1e3 * samp_ms / max(roll_n, 1),
               kernel=rollout_kernel_name(pl, obs, act), chunks_per_tile=pl.tiles()[0], tiles=int(len(pl.tiles()[1])),
               horizon_segments=pl.segments()[0], launches_per_iteration=pl.launches_per_iteration(), select_mode=pl.select_mode(), hip_graph=graph)
    pl.close()
    return out


# The reference's own policies at the shapes it ships (config/policies.yaml:2-20 with config/models.yaml:3 `ensemble_size: 15`; the default
# agent runs safe_cem_mpc, config/agents.yaml:11) and the headline configuration on the SafeCemMpc objective.  Synthetic PointGoal1-shaped
# observations (obs 60, act 2).  B2_safe keeps the Beta threshold the parity tests use at P = 5 (0.3: with the shipped 0.15 no candidate of
# five particles can be safe — (alpha + 0) / (alpha + beta + 5) = 0.163 — which costs the same time but scores every candidate -100).
POLICY_LEGS = (
    # name, K (= E), P, N, H, I, k, variant, thr, post, noise, steps, warmup
    ('B2_safe', 5, 5, 2000, 30, 5, 80, 'safe', -1.0, 0.3, 1e-3, 40, 10),
    ('shipped_safe_cem_mpc', 15, 45, 500, 8, 9, 20, 'safe', -1.0, 0.15, 0.01, 40, 10),
    ('shipped_safe_cem_mpc_early_stop', 15, 45, 500, 8, 9, 20, 'safe', 0.25, 0.15, 0.01, 40, 10),
    ('shipped_cem_mpc', 15, 5, 150, 8, 10, 15, 'cem', -1.0, 0.15, 0.001, 100, 20),
    ('shipped_cem_mpc_early_stop', 15, 5, 150, 8, 10, 15, 'cem', 0.25, 0.15, 0.001, 100, 20),
)


def policy_legs(torch, dev):
    legs = {}
    for name, K, P, N, H, I, k, variant, thr, post, noise, st_, wu_ in POLICY_LEGS:
        what = ('%s: obs=60 act=2 ensemble_size=%d particles=%d n_samples=%d horizon=%d iterations=%d n_elite=%d units=128 layers=4, %s, '
                'stddev_threshold %s, noise_stddev %g' % (name, K, P, N, H, I, k, 'SafeCemMpc (posterior_mean_threashold %g)' % post if variant == 'safe' else 'CemMpc',
                                                          'off (every iteration runs)' if thr <= 0 else '%g as shipped' % thr, noise))
        try:
            legs[name] = config_leg(torch, name, 60, 2, K, N, H, dev, steps=st_, warmup=wu_, P=P, I=I, k=k, variant=variant, thr=thr, post=post,
                                    noise=noise, what=what)
        except Exception as e:
            legs[name] = {'error': str(e)[:300]}
    return legs


def b5_leg(torch, dist, pb, G, rank, dev, native, steps, warmup, rehearse_world=0, ctl_dev=None, precision='fp32', N=65536, k=6554, label='B5'):
    """BASELINE.json configs[4], strong-scaled: N = 65536 candidates (K = P = E = 5, H = 30, I = 5, k = 6554) sharded over the G
    ranks of this run, one all-gather of the scores per CEM iteration.  Timed like the headline: barrier + synchronize on both
    sides, max over ranks.  `rehearse_world` = R > 0 (CEM_BENCH_B5_REHEARSAL=R on a one-GPU box): this process plays rank 0 of R —
    65536/R candidates, the replicated sample and select over all 65536 — with the all-gather replaced by a device copy of its
    own shard (the other ranks' scores keep plausible stale values), as scripts/time_b5_rank.py does; the stepwise C-ABI calls are
    the ones the host-stepped multi-rank driver makes.
    With N = 2000, k = 200, label 'B2' the same leg is the LITERAL metric of BASELINE.json at G GPUs — "(N=2000 K=5 H=30) at 1/2/4/8":
    2000 candidates in total, 2000/G per rank, plain plans/s (`b2_strong`; the headline `value` weak-scales instead)."""
    from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
    obs, act, K, H, I = 60, 2, 5, 30, 5
    W = rehearse_world or G
    cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=k, iterations=I,
                        scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=-1.0, noise_stddev=1e-3, variant='cem',
                        world_size=W, rank=0 if rehearse_world else rank, use_graph=not rehearse_world, precision=precision)
    # Everything that can fail on ONE rank only (handle creation, the communicator, its rank count) happens before the leg's first
    # collective, and the ranks then agree — over the control group — whether all of them got through: a rank that threw here while
    # the others walked into ncclAllGather / dist.barrier would leave them blocked until the launcher's timeout, headline lost.
    pl, n_seen, setup_error = None, 0, None
    try:
        pl = CemPlanner(cfg, device=dev)
        pl.set_weights(pb['weights'])
        pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
        if native and not rehearse_world:
            pl.comm_init()
            n_seen = comm_ranks_or_none(pl)
            if n_seen not in (None, G):
                raise RuntimeError('rank %d: the RCCL communicator has %s ranks, --gpus is %d' % (rank, n_seen, G))
    except Exception as e:
        setup_error = str(e)[:300]
    if not all_ranks_ok(torch, dist, setup_error is None, ctl_dev or dev):
        if pl is not None:
            if pl.has_comm:
                pl.comm_destroy()
            pl.close()
        return {'error': 'skipped on every rank: set-up failed on %s' % ('this rank: ' + setup_error if setup_error else 'another rank')}
    if rehearse_world:
        exchange = 'rehearsal: device copy of this rank\'s shard (1 process playing rank 0 of %d)' % W
        lo, hi = 0, N // W
        frames = synthetic.rehearsal_score_frames(pl, pb['state'], W, I, seed=2028)     # the other ranks' shards: shaped like this rank's own

        def one_plan(i):
            pl.plan_begin(pb['state'], seed=2028, call=i)
            for it in range(I):
                pl.plan_rollout(it)
                with torch.cuda.stream(pl.stream):
                    pl.scores_global(sync=False)[hi:].copy_(frames[it, hi:])
                    pl.scores_global(sync=False)[lo:hi].copy_(pl.scores_local(sync=False))
                pl.plan_select(it)
            return pl.plan_end()
    elif native:
        exchange = 'ncclAllGather inside the library, on the planner stream'

        def one_plan(i):
            return pl.plan(pb['state'], seed=2028, call=i)
    else:
        from ethz_safe_learning_amd.sharded import ShardedCemDriver
        drv = ShardedCemDriver(pl, I, world_size=G, always_exchange=True)
        exchange = 'torch.distributed all_gather between ctypes calls'

        def one_plan(i):
            return drv.plan(pb['state'], seed=2028, call=i)
    for i in range(warmup):
        one_plan(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        a, s, it = one_plan(warmup + i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev or dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert it == I and np.all(np.isfinite(a))
    pl.set_timing(True)
    roll_ms, sel_ms, roll_n = 0.0, 0.0, 0
    for i in range(3):
        one_plan(1000 + i)
        tm = pl.last_timing()
        roll_ms += tm['rollout_ms']; sel_ms += tm['select_ms']; roll_n += tm['rollout_launches']
    pl.set_timing(False)
    flops_launch = synthetic.flops_per_row_step(obs, act) * K * (N // W) * H
    avg_ms = roll_ms / max(roll_n, 1)
    out = dict(workload='%s: obs=60 act=2 K=P=E=5 N=%d H=30 I=5 k=%d, strong-scaled over %d ranks' % (label, N, k, W), scaling='strong', unit='plans/s',
               plans_per_s=steps / dt, ms_per_plan=1e3 * dt / steps, steps=steps, n_ranks=W, candidates_per_rank=N // W,
               candidate_trajectory_steps_per_s=steps / dt * I * N * H,
               rollout_ms_per_launch=avg_ms, rollout_frac_of_fp32_mfma_peak_per_rank=flops_launch / (avg_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS,
               rollout_kernel=(rollout_kernel_name(pl, obs, act) if precision == 'fp32' else 'void cem_rollout_split_kernel<%d, 1, 0>(RolloutParams)' % pl.tiles()[0]),
               precision=precision, select_us_per_iteration=1e3 * sel_ms / max(roll_n, 1),
               chunks_per_tile=pl.tiles()[0], tiles=int(len(pl.tiles()[1])), exchange=exchange, n_ranks_seen_by_rccl=n_seen,
               hip_graph=pl.graph_status() == 'graph')
    if native and not rehearse_world:
        pl.comm_destroy()
    pl.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-split-leg', action='store_true')
    ap.add_argument('--no-configs', action='store_true', help='skip the B1 / B3 / B4 / B5-rank extras of a single-GPU run')
    ap.add_argument('--no-b5', action='store_true', help='skip the B5 extras of a multi-GPU run')
    ap.add_argument('--no-local-leg', action='store_true', help='multi-rank runs: skip the per-rank single-GPU B2 leg in front of the headline')
    ap.add_argument('--extras-timeout', type=int, default=240, help='seconds after which a multi-rank run prints its headline without the extras')
    ap.add_argument('--headline-timeout', type=int, default=420, help='multi-rank runs: seconds after which a run without a headline prints an error line and exits non-zero')
    ap.add_argument('--extras-timeout-status', type=int, default=0, help='exit status of a run whose extras timed out (the line then carries "extras_timed_out": true)')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--chunks', type=int, default=0)
    ap.add_argument('--segments', type=int, default=0, help='rollout work-queue segments: 0 auto, 1 off')
    ap.add_argument('--n-per-gpu', type=int, default=2000)
    args = ap.parse_args()

    import torch
    from ethz_safe_learning_amd import CemPlanner, PlannerConfig, synthetic
    from ethz_safe_learning_amd.sharded import ShardedCemDriver

    G = args.gpus
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # CEM_BENCH_FORCE_DIST=1: run the multi-rank leg (process group, stepwise plan with the score all-gather, max-over-ranks
    # timing) with however many ranks there are, even one — the only way to rehearse that leg on a one-GPU box
    distributed = G > 1 or os.environ.get('CEM_BENCH_FORCE_DIST') == '1'
    # CEM_BENCH_SHARE_GPU=1 (rehearsal on a one-GPU box, tests/test_gpu_multirank.py): every rank uses cuda:0, gloo carries the
    # barrier / max-over-ranks (RCCL refuses two ranks on one device) and CEM_RCCL_LIBRARY names the stand-in for the collective
    share_gpu = os.environ.get('CEM_BENCH_SHARE_GPU') == '1'
    run_guard = None
    if distributed:
        assert world == G, 'launch with torch.distributed.run --nproc-per-node %d' % G
        import torch.distributed as dist
        run_guard = RunGuard(rank, G, args.headline_timeout).start()
        if share_gpu:
            torch.cuda.set_device(0)
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = 'cuda:%d' % (local_rank if distributed and not share_gpu else 0)
    ctl_dev = 'cpu' if share_gpu else dev               # where the few control tensors of the process group live

    obs, act, K, H, I = 60, 2, 5, 30, 5
    N = args.n_per_gpu * G
    k = N // 10                                       # the reference's elite ratio for cem_mpc (policies.yaml:6-7)
    pb = synthetic.problem(obs, act, K)
    cfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=N, horizon=H, n_elite=k,
                        iterations=I, scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=-1.0,
                        noise_stddev=1e-3, variant='cem', world_size=G, rank=rank, chunks_per_tile=args.chunks, rollout_segments=args.segments,
                        use_graph=(not args.no_graph))
    pl = CemPlanner(cfg, device=dev)
    pl.set_weights(pb['weights'])
    pl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
    # Multi-rank: the library owns an RCCL communicator (cem_planner_comm_init) and runs the whole sharded plan natively —
    # kernels + one ncclAllGather of the scores per CEM iteration on the planner's stream, one hipGraph per rank.
    # CEM_BENCH_PYTHON_EXCHANGE=1 keeps the older host-stepped form (torch.distributed all_gather between ctypes calls).
    native = distributed and os.environ.get('CEM_BENCH_PYTHON_EXCHANGE') != '1'
    if native:
        # every rank must end up on the same path: agree on whether the library's communicator came up everywhere, and fall back
        # to the host-stepped exchange (torch.distributed all_gather between the library calls) on all ranks if it did not
        try:
            pl.comm_init()
            n_seen = comm_ranks_or_none(pl)
            assert n_seen in (None, G), 'the RCCL communicator has %s ranks, --gpus is %d' % (n_seen, G)
            ok = 1
        except Exception as e:                        # e.g. librccl not loadable from the library, ncclCommInitRank refused
            sys.stderr.write('rank %d: native RCCL exchange unavailable (%s); using the host-stepped exchange\n' % (rank, e))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=ctl_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            if ok:
                pl.comm_destroy()
            native = False
    drv = ShardedCemDriver(pl, I, world_size=G, always_exchange=distributed)

    def one_plan(i):
        if not distributed or native:
            return pl.plan(pb['state'], seed=2026, call=i)
        return drv.plan(pb['state'], seed=2026, call=i)

    # Single-GPU extras FIRST, the headline after them.  A GPU that has been idle ramps its clock over the first ~13 B2 plans (26 ms of work:
    # 2.07 -> 1.89 ms per plan, profiles/r05_cold_start_steps.txt), which is most of what the driver's `--warmup 5 --steps 20` times; with the
    # other legs in front, the W warm-up plans and the K timed plans of the headline run at the sustained clock — what "planning steps per
    # second" means for a planner called in a loop.  Exactly W warm-up and K timed steps either way; `ms_per_step_each` lists every timed step.
    pre = {}
    if G == 1 and not distributed and not args.no_configs:
        # the other single-GPU BASELINE configs (and one rank of B5's eight) on the driver's line: labelled extras, `value` stays B2
        legs = {}
        for name, O_, A_, K_, N_, H_, st_, wu_ in (('B1', 60, 2, 5, 500, 25, 60, 15), ('B3', 60, 2, 16, 8192, 30, 8, 3), ('B4', 100, 12, 8, 4096, 50, 10, 3)):
            try:
                legs[name] = config_leg(torch, name, O_, A_, K_, N_, H_, dev, steps=st_, warmup=wu_)
            except Exception as e:
                legs[name] = {'error': str(e)[:300]}
        try:
            legs['B5_rank'] = b5_leg(torch, None, pb, 1, 0, dev, False, steps=10, warmup=3, rehearse_world=8, ctl_dev=ctl_dev)
        except Exception as e:
            legs['B5_rank'] = {'error': str(e)[:300]}
        legs.update(policy_legs(torch, dev))          # the reference's default policy and shipped shapes, and the headline shape on SafeCemMpc
        pre['configs'] = legs
    if G == 1 and not distributed and not args.no_split_leg:
        try:
            pre['split_bf16x3'] = split_leg(torch, pb, dev, steps=min(args.steps, 50), warmup=min(max(args.warmup, 3), 10), n_per_gpu=args.n_per_gpu)
        except Exception as e:                             # an extra must never cost the run its headline line
            pre['split_bf16x3'] = {'error': str(e)[:300]}
    if run_guard:
        run_guard.stage = 'per-rank single-GPU leg'
    if distributed and not args.no_local_leg:
        # Multi-rank runs: every rank first plans B2 ALONE on its GPU (a single-rank handle, no collective anywhere: nothing to block in) — the
        # per-GPU rate of the node's GPUs side by side (a slow or throttled GPU shows here, not as a mystery in the sharded number), and the
        # same clock conditioning the single-GPU run gets from its extras, so that the N = 1 and N > 1 lines of a scaling sweep compare like
        # with like.  The rates travel over the control group in one all-gather.
        lcfg = PlannerConfig(obs_dim=obs, act_dim=act, ensemble_size=K, particles=K, n_samples=2000, horizon=H, n_elite=200, iterations=I,
                             scorer=pb['scorer'], act_low=pb['low'], act_high=pb['high'], stddev_threshold=-1.0, noise_stddev=1e-3, variant='cem', use_graph=True)
        lpl = CemPlanner(lcfg, device=dev)
        lpl.set_weights(pb['weights']); lpl.set_normaliser(pb['inputs_min'], pb['inputs_max'])
        for i in range(15):
            lpl.plan(pb['state'], seed=2031, call=i)
        torch.cuda.synchronize()
        tl = time.perf_counter()
        for i in range(40):
            lpl.plan(pb['state'], seed=2031, call=15 + i)
        torch.cuda.synchronize()
        mine = torch.tensor([40.0 / (time.perf_counter() - tl)], dtype=torch.float64, device=ctl_dev)
        lpl.close()
        allr = [torch.zeros_like(mine) for _ in range(G)]
        dist.all_gather(allr, mine)
        pre['per_rank_single_gpu_b2_plans_per_s'] = [round(float(x.item()), 2) for x in allr]
    pre['order'] = ('the other legs first (single-GPU: configs, split leg; multi-rank: every rank planning B2 alone), then W warm-up + K timed headline plans at the sustained clock'
                    if len(pre) else 'headline first (no other leg in front of it in this run): the first timed plans of a cold GPU run during its clock ramp')

    if run_guard:
        run_guard.stage = 'headline warm-up'
    for i in range(args.warmup):
        one_plan(i)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    if run_guard:
        run_guard.stage = 'headline timed steps'
    t0 = time.perf_counter()
    stamps = [t0]
    for i in range(args.steps):
        a, s, it = one_plan(args.warmup + i)      # synchronous: the action is back on the host when it returns
        stamps.append(time.perf_counter())
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    step_ms = 1e3 * np.diff(np.array(stamps))
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=ctl_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert it == I and np.all(np.isfinite(a))
    if run_guard:
        run_guard.stage = 'roofline timing'

    # roofline of the dominant kernel: HIP events on the handle's stream around every rollout launch (eager launches)
    pl.set_timing(True)
    roll_ms, roll_n = 0.0, 0
    for i in range(5):
        if not distributed or native:
            pl.plan(pb['state'], seed=2027, call=i)
        else:
            drv.plan(pb['state'], seed=2027, call=i)
        tm = pl.last_timing()
        roll_ms += tm['rollout_ms']; roll_n += tm['rollout_launches']
    pl.set_timing(False)
    flops_launch = synthetic.flops_per_row_step(obs, act) * K * (N // G) * H
    avg_ms = roll_ms / max(roll_n, 1)
    achieved = flops_launch / (avg_ms * 1e-3) / 1e12

    # HBM-side bytes and MFMA-pipe busy cycles per rollout launch come from the committed rocprofv3 PMC passes of this same
    # workload (counters cannot be collected inside a timed run).  They are only quoted when that profile is of THIS code and THIS
    # kernel: the summary records the kernel name and a hash of the device sources; anything else is reported as stale, not reused.
    kname = rollout_kernel_name(pl, obs, act)
    mfma_util, traffic, profile_note = None, None, None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_b2.json')
    if G == 1 and args.n_per_gpu == 2000:
        if not os.path.exists(tpath):
            profile_note = 'no PMC profile committed'
        else:
            prof = json.load(open(tpath))
            if prof.get('kernel') != kname:
                profile_note = 'stale_profile: profiles/traffic_b2.json is of kernel %r, this run launched %r' % (prof.get('kernel'), kname)
            elif prof.get('source_sha16') != source_sha16():
                profile_note = 'stale_profile: profiles/traffic_b2.json was collected on device sources %s, this run is %s' % (prof.get('source_sha16'), source_sha16())
            else:
                traffic = prof['hbm_bytes_per_launch']
                # MFMA-pipe utilisation: PMC busy cycles per launch over 1024 SIMDs x this run's launch time
                mfma_util = prof.get('sq_valu_mfma_busy_cycles_per_launch', 0.0) / (1024 * avg_ms * 1e-3 * SHADER_CLOCK_HZ) or None

    if run_guard:
        run_guard.cancel()                             # the headline exists: from here on the extras' own guard keeps it
    plans_per_s = args.steps / dt
    b2_equiv = plans_per_s * (N / 2000.0)
    out = {
        'metric': 'planning steps/sec (CEM-MPC, N=2000 K=5 H=30)', 'value': b2_equiv, 'unit': 'plans/s',
        'n_gpus': G, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
        'ms_per_step_median': float(np.median(step_ms)), 'ms_per_step_p90': float(np.percentile(step_ms, 90)),
        'value_at_median_step': (1e3 / float(np.median(step_ms))) * (N / 2000.0),
        'ms_per_step_each': [round(float(x), 4) for x in step_ms[:64]],        # the timed steps one by one (the first 64): a cold GPU shows its clock ramp here
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'B2: obs=60 act=2 K=P=E=5 N=%d H=30 I=5 k=N/10 units=128 layers=4, CemMpc objective, early stop off%s'
                               % (N, '' if G == 1 else ' (weak-scaled: 2000 candidates per GPU, value in B2-equivalent plans/s)'),
                   'candidates_per_gpu': N // G, 'chunks_per_tile': pl.tiles()[0], 'tiles': int(len(pl.tiles()[1])),
                   'horizon_segments': pl.segments()[0], 'launches_per_iteration': pl.launches_per_iteration(),
                   'hip_graph': pl.graph_status() == 'graph',
                   'exchange': 'none (1 rank)' if not distributed else ('ncclAllGather inside the library, on the planner stream' if native
                                                                      else 'torch.distributed all_gather between ctypes calls'),
                   'parallelism': 'candidates sharded x%d, 1 all-gather of scores/iter' % G},
        'candidate_trajectory_steps_per_s': plans_per_s * I * N * H,
        # the same timed interval without the B2-equivalent scaling: whole plans of N = 2000 x G candidates per second
        'weak_plans_per_s': plans_per_s, 'weak_candidates_per_plan': N,
        'roofline': {'bound': 'mfma', 'achieved': achieved, 'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': achieved / PEAK_FP32_MFMA_TFLOPS, 'traffic': traffic, 'mfma_busy_frac': mfma_util,
                     'hbm_gbps': (traffic / (avg_ms * 1e-3) / 1e9) if traffic else None, 'hbm_peak_gbps': PEAK_HBM_GBPS,
                     'kernel': kname, 'profile_note': profile_note, 'avg_launch_ms': avg_ms, 'launches_timed': roll_n,
                     'algorithmic_flops_per_launch': flops_launch},
    }
    # BASELINE config 5 (N = 65536 over the node's GPUs) rides along on every multi-GPU run; CEM_BENCH_B5_REHEARSAL=R rehearses one
    # rank of R on a one-GPU box
    rehearse = int(os.environ.get('CEM_BENCH_B5_REHEARSAL', '0'))
    if (G > 1 or rehearse > 0) and not args.no_b5:
        if native:
            pl.comm_destroy()                          # one communicator at a time
        # an extra must never cost the run its headline: failures a rank can have on its own are agreed on before the leg's first
        # collective (b5_leg), and a leg that blocks is cut off by the guard, which prints the headline as it stands
        with HeadlineGuard(out, rank, args.extras_timeout, args.extras_timeout_status):
            if 2000 % (rehearse or G) == 0:            # the literal metric: N = 2000 candidates in total over the ranks, plain plans/s
                out['b2_strong'] = b5_leg(torch, dist, pb, G, rank, dev, native, steps=min(args.steps, 50), warmup=min(max(args.warmup, 2), 10),
                                          rehearse_world=rehearse if G == 1 else 0, ctl_dev=ctl_dev, N=2000, k=200, label='B2')
            out['b5'] = b5_leg(torch, dist, pb, G, rank, dev, native, steps=min(args.steps, 20), warmup=min(max(args.warmup, 2), 5),
                               rehearse_world=rehearse if G == 1 else 0, ctl_dev=ctl_dev)
            if not args.no_split_leg:                  # the same sharded plan on the opt-in split-product rollout: a labelled extra, never `value`
                out['b5_split_bf16x3'] = b5_leg(torch, dist, pb, G, rank, dev, native, steps=min(args.steps, 20), warmup=min(max(args.warmup, 2), 5),
                                                rehearse_world=rehearse if G == 1 else 0, ctl_dev=ctl_dev, precision='bf16x3')
    out.update(pre)                                    # the single-GPU extras, measured before the headline (see `order`)
    if rank == 0 and G == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
