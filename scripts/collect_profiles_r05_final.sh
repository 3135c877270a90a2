#!/bin/bash
# Round 5 evidence in one gpurun call (repo root on the GPU box).  Every rocprofv3 pass is its own run with --kernel-trace (--stats) only; the
# PMC passes use --kernel-trace --pmc with one counter group each (scripts/collect_profiles.sh, scripts/collect_pmc_configs.sh).
#   1. the B2 bench command: kernel stats + PMC + traffic.json            -> gpurun_out/prof/
#   2. the other BASELINE configs: kernel stats + PMC                      -> gpurun_out/prof/kernel_stats_<cfg>.csv, gpurun_out/pmc_cfg/
#   3. the policy legs (SafeCemMpc at B2, the shipped shapes)              -> gpurun_out/prof05/
#   4. the trainer step, the wide (256-unit) rollout, one agent iteration  -> gpurun_out/prof05/
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
bash "$ROOT/scripts/collect_profiles_r04.sh"
LEGS="B2_safe shipped_safe_cem_mpc shipped_cem_mpc" bash "$ROOT/scripts/collect_profiles_r05.sh"
OUT="$ROOT/gpurun_out/prof05"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_train" -o stats --output-format csv -- python3 "$ROOT/scripts/time_train_kernel.py" > "$OUT/train.log" 2>&1
find "$OUT/stats_train" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_train.csv"; echo "stats train done"
CEM_WIDE_PLAN_ONLY=1 rocprofv3 --kernel-trace --stats -d "$OUT/stats_wide256" -o stats --output-format csv -- python3 "$ROOT/scripts/time_wide_units.py" 256 > "$OUT/wide256.log" 2>&1
find "$OUT/stats_wide256" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_wide256.csv"; echo "stats wide256 done"
cd "$ROOT" && python3 scripts/time_agent_iteration.py 2 > "$OUT/agent_iteration.jsonl" 2> "$OUT/agent_iteration.err"; echo "agent iteration done"; tail -2 "$OUT/agent_iteration.jsonl"
python3 scripts/time_b5_rank.py > "$OUT/b5_rank_timing.jsonl" 2> "$OUT/b5_rank_timing.err"; echo "b5 rank timing done"
