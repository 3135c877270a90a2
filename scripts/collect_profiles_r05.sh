#!/bin/bash
# Round 5 kernel-stats evidence (repo root on the GPU box): the policy legs of bench.py under rocprofv3 --kernel-trace --stats.
#   LEGS="B2_safe shipped_safe_cem_mpc shipped_cem_mpc" bash scripts/collect_profiles_r05.sh
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
LEGS="${LEGS:-B2_safe shipped_safe_cem_mpc shipped_cem_mpc}"
OUT="$ROOT/gpurun_out/prof05"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for leg in $LEGS; do
  D="$OUT/stats_$leg"; mkdir -p "$D"
  CEM_LEG=$leg rocprofv3 --kernel-trace --stats -d "$D" -o stats --output-format csv -- python3 "$ROOT/scripts/run_policy_leg.py" 40 > "$OUT/$leg.log" 2>&1
  find "$D" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$OUT/kernel_stats_$leg.csv"
  echo "stats $leg done"; tail -1 "$OUT/$leg.log"
done
