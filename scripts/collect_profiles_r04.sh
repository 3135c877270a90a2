#!/bin/bash
# Round 4 evidence in one gpurun call (repo root on the GPU box): the B2 bench command under rocprofv3 (kernel stats; PMC passes, each on
# its own) via scripts/collect_profiles.sh, kernel stats of the other BASELINE configs, their PMC counters via scripts/collect_pmc_configs.sh.
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
bash "$ROOT/scripts/collect_profiles.sh"
cd /tmp && export TMPDIR=/tmp
for c in B1 B3 B4 B5; do
  OUT="$ROOT/gpurun_out/prof/stats_$c"; mkdir -p "$OUT"
  CEM_SWEEP_ONLY=$c rocprofv3 --kernel-trace --stats -d "$OUT" -o stats --output-format csv -- python3 "$ROOT/scripts/sweep_configs.py" > "$OUT.log" 2>&1
  find "$OUT" -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} "$ROOT/gpurun_out/prof/kernel_stats_$c.csv"
  echo "stats $c done"
done
cd "$ROOT" && CONFIGS="B1 B3 B4" bash scripts/collect_pmc_configs.sh
