#!/bin/bash
# Round-2 rocprofv3 evidence beyond the B2 bench line (run on the GPU box from the repo root):
#   kernel-trace stats of one BASELINE config at a time (B1..B4; the kernel instantiations differ per config), of what one
#   rank of the 8-GPU B5 plan executes (scripts/time_b5_rank.py), and of the ensemble training step.
# Each pass is its own rocprofv3 run with --kernel-trace --stats only.  Outputs: gpurun_out/prof2/*.csv, *.log
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof2"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
grab() {  # $1 = rocprof output dir, $2 = tag
  f=$(find "$1" -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" "$OUT/kernel_stats_$2.csv"; fi
}
for c in ${CONFIGS:-B1 B2 B3 B4}; do
  CEM_SWEEP_ONLY=$c rocprofv3 --kernel-trace --stats -d "$OUT/raw_$c" -o stats --output-format csv -- python3 "$ROOT/scripts/sweep_configs.py" > "$OUT/sweep_$c.log" 2>&1
  grab "$OUT/raw_$c" "$c"; echo "stats $c done"
done
if [ -z "$SKIP_B5" ]; then
  CASES=b5 REPS=4 rocprofv3 --kernel-trace --stats -d "$OUT/raw_b5rank" -o stats --output-format csv -- python3 "$ROOT/scripts/time_b5_rank.py" > "$OUT/b5rank_profiled.log" 2>&1
  grab "$OUT/raw_b5rank" b5rank; echo "stats b5 rank done"
  python3 "$ROOT/scripts/time_b5_rank.py" > "$OUT/b5rank.jsonl" 2> "$OUT/b5rank.err"; echo "b5 timing done"
fi
if [ -z "$SKIP_TRAIN" ]; then
  rocprofv3 --kernel-trace --stats -d "$OUT/raw_train" -o stats --output-format csv -- python3 "$ROOT/scripts/time_train_kernel.py" > "$OUT/train.log" 2>&1
  grab "$OUT/raw_train" train; echo "stats train done"
fi
rm -rf "$OUT"/raw_*
