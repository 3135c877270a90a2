#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline object cites (run on the GPU box from the repo root):
#   1. --kernel-trace --stats of the default bench line            -> gpurun_out/prof/stats/
#   2. PMC passes, each on its own (no trace domains beside --kernel-trace), graph replay off so every launch is a dispatch:
#      FETCH_SIZE | WRITE_SIZE + hits/misses | SQ busy / MFMA counters (MI355X_MICROARCH.md, HBM/rocprofv3 section)
# then scripts/summarise_profiles.py writes the per-kernel means that get copied to profiles/.
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- python3 "$ROOT/bench.py" --steps 40 --warmup 5 --no-cpu-baseline --no-split-leg --no-configs > "$OUT/stats.log" 2>&1
echo "stats pass done"
for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  tag=$(echo "$pass" | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $pass -d "$OUT/pmc_$tag" -o pmc --output-format csv -- python3 "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline --no-split-leg --no-configs --no-graph > "$OUT/pmc_$tag.log" 2>&1
  echo "pmc pass $tag done"
done
python3 "$ROOT/scripts/summarise_profiles.py" "$OUT"
