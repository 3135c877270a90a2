#!/bin/bash
# PMC counters of the rollout kernel at the BASELINE configs other than the bench line's B2 (which scripts/collect_profiles.sh
# covers): separate rocprofv3 --kernel-trace --pmc passes (never combined with other trace domains) of
# `CEM_SWEEP_ONLY=<config> CEM_SWEEP_NOGRAPH=1 python3 scripts/sweep_configs.py` — graph replay off so every launch is a dispatch.
# Outputs: gpurun_out/pmc_cfg/<config>/summary_pmc.csv, traffic.json.  usage (GPU box, repo root): CONFIGS="B1 B3 B4" bash scripts/collect_pmc_configs.sh
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
export CEM_SWEEP_NOGRAPH=1
for c in ${CONFIGS:-B1 B3 B4}; do
  OUT="$ROOT/gpurun_out/pmc_cfg/$c"; mkdir -p "$OUT"
  export CEM_SWEEP_ONLY=$c
  for pass in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
    tag=$(echo "$pass" | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass -d "$OUT/pmc_$tag" -o pmc --output-format csv -- python3 "$ROOT/scripts/sweep_configs.py" > "$OUT/pmc_$tag.log" 2>&1
    echo "pmc $c $tag done"
  done
  python3 "$ROOT/scripts/summarise_profiles.py" "$OUT" "$c" "CEM_SWEEP_ONLY=$c CEM_SWEEP_NOGRAPH=1 scripts/sweep_configs.py"
  rm -rf "$OUT"/pmc_*/
done
