#!/bin/bash
# One-off sensitivity check of tests/test_gpu_bounds.py (round 4): build the device library three times with a
# deliberate slip in how sampling_params reach the kernels, and show that the new tests FAIL on each (they pass on the real
# library).  Mutated sources live in build_ab/ (untracked); nothing here touches the product build.
#   usage (on the GPU box):  bash scripts/mutation_check_a3.sh build   # here, with hipcc
#                            bash scripts/mutation_check_a3.sh run     # on the GPU box: prints one PASS/FAIL line per mutant
set -u
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
SRC="$ROOT/ethz_safe_learning_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 -fPIC -shared -Wno-unused-function -Wno-unused-value"
declare -A MUT
MUT[init_index]='s/p.musig\[i\] = p.mu0\[i % p.A\]; p.musig\[p.HA + i\] = p.sigma0\[i % p.A\];/p.musig[i] = p.mu0[(i \/ p.A) % p.A]; p.musig[p.HA + i] = p.sigma0[(i \/ p.A) % p.A];/'
MUT[init_swap]='s/p.musig\[i\] = p.mu0\[i % p.A\]; p.musig\[p.HA + i\] = p.sigma0\[i % p.A\];/p.musig[i] = p.sigma0[i % p.A]; p.musig[p.HA + i] = p.mu0[i % p.A];/'
MUT[clip_index]='s/v = fminf(fmaxf(v, p.lb\[a\]), p.ub\[a\]);/v = fminf(fmaxf(v, p.lb[0]), p.ub[0]);/'
case "${1:-}" in
build)
    for m in "${!MUT[@]}"; do
        d="$ROOT/build_ab/mut_$m"; rm -rf "$d"; mkdir -p "$d/csrc" "$d/include"
        cp "$SRC"/*.h "$SRC"/*.hip "$d/csrc/"; cp "$ROOT"/include/*.h "$d/include/"
        sed -i "${MUT[$m]}" "$d/csrc/cem_device.h"
        if cmp -s "$d/csrc/cem_device.h" "$SRC/cem_device.h"; then echo "mutation $m did not apply"; exit 1; fi
        sed -i 's#../../include/cem_mpc.h#../include/cem_mpc.h#' "$d/csrc/cem_capi.hip"
        /opt/rocm/bin/hipcc $FLAGS -o "$d/libcem_mpc_gfx950.so" "$d/csrc/cem_capi.hip" 2>/dev/null || { echo "build of $m failed"; exit 1; }
        echo "built $d/libcem_mpc_gfx950.so"
    done ;;
run)
    cd "$ROOT"
    python -m pytest tests/test_gpu_bounds.py -m gpu -q -x >/dev/null 2>&1 && echo "real library: tests PASS" || echo "real library: tests FAIL (unexpected)"
    for m in "${!MUT[@]}"; do
        if CEM_MPC_LIB="$ROOT/build_ab/mut_$m/libcem_mpc_gfx950.so" python -m pytest tests/test_gpu_bounds.py -m gpu -q >/tmp/mut_$m.log 2>&1; then
            echo "mutant $m: tests PASS  (the slip went unnoticed!)"
        else
            echo "mutant $m: tests FAIL as they should: $(grep -E '^[0-9]+ failed|failed' /tmp/mut_$m.log | tail -1)"
        fi
    done ;;
*) echo "usage: $0 build|run"; exit 2 ;;
esac
