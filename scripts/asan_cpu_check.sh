#!/bin/bash
# Host-side sanitizer pass (no GPU: GPU AddressSanitizer is not available on this pool): the library's HOST code compiled with
# AddressSanitizer + UndefinedBehaviorSanitizer (device code unchanged), then the CPU tests that call into it — configuration
# validation, weight packing, tile plans, the cost model, segment plans — run under it.  usage: bash scripts/asan_cpu_check.sh
set -e -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=/tmp/libcem_mpc_asan.so
(cd "$ROOT/ethz_safe_learning_amd/csrc" && $HIPCC --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-honor-nans -mllvm -amdgpu-mfma-vgpr-form=1 \
    -fPIC -shared -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-omit-frame-pointer -o $OUT cem_capi.hip)
ASAN_SO=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd "$ROOT"
CEM_MPC_LIB=$OUT LD_PRELOAD=$ASAN_SO ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python -m pytest tests/test_capi_cpu.py tests/test_sharded_cpu.py -x -q -m "not gpu"
