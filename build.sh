#!/bin/bash
# Build libscldm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.  The seven translation units are
# compiled in parallel (objects under build/, git-ignored) and linked into one shared library.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${OUT:-scldm_amd/libscldm_hip.so}
OBJ=${OBJ:-build/obj}
# -fno-slp-vectorize: hipcc's SLP pass packs the LayerNorm sum / sum-of-squares sweep into v_pk_fma_f32 /
# v_pk_add_f32; with two waves per SIMD that packed code gave run-to-run different results on MI355X (bisected
# in round 1: identical source, only this flag differs), and packed f32 VALU next to MFMAs is slower anyway
# (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Extra flags may not switch it back on.
case " ${SCLDM_HIPCC_FLAGS:-} " in
  *" -fslp-vectorize"*|*"-fvectorize"*) echo "build.sh: SCLDM_HIPCC_FLAGS must not re-enable SLP vectorisation" >&2; exit 2;;
esac
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize ${SCLDM_HIPCC_FLAGS:-}"
mkdir -p "$OBJ"
# incremental: a translation unit is recompiled when its object is older than any file it included (dependency lists written by
# the compiler next to the objects), when the flags changed, or when SCLDM_REBUILD=1
echo "$FLAGS" > "$OBJ/.flags.new"
if [ "${SCLDM_REBUILD:-0}" = 1 ] || ! cmp -s "$OBJ/.flags.new" "$OBJ/.flags" 2>/dev/null; then rm -f "$OBJ"/*.o "$OBJ"/*.d; fi
mv "$OBJ/.flags.new" "$OBJ/.flags"
stale() {  # $1 = tu
  local o="$OBJ/$1.o" d="$OBJ/$1.d"
  [ -f "$o" ] && [ -f "$d" ] || return 0
  local f
  for f in $(sed -e 's/^[^:]*://' -e 's/\\$//' "$d"); do
    [ -f "$f" ] || return 0
    [ "$f" -nt "$o" ] && return 0
  done
  return 1
}
pids=()
for tu in api vae_api vae_train_api train_api train_fused optim train_step; do
  if stale "$tu"; then
    hipcc $FLAGS -MD -MF "$OBJ/$tu.d" -c "scldm_amd/csrc/$tu.hip" -o "$OBJ/$tu.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/api.o" "$OBJ/vae_api.o" "$OBJ/vae_train_api.o" "$OBJ/train_api.o" "$OBJ/train_fused.o" "$OBJ/optim.o" "$OBJ/train_step.o"
echo "built $OUT"
# the stand-alone GEMM harness (tests/perf/gemm_probe.hip: ablation timings + the bitwise kernel-against-kernel check that
# tests/test_gpu_gemm_probe.py runs); rebuilt when one of its headers is newer
P=tests/perf/gemm_probe
if [ ! -x "$P" ] || [ "$P.hip" -nt "$P" ] || [ scldm_amd/csrc/bgemm8.hpp -nt "$P" ] || [ scldm_amd/csrc/bgemm.hpp -nt "$P" ] || [ scldm_amd/csrc/common.hpp -nt "$P" ]; then
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -I include "$P.hip" -o "$P" 2> build/gemm_probe.log || { cat build/gemm_probe.log >&2; exit 1; }
  echo "built $P"
fi
