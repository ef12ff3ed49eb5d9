#!/bin/bash
# Build libscldm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.  The five translation units are
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
pids=()
for tu in api vae_api vae_train_api train_api train_fused; do
  hipcc $FLAGS -c "scldm_amd/csrc/$tu.hip" -o "$OBJ/$tu.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" "$OBJ/api.o" "$OBJ/vae_api.o" "$OBJ/vae_train_api.o" "$OBJ/train_api.o" "$OBJ/train_fused.o"
echo "built $OUT"
