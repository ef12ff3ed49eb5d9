#!/bin/bash
# Build libscldm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=${OUT:-scldm_amd/libscldm_hip.so}
# -fno-slp-vectorize: hipcc's SLP pass packs the LayerNorm sum / sum-of-squares sweep into v_pk_fma_f32 /
# v_pk_add_f32; with two waves per SIMD that packed code gave run-to-run different results on MI355X (bisected
# in round 1: identical source, only this flag differs), and packed f32 VALU next to MFMAs is slower anyway.
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function -fno-slp-vectorize \
  ${SCLDM_HIPCC_FLAGS:-} -o "$OUT" scldm_amd/csrc/api.hip scldm_amd/csrc/vae_api.hip scldm_amd/csrc/train_api.hip
echo "built $OUT"
