#!/bin/bash
# Build libscldm_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=scldm_amd/libscldm_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
  ${SCLDM_HIPCC_FLAGS:-} -o "$OUT" scldm_amd/csrc/api.hip
echo "built $OUT"
