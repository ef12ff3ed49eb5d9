#!/bin/bash
# Build a variant of libscldm_hip.so in which ONE translation unit is recompiled with extra -D flags (round-5 A/B runs):
#   tools/r5_variant.sh <name> <tu> "<extra hipcc flags>"   ->  build/libx_<name>.so   (run with SCLDM_LIB=build/libx_<name>.so)
set -euo pipefail
cd "$(dirname "$0")/.."
name=$1; tu=$2; extra=${3:-}
mkdir -p build/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-slp-vectorize $extra"
hipcc $FLAGS -c "scldm_amd/csrc/$tu.hip" -o "build/var/${tu}_$name.o"
objs=""
for t in api vae_api vae_train_api train_api train_fused optim train_step; do
  if [ "$t" = "$tu" ]; then objs="$objs build/var/${tu}_$name.o"; else objs="$objs build/obj/$t.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "build/libx_$name.so" $objs
echo "built build/libx_$name.so"
