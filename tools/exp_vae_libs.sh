#!/bin/bash
# A/B of library builds for the MCAB kernels: vae_bench on the default library and every scldm_amd/libx_*.so
for lib in scldm_amd/libscldm_hip.so $(ls scldm_amd/libx_*.so 2>/dev/null); do
  echo "== $lib"; SCLDM_LIB=$PWD/$lib timeout 200 python tests/perf/vae_bench.py 2>&1 | grep "B=1024"
done
