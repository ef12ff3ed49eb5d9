#!/usr/bin/env python3
"""Fold the per-pass PMC listings of tools/collect_evidence.sh (gpurun_out/<tag>_pmc_*.txt) into one JSON summary of the fused
DiT kernel: profiles/<tag>_pmc_dit_forward_kernel.json and profiles/pmc_dit_forward_kernel.json (the file bench.py reads
`roofline.traffic` from).   usage: tools/pmc_summary.py <tag> [layers_per_launch]"""
import hashlib, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
lpl = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vals, kernel, n = {}, None, None
for part in ("sq", "sq2", "fetch", "write"):
    path = os.path.join(root, "gpurun_out", f"{tag}_pmc_{part}.txt")
    if not os.path.exists(path):
        continue
    for line in open(path):
        m = re.match(r"(void scldm::dit_forward_kernel\S.*?) dispatches (\d+)", line)
        if m:
            kernel, n = m.group(1), int(m.group(2))
        m = re.match(r"\s+(\S+)\s+total \S+\s+per dispatch (\S+)", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
fetch_kb, write_kb = vals.get("FETCH_SIZE"), vals.get("WRITE_SIZE")
n_fwd = 12288
alg = (2 * n_fwd * 16 * 16 * 4 + (8 // lpl - 1) * 2 * n_fwd * 16 * 256 * 4 + 8 * 2 * (768 * 256 + 256 * 256 + 3 * 256 * 704)) / (8 // lpl)
out = {
    "kernel": kernel, "launches_profiled": n, "layers_per_launch": lpl,
    "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch_raw": write_kb,
    "fetch_bytes_per_launch_corrected_x2": None if fetch_kb is None else 2 * fetch_kb * 1024,
    "write_bytes_per_launch": None if write_kb is None else write_kb * 1024,
    "hbm_bytes_per_launch": None if fetch_kb is None or write_kb is None else 2 * fetch_kb * 1024 + write_kb * 1024,
    "algorithmic_hbm_bytes_per_launch": alg,
    # bench.py reports `roofline.traffic` from this file only while the kernel's sources are the ones that were profiled
    "kernel_source_sha256": hashlib.sha256(b"".join(open(os.path.join(root, "scldm_amd", "csrc", f), "rb").read()
                                                    for f in ("dit_forward.hpp", "common.hpp"))).hexdigest(),
    "sq_counters_per_launch": {k: v for k, v in vals.items() if k.startswith("SQ_") or k.startswith("GRBM") or k.startswith("TCC")},
    "note": f"round {tag}: separate rocprofv3 --pmc passes (SQ set 1; SQ set 2 + GRBM; FETCH_SIZE; WRITE_SIZE + TCC hit/miss) with --kernel-trace only, "
            "tools/rocprof_pmc.sh over tests/perf/dit_profile.py (default bench workload, 6 evaluations, all launches averaged: the first launch of an "
            "evaluation reads latents instead of a residual, the last writes velocities); FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM "
            "(gfx950 tallies 128-byte requests at 64 bytes); algorithmic = latents in / velocities out once per evaluation, the fp32 residual "
            "between launches, every layer's packed bf16 weights once; 12288 sample-forwards; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles",
}
if out["hbm_bytes_per_launch"]:
    out["traffic_ratio"] = out["hbm_bytes_per_launch"] / alg
for name in (f"{tag}_pmc_dit_forward_kernel.json", "pmc_dit_forward_kernel.json"):
    with open(os.path.join(root, "profiles", name), "w") as f:
        json.dump(out, f, indent=1)
print(json.dumps(out, indent=1)[:1500])
