#!/usr/bin/env python3
"""Fold the MCAB PMC listings of tools/r6_mcab_evidence.sh (gpurun_out/<tag>_mcab_<mode>_<prec>_pmc_*.txt + the kernel statistics)
into profiles/pmc_mcab.json (the file bench.py reads `mcab_roofline.*.traffic` from) and copy the listings to
profiles/<out>_mcab_*.   usage: tools/mcab_pmc_summary.py <tag> [out_tag=r6]

Per kernel and precision: raw counters per launch, HBM bytes per launch (FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950
correction - 128-byte requests are tallied at 64 bytes - plus WRITE_SIZE; both counters are in KB), and the derived figures the
DESIGN quotes: average clock (GRBM_GUI_ACTIVE is summed over the 8 XCDs), matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES over
1 024 SIMDs x elapsed cycles), VALU busy fraction (SQ_ACTIVE_INST_VALU is in quad-cycles), MFMA FLOPs actually issued."""
import csv, glob, json, os, re, shutil, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
out_tag = sys.argv[2] if len(sys.argv) > 2 else "r6"
short = {"enc_pool_kernel": "enc_pool", "enc_cell_kernel": "enc_cell", "dec_cell_kernel": "dec_cell", "dec_gene_kernel": "dec_gene",
         "dec_finalize_kernel": "dec_finalize", "dec_finalize_sample_kernel": "dec_finalize_sample"}
MFMA_FLOPS = {"fp32": 32 * 32 * 2 * 2, "fp16": 32 * 32 * 16 * 2, "bf16": 32 * 32 * 16 * 2}
res = {}
for path in sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_mcab_*_pmc_*.txt"))):
    m = re.search(rf"{tag}_mcab_(\w+?)_(fp32|fp16|bf16)_pmc_(\w+)\.txt$", path)
    if not m:
        continue
    mode, prec, part = m.groups()
    cur = None
    for line in open(path):
        k = re.match(r"(?:void )?scldm::(\w+?)(?:<.*?>)?\(.*dispatches (\d+)", line)
        if k:
            cur = res.setdefault(prec, {}).setdefault(short.get(k.group(1), k.group(1)), {"counters_per_launch": {}})
            cur["launches_profiled"] = int(k.group(2))
            continue
        c = re.match(r"\s+(\S+)\s+total \S+\s+per dispatch (\S+)", line)
        if c and cur is not None:
            cur["counters_per_launch"][c.group(1)] = float(c.group(2))
    shutil.copy(path, os.path.join(root, "profiles", os.path.basename(path).replace(f"{tag}_mcab_", f"{out_tag}_mcab_")))
for path in glob.glob(os.path.join(root, "gpurun_out", f"{tag}_mcab_*_kernel_stats.txt")):
    shutil.copy(path, os.path.join(root, "profiles", os.path.basename(path).replace(f"{tag}_mcab_", f"{out_tag}_mcab_")))
    m = re.search(rf"{tag}_mcab_(\w+?)_(fp32|fp16|bf16)_kernel_stats\.txt$", path)
    if not m or m.group(1) not in ("decode", "encode"):
        continue
    prec = m.group(2)
    for row in csv.reader(open(path)):
        if len(row) > 3 and "scldm::" in row[0]:
            k = re.match(r"(?:void )?scldm::(\w+?)(?:<.*?>)?\(", row[0])
            name = short.get(k.group(1), k.group(1)) if k else None
            if name in res.get(prec, {}):
                res[prec][name]["rocprof_avg_launch_us"] = float(row[3]) / 1e3
                res[prec][name]["rocprof_calls"] = int(row[1])
for prec, ks in res.items():
    for name, r in ks.items():
        c = r["counters_per_launch"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            r["fetch_bytes_per_launch_corrected_x2"] = 2 * c["FETCH_SIZE"] * 1024
            r["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024
            r["hbm_bytes_per_launch"] = r["fetch_bytes_per_launch_corrected_x2"] + r["write_bytes_per_launch"]
        if "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            r["elapsed_cycles"] = cyc
            if "rocprof_avg_launch_us" in r:
                r["avg_clock_GHz"] = cyc / r["rocprof_avg_launch_us"] / 1e3
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                r["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
            if "SQ_ACTIVE_INST_VALU" in c:
                r["valu_busy_frac"] = 4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * cyc)
        if "SQ_INSTS_MFMA" in c:
            r["mfma_flops_issued_per_launch"] = c["SQ_INSTS_MFMA"] * MFMA_FLOPS["fp32" if name in ("enc_cell", "dec_cell") else prec]
        if "SQ_WAIT_INST_ANY" in c and "SQ_WAVE_CYCLES" in c:
            r["wait_inst_frac_of_wave_cycles"] = c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"]
res["note"] = (f"round {out_tag} ({tag}): tools/r6_mcab_evidence.sh = rocprofv3 --kernel-trace --stats and four separate --pmc passes (SQ set 1; SQ set 2 + "
               "GRBM_GUI_ACTIVE; FETCH_SIZE; WRITE_SIZE + TCC hit/miss), kernel trace only, over tests/perf/mcab_profile.py (bench.py's synthetic VAE: "
               "decode 8 192 rows x 17 002 genes, encode 4 096 cells x 6 147 tokens; 4 calls, all launches averaged)")
with open(os.path.join(root, "profiles", "pmc_mcab.json"), "w") as f:
    json.dump(res, f, indent=1)
for prec in ("fp32", "fp16", "bf16"):
    for name, r in res.get(prec, {}).items():
        print(prec, name, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if k != "counters_per_launch"})
