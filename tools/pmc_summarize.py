#!/usr/bin/env python3
"""gpurun_out/pmcb_<config>_{fetch,write,sq,grbm}/**/*counter_collection.csv (tools/pmc_bench.sh) -> per-kernel medians as JSON.
   usage: python tools/pmc_summarize.py <root> <out.json> <config>
FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md); FETCH/WRITE_SIZE are in KiB;
SQ_WAVE_CYCLES / SQ_WAIT_* are quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv, glob, json, os, re, statistics, sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
config = sys.argv[3] if len(sys.argv) > 3 else "C2"
out = sys.argv[2] if len(sys.argv) > 2 else f"profiles/r03/pmc_summary_{config}.json"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_hash  # the build the counters were taken on: bench.py reports them only for the same kernel sources
vals = {}
for tag in ("fetch", "write", "sq", "grbm"):
    for f in glob.glob(os.path.join(root, f"pmcb_{config}_{tag}", "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()
            per.setdefault((name, r["Counter_Name"], r["Dispatch_Id"]), 0.0)
            per[(name, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
        for (name, ctr, _), v in per.items():
            vals.setdefault(name, {}).setdefault(ctr, []).append(v)
res = {}
for name, c in vals.items():
    med = {k: statistics.median(v) for k, v in c.items()}
    if med.get("GRBM_GUI_ACTIVE", 0) / 8 < 20000 and "field" not in name and "hg_" not in name and "hashgrid" not in name and "adam" not in name:
        continue
    e = {"launches": len(next(iter(c.values())))}
    if "FETCH_SIZE" in med:
        e["fetch_bytes_corrected"] = med["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in med:
        e["write_bytes"] = med["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        e["hbm_traffic_bytes"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if "GRBM_GUI_ACTIVE" in med:
        e["kernel_cycles_per_xcd"] = med["GRBM_GUI_ACTIVE"] / 8
    if "SQ_WAVE_CYCLES" in med and med["SQ_WAVE_CYCLES"] > 0:
        e["wait_any_frac"] = round(med.get("SQ_WAIT_ANY", 0) / med["SQ_WAVE_CYCLES"], 3)
        e["wait_inst_frac"] = round(med.get("SQ_WAIT_INST_ANY", 0) / med["SQ_WAVE_CYCLES"], 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in med and "SQ_BUSY_CYCLES" in med and med["SQ_BUSY_CYCLES"] > 0:
        # MFMA-busy cycles summed over SIMDs / (busy cycles per SE-level SQ x 4 SIMDs x CUs ...): report against GRBM cycles x 1024 SIMDs
        if "GRBM_GUI_ACTIVE" in med:
            e["mfma_util"] = round(med["SQ_VALU_MFMA_BUSY_CYCLES"] / (med["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    if "SQ_LDS_BANK_CONFLICT" in med:
        e["lds_bank_conflict_cycles"] = med["SQ_LDS_BANK_CONFLICT"]
    if "SQ_INSTS_MFMA" in med:
        e["mfma_insts"] = med["SQ_INSTS_MFMA"]
    res[name] = e
json.dump({"source": "rocprofv3 --pmc passes of tools/pmc_bench.sh over `bench.py --config " + config + " --steps 6 --warmup 2 --no-cpu-baseline --no-sampler-step` (1 GPU); median per launch; "
           "FETCH_SIZE doubled per MI355X_MICROARCH.md; mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)",
           "csrc_sha": csrc_hash(), "config": config, "kernels": res}, open(out, "w"), indent=1)
print("wrote", out, len(res), "kernels")
