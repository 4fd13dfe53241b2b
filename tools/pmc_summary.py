"""Summarise rocprofv3 --pmc passes (one directory per pass, --output-format csv) for one kernel: per-launch averages of every
counter (instances of a counter within a dispatch are summed), plus the derived figures DESIGN.md / bench.py quote
(effective clock, MFMA-pipe busy fraction, HBM bytes with the gfx950 FETCH_SIZE x2 correction of MI355X_MICROARCH.md section HBM).

    python tools/pmc_summary.py <kernel substring> <kernel_ms> <out.json> <pass dir> [<pass dir> ...]"""
import collections
import csv
import glob
import json
import sys

pat, kernel_ms, out = sys.argv[1], float(sys.argv[2]), sys.argv[3]
per = collections.defaultdict(lambda: collections.defaultdict(float))       # counter -> dispatch -> sum over instances
dur = collections.defaultdict(dict)                                         # counter -> dispatch -> duration of that dispatch
for d in sys.argv[4:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                per[r["Counter_Name"]][(f, r["Dispatch_Id"])] += float(r["Counter_Value"])
                dur[r["Counter_Name"]][(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
res = {c: sum(v.values()) / len(v) for c, v in per.items()}
res["kernel_ms_unprofiled"] = kernel_ms
if "GRBM_GUI_ACTIVE" in dur:        # the clock is derived against the duration of the SAME (profiled) dispatches
    kernel_ms = sum(dur["GRBM_GUI_ACTIVE"].values()) / len(dur["GRBM_GUI_ACTIVE"])
res["kernel_ms"] = kernel_ms
dv = {}
if "GRBM_GUI_ACTIVE" in res:
    dv["effective_clock_GHz"] = res["GRBM_GUI_ACTIVE"] / 8 / (kernel_ms * 1e-3) / 1e9      # summed over the 8 XCDs
if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "GRBM_GUI_ACTIVE" in res:
    dv["mfma_busy_frac"] = res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8 * 256 * 4) / 1.0 if False else \
        res["SQ_VALU_MFMA_BUSY_CYCLES"] / (res["GRBM_GUI_ACTIVE"] / 8 * 1024)               # 1024 SIMDs
if "SQ_WAIT_ANY" in res and "SQ_WAVE_CYCLES" in res:
    dv["wait_any_frac_of_wave"] = res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]
if "SQ_WAIT_INST_ANY" in res and "SQ_WAVE_CYCLES" in res:
    dv["wait_inst_any_frac_of_wave"] = res["SQ_WAIT_INST_ANY"] / res["SQ_WAVE_CYCLES"]
if "SQ_LDS_BANK_CONFLICT" in res and res.get("SQ_LDS_IDX_ACTIVE"):
    dv["lds_bank_conflict_frac"] = res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"]
if "FETCH_SIZE" in res:
    dv["hbm_read_GB_raw"] = res["FETCH_SIZE"] * 1024 / 1e9
    dv["hbm_read_GB_x2_gfx950_correction"] = 2 * res["FETCH_SIZE"] * 1024 / 1e9
if "WRITE_SIZE" in res:
    dv["hbm_write_GB"] = res["WRITE_SIZE"] * 1024 / 1e9
res["derived"] = dv
res["note"] = "rocprofv3 --pmc, separate passes (no trace domains), per-launch averages of kernels matching '%s'" % pat
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
