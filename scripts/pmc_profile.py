"""Turns the PMC pass CSVs of scripts/pmc.sh into the committed profile summary profiles/<name>.json.

HBM/fabric traffic per launch, as MI355X_MICROARCH.md (HBM section) prescribes: FETCH_SIZE and WRITE_SIZE come from
separate --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide
(16 B/lane) coalesced loads — which is what the weight stream is — so it is doubled; WRITE_SIZE is taken as is.
usage: python scripts/pmc_profile.py gpurun_out/<pmc dir> profiles/<name>.json "<workload string>"
"""
import csv, glob, json, os, sys, collections
root, out, workload = sys.argv[1], sys.argv[2], sys.argv[3]
kern = "is_fused_kernel"
vals = {}
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.OrderedDict()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern not in row.get("Kernel_Name", ""):
                continue
            key = (int(row["Dispatch_Id"]), row["Counter_Name"])
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    if acc:
        last = max(k[0] for k in acc)
        for (d, name), v in acc.items():
            if d == last:
                vals[name] = v
res = {"workload": workload, "kernel": kern, "counters_last_dispatch": vals}
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    res["fetch_bytes_corrected"] = 2.0 * vals["FETCH_SIZE"] * 1024.0
    res["write_bytes"] = vals["WRITE_SIZE"] * 1024.0
    res["traffic_bytes_per_launch"] = res["fetch_bytes_corrected"] + res["write_bytes"]
if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and "GRBM_GUI_ACTIVE" in vals:
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 256 CUs x 4 SIMDs issue MFMAs
    res["mfma_pipe_busy_frac"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
    res["cycles_per_mfma"] = vals["SQ_VALU_MFMA_BUSY_CYCLES"] / vals["SQ_INSTS_MFMA"] if vals.get("SQ_INSTS_MFMA") else None
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
