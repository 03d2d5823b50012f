"""Summarises rocprofv3 --pmc CSVs written by scripts/pmc.sh: per counter, the value for the LAST dispatch
of the fused kernel (summed over the dispatch's rows)."""
import csv, glob, os, sys, collections
root = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "is_fused_kernel"
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    acc = collections.OrderedDict()
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if kern not in row.get("Kernel_Name", ""):
                continue
            key = (row["Dispatch_Id"], row["Counter_Name"])
            acc[key] = acc.get(key, 0.0) + float(row["Counter_Value"])
    if not acc:
        continue
    last = max(int(k[0]) for k in acc)
    for (d, name), v in acc.items():
        if int(d) == last:
            print("%-32s %20.0f   (%s)" % (name, v, os.path.relpath(f, root).split(os.sep)[0]))
