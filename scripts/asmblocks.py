"""Basic-block census of one kernel in a hipcc -S dump: MFMAs and other instructions per block."""
import re, sys, collections
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z.*%s.*:\s" % pat, l) or (l.startswith("_Z") and pat in l and l.rstrip().split(";")[0].strip().endswith(":")))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blk, blocks = "entry", collections.OrderedDict()
for i in range(start + 1, end):
    l = lines[i].strip()
    if not l or l.startswith(";") or l.startswith("."):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m: blk = m.group(1)
        continue
    op = l.split()[0]
    blocks.setdefault(blk, collections.Counter())[op] += 1
    blocks[blk]["_line0"] = blocks[blk].get("_line0", 0) or (i + 1)
for b, c in blocks.items():
    n = sum(v for k, v in c.items() if not k.startswith("_"))
    mf = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    if n < int(sys.argv[3]) if len(sys.argv) > 3 else 0: continue
    rest = {k: v for k, v in c.items() if not k.startswith("_") and not k.startswith("v_mfma")}
    print("%-12s line %6d  n=%4d mfma=%4d  %s" % (b, c["_line0"], n, mf, dict(sorted(rest.items(), key=lambda kv: -kv[1])[:14])))
