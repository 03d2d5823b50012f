"""Compiles rt_is.hip with -Rpass-analysis=kernel-resource-usage and prints one line per kernel."""
import re, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "roundtrip_amd/csrc/rt_is.hip"
extra = sys.argv[2:]
p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage", src] + extra, capture_output=True, text=True)
cur = None
rows = {}
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip(); rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1); rows[cur][k.strip()] = v.strip()
for k, r in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
    print("%-45s VGPR %3s AGPR %3s scratch %4s sgprspill %4s vgprspill %4s" % (
        name[-45:], r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("SGPRs Spill"), r.get("VGPRs Spill")))
if p.returncode: print(p.stderr[-2000:])
