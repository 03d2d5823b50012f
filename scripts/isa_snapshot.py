#!/usr/bin/env python3
"""ISA identity check for a refactor: disassemble the gfx950 code object inside librt_is.so and compare, kernel by kernel,
with a snapshot taken before the change.

    python scripts/isa_snapshot.py save  /tmp/before.pkl [name-filter ...]
    python scripts/isa_snapshot.py check /tmp/before.pkl

A snapshot maps each kernel symbol to its list of instructions (addresses, encodings and comments stripped; branch targets
are kept as the disassembler prints them, so a moved block shows up).  Used in round 3 to prove that adding the Jacobian
modes to the skewed schedule left every is_fused_kernel / mlp_fwd_kernel instantiation untouched."""
import os, pickle, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "roundtrip_amd", "librt_is.so")


def disassemble(lib=LIB):
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        co = os.path.join(d, "k.co")
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        return subprocess.check_output([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], text=True)


def kernels(text, filters):
    out, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            if filters and not any(f in cur for f in filters):
                cur = None
            else:
                out[cur] = []
            continue
        if cur is None or not line.startswith(("\t", " ")):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            out[cur].append(ins)
    return out


if __name__ == "__main__":
    mode, path, filters = sys.argv[1], sys.argv[2], sys.argv[3:]
    if mode == "save":
        snap = kernels(disassemble(), filters or ["is_fused_kernel", "mlp_fwd_kernel"])
        pickle.dump(snap, open(path, "wb"))
        print(f"saved {len(snap)} kernels")
    else:
        snap = pickle.load(open(path, "rb"))
        now = kernels(disassemble(), [])
        bad = 0
        for k, v in snap.items():
            if k not in now:
                print("MISSING", k); bad += 1
            elif now[k] != v:
                n = next((i for i, (a, b) in enumerate(zip(v, now[k])) if a != b), min(len(v), len(now[k])))
                print(f"DIFFERS {k}: {len(v)} -> {len(now[k])} instructions, first difference at {n}"); bad += 1
        print(f"{len(snap) - bad}/{len(snap)} kernels identical")
        sys.exit(1 if bad else 0)
