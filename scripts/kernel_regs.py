#!/usr/bin/env python3
"""Register / LDS / scratch table of every kernel of one HIP source, from hipcc's own remarks
(-Rpass-analysis=kernel-resource-usage): what a change cost the kernels it did not mean to touch.
usage: python scripts/kernel_regs.py llama2.zig_amd/csrc/matvec.hip [name filter]"""
import re, subprocess, sys, os

def table(src, flt=""):
    d = os.path.dirname(os.path.abspath(src))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-c", os.path.basename(src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, cwd=d, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: (?:Function )?Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+(\w[\w ]*?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    names = [r["name"] for r in rows]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = []
    for r, n in zip(rows, dem):
        n = re.sub(r"l2z::\(anonymous namespace\)::", "", n)
        n = re.sub(r"\(.*", "", n).replace("void ", "")
        if flt and flt not in n:
            continue
        out.append((n, r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("TotalSGPRs", -1), r.get("ScratchSize", -1),
                    r.get("Occupancy", -1), r.get("LDS Size", -1)))
    return out

if __name__ == "__main__":
    rows = table(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch | waves/SIMD | static LDS |\n|---|---:|---:|---:|---:|---:|---:|")
    for r in rows:
        print("| " + " | ".join(str(x) for x in r) + " |")
