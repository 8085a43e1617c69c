#!/bin/bash
# HBM traffic per launch of the decode kernels (7B shape) from the PMC counters, collected as
# MI355X_MICROARCH.md prescribes: separate passes for FETCH_SIZE and WRITE_SIZE, kernel-trace only.
# Writes gpurun_out/pmc_traffic.json and gpurun_out/<tag>_pmc_traffic.md  (tag = $1, default r01)
tag=${1:-r01}
repo=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  L2Z_NO_GRAPH=1 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o p --output-format csv -- \
    python $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra > /tmp/pmc_$c.log 2>&1 || tail -5 /tmp/pmc_$c.log
done
python - "$tag" "$repo" <<'PY'
import csv, glob, json, sys, collections
tag, repo = sys.argv[1], sys.argv[2]
def kind(name, lds):
    n = name.replace("l2z::(anonymous namespace)::", "")
    if "matvec_row_kernel<1, 1" in n: return "qkv"
    if "matvec_row_kernel<1, 3" in n: return "ffn13"
    if "matvec_row_kernel<1, 4" in n: return "cls"
    if "matvec_row_kernel<0, 2, 12" in n: return "ffn2"   # n = 11008: x staged 12 float4 per thread
    if "matvec_row_kernel<0, 2, 4" in n: return "wo"      # n = 4096
    if "attention" in n: return "attn"
    return None
acc = {c: collections.defaultdict(float) for c in ("FETCH_SIZE", "WRITE_SIZE")}
cnt = collections.Counter()
for c in acc:
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for", c); sys.exit(1)
    recs = sorted((r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == c), key=lambda r: int(r["Dispatch_Id"]))
    n_resid = 0
    for r in recs:
        k = kind(r["Kernel_Name"], int(r["LDS_Block_Size"]))
        if k is None: continue
        acc[c][k] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": cnt[k] += 1
dim, hid, V = 4096, 11008, 32000
alg = {"qkv": 4 * 3 * dim * dim, "wo": 4 * dim * dim, "ffn13": 4 * 2 * hid * dim, "ffn2": 4 * dim * hid, "cls": 4 * V * dim}
out, rows = {}, []
for k in ("qkv", "wo", "ffn13", "ffn2", "cls", "attn"):
    if not cnt[k]: continue
    rd_kb = acc["FETCH_SIZE"][k] / cnt[k]; wr_kb = acc["WRITE_SIZE"][k] / cnt[k]
    rd = rd_kb * 1024 * 2  # gfx950: FETCH_SIZE reports half of a wide coalesced stream (guide, HBM section)
    tot = rd + wr_kb * 1024
    out[k] = int(tot)
    a = alg.get(k)
    rows.append(f"| {k} | {cnt[k]} | {rd_kb:.1f} | {rd:.0f} | {wr_kb:.1f} | {tot:.0f} | {a if a else '-'} | {tot / a:.4f} |" if a else f"| {k} | {cnt[k]} | {rd_kb:.1f} | {rd:.0f} | {wr_kb:.1f} | {tot:.0f} | - | - |")
method = ("rocprofv3 --pmc FETCH_SIZE --kernel-trace and, in a separate pass, --pmc WRITE_SIZE --kernel-trace "
          "(L2Z_NO_GRAPH=1, bench.py --steps 4, llama2-7b shape; scripts/pmc_traffic.sh). FETCH_SIZE is in KB and on gfx950 "
          "reports exactly half of a wide coalesced stream (MI355X_MICROARCH.md, HBM section), so read bytes = "
          "FETCH_SIZE*1024*2; WRITE_SIZE*1024 as is. Values are HBM bytes per launch = corrected read + write.")
json.dump({"_method": method, "llama2-7b": out}, open(f"{repo}/gpurun_out/pmc_traffic.json", "w"), indent=1)
with open(f"{repo}/gpurun_out/{tag}_llama2-7b_pmc_traffic.md", "w") as f:
    f.write(f"# PMC HBM traffic per launch, llama2-7b shape ({tag})\n\n{method}\n\n")
    f.write("| kernel kind | launches | FETCH_SIZE KB (raw) | read bytes (x2) | WRITE_SIZE KB | HBM bytes/launch | algorithmic bytes | traffic / algorithmic |\n|---|---:|---:|---:|---:|---:|---:|---:|\n")
    f.write("\n".join(rows) + "\n")
print(open(f"{repo}/gpurun_out/{tag}_llama2-7b_pmc_traffic.md").read())
PY
