#!/usr/bin/env python3
"""Median duration of the four stream-form GEMM launches of a layer (q|k|v, wo, W1|W3, W2) from a rocprofv3 kernel-trace .db
(wo and W2 are the same kernel: told apart by their order in the stream).  usage: stream_sk_scan.py results.db label"""
import sqlite3, sys, statistics
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x from kernels where name like '%x3_stream%' order by start"))
out = {"qkv": [], "wo": [], "w13": [], "w2": []}
n1 = 0
for name, st, en, grid in rows:
    epi = int(name.split("prefill_x3_stream<")[1].split(",")[0])
    if epi == 6: out["qkv"].append((en - st, grid))
    elif epi == 7: out["w13"].append((en - st, grid))
    elif epi == 1:
        out["wo" if n1 % 2 == 0 else "w2"].append((en - st, grid)); n1 += 1
small = list(cur.execute("select name, avg(end-start), count(*) from kernels where name not like '%x3_stream%' and name not like '%synth%' and name not like '%rocclr%' group by name"))
line = "  ".join(f"{k} {statistics.median(d for d, _ in v)/1e3:6.1f} us ({v[0][1]//512} blocks)" for k, v in out.items() if v)
tot = sum(statistics.median(d for d, _ in v) for v in out.values() if v) / 1e3
print(f"{sys.argv[2]}: {line}  sum {tot:.1f} us")
