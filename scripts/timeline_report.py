#!/usr/bin/env python3
"""Timeline of the decode launches from a rocprofv3 --kernel-trace results .db: per kind of launch the average
duration, the gap to the previous launch of the SAME queue (the stream-ordered boundary), the gap to the end of the
launch that produced its input (the hand-over), and one layer of the last token as a relative timeline.
usage: timeline_report.py results.db [label]"""
import sqlite3, sys, collections
db = sys.argv[1]; label = sys.argv[2] if len(sys.argv) > 2 else db
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name, queue_id, start, end, grid_x, workgroup_x, lds_size from kernels order by start"))
def kind(name, lds):
    if "argmax_kernel" in name: return "argmax"
    if "attention" in name: return "attn"
    if "matvec" in name:
        a = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",")
        pro, epi = a[0], a[1]
        if epi in ("1", "(l2z::Epilogue)1"): return "qkv"
        if epi in ("3", "(l2z::Epilogue)3"): return "ffn13"
        if epi in ("4", "(l2z::Epilogue)4"): return "cls"
        if epi in ("2", "(l2z::Epilogue)2"): return "ffn2" if lds > 40000 else "wo"
    return None
ks = [(kind(n, l), q, s, e, n) for n, q, s, e, g, wg, l in rows]
ks = [k for k in ks if k[0]]
# keep the last 4 tokens (a token ends with argmax)
ends = [i for i, k in enumerate(ks) if k[0] == "argmax"]
if len(ends) < 6: sys.exit("too few tokens in the trace")
lo, hi = ends[-5] + 1, ends[-1] + 1
seq = ks[lo:hi]
order = ["qkv", "attn", "wo", "ffn13", "ffn2", "cls", "argmax"]
prod = {"attn": "qkv", "wo": "attn", "ffn13": "wo", "ffn2": "ffn13", "qkv": "ffn2", "cls": "ffn2", "argmax": "cls"}
dur = collections.defaultdict(list); qgap = collections.defaultdict(list); hand = collections.defaultdict(list); tail = collections.defaultdict(list)
last_in_q = {}; last_of = {}
for k, q, s, e, n in seq:
    dur[k].append(e - s)
    if q in last_in_q: qgap[k].append(s - last_in_q[q])
    p = prod.get(k)
    if p in last_of:
        hand[k].append(s - last_of[p])      # < 0: resident before its producer ended
        tail[k].append(e - last_of[p])      # producer's end -> this launch's end
    last_in_q[q] = e; last_of[k] = e
tok = (seq[-1][3] - ks[ends[-5]][3]) / 4.0
print(f"# decode timeline: {label}\n")
print(f"4 tokens, {len(seq)} launches, {tok / 1e3:.1f} us per token (end of argmax to end of argmax), queues used: {sorted(set(k[1] for k in seq))}\n")
print("| launch | n | duration us | start - end of previous launch in its queue | start - end of its producer | end - end of its producer |")
print("|---|---:|---:|---:|---:|---:|")
avg = lambda x: sum(x) / len(x) / 1e3 if x else float("nan")
for k in order:
    print(f"| {k} | {len(dur[k])} | {avg(dur[k]):.2f} | {avg(qgap[k]):.2f} | {avg(hand[k]):.2f} | {avg(tail[k]):.2f} |")
print(f"\nsum over a layer of (end - end of producer) = {sum(avg(tail[k]) for k in ['qkv', 'attn', 'wo', 'ffn13', 'ffn2']):.2f} us\n")
# one layer of the last token
t_ends = [i for i, k in enumerate(seq) if k[0] == "argmax"]
tokseq = seq[t_ends[-2] + 1:t_ends[-1] + 1]
qi = [i for i, k in enumerate(tokseq) if k[0] == "qkv"]
l0 = qi[len(qi) // 2]
t0 = tokseq[l0][2]
print("layer in the middle of the last token (us relative to its qkv start):\n")
print("| launch | queue | start | end |")
print("|---|---:|---:|---:|")
for k, q, s, e, n in tokseq[l0 - 1:l0 + 7]:
    print(f"| {k} | {q} | {(s - t0) / 1e3:.2f} | {(e - t0) / 1e3:.2f} |")
