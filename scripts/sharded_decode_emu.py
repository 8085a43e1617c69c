"""The decode twin of sharded_prefill_emu.py: what ONE rank's launches of a sharded decode token cost, per kind, on
emulated ranks (bench.py scaling_model), at a few positions.  Run under rocprofv3 --kernel-trace for the kernel table.
   sharded_decode_emu.py [pos ...]"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
import bench
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}["llama2-7b"]
for pos in [int(x) for x in sys.argv[1:]] or [8, 1024]:
    m = bench.scaling_model(B, cfg, shared, 2024, pos)
    print(f"\n## llama2-7b, pos {pos}: one rank's launches of a sharded token (emulated rank 0 of N, one GPU)\n")
    print("| N | per-rank ms | launches | qkv | attn | wo | ffn13 | ffn2 | cls | argmax | bytes / 7.3 TB/s ms | fixed us per launch | tok/s upper bound | speed-up bound |")
    print("|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for n, v in m.items():
        if n == "note": continue
        u = v["us_by_kind"]
        print(f"| {n} | {v['per_rank_ms']:.3f} | {v['launches']} | " + " | ".join(f"{u[k]:.2f}" for k in ("qkv", "attn", "wo", "ffn13", "ffn2", "cls", "argmax")) +
              f" | {v['weight_bytes_per_rank'] / 7.3e12 * 1e3:.3f} | {v['fixed_us_per_launch']:.2f} | {v['predicted_tok_s_upper_bound']:.0f} | {v['speedup_upper_bound_vs_1']:.2f} |")
    print("\n" + m["note"])
