"""ONE rank of an N-rank shard group alone on the GPU, its whole sharded decode pass with free hand-overs
(l2z_comm_p2p_connect_solo: the peers' arenas are a local sink, this rank's zeroed landing slots satisfy every wait): tokens/s of the
rank = an UPPER bound on tokens/s at N GPUs for each structure -- launches, pushes, polls, gather / reduce launches and
graph replay included (scaling_model's per-kind sums leave those out); hand-over latency, rank skew and xGMI are not.
usage: solo_rank.py [workload] [steps]            the table, N = 2 / 4 / 8 x every structure
       solo_rank.py <workload> <steps> <N> <k>    only structure number k (0 .. 2) at N (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, __graft_entry__ as ge
pkg = ge.load_package(); B, ck = pkg.binding, pkg.checkpoint
wl = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg, shared = {n: (c, sh) for n, c, sh in ck.iter_configs()}[wl]
FORMS = [("A: consumer-side words (p2p-consume)", {"L2Z_P2P_CONSUME": 1}),
         ("A: gather launch per vector (p2p-gather)", {"L2Z_P2P_CONSUME": 0}),
         ("B: column shards + reduce launches (p2p-allreduce)", {"L2Z_SCHEME_B": 1})]
RESET = {"L2Z_P2P_CONSUME": -1, "L2Z_SCHEME_B": 0}
B.option_set("L2Z_PREFILL", 0)


PROFS = []


def run(world, opts):
    for k, v in opts.items(): B.option_set(k, v)
    comm = w = s = None
    try:
        if world > 1:
            comm = B.Comm(0, world, None, 0)
            comm.p2p_export(max(cfg.dim, cfg.hidden_dim, cfg.vocab_size, world * cfg.dim), max(cfg.dim, cfg.hidden_dim))
            comm.p2p_connect_solo()
        w = B.Weights(cfg, None, shared, seed=7, comm=comm)
        s = B.RunState(cfg, comm=comm)
        form = s.form()
        s.greedy_begin([]); s.greedy_run(w, 4); s.synchronize()
        best = 0.0
        for _ in range(3):
            s.greedy_begin([]); s.greedy_run(w, 2); s.synchronize()
            t0 = time.perf_counter(); n = len(s.greedy_run(w, steps)); s.synchronize()
            best = max(best, n / (time.perf_counter() - t0))
        prof = None
        if world == 8:   # in situ, an event pair around every launch (adds ~3 us to each)
            acc = {}
            for i in range(4):
                for k, (ms, cnt) in s.profile_forward(1 + i, 8 + i, w).items():
                    a = acc.setdefault(k, [0.0, 0]); a[0] += ms; a[1] += cnt
            prof = {k: (1e3 * a[0] / a[1], a[1] // 4) for k, a in acc.items() if a[1]}
        PROFS.append((world, dict(opts), prof))
        return best, form
    finally:
        for k in opts: B.option_set(k, RESET[k])
        for o in (s, w, comm):
            if o is not None: o.close()


if len(sys.argv) > 4:
    name, opts = FORMS[int(sys.argv[4])]
    v, form = run(int(sys.argv[3]), opts)
    print(f"{wl} N = {sys.argv[3]} {name}: {v:.1f} tok/s = {1e3 / v:.3f} ms per token, runstate form {form}")
    sys.exit(0)
base, _ = run(1, {})
print(f"# {wl}: one rank of N alone on the GPU, hand-overs free (solo connect), best of 3 x {steps} greedy steps; N = 1: {base:.1f} tok/s\n")
print("| structure | " + " | ".join(f"N = {n}: tok/s (x of N = 1)" for n in (2, 4, 8)) + " |\n|---|" + "---:|" * 3)
for name, opts in FORMS:
    cells = []
    for world in (2, 4, 8):
        try:
            v, form = run(world, opts)
            want = 8 if "L2Z_SCHEME_B" in opts else 0
            cells.append(f"{v:.0f} ({v / base:.2f})" + ("" if (form & 8) == want else f" [form {form}!]"))
        except Exception as e:  # noqa: BLE001
            cells.append(f"failed: {str(e)[:40]}")
    print(f"| {name} | " + " | ".join(cells) + " |")
print("\nN = 8, us per launch by kind with an event pair around every launch (launches per token):\n")
for world, opts, prof in PROFS:
    if prof:
        print(f"* {opts or 'default'}: " + ", ".join(f"{k} {v[0]:.1f} ({v[1]})" for k, v in prof.items()))

