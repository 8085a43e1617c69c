"""One rank of a multi-PROCESS sharded run on one GPU (tests/test_gpu_p2p.py spawns these).

usage: p2p_worker.py <rank> <world> <dir> <model.json>
The ranks exchange the IPC handles of their landing arenas through files in <dir>, connect the
peer-write all-gather (l2z_comm_p2p_*), run the greedy loop on a sharded toy checkpoint and write
tokens + final logits to <dir>/out_<rank>.npz.  No RCCL, no torch.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def tie_classifier_rows(ck, cfg, blob, shared, rows) -> None:
    """Classifier rows a, b (different ranks' shards) := 64 x row `base`, rows c, d := -64 x it: whichever sign the final
    activations give, the largest logit is (at most steps) attained at TWO indices owned by different ranks, exactly (the
    same row against the same x).  main.zig:720: the lower index must win -- across the ranks' candidate exchange as well."""
    a, b, c, d, base = rows
    assert not shared, "the tie is built in wcls (an unshared classifier)"
    wcls = ck.carve(cfg, blob, shared)["wcls"]
    v = np.float32(64.0) * wcls[base].copy()
    wcls[a] = v; wcls[b] = v; wcls[c] = -v; wcls[d] = -v


def main() -> None:
    rank, world, d, spec = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], json.load(open(sys.argv[4]))
    pkg = ge.load_package()
    B, ck = pkg.binding, pkg.checkpoint
    cfg = ck.Config(**spec["cfg"])
    shared, seed = spec["shared"], spec["seed"]
    expect = spec.get("expect", "ok")
    comm = B.Comm(rank, world, None, 0)  # every rank on device 0: this is a one-GPU test
    longest = max(cfg.dim, cfg.hidden_dim, cfg.vocab_size)
    scheme_b = os.environ.get("L2Z_SCHEME_B") == "1"
    if scheme_b:  # every rank's whole partial [dim] vector lands in every slot
        longest = max(longest, world * cfg.dim)
    h = comm.p2p_export(longest // 2 if expect == "slot_error" else longest, max(cfg.dim, cfg.hidden_dim))
    with open(os.path.join(d, f"h_{rank}.tmp"), "wb") as f:
        f.write(h)
    os.rename(os.path.join(d, f"h_{rank}.tmp"), os.path.join(d, f"h_{rank}.bin"))
    handles, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(d, f"h_{r}.bin")
        while not os.path.exists(p):
            if time.time() - t0 > 120:
                raise SystemExit(f"rank {rank}: no handle from rank {r}")
            time.sleep(0.01)
        handles.append(open(p, "rb").read())
    comm.p2p_connect(b"".join(handles))
    if expect == "slot_error":
        # landing slots shorter than the longest gathered vector: producers would store past the
        # end of the peers' arenas, so the runstate must be refused (L2Z_ERR_COMM), not built
        try:
            B.RunState(cfg, comm=comm)
        except B.L2ZError as e:
            assert e.code == B.ERR_COMM, e
            open(os.path.join(d, f"ok_{rank}"), "w").write(str(e))
            t0 = time.time()  # keep the arena alive until every rank has mapped it and reported
            while not all(os.path.exists(os.path.join(d, f"ok_{r}")) for r in range(world)) and time.time() - t0 < 60:
                time.sleep(0.02)
            comm.close()
            return
        raise SystemExit("runstate_init accepted landing slots that are too small")
    if expect == "peer_dies" and rank != 0:
        # this rank never runs: rank 0's first wait must time out ONCE and every later wait give up
        # at once, so that L2Z_ERR_COMM reaches its host quickly instead of after steps x gathers x timeout
        t0 = time.time()  # keep the arena mapped until rank 0 has reported
        while not os.path.exists(os.path.join(d, "ok_0")) and time.time() - t0 < 100:
            time.sleep(0.05)
        comm.close()
        return
    blob = ck.synth_blob(cfg, shared, seed) if spec.get("blob", True) else None
    if spec.get("tie_rows"):
        tie_classifier_rows(ck, cfg, blob, shared, spec["tie_rows"])
    w = B.Weights(cfg, blob, shared, seed=seed, comm=comm)
    s = B.RunState(cfg, comm=comm)
    assert (s.form() & 8 != 0) == scheme_b, f"rank {rank}: runstate form {s.form()}, L2Z_SCHEME_B={scheme_b}"
    s.greedy_begin(spec["prompt"])
    if expect == "peer_dies":
        t0 = time.time()
        try:
            s.greedy_run(w, spec["steps"])
        except B.L2ZError as e:
            assert e.code == B.ERR_COMM, e
            open(os.path.join(d, f"ok_{rank}"), "w").write(f"{time.time() - t0:.2f} {e}")
            return  # no clean-up: the group is dead
        raise SystemExit("greedy_run succeeded although the peer never ran")
    toks = s.greedy_run(w, spec["steps"])
    logits = s.logits()
    # the stepped API on top of the same state
    s.transformer(int(toks[-1]), len(toks) % cfg.seq_len, w)
    logits2, am = s.logits(), s.argmax()
    extra = {}
    if spec.get("prefill"):
        # l2z_prefill itself on the sharded runstate: [tokens, n / world] blocks through the bulk regions,
        # then the classifier over this rank's vocabulary rows and the logits gather
        s2 = B.RunState(cfg, comm=comm)
        n1 = spec["prefill_split"]
        if expect == "no_bulk":
            # L2Z_P2P_BULK_MB=0: the arena has no bulk regions, so l2z_prefill must refuse (and the greedy
            # loop above has stepped through its prompt instead)
            try:
                s2.prefill(spec["prefill"][:n1], 0, w)
            except B.L2ZError as e:
                assert e.code == B.ERR_INVALID, e
                np.savez(os.path.join(d, f"out_{rank}.npz"), toks=toks, logits=logits)
                s2.close(); s.close(); w.close(); comm.close()
                return
            raise SystemExit("l2z_prefill ran on a sharded runstate without bulk regions")
        s2.prefill(spec["prefill"][:n1], 0, w)
        s2.prefill(spec["prefill"][n1:], n1, w)
        extra = dict(pf_logits=s2.logits(), pf_key0=s2.read("key_cache", 0, len(spec["prefill"]) * (cfg.kv_dim // world)))
        s2.close()
    np.savez(os.path.join(d, f"out_{rank}.npz"), toks=toks, logits=logits, logits2=logits2, am=am, **extra)
    s.close(); w.close(); comm.close()


if __name__ == "__main__":
    main()
