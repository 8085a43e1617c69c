#!/usr/bin/env python3
"""bench.py -- tokens/s of the llama2.zig forward pass on MI355X, -t 0 (argmax).

Metric (BASELINE.json): "tokens/s (argmax, -t 0) + matvec achieved HBM GB/s vs peak".
A "step" is one pass of the hot path = one generated position: transformer()
(src/main.zig:285) + argmax (:715) + the loop hand-over (:999-1036), all on the
device.  Weights are already resident in HBM when the timed region starts
(synthetic llama2.c-v0 checkpoint of the named shape, generated on device).

  python bench.py                       # N=1, llama2-7b shape, 255 steps after 1 warm-up
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N   # rows/heads sharded over N GPUs (RCCL, strong scaling)

The default (steps=255, warmup=1) is exactly the reference's `-n 256 -t 0 -v`
figure: its clock starts after the first token and the rate is (pos-1)/elapsed
(src/main.zig:1039-1047).

Rank 0 prints ONE JSON line.  It carries `roofline` for the dominant kernel
(HIP-event time measured in situ by l2z_profile_forward) and, at N=1,
`cpu_baseline`: the C oracle (a port, 1 thread -- the reference is single
threaded, main.zig:5 / README.md:107) timed on this box's host cores on a
bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E vendor peak (MI355X_MICROARCH.md); ~6300 measured copy
MFMA_F32_PEAK_TF = 157.3  # dense fp32 matrix peak (256 CUs x 256 flop/clk x 2.4 GHz)


def weight_bytes_by_kind(cfg, world: int = 1) -> dict:
    """Algorithmic HBM bytes ONE launch of each kernel kind must stream:
    a (d,n) mat-vec moves 4*d*n bytes; attention reads K and V rows 0..pos."""
    dim, hid, V = cfg.dim, cfg.hidden_dim, cfg.vocab_size
    kvd = cfg.kv_dim
    return {
        "qkv": 4 * (dim * dim + 2 * kvd * dim) // world,
        "wo": 4 * dim * dim // world,
        "ffn13": 4 * 2 * hid * dim // world,
        "ffn2": 4 * dim * hid // world,
        "cls": 4 * V * dim // world,
    }


def cpu_baseline(ck, cfg, shared, name: str) -> dict:
    """Time the C oracle (port of src/main.zig, 1 thread) on a bounded sample."""
    orc = ge.load_oracle()
    orc.set_mode(8, True, True)  # AVX2 width, fused -- the fastest reading of the reference
    ncpu = os.cpu_count() or 1
    if cfg.n_layers <= 12 and ck.weights_count(cfg, shared) * 4 < (1 << 30):
        blob = orc.synth_fill(cfg.as_i32(), shared, 1, ncpu)
        m = orc.Model(cfg.as_i32(), blob, shared)
        n_tok = 64
        t0 = time.perf_counter()
        toks, _ = m.generate_greedy([], n_tok)
        dt = time.perf_counter() - t0
        m.close()
        return {"value": len(toks) / dt, "unit": "tokens/s", "cores": 1, "kind": "port",
                "sample": f"{name}: full model, {len(toks)} greedy tokens from BOS, C oracle "
                          f"(oracle/llama2_oracle.c, gcc -O3 AVX2+FMA), 1 thread of {ncpu}"}
    # big shape, big host: the real thing -- the full model on one host thread for a dozen tokens
    # (BASELINE.md's plan: "7B on CPU uses -n 16"); the multi-threaded fill is not timed
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:  # noqa: BLE001
        free = 0
    need = ck.weights_count(cfg, shared) * 4
    if free > 2 * need + (8 << 30):
        blob = orc.synth_fill(cfg.as_i32(), shared, 1, ncpu)
        m = orc.Model(cfg.as_i32(), blob, shared)
        m.transformer(1, 0)  # page everything in once
        n_tok = 12
        t0 = time.perf_counter()
        toks, _ = m.generate_greedy([], n_tok)
        dt = time.perf_counter() - t0
        m.close()
        del blob
        return {"value": len(toks) / dt, "unit": "tokens/s", "cores": 1, "kind": "port",
                "sample": f"{name}: full model ({need / 1e9:.1f} GB of weights on the host), {len(toks)} greedy "
                          f"tokens from BOS, C oracle (oracle/llama2_oracle.c, gcc -O3 AVX2+FMA), 1 thread of {ncpu}"}
    # big shape, small host: time 1-layer and 3-layer models of the same dims, extrapolate layers linearly
    times = {}
    n_tok = 3
    for L in (1, 3):
        c = ck.Config(cfg.dim, cfg.hidden_dim, L, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size,
                      cfg.seq_len)
        blob = orc.synth_fill(c.as_i32(), shared, 1, ncpu)
        m = orc.Model(c.as_i32(), blob, shared)
        m.transformer(1, 0)  # touch everything once
        t0 = time.perf_counter()
        for p in range(1, 1 + n_tok):
            m.transformer(7 * p, p)
        times[L] = (time.perf_counter() - t0) / n_tok
        m.close()
        del blob
    t_layer = (times[3] - times[1]) / 2
    t_rest = max(times[1] - t_layer, 0.0)
    t_full = t_rest + cfg.n_layers * t_layer
    return {"value": 1.0 / t_full, "unit": "tokens/s", "cores": 1, "kind": "port",
            "sample": f"{name}: same dims with 1 and 3 layers, {n_tok} tokens each, C oracle 1 thread "
                      f"of {ncpu}; per-layer {t_layer*1e3:.1f} ms, classifier+rest {t_rest*1e3:.1f} ms, "
                      f"extrapolated to {cfg.n_layers} layers"}


def run_once(B, cfg, shared, seed, steps, warmup, comm=None, sync_ok=None):
    """Returns (tokens produced in the timed region, elapsed seconds, runstate, weights).

    sync_ok(ok) -> bool is the multi-rank barrier: every rank reports whether its phase worked and
    learns whether all did, so that a failure on one rank makes ALL ranks raise here instead of
    leaving the others waiting in a barrier."""
    sync_ok = sync_ok or (lambda ok: ok)
    s = w = None
    err = None
    try:
        w = B.Weights(cfg, None, shared, seed=seed, comm=comm)
        s = B.RunState(cfg, comm=comm)
        s.greedy_begin([])
        if warmup > 0:
            s.greedy_run(w, warmup)
        s.synchronize()
    except Exception as e:  # noqa: BLE001
        err = e
    if not sync_ok(err is None):
        for o in (s, w):
            if o is not None:
                o.close()
        raise RuntimeError(f"set-up / warm-up failed on some rank ({err})")
    t0 = time.perf_counter()
    toks = ()
    try:
        toks = s.greedy_run(w, steps)  # synchronises before returning the tokens
        s.synchronize()
    except Exception as e:  # noqa: BLE001
        err = e
    ok = sync_ok(err is None)
    dt = time.perf_counter() - t0
    if not ok:
        s.close(); w.close()
        raise RuntimeError(f"timed region failed on some rank ({err})")
    return len(toks), dt, s, w


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=255)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="llama2-7b",
                    choices=["llama2-7b", "stories110M", "stories15M"])
    ap.add_argument("--seed", type=int, default=2024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the stories15M side measurement")
    args = ap.parse_args()

    pkg = ge.load_package()
    B, ck = pkg.binding, pkg.checkpoint
    shapes = {n: (c, sh) for n, c, sh in ck.iter_configs()}
    cfg, shared = shapes[args.workload]
    steps = max(1, min(args.steps, cfg.seq_len - args.warmup))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    comm, dist = None, None
    transport = None
    force_dist = os.environ.get("L2Z_BENCH_FORCE_DIST") == "1"  # 1-rank RCCL + gloo, for testing
    if args.gpus > 1 or world > 1 or force_dist:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with "
                             "python -m torch.distributed.run --nproc-per-node N ...)")
        # torch is imported BEFORE libllama2_hip.so is loaded: the other order leaves HIP
        # without a visible device on this image (measured on the MI355X box)
        # a peer that never shows up must fail the peer-write attempt quickly, so that the RCCL
        # fallback still fits the run (ranks are within a second of each other after the handshake)
        os.environ.setdefault("L2Z_P2P_TIMEOUT_S", "8")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (the only kind the host driver has)
        import torch
        import torch.distributed as dist
        # control plane only (barrier, handle/id exchange, max-reduce of the clock): gloo on CPU.
        # The data path's all-gathers are issued by libllama2_hip.so itself.
        dist.init_process_group("gloo", rank=rank, world_size=world)

    if B.device_count() < 1:
        raise SystemExit("bench.py: no HIP device visible (the HIP path has no CPU fallback)")
    n_dev = B.device_count()
    device = local_rank % n_dev  # several ranks on one GPU only happens in tests
    if world > n_dev:
        # ranks share a chip: a mat-vec launch that may be polling for a peer's words must leave the
        # peer's kernels room to run (never needed with a GPU per rank).  Measured: with more than
        # 512 polling blocks in total (2 per CU) a peer's producer can be left without a slot and the
        # waits time out; the gather-launch form needs no cap (L2Z_COMM=p2p-gather L2Z_GRID_CAP=0)
        shared_cap = max(32, 512 // ((world + n_dev - 1) // n_dev))
    else:
        shared_cap = 0

    def all_ok(ok: bool) -> bool:
        flags = [None] * world
        dist.all_gather_object(flags, bool(ok))
        return all(flags)

    def make_comm(kind: str):
        """'p2p': peer-write gathers over IPC-mapped arenas (xGMI between GPUs); 'rccl': RCCL."""
        if kind == "p2p":
            c, h = None, b""
            try:
                c = B.Comm(rank, world, None, device)
                h = c.p2p_export(max(cfg.dim, cfg.hidden_dim, cfg.vocab_size))
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] peer-write export failed: {e}", file=sys.stderr)
            hs = [None] * world
            dist.all_gather_object(hs, h)
            ok = c is not None and all(len(x) == B.COMM_IPC_BYTES for x in hs)
            if ok:
                try:
                    c.p2p_connect(b"".join(hs))
                except Exception as e:  # noqa: BLE001
                    print(f"[rank {rank}] peer-write connect failed: {e}", file=sys.stderr)
                    ok = False
            if all_ok(ok):
                return c
            if c is not None:
                c.close()
            return None
        c = None
        try:
            uid = [B.Comm.unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] ncclGetUniqueId failed: {e}", file=sys.stderr)
            uid = [None]
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is not None:
            try:
                c = B.Comm(rank, world, uid[0], device)
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] RCCL communicator failed: {e}", file=sys.stderr)
        if all_ok(c is not None):
            return c
        if c is not None:
            c.close()
        return None

    def ranks_agree(s) -> bool:
        """Every rank must hold the same logits after the same steps (bit for bit)."""
        lg = s.logits()
        sig = (int(np.argmax(lg)), float(lg.astype(np.float64).sum()), float(np.abs(lg).max()))
        sigs = [None] * world
        dist.all_gather_object(sigs, sig)
        return all(x == sigs[0] for x in sigs) and bool(np.isfinite(lg).all())

    agree = None
    attempts = []
    if dist is not None:
        # Transports for the per-layer gathers, best first.  A form that cannot be set
        # up, times out, or leaves the ranks with different logits is dropped for the next one.
        #   p2p-consume  producers store LL words into every rank's landing slot, consumers poll
        #                them while staging x: 5 graph nodes per layer, one gather launch per token
        #   p2p-gather   the same stores, collected by a gather launch per vector (9 nodes per layer)
        #   rccl         ncclAllGather per vector, captured into the step graph when that works
        want = os.environ.get("L2Z_COMM", "")
        order = {"": ["p2p-consume", "p2p-gather", "rccl"], "p2p": ["p2p-consume", "p2p-gather", "rccl"],
                 "p2p-consume": ["p2p-consume"], "p2p-gather": ["p2p-gather"], "rccl": ["rccl"]}[want]
        if force_dist:
            order = ["rccl"]
        for kind in order:
            B.option_set("L2Z_P2P_CONSUME", 1 if kind == "p2p-consume" else 0)
            B.option_set("L2Z_GRID_CAP", shared_cap if kind == "p2p-consume" else 0)
            B.option_set("L2Z_COMM_RCCL", 1 if kind == "rccl" else 0)
            comm = make_comm("rccl" if kind == "rccl" else "p2p")
            if comm is None:
                attempts.append({"transport": kind, "ok": False, "why": "set-up failed"})
                continue
            transport = kind
            s = w = None
            try:
                n_tok, dt, s, w = run_once(B, cfg, shared, args.seed, steps, args.warmup, comm, all_ok)
                ran = True
            except Exception as e:  # noqa: BLE001  (a gather timed out, a launch failed, ...)
                print(f"[rank {rank}] run with transport {transport} failed: {e}", file=sys.stderr)
                ran = False
            agree = ranks_agree(s) if all_ok(ran) else False
            attempts.append({"transport": kind, "ok": bool(agree), "why": None if agree else
                             ("ranks disagree" if ran else "run failed")})
            if agree:
                break
            print(f"[rank {rank}] transport {transport} unusable here (ran={ran}); trying the next one",
                  file=sys.stderr)
            for o in (s, w, comm):
                if o is not None:
                    o.close()
            comm = None
        if comm is None:
            raise SystemExit(f"bench.py: no working transport for the shard group: {attempts}")
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        n_tok, dt, s, w = run_once(B, cfg, shared, args.seed, steps, args.warmup, None, None)

    # ---- roofline of the dominant kernel, HIP events in situ ----
    by_kind = {k: [0.0, 0] for k in B.KINDS}
    pos0 = args.warmup + n_tok
    n_prof = 4
    for i in range(n_prof):
        p = min(pos0 + i, cfg.seq_len - 1)
        for k, (ms, cnt) in s.profile_forward(1 + i, p, w).items():
            by_kind[k][0] += ms
            by_kind[k][1] += cnt
    wb = weight_bytes_by_kind(cfg, world)
    dom = max(wb, key=lambda k: by_kind[k][0])
    avg_ms = by_kind[dom][0] / max(by_kind[dom][1], 1)
    timing = "HIP event pair around every launch, in situ (adds ~3 us per launch)"
    b2b = {}
    if world == 1:
        # kernel durations without the per-launch event overhead: every layer's launch of one kind back to
        # back between ONE event pair (l2z_time_kind) -- what rocprofv3's kernel trace reports as well
        try:
            for k in B.KINDS[:7]:
                if by_kind[k][1]:
                    ms_k, n_k = s.time_kind(k, min(pos0, cfg.seq_len - 16), w, reps=4)
                    b2b[k] = {"ms_per_launch": ms_k, "launches_timed": n_k,
                              "GBps": wb[k] / (ms_k * 1e-3) / 1e9 if k in wb and ms_k > 0 else None}
            if dom in b2b:
                avg_ms = b2b[dom]["ms_per_launch"]
                timing = ("one HIP event pair around the launches of this kind for all layers back to back, "
                          "4 passes (l2z_time_kind): average kernel duration, no per-launch event overhead")
        except Exception as e:  # noqa: BLE001
            b2b = {"error": str(e)}
    achieved = wb[dom] / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath) and world == 1:  # the PMC passes were taken unsharded
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
        except Exception:
            traffic = None
    kernels = {k: {"ms_per_launch": by_kind[k][0] / max(by_kind[k][1], 1),
                   "launches_per_token": by_kind[k][1] // n_prof,
                   "GBps": (wb[k] / (by_kind[k][0] / max(by_kind[k][1], 1) * 1e-3) / 1e9)
                   if k in wb and by_kind[k][0] > 0 else None}
               for k in B.KINDS}
    # the measured ceiling on this box: a pure streaming-read kernel over the same resident
    # weights, in pieces the size of the dominant launch -- context, not the peak
    try:
        rd_avg, rd_best = s.stream_read_probe(w, wb[dom], 12)
    except Exception:
        rd_avg = rd_best = None
    roofline = {"bound": "hbm", "kernel": f"matvec[{dom}]", "achieved": achieved,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic, "algorithmic_bytes_per_launch": wb[dom],
                "avg_launch_ms": avg_ms, "timing": timing, "by_kind_back_to_back": b2b,
                "by_kind": kernels,
                "measured_stream_read": {"avg": rd_avg, "best": rd_best, "unit": "GB/s",
                                         "frac_of_measured": (achieved / rd_avg) if rd_avg else None,
                                         "note": "pure nt-load kernel over the resident weights, "
                                                 "slices of the dominant launch's size"}}
    # ---- batched prompt prefill: the MFMA-bound part, reported beside the
    # decode figure, never folded into `value`
    prefill = None
    if rank == 0 and world == 1 and not args.no_extra:
        try:
            n_p = min(512, cfg.seq_len - 1)
            toks = [1] + np.random.default_rng(args.seed).integers(2, cfg.vocab_size, n_p - 1).tolist()
            s.prefill(toks, 0, w)  # allocations, first-touch
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                s.prefill(toks, 0, w)  # synchronises
            dtp = (time.perf_counter() - t0) / reps
            flops = 2.0 * n_p * (cfg.n_layers * (2 * cfg.dim * cfg.dim + 2 * cfg.dim * cfg.kv_dim
                                                 + 3 * cfg.dim * cfg.hidden_dim))
            prefill = {"prompt_tokens": n_p, "ms": dtp * 1e3, "tokens_per_s": n_p / dtp,
                       "roofline": {"bound": "mfma", "achieved": flops / dtp / 1e12, "peak": MFMA_F32_PEAK_TF,
                                    "unit": "TFLOP/s", "frac": flops / dtp / 1e12 / MFMA_F32_PEAK_TF,
                                    "note": "whole prefill (GEMMs + attention + norms), GEMM flops only "
                                            "in the numerator; v_mfma_f32_32x32x2_f32"}}
            # shorter prompts: other kernels (<= 64 tokens: weight-streaming bound short-prompt GEMMs; 65-256:
            # smaller tiles so that every CU has a block) -- ms per prompt length
            by_len = {}
            for n_s in (16, 128):
                if n_s < cfg.seq_len:
                    s.prefill(toks[:n_s], 0, w)
                    t0 = time.perf_counter()
                    for _ in range(3):
                        s.prefill(toks[:n_s], 0, w)
                    by_len[str(n_s)] = (time.perf_counter() - t0) / 3 * 1e3
            prefill["ms_by_prompt_tokens"] = by_len
        except Exception as e:  # noqa: BLE001
            prefill = {"error": str(e)}
    # sharded runstates: the row-sharded prefill (every rank its column blocks, [tokens, n / world] blocks
    # exchanged through the bulk regions / RCCL) -- a diagnostic beside the decode figure, like the above
    prefill_sharded = None
    if world > 1 and not args.no_extra and os.environ.get("L2Z_BENCH_NO_SHARDED_PREFILL", "") != "1":
        err = None
        dtp = 0.0
        n_p = min(512, cfg.seq_len - 1)
        try:
            toks = [1] + np.random.default_rng(args.seed).integers(2, cfg.vocab_size, n_p - 1).tolist()
            s.prefill(toks, 0, w)
            t0 = time.perf_counter()
            for _ in range(2):
                s.prefill(toks, 0, w)
            dtp = (time.perf_counter() - t0) / 2
        except Exception as e:  # noqa: BLE001  (no bulk regions, a wait timed out, ...)
            err = e
        if all_ok(err is None):
            t = torch.tensor([dtp], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            prefill_sharded = {"prompt_tokens": n_p, "ms": float(t.item()) * 1e3, "tokens_per_s": n_p / float(t.item()),
                               "ranks_agree": ranks_agree(s)}
        else:
            prefill_sharded = {"error": str(err) if err else "failed on another rank"}
    s.close()
    w.close()

    out = {
        "metric": "tokens/s (argmax, -t 0)", "value": n_tok / dt, "unit": "tokens/s",
        "n_gpus": args.gpus, "steps": n_tok, "warmup": args.warmup,
        "ms_per_step": dt / n_tok * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload} shape, fp32 llama2.c-v0 checkpoint "
                               f"(dim {cfg.dim}, hidden {cfg.hidden_dim}, L {cfg.n_layers}, "
                               f"H {cfg.n_heads}, kv {cfg.n_kv_heads}, V {cfg.vocab_size}, "
                               f"S {cfg.seq_len}), greedy from BOS, seeded synthetic weights",
                   "parallelism": (f"rows/heads sharded x{args.gpus}, all-gathers by "
                                   + {"p2p-consume": "peer writes of LL words over IPC-mapped memory (xGMI), "
                                                     "polled by the consuming mat-vec (no gather launch)",
                                      "p2p-gather": "peer writes over IPC-mapped memory (xGMI) + a gather "
                                                    "launch per vector",
                                      "rccl": "RCCL ncclAllGather"}[transport]) if transport else "1 GPU",
                   "ranks_agree": agree,
                   "weight_bytes_per_token": sum(
                       wb[k] * (1 if k == "cls" else cfg.n_layers) for k in wb) * world},
        "roofline": roofline,
    }
    if transport:
        n_g = 4 * cfg.n_layers + 1
        launches = by_kind["gather"][1] // n_prof
        ideal_ms = (sum(wb[k] * (1 if k == "cls" else cfg.n_layers) for k in wb) / (rd_avg * 1e9) * 1e3
                    if rd_avg else None)
        out["comm"] = {
            "transport": transport, "attempts": attempts, "prefill_sharded": prefill_sharded, "gathers_per_token": n_g,
            "gather_launches_per_token": launches,
            "us_per_gather_launch": (by_kind["gather"][0] / max(by_kind["gather"][1], 1) * 1e3) if launches else None,
            "graph_nodes_per_layer": 5 + (4 if launches > 1 else 0),
            # this rank's weight bytes at the streaming-read rate measured on this box: what a step
            # would take with free gathers; the rest of ms_per_step is gather + launch overhead
            "ideal_ms_per_step_at_measured_stream_rate": ideal_ms,
            "overhead_ms_per_step": (dt / n_tok * 1e3 - ideal_ms) if ideal_ms else None,
            "note": "kernel times in roofline.by_kind include the consumer-side polling of the gathered "
                    "input (p2p-consume) -- compare with the N=1 line",
        }
    if rank == 0 and args.gpus == 1:
        if not args.no_extra and args.workload != "stories15M":
            c15, sh15 = shapes["stories15M"]
            n15, dt15, s15, w15 = run_once(B, c15, sh15, args.seed, 255, 1)
            s15.close(); w15.close()
            # stories110M (BASELINE config 3's shape): 438 MB of weights per token do not fit the
            # 256 MB Infinity Cache, so an HBM fraction is meaningful, though launch latency dominates
            c110, sh110 = shapes["stories110M"]
            n110, dt110, s110, w110 = run_once(B, c110, sh110, args.seed, 255, 1)
            s110.close(); w110.close()
            wb110 = weight_bytes_by_kind(c110)
            bytes110 = sum(wb110[k] * (1 if k == "cls" else c110.n_layers) for k in wb110)
            # long context on the headline shape: the prompt fills the cache through the batched
            # prefill, then 32 greedy positions near the end of the 2048-token context are timed
            long_ctx = None
            if args.workload == "llama2-7b":
                try:
                    w7 = B.Weights(cfg, None, shared, seed=args.seed)
                    s7 = B.RunState(cfg)
                    n_p = cfg.seq_len - 48
                    prompt = np.random.default_rng(1).integers(2, cfg.vocab_size, n_p).tolist()
                    s7.greedy_begin(prompt)
                    s7.greedy_run(w7, n_p + 8)  # prefill + 8 warm-up positions
                    s7.synchronize()
                    t0 = time.perf_counter()
                    n_l = len(s7.greedy_run(w7, 32))
                    s7.synchronize()
                    dtl = time.perf_counter() - t0
                    pr = s7.profile_forward(1, cfg.seq_len - 1, w7)
                    kv_bytes = 8 * cfg.n_layers * (cfg.seq_len - 24) * cfg.kv_dim
                    long_ctx = {"positions": [n_p + 8, n_p + 8 + n_l - 1], "tokens_per_s": n_l / dtl,
                                "ms_per_token": dtl / n_l * 1e3,
                                "attention_us_per_layer_at_last_pos": pr["attn"][0] / max(pr["attn"][1], 1) * 1e3,
                                "kv_bytes_per_token": kv_bytes,
                                "hbm_frac_incl_kv": (out["config"]["weight_bytes_per_token"] + kv_bytes)
                                                    / (dtl / n_l) / 1e9 / HBM_PEAK_GBS}
                    s7.close(); w7.close()
                except Exception as e:  # noqa: BLE001
                    long_ctx = {"error": str(e)}
            out["extra"] = {"prefill": prefill,
                            "stories110M": {"tokens_per_s": n110 / dt110, "steps": n110,
                                            "weight_bytes_per_token": bytes110,
                                            "hbm_frac": bytes110 / (dt110 / n110) / 1e9 / HBM_PEAK_GBS},
                            "long_context": long_ctx,
                            "stories15M_tokens_per_s": n15 / dt15, "stories15M_steps": n15,
                            # the only figure the reference publishes: 660 tok/s, -t 0,
                            # stories15M, one Ryzen 9 5900X core, Zig 0.11 -- other hardware, indicative
                            "stories15M_vs_reference_readme_660": (n15 / dt15) / 660.0,
                            "note": "stories15M shape, -t 0 -n 256; weights fit the on-die "
                                    "cache, launch/latency bound, no HBM fraction quoted"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(ck, cfg, shared, args.workload)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
