"""World-size-2 (and 4) CPU test of the multi-GPU shard plan, over gloo.

What runs on the GPUs at N>1 (csrc/forward.cpp enqueue_forward, DESIGN.md
"Sharding") is: every rank owns whole heads of q/k/v, rows of wo / w1 / w3 / w2
and rows of the classifier, given by l2z_shard_range; after attention, after
each residual update, after the SwiGLU and after the classifier the owned
slices are all-gathered in place.  This test executes exactly that schedule
with the CPU oracle's kernels as the per-rank math and torch.distributed/gloo
all_gather as the collective, one process per rank, and checks the logits are
BIT-IDENTICAL to the unsharded pass: every output row is produced by the same
dot product in the same order whichever rank owns it (Scheme A),
so token ids cannot depend on the GPU count.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kw, shared, seed, toks, out_dir, scheme="A"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    ck, B = pkg.checkpoint, pkg.binding
    orc = ge.load_oracle()
    cfg = ck.Config(**kw)
    W = ck.carve(cfg, ck.synth_blob(cfg, shared, seed), shared)
    hs, kv_mul, S = cfg.head_size, cfg.kv_mul, cfg.seq_len
    # the product's own shard plan (C ABI, host-only)
    d0, d1 = B.shard_range(cfg.dim, hs, rank, world)
    k0, k1 = B.shard_range(cfg.kv_dim, hs, rank, world)
    h0, h1 = B.shard_range(cfg.hidden_dim, 1, rank, world)
    v0, v1 = B.shard_range(cfg.vocab_size, 1, rank, world)
    kc = np.zeros((cfg.n_layers, S, k1 - k0), np.float32)
    vc = np.zeros((cfg.n_layers, S, k1 - k0), np.float32)

    n_coll = [0]

    def allgather(buf, lo, hi):
        n_coll[0] += 1
        parts = [torch.zeros(hi - lo) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(buf[lo:hi].copy()))
        return np.concatenate([p.numpy() for p in parts]).astype(np.float32)

    def allreduce(part):
        """Scheme B (csrc/p2p.hip p2p_allreduce_kernel): every rank's partial reaches every rank, each sums them in RANK
        order.  Cross-checked against the library's own all_reduce (the RCCL leg: ncclAllReduce, its order)."""
        n_coll[0] += 1
        parts = [torch.zeros(part.size) for _ in range(world)]
        dist.all_gather(parts, torch.from_numpy(part.copy()))
        acc = parts[0].numpy().astype(np.float32)
        for p in parts[1:]:
            acc = acc + p.numpy()
        lib = torch.from_numpy(part.copy())
        dist.all_reduce(lib, op=dist.ReduceOp.SUM)
        np.testing.assert_allclose(lib.numpy(), acc, rtol=1e-5, atol=1e-6)
        return acc

    logits_all = []
    for pos, tok in enumerate(toks):
        x = W["token_embedding_table"][tok].copy()
        for l in range(cfg.n_layers):
            xb = orc.rmsnorm(x, W["rms_att_weight"][l])
            q = orc.matmul(xb, W["wq"][l][d0:d1])
            k = orc.matmul(xb, W["wk"][l][k0:k1])
            v = orc.matmul(xb, W["wv"][l][k0:k1])
            for i in range(0, d1 - d0, 2):  # RoPE on local rows; (i % hs) is shard independent
                freq = np.float32(1.0) / np.float32(np.power(np.float32(10000.0), np.float32(i % hs) / np.float32(hs), dtype=np.float32))
                val = np.float32(pos) * freq
                fcr, fci = np.cos(val, dtype=np.float32), np.sin(val, dtype=np.float32)
                q[i], q[i + 1] = q[i] * fcr - q[i + 1] * fci, q[i] * fci + q[i + 1] * fcr
                if i < k1 - k0:
                    k[i], k[i + 1] = k[i] * fcr - k[i + 1] * fci, k[i] * fci + k[i + 1] * fcr
            kc[l, pos], vc[l, pos] = k, v
            xb_full = np.zeros(cfg.dim, np.float32)
            for hl in range((d1 - d0) // hs):
                kh = (hl // kv_mul) * hs
                att = np.array([orc.vector_dot_product(q[hl * hs:(hl + 1) * hs], kc[l, t, kh:kh + hs])
                                / np.sqrt(np.float32(hs)) for t in range(pos + 1)], np.float32)
                att = orc.softmax(att)
                rows = np.ascontiguousarray(vc[l, :pos + 1].reshape(-1)[kh:])
                xb_full[d0 + hl * hs:d0 + (hl + 1) * hs] = orc.vector_weighted_sum_rows(
                    hs, rows, k1 - k0, att)
            if scheme == "B":
                # wo by columns: the local heads' outputs against this rank's columns of every row (main.zig:392);
                # rank 0's partial carries the residual (:395)
                part = orc.matmul(np.ascontiguousarray(xb_full[d0:d1]), np.ascontiguousarray(W["wo"][l][:, d0:d1]))
                x = allreduce(x + part if rank == 0 else part)
            else:
                xb_full = allgather(xb_full, d0, d1)
                x[d0:d1] = x[d0:d1] + orc.matmul(xb_full, W["wo"][l][d0:d1])
                x = allgather(x, d0, d1)
            xb = orc.rmsnorm(x, W["rms_ffn_weight"][l])
            a = orc.matmul(xb, W["w1"][l][h0:h1])
            b = orc.matmul(xb, W["w3"][l][h0:h1])
            hb = np.zeros(cfg.hidden_dim, np.float32)
            one = np.float32(1.0)
            hb[h0:h1] = (a * (one / (one + np.exp(-a, dtype=np.float32)))) * b
            if scheme == "B":   # w2 by columns (:419), residual on rank 0 (:422)
                part = orc.matmul(np.ascontiguousarray(hb[h0:h1]), np.ascontiguousarray(W["w2"][l][:, h0:h1]))
                x = allreduce(x + part if rank == 0 else part)
            else:
                hb = allgather(hb, h0, h1)
                x[d0:d1] = x[d0:d1] + orc.matmul(hb, W["w2"][l][d0:d1])
                x = allgather(x, d0, d1)
        xf = orc.rmsnorm(x, W["rms_final_weight"])
        lg = np.zeros(cfg.vocab_size, np.float32)
        lg[v0:v1] = orc.matmul(xf, W["wcls"][v0:v1])
        logits_all.append(allgather(lg, v0, v1))
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), np.stack(logits_all))
    np.save(os.path.join(out_dir, f"ncoll{rank}.npy"), np.array(n_coll))
    dist.barrier()
    dist.destroy_process_group()


def _toy(world):
    """8 ranks (BASELINE config 5's rank count) need 8 kv heads and dims that split eight ways."""
    if world == 8:
        return dict(dim=128, hidden_dim=176, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=512, seq_len=16)
    return dict(dim=64, hidden_dim=172, n_layers=2, n_heads=8, n_kv_heads=4, vocab_size=512, seq_len=16)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_schedule_is_bit_identical(world, tmp_path, ck, orc):
    import torch.multiprocessing as mp

    kw = _toy(world)
    shared, seed, toks = False, 77, [1, 40, 300, 7, 9]
    cfg = ck.Config(**kw)
    mp.spawn(_worker, args=(world, _free_port(), kw, shared, seed, toks, str(tmp_path)),
             nprocs=world, join=True)
    m = orc.Model(cfg.as_i32(), ck.synth_blob(cfg, shared, seed), shared)
    ref = np.stack([m.transformer(t, p) for p, t in enumerate(toks)])
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        # expf differs between numpy and libm by <= 1 ulp in the SwiGLU; everything else is
        # the same arithmetic, so compare tightly and require identical argmax
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
        assert got.argmax(1).tolist() == ref.argmax(1).tolist()
        assert np.array_equal(got, np.load(tmp_path / "rank0.npy"))  # all ranks agree bit for bit
    m.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_scheme_b_schedule(world, tmp_path, ck, orc):
    """Scheme B (L2Z_SCHEME_B; csrc/forward.cpp): Wo / W2 sharded by columns, 2 all-reduces per layer + the logits gather
    instead of 4 all-gathers + it.  The row sums are split across ranks, so the logits hold the parity tolerance against
    the oracle (not bit identity), the ranks agree with each other bit for bit, and a token costs 2L + 1 collectives."""
    import torch.multiprocessing as mp

    kw = _toy(world)
    shared, seed, toks = False, 78, [1, 41, 299, 8, 10]
    cfg = ck.Config(**kw)
    mp.spawn(_worker, args=(world, _free_port(), kw, shared, seed, toks, str(tmp_path), "B"), nprocs=world, join=True)
    m = orc.Model(cfg.as_i32(), ck.synth_blob(cfg, shared, seed), shared)
    ref = np.stack([m.transformer(t, p) for p, t in enumerate(toks)])
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        np.testing.assert_allclose(got, ref, rtol=5e-5, atol=5e-5)
        assert np.array_equal(got, np.load(tmp_path / "rank0.npy"))
        assert int(np.load(tmp_path / f"ncoll{r}.npy")[0]) == len(toks) * (2 * cfg.n_layers + 1)
    m.close()


def _prefill_worker(rank, world, port, kw, shared, seed, toks, out_dir, scheme="A"):
    """The row-sharded BATCHED prefill (csrc/prefill_host.cpp: prefill_stage / comm_bulk_allgather /
    bulk_unpack_kernel) for one chunk: every stage ends in a [P, n] matrix whose columns are split over
    the ranks; a rank's [P, n_loc] block sits contiguously at stage[rank], the blocks are all-gathered and
    unpacked into the row-major matrix the next stage reads."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = ge.load_package()
    ck, B = pkg.checkpoint, pkg.binding
    orc = ge.load_oracle()
    cfg = ck.Config(**kw)
    W = ck.carve(cfg, ck.synth_blob(cfg, shared, seed), shared)
    hs, kv_mul, S, P = cfg.head_size, cfg.kv_mul, cfg.seq_len, len(toks)
    d0, d1 = B.shard_range(cfg.dim, hs, rank, world)
    k0, k1 = B.shard_range(cfg.kv_dim, hs, rank, world)
    h0, h1 = B.shard_range(cfg.hidden_dim, 1, rank, world)
    v0, v1 = B.shard_range(cfg.vocab_size, 1, rank, world)
    kc = np.zeros((cfg.n_layers, S, k1 - k0), np.float32)
    vc = np.zeros((cfg.n_layers, S, k1 - k0), np.float32)

    def gather_unpack(block):
        """block: this rank's [P, n_loc] -> the [P, world * n_loc] matrix (stage[p] -> columns p * n_loc ..)."""
        n_loc = block.shape[1]
        stage = [torch.zeros(P * n_loc) for _ in range(world)]          # [world][P * n_loc], contiguous blocks
        dist.all_gather(stage, torch.from_numpy(np.ascontiguousarray(block).reshape(-1).copy()))
        out = np.empty((P, world * n_loc), np.float32)
        for p in range(world):                                           # bulk_unpack_kernel
            out[:, p * n_loc:(p + 1) * n_loc] = stage[p].numpy().reshape(P, n_loc)
        return out

    def bulk_allreduce(part):
        """Scheme B (comm.cpp comm_bulk_allreduce): part [P, n] -> the sum over ranks IN RANK ORDER, on every rank: every
        peer gets the columns of ITS slice (reduce-scatter; here an all_gather of whole partials, each rank reading its
        slice of every block), adds the world blocks in rank order, and the summed slices are all-gathered and unpacked."""
        n = part.shape[1]
        n_loc = n // world
        blocks = [torch.zeros(P * n) for _ in range(world)]
        dist.all_gather(blocks, torch.from_numpy(np.ascontiguousarray(part).reshape(-1).copy()))
        mine = None
        for p in range(world):                                           # bulk_reduce_kernel: rank order
            blk = blocks[p].numpy().reshape(P, n)[:, rank * n_loc:(rank + 1) * n_loc].astype(np.float32)
            mine = blk.copy() if mine is None else (mine + blk).astype(np.float32)
        return gather_unpack(mine)

    rows = lambda X, Wm: np.stack([orc.matmul(X[t], Wm) for t in range(P)])   # the oracle's dot products per token
    x = np.stack([W["token_embedding_table"][t] for t in toks]).astype(np.float32)
    for l in range(cfg.n_layers):
        xn = np.stack([orc.rmsnorm(x[t], W["rms_att_weight"][l]) for t in range(P)])
        q, k, v = rows(xn, W["wq"][l][d0:d1]), rows(xn, W["wk"][l][k0:k1]), rows(xn, W["wv"][l][k0:k1])
        for t in range(P):
            for i in range(0, d1 - d0, 2):
                freq = np.float32(1.0) / np.float32(np.power(np.float32(10000.0), np.float32(i % hs) / np.float32(hs), dtype=np.float32))
                val = np.float32(t) * freq
                fcr, fci = np.cos(val, dtype=np.float32), np.sin(val, dtype=np.float32)
                q[t, i], q[t, i + 1] = q[t, i] * fcr - q[t, i + 1] * fci, q[t, i] * fci + q[t, i + 1] * fcr
                if i < k1 - k0:
                    k[t, i], k[t, i + 1] = k[t, i] * fcr - k[t, i + 1] * fci, k[t, i] * fci + k[t, i + 1] * fcr
        kc[l, :P], vc[l, :P] = k, v
        att = np.zeros((P, d1 - d0), np.float32)
        for t in range(P):
            for hl in range((d1 - d0) // hs):
                kh = (hl // kv_mul) * hs
                a = np.array([orc.vector_dot_product(q[t, hl * hs:(hl + 1) * hs], kc[l, u, kh:kh + hs])
                              / np.sqrt(np.float32(hs)) for u in range(t + 1)], np.float32)
                a = orc.softmax(a)
                att[t, hl * hs:(hl + 1) * hs] = orc.vector_weighted_sum_rows(
                    hs, np.ascontiguousarray(vc[l, :t + 1].reshape(-1)[kh:]), k1 - k0, a)
        one = np.float32(1.0)
        if scheme == "B":
            # prefill_half_b: the local heads' output against this rank's COLUMNS of every row of Wo (:392), rank 0's
            # partial with the residual (:395); likewise the local hidden rows against W2's columns (:419-422)
            part = rows(att, np.ascontiguousarray(W["wo"][l][:, d0:d1]))
            x = bulk_allreduce((x + part).astype(np.float32) if rank == 0 else part)
            xn = np.stack([orc.rmsnorm(x[t], W["rms_ffn_weight"][l]) for t in range(P)])
            a, b = rows(xn, W["w1"][l][h0:h1]), rows(xn, W["w3"][l][h0:h1])
            hloc = ((a * (one / (one + np.exp(-a, dtype=np.float32)))) * b).astype(np.float32)
            part = rows(hloc, np.ascontiguousarray(W["w2"][l][:, h0:h1]))
            x = bulk_allreduce((x + part).astype(np.float32) if rank == 0 else part)
            continue
        att_full = gather_unpack(att)                                                    # PF_ATT
        x = gather_unpack(x[:, d0:d1] + rows(att_full, W["wo"][l][d0:d1]))               # PF_WO (res = x[:, d0:d1])
        xn = np.stack([orc.rmsnorm(x[t], W["rms_ffn_weight"][l]) for t in range(P)])
        a, b = rows(xn, W["w1"][l][h0:h1]), rows(xn, W["w3"][l][h0:h1])
        h1m = gather_unpack((a * (one / (one + np.exp(-a, dtype=np.float32)))) * b)      # PF_H1
        x = gather_unpack(x[:, d0:d1] + rows(h1m, W["w2"][l][d0:d1]))                    # PF_W2
    xf = orc.rmsnorm(x[P - 1], W["rms_final_weight"])
    lg = np.zeros(cfg.vocab_size, np.float32)
    lg[v0:v1] = orc.matmul(xf, W["wcls"][v0:v1])
    parts = [torch.zeros(v1 - v0) for _ in range(world)]
    dist.all_gather(parts, torch.from_numpy(lg[v0:v1].copy()))
    np.save(os.path.join(out_dir, f"pf_rank{rank}.npy"), np.concatenate([p.numpy() for p in parts]).astype(np.float32))
    np.save(os.path.join(out_dir, f"pf_k_rank{rank}.npy"), kc[:, :P])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_scheme_b_prefill_schedule(world, tmp_path, ck, orc):
    """Scheme B's batched prefill (prefill_host.cpp prefill_half_b + comm_bulk_allreduce) on gloo ranks: column shards of
    Wo / W2, partial [tokens, dim] products summed in rank order by the slice owners and re-gathered.  The last
    position's logits hold the parity tolerance against the oracle's stepped pass (the row sums are split across ranks),
    and every rank holds the same bits."""
    import torch.multiprocessing as mp

    kw = _toy(world)
    shared, seed, toks = False, 79, [1, 40, 300, 7, 9, 11, 500]
    cfg = ck.Config(**kw)
    mp.spawn(_prefill_worker, args=(world, _free_port(), kw, shared, seed, toks, str(tmp_path), "B"), nprocs=world, join=True)
    m = orc.Model(cfg.as_i32(), ck.synth_blob(cfg, shared, seed), shared)
    ref = None
    for p, t in enumerate(toks):
        ref = m.transformer(t, p)
    for r in range(world):
        got = np.load(tmp_path / f"pf_rank{r}.npy")
        np.testing.assert_allclose(got, ref, rtol=5e-5, atol=5e-5)
        assert np.array_equal(got, np.load(tmp_path / "pf_rank0.npy"))
    m.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_prefill_schedule(world, tmp_path, ck, orc):
    """The stage / staging-block / all-gather / unpack schedule of the row-sharded batched prefill on gloo
    ranks with the oracle's kernels as the per-rank math: the last position's logits equal the oracle's
    stepped pass (same dot products in the same order whichever rank owns a row), every rank holds the same
    logits, and the ranks' key-cache shards tile the unsharded cache."""
    import torch.multiprocessing as mp

    kw = _toy(world)
    shared, seed, toks = False, 78, [1, 40, 300, 7, 9, 11, 500]
    cfg = ck.Config(**kw)
    mp.spawn(_prefill_worker, args=(world, _free_port(), kw, shared, seed, toks, str(tmp_path)), nprocs=world, join=True)
    m = orc.Model(cfg.as_i32(), ck.synth_blob(cfg, shared, seed), shared)
    ref = None
    for p, t in enumerate(toks):
        ref = m.transformer(t, p)
    for r in range(world):
        got = np.load(tmp_path / f"pf_rank{r}.npy")
        np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-5)
        assert int(got.argmax()) == int(ref.argmax())
        assert np.array_equal(got, np.load(tmp_path / "pf_rank0.npy"))
    kvl = cfg.kv_dim // world
    full_k = np.concatenate([np.load(tmp_path / f"pf_k_rank{r}.npy") for r in range(world)], axis=2)
    assert full_k.shape == (cfg.n_layers, len(toks), cfg.kv_dim) and kvl * world == cfg.kv_dim
    assert np.isfinite(full_k).all() and float(np.abs(full_k).max()) > 0
    m.close()


def _xchg_worker(rank, world, port, vocab, seed, out_dir):
    """The greedy step's hand-over of a shard group (csrc/misc_kernels.hip argmax_kernel, ArgmaxArgs::xchg): every rank
    reduces ITS vocabulary rows to one (max, first index) candidate, the ranks exchange the N pairs (here: gloo
    all_gather; on the GPUs: two LL words per rank in every peer's landing slot) and every rank applies main.zig:720's
    rule to them -- larger value, equal values: lower index."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge

    dist.init_process_group("gloo", rank=rank, world_size=world)
    B = ge.load_package().binding
    v0, v1 = B.shard_range(vocab, 1, rank, world)
    rng = np.random.default_rng(seed)
    picks = []
    for case in range(24):
        lg = rng.standard_normal(vocab).astype(np.float32)
        if case % 3 != 2:  # plant the maximum at several indices, on different ranks and inside one rank
            hot = rng.choice(vocab, size=2 + case % 4, replace=False)
            lg[hot] = np.float32(lg.max() + 1.0 if case % 3 == 0 else lg.max())
        mine = lg[v0:v1]
        loc = int(np.argmax(mine))  # first index of the maximum: strict '>' (main.zig:720)
        pair = torch.tensor([float(mine[loc]), float(v0 + loc)], dtype=torch.float64)
        pairs = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(pairs, pair)
        best, bi = -np.inf, 0x7fffffff
        for p in pairs:  # any order gives the same winner: the rule is a total order on (value desc, index asc)
            v, i = np.float32(p[0].item()), int(p[1].item())
            if v > best or (v == best and i < bi):
                best, bi = v, i
        assert bi == int(np.argmax(lg)), (case, bi, int(np.argmax(lg)))
        picks.append(bi)
    np.save(os.path.join(out_dir, f"picks{rank}.npy"), np.array(picks))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_greedy_candidate_exchange_rule(world, tmp_path):
    """N pairs instead of 32000 logits: the winner equals the argmax of the whole vector, ties included, on every rank."""
    import torch.multiprocessing as mp

    mp.spawn(_xchg_worker, args=(world, _free_port(), 32000, 5, str(tmp_path)), nprocs=world, join=True)
    p0 = np.load(tmp_path / "picks0.npy")
    for r in range(1, world):
        assert np.array_equal(np.load(tmp_path / f"picks{r}.npy"), p0)
