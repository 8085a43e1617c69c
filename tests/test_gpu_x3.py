"""The prefill GEMMs on the bf16 matrix cores (csrc/prefill_common.h split3 / x3_mfma, prefill_gemm.hip planes kernels):
an f32 product as six bf16 products of three-term splits.  The claim these tests hold the kernels to: the result is an
f32-accurate product -- its error against FLOAT64 is not above the error of the f32 matrix cores' own fmaf chain
(v_mfma_f32_32x32x2_f32, the arithmetic of rounds 2-5, L2Z_PF_X3=0) on the same inputs."""
import numpy as np
import pytest



def _layer0_v_rows(ck, cfg, seed, toks):
    """float64 truth of the value-cache rows of layer 0: Wv . rmsnorm(embedding row) -- ONE GEMM of the prefill, inputs
    regenerated on the host (the rmsnorm in float64, then rounded to the f32 the device holds, up to its own rounding)."""
    t = {t.name: t for t in ck.tensor_table(cfg, False)}

    def tensor(name, row0, rows, width):
        return ck.synth_values(t[name].offset + row0 * width, rows * width, seed, t[name].scale, t[name].bias).reshape(rows, width)
    rms = tensor("rms_att_weight", 0, 1, cfg.dim)[0].astype(np.float64)
    wv = tensor("wv", 0, cfg.kv_dim, cfg.dim).astype(np.float64)
    emb = np.stack([tensor("token_embedding_table", tk, 1, cfg.dim)[0] for tk in toks]).astype(np.float64)
    xn = emb * (1.0 / np.sqrt((emb * emb).mean(axis=1, keepdims=True) + 1e-5)) * rms
    xn = xn.astype(np.float32).astype(np.float64)
    return xn @ wv.T, np.abs(xn) @ np.abs(wv.T)


# (n_tokens, what the bf16 path runs there)
FORMS = [(56, "stream form, two token tiles"), (100, "stream form, four token tiles"), (300, "tile forms")]


@pytest.mark.gpu
@pytest.mark.parametrize("n,what", FORMS, ids=[f[1].replace(" ", "-").replace(",", "") for f in FORMS])
def test_bf16_split_product_is_as_accurate_as_the_f32_chain(gpu, ck, options, n, what):
    """One GEMM (K = 2048, matrices that stream from HBM) on both kinds of matrix cores against float64.
    Bars: max |err| <= 4e-7 * sum |a_i b_i| (the single-kernel bar of tests/test_gpu_parity.py is 4e-6), the rms error
    of the bf16 path <= 1.15 x the f32 chain's, and no bias (|mean err * sign(value)| <= 3e-8 * mean |value|: a
    truncating split or accumulator would show up here first).  Observed at K = 4096 (scripts/x3_accuracy.py): rms
    7.1e-7 (tile forms) / 3.2e-7 (stream form: shorter chains, K ranges) against 8.1e-7 for the f32 chain."""
    cfg = ck.Config(dim=2048, hidden_dim=5632, n_layers=1, n_heads=16, n_kv_heads=16, vocab_size=4096, seq_len=320)
    seed = 11
    w = gpu.Weights(cfg, None, False, seed=seed)
    toks = [1] + np.random.default_rng(1).integers(2, cfg.vocab_size, n - 1).tolist()
    truth, scale = _layer0_v_rows(ck, cfg, seed, toks)
    S, kvd = cfg.seq_len, cfg.kv_dim
    rms = {}
    for x3 in (0, 1):
        options(L2Z_PF_X3=x3, L2Z_PF_PANEL=0 if x3 == 0 else 1)   # (f32 side: the tile GEMM's chain, not the panel kernel's ranges)
        s = gpu.RunState(cfg)
        s.prefill(toks, 0, w)
        got = s.read("value_cache", 0, S * kvd).astype(np.float64).reshape(S, kvd)[:n]
        s.close()
        err = got - truth
        rms[x3] = float(np.sqrt((err ** 2).mean()))
        bias = float((err * np.sign(truth)).mean() / np.abs(truth).mean())
        print(f"{what}, {n} tokens, L2Z_PF_X3={x3}: max |err| / sum|ab| {float((np.abs(err) / scale).max()):.3e}, rms err {rms[x3]:.3e}, bias {bias:+.2e}")
        assert float((np.abs(err) / scale).max()) <= 4e-7
        assert abs(bias) <= 3e-8
    assert rms[1] <= 1.15 * rms[0], rms
    w.close()


def test_split_terms_recompose_exactly():
    """split3 on the host's own floats, restated in numpy: x1 + x2 + x3 == x bit for bit (round-to-nearest-even bf16 of
    what the terms before left; the subtractions are exact), the property the six-product sum rests on.  This is the
    restatement the device code follows (prefill_common.h), checked on the values the checkpoints hold -- including
    denormal-free extremes."""
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.standard_normal(1 << 16).astype(np.float32) * np.float32(s) for s in (1e-3, 1.0, 37.5, 1e4)] +
                       [np.array([0.0, -0.0, 1.0, -1.0, 3.3895314e38 / 2, 1.1754944e-38 * 2 ** 30], np.float32)])

    def bf16(v):   # round to nearest even to 8 significand bits, as v_cvt_pk_bf16_f32 does
        u = v.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)
    x1 = bf16(x); r1 = x - x1
    x2 = bf16(r1); r2 = r1 - x2
    x3 = bf16(r2)
    assert np.array_equal((x1.astype(np.float64) + x2 + x3), x.astype(np.float64))
    assert np.array_equal(r2 - x3, np.zeros_like(x))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [56, 100, 128, 300])
def test_planes_written_by_the_producers_equal_the_split_launch(gpu, ck, options, n):
    """The unsharded pass writes an activation matrix's planes of bf16 terms in the kernel that PRODUCES the matrix
    (rmsnorm, the attention output, the SwiGLU epilogue of the stream form; prefill_host.cpp planes_for) instead of a split
    launch before every GEMM: an element's three terms depend on that element alone, so logits, the KV cache and the
    activations must come out BIT-identical with L2Z_PF_FUSE_PLANES=0 (every consumer splits for itself).  Stream form at
    two / four token tiles and the tile forms; a model wide enough that every product takes the bf16 cores."""
    cfg = ck.Config(dim=2048, hidden_dim=5632, n_layers=2, n_heads=16, n_kv_heads=16, vocab_size=4096, seq_len=320)
    w = gpu.Weights(cfg, None, False, seed=5)
    toks = [1] + np.random.default_rng(2).integers(2, cfg.vocab_size, n - 1).tolist()
    got = {}
    for fuse in (0, 1):
        options(L2Z_PF_FUSE_PLANES=fuse)
        s = gpu.RunState(cfg)
        s.prefill(toks, 0, w)
        S, kvd, L = cfg.seq_len, cfg.kv_dim, cfg.n_layers - 1   # (the last layer's rows: every launch of every layer before feeds them)
        got[fuse] = (s.logits().copy(), s.read("key_cache", L * S * kvd, n * kvd), s.read("value_cache", L * S * kvd, n * kvd))
    for a, b, what in zip(got[0], got[1], ("logits", "key cache", "value cache")):
        assert np.array_equal(a, b), f"{n} tokens: {what} differ between fused and split planes (max |diff| {np.abs(a - b).max():.3e})"


@pytest.mark.gpu
@pytest.mark.parametrize("n", [40, 64])
def test_stream_form_tiles_of_every_width_equal_the_stepped_loop(gpu, ck, options, n):
    """Chunks of <= 64 tokens pick their tile by grid fill (prefill_gemm.hip launch_x3_stream: the narrowest of 128 / 192 / 256
    features -- eight / twelve / sixteen waves -- whose blocks fit one round of 256): a shape whose four products take all
    three.  dim 2048, hidden_dim 14336: q | k | v (6144 features, 8 K ranges) 32 tiles of 192 -- tiles lie across the three
    matrices; wo (16 MB: cache resident) stays on the f32 cores; W1 | W3 (28672 features, 2 ranges: 150 tiles of 192 would be
    300 blocks) 112 tiles of 256 on SIXTEEN waves; W2 (K = 14336, 8 ranges) 16 tiles of 128 (rocprofv3 of this test:
    prefill_x3_stream<7, 2, 3, 8>, <1, 2, 5, 4>, <6, 2, 4, 6>).  The K ranges -- the arithmetic -- do not depend on the tile.  Logits and
    KV rows against the stepped loop of f32 mat-vec kernels (itself pinned against the oracle), at the bar of
    test_7b_prefill_equals_stepped_loop."""
    cfg = ck.Config(dim=2048, hidden_dim=14336, n_layers=2, n_heads=16, n_kv_heads=16, vocab_size=4096, seq_len=96)
    w = gpu.Weights(cfg, None, False, seed=17)
    toks = [1] + np.random.default_rng(n).integers(2, cfg.vocab_size, n - 1).tolist()
    s = gpu.RunState(cfg)
    for pos, t in enumerate(toks):
        s.transformer(t, pos, w)
    ref = s.logits()
    S, kvd = cfg.seq_len, cfg.kv_dim
    ref_kv = {(nm, l): s.read(nm, l * S * kvd, n * kvd) for nm in ("key_cache", "value_cache") for l in range(cfg.n_layers)}
    s2 = gpu.RunState(cfg)
    s2.prefill(toks, 0, w)
    got = s2.logits()
    print(f"stream form, tiles of 128 / 192 / 256 features, {n} tokens: max |logit diff| {float(np.abs(got - ref).max()):.3e}")
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)
    for (nm, l), a in ref_kv.items():
        np.testing.assert_allclose(s2.read(nm, l * S * kvd, n * kvd), a, rtol=5e-5, atol=5e-5, err_msg=f"{nm} layer {l}")
    s.close(); s2.close(); w.close()
