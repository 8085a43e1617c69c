"""GPU parity: the HIP path (through the C ABI) against the CPU oracle.

Bars (BASELINE.json north_star):
  * token ids at temperature 0: IDENTICAL to the oracle's greedy loop;
  * logits: within the fp32 tolerance below.

The tolerance.  The reference's own arithmetic is only defined up to (a) the
host's DEFAULT_VECTOR_WIDTH (main.zig:7), (b) LLVM's FMA contraction and
(c) the @reduce order under @setFloatMode(.optimized) (main.zig:11-13).  The
oracle can emulate all 12 combinations; LOGIT_RTOL/ATOL below bound
|gpu - oracle(default mode)| and `test_gpu_inside_reference_spread` checks
that the GPU is no further from the default reading than the readings are
from each other (times a small factor).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# |gpu - oracle| <= LOGIT_ATOL + LOGIT_RTOL * |oracle|, logits are O(1..10)
LOGIT_RTOL = 5e-5   # observed on MI355X: max |diff| 3e-6 on toy models, 1.05e-5 on the 7B shape
LOGIT_ATOL = 5e-5
# single kernels (one dot product deep): relative to sum |a_i b_i|
KERNEL_RTOL = 4e-6

TOY = dict(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=32)


def rel_dot_err(got, ref, absdot):
    return float(np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64)) / (absdot + 1e-30)))


# ---------------------------------------------------------------- reference KATs
def test_kat_matrix_multiplies(gpu):
    """src/main.zig:1078-1087, exact"""
    w = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    x = np.array([1, 2, 3], np.float32)
    assert gpu.matmul(x, w).tolist() == [14.0, 32.0, 50.0]


def test_kat_vector_length_less_than_width(gpu):
    """src/main.zig:1089-1103, exact (small integers)"""
    w = np.arange(1, 25, dtype=np.float32).reshape(2, 12)
    x = np.arange(1, 13, dtype=np.float32)
    exp = [float(sum(w[i, j] * x[j] for j in range(12))) for i in range(2)]
    assert gpu.matmul(x, w).tolist() == exp


@pytest.mark.parametrize("vw", [4, 8, 16])
def test_kat_vector_weighted_sum_rows(gpu, vw):
    """src/main.zig:1117-1139: width = VW+3, stride = width+2, abs tol 1e-5"""
    width, weights = vw + 3, np.array([0.25, -0.5, 1.5], np.float32)
    stride = width + 2
    rows = np.zeros(stride * 3, np.float32)
    for r in range(3):
        for i in range(width):
            rows[r * stride + i] = r * width + i + 1
    out = gpu.vector_weighted_sum_rows(width, rows, stride, weights)
    for i in range(width):
        exp = sum(float(rows[r * stride + i]) * float(weights[r]) for r in range(3))
        assert abs(out[i] - exp) <= 1e-5


def test_kat_softmax_sums_to_one(gpu, orc):
    """src/main.zig:1141-1150.  The reference asserts sum == 1.0 exactly; that is a
    property of its strictly sequential `sum += exp(..)` (main.zig:698-701), which the
    oracle reproduces (asserted here).  The GPU reduces the denominator with a wave
    xor-shuffle tree, so its four quotients may each differ by 1 ulp: the stated
    tolerance is |sum - 1| <= 2^-23 (1 ulp of 1.0) and 2 ulp per element."""
    x = np.array([1, 2, 3, 4], np.float32)
    ref = orc.softmax(x)
    acc = np.float32(0)
    for v in ref:
        acc = np.float32(acc + v)
    assert acc == np.float32(1.0)          # the reference's own assertion, on the oracle
    s = gpu.softmax(x)
    acc = np.float32(0)
    for v in s:
        acc = np.float32(acc + v)
    assert abs(float(acc) - 1.0) <= 2.0 ** -23
    assert np.all(np.abs(s - ref) <= 2 * np.spacing(ref))


# ---------------------------------------------------------------- kernels vs oracle
@pytest.mark.parametrize("d,n", [(3, 3), (2, 12), (7, 5), (64, 64), (33, 172), (288, 288),
                                 (768, 288), (288, 768), (130, 4096), (16, 11008), (5, 1027)])
def test_matmul_vs_oracle(gpu, orc, d, n):
    rng = np.random.default_rng(d * 100003 + n)
    w = rng.standard_normal((d, n), dtype=np.float32)
    x = rng.standard_normal(n, dtype=np.float32)
    got, ref = gpu.matmul(x, w), orc.matmul(x, w)
    absdot = np.abs(w.astype(np.float64)) @ np.abs(x.astype(np.float64))
    assert rel_dot_err(got, ref, absdot) <= KERNEL_RTOL


@pytest.mark.parametrize("N", [2, 3])
def test_matmul_fused_vs_oracle(gpu, orc, N):
    rng = np.random.default_rng(N)
    d, n = 96, 320
    ws = [rng.standard_normal((d, n), dtype=np.float32) for _ in range(N)]
    x = rng.standard_normal(n, dtype=np.float32)
    got, ref = gpu.matmul_fused(x, ws), orc.matmul_fused(x, ws)
    for j in range(N):
        absdot = np.abs(ws[j].astype(np.float64)) @ np.abs(x.astype(np.float64))
        assert rel_dot_err(got[j], ref[j], absdot) <= KERNEL_RTOL
    # fusing must not change a row's value: same order as the unfused launch
    for j in range(N):
        assert np.array_equal(got[j], gpu.matmul(x, ws[j]))


@pytest.mark.parametrize("n", [1, 3, 64, 288, 300, 4096])
def test_rmsnorm_vs_oracle(gpu, orc, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n, dtype=np.float32) * 3
    w = 1 + 0.1 * rng.standard_normal(n, dtype=np.float32)
    np.testing.assert_allclose(gpu.rmsnorm(x, w), orc.rmsnorm(x, w), rtol=3e-6, atol=1e-7)


@pytest.mark.parametrize("n", [1, 2, 5, 64, 257, 2048, 32000])
def test_softmax_vs_oracle(gpu, orc, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n, dtype=np.float32) * 4
    got, ref = gpu.softmax(x), orc.softmax(x)
    # the denominator is an n-term f32 sum taken in a different order: allow
    # sqrt(n)*eps relative on top of 1 ulp of exp()
    np.testing.assert_allclose(got, ref, rtol=2e-6 + 6e-8 * 4 * np.sqrt(n), atol=1e-12)
    assert abs(float(got.astype(np.float64).sum()) - 1.0) < 1e-5


@pytest.mark.parametrize("n", [1, 3, 48, 64, 128, 130])
def test_dot_vs_oracle(gpu, orc, n):
    rng = np.random.default_rng(n)
    x, y = rng.standard_normal(n, dtype=np.float32), rng.standard_normal(n, dtype=np.float32)
    absdot = float(np.abs(x.astype(np.float64)) @ np.abs(y.astype(np.float64)))
    assert abs(float(gpu.vector_dot_product(x, y)) - float(orc.vector_dot_product(x, y))) <= KERNEL_RTOL * absdot


@pytest.mark.parametrize("hs,stride,T", [(48, 288, 1), (48, 288, 77), (64, 768, 300), (128, 4096, 200), (11, 13, 9)])
def test_weighted_sum_rows_vs_oracle(gpu, orc, hs, stride, T):
    rng = np.random.default_rng(hs + T)
    rows = rng.standard_normal((T - 1) * stride + hs, dtype=np.float32)
    wts = rng.random(T, dtype=np.float32)
    got = gpu.vector_weighted_sum_rows(hs, rows, stride, wts)
    ref = orc.vector_weighted_sum_rows(hs, rows, stride, wts)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.sqrt(T))


def attention_ref(orc, q, kc, vc, pos, n_heads, n_kv_heads, hs):
    """src/main.zig:361-389 for one layer out of the oracle's own kernels (:503 dot, :687 softmax,
    :657 weighted row sum)."""
    kv_dim, kv_mul = n_kv_heads * hs, n_heads // n_kv_heads
    K, V = kc.reshape(-1, kv_dim), vc.reshape(-1)
    out = np.empty(n_heads * hs, np.float32)
    div = np.float32(np.sqrt(np.float32(hs)))
    for h in range(n_heads):
        o = (h // kv_mul) * hs                                               # :369
        att = np.array([orc.vector_dot_product(q[h * hs:(h + 1) * hs], K[t, o:o + hs]) for t in range(pos + 1)],
                       np.float32) / div                                    # :372
        att = orc.softmax(att)                                              # :378
        out[h * hs:(h + 1) * hs] = orc.vector_weighted_sum_rows(hs, V[o:], kv_dim, att)   # :381-388
    return out


ATTN_SHAPES = {  # n_heads, n_kv_heads, head_size, seq_len
    "stories15M": (6, 6, 48, 256), "stories110M": (12, 12, 64, 1024), "llama2-7b": (32, 32, 128, 2048),
    "gqa": (8, 2, 32, 64), "mqa-wide-head": (4, 1, 256, 96),
}
# (shape, form, nch, positions): every kernel the forward pass can pick for these models, driven
# directly -- attention_fast_kernel<256, speculative>, <1024>, the single-launch split form -- plus
# the generic kernel; positions cover pos 0, fewer timesteps than groups / chunks, the last row
ATTN_CASES = [
    ("stories15M", "fast256", 0, (0, 1, 31, 255)),
    ("stories15M", "generic", 0, (0, 255)),
    ("stories15M", "split", 3, (0, 1, 2, 255)),
    ("stories110M", "fast1024", 0, (0, 5, 255, 1023)),
    ("stories110M", "fast256", 0, (300,)),
    ("stories110M", "split", 0, (256, 1023)),
    ("llama2-7b", "fast1024", 0, (0, 100, 255)),
    ("llama2-7b", "split", 0, (0, 3, 300, 1023, 1024, 2047)),   # 256 threads per block below pos 1024, 1024 from there
    ("llama2-7b", "split", 16, (2047,)),
    ("gqa", "fast256", 0, (0, 63)),
    ("gqa", "split", 2, (0, 1, 63)),
    ("gqa", "generic", 0, (63,)),
    ("mqa-wide-head", "fast256", 0, (95,)),
    ("mqa-wide-head", "split", 4, (2, 95)),
]


@pytest.mark.parametrize("shape,form,nch,positions", ATTN_CASES,
                         ids=[f"{c[0]}-{c[1]}{c[2] or ''}" for c in ATTN_CASES])
def test_attention_kernels_vs_oracle(gpu, orc, shape, form, nch, positions):
    """The decode attention kernels the forward pass launches, one layer at a time, against the
    oracle's dot / softmax / weighted-row-sum (main.zig:361-389).  Outputs are convex combinations
    of V entries (|v| <= 2 here); observed max |diff| 4e-7 on MI355X (profiles/r02_parity_numbers.txt), bound 2e-6."""
    H, KV, hs, S = ATTN_SHAPES[shape]
    rng = np.random.default_rng(H * 1000 + hs)
    q = rng.standard_normal(H * hs, dtype=np.float32)
    kc = rng.standard_normal(S * KV * hs, dtype=np.float32)
    vc = rng.uniform(-2, 2, S * KV * hs).astype(np.float32)
    worst = 0.0
    for pos in positions:
        got = gpu.attention_decode(q, kc, vc, pos, H, KV, hs, S, form=form, nch=nch)
        ref = attention_ref(orc, q, kc, vc, pos, H, KV, hs)
        worst = max(worst, float(np.abs(got - ref).max()))
        np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6, err_msg=f"{shape} {form} pos {pos}")
    print(f"attention {shape} {form}{nch or ''}: max |diff| {worst:.2e}")


PF_ATTN_FORMS = {"per-query": 1, "tiled": 2, "flash": 3}


@pytest.mark.parametrize("H,KV,hs", [(4, 2, 64), (4, 4, 128), (6, 6, 48)], ids=["gqa-hs64", "mha-hs128", "hs48"])
@pytest.mark.parametrize("form", list(PF_ATTN_FORMS))
def test_prefill_attention_kernels_vs_softmax_reference(gpu, orc, H, KV, hs, form):
    """The batched prefill's attention kernels driven directly (l2z_prefill_attention): block per (head,
    query); tiled with the softmax in LDS; the flash form (S^T / O^T on MFMA 16x16x4, P kept in the
    accumulators' registers; two key parts per tile).  150 queries at positions 37 .. 186 (three query
    tiles, the last one partial, key tiles that start before the chunk) against main.zig:361-389 in
    float64, and a few (query, head) pairs against the oracle's own dot / softmax / weighted sum.  Outputs
    are convex combinations of V entries (|v| <= 2): bound 3e-6."""
    if form.startswith("flash") and hs not in (64, 128):
        pytest.skip("the flash form takes head sizes 64 and 128")
    pos0, P, S = 37, 150, 200
    dim, kvd, kv_mul = H * hs, KV * hs, H // KV
    rng = np.random.default_rng(hs + H)
    q = rng.standard_normal((P, dim), dtype=np.float32)
    kc = rng.standard_normal((S, kvd), dtype=np.float32)
    vc = rng.uniform(-2, 2, (S, kvd)).astype(np.float32)
    got = gpu.prefill_attention(PF_ATTN_FORMS[form], q, kc, vc, pos0, H, KV, hs)
    ref = np.empty((P, dim), np.float64)
    for h in range(H):
        o = (h // kv_mul) * hs
        K, V = kc[:, o:o + hs].astype(np.float64), vc[:, o:o + hs].astype(np.float64)
        sc = q[:, h * hs:(h + 1) * hs].astype(np.float64) @ K.T / np.sqrt(np.float64(hs))
        for i in range(P):
            a = sc[i, :pos0 + i + 1]
            e = np.exp(a - a.max())
            ref[i, h * hs:(h + 1) * hs] = (e / e.sum()) @ V[:pos0 + i + 1]
    worst = float(np.abs(got - ref).max())
    print(f"prefill attention {form} H{H} kv{KV} hs{hs}: max |diff| vs float64 {worst:.2e}")
    assert worst <= 3e-6
    for i, h in ((0, 0), (63, 1), (64, H - 1), (P - 1, H - 1)):   # the oracle's kernels on a few rows
        r = attention_ref(orc, q[i], kc.reshape(-1), vc.reshape(-1), pos0 + i, H, KV, hs)
        np.testing.assert_allclose(got[i, h * hs:(h + 1) * hs], r[h * hs:(h + 1) * hs], rtol=0, atol=3e-6)


def test_attention_auto_picks_the_documented_form(gpu):
    """form 'auto' is what enqueue_forward launches at a position (csrc/forward.cpp attn_variant): 256
    threads per head with the speculative first round while the context is short -- every position of a
    seq_len <= 512 model, pos < 256 at head sizes <= 64, pos < 128 at head size 128 (attention_short_pos)
    -- then 1024 threads per head, and the split form from pos 256 on (256 threads per block below pos 1024,
    1024 from there: the hook's 'split' takes the block size of the position too); bit-identical to the same
    form asked for by name."""
    rng = np.random.default_rng(5)
    for shape, pos, want, nch in (("stories15M", 200, "fast256", 0), ("stories110M", 200, "fast256", 0),
                                  ("llama2-7b", 100, "fast256", 0), ("llama2-7b", 200, "fast1024", 0),
                                  ("stories110M", 700, "split", 0), ("llama2-7b", 2047, "split", 0)):
        H, KV, hs, S = ATTN_SHAPES[shape]
        q = rng.standard_normal(H * hs, dtype=np.float32)
        kc = rng.standard_normal(S * KV * hs, dtype=np.float32)
        vc = rng.standard_normal(S * KV * hs, dtype=np.float32)
        a = gpu.attention_decode(q, kc, vc, pos, H, KV, hs, S, form="auto")
        b = gpu.attention_decode(q, kc, vc, pos, H, KV, hs, S, form=want, nch=nch)
        assert np.array_equal(a, b), (shape, pos, want)


def test_probs_read_is_softmax_of_logits_over_temperature(gpu, ck, orc):
    """l2z_probs_read (main.zig:1005-1008 on the device) against the oracle's softmax of the same logits
    divided by the temperature on the host, and against the exact (float64) softmax: the device sums in a
    tree, the host in order; expf differs in the last bit."""
    cfg = ck.Config(dim=128, hidden_dim=352, n_layers=2, n_heads=8, n_kv_heads=4, vocab_size=32000, seq_len=16)
    w, s = gpu.Weights(cfg, None, False, seed=9), gpu.RunState(cfg)
    s.transformer(5, 0, w)
    lg = s.logits()
    for temp in (1.0, 0.7, 0.05):
        got = s.probs(temp)
        x = (lg / np.float32(temp)).astype(np.float32)
        e = np.exp(x.astype(np.float64) - float(x.max()))
        assert abs(float(got.astype(np.float64).sum()) - 1.0) < 1e-5
        np.testing.assert_allclose(got, e / e.sum(), rtol=3e-6, atol=1e-12)   # the exact softmax
        # the oracle adds its 32000 terms in order in float32: its normalisation alone is off by ~2e-5
        np.testing.assert_allclose(got, orc.softmax(x), rtol=1e-4, atol=1e-9)
        assert int(np.argmax(got)) == int(np.argmax(lg))
    with pytest.raises(gpu.L2ZError):
        s.probs(0.0)
    s.close(); w.close()


def test_argmax_tie_rule(gpu, orc):
    """main.zig:720 strict '>' : the lowest index wins ties"""
    x = np.zeros(32000, np.float32)
    x[[31999, 777, 20000]] = 5.0
    assert gpu.argmax(x) == 777 == orc.argmax(x)
    assert gpu.argmax(np.full(1000, -np.inf, np.float32)) == 0
    rng = np.random.default_rng(3)
    for n in (1, 2, 63, 64, 65, 1023, 1025, 32000):
        y = rng.standard_normal(n, dtype=np.float32)
        assert gpu.argmax(y) == orc.argmax(y)


# ---------------------------------------------------------------- synthetic generator
def test_synth_generator_device_matches_host(gpu, ck, orc):
    cfg = ck.Config(**TOY)
    for shared in (True, False):
        w = gpu.Weights(cfg, None, shared, seed=99)
        n = ck.weights_count(cfg, shared)
        dev = w.read(0, n)
        assert np.array_equal(dev, ck.synth_blob(cfg, shared, 99))
        assert np.array_equal(dev, orc.synth_fill(cfg.as_i32(), shared, 99))
        w.close()


def test_upload_is_byte_identical(gpu, ck):
    cfg = ck.Config(**TOY)
    blob = ck.synth_blob(cfg, True, 5)
    w = gpu.Weights(cfg, blob, True)
    assert np.array_equal(w.read(0, blob.size), blob)
    w.close()


def test_weights_read_serves_file_order_across_the_interleaved_slot(gpu, ck):
    """The device copy keeps W1 | W3 row-interleaved in one slot; l2z_weights_read addresses the
    FILE's order: ranges that start and end inside rows of w1 / w3, run from w1 into w2 and from w2 into w3 and
    past it, for an uploaded blob and for the device-side generator."""
    cfg = ck.Config(dim=48, hidden_dim=136, n_layers=3, n_heads=4, n_kv_heads=2, vocab_size=300, seq_len=24)
    blob = ck.synth_blob(cfg, False, 11)
    t = {d.name: d for d in ck.tensor_table(cfg, False)}
    dim, hid = cfg.dim, cfg.hidden_dim
    w1, w2, w3 = t["w1"].offset, t["w2"].offset, t["w3"].offset
    ranges = [(w1 + 5, 7), (w1 + 5, 3 * dim), (w1 + dim * hid - 10, 30 + dim), (w2 - 17, 40),
              (w3 - 9, 2 * dim + 20), (w3 + (2 * hid + 3) * dim + 11, 5 * dim), (w3 + 3 * hid * dim - 4, 40), (w1, w3 + 3 * hid * dim - w1)]
    for w in (gpu.Weights(cfg, blob, False), gpu.Weights(cfg, None, False, seed=11)):
        for off, n in ranges:
            assert np.array_equal(w.read(off, n), blob[off:off + n]), (off, n)
        w.close()


# ---------------------------------------------------------------- an implementation that is not ours
HF_CONFIGS = [
    ("gqa-toy", dict(TOY), False, 32),
    ("stories15M-shape", dict(dim=288, hidden_dim=768, n_layers=6, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=256), True, 64),
    ("gqa-head-128", dict(dim=512, hidden_dim=1408, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=1024, seq_len=320), False, 300),
]


@pytest.mark.parametrize("name,kw,shared,n_pos", HF_CONFIGS, ids=[c[0] for c in HF_CONFIGS])
def test_gpu_logits_agree_with_hf_llama(gpu, ck, name, kw, shared, n_pos):
    """The HIP pass against Hugging Face's LlamaForCausalLM (CPU, fp32) holding the same seeded checkpoint -- an
    implementation of the reference's architecture that shares no code or author with the oracle
    (tests/test_oracle_vs_hf.py checks the oracle against it): logits at every position, stepped on the GPU (its KV
    cache, the attention forms by position up to the split form at pos >= 256), one causal pass in HF."""
    import importlib.util
    if importlib.util.find_spec("transformers") is None:
        pytest.skip("no transformers")
    import hf_llama
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=4243)
    rng = np.random.default_rng(4)
    toks = [1] + rng.integers(0, cfg.vocab_size, n_pos - 1).tolist()
    hf = hf_llama.logits_in_subprocess(kw, shared, 4243, toks)   # torch stays out of this process
    w, s = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
    worst = 0.0
    for pos, tok in enumerate(toks):
        s.transformer(tok, pos, w)
        got = s.logits()
        np.testing.assert_allclose(got, hf[pos], rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{name} pos {pos}")
        worst = max(worst, float(np.abs(got - hf[pos]).max()))
    print(f"GPU vs HF LlamaForCausalLM {name}: max |logit diff| over {n_pos} positions {worst:.2e}")
    s.close(); w.close()


def test_gpu_prefill_agrees_with_hf_llama(gpu, ck):
    """The batched prompt pass (MFMA GEMMs, prefill attention) against the same HF model: its logits after a 300-token
    prompt and the KV cache it leaves -- checked through the next stepped token -- against HF's causal pass."""
    import importlib.util
    if importlib.util.find_spec("transformers") is None:
        pytest.skip("no transformers")
    import hf_llama
    name, kw, shared, n_pos = HF_CONFIGS[2]
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=4243)
    toks = [1] + np.random.default_rng(4).integers(0, cfg.vocab_size, n_pos).tolist()
    hf = hf_llama.logits_in_subprocess(kw, shared, 4243, toks)
    w, s = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
    s.prefill(toks[:n_pos], 0, w)
    np.testing.assert_allclose(s.logits(), hf[n_pos - 1], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    s.transformer(toks[n_pos], n_pos, w)   # reads every KV row the batched pass wrote
    np.testing.assert_allclose(s.logits(), hf[n_pos], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    print(f"GPU prefill vs HF: max |logit diff| {np.abs(s.logits() - hf[n_pos]).max():.2e}")
    s.close(); w.close()


def test_gpu_panel_prefill_agrees_with_hf_llama(gpu, ck):
    """The K-range panel kernel (csrc/prefill_panel.hip: chunks of 17 ... 64 tokens of matrices that stream) against the
    HF model as well: a wide two-layer GQA shape, prompts of 20, 33 and 64 tokens (two, three and four token tiles), the
    logits after each prompt and -- through the next stepped token -- the KV rows it left."""
    import importlib.util
    if importlib.util.find_spec("transformers") is None:
        pytest.skip("no transformers")
    import hf_llama
    kw = dict(dim=3072, hidden_dim=8448, n_layers=2, n_heads=24, n_kv_heads=8, vocab_size=2048, seq_len=80)
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, False, seed=4244)
    toks = [1] + np.random.default_rng(6).integers(0, cfg.vocab_size, 65).tolist()
    hf = hf_llama.logits_in_subprocess(kw, False, 4244, toks)
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    worst = 0.0
    for n in (20, 33, 64):
        s.prefill(toks[:n], 0, w)
        np.testing.assert_allclose(s.logits(), hf[n - 1], rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{n} tokens")
        worst = max(worst, float(np.abs(s.logits() - hf[n - 1]).max()))
        s.transformer(toks[n], n, w)   # reads every KV row the batched pass wrote
        np.testing.assert_allclose(s.logits(), hf[n], rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"step after {n} tokens")
    print(f"GPU panel prefill vs HF: max |logit diff| {worst:.2e}")
    s.close(); w.close()


# ---------------------------------------------------------------- whole forward pass
CONFIGS = [
    ("toy-gqa-unshared", dict(TOY), False),
    ("toy-mha-shared", dict(dim=48, hidden_dim=128, n_layers=3, n_heads=4, n_kv_heads=4, vocab_size=300, seq_len=24), True),
    ("toy-mqa", dict(dim=96, hidden_dim=256, n_layers=2, n_heads=6, n_kv_heads=1, vocab_size=1000, seq_len=40), True),
    ("odd-headsize-6", dict(dim=36, hidden_dim=100, n_layers=2, n_heads=6, n_kv_heads=3, vocab_size=97, seq_len=16), False),
    ("stories15M-shape-2layers", dict(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=64), True),
]


@pytest.mark.parametrize("name,kw,shared", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_transformer_logits_and_state(gpu, ck, orc, name, kw, shared):
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=11)
    w, s = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
    m = orc.Model(cfg.as_i32(), blob, shared)
    rng = np.random.default_rng(7)
    toks = [1] + rng.integers(0, cfg.vocab_size, size=min(cfg.seq_len, 12) - 1).tolist()
    for pos, tok in enumerate(toks):
        ref = m.transformer(tok, pos)
        s.transformer(tok, pos, w)
        got = s.logits()
        np.testing.assert_allclose(got, ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL,
                                   err_msg=f"{name} pos {pos}")
        assert s.argmax() == orc.argmax(got)
    # KV cache rows written so far (main.zig:354-358) and the last q
    kvd, S = cfg.kv_dim, cfg.seq_len
    n_pos = len(toks)
    for l in range(cfg.n_layers):
        k_ref = m.state("key_cache", cfg.n_layers * S * kvd)[l * S * kvd:(l * S + n_pos) * kvd]
        v_ref = m.state("value_cache", cfg.n_layers * S * kvd)[l * S * kvd:(l * S + n_pos) * kvd]
        # the bound the prefill tests hold their KV rows to (round 3: 2e-4)
        np.testing.assert_allclose(s.read("key_cache", l * S * kvd, n_pos * kvd), k_ref, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(s.read("value_cache", l * S * kvd, n_pos * kvd), v_ref, rtol=2e-5, atol=2e-5)
    s.close(); w.close(); m.close()


@pytest.mark.parametrize("name,kw,shared", CONFIGS, ids=[c[0] for c in CONFIGS])
def test_greedy_token_ids_identical(gpu, ck, orc, name, kw, shared):
    """-t 0: token ids must be identical; report the tie margin when they are not."""
    cfg = ck.Config(**kw)
    for seed, prompt in ((21, []), (22, [5, 9, 2])):
        blob = ck.synth_blob(cfg, shared, seed=seed)
        w, s = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
        m = orc.Model(cfg.as_i32(), blob, shared)
        ref, margins = m.generate_greedy(prompt, cfg.seq_len)
        s.greedy_begin(prompt)
        got = np.concatenate([s.greedy_run(w, 5), s.greedy_run(w, cfg.seq_len)])
        if not np.array_equal(got, ref):
            k = int(np.argmax(got[:min(len(got), len(ref))] != ref[:min(len(got), len(ref))]))
            pytest.fail(f"{name}: first divergence at pos {k}: gpu {got[k]} vs oracle {ref[k]}, "
                        f"oracle top1-top2 margin there {margins[k]:.3e} (near-tie if ~1e-6)")
        # a second sequence on the same runstate must restart cleanly
        s.greedy_begin(prompt)
        assert np.array_equal(s.greedy_run(w, cfg.seq_len), ref)
        s.close(); w.close(); m.close()


def test_transformer_vs_greedy_paths_agree(gpu, ck):
    """l2z_transformer + l2z_argmax (host loop) == l2z_greedy_run (device loop), bit for bit."""
    cfg = ck.Config(**TOY)
    blob = ck.synth_blob(cfg, False, 31)
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    s.greedy_begin([])
    dev = s.greedy_run(w, cfg.seq_len)
    tok, host = 1, []
    for pos in range(len(dev)):
        s.transformer(tok, pos, w)
        tok = s.argmax()
        host.append(tok)
    assert host == dev.tolist()
    s.close(); w.close()


def test_gpu_inside_reference_spread(gpu, ck, orc):
    """The GPU's distance to the default oracle reading is of the same order as the
    distance between the 12 readings of the reference among themselves."""
    cfg = ck.Config(**CONFIGS[4][1])
    blob = ck.synth_blob(cfg, True, 41)
    w, s = gpu.Weights(cfg, blob, True), gpu.RunState(cfg)
    toks = [1, 100, 2000, 31999, 5, 6, 7, 8]
    runs = {}
    for mode in orc.ALL_MODES:
        orc.set_mode(*mode)
        m = orc.Model(cfg.as_i32(), blob, True)
        runs[mode] = np.stack([m.transformer(t, p) for p, t in enumerate(toks)])
        m.close()
    orc.set_mode(8, False, False)
    base = runs[(8, 0, 0)]
    spread = max(float(np.abs(r - base).max()) for r in runs.values())
    got = []
    for p, t in enumerate(toks):
        s.transformer(t, p, w)
        got.append(s.logits())
    gpu_err = float(np.abs(np.stack(got) - base).max())
    print(f"reference-reading spread {spread:.3e}, gpu distance {gpu_err:.3e}")
    assert gpu_err <= 4 * spread + 1e-6
    s.close(); w.close()


def test_errors_are_loud(gpu, ck):
    cfg = ck.Config(**TOY)
    blob = ck.synth_blob(cfg, False, 1)
    with pytest.raises(gpu.L2ZError):  # blob too small
        gpu.Weights(cfg, blob[:100], False)
    with pytest.raises(gpu.L2ZError):  # n_heads does not divide dim
        gpu.RunState(ck.Config(dim=65, hidden_dim=8, n_layers=1, n_heads=4, n_kv_heads=2, vocab_size=8, seq_len=4))
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    with pytest.raises(gpu.L2ZError):
        s.transformer(1, cfg.seq_len, w)  # pos out of range
    with pytest.raises(gpu.L2ZError):
        s.transformer(cfg.vocab_size, 0, w)  # token out of range
    s.close(); w.close()


# ---------------------------------------------------------------- committed golden fixtures
def test_golden_toy_checkpoints_on_gpu(gpu, ck):
    """tests/golden/*.bin through l2z_weights_init (file layout) -> token ids identical to
    the committed expectation, logits within tolerance."""
    import json
    import os
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    meta = json.load(open(os.path.join(gdir, "toy_models.json")))
    for ent in meta["models"]:
        c, shared, blob = ck.read_checkpoint(os.path.join(gdir, ent["checkpoint"]))
        exp = np.load(os.path.join(gdir, ent["expected"]))
        w, s = gpu.Weights(c, np.asarray(blob), shared), gpu.RunState(c)
        s.greedy_begin(ent["prompt"])
        assert s.greedy_run(w, c.seq_len).tolist() == exp["tokens"].tolist()
        for pos, t in enumerate(exp["fed_tokens"]):
            s.transformer(int(t), pos, w)
            np.testing.assert_allclose(s.logits(), exp["logits"][pos], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
        s.close(); w.close()


FUSED_SHAPES = [  # MHA shapes the fused qkv+attention launch takes (fused_small.hip)
    ("stories15M-3layers", dict(dim=288, hidden_dim=768, n_layers=3, n_heads=6, n_kv_heads=6, vocab_size=4096, seq_len=256)),
    ("hs12", dict(dim=48, hidden_dim=128, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=300, seq_len=24)),
    ("hs64-one-head", dict(dim=64, hidden_dim=160, n_layers=2, n_heads=1, n_kv_heads=1, vocab_size=200, seq_len=64)),
    ("hs128-one-head", dict(dim=128, hidden_dim=256, n_layers=1, n_heads=1, n_kv_heads=1, vocab_size=300, seq_len=128)),
]


@pytest.mark.parametrize("name,kw", FUSED_SHAPES, ids=[c[0] for c in FUSED_SHAPES])
def test_fused_qkv_attention_launch_vs_oracle_and_unfused(gpu, ck, orc, name, kw, options):
    """Small MHA models run rmsnorm + q/k/v + RoPE + KV write + attention of a head as ONE launch
    (main.zig:305-389).  Against the oracle at every position of a full context (logits, K/V cache
    rows, q), and against the separate launches (L2Z_FUSE_SMALL=0): same tokens, logits within the
    tolerance (the summation orders differ, the values do not)."""
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, True, seed=88)
    w = gpu.Weights(cfg, blob, True)
    s_f = gpu.RunState(cfg)
    options(L2Z_FUSE_SMALL=0)
    s_u = gpu.RunState(cfg)
    options(L2Z_FUSE_SMALL=1)
    m = orc.Model(cfg.as_i32(), blob, True)
    ref_toks, margins = m.generate_greedy([4, 5], cfg.seq_len)
    for s in (s_f, s_u):
        s.greedy_begin([4, 5])
        assert np.array_equal(s.greedy_run(w, cfg.seq_len), ref_toks), (name, margins.min())
    m2 = orc.Model(cfg.as_i32(), blob, True)
    tok, worst = 1, 0.0
    kvd = cfg.kv_dim
    for pos in range(cfg.seq_len):
        ref = m2.transformer(tok, pos)
        s_f.transformer(tok, pos, w)
        got = s_f.logits()
        worst = max(worst, float(np.abs(got - ref).max()))
        np.testing.assert_allclose(got, ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{name} pos {pos}")
        if pos in (0, 1, cfg.seq_len // 2, cfg.seq_len - 1):
            s_u.transformer(tok, pos, w)
            np.testing.assert_allclose(got, s_u.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
            L = cfg.n_layers - 1  # the last layer's cache row and q of this position
            np.testing.assert_allclose(s_f.read("key_cache", (L * cfg.seq_len + pos) * kvd, kvd),
                                       m2.state("key_cache", cfg.n_layers * cfg.seq_len * kvd)[(L * cfg.seq_len + pos) * kvd:][:kvd],
                                       rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(s_f.read("value_cache", (L * cfg.seq_len + pos) * kvd, kvd),
                                       m2.state("value_cache", cfg.n_layers * cfg.seq_len * kvd)[(L * cfg.seq_len + pos) * kvd:][:kvd],
                                       rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(s_f.read("q", 0, cfg.dim), m2.state("q", cfg.dim), rtol=2e-5, atol=2e-5)
        else:
            s_u.transformer(tok, pos, w)  # keep its cache in step
        tok = int(ref_toks[pos])
    print(f"fused qkv+attention {name}: max |logit diff| vs oracle {worst:.2e}")
    for o in (s_f, s_u, w, m, m2):
        o.close()


def test_rccl_call_path_world1(gpu, ck):
    """A 1-rank RCCL communicator exercises the N>1 code path on one GPU: dlopen of
    librccl, ncclCommInitRank, the in-place ncclAllGather after every shard step, eager
    launches instead of the graph.  Tokens must equal the communicator-free run."""
    cfg = ck.Config(**TOY)
    blob = ck.synth_blob(cfg, False, 55)
    w0, s0 = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    s0.greedy_begin([3, 4])
    ref = s0.greedy_run(w0, cfg.seq_len)
    comm = gpu.Comm(0, 1, gpu.Comm.unique_id(), 0)
    w1, s1 = gpu.Weights(cfg, blob, False, comm=comm), gpu.RunState(cfg, comm=comm)
    s1.greedy_begin([3, 4])
    got = s1.greedy_run(w1, cfg.seq_len)
    assert np.array_equal(got, ref)
    s1.transformer(1, 0, w1)
    s0.transformer(1, 0, w0)
    assert np.array_equal(s1.logits(), s0.logits())
    for o in (s0, s1, w0, w1):
        o.close()
    comm.close()


@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("from_blob", [True, False], ids=["uploaded", "synthetic"])
def test_sharded_hip_path_emulated_ranks(gpu, ck, world, from_blob, options):
    """The real HIP shard path (sharded upload / on-device generation, shard offsets, sharded
    KV cache, GQA head mapping) for N = 2, 4, 8 emulated ranks on one GPU: every rank's
    logits must be BIT-IDENTICAL to the unsharded pass (scheme A: a row's dot product does
    not depend on which rank owns it)."""
    options(L2Z_FUSE_SMALL=0)  # the unsharded reference runs the launches the shards run
    cfg = ck.Config(dim=128, hidden_dim=352, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=512, seq_len=16)
    seed, shared = 17, False
    blob = ck.synth_blob(cfg, shared, seed)
    w0, s0 = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
    comms = [gpu.Comm(r, world, None, 0, emulated=True) for r in range(world)]
    ws = [gpu.Weights(cfg, blob if from_blob else None, shared, seed=seed, comm=c) for c in comms]
    ss = [gpu.RunState(cfg, comm=c) for c in comms]
    toks = [1, 77, 300, 5, 9, 400]
    for pos, t in enumerate(toks):
        s0.transformer(t, pos, w0)
        ref = s0.logits()
        gpu.emu_transformer(ss, ws, t, pos)
        for r in range(world):
            assert np.array_equal(ss[r].logits(), ref), f"rank {r} pos {pos}"
            assert ss[r].argmax() == s0.argmax()
    for o in ss + ws + [s0, w0]:
        o.close()
    for c in comms:
        c.close()


def test_prefill_scratch_follows_the_chunk_length(gpu, ck, options):
    """The prefill scratch is sized for the chunk length in force at first use; a longer chunk set later
    (L2Z_PF_CHUNK through l2z_option_set) must re-allocate it, not run past it.  (64-token chunks take
    the short-prompt GEMMs, 256-token chunks the tile GEMMs: same sums in another order.)"""
    cfg = ck.Config(dim=256, hidden_dim=704, n_layers=2, n_heads=8, n_kv_heads=4, vocab_size=512, seq_len=320)
    w, s = gpu.Weights(cfg, None, False, seed=3), gpu.RunState(cfg)
    toks = [1] + np.random.default_rng(2).integers(2, cfg.vocab_size, 299).tolist()
    options(L2Z_PF_CHUNK=64)
    s.prefill(toks, 0, w)
    small = s.logits()
    options(L2Z_PF_CHUNK=256)
    s.prefill(toks, 0, w)
    np.testing.assert_allclose(s.logits(), small, rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    s.close(); w.close()


SHARDED_PREFILL = [
    # GQA, 530 tokens: two chunks, 128- and 64-token tiles, tiled attention, a partial last chunk
    ("gqa", dict(dim=512, hidden_dim=1408, n_layers=2, n_heads=16, n_kv_heads=8, vocab_size=1024, seq_len=544), 530),
    # few heads: the per-query attention kernel; 40 tokens: the skinny (P <= 64) GEMM forms
    ("mha-short", dict(dim=256, hidden_dim=704, n_layers=3, n_heads=8, n_kv_heads=8, vocab_size=512, seq_len=64), 40),
    # matrices that stream from HBM (> 16 MB each): the tile GEMM's split-K family -- 4 K ranges per tile for the
    # 51-token second call, 2 for a 101-token one -- whose range count comes from the WHOLE model's shape, so a
    # rank's [tokens, n / world] block has the bits of the unsharded pass
    ("streams-60", dict(dim=3072, hidden_dim=8192, n_layers=2, n_heads=24, n_kv_heads=8, vocab_size=2048, seq_len=128), 60),
    ("streams-110", dict(dim=3072, hidden_dim=8192, n_layers=2, n_heads=24, n_kv_heads=8, vocab_size=2048, seq_len=128), 110),
    # the same matrices, chunks of 9 and 21 tokens: the short-prompt kernels on matrices that stream (one and two
    # token tiles per block by the chunk length alone, W1 | W3 and wk | wv paired)
    ("streams-30", dict(dim=3072, hidden_dim=8192, n_layers=2, n_heads=24, n_kv_heads=8, vocab_size=2048, seq_len=128), 30),
]


@pytest.mark.parametrize("x3", [1, 2], ids=["cores-by-shape", "bf16-cores-forced"])
@pytest.mark.parametrize("world", [2, 4, 8])
@pytest.mark.parametrize("name,kw,n_tok", SHARDED_PREFILL, ids=[c[0] for c in SHARDED_PREFILL])
def test_sharded_prefill_emulated_ranks_bit_identical(gpu, ck, world, name, kw, n_tok, options, x3):
    """Row-sharded batched prefill (prefill_host.cpp stages: heads / rows of wo, w1|w3, w2 per rank,
    [tokens, n / world] blocks exchanged and unpacked) on N emulated ranks: every rank's logits, and
    its shard of every layer's KV cache, are BIT-IDENTICAL to the unsharded l2z_prefill -- a tile's k
    order does not depend on which rank owns its rows, and shards take the attention form the whole
    model takes.  Also with pos0 > 0 (a second call continuing the context).  x3 = 2: every product on the bf16 matrix cores
    (the streams-* shapes take them by default: stream form at 51 / 101 tokens, its K ranges from the WHOLE model's rows);
    the planes kernels' k order and ranges do not depend on the rows a rank owns either."""
    if x3 == 2 and name.startswith("streams"):
        pytest.skip("these matrices take the bf16 cores by default")
    options(L2Z_FUSE_SMALL=0, L2Z_PF_X3=x3)  # (fuse: the decode step at the end -- the unsharded pass runs the launches the shards run)
    cfg = ck.Config(**kw)
    w0, s0 = gpu.Weights(cfg, None, False, seed=23), gpu.RunState(cfg)
    comms = [gpu.Comm(r, world, None, 0, emulated=True) for r in range(world)]
    ws = [gpu.Weights(cfg, None, False, seed=23, comm=c) for c in comms]
    ss = [gpu.RunState(cfg, comm=c) for c in comms]
    rng = np.random.default_rng(4)
    toks = [1] + rng.integers(2, cfg.vocab_size, n_tok - 1).tolist()
    split = 9
    for lo, hi in ((0, split), (split, n_tok)):
        s0.prefill(toks[lo:hi], lo, w0)
        gpu.emu_prefill(ss, ws, toks[lo:hi], lo)
        ref = s0.logits()
        for r in range(world):
            assert np.array_equal(ss[r].logits(), ref), f"rank {r} logits after tokens {lo}..{hi}"
    kvd, S = cfg.kv_dim, cfg.seq_len
    kvl = kvd // world
    for l in range(cfg.n_layers):
        for nm in ("key_cache", "value_cache"):
            full = s0.read(nm, l * S * kvd, n_tok * kvd).reshape(n_tok, kvd)
            for r in range(world):
                mine = ss[r].read(nm, l * S * kvl, n_tok * kvl).reshape(n_tok, kvl)
                assert np.array_equal(mine, full[:, r * kvl:(r + 1) * kvl]), f"{nm} layer {l} rank {r}"
    # decoding continues identically from the sharded state
    nxt = s0.argmax()
    s0.transformer(nxt, n_tok, w0)
    gpu.emu_transformer(ss, ws, nxt, n_tok)
    for r in range(world):
        assert np.array_equal(ss[r].logits(), s0.logits())
    for o in ss + ws + [s0, w0]:
        o.close()
    for c in comms:
        c.close()


def test_shard_that_cannot_take_the_unsharded_kernel_is_refused_not_rerouted(gpu, ck, options):
    """which prefill kernel a product takes is decided from the WHOLE model (so a shard sums in the order the
    unsharded pass sums in), but the K-range panel kernel also needs a multiple of 16 rows per matrix ON THE RANK.  Here
    W1 | W3 of the whole model (2 x 2056 rows x 1152: 18.9 MB, streams from HBM) takes the panel kernel at 20 tokens and
    a rank of 2 holds 2 x 1028 = 2056 rows, 2056 % 16 = 8: until round 5 that rank silently fell back to another kernel
    (another summation order).  Now the group keeps off the batched path: l2z_emu_prefill refuses, and the world that
    does split into multiples of 16 (hidden_dim 2048) still agrees bit for bit."""
    options(L2Z_FUSE_SMALL=0)
    toks = [1] + np.random.default_rng(8).integers(2, 500, 19).tolist()
    for hidden, ok in ((2056, False), (2048, True)):
        cfg = ck.Config(dim=1152, hidden_dim=hidden, n_layers=1, n_heads=18, n_kv_heads=18, vocab_size=512, seq_len=64)
        w0, s0 = gpu.Weights(cfg, None, True, seed=5), gpu.RunState(cfg)
        comms = [gpu.Comm(r, 2, None, 0, emulated=True) for r in range(2)]
        ws = [gpu.Weights(cfg, None, True, seed=5, comm=c) for c in comms]
        ss = [gpu.RunState(cfg, comm=c) for c in comms]
        s0.prefill(toks, 0, w0)   # the unsharded pass takes the panel kernel either way (4112 and 4096 rows: % 16 == 0)
        if ok:
            gpu.emu_prefill(ss, ws, toks, 0)
            assert all(np.array_equal(s.logits(), s0.logits()) for s in ss)
        else:
            with pytest.raises(gpu.L2ZError, match="multiples of 16"):
                gpu.emu_prefill(ss, ws, toks, 0)
        for o in ss + ws + [s0, w0]:
            o.close()
        for c in comms:
            c.close()


@pytest.mark.parametrize("nch", [2, 3, 16])
def test_split_attention_matches_oracle(gpu, ck, orc, nch, options):
    """The flash-decoding attention (nch blocks per head, the last arriver combines) is forced on
    for toy models (it is auto-enabled only for seq_len > 256): logits within tolerance at every
    position, greedy tokens identical, including positions < nch (empty chunks)."""
    options(L2Z_ATTN_SPLIT=nch)
    for name, kw, shared in CONFIGS[:3]:
        cfg = ck.Config(**kw)
        blob = ck.synth_blob(cfg, shared, seed=61)
        w, s = gpu.Weights(cfg, blob, shared), gpu.RunState(cfg)
        m = orc.Model(cfg.as_i32(), blob, shared)
        ref_toks, _ = m.generate_greedy([2, 3], cfg.seq_len)
        s.greedy_begin([2, 3])
        assert np.array_equal(s.greedy_run(w, cfg.seq_len), ref_toks), name
        m2 = orc.Model(cfg.as_i32(), blob, shared)
        tok = 1
        for pos in range(cfg.seq_len):
            ref = m2.transformer(tok, pos)
            s.transformer(tok, pos, w)
            np.testing.assert_allclose(s.logits(), ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
            tok = int(ref_toks[pos])
        s.close(); w.close(); m.close(); m2.close()


def test_greedy_loop_edges(gpu, ck, orc):
    """main.zig:1017 (a BOS -- even one forced by the prompt -- ends the sequence), :992-993
    (steps clamp to seq_len), chunked calls, restart."""
    cfg = ck.Config(**TOY)
    blob = ck.synth_blob(cfg, False, 77)
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    m = orc.Model(cfg.as_i32(), blob, False)
    # BOS inside the prompt: the loop stops right there, on both sides
    ref, _ = m.generate_greedy([5, 1, 7], cfg.seq_len)
    assert ref.tolist() == [5, 1]
    s.greedy_begin([5, 1, 7])
    assert s.greedy_run(w, cfg.seq_len).tolist() == [5, 1]
    assert s.greedy_run(w, 4).size == 0            # the sequence is over until the next begin
    # more steps than seq_len: clamped (main.zig:993)
    ref, _ = m.generate_greedy([], cfg.seq_len)
    s.greedy_begin([])
    got = s.greedy_run(w, 10 * cfg.seq_len)
    assert np.array_equal(got, ref) and len(got) <= cfg.seq_len
    assert s.greedy_run(w, 1).size == 0            # pos == seq_len: nothing left
    # one token at a time == all at once
    s.greedy_begin([])
    one = [int(s.greedy_run(w, 1)[0]) for _ in range(len(ref))]
    assert one == ref.tolist()
    assert s.greedy_run(w, 0).size == 0
    s.close(); w.close(); m.close()


# ---------------------------------------------------------------- batched prefill (MFMA GEMM)
@pytest.mark.gpu
@pytest.mark.parametrize("n_prompt", [3, 4, 21, 70])
def test_greedy_prompt_through_prefill(gpu, ck, n_prompt):
    """l2z_greedy_run with a prompt (batched pass over the prompt positions when it has >= 4
    tokens) produces the tokens of the stepped loop: prompt echoed, then argmax (main.zig:995-1036).
    A differing token is accepted only at a near tie of the stepped path's top two logits."""
    cfg = ck.Config(dim=64, hidden_dim=172, n_layers=3, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=128)
    blob = ck.synth_blob(cfg, False, seed=17)
    w = gpu.Weights(cfg, blob, False)
    s1, s2 = gpu.RunState(cfg), gpu.RunState(cfg)
    prompt = np.random.default_rng(3).integers(2, cfg.vocab_size, n_prompt).tolist()
    n_steps = 100
    s2.greedy_begin(prompt)
    got = s2.greedy_run(w, n_steps).tolist()
    token, want = 1, []
    for pos in range(n_steps):
        s1.transformer(token, pos, w)
        if pos < n_prompt:
            nxt = prompt[pos]
        else:
            nxt = s1.argmax()
            if len(got) > pos and got[pos] != nxt:
                lg = np.sort(s1.logits())
                assert lg[-1] - lg[-2] < 1e-4, f"pos {pos}: {got[pos]} vs {nxt}, margin {lg[-1]-lg[-2]}"
                break
        want.append(nxt)
        if nxt == 1:
            break
        token = nxt
    assert got[: len(want)] == want
    assert got[:n_prompt] == prompt
    # a first call shorter than the prompt keeps the stepped loop and gives the same tokens
    s2.greedy_begin(prompt)
    a = s2.greedy_run(w, 2).tolist() + s2.greedy_run(w, n_steps - 2).tolist()
    assert a[: len(want)] == want
    # a BOS inside a long prompt: no batched pass, the loop ends on it (main.zig:1017)
    if n_prompt >= 8:
        p2 = list(prompt)
        p2[6] = 1
        s2.greedy_begin(p2)
        assert s2.greedy_run(w, n_steps).tolist() == p2[:7]
    # a prompt that fills the whole context: every position is forced, nothing is generated
    full = np.random.default_rng(4).integers(2, cfg.vocab_size, cfg.seq_len).tolist()
    s2.greedy_begin(full)
    assert s2.greedy_run(w, 10 * cfg.seq_len).tolist() == full
    assert s2.greedy_run(w, 1).size == 0
    for o in (s1, s2, w):
        o.close()



PREFILL_CONFIGS = [
    ("toy-gqa", dict(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=96), False),
    ("toy-mqa", dict(dim=96, hidden_dim=256, n_layers=2, n_heads=6, n_kv_heads=1, vocab_size=1000, seq_len=80), True),
    ("stories15M-shape-2layers", dict(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=560), True),
    # many heads: the tiled (flash-form) prefill attention runs from n_heads * ceil(P/64) >= 128
    ("32-heads-hs16-gqa", dict(dim=512, hidden_dim=1024, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=1024, seq_len=560), False),
    ("16-heads-hs48", dict(dim=768, hidden_dim=1024, n_layers=2, n_heads=16, n_kv_heads=16, vocab_size=1024, seq_len=560), True),
    ("16-heads-hs64-gqa", dict(dim=1024, hidden_dim=1536, n_layers=2, n_heads=16, n_kv_heads=4, vocab_size=1024, seq_len=560), False),
    # matrices that stream from HBM (> 16 MB): the K-range panel kernel at 17 ... 96 tokens, the split-K family at 65 / 79
    # (wo, W2: 96 blocks of 32 x 64), and at 523 tokens 128 x 64 tiles with the two k-groups on two blocks
    ("streams-2048", dict(dim=2048, hidden_dim=5632, n_layers=2, n_heads=16, n_kv_heads=16, vocab_size=1024, seq_len=560), False),
]


@pytest.mark.parametrize("x3", [1, 2], ids=["cores-by-shape", "bf16-cores-forced"])
@pytest.mark.parametrize("name,kw,shared", PREFILL_CONFIGS, ids=[c[0] for c in PREFILL_CONFIGS])
def test_prefill_equals_token_by_token(gpu, ck, orc, options, name, kw, shared, x3):
    """l2z_prefill(tokens, pos0) leaves the KV cache and the last position's logits as n calls
    of l2z_transformer do (within the logit tolerance: the GEMM sums in MFMA k-order), also
    when it continues an existing context (pos0 > 0) and spans more than one 512-token chunk.
    x3 = 1: matrices that stream from HBM multiply on the bf16 matrix cores (three-term splits), the others on the f32
    ones -- the default; x3 = 2: every matrix on the bf16 cores, so that the planes kernels (tile forms from 129 tokens,
    the stream form at 33 ... 128 where K >= 256) meet every toy shape, head layout and ragged width here."""
    options(L2Z_PF_X3=x3)
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=91)
    w = gpu.Weights(cfg, blob, shared)
    s1, s2 = gpu.RunState(cfg), gpu.RunState(cfg)
    rng = np.random.default_rng(9)
    n_total = min(cfg.seq_len - 4, 530)
    toks = [1] + rng.integers(2, cfg.vocab_size, n_total - 1).tolist()
    for pos, t in enumerate(toks):
        s1.transformer(t, pos, w)
    ref_logits = s1.logits()
    split = 7  # first 7 tokens one call, the rest a second call (pos0 > 0)
    s2.prefill(toks[:split], 0, w)
    s2.prefill(toks[split:], split, w)
    got = s2.logits()
    np.testing.assert_allclose(got, ref_logits, rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    assert s2.argmax() == s1.argmax()
    kvd, S = cfg.kv_dim, cfg.seq_len
    for l in range(cfg.n_layers):
        for name_ in ("key_cache", "value_cache"):
            a = s1.read(name_, l * S * kvd, n_total * kvd)
            b = s2.read(name_, l * S * kvd, n_total * kvd)
            np.testing.assert_allclose(b, a, rtol=2e-4, atol=2e-4)
    # decoding continues from the prefilled state exactly like from the stepped one
    nxt = s1.argmax()
    s1.transformer(nxt, n_total, w); s2.transformer(nxt, n_total, w)
    np.testing.assert_allclose(s2.logits(), s1.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    # and against the CPU oracle
    m = orc.Model(cfg.as_i32(), blob, shared)
    for pos, t in enumerate(toks[:24]):
        ref = m.transformer(t, pos)
    s3 = gpu.RunState(cfg)
    s3.prefill(toks[:24], 0, w)
    np.testing.assert_allclose(s3.logits(), ref, rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    m.close()
    # every tile path: 16x16x4 skinny kernel with 1, 2, 4 token tiles (n <= 16, 32, 64), the
    # 64x64 LDS-tiled GEMM beyond, partial tiles on both sides of each boundary
    for n in (1, 2, 15, 16, 17, 32, 33, 50, 64, 65, min(79, n_total)):
        for pos, t in enumerate(toks[:n]):
            s1.transformer(t, pos, w)
        s3.prefill(toks[:n], 0, w)
        np.testing.assert_allclose(s3.logits(), s1.logits(), rtol=LOGIT_RTOL, atol=LOGIT_ATOL,
                                   err_msg=f"n={n}")
        for l in range(cfg.n_layers):
            for name_ in ("key_cache", "value_cache"):
                a = s1.read(name_, l * S * kvd, n * kvd)
                b = s3.read(name_, l * S * kvd, n * kvd)
                np.testing.assert_allclose(b, a, rtol=2e-4, atol=2e-4, err_msg=f"n={n} {name_} l={l}")
    for o in (s1, s2, s3, w):
        o.close()


@pytest.mark.parametrize("kv_heads", [24, 8])
def test_prefill_panel_kernel_vs_oracle(gpu, ck, orc, options, kv_heads):
    """prefill_panel.hip: chunks of 17 ... 96 tokens of matrices that stream from HBM (K cut into ranges with a resident X
    panel, the ranges added in order by a second launch that also runs the epilogue -- RoPE + cache rows, residual,
    SiLU * mul).  A wide two-layer shape whose hidden_dim leaves a SHORT last range (8448 = 16.5 x 512 = 33 x 256), MHA and
    GQA: logits and KV rows against the CPU oracle's stepped loop for chunks of 16 (short-prompt GEMMs: below the panel
    kernel's range), 17 / 32 (two token tiles, ranges of 512), 33 / 48 (three, 512), 49 / 64 (four, 256), 65 / 80 / 81 / 96
    (five and six, 256: round 6), a second call continuing the context; and against the short-prompt GEMMs
    (L2Z_PF_PANEL=0) within the tolerance."""
    kw = dict(dim=3072, hidden_dim=8448, n_layers=2, n_heads=24, n_kv_heads=kv_heads, vocab_size=2048, seq_len=112)
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, False, seed=55)
    w, s = gpu.Weights(cfg, blob, False), gpu.RunState(cfg)
    m = orc.Model(cfg.as_i32(), blob, False)
    rng = np.random.default_rng(12)
    toks = [1] + rng.integers(2, cfg.vocab_size, 103).tolist()
    kvd, S = cfg.kv_dim, cfg.seq_len
    lens = (16, 17, 32, 33, 48, 49, 64, 65, 80, 81, 96)
    ref = {}
    for pos, t in enumerate(toks):       # ONE pass of the oracle: its logits after position n - 1 are the n-token prompt's
        lg = m.transformer(t, pos)
        if pos + 1 in lens or pos + 1 in (40, 104):
            ref[pos + 1] = lg
    ref_cache = {nm: m.state(nm, cfg.n_layers * S * kvd).reshape(cfg.n_layers, S, kvd) for nm in ("key_cache", "value_cache")}
    worst = 0.0
    # L2Z_PF_X3=0 first: every GEMM on the f32 matrix cores -- the panel kernel at all of 17 ... 96 tokens (by default chunks
    # of 33 ... 128 tokens of such matrices take the stream form of the bf16-core kernel); then the defaults
    for x3 in (0, 1):
        options(L2Z_PF_X3=x3)
        for n in lens:
            s.prefill(toks[:n], 0, w)
            got = s.logits()
            worst = max(worst, float(np.abs(got - ref[n]).max()))
            np.testing.assert_allclose(got, ref[n], rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=f"{n} tokens, L2Z_PF_X3={x3}")
            for l in range(cfg.n_layers):
                for nm in ("key_cache", "value_cache"):
                    np.testing.assert_allclose(s.read(nm, l * S * kvd, n * kvd), ref_cache[nm][l, :n].ravel(), rtol=2e-5, atol=2e-5,
                                               err_msg=f"{nm} l={l} n={n} L2Z_PF_X3={x3}")
    # a second call continuing the context (pos0 = 96), 8 more tokens
    s.prefill(toks[96:104], 96, w)
    np.testing.assert_allclose(s.logits(), ref[104], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    # 32 + 8 tokens: the panel kernel and the short-prompt GEMMs: another summation order, the same tolerance
    got = {}
    for tag, opts in (("default", {}), ("skinny", dict(L2Z_PF_PANEL=0))):
        options(**opts)
        s.prefill(toks[:32], 0, w); s.prefill(toks[32:40], 32, w)
        got[tag] = s.logits()
        np.testing.assert_allclose(got[tag], ref[40], rtol=LOGIT_RTOL, atol=LOGIT_ATOL, err_msg=tag)
        options(L2Z_PF_PANEL=1)
    assert not np.array_equal(got["skinny"], got["default"]), "L2Z_PF_PANEL=0 did not change the path"
    # ... and the two sides of the upper switch-over: 96 tokens on the tile GEMM (L2Z_PF_PANEL_MAX=64)
    options(L2Z_PF_PANEL_MAX=64)
    s.prefill(toks[:96], 0, w)
    tile = s.logits()
    options(L2Z_PF_PANEL_MAX=-1)
    np.testing.assert_allclose(tile, ref[96], rtol=LOGIT_RTOL, atol=LOGIT_ATOL)
    print(f"panel kernel, kv heads {kv_heads}: max |logit - oracle| {worst:.2e}")
    m.close(); s.close(); w.close()


@pytest.mark.parametrize("temp", [1.0, 0.7])
def test_probs_read_vs_host_softmax(gpu, ck, orc, temp):
    """l2z_probs_read = main.zig:1005-1008 (logits / temperature, softmax) on the device, for the host
    samplers -- a deliberate parity trade, bounded here.  The device divides and exponentiates like the host
    (IEEE divide, expf within an ulp or two) but reduces the DENOMINATOR as a tree; the reference adds the
    32000 terms in order, and that in-order fp32 sum itself sits ~1e-5 away from the exact sum (measured:
    5e-6 at -t 1.0, 1.6e-5 at -t 0.7).  So: (a) against an exact (float64) softmax of the same logits the
    device is within 2e-6 relative; (b) against the host's in-order softmax within 5e-5 relative, and the
    difference is ONE common factor (the two denominators): got / ref is constant over the vocabulary to
    1e-6.  A common factor cancels in sample_top_p's r = coin * cumulative, so draws only move at a cdf
    boundary (tests/test_host_cli.py compares token ids against a host replay and reports the margin)."""
    cfg = ck.Config(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=64)
    w, s = gpu.Weights(cfg, None, True, seed=3), gpu.RunState(cfg)
    for pos, tok in enumerate([1, 500, 9999]):
        s.transformer(tok, pos, w)
        lg = s.logits()
        got = s.probs(temp)
        x = (lg / np.float32(temp)).astype(np.float32)
        ref = orc.softmax(x)
        e = np.exp(x.astype(np.float64) - float(x.max()))
        exact = e / e.sum()
        rel_exact = float(np.max(np.abs(got - exact) / exact))
        rel_host = float(np.max(np.abs(got - ref) / (ref + 1e-30)))
        ratio = got.astype(np.float64) / ref.astype(np.float64)
        big = ref > 1e-7   # where an ulp of the value is small against the common factor
        print(f"-t {temp} pos {pos}: device vs exact {rel_exact:.2e}, device vs in-order host {rel_host:.2e}, "
              f"spread of got/ref {float(ratio[big].max() - ratio[big].min()):.2e}")
        assert rel_exact <= 2e-6 and rel_host <= 5e-5
        assert float(ratio[big].max() - ratio[big].min()) <= 1e-6
        assert abs(float(got.astype(np.float64).sum()) - 1.0) < 1e-5
        assert int(np.argmax(got)) == int(np.argmax(ref))
        assert np.array_equal(s.logits(), lg)   # the logits themselves are left untouched
    s.close(); w.close()


# ---------------------------------------------------------------- RoPE, driven directly (main.zig:336-351)
ROPE_SHAPES = [("hs48-mha", 288, 6, 6), ("hs48-gqa", 288, 6, 2), ("hs64-gqa", 512, 8, 2),
               ("hs128-gqa", 1024, 8, 4), ("hs128-rowkernel-gqa", 4096, 32, 8)]


@pytest.mark.parametrize("name,dim,H,KV", ROPE_SHAPES, ids=[c[0] for c in ROPE_SHAPES])
def test_rope_epilogue_at_positions_vs_oracle(gpu, ck, orc, options, name, dim, H, KV):
    """the EPI_ROPE epilogue of the q|k|v launch (matvec_device.h; fused_small.hip for small MHA
    models) at pos > 0, up to the last row of a 2048-row cache, head sizes 48 / 64 / 128, MHA and GQA (pairs
    with i >= kv_dim rotate q only, main.zig:343).  Two bars on a one-layer model (so RunState.q is layer 0's):
      * the rotation ALONE: the device's own pos-0 q / k rows (cos 1, sin 0: un-rotated) run through the
        oracle's inline RoPE for `pos` must give the device's q / K-cache row at `pos` to an ulp of the pair's
        magnitude -- same inputs, only the rotation differs; V rows do not change with pos at all;
      * the whole launch: q and the K row against the oracle's own pass at `pos`, one dot product deep."""
    hs, kv_dim = dim // H, (dim // H) * KV
    cfg = ck.Config(dim=dim, hidden_dim=64, n_layers=1, n_heads=H, n_kv_heads=KV, vocab_size=64, seq_len=2048)
    blob = ck.synth_blob(cfg, True, seed=77)
    m = orc.Model(cfg.as_i32(), blob, True)
    tok = 5
    forms = (1, 0) if KV == H else (0,)   # the fused small-model launch only exists for MHA shapes
    for fuse in forms:
        options(L2Z_FUSE_SMALL=fuse)
        w, s = gpu.Weights(cfg, blob, True), gpu.RunState(cfg)
        s.transformer(tok, 0, w)
        q0 = s.read("q", 0, dim)
        k0 = s.read("key_cache", 0, kv_dim)
        v0 = s.read("value_cache", 0, kv_dim)
        m.transformer(tok, 0)
        assert np.allclose(q0, m.state("q", dim), rtol=2e-5, atol=2e-5)
        for pos in (1, 2, 7, 300, 1023, 2047):
            s.transformer(tok, pos, w)
            q = s.read("q", 0, dim)
            k = s.read("key_cache", pos * kv_dim, kv_dim)
            v = s.read("value_cache", pos * kv_dim, kv_dim)
            assert np.array_equal(v, v0), (name, fuse, pos)
            want_q, want_k = orc.rope(q0, k0, pos, hs)
            for got, want, src in ((q, want_q, q0), (k, want_k, k0)):
                pair_mag = np.repeat(np.abs(src.reshape(-1, 2)).sum(axis=1), 2)
                assert np.all(np.abs(got - want) <= 2.4e-7 * pair_mag + 1e-12), \
                    (name, fuse, pos, float(np.max(np.abs(got - want) / (pair_mag + 1e-30))))
            assert not np.array_equal(q, q0) and not np.array_equal(k, k0)
            m.transformer(tok, pos)
            assert np.allclose(q, m.state("q", dim), rtol=2e-5, atol=2e-5), (name, fuse, pos)
            assert np.allclose(k, m.state("key_cache", 2048 * kv_dim)[pos * kv_dim:(pos + 1) * kv_dim],
                               rtol=2e-5, atol=2e-5), (name, fuse, pos)
        s.close(); w.close()
    m.close()
