"""CPU tests (-m "not gpu"): the oracle against the reference's own known-answer
tests and against an independent float64 restatement; checkpoint format;
golden fixtures; the C ABI library loads and exports every declared symbol.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from ref_numpy import NumpyModel

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(params=[(4, 0, 0), (8, 0, 0), (8, 1, 1), (16, 0, 1), (16, 1, 0)],
                ids=lambda m: f"vw{m[0]}-fma{m[1]}-tree{m[2]}")
def mode(request, orc):
    orc.set_mode(*request.param)
    yield request.param
    orc.set_mode(8, False, False)


# ---- the reference's known-answer tests (src/main.zig:1078-1150), all readings ----
def test_kat_matrix_multiplies(orc, mode):
    w = np.arange(1, 10, dtype=np.float32).reshape(3, 3)
    assert orc.matmul(np.array([1, 2, 3], np.float32), w).tolist() == [14.0, 32.0, 50.0]


def test_kat_vector_length_less_than_width_case(orc, mode):
    w = np.arange(1, 25, dtype=np.float32).reshape(2, 12)
    x = np.arange(1, 13, dtype=np.float32)
    exp = [float(sum(w[i, j] * x[j] for j in range(12))) for i in range(2)]
    assert orc.matmul(x, w).tolist() == exp


def test_kat_vector_weighted_sum(orc, mode):
    """main.zig:1105-1115 (dead code in the reference, one-sided assertion)"""
    x = np.arange(1, 13, dtype=np.float32)
    out = orc.vector_weighted_sum(x, x, 3.0)
    assert np.all((out - (x * 3.0 + x)) < 1e-4)


def test_kat_vector_weighted_sum_rows(orc, mode):
    """main.zig:1117-1139: width = DEFAULT_VECTOR_WIDTH + 3, stride = width + 2"""
    width = mode[0] + 3
    stride, weights = width + 2, np.array([0.25, -0.5, 1.5], np.float32)
    rows = np.zeros(stride * 3, np.float32)
    for r in range(3):
        rows[r * stride : r * stride + width] = np.arange(r * width + 1, r * width + width + 1)
    out = orc.vector_weighted_sum_rows(width, rows, stride, weights)
    exp = sum(rows[r * stride : r * stride + width].astype(np.float64) * float(weights[r]) for r in range(3))
    assert np.all(np.abs(out - exp) <= 1e-5)


def test_kat_softmax(orc, mode):
    s = orc.softmax(np.array([1, 2, 3, 4], np.float32))
    acc = np.float32(0)
    for v in s:
        acc = np.float32(acc + v)
    assert acc == np.float32(1.0)


def test_golden_kat_file(orc):
    """tests/golden/reference_kats.json holds the reference's vectors as data."""
    kats = json.load(open(os.path.join(GOLDEN, "reference_kats.json")))
    for k in kats["matmul"]:
        w = np.array(k["w"], np.float32).reshape(k["d"], k["n"])
        assert orc.matmul(np.array(k["x"], np.float32), w).tolist() == k["expect"]


# ---- kernels against float64 ----
@pytest.mark.parametrize("d,n", [(1, 1), (5, 7), (3, 64), (9, 130), (4, 288), (2, 4096)])
def test_matmul_fused_matches_float64(orc, mode, d, n):
    rng = np.random.default_rng(n * 31 + d)
    x = rng.standard_normal(n, dtype=np.float32)
    ws = [rng.standard_normal((d, n), dtype=np.float32) for _ in range(3)]
    for N in (1, 2, 3):
        outs = orc.matmul_fused(x, ws[:N])
        for j in range(N):
            ref = ws[j].astype(np.float64) @ x.astype(np.float64)
            bound = 4e-6 * (np.abs(ws[j].astype(np.float64)) @ np.abs(x.astype(np.float64)))
            assert np.all(np.abs(outs[j] - ref) <= bound + 1e-30)
    # N=1 uses 8 accumulators, N>1 uses 4 (main.zig:546): same math, different order
    assert np.allclose(orc.matmul(x, ws[0]), orc.matmul_fused(x, ws[:2])[0], rtol=1e-4, atol=1e-4)


def test_rmsnorm_eps_after_divide(orc, mode):
    """main.zig:452-453: eps is added after sum/n -- visible for tiny inputs."""
    x = np.full(16, 1e-3, np.float32)
    w = np.ones(16, np.float32)
    out = orc.rmsnorm(x, w)
    assert np.allclose(out, 1e-3 / np.sqrt(1e-6 + 1e-5), rtol=1e-5)
    y = np.array(x)
    # aliasing o == x (main.zig:426) is allowed
    assert np.array_equal(orc.rmsnorm(y, w), out)


def test_argmax_first_maximum(orc):
    assert orc.argmax(np.array([1, 3, 3, 2], np.float32)) == 1
    assert orc.argmax(np.array([-np.inf, -np.inf], np.float32)) == 0


# ---- whole pass against the independent numpy model ----
CASES = [
    ("gqa-unshared", dict(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=32), False),
    ("mha-shared", dict(dim=48, hidden_dim=128, n_layers=3, n_heads=4, n_kv_heads=4, vocab_size=300, seq_len=24), True),
    ("mqa", dict(dim=96, hidden_dim=256, n_layers=2, n_heads=6, n_kv_heads=1, vocab_size=200, seq_len=16), True),
]


@pytest.mark.parametrize("name,kw,shared", CASES, ids=[c[0] for c in CASES])
def test_transformer_matches_independent_float64(orc, ck, mode, name, kw, shared):
    cfg = ck.Config(**kw)
    blob = ck.synth_blob(cfg, shared, seed=3)
    m, ref = orc.Model(cfg.as_i32(), blob, shared), NumpyModel(ck, cfg, blob, shared)
    toks = [1, 17, 3, 250 % cfg.vocab_size, 9, 11, 2, 5]
    for pos, t in enumerate(toks):
        got, exp = m.transformer(t, pos), ref.transformer(t, pos)
        np.testing.assert_allclose(got, exp, rtol=2e-4, atol=2e-4, err_msg=f"{name} pos {pos}")
    m.close()


def test_reference_readings_spread_is_small(orc, ck):
    """How far apart the 12 readings of the reference are from each other: the
    yardstick for the GPU tolerance (tests/test_gpu_parity.py)."""
    cfg = ck.Config(**CASES[0][1])
    blob = ck.synth_blob(cfg, False, 3)
    outs = []
    for md in orc.ALL_MODES:
        orc.set_mode(*md)
        m = orc.Model(cfg.as_i32(), blob, False)
        outs.append(np.stack([m.transformer(t, p) for p, t in enumerate([1, 5, 9, 2])]))
        m.close()
    orc.set_mode(8, False, False)
    spread = max(float(np.abs(o - outs[0]).max()) for o in outs)
    assert 0 < spread < 1e-4


# ---- golden fixtures generated by tests/golden/make_golden.py ----
def test_golden_toy_checkpoints(orc, ck):
    meta = json.load(open(os.path.join(GOLDEN, "toy_models.json")))
    for ent in meta["models"]:
        c, shared, blob = ck.read_checkpoint(os.path.join(GOLDEN, ent["checkpoint"]), mmap=False)
        assert shared == ent["shared"]
        assert np.array_equal(blob, ck.synth_blob(c, shared, ent["seed"]))  # generator is stable
        exp = np.load(os.path.join(GOLDEN, ent["expected"]))
        m = orc.Model(c.as_i32(), blob, shared)
        toks, _ = m.generate_greedy(ent["prompt"], c.seq_len)
        assert toks.tolist() == exp["tokens"].tolist()
        m2 = orc.Model(c.as_i32(), blob, shared)
        for pos, t in enumerate(exp["fed_tokens"]):
            lg = m2.transformer(int(t), pos)
            assert np.array_equal(lg, exp["logits"][pos])  # the oracle is deterministic
        m.close(); m2.close()


# ---- checkpoint format ----
def test_checkpoint_sizes_match_public_files(ck):
    """the computed sizes equal the public files' sizes."""
    assert ck.file_size(ck.STORIES15M, True) == 60_816_028
    assert ck.file_size(ck.STORIES110M, True) == 438_381_596
    assert ck.file_size(ck.LLAMA2_7B, False) == 26_954_711_068


def test_checkpoint_roundtrip_and_sign_convention(ck, tmp_path):
    cfg = ck.Config(dim=16, hidden_dim=40, n_layers=1, n_heads=2, n_kv_heads=1, vocab_size=20, seq_len=8)
    for shared in (True, False):
        blob = ck.synth_blob(cfg, shared, 1)
        p = tmp_path / f"m{int(shared)}.bin"
        ck.write_checkpoint(p, cfg, blob, shared)
        raw = np.fromfile(p, dtype="<i4", count=7)
        assert (raw[5] > 0) == shared and abs(raw[5]) == 20  # main.zig:943
        c2, sh2, b2 = ck.read_checkpoint(p)
        assert c2 == cfg and sh2 == shared and np.array_equal(np.asarray(b2), blob)
        assert os.path.getsize(p) == ck.file_size(cfg, shared)
    t = {x.name: x for x in ck.tensor_table(cfg, False)}
    # freq_cis gap sits between rms_final and wcls (main.zig:108-112)
    assert t["wcls"].offset == t["freq_cis_imag"].offset + cfg.seq_len * cfg.head_size // 2
    assert ck.carve(cfg, ck.synth_blob(cfg, True, 1), True)["wcls"].base is not None


def test_oracle_weights_count_matches_python(orc, ck):
    import ctypes as C
    for _, cfg, shared in ck.iter_configs():
        c = orc.OrcConfig(*[int(v) for v in cfg.as_i32()])
        assert orc.lib().orc_weights_count(C.byref(c), int(shared)) == ck.weights_count(cfg, shared)


def test_synth_generators_agree(orc, ck):
    cfg = ck.Config(dim=32, hidden_dim=88, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=100, seq_len=16)
    for shared in (True, False):
        assert np.array_equal(ck.synth_blob(cfg, shared, 42), orc.synth_fill(cfg.as_i32(), shared, 42, 3))


# ---- the product library: loads without a GPU, exports every declared symbol ----
def test_c_abi_exports_every_declared_symbol(B):
    lib = B.lib()
    decl = B.declared_symbols()
    assert len(decl) >= 25
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.l2z_abi_version() == 2  # 2: test-only entry points moved to llama2_hip_test.h


def test_no_cpu_fallback_without_device(B):
    """On a machine with no GPU the compute entry points must fail loudly."""
    if B.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(B.L2ZError) as e:
        B.matmul(np.ones(3, np.float32), np.ones((3, 3), np.float32))
    assert e.value.code == B.ERR_NO_DEVICE
    cfg = [8, 16, 1, 2, 2, 10, 4]
    with pytest.raises(B.L2ZError):
        B.RunState(cfg)
    with pytest.raises(B.L2ZError):
        B.Weights(cfg, None, True, seed=1)


def test_product_does_not_reference_the_oracle():
    """The product path must not import, link or fall back to anything under oracle/."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "llama2.zig_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                for needle in ("liboracle", "llama2_oracle.h", "from oracle", "import oracle",
                               "load_oracle"):
                    assert needle not in txt, f"{f} references the oracle ({needle})"
                import re
                assert not re.search(r"\borc_[a-z_0-9]+\s*\(", txt), f"{f} calls an oracle function"
    for so in (os.path.join(pkg, "libllama2_hip.so"), os.path.join(pkg, "libllama2_hip_test.so")):
        if os.path.exists(so):
            import subprocess
            needed = subprocess.run(["readelf", "-d", so], capture_output=True, text=True).stdout
            assert "liboracle" not in needed


def test_shard_plan_rows_and_scheme_b_widths(B, ck):
    """Host-side shard geometry (l2z_shard_plan, no device): the row / head ranges of scheme A tile the model exactly, and
    scheme B's column shards of Wo / W2 are padded to widths the vector mat-vecs take -- at least the shard, a multiple of 4,
    and of 256 floats (whole 64-lane sweeps of 16 bytes) above 768."""
    shapes = [dict(dim=4096, hidden_dim=11008, n_layers=2, n_heads=32, n_kv_heads=32, vocab_size=32000, seq_len=64),
              dict(dim=4096, hidden_dim=8192, n_layers=2, n_heads=32, n_kv_heads=8, vocab_size=8192, seq_len=64),
              dict(dim=768, hidden_dim=2048, n_layers=2, n_heads=12, n_kv_heads=12, vocab_size=32000, seq_len=64),
              dict(dim=36, hidden_dim=102, n_layers=1, n_heads=6, n_kv_heads=3, vocab_size=96, seq_len=16)]
    for kw in shapes:
        cfg = ck.Config(**kw)
        for world in (1, 2, 3, 4, 6, 8):
            if cfg.n_kv_heads % world or cfg.hidden_dim % world or cfg.vocab_size % world:
                with pytest.raises(B.L2ZError):
                    B.shard_plan(cfg, 0, world)
                continue
            plans = [B.shard_plan(cfg, r, world) for r in range(world)]
            assert [p["dim0"] for p in plans] == [r * cfg.dim // world for r in range(world)]
            assert sum(p["dim_loc"] for p in plans) == cfg.dim and sum(p["hid_loc"] for p in plans) == cfg.hidden_dim
            assert sum(p["v_loc"] for p in plans) == cfg.vocab_size and sum(p["kvd_loc"] for p in plans) == cfg.kv_dim
            assert all(p["heads_loc"] * cfg.head_size == p["dim_loc"] for p in plans)
            for p in plans:
                for loc, pad in ((p["dim_loc"], p["dimc_pad"]), (p["hid_loc"], p["hidc_pad"])):
                    assert loc <= pad < loc + 256 and pad % 4 == 0 and (pad <= 768 or pad % 256 == 0), (kw, world, loc, pad)
    assert B.shard_plan(ck.Config(**shapes[0]), 7, 8)["hidc_pad"] == 1536   # 1376 -> 1536


def test_shard_range(B):
    assert B.shard_range(4096, 128, 3, 8) == (1536, 2048)
    assert B.shard_range(11008, 1, 7, 8) == (9632, 11008)
    assert B.shard_range(32000, 1, 0, 1) == (0, 32000)
    with pytest.raises(B.L2ZError):
        B.shard_range(288, 48, 0, 4)   # 6 heads do not split over 4 ranks
    with pytest.raises(B.L2ZError):
        B.shard_range(10, 1, 5, 4)


def test_rope_function_is_the_pass_inline_rope(orc, ck):
    """orc_rope is transformer()'s inline RoPE (main.zig:336-351) factored out for the GPU test of the
    EPI_ROPE epilogue: identity at pos 0, equal to the float64 formula to f32 rounding elsewhere, k pairs
    beyond kv_dim untouched (GQA), and the pass's own q / K row at pos p equals rope(pos-0 q / K row)."""
    rng = np.random.default_rng(5)
    for dim, kv_dim, hs in ((96, 96, 48), (128, 32, 64), (256, 128, 128)):
        q, k = rng.standard_normal(dim).astype(np.float32), rng.standard_normal(kv_dim).astype(np.float32)
        q0, k0 = orc.rope(q, k, 0, hs)
        assert np.array_equal(q0, q) and np.array_equal(k0, k)
        for pos in (1, 17, 2047):
            qr, kr = orc.rope(q, k, pos, hs)
            tol = 3e-6 + 1e-6 * pos   # an ulp of freq (powf vs numpy's pow) is an angle error of pos * 6e-8
            i = np.arange(0, dim, 2)
            ang = pos * (1.0 / np.power(10000.0, (i % hs) / hs).astype(np.float32)).astype(np.float32)
            c, s_ = np.cos(ang.astype(np.float64)), np.sin(ang.astype(np.float64))
            wq = np.empty(dim); wq[0::2] = q[0::2] * c - q[1::2] * s_; wq[1::2] = q[0::2] * s_ + q[1::2] * c
            assert np.allclose(qr, wq, atol=tol)
            half = kv_dim // 2
            wk = np.empty(kv_dim); wk[0::2] = k[0::2] * c[:half] - k[1::2] * s_[:half]; wk[1::2] = k[0::2] * s_[:half] + k[1::2] * c[:half]
            assert np.allclose(kr, wk, atol=tol)
    cfg = ck.Config(dim=64, hidden_dim=32, n_layers=1, n_heads=4, n_kv_heads=2, vocab_size=32, seq_len=64)
    m = orc.Model(cfg.as_i32(), ck.synth_blob(cfg, True, seed=3), True)
    m.transformer(3, 0)
    q0, k0 = m.state("q", 64), m.state("key_cache", 32)
    m.transformer(3, 41)
    wq, wk = orc.rope(q0, k0, 41, 16)
    assert np.array_equal(m.state("q", 64), wq) and np.array_equal(m.state("key_cache", 64 * 32)[41 * 32:42 * 32], wk)
    m.close()
