"""Host side above the C ABI (llama2.zig_amd/host): tokenizer against the reference's
`bpe` test vectors (src/main.zig:1152-1180, fixture tests/golden/tokenizer.bin is the
reference's own data file), samplers, raw-byte formatting, CLI flag handling; and, on
the GPU, the `llama2` binary end to end against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "llama2.zig_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOK = os.path.join(GOLDEN, "tokenizer.bin")


@pytest.fixture(scope="module")
def H(B):  # B builds everything if needed
    if not os.path.exists(os.path.join(HOST, "libllama2_host.so")):
        subprocess.check_call(["make", "-C", HOST, "-s"])
    L = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
    L.l2zh_tokenizer_open.restype = C.c_void_p
    L.l2zh_tokenizer_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.l2zh_tokenizer_close.argtypes = [C.c_void_p]
    L.l2zh_tokenizer_lookup.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    L.l2zh_tokenizer_max_token_len.argtypes = [C.c_void_p]
    L.l2zh_tokenizer_max_token_len.restype = C.c_uint32
    L.l2zh_tokenizer_token.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
    L.l2zh_tokenizer_token.restype = C.c_size_t
    L.l2zh_tokenizer_encode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.c_size_t]
    L.l2zh_tokenizer_encode.restype = C.c_long
    L.l2zh_tokenizer_encode_quadratic.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int32), C.c_size_t]
    L.l2zh_tokenizer_encode_quadratic.restype = C.c_long
    L.l2zh_is_raw_byte.argtypes = [C.c_char_p, C.c_size_t]
    L.l2zh_prng_floats.argtypes = [C.c_uint64, C.POINTER(C.c_float), C.c_size_t]
    L.l2zh_prng_u64.argtypes = [C.c_uint64, C.c_size_t]
    L.l2zh_prng_u64.restype = C.c_uint64
    L.l2zh_sample.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_uint64]
    L.l2zh_sample.restype = C.c_size_t
    L.l2zh_sample_top_p.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.c_float, C.c_uint64]
    L.l2zh_sample_top_p.restype = C.c_size_t
    L.l2zh_softmax.argtypes = [C.POINTER(C.c_float), C.c_size_t]
    return L


@pytest.fixture(scope="module")
def tok(H):
    err = C.create_string_buffer(256)
    t = H.l2zh_tokenizer_open(TOK.encode(), 32000, err, 256)
    assert t, err.value
    yield t
    H.l2zh_tokenizer_close(t)


def encode(H, tok, text: str):
    b = text.encode("utf-8")
    out = (C.c_int32 * (len(b) + 1))()
    n = H.l2zh_tokenizer_encode(tok, b, len(b), out, len(b) + 1)
    return None if n < 0 else list(out[:n])


def test_bpe_reference_vectors(H, tok):
    """src/main.zig:1152-1180, every assertion of the reference's `bpe` test"""
    ae = "æ".encode()
    assert H.l2zh_tokenizer_lookup(tok, ae, len(ae)) == 233
    buf = C.create_string_buffer(64)
    assert H.l2zh_tokenizer_token(tok, 100, buf, 64) == 1 and buf.raw[:1] == b"a"
    assert H.l2zh_tokenizer_max_token_len(tok) == 27
    assert H.l2zh_tokenizer_lookup(tok, b"a", 1) == 100
    assert encode(H, tok, "A man dying of thirst is suddenly a mineral water critic?") == \
        [68, 767, 27116, 310, 266, 765, 338, 11584, 263, 1375, 13537, 4094, 11164, 66]
    assert encode(H, tok, "中") == [30275]


def test_bpe_fast_form_equals_the_reference_merge_loop(H, tok):
    """encode() runs the reference's greedy merges (main.zig:247-278: best score, leftmost among equals)
    off a heap in O(n log n); encode_quadratic() is that loop as written.  Same token ids on English,
    repeated and random-word text, multi-byte characters, runs of spaces and of one letter (every pair
    ties), and a 6000-character prompt -- where the loop as written needs about a second."""
    import time
    rng = np.random.default_rng(12)
    words = ["the", "a", "Once", "upon", "time", "there", "was", "little", "girl", "named", "Lily", "mineral", "water",
             "critic", "æther", "naïve", "中文", "日本語", "🙂", "x", "zzzz", "aaaaaaaa", "  ", "\n", "don't", "1234567890"]
    words = [w for w in words if encode(H, tok, w) is not None]   # code points the vocabulary has (no raw newline, no emoji)
    assert len(words) >= 20
    texts = ["", "a", "aa", "aaa", "a" * 64, " " * 33, "abababababababab", "A man dying of thirst is suddenly a mineral water critic?"]
    for _ in range(40):
        k = int(rng.integers(1, 60))
        texts.append(" ".join(words[int(i)] for i in rng.integers(0, len(words), k)))
    long_text = " ".join(words[int(i)] for i in rng.integers(0, len(words), 1300))[:6000].rsplit(" ", 1)[0]
    texts.append(long_text)
    for t in texts:
        b = t.encode("utf-8")
        o1, o2 = (C.c_int32 * (len(b) + 1))(), (C.c_int32 * (len(b) + 1))()
        n1 = H.l2zh_tokenizer_encode(tok, b, len(b), o1, len(b) + 1)
        n2 = H.l2zh_tokenizer_encode_quadratic(tok, b, len(b), o2, len(b) + 1)
        assert n1 == n2 >= 0 and list(o1[:n1]) == list(o2[:n2]), t[:60]
    b = long_text.encode("utf-8")
    o = (C.c_int32 * (len(b) + 1))()
    t0 = time.perf_counter()
    n = H.l2zh_tokenizer_encode(tok, b, len(b), o, len(b) + 1)
    assert n > 1000 and time.perf_counter() - t0 < 0.25   # the loop as written: ~1 s


def test_tokenizer_edges(H, tok):
    assert encode(H, tok, "") == []
    assert encode(H, tok, "a") == [100]
    b = b"\xff"  # invalid UTF-8 start byte -> error like std.unicode
    out = (C.c_int32 * 4)()
    assert H.l2zh_tokenizer_encode(tok, b, 1, out, 4) == -1
    err = C.create_string_buffer(256)
    assert not H.l2zh_tokenizer_open(b"/nonexistent/tokenizer.bin", 10, err, 256)
    assert b"cannot open" in err.value
    # truncated vocabulary (main.zig:173 reads only vocab_size entries)
    t = H.l2zh_tokenizer_open(TOK.encode(), 512, err, 256)
    assert t and H.l2zh_tokenizer_lookup(t, b"a", 1) == 100
    H.l2zh_tokenizer_close(t)


def test_is_raw_byte(H):
    """src/main.zig:1055-1076"""
    f = lambda s: H.l2zh_is_raw_byte(s, len(s))
    assert f(b"<0x41>") == 0x41 and f(b"<0x0A>") == 10 and f(b"<0x0a>") == 10
    assert f(b"<0x00>") == -1 and f(b"<0x7F>") == -1     # not printable, not whitespace
    assert f(b"<0xZZ>") == -1 and f(b"<0x4>") == -1 and f(b"hello!") == -1


def test_prng_is_xoshiro256pp(H):
    """Independent Python restatement of SplitMix64 -> Xoshiro256++ (Zig DefaultPrng)."""
    M = (1 << 64) - 1

    def stream(seed, n):
        s, sm = [], seed
        for _ in range(4):
            sm = (sm + 0x9E3779B97F4A7C15) & M
            z = sm
            z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
            z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
            s.append(z ^ (z >> 31))
        rot = lambda x, k: ((x << k) | (x >> (64 - k))) & M
        out = []
        for _ in range(n):
            out.append((rot((s[0] + s[3]) & M, 23) + s[0]) & M)
            t = (s[1] << 17) & M
            s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rot(s[3], 45)
        return out

    for seed in (0, 1, 42, 2**63 + 5):
        exp = stream(seed, 5)
        assert [H.l2zh_prng_u64(seed, k) for k in range(5)] == exp
    fl = (C.c_float * 10000)()
    H.l2zh_prng_floats(7, fl, 10000)
    a = np.array(fl[:])
    assert a.min() >= 0.0 and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.02


def test_samplers(H):
    """main.zig:728-798: cdf sampling, nucleus sampling restricted to the top-p set"""
    p = np.array([0.05, 0.6, 0.05, 0.3], np.float32)
    pp = p.ctypes.data_as(C.POINTER(C.c_float))
    draws = [H.l2zh_sample(pp, 4, s) for s in range(400)]
    assert set(draws) <= {0, 1, 2, 3} and 180 < draws.count(1) < 300
    nuc = [H.l2zh_sample_top_p(pp, 4, 0.8, s) for s in range(400)]
    assert set(nuc) <= {1, 3}           # 0.6 + 0.3 > 0.8: tokens 0 and 2 can never be drawn
    one = np.array([0, 0, 1, 0], np.float32)
    assert H.l2zh_sample(one.ctypes.data_as(C.POINTER(C.c_float)), 4, 3) == 2
    x = np.array([1, 2, 3, 4], np.float32)
    H.l2zh_softmax(x.ctypes.data_as(C.POINTER(C.c_float)), 4)
    assert abs(float(x.sum()) - 1) < 1e-6 and np.all(np.diff(x) > 0)


def _top_p_reference(p, top_p, r01):
    """main.zig:745-798 with the candidates ordered by (probability descending, token id ascending) --
    the total order the host sampler documents -- and float32 running sums."""
    n = len(p)
    cutoff = np.float32((np.float32(1.0) - np.float32(top_p)) / (np.float32(n) - np.float32(1.0)))
    idx = np.nonzero(p >= cutoff)[0]
    order = idx[np.lexsort((idx, -p[idx].astype(np.float64)))]
    cum = np.float32(0.0)
    last = len(order) - 1
    for i, t in enumerate(order):
        cum = np.float32(cum + p[t])
        if cum > np.float32(top_p):
            last = i
            break
    r = np.float32(np.float32(r01) * cum)
    cdf = np.float32(0.0)
    for t in order[:last + 1]:
        cdf = np.float32(cdf + p[t])
        if r < cdf:
            return int(t)
    return int(order[last])


def test_top_p_large_vocabulary_matches_the_sorted_reference(H):
    """The nucleus sampler over a whole 32000-token vocabulary (the reference's default -p 0.9 runs it for
    every generated token): the radix sort behind it must give the permutation of the documented total
    order -- near-uniform probabilities with MANY exact ties, a peaked distribution (few candidates: the
    comparison-sort path), and an exactly uniform one."""
    rng = np.random.default_rng(11)
    V = 32000
    flat = np.round(rng.uniform(1.0, 2.0, V) * 64) / 64          # 64 distinct values: ties everywhere
    peaked = np.exp(rng.standard_normal(V) * 4.0)
    cases = [flat, peaked, np.ones(V), np.concatenate([np.full(2000, 3.0), rng.uniform(0.0, 1e-9, V - 2000)])]
    fl = (C.c_float * 1)()
    for ci, w in enumerate(cases):
        p = (w / w.sum()).astype(np.float32)
        pp = p.ctypes.data_as(C.POINTER(C.c_float))
        for top_p in (0.9, 0.5, 0.999):
            for seed in range(12):
                H.l2zh_prng_floats(seed, fl, 1)
                want = _top_p_reference(p, top_p, fl[0])
                got = H.l2zh_sample_top_p(pp, V, C.c_float(top_p), seed)
                assert got == want, (ci, top_p, seed, got, want)


def test_cli_usage_and_errors():
    exe = os.path.join(HOST, "llama2")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Usage:   llama2 <checkpoint> [options]")
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert "-z, --tokenizer <path>" in r.stdout
    r = subprocess.run([exe, "a.bin", "b.bin"], capture_output=True, text=True)
    assert r.returncode == 1 and "multiple checkpoint paths" in r.stderr          # main.zig:857
    r = subprocess.run([exe, "a.bin", "-t"], capture_output=True, text=True)
    assert r.returncode == 1 and "missing argument for temperature" in r.stderr   # main.zig:865
    r = subprocess.run([exe, "a.bin", "-n", "abc"], capture_output=True, text=True)
    assert r.returncode == 1 and "unable to parse --seq-len" in r.stderr
    r = subprocess.run([exe, "a.bin", "--bogus"], capture_output=True, text=True)
    assert "unknown argument '--bogus'" in r.stderr and r.stdout.startswith("Usage")  # :930
    r = subprocess.run([exe, "/nonexistent.bin"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open checkpoint" in r.stderr


@pytest.mark.gpu
def test_cli_greedy_matches_oracle(gpu, ck, orc, tmp_path):
    """`llama2 ckpt -t 0 -n N [-i prompt]` prints the oracle's greedy tokens."""
    exe = os.path.join(HOST, "llama2")
    ckpt = os.path.join(GOLDEN, "toy_gqa_unshared.bin")
    c, shared, blob = ck.read_checkpoint(ckpt, mmap=False)
    H = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
    for prompt_text in (None, "a b"):
        args = [exe, ckpt, "-t", "0", "-n", "24", "-z", TOK, "-v", "--tokens"]
        prompt = []
        if prompt_text:
            args += ["-i", prompt_text]
            err = C.create_string_buffer(64)
            H.l2zh_tokenizer_open.restype = C.c_void_p
            t = H.l2zh_tokenizer_open(TOK.encode(), c.vocab_size, err, 64)
            out = (C.c_int32 * 16)()
            H.l2zh_tokenizer_encode.restype = C.c_long
            n = H.l2zh_tokenizer_encode(C.c_void_p(t), prompt_text.encode(), len(prompt_text), out, 16)
            prompt = list(out[:n])
            assert all(p < c.vocab_size for p in prompt)
        r = subprocess.run(args, capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        line = [l for l in r.stderr.decode(errors="replace").splitlines() if l.startswith("tokens:")][0]
        got = [int(v) for v in line.split()[1:]]
        m = orc.Model(c.as_i32(), blob, shared)
        ref, _ = m.generate_greedy(prompt, 24)
        assert got == ref.tolist()
        assert b"tokens per second" in r.stderr
        m.close()


@pytest.mark.gpu
def test_cli_sampling_runs(gpu):
    """-t 1.0 -p 0.9 (BASELINE config 3 flags): the sampled path runs and is seed-deterministic."""
    exe = os.path.join(HOST, "llama2")
    ckpt = os.path.join(GOLDEN, "toy_mha_shared.bin")
    outs = []
    for _ in range(2):
        r = subprocess.run([exe, ckpt, "-t", "1.0", "-p", "0.9", "-n", "16", "-s", "123", "-z", TOK,
                            "--tokens"], capture_output=True, timeout=120)
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        outs.append([l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0])
    assert outs[0] == outs[1] and len(outs[0].split()) > 2


@pytest.mark.gpu
def test_cli_sampling_with_long_prompt_uses_batched_prefill(gpu, ck):
    """-t 1.0 with a prompt of >= 4 tokens: the prompt positions go through l2z_prefill, the
    output starts with the prompt's tokens, and the run is seed-deterministic; with L2Z_PREFILL=0
    (stepped prompt) the echoed prompt is the same."""
    exe = os.path.join(HOST, "llama2")
    ckpt = os.path.join(GOLDEN, "toy_mha_shared.bin")
    c, _, _ = ck.read_checkpoint(ckpt, mmap=False)
    text = "a b c d e f g h"
    H = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
    err = C.create_string_buffer(64)
    H.l2zh_tokenizer_open.restype = C.c_void_p
    tk = H.l2zh_tokenizer_open(TOK.encode(), c.vocab_size, err, 64)
    out = (C.c_int32 * 64)()
    H.l2zh_tokenizer_encode.restype = C.c_long
    n = H.l2zh_tokenizer_encode(C.c_void_p(tk), text.encode(), len(text), out, 64)
    prompt = list(out[:n])
    assert n >= 4 and 1 not in prompt

    def run(env_extra):
        r = subprocess.run([exe, ckpt, "-t", "1.0", "-p", "0.9", "-n", "28", "-s", "7", "-z", TOK, "-i", text,
                            "--tokens"], capture_output=True, timeout=120, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        line = [l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0]
        return [int(v) for v in line.split()[1:]]

    a, b, stepped = run({}), run({}), run({"L2Z_PREFILL": "0"})
    assert a == b and a[:n] == prompt and stepped[:n] == prompt and len(a) > n


@pytest.mark.gpu
@pytest.mark.parametrize("n_gpus", [2, 4])
def test_cli_multi_gpu_mode_matches_single(gpu, n_gpus):
    """`-g N`: one process per rank (forked by the CLI), each rank mmaps the checkpoint and uploads
    only its rows, peer-write gathers over IPC; greedy (-t 0) and sampled (-t 1.0, fixed seed) token
    ids equal the single-GPU run's.  The ranks share this box's one GPU (rank % device count)."""
    exe = os.path.join(HOST, "llama2")
    ckpt = os.path.join(GOLDEN, "toy_gqa_unshared.bin")  # 4 heads, 2 kv heads: 2 ranks; hidden 172 = 4 * 43
    if n_gpus == 4:
        ckpt = os.path.join(GOLDEN, "toy_mha_shared.bin")  # 4 heads, 4 kv heads, hidden 128, vocab 300
    env = dict(os.environ, L2Z_GRID_CAP=str(max(32, 512 // n_gpus)), L2Z_P2P_TIMEOUT_S="30", L2Z_FUSE_SMALL="0")
    for flags in (["-t", "0"], ["-t", "1.0", "-p", "0.9", "-s", "99"]):
        outs = []
        for g in (1, n_gpus):
            r = subprocess.run([exe, ckpt, *flags, "-n", "20", "-z", TOK, "-i", "a b", "--tokens", "-g", str(g)],
                               capture_output=True, timeout=180, env=env)
            assert r.returncode == 0, r.stderr.decode(errors="replace")
            outs.append(([l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0], r.stdout))
        assert outs[0] == outs[1], flags


@pytest.mark.gpu
@pytest.mark.parametrize("n_gpus", [2, 4])
def test_cli_multi_gpu_scheme_b(gpu, n_gpus):
    """`-g N` under L2Z_SCHEME_B=1 (Wo / W2 sharded by columns, two all-reduces per layer, csrc/forward.cpp): the ranks
    hold bit-identical logits among themselves -- which is what keeps the CLI's ranks in step, greedy and sampled -- and
    logits within ~1e-6 of the single-GPU pass; these toys' argmax margins are far wider, so the greedy tokens are the
    single-GPU run's."""
    exe = os.path.join(HOST, "llama2")
    ckpt = os.path.join(GOLDEN, "toy_gqa_unshared.bin" if n_gpus == 2 else "toy_mha_shared.bin")
    env = dict(os.environ, L2Z_GRID_CAP=str(max(32, 512 // n_gpus)), L2Z_P2P_TIMEOUT_S="30", L2Z_FUSE_SMALL="0")
    outs = []
    for g, extra in ((1, {}), (n_gpus, {"L2Z_SCHEME_B": "1"})):
        r = subprocess.run([exe, ckpt, "-t", "0", "-n", "20", "-z", TOK, "-i", "a b", "--tokens", "-g", str(g)],
                           capture_output=True, timeout=180, env=dict(env, **extra))
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        outs.append(([l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0], r.stdout))
    assert outs[0] == outs[1]
    # sampled: every rank draws from its own copy of the distribution with the same seed; they finish together
    r = subprocess.run([exe, ckpt, "-t", "1.0", "-p", "0.9", "-s", "99", "-n", "20", "-z", TOK, "-i", "a b", "--tokens", "-g", str(n_gpus)],
                       capture_output=True, timeout=180, env=dict(env, L2Z_SCHEME_B="1"))
    assert r.returncode == 0, r.stderr.decode(errors="replace")


@pytest.mark.gpu
def test_cli_multi_gpu_mode_ends_cleanly_on_bos(gpu, ck, orc, tmp_path):
    """`-g 2 -t 0` on a model whose greedy sequence ends with BOS (main.zig:1017) in the middle of a
    device call: the ranks share no control plane, so they only stay in step if every rank asks for the
    same number of positions per call (the step is a function of the model and the rank count, not of a
    rank's own clock -- ADVICE r2).  The run must print the oracle's tokens, exit 0 and not sit in a
    gather timeout."""
    import time
    exe = os.path.join(HOST, "llama2")
    cfg = ck.Config(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=64, seq_len=128)
    blob = ck.synth_blob(cfg, False, 7)   # seed found with the oracle: BOS is the 71st greedy token
    m = orc.Model(cfg.as_i32(), blob, False)
    ref, margins = m.generate_greedy([], 128)
    m.close()
    assert len(ref) < 128 and ref[-1] == 1 and len(ref) > 20, "the seed no longer ends on BOS"
    path = str(tmp_path / "bos.bin")
    ck.write_checkpoint(path, cfg, blob, False)
    env = dict(os.environ, L2Z_P2P_TIMEOUT_S="20", L2Z_FUSE_SMALL="0")
    env.pop("L2Z_GRID_CAP", None)   # the CLI sets the cap itself when ranks share a GPU
    for g in (1, 2):
        t0 = time.time()
        r = subprocess.run([exe, path, "-t", "0", "-n", "128", "-z", TOK, "--tokens", "-g", str(g)],
                           capture_output=True, timeout=180, env=env)
        took = time.time() - t0
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        line = [l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0]
        got = [int(v) for v in line.split()[1:]]
        assert got == ref.tolist(), (g, got, ref.tolist(), float(margins.min()))
        assert took < 15.0, f"-g {g} needed {took:.1f} s: a rank sat in a gather timeout"


def _cli_tokens(r):
    return [int(v) for v in [l for l in r.stderr.decode().splitlines() if l.startswith("tokens:")][0].split()[1:]]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["toy", "stories15M-2layers"])
def test_cli_sampled_tokens_equal_host_replay(gpu, ck, orc, tmp_path, shape):
    """BASELINE config 3's flags (-t 1.0 -p 0.9) as a COMPARISON, not a property: the CLI's token ids
    against a host replay of main.zig:996-1012 -- the ORACLE's logits -> logits / temperature ->
    softmax (:1006-1008) -> sample_top_p (:1011) with the same seed through the same host sampler
    (libllama2_host.so).  Both the device softmax (l2z_probs_read, default) and the host softmax
    (L2Z_HOST_SOFTMAX=1) must give the replay's ids; a divergence is only accepted where the drawn
    number lies within 2e-6 of a boundary of the cumulative distribution (the two sides differ by
    ~1e-6 in the logits and by ulps in the probabilities), and is reported."""
    exe = os.path.join(HOST, "llama2")
    H = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
    if shape == "toy":
        cfg, shared, seed_w = ck.Config(dim=64, hidden_dim=172, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=512, seq_len=96), False, 5
    else:
        cfg, shared, seed_w = ck.Config(dim=288, hidden_dim=768, n_layers=2, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=256), True, 15
    blob = ck.synth_blob(cfg, shared, seed_w)
    path = str(tmp_path / "m.bin")
    ck.write_checkpoint(path, cfg, blob, shared)
    n, temp, top_p, seed = 64, 1.0, 0.9, 4242
    # the replay: ONE generator across the positions, as the loop holds it (main.zig:845 / :926)
    H.l2zh_prng_open.restype = C.c_void_p
    H.l2zh_sample_top_p_rng.restype = C.c_size_t
    rng = C.c_void_p(H.l2zh_prng_open(C.c_uint64(seed)))
    m = orc.Model(cfg.as_i32(), blob, shared)
    tok, want, margins = 1, [], []
    for pos in range(n):
        lg = m.transformer(tok, pos)
        pp = np.ascontiguousarray(lg / np.float32(temp), np.float32)      # :1006
        H.l2zh_softmax(pp.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(pp.size))   # :1008
        mg = C.c_float(0)
        nxt = H.l2zh_sample_top_p_rng(pp.ctypes.data_as(C.POINTER(C.c_float)), C.c_size_t(cfg.vocab_size),
                                      C.c_float(top_p), rng, C.byref(mg))  # :1011
        margins.append(float(mg.value))
        want.append(int(nxt))
        if nxt == 1:
            break
        tok = int(nxt)
    H.l2zh_prng_close(rng)
    m.close()
    for env_extra in ({}, {"L2Z_HOST_SOFTMAX": "1"}):
        r = subprocess.run([exe, path, "-t", str(temp), "-p", str(top_p), "-n", str(n), "-s", str(seed), "-z", TOK,
                            "--tokens"], capture_output=True, timeout=180, env=dict(os.environ, **env_extra))
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        got = _cli_tokens(r)
        k = next((i for i, (a, b) in enumerate(zip(got, want)) if a != b), min(len(got), len(want)))
        if k < min(len(got), len(want)):
            print(f"{shape} {env_extra}: first divergence at pos {k} (cli {got[k]} vs replay {want[k]}), "
                  f"cdf margin there {margins[k]:.3e}")
            assert margins[k] < 2e-6, (shape, env_extra, k, got[k], want[k], margins[k])
        else:
            assert len(got) == len(want)
            print(f"{shape} {env_extra}: {len(got)} sampled token ids identical to the host replay; "
                  f"min cdf margin {min(margins):.3e}")
