"""First-contact kit for the two proofs the build image cannot give (no Zig 0.16 here):

* bindings/zig/{llama2_hip.zig, main_hip.patch, build_hip.patch} -- the reference with its
  forward pass replaced by this library -- builds and prints what `host/llama2` prints;
* the UNPATCHED reference (`zig build -Doptimize=ReleaseFast`, README.md:44) run at `-t 0`
  prints exactly the oracle's greedy token ids: the only thing that can move SURVEY.md 8(c)
  from "kernel-level pinned" to "pinned end to end".

Tests that need Zig skip when `zig version` is not 0.16.x; tests that need the reference's
sources take them from $L2Z_REFERENCE_DIR (default /root/reference) and skip when it is
absent (the GPU box has none unless the caller supplies it).  The text checks (patches apply,
the binding declares the header's functions with the header's arity) run everywhere."""
import ctypes as C
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ZIGDIR = os.path.join(ROOT, "bindings", "zig")
HOST = os.path.join(ROOT, "llama2.zig_amd", "host")
PKG = os.path.join(ROOT, "llama2.zig_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
TOK = os.path.join(GOLDEN, "tokenizer.bin")
REF = os.environ.get("L2Z_REFERENCE_DIR", "/root/reference")


def zig_exe():
    exe = os.environ.get("L2Z_ZIG", shutil.which("zig") or "")
    if not exe:
        pytest.skip("no zig on PATH (set L2Z_ZIG)")
    try:
        ver = subprocess.run([exe, "version"], capture_output=True, text=True, timeout=60).stdout.strip()
    except OSError as e:
        pytest.skip(f"zig not runnable: {e}")
    if not ver.startswith("0.16"):
        pytest.skip(f"zig {ver}: the reference needs 0.16 (build.zig.zon:5)")
    return exe


def reference_tree(tmp_path, patched: bool):
    if not os.path.isfile(os.path.join(REF, "src", "main.zig")):
        pytest.skip(f"no reference sources at {REF} (set L2Z_REFERENCE_DIR)")
    dst = tmp_path / ("ref_hip" if patched else "ref_plain")
    dst.mkdir()
    shutil.copytree(os.path.join(REF, "src"), dst / "src")
    for f in ("build.zig", "build.zig.zon"):
        shutil.copy(os.path.join(REF, f), dst / f)
    if patched:
        for p in ("main_hip.patch", "build_hip.patch"):
            r = subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", os.path.join(ZIGDIR, p)],
                               cwd=dst, capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
        shutil.copy(os.path.join(ZIGDIR, "llama2_hip.zig"), dst / "src" / "llama2_hip.zig")
    return dst


def zig_build(zig, tree, *extra):
    env = dict(os.environ, ZIG_GLOBAL_CACHE_DIR=str(tree / ".zig-global"), ZIG_LOCAL_CACHE_DIR=str(tree / ".zig-cache"))
    r = subprocess.run([zig, "build", "-Doptimize=ReleaseFast", *extra], cwd=tree, capture_output=True, text=True,
                       timeout=1800, env=env)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-4000:]
    exe = tree / "zig-out" / "bin" / "llama2"
    assert exe.exists()
    return str(exe)


class Detok:
    """stdout of main.zig:1020-1034 for a list of `next` tokens (libllama2_host.so holds the
    tokenizer; the formatting rule is restated here: no leading space right after BOS, raw bytes)."""

    def __init__(self, vocab_size):
        if not os.path.exists(os.path.join(HOST, "libllama2_host.so")):
            subprocess.check_call(["make", "-C", HOST, "-s"])
        self.H = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
        self.H.l2zh_tokenizer_open.restype = C.c_void_p
        self.H.l2zh_tokenizer_token.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_size_t]
        self.H.l2zh_tokenizer_token.restype = C.c_size_t
        err = C.create_string_buffer(256)
        self.t = self.H.l2zh_tokenizer_open(TOK.encode(), vocab_size, err, 256)
        assert self.t, err.value

    def text(self, nexts) -> bytes:
        out, token = b"", 1
        buf = C.create_string_buffer(256)
        for nx in nexts:
            if nx == 1:
                break
            n = self.H.l2zh_tokenizer_token(C.c_void_p(self.t), int(nx), buf, 256)
            s = buf.raw[:n]
            if token == 1 and s[:1] == b" ":
                s = s[1:]
            m = re.fullmatch(rb"<0x([0-9A-Fa-f]{2})>", s)
            out += bytes([int(m.group(1), 16)]) if m else s
            token = nx
        return out


# ---------------------------------------------------------------- text checks (run everywhere)

def header_functions():
    src = open(os.path.join(ROOT, "include", "llama2_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    fns = {}
    for m in re.finditer(r"\b(l2z_\w+)\s*\(([^;{]*?)\)\s*;", src):
        args = m.group(2).strip()
        fns[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return fns


def test_zig_binding_matches_the_header():
    """every `extern fn` of llama2_hip.zig is a function of include/llama2_hip.h with the same
    number of parameters, and the calls the patch makes are declared"""
    hdr = header_functions()
    z = open(os.path.join(ZIGDIR, "llama2_hip.zig")).read()
    ext = {}
    for m in re.finditer(r"pub extern fn (\w+)\(([^)]*)\)", z, flags=re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        ext[m.group(1)] = len(args)
    assert len(ext) >= 12
    for name, n in ext.items():
        assert name in hdr, f"{name} is not in include/llama2_hip.h"
        assert hdr[name] == n, f"{name}: header takes {hdr[name]} parameters, binding {n}"
    for need in ("l2z_weights_init", "l2z_runstate_init", "l2z_transformer", "l2z_argmax", "l2z_logits_read",
                 "l2z_weights_free", "l2z_runstate_free", "l2z_last_error"):
        assert need in ext
    m = re.search(r"#define L2Z_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "llama2_hip.h")).read())
    assert f"abi_version: c_int = {m.group(1)};" in z


def test_zig_patches_apply_to_the_reference(tmp_path):
    """main_hip.patch / build_hip.patch apply cleanly to the reference's sources and replace exactly
    the sites INTEGRATION.md names (main.zig:967, :974-975, :996, :1003, :1005-1012)"""
    tree = reference_tree(tmp_path, patched=True)
    main = (tree / "src" / "main.zig").read_text()
    assert 'const hip = @import("llama2_hip.zig");' in main
    body = main[main.index("pub fn main("):]
    assert "Weights.init(" not in body and "RunState.init(" not in body
    assert "try dev.transformer(token, pos);" in body and "next = try dev.argmax();" in body
    assert "state." not in body
    assert 'linkSystemLibrary("llama2_hip"' in (tree / "build.zig").read_text()


def test_detokenizer_round_trips_the_reference_bpe_vector():
    """the checker's own formatting rule, on the reference's `bpe` vector (main.zig:1164-1179): the ids of
    the sentence print the sentence (no leading space after BOS), raw-byte tokens print their byte"""
    d = Detok(32000)
    ids = [68, 767, 27116, 310, 266, 765, 338, 11584, 263, 1375, 13537, 4094, 11164, 66]
    assert d.text(ids) == b"A man dying of thirst is suddenly a mineral water critic?"
    assert d.text([3 + 0x41, 3 + 0x0A]) == b"A\n"      # <0x41>, <0x0A> (ids 3..258 are the raw bytes)
    assert d.text([767, 1, 767]) == b"man"               # BOS ends the text (main.zig:1017)


# ---------------------------------------------------------------- needs Zig 0.16

def greedy_cases(ck):
    """(name, checkpoint path or (cfg, shared, seed), prompt text, steps)"""
    cases = [("toy_gqa_unshared", os.path.join(GOLDEN, "toy_gqa_unshared.bin"), None, 24),
             ("toy_mha_shared", os.path.join(GOLDEN, "toy_mha_shared.bin"), "a b", 24)]
    real = os.environ.get("L2Z_STORIES15M")
    if real and os.path.exists(real):
        cases.append(("stories15M.bin", real, None, 256))
        cases.append(("stories15M.bin+prompt", real, "Once upon a time", 256))
    return cases


def write_synth_15m(ck, tmp_path):
    cfg = ck.Config(dim=288, hidden_dim=768, n_layers=6, n_heads=6, n_kv_heads=6, vocab_size=32000, seq_len=256)
    path = tmp_path / "synth15M.bin"
    ck.write_checkpoint(str(path), cfg, ck.synth_blob(cfg, True, seed=15), True)
    return str(path)


def oracle_tokens(ck, orc, path, prompt_text, steps):
    c, shared, blob = ck.read_checkpoint(path, mmap=False)
    prompt = []
    if prompt_text:
        H = C.CDLL(os.path.join(HOST, "libllama2_host.so"))
        H.l2zh_tokenizer_open.restype = C.c_void_p
        H.l2zh_tokenizer_encode.restype = C.c_long
        err = C.create_string_buffer(64)
        t = H.l2zh_tokenizer_open(TOK.encode(), c.vocab_size, err, 64)
        out = (C.c_int32 * 256)()
        n = H.l2zh_tokenizer_encode(C.c_void_p(t), prompt_text.encode(), len(prompt_text.encode()), out, 256)
        prompt = list(out[:n])
    m = orc.Model(c.as_i32(), blob, shared)
    toks, margins = m.generate_greedy(prompt, steps)
    m.close()
    return c, toks.tolist(), margins


def test_unpatched_reference_pins_the_oracle(tmp_path, ck, orc):
    """build the reference as it is and compare what it prints at -t 0 with the
    oracle's greedy token ids (detokenized by main.zig:1020-1034's rule).  Synthetic checkpoints in
    the reference's file layout need nothing but Zig; $L2Z_STORIES15M adds the real file."""
    zig = zig_exe()
    exe = zig_build(zig, reference_tree(tmp_path, patched=False))
    cases = greedy_cases(ck) + [("synth15M", write_synth_15m(ck, tmp_path), None, 256)]
    for name, path, prompt_text, steps in cases:
        c, toks, margins = oracle_tokens(ck, orc, path, prompt_text, steps)
        args = [exe, path, "-t", "0", "-n", str(steps), "-s", "1", "-z", TOK]
        if prompt_text:
            args += ["-i", prompt_text]
        r = subprocess.run(args, capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode(errors="replace")
        want = Detok(c.vocab_size).text(toks)
        assert r.stdout == want, (f"{name}: the Zig binary and the oracle part ways (smallest top-1/top-2 margin "
                                  f"{float(np.min(margins)):.3e}: a near-tie if ~1e-6, else a bug)")


@pytest.mark.gpu
def test_zig_shim_prints_what_the_host_cli_prints(tmp_path, gpu, ck, orc):
    """the reference's own main() with the forward pass behind include/llama2_hip.h (the patch) against
    this repository's host driver and the oracle, greedy and sampled (same seed => same tokens needs
    Zig's own xoshiro256++, which the patched binary still uses: compare greedy text, sampled exit code)"""
    zig = zig_exe()
    exe = zig_build(zig, reference_tree(tmp_path, patched=True), f"-Dhip-lib-dir={PKG}")
    ours = os.path.join(HOST, "llama2")
    cases = greedy_cases(ck) + [("synth15M", write_synth_15m(ck, tmp_path), None, 256)]
    for name, path, prompt_text, steps in cases:
        c, toks, _ = oracle_tokens(ck, orc, path, prompt_text, steps)
        tail = [path, "-t", "0", "-n", str(steps), "-z", TOK] + (["-i", prompt_text] if prompt_text else [])
        a = subprocess.run([exe] + tail, capture_output=True, timeout=600)
        b = subprocess.run([ours] + tail, capture_output=True, timeout=600)
        assert a.returncode == 0, a.stderr.decode(errors="replace")
        assert b.returncode == 0, b.stderr.decode(errors="replace")
        assert a.stdout == b.stdout == Detok(c.vocab_size).text(toks), name
    r = subprocess.run([exe, cases[0][1], "-t", "1.0", "-p", "0.9", "-s", "7", "-n", "16", "-z", TOK],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")
