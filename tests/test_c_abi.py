"""The drop-in boundary from plain C: include/llama2_hip.h must compile as strict C99 and a C host
must be able to run the whole hot path through it ("nothing proves the header is valid
C").  The reference's only caller is Zig (src/main.zig:996); Zig's `extern fn` binds the same C
symbols this program links against."""
import json
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "c_abi", "abi_smoke.c")
GOLD = os.path.join(HERE, "golden")


def build_smoke(tmp_path, B):
    exe = str(tmp_path / "abi_smoke")
    lib_dir = os.path.dirname(B.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           SRC, "-o", exe, "-L", lib_dir, "-lllama2_hip", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_strict_c99_and_config_layout_matches_the_reference(B, tmp_path):
    """sizeof(l2z_config) == 28, fields at 0,4,...,24: src/main.zig:17-25 ConfigReader."""
    # the header alone, as a translation unit
    tu = tmp_path / "hdr.c"
    tu.write_text('#include "llama2_hip.h"\n#include "llama2_hip_test.h"\nint main(void) { return 0; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I",
                    os.path.join(ROOT, "include"), str(tu)], check=True, capture_output=True, text=True)
    out = subprocess.run([build_smoke(tmp_path, B), "layout"], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    assert got.pop("abi") == str(B.lib().l2z_abi_version())
    assert {k: int(v) for k, v in got.items()} == {
        "sizeof": 28, "dim": 0, "hidden_dim": 4, "n_layers": 8, "n_heads": 12, "n_kv_heads": 16,
        "vocab_size": 20, "seq_len": 24}


def test_product_and_test_headers_do_not_overlap(B):
    prod, test = set(B.declared_symbols("product")), set(B.declared_symbols("test"))
    assert not (prod & test)
    # nothing a drop-in host needs lives in the test header, and vice versa
    assert {"l2z_weights_init", "l2z_runstate_init", "l2z_transformer", "l2z_argmax", "l2z_logits_read",
            "l2z_greedy_run", "l2z_prefill", "l2z_comm_init"} <= prod
    assert {"l2z_emu_transformer", "l2z_comm_init_emulated", "l2z_weights_read", "l2z_runstate_read",
            "l2z_matmul", "l2z_option_set"} <= test


def exported_l2z(so):
    out = subprocess.run(["nm", "-D", "--defined-only", so], check=True, capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("l2z_")}


def test_product_library_exports_the_boundary_and_nothing_else(B):
    """round 6: `nm -D libllama2_hip.so | grep l2z_` is EXACTLY include/llama2_hip.h (csrc/llama2_hip.map) -- the test and
    measurement entry points (emulated ranks, solo connect, kernel hooks, l2z_option_set, ...) are not in the library a
    host links; libllama2_hip_test.so, the same objects, exports both headers and is what tests / bench load."""
    prod, test = set(B.declared_symbols("product")), set(B.declared_symbols("test"))
    assert exported_l2z(B.PRODUCT_LIB_PATH) == prod
    assert exported_l2z(os.path.join(os.path.dirname(B.PRODUCT_LIB_PATH), "libllama2_hip_test.so")) >= prod | test
    # the version script is the header: a function added to one and not the other fails here before it fails a link
    mp = open(os.path.join(ROOT, "llama2.zig_amd", "csrc", "llama2_hip.map")).read()
    import re
    assert set(re.findall(r"^\s+(l2z_\w+);", mp, flags=re.M)) == prod
    # and nothing but the C ABI leaks out of the product library (no C++ symbols of namespace l2z)
    out = subprocess.run(["nm", "-D", "--defined-only", B.PRODUCT_LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert not [ln for ln in out.splitlines() if "l2z" in ln and not ln.split()[-1].startswith("l2z_")], out[:2000]


@pytest.mark.gpu
def test_c_host_runs_the_golden_toy_checkpoints(gpu, tmp_path):
    """init -> l2z_transformer -> l2z_argmax -> free from C on the committed toy checkpoints:
    the printed token ids equal the golden greedy tokens (tests/golden/*.npz)."""
    exe = build_smoke(tmp_path, gpu)
    meta = json.load(open(os.path.join(GOLD, "toy_models.json")))
    for m in meta["models"]:
        exp = np.load(os.path.join(GOLD, m["expected"]))["tokens"]
        n = m["config"]["seq_len"]
        out = subprocess.run([exe, "run", os.path.join(GOLD, m["checkpoint"]), str(n)] +
                             [str(t) for t in m["prompt"]], check=True, capture_output=True, text=True).stdout
        got = np.array([int(x) for x in out.split()], np.int32)
        assert np.array_equal(got, exp[:len(got)]) and len(got) == len(exp), (m["checkpoint"], got, exp)


def test_prefill_planning_is_pinned(B, ck):
    """Host logic behind the batched prefill, no device needed (the library loads on a CPU-only machine):
    a prompt is cut into 1024-token chunks while that many tokens remain, then 512, then the rest; the
    direct-to-LDS GEMM's output tile follows the grid (256 CUs assumed without a device).  The tile never
    changes a bit of the result (GPU tests), but these choices are what DESIGN.md 4.5 quotes timings for."""
    assert B.prefill_plan(0) == [] and B.prefill_plan(5) == [5] and B.prefill_plan(512) == [512]
    assert B.prefill_plan(600) == [512, 88] and B.prefill_plan(1024) == [1024]
    assert B.prefill_plan(1500) == [1024, 476] and B.prefill_plan(2047) == [1024, 512, 511]
    # a model whose matrices take the K-range panel kernel (they stream from HBM): a tail of 65 ... 96 tokens is cut in
    # two chunks of that kernel's range; small models and other lengths keep the plain plan
    c7b, c110 = ck.Config(4096, 11008, 32, 32, 32, 32000, 2048), ck.Config(768, 2048, 12, 12, 12, 32000, 1024)
    assert [B.prefill_plan(n, c7b) for n in (64, 65, 80, 81, 96, 97, 600)] == [[64], [65], [80], [81], [96], [97], [512, 88]]   # (round 6: the panel kernel takes up to 96 tokens itself)
    # ... and a tail past a step of the tile GEMM's cost staircase gives the step's worth to the tile forms and the rest to the
    # short-chunk kernels: on the bf16 cores 129 ... 224 tokens = 128 + rest (<= 96), 257 ... 384 = 256 + rest (<= 128)
    assert [B.prefill_plan(n, c7b) for n in (128, 129, 160, 161, 224, 225, 256, 257, 288, 289, 384, 385, 650)] == \
        [[128], [128, 1], [128, 32], [128, 33], [128, 96], [225], [256], [256, 1], [256, 32], [256, 33], [256, 128], [385], [512, 128, 10]]
    # ... and 673 ... 1023 tokens are ONE chunk there (512 first only where the rest is cheap); 1500 tokens: 1024 + 476
    assert [B.prefill_plan(n, c7b) for n in (672, 673, 900, 1023, 1500, 1900)] == [[512, 128, 32], [673], [900], [1023], [1024, 476], [1024, 876]]
    assert [B.prefill_plan(n, c110) for n in (65, 96, 140, 600)] == [[65], [96], [140], [512, 88]]
    want = {(4096, 512, False): "128x64",    # 7B q / k / v / wo / W2: one 128 x 64 tile per CU
            (4096, 1024, False): "128x128",  # a 1024-token chunk: fewer bytes per flop into the CU
            (4096, 256, False): "64x64", (4096, 128, False): "32x64", (4096, 80, False): "32x64",
            (11008, 512, True): "128x64",    # W1 | W3 paired: 128 tokens x (64 + 64) features
            (5504, 512, True): "64x64",      # the same on a 2-rank row shard: 2.7 waves of the larger tile -> 3 of the smaller
            (768, 256, False): "32x32", (512, 512, False): "32x32"}   # stories110M; a 7B row shard of 8 ranks
    for (n, p, pair), form in want.items():
        assert B.prefill_tile(n, p, pair) == form, (n, p, pair)
    # the split-K family (K ranges per output tile; the partials are added in range order, so this is part of
    # the arithmetic and a function of the WHOLE model's matrix): matrices that stream from HBM take 4 ranges at
    # 49-64 tokens, the block-starved ones (wo, W2: N = 4096) 2 at 65-128; nothing else, and never a matrix
    # that stays in the on-die caches
    sk = {(4096, 64, 4096): 4, (4096, 49, 4096): 4, (4096, 48, 4096): 1, (12288, 60, 4096): 4, (11008, 64, 4096): 4,
          (4096, 100, 4096): 2, (4096, 128, 11008): 2, (12288, 128, 4096): 1, (11008, 128, 4096): 1,
          (4096, 129, 4096): 1, (4096, 256, 4096): 1, (4096, 512, 4096): 1,
          (768, 64, 768): 1, (2048, 100, 768): 1,          # stories110M: cache resident
          (4096, 64, 4160): 1}                             # K not a multiple of 64 x ranges: fewer ranges
    for (n, p, k), want_sk in sk.items():
        assert B.prefill_split_k(n, p, k) == want_sk, (n, p, k)


def test_prefill_matrix_cores_and_stream_ranges_are_pinned(B):
    """Which matrix cores a prefill product takes, and the stream form's K ranges (host logic, no device; round 6).
    Matrices that stream from HBM (> 16 MB over the whole model) multiply on the bf16 cores over three-term splits; chunks
    of 33 ... 128 tokens of them take the stream form, whose K ranges -- part of the arithmetic -- follow from the WHOLE
    model's rows by a round count on the 256-CU part the rule was made for: q | k | v of the 7B shape (96 tiles of 128
    features) 2 ranges (1, 2, 4 or 8: the ranges of a tile share its epilogue in equal parts), wo 8, W1 | W3 (172 tiles: one round unsplit) 1, W2 (K = 11008) 8."""
    cores = {(12288, 512, 4096): 1, (4096, 1024, 4096): 1, (22016, 300, 4096): 1, (4096, 129, 11008): 1,   # tile forms
             (12288, 128, 4096): 1 + 2, (4096, 128, 4096): 1 + 8, (22016, 100, 4096): 1 + 1, (4096, 64, 11008): 1 + 8,
             (4096, 49, 4096): 1 + 8, (4096, 33, 4096): 1 + 8, (4096, 32, 4096): 1,          # below the switch-over: the panel kernel (f32 cores) takes the chunk
             (768, 300, 768): 0, (4096, 300, 288): 0, (32000, 64, 288): 1 + 1,   # stories110M / 15M: cache resident; a wide classifier-like matrix streams
             (2304, 100, 768): 0}
    for (n, p, k), want in cores.items():
        assert B.prefill_cores(n, p, k) == want, (n, p, k, B.prefill_cores(n, p, k))
