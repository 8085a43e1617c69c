"""Seeded random model shapes (head sizes 2..128, GQA ratios, odd hidden sizes, tiny and odd
vocabularies, contexts across the pos-256 attention switch) through the C ABI against the CPU
oracle -- the sweep that found the classifier path refusing widths like dim 1152 (scripts/
fuzz_shapes.py; run it with more configs by hand)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fuzz(name="fuzz_shapes"):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "scripts", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_shapes_against_oracle(gpu, ck, orc):
    lines = []
    bad = _fuzz().run(24, 20260926, lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok "))


def test_width_with_a_partial_last_step_and_width_that_takes_the_scalar_kernel(gpu, ck, orc):
    """dim 1152: (dim/4) % 64 != 0 above the narrow-row range -- the generic scalar kernel until round 5, since
    round 6 the 64-lane vector kernel with a partial last float4 step (the reference handles any n with a scalar
    TAIL, main.zig:589-594, not a scalar kernel) and the fused-argmax classifier.  dim 1150 (n % 4 != 0) still takes
    the generic scalar kernel, which has no fused-argmax epilogue: the classifier launch must fall back, not fail."""
    f = _fuzz()
    for cfg in (ck.Config(1152, 2304, 1, 12, 3, 1000, 96), ck.Config(1150, 2302, 1, 25, 5, 1000, 96)):
        lines = []
        assert f.check_config(gpu, ck, orc, np.random.default_rng(5), cfg, False, 77, lines.append), lines


def test_random_shapes_and_world_sizes_sharded_bit_identical(gpu):
    """Emulated ranks (world 2, 3, 4, 6, 8) on random shapes: every rank's logits equal the
    unsharded pass bit for bit, also beyond the pos-256 attention switch (scripts/fuzz_shards.py)."""
    lines = []
    bad = _fuzz("fuzz_shards").run(16, 20260926, lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok "))


def test_fuzz_scheme_b_ranks_agree_and_hold_the_tolerance(gpu):
    """The same random shapes x world sizes under L2Z_SCHEME_B (Wo / W2 by columns, all-reduces): the emulated ranks equal
    each other bit for bit and the unsharded pass within the parity tolerance (scripts/fuzz_shards.py ... b)."""
    lines = []
    bad = _fuzz("fuzz_shards").run(16, 20260928, lines.append, scheme_b=True)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok "))


def test_fuzz_prefill_vs_stepped_and_sharded_vs_unsharded(gpu):
    """Random shapes x prompt lengths (1 .. 530 tokens: every GEMM form and both attention kernels) x
    world sizes: the batched prefill against the stepped loop (logit tolerance), and the row-sharded
    prefill on emulated ranks against the unsharded one, bit for bit (scripts/fuzz_prefill.py; that run
    found the short-prompt kernel form depending on the SHARD's matrix size)."""
    lines = []
    bad = _fuzz("fuzz_prefill").run(28, 20260927, lines.append)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok "))


def test_fuzz_prefill_wide_shapes_panel_kernel_and_scheme_b(gpu):
    """The same harness on matrices that stream from HBM (dim 1024 ... 3072, hidden_dim multiples of 128) and chunks of
    9 ... 80 tokens: the K-range panel kernel (csrc/prefill_panel.hip) on both sides of each of its switch-overs,
    row-sharded bit-identical to the unsharded pass, and a third of the sharded cases under scheme B (column-sharded Wo /
    W2 + the bulk all-reduce: ranks identical to each other, logits at the tolerance)."""
    lines = []
    bad = _fuzz("fuzz_prefill").run(10, 20260929, lines.append, wide=True)
    assert bad == 0, "\n".join(l for l in lines if not l.startswith("ok "))
