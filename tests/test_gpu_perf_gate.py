"""Perf gates (round 5: the 7B decode kernels; round 6: everything else that is measured) against profiles/perf_floor.json.

A number in the bench line is no protection: round 4 slowed wo by 12 % and nobody saw it for a round.  Here every kind of
launch of every BASELINE shape, the batched prefill at the chunk lengths each of its kernel families serves, long-context
attention and a rank's whole sharded pass are measured inside `pytest -m gpu` with the functions of scripts/perf_floor.py
(which also wrote the floors) and held to the committed figures.

Noise: some processes run EVERY launch ~3 % slower (where the big allocations land, clock state).  So groups of figures
are compared after dividing out the group's common factor (the lower quartile of measured / floor): one kernel falling behind
fails at `slack`, everything drifting together only at `common_slack`.  The floors belong to ONE part: on another device
name / CU count the gates skip -- they are regression tests of this code on the machine it is tuned for, not
portability tests."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
FLOOR = json.load(open(os.path.join(ROOT, "profiles", "perf_floor.json")))


@pytest.fixture(scope="module")
def pf(gpu):
    name, cus, _ = gpu.device_info(0)
    want = FLOOR["device"]
    if want["name_contains"] not in name or cus != want["cus"]:
        pytest.skip(f"perf floors were recorded on {want['name_contains']} / {want['cus']} CUs; this is {name} / {cus}")
    import perf_floor
    return perf_floor


def check_group(what, got, floor, slack, common_slack, higher_is_better=False, slack_by_key=None, common_over=None):
    """got / floor: dicts with the same keys.  ratio > 1 = worse."""
    ratio = {k: (floor[k] / got[k] if higher_is_better else got[k] / floor[k]) for k in floor}
    keys = [k for k in ratio if common_over is None or k in common_over]
    # the common factor is the LOWER QUARTILE of the ratios: a process-wide slowdown lifts every ratio alike, a regression
    # lifts some -- even most -- of them above the rest (a median would swallow a kernel that serves most of the group;
    # the minimum is one noisy reading: the 19-us classifier launch of a small model came in 3 % under its floor and
    # failed everybody else)
    common = float(np.percentile([ratio[k] for k in keys], 25)) if len(keys) >= 3 else 1.0
    print(what, {k: round(v, 3) for k, v in got.items()}, f"common factor {common:.3f}")
    bad = [f"{k}: {got[k]:.3f} vs its floor {floor[k]} = {ratio[k] / common:.3f} x after the common factor {common:.3f}"
           for k in ratio if ratio[k] / common > 1.0 + (slack_by_key or {}).get(k, slack)]
    assert not bad, what + ": " + "; ".join(bad)
    assert common <= 1.0 + common_slack, f"{what}: everything is {common:.3f} x its floor"


@pytest.mark.parametrize("shape", list(FLOOR["decode_us_per_launch"]["floors"]))
def test_decode_launches_stay_at_their_floor(gpu, ck, pf, shape):
    """every kind of launch of the decode pass (the kind's launches of all layers back to back between ONE event pair =
    rocprofv3's kernel durations, best of 3): 7B (HBM streaming), stories110M / 42M / 15M (launch-latency bound; the
    fused rmsnorm + q|k|v + RoPE + attention launch of small MHA models is their `qkv`)"""
    g = FLOOR["decode_us_per_launch"]
    got = pf.decode_kinds(gpu, ck, shape, g["pos"])
    assert set(got) == set(g["floors"][shape]), (sorted(got), sorted(g["floors"][shape]))
    check_group(f"{shape} decode, us per launch", got, g["floors"][shape], g.get("slack_by_shape", {}).get(shape, g["slack"]), g["common_slack"],
                slack_by_key=g.get("slack_by_kind"), common_over=[k for k in got if k != "attn"])


def test_small_model_tokens_per_s_stay_at_their_floor(gpu, ck, pf):
    """stories15M / 42M / 110M, -t 0, 255 graph replays, best of 3"""
    g = FLOOR["decode_tokens_per_s"]
    got = {nm: pf.decode_tokens_per_s(gpu, ck, nm) for nm in g["floors"]}
    check_group("tokens/s", got, g["floors"], g["slack"], g["common_slack"], higher_is_better=True)


def test_prefill_and_long_context_attention_stay_at_their_floor(gpu, ck, pf):
    """the batched prefill of the 7B shape at 16 (short-prompt GEMMs), 32 (the K-range panel kernel, f32 matrix cores),
    48 / 64 (the stream form of the bf16-core kernel at two token tiles: the weight stream's pace) and 96 / 128 (three / four
    token tiles) / 256 / 512 / 1024 tokens (its tile forms: the matrix cores' pace, which differs by 7 % between boxes -- that
    group's common factor may reach 12 %), best of 6; and the split decode attention at the last position of the
    2048-token context"""
    cfg = ck.LLAMA2_7B
    w, s = gpu.Weights(cfg, None, False, seed=2024), gpu.RunState(cfg)
    try:
        g = FLOOR["prefill_ms"]
        got = {n: pf.prefill_ms(gpu, ck, w, s, cfg, int(n)) for n in g["floors"]}
        for name, keys in g["groups"].items():   # (the bf16-core kernels move together with the clock: their own common factor)
            check_group(f"7B prefill, ms, {name}", {k: got[k] for k in keys}, {k: g["floors"][k] for k in keys}, g["slack"],
                        g.get("common_slack_by_group", {}).get(name, g["common_slack"]))
        s.greedy_begin([]); s.greedy_run(w, 2)
        a = FLOOR["attention_us_per_layer_pos2047"]
        us = pf.attention_long_us(gpu, w, s)
        print(f"attention at pos 2047: {us:.2f} us per layer (floor {a['floor']})")
        assert us <= a["floor"] * (1.0 + a["slack"]), (us, a)
    finally:
        s.close(); w.close()


def test_one_rank_of_eight_alone_stays_at_its_floor(gpu, ck, pf):
    """rank 0 of 8 alone on the GPU, its whole sharded pass with free hand-overs (l2z_comm_p2p_connect_solo):
    scheme A with gather launches and scheme B -- the figures config 5's first real run will be read against"""
    g = FLOOR["solo_rank_tokens_per_s"]
    got = pf.solo_rank(gpu, ck, g["world"])
    assert all(got.get(k) for k in g["floors"]), got
    for k, f in g["floors"].items():
        print(f"solo rank of {g['world']}, {k}: {got[k]:.1f} tok/s (floor {f})")
        assert got[k] >= f * (1.0 - g["slack"]), (k, got[k], f)
