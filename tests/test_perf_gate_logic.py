"""The perf gate's comparison rule on made-up readings (CPU; the measurements themselves need the GPU:
tests/test_gpu_perf_gate.py).  What it must do: fail a 7 % slowdown of ONE kernel family even when that family serves
many of the group's figures (the batched prefill: the stream form of the bf16 kernel serves 4 of 9 chunk lengths, its tile forms
3 more), pass a process in which
everything runs 3 % slower, and fail when everything drifts past the common slack."""
import json
import os

import pytest

from test_gpu_perf_gate import FLOOR, check_group


def test_floor_file_has_every_group_and_names_its_device():
    assert FLOOR["device"]["cus"] == 256 and FLOOR["device"]["name_contains"]
    assert set(FLOOR["decode_us_per_launch"]["floors"]) == {"llama2-7b", "stories110M", "stories42M", "stories15M"}
    assert set(FLOOR["prefill_ms"]["floors"]) == {"16", "32", "48", "64", "96", "128", "256", "512", "1024"}
    assert set(FLOOR["decode_tokens_per_s"]["floors"]) == {"stories15M", "stories42M", "stories110M"}
    assert set(FLOOR["solo_rank_tokens_per_s"]["floors"]) == {"p2p-gather", "p2p-allreduce"}
    assert FLOOR["attention_us_per_layer_pos2047"]["floor"] > 0


def test_gate_rule_on_synthetic_readings():
    g = FLOOR["prefill_ms"]
    assert sorted(sum(g["groups"].values(), [])) == sorted(g["floors"])
    name = [n for n in g["groups"] if "matrix cores' pace" in n][0]
    f = {k: g["floors"][k] for k in g["groups"][name]}
    cs = g["common_slack_by_group"][name]
    ok = {k: v * 1.07 for k, v in f.items()}                       # a box whose matrix cores clock 7 % lower: everything + 7 %
    check_group("prefill", ok, f, g["slack"], cs)
    stream = {k: v * (1.07 * 0.85 + 0.15 if k in ("96", "128") else 1.0) for k, v in f.items()}  # stream kernel + 7 %: 85 % of those chunks' time
    with pytest.raises(AssertionError, match="96"):
        check_group("prefill", stream, f, g["slack"], cs)
    tiles = {k: v * (1.07 * 0.9 + 0.1 if k in ("256", "512", "1024") else 1.0) for k, v in f.items()}   # the tile forms + 7 %
    with pytest.raises(AssertionError, match="512"):
        check_group("prefill", tiles, f, g["slack"], cs)
    with pytest.raises(AssertionError, match="everything"):
        check_group("prefill", {k: v * 1.14 for k, v in f.items()}, f, g["slack"], cs)
    d = FLOOR["decode_us_per_launch"]
    f15 = d["floors"]["stories15M"]
    fused = dict(f15, qkv=f15["qkv"] * 1.5)                         # fused_qkv_attn_kernel + 50 % (launch-floor launches: 40 % per launch; tok/s: 4 %)
    with pytest.raises(AssertionError, match="qkv"):
        check_group("15M", fused, f15, d["slack_by_shape"]["stories15M"], d["common_slack"], slack_by_key=d["slack_by_kind"])
    noisy = dict(f15, cls=f15["cls"] * 0.97, wo=f15["wo"] * 1.03)   # one reading 3 % under its floor must not fail the others
    check_group("15M", noisy, f15, d["slack_by_shape"]["stories15M"], d["common_slack"], slack_by_key=d["slack_by_kind"])
    f7 = d["floors"]["llama2-7b"]
    with pytest.raises(AssertionError, match="wo"):                 # round 4's wo regression
        check_group("7B", dict(f7, wo=f7["wo"] * 1.12), f7, d["slack"], d["common_slack"], slack_by_key=d["slack_by_kind"], common_over=[k for k in f7 if k != "attn"])
    t = FLOOR["decode_tokens_per_s"]
    check_group("tok/s", {k: v * 0.98 for k, v in t["floors"].items()}, t["floors"], t["slack"], t["common_slack"], higher_is_better=True)
    with pytest.raises(AssertionError):
        check_group("tok/s", dict(t["floors"], stories15M=t["floors"]["stories15M"] * 0.94), t["floors"], t["slack"], t["common_slack"],
                    higher_is_better=True)
