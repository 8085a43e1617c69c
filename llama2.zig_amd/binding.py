"""ctypes binding over libllama2_hip_test.so (include/llama2_hip.h + include/llama2_hip_test.h; the product library
libllama2_hip.so exports the first header only).

This is what tests/ and bench.py call: every compute path goes through the
C ABI into the hand-written HIP kernels.  There is no Python or CPU fallback
here -- if the library is missing or no gfx950 device is present the calls
raise (`L2ZError`), they never silently compute on the host.
"""
from __future__ import annotations

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product library (exports include/llama2_hip.h only: what a host links) and the library this binding loads: the same
# objects with the test / measurement entry points of include/llama2_hip_test.h exported as well (csrc/Makefile).
PRODUCT_LIB_PATH = os.path.join(_HERE, "libllama2_hip.so")
LIB_PATH = os.environ.get("L2Z_LIB") or os.path.join(_HERE, "libllama2_hip_test.so")  # L2Z_LIB: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "llama2_hip.h")
TEST_HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "llama2_hip_test.h")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_COMM, ERR_STATE = 0, -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128
COMM_IPC_BYTES = 64
KINDS = ["qkv", "attn", "wo", "ffn13", "ffn2", "cls", "argmax", "gather"]


class L2ZError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"llama2_hip error {code}: {msg}")
        self.code = code


class L2ZConfig(C.Structure):
    """src/main.zig:17-25 ConfigReader layout."""

    _fields_ = [(n, C.c_int32) for n in
                ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len")]


def declared_symbols(which: str = "all") -> list[str]:
    """Every function the headers declare (for the export check): the drop-in boundary
    include/llama2_hip.h ("product"), the test / measurement entry points of
    include/llama2_hip_test.h ("test"), or both."""
    paths = {"product": [HEADER_PATH], "test": [TEST_HEADER_PATH],
             "all": [HEADER_PATH, TEST_HEADER_PATH]}[which]
    out = set()
    for path in paths:
        txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        out |= set(re.findall(r"\b(l2z_[a-z0-9_]+)\s*\(", txt))
    return sorted(out)


_lib = None


def lib():
    """Load the library (raises if it was not built -- run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} not built; run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback)")
    L = C.CDLL(LIB_PATH)
    fp, sz, vp = C.POINTER(C.c_float), C.c_size_t, C.c_void_p
    cfgp = C.POINTER(L2ZConfig)
    i32p, ip = C.POINTER(C.c_int32), C.POINTER(C.c_int)
    L.l2z_last_error.restype = C.c_char_p
    L.l2z_device_count.argtypes = [ip]
    L.l2z_device_info.argtypes = [C.c_int, C.c_char_p, sz, ip, C.POINTER(C.c_uint64)]
    L.l2z_weights_init.argtypes = [cfgp, fp, sz, C.c_int, vp, C.POINTER(vp)]
    L.l2z_weights_init_synthetic.argtypes = [cfgp, C.c_int, C.c_uint64, vp, C.POINTER(vp)]
    L.l2z_weights_read.argtypes = [vp, sz, sz, fp]
    L.l2z_weights_free.argtypes = [vp]
    L.l2z_weights_free.restype = None
    L.l2z_runstate_init.argtypes = [cfgp, vp, C.POINTER(vp)]
    L.l2z_runstate_free.argtypes = [vp]
    L.l2z_runstate_free.restype = None
    L.l2z_transformer.argtypes = [C.c_int, C.c_int, cfgp, vp, vp]
    L.l2z_argmax.argtypes = [vp, ip]
    L.l2z_logits_read.argtypes = [vp, fp]
    L.l2z_probs_read.argtypes = [vp, C.c_float, fp]
    L.l2z_runstate_read.argtypes = [vp, C.c_char_p, sz, sz, fp]
    L.l2z_prefill.argtypes = [i32p, C.c_int, C.c_int, cfgp, vp, vp]
    L.l2z_greedy_begin.argtypes = [vp, i32p, C.c_int]
    L.l2z_greedy_run.argtypes = [cfgp, vp, vp, C.c_int, i32p, ip]
    L.l2z_profile_forward.argtypes = [C.c_int, C.c_int, cfgp, vp, vp, C.POINTER(C.c_double), ip,
                                      C.c_int]
    L.l2z_stream_read_probe.argtypes = [vp, vp, sz, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if hasattr(L, "l2z_d2d_copy_probe"):
        L.l2z_d2d_copy_probe.argtypes = [vp, vp, sz, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.l2z_kind_name.argtypes = [C.c_int, C.c_char_p, sz]
    L.l2z_time_kind.argtypes = [C.c_int, C.c_int, cfgp, vp, vp, C.c_int, C.POINTER(C.c_double), ip]
    L.l2z_synchronize.argtypes = [vp]
    L.l2z_matmul.argtypes = [fp, fp, fp, sz, sz]
    L.l2z_matmul_fused.argtypes = [C.c_int, C.POINTER(fp), fp, C.POINTER(fp), sz, sz]
    L.l2z_rmsnorm.argtypes = [fp, fp, fp, sz]
    L.l2z_softmax.argtypes = [fp, sz]
    L.l2z_vector_dot_product.argtypes = [fp, fp, fp, sz]
    L.l2z_vector_weighted_sum_rows.argtypes = [fp, sz, fp, sz, sz, fp, sz]
    L.l2z_argmax_host.argtypes = [fp, sz, C.POINTER(sz)]
    L.l2z_attention_decode.argtypes = [C.c_int, C.c_int, fp, fp, fp, fp] + [C.c_int] * 5
    L.l2z_option_set.argtypes = [C.c_char_p, C.c_longlong]
    L.l2z_comm_unique_id.argtypes = [vp]
    L.l2z_comm_init.argtypes = [C.c_int, C.c_int, vp, C.c_int, C.POINTER(vp)]
    L.l2z_comm_p2p_export.argtypes = [vp, sz, vp]
    if hasattr(L, "l2z_comm_p2p_export_sized"):
        L.l2z_comm_p2p_export_sized.argtypes = [vp, sz, sz, vp]
    L.l2z_comm_p2p_connect.argtypes = [vp, vp]
    L.l2z_comm_rank.argtypes = [vp, ip, ip]
    if hasattr(L, "l2z_comm_p2p_connect_solo"):
        L.l2z_comm_p2p_connect_solo.argtypes = [vp]
    if hasattr(L, "l2z_comm_rccl_info"):
        L.l2z_comm_rccl_info.argtypes = [C.c_char_p, sz, ip]
    if hasattr(L, "l2z_runstate_form"):
        L.l2z_runstate_form.argtypes = [vp, ip]
    if hasattr(L, "l2z_comm_p2p_pingpong"):
        L.l2z_comm_p2p_pingpong.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
        L.l2z_comm_peer_copy_probe.argtypes = [vp, C.c_int, sz, C.c_int, C.POINTER(C.c_double)]
    if hasattr(L, "l2z_comm_transports"):  # an older build loaded through L2Z_LIB (A/B runs) lacks the newer entry points
        L.l2z_comm_transports.argtypes = [vp, ip, ip]
    L.l2z_comm_free.argtypes = [vp]
    L.l2z_comm_free.restype = None
    L.l2z_comm_init_emulated.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.l2z_emu_transformer.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, C.c_int]
    L.l2z_prefill_attention.argtypes = [C.c_int, fp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    L.l2z_prefill_plan.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_int]
    L.l2z_prefill_plan_model.argtypes = [cfgp, C.c_int, C.POINTER(C.c_int), C.c_int]
    if hasattr(L, "l2z_shard_plan"):
        L.l2z_shard_plan.argtypes = [cfgp, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.l2z_prefill_tile.argtypes = [C.c_int, C.c_int, C.c_int]
    if hasattr(L, "l2z_prefill_cores"):
        L.l2z_prefill_cores.argtypes = [C.c_longlong, C.c_int, C.c_int]
    if hasattr(L, "l2z_prefill_split_k"):
        L.l2z_prefill_split_k.argtypes = [C.c_longlong, C.c_int, C.c_int, C.c_int]
    L.l2z_emu_prefill.argtypes = [C.c_int, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32), C.c_int, C.c_int]
    L.l2z_shard_range.argtypes = [C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64)]
    _lib = L
    return L


def _chk(code: int) -> None:
    if code != OK:
        raise L2ZError(code, lib().l2z_last_error().decode(errors="replace"))


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def device_count() -> int:
    n = C.c_int(0)
    _chk(lib().l2z_device_count(C.byref(n)))
    return n.value


def device_info(dev: int = 0):
    name = C.create_string_buffer(256)
    cus, hbm = C.c_int(0), C.c_uint64(0)
    _chk(lib().l2z_device_info(dev, name, 256, C.byref(cus), C.byref(hbm)))
    return name.value.decode(), cus.value, hbm.value


def shard_range(rows: int, granule: int, rank: int, world: int):
    r0, r1 = C.c_int64(0), C.c_int64(0)
    _chk(lib().l2z_shard_range(rows, granule, rank, world, C.byref(r0), C.byref(r1)))
    return r0.value, r1.value


def _cfg(cfg) -> L2ZConfig:
    if isinstance(cfg, L2ZConfig):
        return cfg
    vals = cfg.as_i32() if hasattr(cfg, "as_i32") else cfg
    return L2ZConfig(*[int(v) for v in vals])


class Comm:
    """Multi-GPU shard group (l2z_comm_*)."""

    def __init__(self, rank: int, world: int, uid: bytes | None, device: int, emulated=False):
        self.h = C.c_void_p()
        if emulated:  # rank descriptor only (l2z_emu_transformer), no RCCL
            _chk(lib().l2z_comm_init_emulated(rank, world, device, C.byref(self.h)))
        else:
            buf = C.create_string_buffer(uid, COMM_ID_BYTES) if uid is not None else None
            _chk(lib().l2z_comm_init(rank, world, buf, device, C.byref(self.h)))
        self.rank, self.world = rank, world

    def p2p_export(self, max_vector_floats: int, max_matrix_width: int | None = None) -> bytes:
        """Allocate this rank's landing arena; returns its 64-byte IPC handle (to be all-gathered).
        max_matrix_width = max(dim, hidden_dim) sizes the sharded prefill's bulk regions."""
        buf = C.create_string_buffer(COMM_IPC_BYTES)
        if max_matrix_width is None:
            _chk(lib().l2z_comm_p2p_export(self.h, max_vector_floats, buf))
        else:
            _chk(lib().l2z_comm_p2p_export_sized(self.h, max_vector_floats, max_matrix_width, buf))
        return buf.raw

    def p2p_connect(self, handles: bytes) -> None:
        """handles: every rank's p2p_export() result concatenated in rank order."""
        assert len(handles) == COMM_IPC_BYTES * self.world
        buf = C.create_string_buffer(handles, len(handles))
        _chk(lib().l2z_comm_p2p_connect(self.h, buf))

    def p2p_connect_solo(self) -> None:
        """Measurement: this rank alone, peers' arenas a local sink, no wait ever blocks (l2z_comm_p2p_connect_solo)."""
        _chk(lib().l2z_comm_p2p_connect_solo(self.h))

    def p2p_pingpong(self, other: int, initiator: bool, iters: int = 2000) -> float:
        """microseconds per round trip of one LL word each way between this rank and `other` (both ranks call)"""
        v = C.c_double(0)
        _chk(lib().l2z_comm_p2p_pingpong(self.h, other, 1 if initiator else 0, iters, C.byref(v)))
        return v.value

    def peer_copy_probe(self, other: int, nbytes: int = 16384, iters: int = 200) -> float:
        """microseconds per synchronised device-to-device copy of `nbytes` into rank `other`'s arena (one rank calls)"""
        v = C.c_double(0)
        _chk(lib().l2z_comm_peer_copy_probe(self.h, other, nbytes, iters, C.byref(v)))
        return v.value

    def transports(self) -> dict:
        """{"rccl_ranks": ranks RCCL reports for the communicator (0: none), "p2p": arenas connected}"""
        n, p = C.c_int(0), C.c_int(0)
        _chk(lib().l2z_comm_transports(self.h, C.byref(n), C.byref(p)))
        return {"rccl_ranks": n.value, "p2p": bool(p.value)}

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _chk(lib().l2z_comm_unique_id(buf))
        return buf.raw

    def close(self):
        if self.h:
            lib().l2z_comm_free(self.h)
            self.h = C.c_void_p()


class Weights:
    """src/main.zig:53 Weights, device resident."""

    def __init__(self, cfg, blob: np.ndarray | None, shared: bool, *, seed: int | None = None,
                 comm: Comm | None = None):
        self.cfg = _cfg(cfg)
        self.h = C.c_void_p()
        ch = comm.h if comm is not None else None
        if blob is not None:
            b = blob if (blob.dtype == np.float32 and blob.flags["C_CONTIGUOUS"]) else _f32(blob)
            _chk(lib().l2z_weights_init(C.byref(self.cfg), _fp(b), b.size, int(shared), ch,
                                        C.byref(self.h)))
        else:
            assert seed is not None
            _chk(lib().l2z_weights_init_synthetic(C.byref(self.cfg), int(shared), seed, ch,
                                                  C.byref(self.h)))

    def read(self, offset: int, count: int) -> np.ndarray:
        out = np.empty(count, np.float32)
        _chk(lib().l2z_weights_read(self.h, offset, count, _fp(out)))
        return out

    def close(self):
        if self.h:
            lib().l2z_weights_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RunState:
    """src/main.zig:119 RunState, device resident."""

    def __init__(self, cfg, comm: Comm | None = None):
        self.cfg = _cfg(cfg)
        self.h = C.c_void_p()
        _chk(lib().l2z_runstate_init(C.byref(self.cfg), comm.h if comm is not None else None,
                                     C.byref(self.h)))

    def transformer(self, token: int, pos: int, w: Weights) -> None:
        """src/main.zig:285"""
        _chk(lib().l2z_transformer(token, pos, C.byref(self.cfg), self.h, w.h))

    def prefill(self, tokens, pos0: int, w: Weights) -> None:
        """l2z_prefill: the state change of transformer(tokens[i], pos0+i) for all i, batched."""
        t = np.ascontiguousarray(tokens, np.int32)
        _chk(lib().l2z_prefill(t.ctypes.data_as(C.POINTER(C.c_int32)), t.size, pos0,
                               C.byref(self.cfg), self.h, w.h))

    def argmax(self) -> int:
        t = C.c_int(0)
        _chk(lib().l2z_argmax(self.h, C.byref(t)))
        return t.value

    def logits(self) -> np.ndarray:
        out = np.empty(self.cfg.vocab_size, np.float32)
        _chk(lib().l2z_logits_read(self.h, _fp(out)))
        return out

    def probs(self, temperature: float = 1.0) -> np.ndarray:
        """softmax(logits / temperature) computed on the device (l2z_probs_read)."""
        out = np.empty(self.cfg.vocab_size, np.float32)
        _chk(lib().l2z_probs_read(self.h, C.c_float(temperature), _fp(out)))
        return out

    def read(self, name: str, offset: int, count: int) -> np.ndarray:
        out = np.empty(count, np.float32)
        _chk(lib().l2z_runstate_read(self.h, name.encode(), offset, count, _fp(out)))
        return out

    def greedy_begin(self, prompt=()) -> None:
        p = np.ascontiguousarray(prompt, np.int32)
        _chk(lib().l2z_greedy_begin(self.h, p.ctypes.data_as(C.POINTER(C.c_int32)), p.size))

    def greedy_run(self, w: Weights, n_steps: int) -> np.ndarray:
        out = np.zeros(max(n_steps, 1), np.int32)
        n = C.c_int(0)
        _chk(lib().l2z_greedy_run(C.byref(self.cfg), self.h, w.h, n_steps,
                                  out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(n)))
        return out[: n.value].copy()

    def profile_forward(self, token: int, pos: int, w: Weights):
        ms = (C.c_double * len(KINDS))()
        cnt = (C.c_int * len(KINDS))()
        _chk(lib().l2z_profile_forward(token, pos, C.byref(self.cfg), self.h, w.h, ms, cnt,
                                       len(KINDS)))
        return {k: (ms[i], cnt[i]) for i, k in enumerate(KINDS)}

    def form(self) -> int:
        """The decode structure this state runs: bit 0 paired mat-vec blocks, bit 1 two chains, bit 2 persistent launches."""
        f = C.c_int(0)
        _chk(lib().l2z_runstate_form(self.h, C.byref(f)))
        return f.value

    def time_kind(self, kind: str, pos: int, w: Weights, reps: int = 4):
        """(average ms per launch, launches) of one kind of launch, back to back between one event pair."""
        ms, n = C.c_double(0), C.c_int(0)
        _chk(lib().l2z_time_kind(KINDS.index(kind), pos, C.byref(self.cfg), self.h, w.h, reps, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def stream_read_probe(self, w: Weights, slice_bytes: int = 0, reps: int = 8):
        """(average, best) GB/s of a pure streaming-read kernel over the resident weight blob."""
        avg, best = C.c_double(0), C.c_double(0)
        _chk(lib().l2z_stream_read_probe(self.h, w.h, slice_bytes, reps, C.byref(avg), C.byref(best)))
        return avg.value, best.value

    def d2d_copy_probe(self, w: Weights, slice_bytes: int = 0, reps: int = 8):
        """(average, best) GB/s of bytes copied by a device-to-device hipMemcpyAsync of pieces of the weight blob."""
        avg, best = C.c_double(0), C.c_double(0)
        _chk(lib().l2z_d2d_copy_probe(self.h, w.h, slice_bytes, reps, C.byref(avg), C.byref(best)))
        return avg.value, best.value

    def synchronize(self) -> None:
        _chk(lib().l2z_synchronize(self.h))

    def close(self):
        if self.h:
            lib().l2z_runstate_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def emu_transformer(states, weights, token: int, pos: int) -> None:
    """One forward pass of N emulated ranks on one GPU (l2z_emu_transformer)."""
    n = len(states)
    ss = (C.c_void_p * n)(*[s.h for s in states])
    ws = (C.c_void_p * n)(*[w.h for w in weights])
    _chk(lib().l2z_emu_transformer(n, ss, ws, token, pos))


def prefill_attention(form: int, q, kcache, vcache, pos0: int, n_heads: int, n_kv_heads: int, head_size: int) -> np.ndarray:
    """The batched prefill's attention kernels on (q [P, dim], caches [seq_len, kv_dim]); form as in the header."""
    q, kcache, vcache = _f32(q), _f32(kcache), _f32(vcache)
    out = np.empty_like(q)
    _chk(lib().l2z_prefill_attention(form, _fp(out), _fp(q), _fp(kcache), _fp(vcache), pos0, q.shape[0], n_heads,
                                     n_kv_heads, head_size, kcache.shape[0]))
    return out


def shard_plan(cfg, rank: int, world: int) -> dict:
    """What rank `rank` of `world` owns (host logic, no device): scheme A's row / head ranges and scheme B's padded widths."""
    buf = (C.c_int * 10)()
    c = _cfg(cfg)
    _chk(lib().l2z_shard_plan(C.byref(c), rank, world, buf, 10))
    keys = ("dim0", "dim_loc", "kvd_loc", "heads_loc", "hid0", "hid_loc", "v0", "v_loc", "dimc_pad", "hidc_pad")
    return dict(zip(keys, list(buf)))


def prefill_plan(n_tokens: int, cfg=None) -> list[int]:
    """Chunk lengths of a batched prefill of n_tokens (host logic, no device); with a config: of that model's."""
    buf = (C.c_int * 64)()
    if cfg is not None:
        c = _cfg(cfg)
        n = lib().l2z_prefill_plan_model(C.byref(c), n_tokens, buf, 64)
    else:
        n = lib().l2z_prefill_plan(n_tokens, buf, 64)
    if n < 0:
        raise L2ZError(n, lib().l2z_last_error().decode(errors="replace"))
    return list(buf[:n])


TILE_FORMS = ("128x64", "64x64", "32x64", "32x32", "128x128")


def prefill_tile(n_features: int, n_tokens: int, paired: bool = False) -> str:
    """Output tile the direct-to-LDS GEMM takes for an [n_tokens, n_features] product (host logic)."""
    t = lib().l2z_prefill_tile(n_features, n_tokens, 1 if paired else 0)
    if t < 0:
        raise L2ZError(t, lib().l2z_last_error().decode(errors="replace"))
    return TILE_FORMS[t]


def prefill_split_k(n_features_whole: int, n_tokens: int, k: int, paired: bool = False) -> int:
    """K ranges per output tile the tile GEMM takes for this product (host logic; 1 = the unsplit family)."""
    r = lib().l2z_prefill_split_k(n_features_whole, n_tokens, k, 1 if paired else 0)
    if r < 0:
        raise L2ZError(r, lib().l2z_last_error().decode(errors="replace"))
    return r


def prefill_cores(n_features_whole: int, n_tokens: int, k: int) -> int:
    """Matrix cores of a prefill product (host logic): 0 f32, 1 bf16 over three-term splits (tile forms), n >= 2 the
    stream form of the bf16 kernel with n - 1 K ranges."""
    r = lib().l2z_prefill_cores(n_features_whole, n_tokens, k)
    if r < 0:
        raise L2ZError(r, lib().l2z_last_error().decode(errors="replace"))
    return r


def prefill_on_bf16_cores(cfg) -> bool:
    """Whether the model's widest product (W1 | W3) multiplies on the bf16 matrix cores in the batched prefill."""
    return prefill_cores(2 * cfg.hidden_dim, 512, cfg.dim) > 0


def emu_prefill(states, weights, tokens, pos0: int) -> None:
    """l2z_prefill for N emulated ranks on one GPU (l2z_emu_prefill)."""
    n = len(states)
    ss = (C.c_void_p * n)(*[s.h for s in states])
    ws = (C.c_void_p * n)(*[w.h for w in weights])
    t = np.ascontiguousarray(tokens, dtype=np.int32)
    _chk(lib().l2z_emu_prefill(n, ss, ws, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), pos0))


# ---- kernel-level hooks (names follow src/main.zig) ----
def matmul(x, w) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    d, n = w.shape
    out = np.empty(d, np.float32)
    _chk(lib().l2z_matmul(_fp(out), _fp(x), _fp(w), n, d))
    return out


def matmul_fused(x, ws) -> list[np.ndarray]:
    x = _f32(x)
    ws = [_f32(w) for w in ws]
    d, n = ws[0].shape
    outs = [np.empty(d, np.float32) for _ in ws]
    FP = C.POINTER(C.c_float)
    N = len(ws)
    _chk(lib().l2z_matmul_fused(N, (FP * N)(*[_fp(o) for o in outs]), _fp(x),
                                (FP * N)(*[_fp(w) for w in ws]), n, d))
    return outs


def rmsnorm(x, w) -> np.ndarray:
    x, w = _f32(x), _f32(w)
    o = np.empty_like(x)
    _chk(lib().l2z_rmsnorm(_fp(o), _fp(x), _fp(w), x.size))
    return o


def softmax(x) -> np.ndarray:
    o = np.array(x, np.float32, copy=True)
    _chk(lib().l2z_softmax(_fp(o), o.size))
    return o


def vector_dot_product(x, y) -> np.float32:
    x, y = _f32(x), _f32(y)
    o = np.zeros(1, np.float32)
    _chk(lib().l2z_vector_dot_product(_fp(o), _fp(x), _fp(y), x.size))
    return o[0]


def vector_weighted_sum_rows(xout_len: int, rows, row_stride: int, weights) -> np.ndarray:
    rows, weights = _f32(rows), _f32(weights)
    o = np.empty(xout_len, np.float32)
    _chk(lib().l2z_vector_weighted_sum_rows(_fp(o), xout_len, _fp(rows), rows.size, row_stride,
                                            _fp(weights), weights.size))
    return o


ATTN_FORMS = {"auto": 0, "fast256": 1, "fast1024": 2, "split": 3, "generic": 4}


def attention_decode(q, kcache, vcache, pos: int, n_heads: int, n_kv_heads: int, head_size: int,
                     seq_len: int, form: str = "auto", nch: int = 0) -> np.ndarray:
    """One layer's decode attention (src/main.zig:361-389) through the forward pass's kernels."""
    q, kcache, vcache = _f32(q), _f32(kcache), _f32(vcache)
    assert q.size == n_heads * head_size and kcache.size == seq_len * n_kv_heads * head_size
    out = np.empty(n_heads * head_size, np.float32)
    _chk(lib().l2z_attention_decode(ATTN_FORMS[form], nch, _fp(out), _fp(q), _fp(kcache), _fp(vcache),
                                    pos, n_heads, n_kv_heads, head_size, seq_len))
    return out


def rccl_info() -> dict:
    """Loads RCCL now; {'path': the file the process got, 'version': ncclGetVersion's code}."""
    buf, v = C.create_string_buffer(512), C.c_int(0)
    _chk(lib().l2z_comm_rccl_info(buf, 512, C.byref(v)))
    return {"path": buf.value.decode(errors="replace"), "version": v.value}


def option_set(name: str, value: int) -> None:
    """A tuning knob of csrc/tunables.h by its environment name; applies to objects created afterwards."""
    _chk(lib().l2z_option_set(name.encode(), int(value)))


def argmax(x) -> int:
    x = _f32(x)
    i = C.c_size_t(0)
    _chk(lib().l2z_argmax_host(_fp(x), x.size, C.byref(i)))
    return int(i.value)
