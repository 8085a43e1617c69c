"""ctypes wrapper over oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may
import this module, and only as the checker / the timed CPU baseline.  The
product package (llama2.zig_amd) never imports it.

The library is the C restatement of /root/reference/src/main.zig (see
llama2_oracle.h for the per-function citations and the parity status).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")


class OrcConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("dim", "hidden_dim", "n_layers", "n_heads", "n_kv_heads", "vocab_size", "seq_len")]


class OrcWeights(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("token_embedding_table", "rms_att_weight", "rms_ffn_weight", "wq", "wk", "wv",
                 "wo", "w1", "w2", "w3", "rms_final_weight", "freq_cis_real", "freq_cis_imag",
                 "wcls")]


class OrcRunState(C.Structure):
    _fields_ = [(n, C.POINTER(C.c_float)) for n in
                ("x", "xb", "xb2", "hb", "hb2", "q", "k", "v", "att", "logits", "key_cache",
                 "value_cache")]


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


NATIVE_FLAGS = ["-O3", "-march=native", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-std=c11", "-D_GNU_SOURCE"]
SHIPPED_FLAGS = "-O3 -march=x86-64-v3 -ffp-contract=off (oracle/Makefile, built where the tree was built)"


def build_native():
    """BASELINE.md's CPU-baseline plan: the port at `-O3 -march=native` on the host that is timed.  Compiles
    oracle/liboracle_native.so with this host's gcc and returns (path, flags); None when there is no compiler
    or the build fails (the caller then times the shipped library and says so).  Same source, same
    -ffp-contract=off: only the instruction selection changes, the arithmetic is still orc_set_mode's."""
    import shutil
    cc = shutil.which(os.environ.get("CC", "gcc"))
    if not cc:
        return None
    out = os.path.join(_HERE, "liboracle_native.so")
    cmd = [cc, *NATIVE_FLAGS, "-shared", "-o", out, os.path.join(_HERE, "llama2_oracle.c"), "-lm", "-lpthread"]
    try:
        subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    except (subprocess.SubprocessError, OSError):
        return None
    return out, f"{os.path.basename(cc)} " + " ".join(NATIVE_FLAGS[:-1])


def use_library(path: str) -> None:
    """Load the oracle from `path` from now on (bench.py's cpu_baseline: the -march=native build)."""
    global _LIB_PATH, _lib
    _LIB_PATH, _lib = path, None


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp, sz, i32p = C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_int32)
        L.orc_set_mode.argtypes = [C.c_int] * 3
        L.orc_rmsnorm.argtypes = [fp, fp, fp, sz]
        L.orc_matmul.argtypes = [fp, fp, fp, sz, sz]
        L.orc_matmul_fused.argtypes = [C.c_int, C.POINTER(fp), fp, C.POINTER(fp), sz, sz]
        L.orc_vector_dot_product.argtypes = [fp, fp, sz]
        L.orc_vector_dot_product.restype = C.c_float
        L.orc_vector_mul.argtypes = [fp, fp, sz]
        L.orc_vector_weighted_sum.argtypes = [fp, fp, C.c_float, sz]
        L.orc_vector_weighted_sum_rows.argtypes = [fp, sz, fp, sz, fp, sz]
        L.orc_softmax.argtypes = [fp, sz]
        L.orc_accum.argtypes = [fp, fp, sz]
        L.orc_argmax.argtypes = [fp, sz]
        L.orc_argmax.restype = sz
        L.orc_rope.argtypes = [fp, fp, sz, sz, sz, sz]
        L.orc_weights_count.argtypes = [C.POINTER(OrcConfig), C.c_int]
        L.orc_weights_count.restype = sz
        L.orc_weights_init.argtypes = [C.POINTER(OrcWeights), C.POINTER(OrcConfig), fp, C.c_int]
        L.orc_runstate_init.argtypes = [C.POINTER(OrcRunState), C.POINTER(OrcConfig)]
        L.orc_runstate_init.restype = C.c_int
        L.orc_runstate_free.argtypes = [C.POINTER(OrcRunState)]
        L.orc_transformer.argtypes = [sz, sz, C.POINTER(OrcConfig), C.POINTER(OrcRunState),
                                      C.POINTER(OrcWeights)]
        L.orc_generate_greedy.argtypes = [C.POINTER(OrcConfig), C.POINTER(OrcRunState),
                                          C.POINTER(OrcWeights), i32p, sz, sz, i32p, fp]
        L.orc_generate_greedy.restype = sz
        L.orc_synth_value.argtypes = [C.c_uint64, C.c_uint64, C.c_float, C.c_float]
        L.orc_synth_value.restype = C.c_float
        L.orc_synth_fill.argtypes = [fp, C.POINTER(OrcConfig), C.c_int, C.c_uint64, C.c_int]
        L.orc_synth_fill_range.argtypes = [fp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_float,
                                           C.c_float]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def set_mode(vector_width: int = 8, use_fma: bool = False, tree_reduce: bool = False) -> None:
    lib().orc_set_mode(int(vector_width), int(use_fma), int(tree_reduce))


# All modes: the readings of the reference that cannot be told apart without
# a Zig compiler (DEFAULT_VECTOR_WIDTH x FMA contraction x @reduce order).
ALL_MODES = [(vw, fma, tree) for vw in (4, 8, 16) for fma in (0, 1) for tree in (0, 1)]


# ---- kernel wrappers (numpy in, numpy out) ----
def rmsnorm(x, w):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    o = np.empty_like(x)
    lib().orc_rmsnorm(_fp(o), _fp(x), _fp(w), x.size)
    return o


def matmul(x, w):
    """w: (d, n) row-major, x: (n,) -> (d,)"""
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    d, n = w.shape
    o = np.empty(d, np.float32)
    lib().orc_matmul(_fp(o), _fp(x), _fp(w), n, d)
    return o


def matmul_fused(x, ws):
    x = np.ascontiguousarray(x, np.float32)
    ws = [np.ascontiguousarray(w, np.float32) for w in ws]
    d, n = ws[0].shape
    outs = [np.empty(d, np.float32) for _ in ws]
    N = len(ws)
    FP = C.POINTER(C.c_float)
    o_arr = (FP * N)(*[_fp(o) for o in outs])
    w_arr = (FP * N)(*[_fp(w) for w in ws])
    lib().orc_matmul_fused(N, o_arr, _fp(x), w_arr, n, d)
    return outs


def vector_dot_product(x, y):
    x = np.ascontiguousarray(x, np.float32); y = np.ascontiguousarray(y, np.float32)
    return np.float32(lib().orc_vector_dot_product(_fp(x), _fp(y), x.size))


def vector_weighted_sum_rows(xout_len, rows, row_stride, weights):
    rows = np.ascontiguousarray(rows, np.float32); weights = np.ascontiguousarray(weights, np.float32)
    o = np.empty(xout_len, np.float32)
    lib().orc_vector_weighted_sum_rows(_fp(o), xout_len, _fp(rows), row_stride, _fp(weights),
                                       weights.size)
    return o


def vector_weighted_sum(xout, x, y):
    o = np.array(xout, np.float32, copy=True); x = np.ascontiguousarray(x, np.float32)
    lib().orc_vector_weighted_sum(_fp(o), _fp(x), float(y), x.size)
    return o


def softmax(x):
    o = np.array(x, np.float32, copy=True)
    lib().orc_softmax(_fp(o), o.size)
    return o


def rope(q, k, pos: int, head_size: int):
    """main.zig:336-351 on copies of q [dim] and k [kv_dim]; returns (q, k) rotated for `pos`"""
    q = np.array(q, np.float32, copy=True); k = np.array(k, np.float32, copy=True)
    lib().orc_rope(_fp(q), _fp(k), pos, q.size, k.size, head_size)
    return q, k


def argmax(x):
    x = np.ascontiguousarray(x, np.float32)
    return int(lib().orc_argmax(_fp(x), x.size))


def synth_fill(cfg_i32, shared: bool, seed: int, n_threads: int = 1) -> np.ndarray:
    c = OrcConfig(*[int(v) for v in cfg_i32])
    n = lib().orc_weights_count(C.byref(c), int(shared))
    blob = np.empty(n, np.float32)
    lib().orc_synth_fill(_fp(blob), C.byref(c), int(shared), seed, n_threads)
    return blob


class Model:
    """Config + Weights + RunState over a host blob; mirrors main.zig:946-975."""

    def __init__(self, cfg_i32, blob: np.ndarray, shared: bool):
        self.cfg = OrcConfig(*[int(v) for v in cfg_i32])
        self.blob = np.ascontiguousarray(blob, np.float32)
        need = lib().orc_weights_count(C.byref(self.cfg), int(shared))
        assert self.blob.size >= need, (self.blob.size, need)
        self.w = OrcWeights()
        lib().orc_weights_init(C.byref(self.w), C.byref(self.cfg), _fp(self.blob), int(shared))
        self.s = OrcRunState()
        if lib().orc_runstate_init(C.byref(self.s), C.byref(self.cfg)) != 0:
            raise MemoryError("orc_runstate_init")

    def transformer(self, token: int, pos: int) -> np.ndarray:
        lib().orc_transformer(token, pos, C.byref(self.cfg), C.byref(self.s), C.byref(self.w))
        return np.ctypeslib.as_array(self.s.logits, shape=(self.cfg.vocab_size,)).copy()

    def state(self, name: str, n: int) -> np.ndarray:
        return np.ctypeslib.as_array(getattr(self.s, name), shape=(n,)).copy()

    def generate_greedy(self, prompt, steps: int):
        prompt = np.ascontiguousarray(prompt, np.int32)
        cap = max(int(steps) if steps else self.cfg.seq_len, 1)
        out = np.zeros(cap, np.int32)
        margins = np.zeros(cap, np.float32)
        n = lib().orc_generate_greedy(C.byref(self.cfg), C.byref(self.s), C.byref(self.w),
                                      prompt.ctypes.data_as(C.POINTER(C.c_int32)), prompt.size,
                                      steps, out.ctypes.data_as(C.POINTER(C.c_int32)), _fp(margins))
        return out[:n].copy(), margins[:n].copy()

    def close(self):
        if self.s is not None:
            lib().orc_runstate_free(C.byref(self.s))
            self.s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
