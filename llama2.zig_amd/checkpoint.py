"""llama2.c "legacy v0" checkpoint format, as the reference reads it.

Layout (cited against /root/reference/src/main.zig):
  * 28-byte header = ConfigReader, 7 x i32 little endian (main.zig:17-25, read
    at :941).  vocab_size < 0 signals an UNSHARED classifier (main.zig:943-944).
  * one flat f32 blob carved by Weights.init in a fixed order (main.zig:85-112),
    including the freq_cis_real/imag region that is skipped but never read.

No real checkpoint exists in the build image, so this module also produces
seeded synthetic blobs.  The generator is specified in DESIGN.md ("Synthetic
checkpoints") and implemented three times -- here (numpy), in the CPU oracle
(oracle/llama2_oracle.c: orc_synth_*) and on the device (csrc/misc_kernels.hip: synth_fill_kernel) --
tests check the three agree bit for bit.
"""
from __future__ import annotations

import dataclasses
import struct
from typing import Iterator, NamedTuple

import numpy as np

HEADER_BYTES = 28


@dataclasses.dataclass(frozen=True)
class Config:
    """main.zig:41-49 (vocab_size already abs()'d)."""

    dim: int
    hidden_dim: int
    n_layers: int
    n_heads: int
    n_kv_heads: int
    vocab_size: int
    seq_len: int

    @property
    def head_size(self) -> int:
        return self.dim // self.n_heads

    @property
    def kv_dim(self) -> int:
        return (self.dim * self.n_kv_heads) // self.n_heads

    @property
    def kv_mul(self) -> int:
        return self.n_heads // self.n_kv_heads

    def as_i32(self) -> np.ndarray:
        return np.array(dataclasses.astuple(self), dtype=np.int32)


# The shapes BASELINE.json names.
STORIES15M = Config(288, 768, 6, 6, 6, 32000, 256)
STORIES110M = Config(768, 2048, 12, 12, 12, 32000, 1024)
# llama2.c's public stories42M checkpoint: not a BASELINE config, but a model the reference runs whose hidden_dim
# (1376 = 4 * 344, 344 = 5 * 64 + 24) is not a whole number of 64-lane float4 steps: the mat-vec's partial last step
STORIES42M = Config(512, 1376, 8, 8, 8, 32000, 1024)
LLAMA2_7B = Config(4096, 11008, 32, 32, 32, 32000, 2048)


class Tensor(NamedTuple):
    name: str
    offset: int  # in f32, from the start of the blob
    shape: tuple  # logical shape
    scale: np.float32  # synthetic generator: value = bias + scale * r
    bias: np.float32

    @property
    def count(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def tensor_table(c: Config, shared_weights: bool) -> list[Tensor]:
    """The Weights.init pointer walk (main.zig:85-112) as a table."""
    L, dim, hid, V, S = c.n_layers, c.dim, c.hidden_dim, c.vocab_size, c.seq_len
    hs, kvd = c.head_size, c.kv_dim
    f = np.float32
    s_dim = np.sqrt(f(3.0) / f(dim), dtype=np.float32)
    s_hid = np.sqrt(f(3.0) / f(hid), dtype=np.float32)
    s_emb = f(2.0) * s_dim
    rows = [
        ("token_embedding_table", (V, dim), s_emb, f(0)),
        ("rms_att_weight", (L, dim), f(0.1), f(1)),
        ("wq", (L, dim, dim), s_dim, f(0)),
        ("wk", (L, kvd, dim), s_dim, f(0)),
        ("wv", (L, kvd, dim), s_dim, f(0)),
        ("wo", (L, dim, dim), s_dim, f(0)),
        ("rms_ffn_weight", (L, dim), f(0.1), f(1)),
        ("w1", (L, hid, dim), s_dim, f(0)),
        ("w2", (L, dim, hid), s_hid, f(0)),
        ("w3", (L, hid, dim), s_dim, f(0)),
        ("rms_final_weight", (dim,), f(0.1), f(1)),
        ("freq_cis_real", (S * hs // 2,), f(1), f(0)),
        ("freq_cis_imag", (S * hs // 2,), f(1), f(0)),
    ]
    if not shared_weights:
        rows.append(("wcls", (V, dim), s_emb, f(0)))
    out, off = [], 0
    for name, shape, scale, bias in rows:
        t = Tensor(name, off, shape, f(scale), f(bias))
        out.append(t)
        off += t.count
    return out


def weights_count(c: Config, shared_weights: bool) -> int:
    t = tensor_table(c, shared_weights)[-1]
    return t.offset + t.count


def file_size(c: Config, shared_weights: bool) -> int:
    return HEADER_BYTES + 4 * weights_count(c, shared_weights)


def carve(c: Config, blob: np.ndarray, shared_weights: bool) -> dict[str, np.ndarray]:
    """Views into the blob, one per tensor (Weights.init)."""
    assert blob.dtype == np.float32 and blob.ndim == 1
    out = {}
    for t in tensor_table(c, shared_weights):
        out[t.name] = blob[t.offset : t.offset + t.count].reshape(t.shape)
    if shared_weights:
        out["wcls"] = out["token_embedding_table"]  # main.zig:112
    return out


def _mix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def synth_values(base_idx: int, count: int, seed: int, scale, bias) -> np.ndarray:
    """value(idx) = bias + scale * r(idx, seed); r on a 2^-22 grid in [-1, 1)."""
    with np.errstate(over="ignore"):
        idx = np.arange(base_idx, base_idx + count, dtype=np.uint64)
        z = _mix64(idx + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15))
    u = (z >> np.uint64(41)).astype(np.float32)  # 23 bits, exact
    r = u * np.float32(2.0**-22) - np.float32(1.0)  # exact
    return (np.float32(bias) + np.float32(scale) * r).astype(np.float32)


def synth_blob(c: Config, shared_weights: bool, seed: int) -> np.ndarray:
    blob = np.empty(weights_count(c, shared_weights), dtype=np.float32)
    for t in tensor_table(c, shared_weights):
        chunk = 1 << 22
        for lo in range(0, t.count, chunk):
            n = min(chunk, t.count - lo)
            blob[t.offset + lo : t.offset + lo + n] = synth_values(
                t.offset + lo, n, seed, t.scale, t.bias
            )
    return blob


def write_checkpoint(path, c: Config, blob: np.ndarray, shared_weights: bool) -> None:
    assert blob.dtype == np.float32 and blob.size == weights_count(c, shared_weights)
    hdr = list(dataclasses.astuple(c))
    if not shared_weights:
        hdr[5] = -hdr[5]  # main.zig:943: negative vocab = unshared classifier
    with open(path, "wb") as f:
        f.write(struct.pack("<7i", *hdr))
        f.write(blob.astype("<f4", copy=False).tobytes())


def read_checkpoint(path, mmap: bool = True):
    """Returns (Config, shared_weights, blob[f32]).  main.zig:936-967."""
    with open(path, "rb") as f:
        hdr = struct.unpack("<7i", f.read(HEADER_BYTES))
    shared = hdr[5] > 0
    c = Config(hdr[0], hdr[1], hdr[2], hdr[3], hdr[4], abs(hdr[5]), hdr[6])
    if mmap:
        blob = np.memmap(path, dtype="<f4", mode="r", offset=HEADER_BYTES)
    else:
        blob = np.fromfile(path, dtype="<f4", offset=HEADER_BYTES)
    need = weights_count(c, shared)
    if blob.size < need:
        raise ValueError(f"checkpoint too small: {blob.size} f32 < {need}")
    return c, shared, blob[:need]


def iter_configs() -> Iterator[tuple[str, Config, bool]]:
    yield "stories15M", STORIES15M, True
    yield "stories110M", STORIES110M, True
    yield "stories42M", STORIES42M, True
    yield "llama2-7b", LLAMA2_7B, False
