"""llama2.zig_amd -- MI355X (gfx950) forward pass for cgbur/llama2.zig.

Holds only what the hot path needs:
  csrc/        hand-written HIP kernels + the C ABI (include/llama2_hip.h)
  host/        C++ host driver mirroring the reference CLI (main.zig:800-1051)
  binding.py   ctypes over libllama2_hip.so (tests / bench call through this)
  checkpoint.py  llama2.c v0 checkpoint layout (main.zig:17-25, :85-112)

The directory name contains a dot, so it is loaded under the module name
`llama2_zig_amd` by __graft_entry__.load_package().
"""
from . import checkpoint  # noqa: F401
from . import binding  # noqa: F401

__all__ = ["checkpoint", "binding"]
