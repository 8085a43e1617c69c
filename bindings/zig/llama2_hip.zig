//! llama2_hip.zig -- Zig declarations for include/llama2_hip.h (ABI version 2).
//!
//! Drop this file next to the reference's src/main.zig; main_hip.patch makes
//! main() call through it and build_hip.patch links libllama2_hip.so.
//!
//! Each extern replaces one site of the reference (cgbur/llama2.zig, src/main.zig):
//!   ConfigReader :17-25          L2zConfig (same 7 x i32, extern layout)
//!   Weights.init :73, :967       l2z_weights_init
//!   RunState.init :137, :974     l2z_runstate_init
//!   RunState.deinit :156, :975   l2z_runstate_free / l2z_weights_free
//!   transformer :285, :996       l2z_transformer
//!   argmax :715, :1003           l2z_argmax
//!   state.logits :1005-1012      l2z_logits_read / l2z_probs_read
//!   the -t 0 loop :995-1042      l2z_greedy_begin / l2z_greedy_run (optional fast path)
//!   prompt positions :999-1000   l2z_prefill (optional)
//!
//! Written against Zig 0.16 (build.zig.zon:5 of the reference).  The build
//! image of this repository has no Zig compiler: tests/test_zig_shim.py builds
//! and runs this file wherever `zig version` reports 0.16 and skips otherwise.
const std = @import("std");

pub const abi_version: c_int = 2;

/// == ConfigReader (main.zig:17-25) after :944 made vocab_size positive.
pub const L2zConfig = extern struct {
    dim: i32,
    hidden_dim: i32,
    n_layers: i32,
    n_heads: i32,
    n_kv_heads: i32,
    vocab_size: i32,
    seq_len: i32,
};

comptime {
    std.debug.assert(@sizeOf(L2zConfig) == 28);
    std.debug.assert(@offsetOf(L2zConfig, "seq_len") == 24);
}

pub const L2zWeights = opaque {};
pub const L2zRunState = opaque {};
pub const L2zComm = opaque {};

/// l2z_status of include/llama2_hip.h
pub const Status = enum(c_int) {
    ok = 0,
    invalid = -1,
    no_device = -2,
    hip = -3,
    oom = -4,
    comm = -5,
    state = -6,
    _,
};

pub extern fn l2z_abi_version() c_int;
pub extern fn l2z_last_error() [*:0]const u8;
pub extern fn l2z_device_count(out_n: *c_int) c_int;

pub extern fn l2z_weights_init(
    config: *const L2zConfig,
    data: [*]const f32,
    n_floats: usize,
    shared_weights: c_int,
    comm: ?*const L2zComm,
    out: *?*L2zWeights,
) c_int;
pub extern fn l2z_weights_free(w: ?*L2zWeights) void;

pub extern fn l2z_runstate_init(config: *const L2zConfig, comm: ?*const L2zComm, out: *?*L2zRunState) c_int;
pub extern fn l2z_runstate_free(s: ?*L2zRunState) void;

pub extern fn l2z_transformer(
    token: c_int,
    pos: c_int,
    config: *const L2zConfig,
    s: *L2zRunState,
    w: *const L2zWeights,
) c_int;
pub extern fn l2z_argmax(s: *L2zRunState, out_token: *c_int) c_int;
pub extern fn l2z_logits_read(s: *L2zRunState, out_logits: [*]f32) c_int;
pub extern fn l2z_probs_read(s: *L2zRunState, temperature: f32, out_probs: [*]f32) c_int;

pub extern fn l2z_greedy_begin(s: *L2zRunState, prompt: ?[*]const i32, n_prompt: c_int) c_int;
pub extern fn l2z_greedy_run(
    config: *const L2zConfig,
    s: *L2zRunState,
    w: *const L2zWeights,
    n_steps: c_int,
    out_tokens: [*]i32,
    out_n: *c_int,
) c_int;
pub extern fn l2z_prefill(
    tokens: [*]const i32,
    n_tokens: c_int,
    pos0: c_int,
    config: *const L2zConfig,
    s: *L2zRunState,
    w: *const L2zWeights,
) c_int;
pub extern fn l2z_synchronize(s: *L2zRunState) c_int;

pub const Error = error{DeviceForwardFailed};

/// The reference's transformer() cannot fail; a device path can.  Every entry
/// point returns 0 or a negative l2z_status: print the library's message and
/// turn it into a Zig error.
pub fn check(rc: c_int) Error!void {
    if (rc != 0) {
        std.debug.print("llama2_hip: {s} (status {d})\n", .{ l2z_last_error(), rc });
        return error.DeviceForwardFailed;
    }
}

/// Device-side twins of the reference's Weights + RunState, made once after
/// the checkpoint is in memory (main.zig:967, :974) and freed with deinit().
pub const Device = struct {
    cfg: L2zConfig,
    w: *L2zWeights,
    s: *L2zRunState,

    /// `data` is the blob that follows the 28-byte header (main.zig:957-965);
    /// it may be freed as soon as this returns.
    pub fn init(cfg: L2zConfig, data: []align(4) const u8, shared_weights: bool) Error!Device {
        if (l2z_abi_version() != abi_version) {
            std.debug.print("llama2_hip: library ABI {d}, binding ABI {d}\n", .{ l2z_abi_version(), abi_version });
            return error.DeviceForwardFailed;
        }
        var w: ?*L2zWeights = null;
        try check(l2z_weights_init(&cfg, @ptrCast(data.ptr), data.len / 4, @intFromBool(shared_weights), null, &w));
        errdefer l2z_weights_free(w);
        var s: ?*L2zRunState = null;
        try check(l2z_runstate_init(&cfg, null, &s));
        return .{ .cfg = cfg, .w = w.?, .s = s.? };
    }

    pub fn deinit(self: *Device) void {
        l2z_runstate_free(self.s);
        l2z_weights_free(self.w);
        self.* = undefined;
    }

    /// main.zig:996
    pub fn transformer(self: *Device, token: usize, pos: usize) Error!void {
        try check(l2z_transformer(@intCast(token), @intCast(pos), &self.cfg, self.s, self.w));
    }

    /// main.zig:1003 -- on the device, strict '>' (lowest index wins ties, :720)
    pub fn argmax(self: *Device) Error!usize {
        var t: c_int = 0;
        try check(l2z_argmax(self.s, &t));
        return @intCast(t);
    }

    /// state.logits for the host samplers (main.zig:1005-1012)
    pub fn readLogits(self: *Device, logits: []f32) Error!void {
        std.debug.assert(logits.len == @as(usize, @intCast(self.cfg.vocab_size)));
        try check(l2z_logits_read(self.s, logits.ptr));
    }
};
