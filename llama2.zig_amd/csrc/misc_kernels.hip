// misc_kernels.hip -- argmax + loop hand-over, state set-up, the stand-alone kernels behind the
// test hooks (rmsnorm, softmax, dot, weighted row sum) and the synthetic-checkpoint generator.
#include "kernel_common.h"

namespace l2z {
namespace {

// ---------------------------------------------------------------------------
// argmax (main.zig:715-726) + the loop's hand-over (main.zig:999-1003, :1036)
// ---------------------------------------------------------------------------
constexpr int kArgmaxFailed = -2;  // the candidate exchange of a shard group did not complete (a peer timed out)

__global__ __launch_bounds__(1024) void argmax_kernel(const ArgmaxArgs a)
{
    __shared__ float s_val[16];
    __shared__ int s_idx[16];
    __shared__ int s_next;
    const int tid = threadIdx.x;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    if (a.part_val != nullptr) {  // per-block candidates left by the classifier launch
        for (int i = tid; i < a.n_part; i += blockDim.x) {
            const float v = a.part_val[i];
            const int id = a.part_idx[i];
            if (id != 0x7fffffff && (bi == 0x7fffffff || v > best || (v == best && id < bi))) {
                best = v;
                bi = id;
            }
        }
    } else if ((a.vocab & 3) == 0 && ((unsigned long long)a.logits & 15ull) == 0) {
        // the whole vocabulary in flight at once where it fits (32000 logits = 8 x 16 bytes per thread): the scan of a
        // gathered logits vector -- every sharded token -- took 11 us as 32 dependent rounds of 4-byte loads
        constexpr int kU = 8;
        const int n4 = a.vocab >> 2;
        for (int base = tid; base < n4; base += kU * (int)blockDim.x) {
            v4f r[kU];
#pragma unroll
            for (int k = 0; k < kU; k++) {
                const int j = base + k * (int)blockDim.x;
                r[k] = ((const v4f *)a.logits)[j < n4 ? j : base];
            }
#pragma unroll
            for (int k = 0; k < kU; k++) {  // increasing index inside a thread: strict '>' keeps the lowest (:720)
                const int j = base + k * (int)blockDim.x;
                if (j < n4) {
                    const float e[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        if (e[q] > best || bi == 0x7fffffff) {
                            best = e[q];
                            bi = 4 * j + q;
                        }
                }
            }
        }
    } else {
        for (int i = tid; i < a.vocab; i += blockDim.x) {
            const float v = a.logits[i];
            if (v > best || bi == 0x7fffffff) {  // strict '>' keeps the lowest index (:720)
                best = v;
                bi = i;
            }
        }
    }
    // wave reduce: larger value wins, equal values -> lower index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) {
            best = ov;
            bi = oi;
        }
    }
    if ((tid & 63) == 0) {
        s_val[tid >> 6] = best;
        s_idx[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid < kWave) {
        // every lane of the first wave forms the block's candidate (<= 16 wave candidates, in wave order)
        const int nw = blockDim.x >> 6;
        best = s_val[0];
        bi = s_idx[0];
        for (int w = 1; w < nw; w++) {
            if (s_idx[w] != 0x7fffffff &&
                (bi == 0x7fffffff || s_val[w] > best || (s_val[w] == best && s_idx[w] < bi))) {
                best = s_val[w];
                bi = s_idx[w];
            }
        }
        if (a.xchg != nullptr) {
            // The ranks' candidates: lane p sends this rank's pair to rank p -- two LL words {bits, epoch}, each valid by
            // itself -- and waits for rank p's pair in this rank's own slot; then the wave reduces the N pairs by the
            // same rule.  Every rank sees the same N pairs, so every rank picks the same token.
            const P2pArgs *x = a.xchg;
            const int world = x->world, rank = x->rank;
            const unsigned e = (unsigned)(x->ctl[kCtlEpoch] + a.xchg_gi);
            const unsigned long long tag = (unsigned long long)e << 32;
            const size_t off = (size_t)(e & 1u) * x->slot_floats;
            float cv = -INFINITY;
            int ci = 0x7fffffff;
            bool late = false;
            const bool dead = __hip_atomic_load(x->ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            if (tid == rank) {
                cv = best;
                ci = bi;
            } else if (tid < world && !dead) {
                unsigned long long *dst = (unsigned long long *)(x->peer_arena[tid] + kP2pFlagBytes) + off + 2 * (size_t)rank;
                __hip_atomic_store(dst, tag | (unsigned long long)__float_as_uint(best), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(dst + 1, tag | (unsigned long long)(unsigned)bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long *src = (const unsigned long long *)(x->peer_arena[rank] + kP2pFlagBytes) + off + 2 * (size_t)tid;
                const long long t0 = wall_clock64();
                unsigned long long wv, wi;
                for (;;) {
                    wv = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    wi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if ((unsigned)(wv >> 32) == e && (unsigned)(wi >> 32) == e) break;
                    if (wall_clock64() - t0 > x->timeout_ticks ||
                        __hip_atomic_load(x->ctl + kCtlErr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        late = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!late) {
                    cv = __uint_as_float((unsigned)wv);
                    ci = (int)(unsigned)wi;
                } else {
                    __hip_atomic_store(x->ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *x->err = 1 + tid;
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(cv, o, 64);
                const int oi = __shfl_xor(ci, o, 64);
                if (oi != 0x7fffffff && (ci == 0x7fffffff || ov > cv || (ov == cv && oi < ci))) {
                    cv = ov;
                    ci = oi;
                }
            }
            best = cv;
            bi = ci;
            // a candidate that never arrived (or a group already marked dead): the N pairs are not what the other ranks
            // see, so this rank must not hand a token over as if they were -- mark the step invalid instead of diverging
            if (__any(late ? 1 : 0) || dead) bi = kArgmaxFailed;
        }
    }
    if (tid == 0 && bi == kArgmaxFailed) {
        // failed candidate exchange: token, pos, x and the group's epochs stay as they are (every replay that follows
        // finds the error latched and ends at once), the step's output slot gets -1; the host's comm_check after the
        // stream sync turns the latch into L2Z_ERR_COMM before any token is consumed
        if (a.argmax_out) *a.argmax_out = -1;
        if (a.advance) a.out_tokens[*a.pos_ptr] = -1;
        s_next = -1;
    } else if (tid == 0) {
        // (an index outside the vocabulary can only come from a corrupted exchange -- a solo rank's collapsed epochs, a
        // peer that died mid-word: never let it address the embedding table)
        if (bi == 0x7fffffff || (unsigned)bi >= (unsigned)a.vocab) bi = 0;
        if (a.epoch_ctl) a.epoch_ctl[0] += a.epoch_add;  // every launch of the pass has read its epoch long ago
        if (a.argmax_out) *a.argmax_out = bi;
        int next = bi;
        if (a.advance) {
            const int pos = *a.pos_ptr;
            if (pos < *a.n_prompt_ptr) next = a.prompt[pos];  // :999-1000
            a.out_tokens[pos] = next;
            *a.token_ptr = next;                              // :1036
            *a.pos_ptr = pos + 1;                             // :995
        }
        s_next = next;
    }
    __syncthreads();
    if (a.advance && s_next >= 0) {
        // next step's embedding row -> x (main.zig:295-296), saves a launch
        const float *row = a.tok_emb + (size_t)s_next * (size_t)a.dim;
        for (int i = tid; i < a.dim; i += blockDim.x) a.x[i] = row[i];
    }
}

// token/pos from the host + embedding copy (main.zig:295-296)
__global__ void set_state_kernel(int token, int pos, int *token_ptr, int *pos_ptr,
                                 const float *tok_emb, float *x, int dim)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        *token_ptr = token;
        *pos_ptr = pos;
    }
    const float *row = tok_emb + (size_t)token * (size_t)dim;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < dim; i += gridDim.x * blockDim.x)
        x[i] = row[i];
}

// ---------------------------------------------------------------------------
// Stand-alone wrappers for the test hooks: same device functions as above.
// ---------------------------------------------------------------------------
// VEC: the vector staging path the fused mat-vec uses (XC = 4, zero padded)
template <bool VEC>
__global__ __launch_bounds__(kBlock) void rmsnorm_kernel(float *o, const float *x, const float *w,
                                                         int n)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (VEC) {
        const int n4 = n >> 2, n4_pad = (n4 + 255) & ~255;
        float *xs = lds, *scratch = lds + 4 * n4_pad;
        v4f xr[4], gr[4];
        xload_issue<PRO_RMS, 4>(x, w, n4, xr, gr);
        xstage_finish<PRO_RMS, 4>(x, w, n, n4_pad, xr, gr, xs, scratch);
        for (int j = threadIdx.x; j < n; j += blockDim.x) o[j] = xs[j];
    } else {
        float *xs = lds, *scratch = lds + ((n + 3) & ~3);
        stage_x_scalar<PRO_RMS>(x, w, n, xs, scratch);
        for (int j = threadIdx.x; j < n; j += blockDim.x) o[j] = xs[j];
    }
}

__global__ __launch_bounds__(kBlock) void softmax_kernel(float *x, int n)
{
    __shared__ float scratch[kScratch];
    block_softmax(x, n, scratch);
}

// main.zig:1005-1008 for the samplers: probs = softmax(logits / temperature), one block of 1024 threads
// over the vocabulary (the logits are L2 resident: the classifier or the gather has just written them)
__global__ __launch_bounds__(1024) void probs_kernel(float *probs, const float *logits, int n, float temperature)
{
    __shared__ float scratch[kScratch];
    for (int i = threadIdx.x; i < n; i += blockDim.x) probs[i] = logits[i] / temperature;  // :1006
    __syncthreads();
    block_softmax(probs, n, scratch);                                                       // :1008
}

// Seeded synthetic weights: value(idx) = bias + scale*r(idx,seed); must match
// oracle/llama2_oracle.c orc_synth_value and checkpoint.py synth_values bit for bit.
// row_len != 0: element i lands at dst[(i / row_len) * row_pitch + i % row_len] (rows of a strided matrix);
// idx_pitch != 0: its index in the blob is base_idx + (i / row_len) * idx_pitch + i % row_len (a column range of wider rows)
__global__ void synth_fill_kernel(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                                  float scale, float bias, uint64_t row_len, uint64_t row_pitch, uint64_t idx_pitch)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const uint64_t idx = idx_pitch ? (i / row_len) * idx_pitch + i % row_len : i;
        uint64_t z = (base_idx + idx) + seed * 0x9E3779B97F4A7C15ULL;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        const uint32_t u = (uint32_t)(z >> 41);
        const float r = __fsub_rn(__fmul_rn((float)u, 0x1p-22f), 1.0f);
        const uint64_t at = row_len ? (i / row_len) * row_pitch + i % row_len : i;
        dst[at] = __fadd_rn(bias, __fmul_rn(scale, r));
    }
}

struct PartPtrs { const float *p[16]; };
__global__ void sum_parts_kernel(float *out, const PartPtrs parts, int n_parts, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float acc = parts.p[0][i];
    for (int r = 1; r < n_parts; r++) acc = __fadd_rn(acc, parts.p[r][i]);
    out[i] = acc;
}

// dst row r (dpitch floats apart) = src row r (contiguous rows of cols floats)
__global__ void copy_rows_kernel(float *dst, size_t dpitch, const float *src, size_t rows, size_t cols)
{
    const size_t n = rows * cols, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[(i / cols) * dpitch + i % cols] = src[i];
}

}  // namespace

hipError_t launch_argmax(const ArgmaxArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(argmax_kernel, dim3(1), dim3(1024), 0, st, a);
    return hipGetLastError();
}

hipError_t launch_set_state(int token, int pos, int *token_ptr, int *pos_ptr, const float *tok_emb,
                            float *x, int dim, hipStream_t st)
{
    const int grid = (dim + 255) / 256 > 64 ? 64 : (dim + 255) / 256;
    hipLaunchKernelGGL(set_state_kernel, dim3(grid), dim3(256), 0, st, token, pos, token_ptr,
                       pos_ptr, tok_emb, x, dim);
    return hipGetLastError();
}

hipError_t launch_rmsnorm(float *o, const float *x, const float *w, int n, hipStream_t st)
{
    const bool vec = (n % 4) == 0 && aligned16(x) && aligned16(w);
    const size_t lds = matvec_lds_bytes(n);
    if (vec) {
        hipError_t e = ensure_lds(rmsnorm_kernel<true>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rmsnorm_kernel<true>), dim3(1), dim3(kBlock), lds, st, o, x, w, n);
    } else {
        hipError_t e = ensure_lds(rmsnorm_kernel<false>, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((rmsnorm_kernel<false>), dim3(1), dim3(kBlock), lds, st, o, x, w, n);
    }
    return hipGetLastError();
}

hipError_t launch_probs(float *probs, const float *logits, int n, float temperature, hipStream_t st)
{
    hipLaunchKernelGGL(probs_kernel, dim3(1), dim3(1024), 0, st, probs, logits, n, temperature);
    return hipGetLastError();
}

hipError_t launch_softmax(float *x, int n, hipStream_t st)
{
    hipLaunchKernelGGL(softmax_kernel, dim3(1), dim3(kBlock), 0, st, x, n);
    return hipGetLastError();
}

hipError_t launch_synth_fill(float *dst, uint64_t base_idx, uint64_t count, uint64_t seed,
                             float scale, float bias, hipStream_t st, uint64_t row_len, uint64_t row_pitch, uint64_t idx_pitch)
{
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, base_idx,
                       count, seed, scale, bias, row_len, row_pitch, idx_pitch);
    return hipGetLastError();
}

hipError_t launch_sum_parts(float *out, const float *const *parts, int n_parts, int n, hipStream_t st)
{
    if (n_parts < 1 || n_parts > 16) return hipErrorInvalidValue;
    PartPtrs pp = {};
    for (int r = 0; r < n_parts; r++) pp.p[r] = parts[r];
    hipLaunchKernelGGL(sum_parts_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out, pp, n_parts, n);
    return hipGetLastError();
}

hipError_t launch_copy_rows(float *dst, size_t dpitch, const float *src, size_t rows, size_t cols, hipStream_t st)
{
    if (rows == 0 || cols == 0) return hipSuccess;
    size_t blocks = (rows * cols + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dst, dpitch, src, rows, cols);
    return hipGetLastError();
}

}  // namespace l2z
