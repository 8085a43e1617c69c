// engine.hip -- the persistent decode launch: several consecutive mat-vecs of the forward pass
// (main.zig:392-422 of one layer and :305-358 of the next: wo, w1|w3, w2, q|k|v) as ONE launch whose blocks stay
// resident, hand their vectors over inside the launch and never stop streaming weights while they wait for them.
//
// Why.  Launch by launch, every mat-vec costs ~4-6 us on top of its bytes / 7.3 TB/s: its producer's tail, the launch
// boundary (1.3 us), the first read of x and the rmsnorm (3 us), the refill of an empty memory pipe (2 us) -- 0.6 ms
// of a 4.4 ms token (profiles/r04_overlap_timeline.md: the GPU's own clock, per block).  Overlapping two launches of
// the chain on two streams moved the hand-over into the waiting launch but not off its critical path: the waves that
// wait for x are the waves that stream the weights -- their poll and their sweep of x queue behind their own weight
// requests (loads return in order per wave) -- and one weight batch of run-ahead is 2 us of stream against a 6 us
// hand-over, so every block restarts from an empty pipe exactly like a fresh launch.  What it takes, measured there:
//   * run-ahead deep enough to cover a hand-over: a RING of R weight batches per streaming wave in registers
//     (R * 64 KB per CU in flight: 7 us of stream at R = 3), issued across mat-vec boundaries -- the weights do not
//     depend on x;
//   * the hand-over off the streaming waves' memory queue: a ninth wave per block (the GATHERER) polls, sweeps the
//     handed-over vector into the OTHER x buffer in LDS, normalises it (main.zig:432-468), and runs every unit's
//     EPILOGUE (RoPE + cache write, residual, SwiGLU, logits + argmax candidate) from the streaming waves' partial
//     sums -- lane (h, k) owns the k-th unit of half h for the whole mat-vec: its residual or RoPE values, its
//     outputs, and the {value, epoch} words that publish them when the block's units are done -- so a streaming
//     wave never issues or waits for anything but its weight loads;
//   * no launch boundary between the mat-vecs, so nothing drains the ring.
// The ring's loads and waits are inline assembly: hipcc counts the memory operations it can see, and a wait it
// cannot count exactly (a loop or a branch with a memory operation anywhere between issue and use) becomes
// vmcnt(0) -- the ring would drain at every use.  The streaming waves therefore contain no other load at all.
//
// Arithmetic.  A block is two halves of four streaming waves; half h of block b is virtual block 2b + h of the row
// kernel's grid: same units (row pairs), same thread -> column map, same summation order, same epilogues
// (matvec_device.h) -- the same bits as matvec_row_kernel / matvec_duo_kernel.  The gatherer forms the rmsnorm's sum
// of squares in the 256-thread kernels' order (lane L plays threads L, L + 64, L + 128, L + 192 one after the other).
//
// Synchronisation inside a block is all in LDS (the hardware barrier would include the gatherer): a counting barrier
// among the eight streaming waves per unit (it keeps their sweeps of the matrix in step, as in the row kernel), and
// monotonic words between them and the gatherer (ready: x of mat-vec k staged; units: unit steps whose partial sums
// are in LDS; epi: unit steps whose epilogue has run -- the partials wait in a ring of eight slots).  Between blocks: the LL
// words of p2p.hip in this process's own landing slots (comm_self_create), two slots by epoch parity; a block writes
// hand-over g + 2 only after it has read all of g + 1, which every block wrote only after reading g.
// Every wait is bounded (timeout -> error latch -> the host reports L2Z_ERR_COMM).
#include "matvec_device.h"

namespace l2z {
namespace {

#define L2Z_S __attribute__((address_space(3)))

constexpr int kEngStream = 512;               // streaming threads: two halves of kBlock
constexpr int kEngGather = 256;                // the gatherer: four waves (its leader, wave 8, also runs the epilogues and publishes)
constexpr int kEngThreads = kEngStream + kEngGather;
constexpr int kEngR = 3;                      // weight batches in flight per streaming wave
constexpr int kEngU = 4;                      // float4 per row per thread per batch (as the row kernel)
constexpr int kEngUnits = 32;                 // most units of one mat-vec a half may have (lanes of the gatherer: 2 x 32)

enum { EC_BAR = 0, EC_READY, EC_UNITS, EC_EPI, EC_ERR, EC_DESC, EC_GO, EC_GBAR, EC_WORDS = 8 };  // control words in LDS
constexpr int kEngSlots = 8;                  // units whose wave partials may wait in LDS for the gatherer's epilogue

// what the streaming waves need to know of a mat-vec, in LDS (written once by the gatherer)
struct EngDesc {
    unsigned long long w0, w1, w2;
    int rows0, r01, total_rows, n_pairs, n4, nb, epi, pad;
};

typedef MvLocalsT<true> MvG;

__device__ __forceinline__ MvG eng_locals(const MatvecArgs &a, int epi)
{
    MvG m;
    m.w0 = as_g(a.w0); m.w1 = as_g(a.w1); m.w2 = as_g(a.w2);
    m.out0 = as_g(a.out0); m.out1 = as_g(a.out1); m.out2 = as_g(a.out2);
    m.resid = as_g(a.resid); m.rope = (const L2Z_G float2 *)a.rope;
    m.rows0 = a.rows0; m.r01 = a.rows0 + a.rows1; m.total_rows = a.rows0 + a.rows1 + a.rows2;
    m.n_pairs = (epi == EPI_SWIGLU) ? a.rows0 : (m.total_rows + 1) >> 1;
    m.n = a.n; m.head_size = a.head_size; m.rope_segs = a.rope_segs;
    m.pos = (epi == EPI_ROPE) ? *as_g(a.pos_ptr) : 0;
    m.ps1 = (size_t)m.pos * (size_t)a.pos_stride1;
    m.ps2 = (size_t)m.pos * (size_t)a.pos_stride2;
    m.kv_head_stride = (epi == EPI_ROPE) ? a.kv_head_stride : 0;
    m.push = nullptr; m.push_e = 0; m.push_base = 0;   // published by the gatherer from the stash
    m.resid_slot = nullptr; m.resid_e = 0; m.resid_ctl = nullptr; m.resid_herr = nullptr; m.resid_timeout = 0;
    m.resid_pre = false;                               // residual VALUES come validated from the gatherer (EpiIn::ra / rb)
    return m;
}

// ---- the ring -------------------------------------------------------------------------------------------------
// Fixed physical registers, named in inline assembly (engine_ring.inc, generated by scripts/gen_engine_ring.py):
//   v[48:63] x of the batch being consumed, v[64:159] the ring (slot s = v[64 + 32 s ...]: four float4 of row a, four
//   of row b), v[160:167] the unit's accumulators.
// A register ring written by loads that are still in flight cannot be left to hipcc: it treats the destination of
// a load (its own, or an asm output) as a value it may copy, spill or rotate at will -- with its own loads it puts a
// wait in front of every such move (the pipeline drains at each: the CDNA guide's "HIP does not preserve the
// pipeline"), with asm outputs it moves registers whose data has not arrived.  So a whole step -- wait for the slot,
// read x from LDS, 32 fused multiply-adds, refill the slot -- is ONE asm statement on registers the compiler never
// allocates in these waves (every statement clobbers them all; scripts/check_engine_regs.py verifies the build).
#include "engine_ring.inc"

#ifdef L2Z_TIMELINE
// measurement build (scripts/timeline_build.sh): per (launch, block) wall-clock stamps (100 MHz), stored at the end --
// [0] entry; per mat-vec k: [1 + 6k] streaming waves start waiting for x, [2 + 6k] x ready, [3 + 6k] their last unit
// done; [4 + 6k] gatherer past the gate, [5 + 6k] x staged, [6 + 6k] outputs published.  Read with l2z_engine_timeline_dump.
constexpr int kEtlMax = 1024, kEtlBlocks = 256, kEtlSlots = 32;
__device__ long long g_etl[kEtlMax * kEtlBlocks * kEtlSlots];
#define L2Z_ETL(slot) do { if (tl_on) tl[(slot)] = wall_clock64(); } while (0)
#else
#define L2Z_ETL(slot) do { } while (0)
#endif

// units half 0 of block b has in a mat-vec of n_pairs pairs: the block's unit steps for that mat-vec
__device__ __forceinline__ int eng_steps(int n_pairs, int b, int vgrid) { return (n_pairs - 2 * b + vgrid - 1) / vgrid; }

// ---------------------------------------------------------------------------------------------------------------
// One chunk of consecutive mat-vecs.  chunk->op[k].a is the MatvecArgs launch_matvec would get for mat-vec k in its
// duo form: x plain (first mat-vec of the chunk, written by the launch before this one) or xin (LL words written by
// mat-vec k - 1 of this launch), resid plain or resid_in, push (outputs as LL words) for all but the last.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kEngThreads) void engine_kernel(const EngChunk *__restrict__ chunk, int xs_floats, int tl_seq)
{
#ifdef L2Z_TIMELINE
    const bool tl_on = tl_seq >= 0 && tl_seq < kEtlMax && blockIdx.x < kEtlBlocks && (threadIdx.x == 0 || threadIdx.x == kEngStream);
    long long *tl = g_etl + ((size_t)(tl_on ? tl_seq : 0) * kEtlBlocks + blockIdx.x) * kEtlSlots;
    if (tl_on && threadIdx.x == 0) tl[0] = wall_clock64();
#else
    (void)tl_seq;
#endif
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs_base = lds;                                   // [2][xs_floats] x of the running and of the next mat-vec
    volatile L2Z_S int *ctrl = (volatile L2Z_S int *)(lds + 2 * (size_t)xs_floats);  // EC_WORDS control words
    float *part = lds + 2 * (size_t)xs_floats + EC_WORDS;   // [kEngSlots][half][2][kWaves] wave partials of a unit step
    EngDesc *desc = (EngDesc *)(part + kEngSlots * 2 * 2 * kWaves);  // [kEngMaxOps]
    float *gbest = (float *)(desc + kEngMaxOps);            // [2][kEngUnits][2] classifier: per-unit candidates (value, index bits)
    float *gpart = gbest + 2 * kEngUnits * 2;               // [kWaves] the gatherer waves' partial sums of squares
    EngChunk *lch = (EngChunk *)(gpart + 2 * kWaves);       // the chunk description, copied once: every field the gatherer reads from
                                                            // device memory was a dependent ~1 us round trip (8 us per hand-over)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_ops = chunk->n_ops;
    const int vgrid = 2 * gridDim.x;
    const long long timeout = chunk->timeout_ticks;
    if (tid < EC_WORDS) ctrl[tid] = 0;
    {   // the chunk description into LDS: one coalesced round trip by everybody
        const L2Z_G unsigned *src = (const L2Z_G unsigned *)chunk;
        unsigned *dst = (unsigned *)lch;
        for (int i = tid; i < (int)(sizeof(EngChunk) / 4); i += kEngThreads) dst[i] = src[i];
    }
    __syncthreads();  // the only hardware barrier: all twelve waves, before the roles part

    if (wave >= 8) {
        // =========================================================================================================
        // The gatherer.  Per mat-vec k, in order: stage x (+ this block's per-unit epilogue inputs), then run the
        // epilogue of every unit step as its partial sums arrive, then publish the block's outputs.
        // =========================================================================================================
        asm volatile("; L2Z_GATHER_BEGIN (scripts/check_engine_regs.py)" ::: "memory");
        // Four waves.  Wave 8 leads: it waits at the gate, runs the epilogues and publishes (lane = (half, k-th unit));
        // all four sweep x -- gatherer lane t plays thread t of the 256-thread kernels' staging (float4 t, t + 256, ...;
        // sum of squares per thread, wave sum, four partials in wave order: main.zig:432-468 in that order, bit for bit).
        // One wave alone took 8-14 us per hand-over (profiles/r04_engine_timeline*.md): 16 KB ... 88 KB of words at one
        // or two round trips per 16 loads.
        int *g_ctl = lch->ctl;
        int *h_err = lch->h_err;
        const int gw = wave - 8, gl = tid - kEngStream;
        const bool leader = gw == 0;
        if (leader && lane < n_ops) {
            const MatvecArgs &a = lch->op[lane].a;
            EngDesc d;
            d.w0 = (unsigned long long)a.w0; d.w1 = (unsigned long long)a.w1; d.w2 = (unsigned long long)a.w2;
            d.rows0 = a.rows0; d.r01 = a.rows0 + a.rows1; d.total_rows = a.rows0 + a.rows1 + a.rows2;
            d.n_pairs = lch->op[lane].n_pairs; d.n4 = a.n >> 2; d.nb = lch->op[lane].nb; d.epi = lch->op[lane].epi; d.pad = 0;
            desc[lane] = d;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (leader && lane == 0) ctrl[EC_DESC] = 1;
        const int gh = lane >> 5, gk = lane & 31;          // the leader's lanes: (half, k-th unit)
        int useq = 0;                                      // unit steps whose epilogue has run (all mat-vecs)
        bool bad = false;
        int gbar_target = 0;
        L2Z_S int *gbar = (L2Z_S int *)&ctrl[EC_GBAR];
        auto gather_barrier = [&]() {   // among the four gatherer waves
            gbar_target += kEngGather / 64;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_fetch_add(gbar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(gbar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gbar_target) {
                if (ctrl[EC_ERR] || wall_clock64() - t0 > timeout) { ctrl[EC_ERR] = 1; bad = true; break; }
                __builtin_amdgcn_s_sleep(0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        };
        for (int g = 0; g < n_ops && !bad; g++) {
            const MatvecArgs &a = lch->op[g].a;
            const int pro = lch->op[g].pro, epi = lch->op[g].epi;
            const int n_pairs = lch->op[g].n_pairs;
            const int total_rows = a.rows0 + a.rows1 + a.rows2;
            const int n = a.n, n4 = n >> 2;
            const int n4_pad = ((n4 + kBlock * kEngU - 1) / (kBlock * kEngU)) * (kBlock * kEngU);
            v4f *xs4 = (v4f *)(xs_base + (size_t)(g & 1) * xs_floats);
            const v4f zero = {0.f, 0.f, 0.f, 0.f};
            const int uk = 2 * blockIdx.x + gh + gk * vgrid;   // the leader lane's unit of this mat-vec (if < n_pairs)
            const bool ll = a.xin.slots != nullptr;
            LLPoll lp = {};
            if (ll) lp = ll_poll_init(a.xin);
            if (leader) {
                // ---- the gate: 16 producer blocks' last words (this block's own outputs of mat-vec g - 1 are out: below)
                if (ll && a.xin.hint_n != 0) {
                    const unsigned step = a.xin.hint_n >= kHintLanes ? a.xin.hint_n / kHintLanes : 1u;
                    const unsigned idx = a.xin.hint0 + ((blockIdx.x + (lane & (kHintLanes - 1)) * step) % a.xin.hint_n) * a.xin.hint_stride;
                    const long long t0 = wall_clock64();
                    for (;;) {
                        const unsigned long long w = __hip_atomic_load(lp.slot + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (__all((unsigned)(w >> 32) == lp.e)) break;
                        if (ctrl[EC_ERR] || wall_clock64() - t0 > timeout) { bad = true; break; }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                if (bad) ctrl[EC_ERR] = 1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) ctrl[EC_GO] = g + 1;
                if (bad) break;
            } else {
                const long long t0 = wall_clock64();
                while (ctrl[EC_GO] < g + 1) {
                    if (ctrl[EC_ERR] || wall_clock64() - t0 > 4 * timeout) { bad = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (bad) break;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            L2Z_ETL(4 + 6 * g);
            // ---- this lane's epilogue input: residual values (validated once, here) or the RoPE pair
            v4u ein = {0u, 0u, 0u, 0u};
            if (leader && uk < n_pairs) {
                if (epi == EPI_RESID) {
                    const bool two = 2 * uk + 1 < total_rows;
                    if (a.resid_in.slots != nullptr) {
                        const unsigned e = (unsigned)(a.resid_in.ctl[kCtlEpoch] + a.resid_in.gi);
                        const unsigned long long *slot = a.resid_in.slots + (size_t)(e & 1) * a.resid_in.slot_floats;
                        v4u w = ll_load2(slot, (size_t)(a.row_offset + 2 * uk));
                        const long long t0 = wall_clock64();
                        while (!(w.y == e && (w.w == e || !two))) {
                            if (wall_clock64() - t0 > timeout) { ctrl[EC_ERR] = 1; break; }
                            __builtin_amdgcn_s_sleep(16);
                            w = ll_load2(slot, (size_t)(a.row_offset + 2 * uk));
                        }
                        ein.x = w.x; ein.z = w.z;
                    } else {
                        const L2Z_G float *rp = as_g(a.resid);
                        ein.x = __float_as_uint(rp[2 * uk]);
                        ein.z = two ? __float_as_uint(rp[2 * uk + 1]) : 0u;
                    }
                } else if (epi == EPI_ROPE) {
                    const int ga = 2 * uk;
                    const int r01 = a.rows0 + a.rows1;
                    const bool a1 = ga >= a.rows0, a2 = ga >= r01;
                    const int row_a = ga - (a2 ? r01 : (a1 ? a.rows0 : 0));
                    float2 cs = make_float2(1.0f, 0.0f);
                    if ((a2 ? 2 : (a1 ? 1 : 0)) < a.rope_segs) {
                        const int hs = a.head_size, pos = *as_g(a.pos_ptr);
                        const L2Z_G float *rp = (const L2Z_G float *)a.rope + 2 * ((size_t)pos * (size_t)(hs >> 1) + (size_t)((row_a % hs) >> 1));
                        cs = make_float2(rp[0], rp[1]);
                    }
                    ein.x = __float_as_uint(cs.x);
                    ein.y = __float_as_uint(cs.y);
                }
            }
            // ---- x into the buffer of this mat-vec's parity (free: the streaming waves finished mat-vec g - 2 before
            // mat-vec g - 1, whose epilogues all ran): gatherer lane t holds float4 t + 256 i
            constexpr int RG = 11;  // float4 per lane per round: the longest vector here (11008 floats) in one round of 22 loads
            for (int j0 = gl; j0 < n4_pad; j0 += kEngGather * RG) {
                if (ll) {
                    v4u w[2 * RG];
#pragma unroll
                    for (int i = 0; i < RG; i++) {
                        const int j = j0 + kEngGather * i;
                        const int jc = j < n4 ? j : 0;
                        w[2 * i] = ll_load2(lp.slot, (size_t)4 * jc);
                        w[2 * i + 1] = ll_load2(lp.slot, (size_t)4 * jc + 2);
                    }
#pragma unroll
                    for (int i = 0; i < RG; i++) {
                        const int j = j0 + kEngGather * i;
                        if (j < n4_pad) {
                            const v4f v = ll_wait4(lp, j < n4 ? j : 0, w[2 * i], w[2 * i + 1]);
                            xs4[j] = j < n4 ? v : zero;
                        }
                    }
                } else {
                    const L2Z_G v4f *x4 = (const L2Z_G v4f *)a.x;
                    v4f v[RG];
#pragma unroll
                    for (int i = 0; i < RG; i++) {
                        const int j = j0 + kEngGather * i;
                        v[i] = j < n4 ? x4[j] : zero;
                    }
#pragma unroll
                    for (int i = 0; i < RG; i++) {
                        const int j = j0 + kEngGather * i;
                        if (j < n4_pad) xs4[j] = v[i];
                    }
                }
            }
            if (pro == PRO_RMS) {
                // main.zig:432-468 in the 256-thread kernels' order: thread t sums its float4 t, t + 256, ... (fmaf per
                // component), wave sum, the four wave partials added in wave order
                float ss = 0.0f;
                for (int j = gl; j < n4; j += kEngGather) {
                    const v4f x = xs4[j];   // this lane's own float4 (it wrote them)
                    ss = fmaf(x.x, x.x, ss);
                    ss = fmaf(x.y, x.y, ss);
                    ss = fmaf(x.z, x.z, ss);
                    ss = fmaf(x.w, x.w, ss);
                }
                ss = wave_sum(ss);
                if (lane == 0) gpart[gw] = ss;
                gather_barrier();
                float tot = gpart[0];
#pragma unroll
                for (int v = 1; v < kWaves; v++) tot += gpart[v];
                float sc = tot / (float)n;  // :452
                sc += 1e-5f;                // :453
                sc = 1.0f / sqrtf(sc);      // :454
                const L2Z_G v4f *g4 = (const L2Z_G v4f *)a.rms_w;
                for (int j = gl; j < n4; j += kEngGather) {
                    v4f x = xs4[j];
                    const v4f gw4 = g4[j];
                    x.x = (x.x * sc) * gw4.x;  // :462
                    x.y = (x.y * sc) * gw4.y;
                    x.z = (x.z * sc) * gw4.z;
                    x.w = (x.w * sc) * gw4.w;
                    xs4[j] = x;
                }
            }
            gather_barrier();   // x (and, with the rmsnorm, every lane's share of it) is in LDS
            if (bad) break;
            if (!leader) continue;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) ctrl[EC_READY] = g + 1;
            L2Z_ETL(5 + 6 * g);

            // ---- the epilogues of this mat-vec's unit steps, as their partial sums arrive
            const MvG me = eng_locals(a, epi);
            float hold_a = 0.0f, hold_b = 0.0f;   // this lane's outputs, published when the block is done with the mat-vec
            const int steps = eng_steps(n_pairs, blockIdx.x, vgrid);
            for (int k = 0; k < steps && !bad; k++, useq++) {
                if (ctrl[EC_UNITS] <= useq) {
                    const long long t0 = wall_clock64();
                    while (ctrl[EC_UNITS] <= useq) {
                        if (ctrl[EC_ERR] || wall_clock64() - t0 > timeout) { bad = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (bad) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (gk == k && uk < n_pairs) {
                    const float *pp = part + ((useq % kEngSlots) * 2 + gh) * (2 * kWaves);
                    const float ta = ((pp[0] + pp[1]) + pp[2]) + pp[3];
                    const float tb = ((pp[kWaves] + pp[kWaves + 1]) + pp[kWaves + 2]) + pp[kWaves + 3];
                    EpiIn in;
                    in.ra = __uint_as_float(ein.x); in.rb = __uint_as_float(ein.z);
                    in.cs = make_float2(__uint_as_float(ein.x), __uint_as_float(ein.y));
                    in.rw = ein;
                    float st[2] = {0.0f, 0.0f};
                    if (epi == EPI_ROPE) pair_epilogue<EPI_ROPE>(me, uk, ta, tb, true, in, nullptr);
                    else if (epi == EPI_RESID) pair_epilogue<EPI_RESID>(me, uk, ta, tb, true, in, st);
                    else if (epi == EPI_SWIGLU) pair_epilogue<EPI_SWIGLU>(me, uk, ta, tb, true, in, st);
                    else if (epi == EPI_STORE) {   // a shard's classifier rows: logits slice, published for the logits gather
                        pair_epilogue<EPI_STORE>(me, uk, ta, tb, true, in, nullptr);
                        st[0] = ta; st[1] = tb;
                    } else {
                        pair_epilogue<EPI_ARGMAX>(me, uk, ta, tb, true, in, nullptr);
                        // the unit's candidate (rows 2 uk, 2 uk + 1: strict > keeps the first)
                        float bv = ta; int bi = 2 * uk;
                        if (2 * uk + 1 < total_rows && tb > bv) { bv = tb; bi = 2 * uk + 1; }
                        st[0] = bv; st[1] = __int_as_float(bi);
                    }
                    hold_a = st[0]; hold_b = st[1];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) ctrl[EC_EPI] = useq + 1;
            }
            if (bad) break;
            // ---- publish: every lane its unit's outputs (the {value, epoch} words the next mat-vec's gatherers sweep)
            // (row_offset: this rank's first element of the vector -- 0 unsharded)
            if (uk < n_pairs && a.push != nullptr && (epi == EPI_RESID || epi == EPI_SWIGLU || epi == EPI_STORE)) {
                const int e = a.push_ctl[kCtlEpoch] + a.push_gi;
                const size_t base = (size_t)a.row_offset;
                if (epi == EPI_SWIGLU) {
                    p2p_ll_push(a.push, e, base + (size_t)uk, hold_a);
                } else {
                    p2p_ll_push(a.push, e, base + (size_t)(2 * uk), hold_a);
                    if (2 * uk + 1 < total_rows) p2p_ll_push(a.push, e, base + (size_t)(2 * uk + 1), hold_b);
                }
            }
            L2Z_ETL(6 + 6 * g);
            if (epi == EPI_ARGMAX) {  // one candidate per virtual block: its units in ascending order, strict >
                float *gb = gbest + (gh * kEngUnits + gk) * 2;
                gb[0] = hold_a; gb[1] = hold_b;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (gk == 0) {
                    float bv = -INFINITY; int bi = 0x7fffffff;
                    for (int k = 0; k < steps; k++) {
                        const int u = 2 * blockIdx.x + gh + k * vgrid;
                        if (u >= n_pairs) break;
                        const float v = gbest[(gh * kEngUnits + k) * 2];
                        const int i = __float_as_int(gbest[(gh * kEngUnits + k) * 2 + 1]);
                        if (v > bv || bi == 0x7fffffff) { bv = v; bi = i; }
                    }
                    as_g(a.part_val)[2 * blockIdx.x + gh] = bv;
                    as_g(a.part_idx)[2 * blockIdx.x + gh] = bi == 0x7fffffff ? bi : bi + a.row_offset;
                }
            }
        }
        if ((bad || ctrl[EC_ERR]) && leader && lane == 0) {  // a wait in this block gave up: latch it for every later wait, tell the host
            ctrl[EC_ERR] = 1;
            __hip_atomic_store(g_ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *h_err = 1;
        }
        return;
    }

    // =============================================================================================================
    // The streaming waves: a flat sequence of weight batches over (mat-vec, unit, batch); the issue cursor runs R
    // batches ahead of the consume cursor, across mat-vec boundaries.  No memory operation here but the ring's loads.
    // =============================================================================================================
    asm volatile("; L2Z_STREAM_BEGIN (scripts/check_engine_regs.py)" ::: "memory");
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);  // wave-uniform: what derives from it stays in SGPRs
    const int ht = tid & (kBlock - 1), hw = __builtin_amdgcn_readfirstlane(ht >> 6);
    const int vb = 2 * blockIdx.x + half;
    const L2Z_G float *dummy = as_g(chunk->dummy);  // (a kernel-argument load: uniform)

    bool failed = false;
    auto wait_ge = [&](int word, int target) {   // until ctrl[word] >= target (LDS only)
        if (ctrl[word] >= target) return;
        const long long t0 = wall_clock64();
        while (ctrl[word] < target) {
            if (ctrl[EC_ERR] || wall_clock64() - t0 > timeout) {
                ctrl[EC_ERR] = 1;
                failed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    wait_ge(EC_DESC, 1);

    // issue cursor
    int iop = 0, iu = vb, ib = 0;
    EngDesc di = desc[0];
    const L2Z_G float *pa, *pb;
    auto rows_of = [&](int u) {  // pair_rows (matvec_device.h) on the description in LDS; clamped to the last pair
        if (u >= di.n_pairs) u = di.n_pairs - 1;
        const L2Z_G float *w0 = (const L2Z_G float *)di.w0, *w1 = (const L2Z_G float *)di.w1, *w2 = (const L2Z_G float *)di.w2;
        const size_t n = (size_t)di.n4 * 4;
        if (di.epi == EPI_SWIGLU) {
            pa = w0 + (size_t)(2 * u) * n;
            pb = pa + n;
        } else {
            const int ga = 2 * u;
            const int gb = (ga + 1 < di.total_rows) ? ga + 1 : ga;
            const bool a1 = ga >= di.rows0, a2 = ga >= di.r01;
            const bool b1 = gb >= di.rows0, b2 = gb >= di.r01;
            const int row_a = ga - (a2 ? di.r01 : (a1 ? di.rows0 : 0));
            const int row_b = gb - (b2 ? di.r01 : (b1 ? di.rows0 : 0));
            const L2Z_G float *wa_ = a2 ? w2 : (a1 ? w1 : w0);
            const L2Z_G float *wb_ = b2 ? w2 : (b1 ? w1 : w0);
            pa = wa_ + (size_t)row_a * n;
            pb = wb_ + (size_t)row_b * n;
        }
    };
    auto uni = [](const L2Z_G float *p) {  // the row pointers are wave-uniform: say so (scalar base of the loads)
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
        return (const L2Z_G float *)(((unsigned long long)hi << 32) | lo);
    };
    rows_of(iu);
    pa = uni(pa); pb = uni(pb);
    bool i_valid = true;
    // software barrier among the eight streaming waves (monotonic counter in LDS)
    int bar_target = 0;
    L2Z_S int *bar = (L2Z_S int *)&ctrl[EC_BAR];
    auto stream_barrier = [&]() {
        bar_target += 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add(bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target) {
                if (ctrl[EC_ERR] || wall_clock64() - t0 > timeout) {  // a wave of this block gave up (or never arrives)
                    ctrl[EC_ERR] = 1;
                    failed = true;
                    break;
                }
                __builtin_amdgcn_s_sleep(0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    // per-step operands of the loads: byte offsets of this thread's float4 in the rows (or in the dummy KB) and the
    // wave-uniform bases -- computed by issue_prep() for the batch the issue cursor points at, which it then advances
    unsigned vo[kEngU];
    const L2Z_G float *ba[kEngU], *bb[kEngU];
    auto issue_prep = [&]() {
        const int cbase = ib * (kBlock * kEngU);
        // Steps past the row end (a row's last, partly filled batch; their x is the zero padding) read a fixed, cache
        // resident, finite 1 KB instead: no branch around a load, no HBM traffic (wave-uniform select of the base)
#pragma unroll
        for (int k = 0; k < kEngU; k++) {
            const bool in_row = cbase + hw * 64 + kBlock * k < di.n4;
            vo[k] = in_row ? (unsigned)(cbase + ht + kBlock * k) * 16u : (unsigned)lane * 16u;
            ba[k] = in_row ? pa : dummy;
            bb[k] = in_row ? pb : dummy + 256;
        }
        if (++ib == di.nb) {  // (block-uniform: half 0's unit decides where a mat-vec ends)
            ib = 0;
            iu += vgrid;
            if (iu - half >= di.n_pairs) {
                iop++;
                if (iop < n_ops) {
                    di = desc[iop];
                    iu = vb;
                } else {
                    i_valid = false;
                }
            }
            if (i_valid) { rows_of(iu); pa = uni(pa); pb = uni(pb); }
        }
    };
#define L2Z_ENG_LOAD_OPS [vo0] "v"(vo[0]), [vo1] "v"(vo[1]), [vo2] "v"(vo[2]), [vo3] "v"(vo[3]), \
                         [ba0] "s"(ba[0]), [ba1] "s"(ba[1]), [ba2] "s"(ba[2]), [ba3] "s"(ba[3]), \
                         [bb0] "s"(bb[0]), [bb1] "s"(bb[1]), [bb2] "s"(bb[2]), [bb3] "s"(bb[3])
    asm volatile(L2Z_ENG_ACC_ZERO ::: "memory", L2Z_ENG_CLOBBERS);
    // (Measured, not kept: holding the ring back until the gatherer has requested the first mat-vec's plain x -- which
    // otherwise comes back behind this CU's 192 KB of weight requests, staged 8 us into the launch instead of 3 -- cost
    // more than it saved: 194.1 against 196.3 tok/s; the weights requested at entry are the shorter path.)
    issue_prep(); asm volatile(L2Z_ENG_ISSUE_0 :: L2Z_ENG_LOAD_OPS : "memory", L2Z_ENG_CLOBBERS);   // a chunk has more than R batches per half
    issue_prep(); asm volatile(L2Z_ENG_ISSUE_1 :: L2Z_ENG_LOAD_OPS : "memory", L2Z_ENG_CLOBBERS);
    issue_prep(); asm volatile(L2Z_ENG_ISSUE_2 :: L2Z_ENG_LOAD_OPS : "memory", L2Z_ENG_CLOBBERS);

    // consume cursor
    int cop = 0, cu = vb, cb = 0, useq = 0;
    int c_nb = desc[0].nb, c_pairs = desc[0].n_pairs;
    int xs_off = 0;   // floats from xs_base to the running mat-vec's x
    L2Z_ETL(1);
    wait_ge(EC_READY, 1);
    L2Z_ETL(2);
    bool c_valid = true;

    auto unit_end = [&]() {
        float a0, a1, a2, a3, b0, b1, b2, b3;
        asm volatile(L2Z_ENG_ACC_OUT : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3) :: "memory", L2Z_ENG_CLOBBERS);
        const float sa = wave_sum((a0 + a1) + (a2 + a3));   // hsum4
        const float sb = wave_sum((b0 + b1) + (b2 + b3));
        if (useq >= kEngSlots) wait_ge(EC_EPI, useq - kEngSlots + 1);   // the slot's previous unit step has had its epilogue
        float *pp = part + ((useq % kEngSlots) * 2 + half) * (2 * kWaves);
        if (lane == 0) {
            pp[hw] = sa;
            pp[kWaves + hw] = sb;
        }
        stream_barrier();      // all eight waves' partials of this unit step are in LDS (and the sweeps stay in step)
        useq++;
        if (tid == 0) ctrl[EC_UNITS] = useq;
        cu += vgrid;
        if (cu - half < c_pairs) return;
        L2Z_ETL(3 + 6 * cop);
        cop++;                 // this block is done with the mat-vec
        if (cop >= n_ops) {
            c_valid = false;
            return;
        }
        c_pairs = desc[cop].n_pairs; c_nb = desc[cop].nb;
        cu = vb;
        xs_off = (cop & 1) * xs_floats;
        L2Z_ETL(1 + 6 * cop);
        wait_ge(EC_READY, cop + 1);   // x of this mat-vec staged
        L2Z_ETL(2 + 6 * cop);
    };
    // LDS byte address of this thread's first float4 of the batch (the asm adds k * 4096)
    auto xaddr_of = [&]() { return (unsigned)(size_t)(L2Z_S float *)(xs_base + xs_off) + (unsigned)(cb * (kBlock * kEngU) + ht) * 16u; };
    // main-loop step: wait for the slot (its loads were followed by 2 * 8 younger ones), consume it, refill it, and
    // close the unit if this was its last batch
#define L2Z_ENG_DO_STEP(S)                                                                                     \
    do {                                                                                                       \
        const unsigned xa = xaddr_of();                                                                        \
        issue_prep();                                                                                          \
        asm volatile(L2Z_ENG_STEP_##S :: [xaddr] "v"(xa), L2Z_ENG_LOAD_OPS : "memory", L2Z_ENG_CLOBBERS);     \
        if (++cb == c_nb) { cb = 0; unit_end(); }                                                              \
    } while (0)
#define L2Z_ENG_DO_DRAIN(S)                                                                                    \
    do {                                                                                                       \
        const unsigned xa = xaddr_of();                                                                        \
        asm volatile(L2Z_ENG_DRAIN_##S :: [xaddr] "v"(xa) : "memory", L2Z_ENG_CLOBBERS);                      \
        if (++cb == c_nb) { cb = 0; unit_end(); }                                                              \
    } while (0)
    static_assert(kEngR == 3, "the main loop and its drain are written out for a ring of three");
    int phase;
    for (;;) {  // i_valid goes false in the step that issues the chunk's last batch: R batches are left, in ring order
        L2Z_ENG_DO_STEP(0); if (!i_valid || failed) { phase = 0; break; }
        L2Z_ENG_DO_STEP(1); if (!i_valid || failed) { phase = 1; break; }
        L2Z_ENG_DO_STEP(2); if (!i_valid || failed) { phase = 2; break; }
    }
    if (!failed) {
        if (phase == 0) { L2Z_ENG_DO_DRAIN(1); L2Z_ENG_DO_DRAIN(2); L2Z_ENG_DO_DRAIN(0); }
        else if (phase == 1) { L2Z_ENG_DO_DRAIN(2); L2Z_ENG_DO_DRAIN(0); L2Z_ENG_DO_DRAIN(1); }
        else { L2Z_ENG_DO_DRAIN(0); L2Z_ENG_DO_DRAIN(1); L2Z_ENG_DO_DRAIN(2); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (a failed wait leaves loads in flight: land them before the end)
    (void)c_valid;
}

}  // namespace

size_t engine_lds_bytes(int xs_floats)
{
    // the carve at the top of engine_kernel: x twice, control words, partial-sum slots, descriptions, classifier
    // candidates, the gatherer's partials, the chunk description
    return (size_t)(2 * (size_t)xs_floats + EC_WORDS + kEngSlots * 2 * 2 * kWaves + 2 * kEngUnits * 2 + 2 * kWaves) * sizeof(float) +
           kEngMaxOps * sizeof(EngDesc) + sizeof(EngChunk) + 64;
}

// x buffer size (floats) for a chunk whose widest mat-vec has n columns
int engine_xs_floats(int n_max)
{
    const int n4 = n_max >> 2;
    return 4 * (((n4 + kBlock * kEngU - 1) / (kBlock * kEngU)) * (kBlock * kEngU));
}

// whether a mat-vec of n_pairs pairs fits a grid of `grid` blocks: every half has a unit, none more than kEngUnits
bool engine_units_ok(int n_pairs, int grid)
{
    const int vgrid = 2 * grid;
    return vgrid >= 2 && vgrid <= n_pairs && (n_pairs + vgrid - 1) / vgrid <= kEngUnits;
}

hipError_t launch_engine(const EngChunk *d_chunk, int grid, int xs_floats, hipStream_t st, int tl_seq)
{
    const size_t lds = engine_lds_bytes(xs_floats);
    hipError_t e = ensure_lds(engine_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(engine_kernel, dim3(grid), dim3(kEngThreads), lds, st, d_chunk, xs_floats, tl_seq);
    return hipGetLastError();
}

}  // namespace l2z

#ifdef L2Z_TIMELINE
extern "C" int l2z_engine_timeline_dump(long long *out, int max_launches)
{
    const int n = max_launches < l2z::kEtlMax ? max_launches : l2z::kEtlMax;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(l2z::g_etl), (size_t)n * l2z::kEtlBlocks * l2z::kEtlSlots * sizeof(long long)) != hipSuccess) return 1;
    return 0;
}
#endif
