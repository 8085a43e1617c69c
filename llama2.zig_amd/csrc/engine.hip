// engine.hip -- the persistent decode launch: several consecutive mat-vecs of the forward pass
// (main.zig:392-422 of one layer and :305-358 of the next: wo, w1|w3, w2, q|k|v) as ONE launch whose blocks stay
// resident, hand their vectors over inside the launch and never stop streaming weights while they wait for them.
//
// Why.  Launch by launch, every mat-vec costs 3.8 us on top of its bytes / 7.3 TB/s: its producer's tail, the launch
// boundary, the first read of x, the refill of an empty memory pipe -- 0.6 ms of a 4.4 ms token (profiles/r04_*).
// Overlapping two launches of a chain on two streams moved the hand-over into the waiting launch but not off its
// critical path: the waves that wait for x are the waves that stream the weights, their poll and their sweep of x
// queue behind their own weight requests (loads return in order per wave), and one weight batch of run-ahead is
// 2 us of stream against a 6 us hand-over (profiles/r04_overlap_*).  What it takes, measured there:
//   * run-ahead deep enough to cover a hand-over: a RING of R weight batches per streaming wave in registers
//     (R * 64 KB per CU in flight: 7 us of stream at R = 3), issued across mat-vec boundaries -- the weights do not
//     depend on x;
//   * the hand-over off the streaming waves' memory queue: a ninth wave per block (the GATHERER) polls, sweeps the
//     handed-over vector into the OTHER x buffer in LDS, normalises it (main.zig:432-468), and also does the
//     block's own publishing -- the {value, epoch} words of its outputs -- so no streaming wave ever waits on
//     a system-scope store or an uncached load;
//   * no launch boundary between the mat-vecs, so nothing drains the ring.
//
// Arithmetic.  A block is two halves of four streaming waves; half h of block b is virtual block 2b + h of the row
// kernel's grid: same units (row pairs), same thread -> column map, same summation order, same epilogues
// (matvec_device.h) -- the same bits as matvec_row_kernel / matvec_duo_kernel.  The gatherer forms the rmsnorm's sum
// of squares in the 256-thread kernels' order (lane L plays threads L, L + 64, L + 128, L + 192 one after the other).
//
// Synchronisation inside a block is all in LDS (the hardware barrier would include the gatherer): a counting barrier
// among the eight streaming waves per unit, and three monotonic words between them and the gatherer (ready: x of
// mat-vec k staged; done: the halves' epilogues of mat-vec k finished; pub: its outputs published).  Between blocks:
// the LL words of p2p.hip in this process's own landing slots (comm_self_create), two slots by epoch parity; a block
// writes hand-over g + 2 only after it has read all of g + 1, which every block wrote only after reading g.
// Every wait is bounded (timeout -> error latch -> the host reports L2Z_ERR_COMM).
#include <type_traits>

#include "matvec_device.h"

namespace l2z {
namespace {

constexpr int kEngStream = 512;            // streaming threads: two halves of kBlock
constexpr int kEngThreads = kEngStream + 64;  // + the gatherer wave
constexpr int kEngR = 3;                   // weight batches in flight per streaming wave
constexpr int kEngU = 4;                   // float4 per row per thread per batch (as the row kernel)
constexpr int kEngUnits = 32;              // most units of one mat-vec a half may have (lanes of the gatherer: 2 x 32)

// control words in LDS (ints)
enum { EC_BAR = 0, EC_READY, EC_DONE, EC_PUB, EC_ERR, EC_WORDS = 8 };

// runtime (per mat-vec) forms of the templated helpers of matvec_device.h
__device__ __forceinline__ MvLocals eng_locals(const MatvecArgs &a, int epi)
{
    MvLocals m;
    m.w0 = a.w0; m.w1 = a.w1; m.w2 = a.w2;
    m.out0 = a.out0; m.out1 = a.out1; m.out2 = a.out2;
    m.resid = a.resid; m.rope = a.rope;
    m.rows0 = a.rows0; m.r01 = a.rows0 + a.rows1; m.total_rows = a.rows0 + a.rows1 + a.rows2;
    m.n_pairs = (epi == EPI_SWIGLU) ? a.rows0 : (m.total_rows + 1) >> 1;
    m.n = a.n; m.head_size = a.head_size; m.rope_segs = a.rope_segs;
    m.pos = (epi == EPI_ROPE) ? *as_g(a.pos_ptr) : 0;
    m.ps1 = (size_t)m.pos * (size_t)a.pos_stride1;
    m.ps2 = (size_t)m.pos * (size_t)a.pos_stride2;
    m.kv_head_stride = (epi == EPI_ROPE) ? a.kv_head_stride : 0;
    m.push = a.push;
    m.push_e = m.push ? as_g(a.push_ctl)[kCtlEpoch] + a.push_gi : 0;
    m.push_base = 0;
    m.resid_slot = nullptr; m.resid_e = 0; m.resid_ctl = nullptr; m.resid_herr = nullptr; m.resid_timeout = 0;
    m.resid_pre = false;
    if (epi == EPI_RESID && a.resid_in.slots != nullptr) {
        const int e = a.resid_in.ctl[kCtlEpoch] + a.resid_in.gi;
        m.resid_e = (unsigned)e;
        m.resid_slot = a.resid_in.slots + (size_t)(e & 1) * a.resid_in.slot_floats;
        m.resid_ctl = a.resid_in.ctl; m.resid_herr = a.resid_in.h_err; m.resid_timeout = a.resid_in.timeout_ticks;
        m.resid_pre = true;
    }
    return m;
}

__device__ __forceinline__ void eng_pair_rows(const MvLocals &m, int epi, int p, const float *&pa, const float *&pb)
{
    if (epi == EPI_SWIGLU) pair_rows<EPI_SWIGLU>(m, p, pa, pb);
    else pair_rows<EPI_STORE>(m, p, pa, pb);
}

__device__ __forceinline__ int eng_spin_failed(volatile int *ctrl, long long t0, long long timeout, int *g_ctl, int *h_err, int code)
{
    if (ctrl[EC_ERR]) return 1;
    if (wall_clock64() - t0 > timeout) {
        ctrl[EC_ERR] = 1;
        __hip_atomic_store(g_ctl + kCtlErr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *h_err = code;
        return 1;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// One chunk of consecutive mat-vecs.  ops[k].a is the MatvecArgs launch_matvec would get for mat-vec k in its duo
// form: x plain (first mat-vec of the chunk, written by the launch before this one) or xin (LL words written by
// mat-vec k - 1 of this launch), resid plain or resid_in, push (outputs as LL words) for all but the last.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kEngThreads) void engine_kernel(const EngChunk *__restrict__ chunk, int xs_floats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *xs_base = lds;                                   // [2][xs_floats] x of the running and of the next mat-vec
    volatile int *ctrl = (volatile int *)(lds + 2 * (size_t)xs_floats);      // EC_WORDS control words
    float *part = (float *)(ctrl + EC_WORDS);               // [half][parity][2][kWaves] wave partials
    float *stash = part + 4 * (2 * kWaves);                 // [parity of k][half][kEngUnits][2] outputs to publish
    v4u *epin = (v4u *)(stash + 2 * 2 * kEngUnits * 2);     // [parity of k][half][kEngUnits] epilogue inputs
    MvLocals *lmc = (MvLocals *)(epin + 2 * 2 * kEngUnits);  // [parity of k] what the epilogues of mat-vec k need (read by two threads per unit:
                                                            // held in registers by every streaming wave it cost ~40 SGPRs and spilled)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n_ops = chunk->n_ops;
    const int vgrid = 2 * gridDim.x;
    int *g_ctl = chunk->ctl;
    int *h_err = chunk->h_err;
    const long long timeout = chunk->timeout_ticks;
    if (tid < EC_WORDS) ctrl[tid] = 0;
    __syncthreads();  // the only hardware barrier: all nine waves, before the roles part

    if (wave == 8) {
        // =========================================================================================================
        // The gatherer: stages x of mat-vec g (in consume order, at most one mat-vec ahead of the streaming waves) and
        // publishes the block's outputs of mat-vec p as soon as its two halves are done with it.
        // =========================================================================================================
        int g = 0, p = 0;
        const int gh = lane >> 5, gk = lane & 31;          // this lane's (half, k-th unit) for the per-unit duties
        while (g < n_ops || p < n_ops) {
            bool progressed = false;
            // ---- publish mat-vec p
            if (p < n_ops && ctrl[EC_DONE] >= 2 * (p + 1)) {
                const MatvecArgs &a = chunk->op[p].a;
                const int epi = chunk->op[p].epi;
                if (a.push != nullptr && (epi == EPI_RESID || epi == EPI_SWIGLU)) {
                    const int n_pairs = chunk->op[p].n_pairs;
                    const int total_rows = a.rows0 + a.rows1 + a.rows2;
                    const int e = a.push_ctl[kCtlEpoch] + a.push_gi;
                    const int uk = 2 * blockIdx.x + gh + gk * vgrid;
                    if (uk < n_pairs) {
                        const float *sv = stash + (((p & 1) * 2 + gh) * kEngUnits + gk) * 2;
                        if (epi == EPI_SWIGLU) {
                            p2p_ll_push(a.push, e, (size_t)uk, sv[0]);
                        } else {
                            p2p_ll_push(a.push, e, (size_t)(2 * uk), sv[0]);
                            if (2 * uk + 1 < total_rows) p2p_ll_push(a.push, e, (size_t)(2 * uk + 1), sv[1]);
                        }
                    }
                }
                p++;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                if (lane == 0) ctrl[EC_PUB] = p;
                progressed = true;
            }
            // ---- stage x of mat-vec g: its buffer is free once the streaming waves are done with mat-vec g - 2
            // (p >= g: this block's own words of the vector are out -- a sweep that waited for them would wait for itself)
            if (g < n_ops && p >= g && ctrl[EC_DONE] >= 2 * (g - 1)) {
                const MatvecArgs &a = chunk->op[g].a;
                const int pro = chunk->op[g].pro, epi = chunk->op[g].epi;
                const int n = a.n, n4 = n >> 2;
                const int n4_pad = ((n4 + kBlock * kEngU - 1) / (kBlock * kEngU)) * (kBlock * kEngU);
                v4f *xs4 = (v4f *)(xs_base + (size_t)(g & 1) * xs_floats);
                const v4f zero = {0.f, 0.f, 0.f, 0.f};
                bool ready = true;
                LLPoll lp = {};
                const bool ll = a.xin.slots != nullptr;
                if (ll) {
                    lp = ll_poll_init(a.xin);
                    // the gate: 16 producer blocks' last words (kernel_common.h ll_hint_wait, one poll per visit)
                    if (a.xin.hint_n != 0) {
                        const unsigned step = a.xin.hint_n >= kHintLanes ? a.xin.hint_n / kHintLanes : 1u;
                        const unsigned idx = a.xin.hint0 + ((blockIdx.x + (lane & (kHintLanes - 1)) * step) % a.xin.hint_n) * a.xin.hint_stride;
                        const unsigned long long w = __hip_atomic_load(lp.slot + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        ready = __all((unsigned)(w >> 32) == lp.e);
                    }
                }
                if (ready) {
                    // per-unit epilogue inputs of this block's units (lane = (half, k)): residual words / values, RoPE pair
                    const int n_pairs = chunk->op[g].n_pairs;
                    const int uk = 2 * blockIdx.x + gh + gk * vgrid;
                    v4u ein = {0u, 0u, 0u, 0u};
                    if (uk < n_pairs) {
                        if (epi == EPI_RESID) {
                            if (a.resid_in.slots != nullptr) {
                                const int e = a.resid_in.ctl[kCtlEpoch] + a.resid_in.gi;
                                ein = ll_load2(a.resid_in.slots + (size_t)(e & 1) * a.resid_in.slot_floats, (size_t)(2 * uk));
                            } else {
                                const int total_rows = a.rows0 + a.rows1 + a.rows2;
                                const L2Z_G float *rp = as_g(a.resid);
                                ein.x = __float_as_uint(rp[2 * uk]);
                                ein.z = 2 * uk + 1 < total_rows ? __float_as_uint(rp[2 * uk + 1]) : 0u;
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
                    // x: lane L holds float4 L + 64 i
                    constexpr int RG = 8;   // float4 per lane per round (16 loads of 16 bytes in flight in the LL form)
                    for (int j0 = lane; j0 < n4_pad; j0 += 64 * RG) {
                        if (ll) {
                            v4u w[2 * RG];
#pragma unroll
                            for (int i = 0; i < RG; i++) {
                                const int j = j0 + 64 * i;
                                const int jc = j < n4 ? j : 0;
                                w[2 * i] = ll_load2(lp.slot, (size_t)4 * jc);
                                w[2 * i + 1] = ll_load2(lp.slot, (size_t)4 * jc + 2);
                            }
#pragma unroll
                            for (int i = 0; i < RG; i++) {
                                const int j = j0 + 64 * i;
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
                                const int j = j0 + 64 * i;
                                v[i] = j < n4 ? x4[j] : zero;
                            }
#pragma unroll
                            for (int i = 0; i < RG; i++) {
                                const int j = j0 + 64 * i;
                                if (j < n4_pad) xs4[j] = v[i];
                            }
                        }
                    }
                    if (pro == PRO_RMS) {
                        // main.zig:432-468 in the 256-thread kernels' order: thread t sums float4 t, t + 256, ... (fmaf per
                        // component), wave sum, the four wave partials added in wave order.  Lane L plays t = L + 64 v.
                        float pv[kWaves];
#pragma unroll
                        for (int v = 0; v < kWaves; v++) {
                            float ss = 0.0f;
                            for (int j = lane + 64 * v; j < n4; j += kBlock) {
                                const v4f x = xs4[j];
                                ss = fmaf(x.x, x.x, ss);
                                ss = fmaf(x.y, x.y, ss);
                                ss = fmaf(x.z, x.z, ss);
                                ss = fmaf(x.w, x.w, ss);
                            }
                            pv[v] = wave_sum(ss);
                        }
                        float tot = pv[0];
#pragma unroll
                        for (int v = 1; v < kWaves; v++) tot += pv[v];
                        float s = tot / (float)n;  // :452
                        s += 1e-5f;                // :453
                        s = 1.0f / sqrtf(s);       // :454
                        const L2Z_G v4f *g4 = (const L2Z_G v4f *)a.rms_w;
                        for (int j = lane; j < n4; j += 64) {
                            v4f x = xs4[j];
                            const v4f gw = g4[j];
                            x.x = (x.x * s) * gw.x;  // :462
                            x.y = (x.y * s) * gw.y;
                            x.z = (x.z * s) * gw.z;
                            x.w = (x.w * s) * gw.w;
                            xs4[j] = x;
                        }
                    }
                    epin[((g & 1) * 2 + gh) * kEngUnits + gk] = ein;
                    if (lane == 0) {
                        MvLocals mg = eng_locals(a, epi);
                        mg.push = nullptr;  // published by this wave from the stash
                        lmc[g & 1] = mg;
                    }
                    g++;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) ctrl[EC_READY] = g;
                    progressed = true;
                }
            }
            if (!progressed) {
                if (ctrl[EC_ERR]) break;
                __builtin_amdgcn_s_sleep(4);
            }
        }
        return;
    }

    // =============================================================================================================
    // The streaming waves.  Flat sequence of weight batches over (mat-vec, unit, batch); the issue cursor runs R
    // batches ahead of the consume cursor, across mat-vec boundaries.
    // =============================================================================================================
    const int half = __builtin_amdgcn_readfirstlane(tid >> 8);  // wave-uniform: keep what derives from it in SGPRs
    const int ht = tid & (kBlock - 1), hw = __builtin_amdgcn_readfirstlane(ht >> 6);
    const int vb = 2 * blockIdx.x + half;

    // issue cursor
    int iop = 0, iu = vb, ib = 0;
    MvLocals mi = eng_locals(chunk->op[0].a, chunk->op[0].epi);
    int i_epi = chunk->op[0].epi, i_nb = chunk->op[0].nb, i_n4 = mi.n >> 2;
    const L2Z_G v4f *dummy4 = (const L2Z_G v4f *)chunk->dummy;
    const float *pa, *pb;  // generic to the compiler (read out of the chunk description): cast to L2Z_G where they are loaded through
    eng_pair_rows(mi, i_epi, iu, pa, pb);
    // consume cursor
    int cop = 0, cu = vb, cb = 0, ck = 0;
    int c_epi = i_epi, c_nb = i_nb, c_pairs = mi.n_pairs;
    const v4f *xs4 = (const v4f *)xs_base;

    v4f wa[kEngR][kEngU], wb[kEngR][kEngU];
    bool i_valid = true;

    auto issue = [&](v4f (&ra)[kEngU], v4f (&rb)[kEngU]) {
        const int cbase = ib * (kBlock * kEngU);
        const L2Z_G v4f *a4 = (const L2Z_G v4f *)pa + cbase + ht, *b4 = (const L2Z_G v4f *)pb + cbase + ht;
        const int wbase = cbase + (ht & ~63);
        // Steps past the row end (a row's last, partly filled batch; their x is the zero padding) read a fixed, cache
        // resident, finite 1 KB instead (dummy): no branch around a load -- the ring's waits count the loads issued
        // since -- and no HBM traffic (matvec_row_kernel skips those loads, which a counted ring cannot).
#pragma unroll
        for (int k = 0; k < kEngU; k++) {
            const bool in_row = wbase + kBlock * k < i_n4;  // wave-uniform
            const L2Z_G v4f *sa_ = in_row ? a4 + kBlock * k : dummy4 + lane;
            const L2Z_G v4f *sb_ = in_row ? b4 + kBlock * k : dummy4 + 64 + lane;
            ra[k] = ldg_nt(sa_);
            rb[k] = ldg_nt(sb_);
        }
        // advance (block-uniform: half 0's unit decides where a mat-vec ends)
        if (++ib == i_nb) {
            ib = 0;
            iu += vgrid;
            if (iu - half >= mi.n_pairs) {
                iop++;
                if (iop < n_ops) {
                    i_epi = chunk->op[iop].epi;
                    mi = eng_locals(chunk->op[iop].a, i_epi);
                    i_nb = chunk->op[iop].nb; i_n4 = mi.n >> 2;
                    iu = vb;
                } else {
                    i_valid = false;
                }
            }
            if (i_valid) eng_pair_rows(mi, i_epi, iu, pa, pb);  // clamped to the last pair for a half without this unit
        }
    };

    // software barrier among the eight streaming waves (monotonic counter)
    int bar_target = 0;
    auto stream_barrier = [&]() {
        bar_target += 8;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_fetch_add((int *)&ctrl[EC_BAR], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load((int *)&ctrl[EC_BAR], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < bar_target) __builtin_amdgcn_s_sleep(0);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto wait_word = [&](int word, int target, int code) {
        if (ctrl[word] >= target) return;
        const long long t0 = wall_clock64();
        while (ctrl[word] < target) {
            if (eng_spin_failed(ctrl, t0, timeout, g_ctl, h_err, code)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };

    // fill the ring, then wait for x of the first mat-vec
#pragma unroll
    for (int s = 0; s < kEngR; s++) issue(wa[s], wb[s]);  // a chunk has more than R batches per half (launch_engine's callers)
    wait_word(EC_READY, 1, 1);

    v4f acc_a = {0.f, 0.f, 0.f, 0.f}, acc_b = {0.f, 0.f, 0.f, 0.f};
    float best_v = -INFINITY;
    int best_i = 0x7fffffff;
    int parity = 0;
    bool c_valid = true;
    (void)c_valid;

    auto step = [&](auto refill, v4f (&ra)[kEngU], v4f (&rb)[kEngU]) {
        // consume
#pragma unroll
        for (int k = 0; k < kEngU; k++) {
            const v4f xv = xs4[cb * (kBlock * kEngU) + ht + kBlock * k];
            acc_a = fma4(ra[k], xv, acc_a);
            acc_b = fma4(rb[k], xv, acc_b);
        }
        // refill this slot (the main loop: always -- a conditional issue would make every wait on the ring a full drain)
        if constexpr (decltype(refill)::value) issue(ra, rb);
        if (++cb < c_nb) return;
        // ---- unit done
        cb = 0;
        const float sa = wave_sum(hsum4(acc_a));
        const float sb = wave_sum(hsum4(acc_b));
        float *pp = part + (half * 2 + parity) * (2 * kWaves);
        if (lane == 0) {
            pp[hw] = sa;
            pp[kWaves + hw] = sb;
        }
        stream_barrier();
        if (ht == 0 && cu < c_pairs) {
            const MvLocals me = lmc[cop & 1];
            const float ta = ((pp[0] + pp[1]) + pp[2]) + pp[3];
            const float tb = ((pp[kWaves] + pp[kWaves + 1]) + pp[kWaves + 2]) + pp[kWaves + 3];
            const v4u ev = epin[((cop & 1) * 2 + half) * kEngUnits + ck];
            EpiIn ein;
            ein.ra = __uint_as_float(ev.x); ein.rb = __uint_as_float(ev.z);
            ein.cs = make_float2(__uint_as_float(ev.x), __uint_as_float(ev.y));
            ein.rw = ev;
            float *st = stash + (((cop & 1) * 2 + half) * kEngUnits + ck) * 2;
            if (c_epi == EPI_ROPE) pair_epilogue<EPI_ROPE>(me, cu, ta, tb, true, ein, nullptr);
            else if (c_epi == EPI_RESID) pair_epilogue<EPI_RESID>(me, cu, ta, tb, true, ein, st);
            else if (c_epi == EPI_SWIGLU) pair_epilogue<EPI_SWIGLU>(me, cu, ta, tb, true, ein, st);
            else {
                pair_epilogue<EPI_ARGMAX>(me, cu, ta, tb, true, ein, nullptr);
                const int ra_ = 2 * cu, rb_ = ra_ + 1;
                if (ta > best_v || best_i == 0x7fffffff) { best_v = ta; best_i = ra_; }
                if (rb_ < me.total_rows && tb > best_v) { best_v = tb; best_i = rb_; }
            }
        }
        parity ^= 1;
        acc_a = v4f{0.f, 0.f, 0.f, 0.f};
        acc_b = v4f{0.f, 0.f, 0.f, 0.f};
        cu += vgrid;
        ck++;
        if (cu - half < c_pairs) return;
        // ---- mat-vec done for this block: tell the gatherer, move on
        if (ht == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __hip_atomic_fetch_add((int *)&ctrl[EC_DONE], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        cop++;
        if (cop >= n_ops) {
            c_valid = false;
            return;
        }
        c_epi = chunk->op[cop].epi;
        c_pairs = chunk->op[cop].n_pairs;
        c_nb = chunk->op[cop].nb;
        cu = vb; ck = 0;
        xs4 = (const v4f *)(xs_base + (size_t)(cop & 1) * xs_floats);
        wait_word(EC_READY, cop + 1, 2 + cop);   // x of this mat-vec staged (and its epilogue inputs)
        if (cop >= 2) wait_word(EC_PUB, cop - 1, 8 + cop);  // the stash of this parity has been published
    };

    static_assert(kEngR == 3, "the main loop and its drain are written out for a ring of three");
    constexpr std::true_type kRefill{};
    constexpr std::false_type kDrain{};
    int phase;
    for (;;) {  // i_valid goes false inside the step that issues the chunk's last batch: R batches are left, in ring order
        step(kRefill, wa[0], wb[0]); if (!i_valid) { phase = 0; break; }
        step(kRefill, wa[1], wb[1]); if (!i_valid) { phase = 1; break; }
        step(kRefill, wa[2], wb[2]); if (!i_valid) { phase = 2; break; }
    }
    if (phase == 0) { step(kDrain, wa[1], wb[1]); step(kDrain, wa[2], wb[2]); step(kDrain, wa[0], wb[0]); }
    else if (phase == 1) { step(kDrain, wa[2], wb[2]); step(kDrain, wa[0], wb[0]); step(kDrain, wa[1], wb[1]); }
    else { step(kDrain, wa[0], wb[0]); step(kDrain, wa[1], wb[1]); step(kDrain, wa[2], wb[2]); }
    if (c_epi == EPI_ARGMAX && ht == 0) {  // the chunk ended with the classifier: one candidate per virtual block
        const MatvecArgs &a = chunk->op[n_ops - 1].a;
        a.part_val[vb] = best_v;
        a.part_idx[vb] = best_i == 0x7fffffff ? best_i : best_i + a.row_offset;
    }
}

}  // namespace

size_t engine_lds_bytes(int xs_floats)
{
    return (size_t)(2 * (size_t)xs_floats + EC_WORDS + 4 * (2 * kWaves) + 2 * 2 * kEngUnits * 2 + 4 * (2 * 2 * kEngUnits) + 16) * sizeof(float) + 2 * sizeof(MvLocals);
}

// x buffer size (floats) for a chunk whose widest mat-vec has n columns
int engine_xs_floats(int n_max)
{
    const int n4 = n_max >> 2;
    return 4 * (((n4 + kBlock * kEngU - 1) / (kBlock * kEngU)) * (kBlock * kEngU));
}

// units per half a mat-vec of n_pairs pairs has on a grid of `grid` blocks (0: the engine cannot take it)
bool engine_units_ok(int n_pairs, int grid)
{
    const int vgrid = 2 * grid;
    return vgrid >= 2 && vgrid <= n_pairs && (n_pairs + vgrid - 1) / vgrid <= kEngUnits;
}

hipError_t launch_engine(const EngChunk *d_chunk, int grid, int xs_floats, hipStream_t st)
{
    const size_t lds = engine_lds_bytes(xs_floats);
    hipError_t e = ensure_lds(engine_kernel, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(engine_kernel, dim3(grid), dim3(kEngThreads), lds, st, d_chunk, xs_floats);
    return hipGetLastError();
}

}  // namespace l2z
