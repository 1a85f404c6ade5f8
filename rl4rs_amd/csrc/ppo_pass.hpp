// One PPO SGD pass (all minibatches: forward, loss, backward, gradient reductions, Adam) as ONE persistent kernel.
// Included by policy.hip inside namespace rl4rs (needs PolDims, LossArgs, policy_row_loss, wave_max / wave_sum, f32x16).
//
// Reference: the SGD loop of RLlib 1.5.1's PPO as script/modelfree_train.py:179-247 configures it (sgd_minibatch_size 256,
// num_sgd_iter passes over the train batch; third-party, parity unpinned - checked against the fp64 restatement in tests).
//
// The per-minibatch path (k_policy_train + k_gemm_tn + ... in policy.hip) is a chain of 7 small dependent kernels (144 x 7 per
// pass at RLlib's 256-sample minibatches) and is bound by that serial depth.  Here MB / R workgroups of 8 waves stay resident for
// the whole pass and meet at two grid barriers per minibatch:
//   phase A  workgroup g owns samples [R g, R g + R) of the minibatch: obs tile -> LDS, hidden = tanh(obs W1 + b1) and
//            out = h W2e + b2e on 32x32x2 fp32 MFMA tiles (layer 1 split over K across waves, partials summed in a fixed
//            order), the A2C/PPO row loss (policy_row_loss, one wave per row), dH = dOut W2e^T (B operand from the transposed
//            copy w2t), dHpre = dH (1 - h^2); H, dOut, dHpre go to a [MB, .] scratch
//   barrier
//   phase B  45 tasks over groups of 4 waves: the 32x32 tiles of dW1 = obs^T dHpre and dW2e = H^T dOut (sample-axis MFMA
//            reductions) and the bias column sums, each followed by the Adam update of exactly the parameters it produced
//            (no separate gradient / Adam kernels, no clip: the caller falls back to the per-minibatch path when grad_clip > 0)
//   barrier
//
// Round 6: the pass is a chain of dependent memory round trips (weights written by OTHER workgroups one barrier ago can only come
// from the memory side: ~1.5 - 2.7 us each), not of arithmetic - 31.3 us per minibatch of which the MFMAs are ~5.  What changed:
//   * STD = the shape everything in BASELINE.json runs (256 -> 64 -> 284 + 1, 9 mask words, 8 rows per workgroup, minibatch a
//     multiple of 256) is a compile-time instantiation: addresses are scalar base + 32-bit lane offset with immediates, no integer
//     divisions, and the kernel fits its 256 registers without the 27 spills of the all-runtime form (each spilled address was a
//     scratch reload + s_waitcnt vmcnt(0) in front of its load: twelve serialised round trips per minibatch, tools/isa_wait_scan.py)
//   * every weight value phase A needs that nobody rewrites during phase A is requested in ONE round trip at the top of the phase
//     (W1 slice, both W2e tiles of the wave, both biases) and the w2t slice of the dH stage behind layer 2's MFMAs, one stage ahead
//     of its use: layer 2 and dH no longer start with a round trip of their own
//   * Adam's operands never come from memory: the task -> workgroup map is the same for every minibatch, so a wave keeps the
//     parameters and both moments of its tile slice in 12 registers for the whole pass (single-trip task maps only)
//   * the next minibatch's observation rows / old logits / mask words / scalars are requested behind phase B's tile loads instead
//     of inside the grid barrier, where they sat in front of the polling wave's own loads (vmcnt retires in order)
//   * the barrier polls both status words with ONE 64-bit load per iteration (was two dependent round trips)
// All of it leaves every sum in the order it had: results are bit-identical to round 5's kernel.
#pragma once

struct PassArgs {
    PolDims d;
    int N, MB, rows;
    float *prm, *am, *av;
    float* w2t;              // [AE, HID] transposed copy of W2e, kept current by the Adam updates (coalesced dH operand)
    const float* obs;
    const uint32_t* mask;
    LossArgs L;
    float *H, *dOut, *dHpre;
    float4* terms;
    float* grad;
    unsigned* bar;           // bar[0] arrival counter, bar[1] sticky "a barrier timed out" (8-byte aligned: polled as one word)
    unsigned* dead_host;     // pinned host word raised next to bar[1]: the host sees a timeout without synchronising
    float lr, b1, b2, eps;
    long long t0;
    int mb_begin, mb_end;    // minibatches [mb_begin, mb_end) of the pass
    int apply;               // 1: Adam inside phase B (single GPU); 0: gradient only (the caller all-reduces it, then rl4rs_policy_adam_step)
    unsigned long long* trace;
};

#ifdef RL4RS_PASS_TRACE     // s_memtime marks of workgroup 0 / wave 0 in minibatch 10 (timing experiments)
#define RL4RS_PT(k) do { if (a.trace && blockIdx.x == 0 && tid == 0 && mb == 10) a.trace[k] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RL4RS_PT(k) do { } while (0)
#endif

// Grid barrier (all workgroups resident): monotonically increasing arrival counter.  Producer side: every wave drains its
// stores (everything another workgroup reads is stored write-through, store_wt: no L2 write-back fence needed), and - the compiler
// may drop the wait when it believes nothing is outstanding - an explicit s_waitcnt before the RELAXED arrival, so the counter
// cannot overtake the payload.  Consumer side: relaxed polling of (counter, dead flag) as ONE 8-byte load, ONE agent acquire
// (invalidates this CU's L1), workgroup barrier, then plain loads.
// Returns false once any workgroup has given up (bar[1] != 0): the caller then stops touching the parameters, so a pass whose
// workgroups were not co-resident leaves them at the last consistent minibatch instead of running racy updates.
#ifndef RL4RS_PASS_WT
#define RL4RS_PASS_WT 1      // 1: everything another workgroup reads is stored write-through (store_wt) and the barrier has no release
#endif                       //    fence; 0: plain stores + lane-0 agent release (L2 write-back) before the arrival
#if RL4RS_PASS_WT
#define PUB(ptr, val) store_wt(&(ptr), (val))
#else
#define PUB(ptr, val) ((ptr) = (val))
#endif
// `mid` runs on every thread between the arrival and the poll: the place to REQUEST data that does not depend on what the
// other workgroups publish - the loads fly while the barrier waits (used once per pass, for the first minibatch's inputs).
struct NoMid { __device__ void operator()() const {} };
// DRAINED: the caller has already waited for every store another workgroup reads (each wave, before the call) and may still have
// PRIVATE stores in flight behind them - the barrier then adds no wait of its own.
//
// Arrival, grids of up to 126 workgroups (RL4RS_PASS_FLAGS): every workgroup publishes its generation number in a word of ITS OWN
// (bar[64 + workgroup], write-through) and wave 0 looks at all of them with ONE 8-byte load per lane and poll (lane l reads the words
// of workgroups 2l and 2l + 1, lane 63 the timeout flag) - arrivals no longer queue up as read-modify-writes of one address.  Larger grids: the counter.
#ifndef RL4RS_PASS_FLAGS
#define RL4RS_PASS_FLAGS 1
#endif
template <typename Mid = NoMid, bool DRAINED = false>
__device__ __forceinline__ bool grid_barrier(unsigned* bar, unsigned* dead_host, unsigned nwg, unsigned& gen, Mid mid = Mid()) {
    __shared__ unsigned s_dead;
    if (!DRAINED || !RL4RS_PASS_WT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    gen += 1;
    const bool flags = RL4RS_PASS_FLAGS && RL4RS_PASS_WT && nwg <= 126u;       // (a second pair per lane for grids up to 252 measured slower than the counter at 128: 29.9 vs 28.3 us)
    if (threadIdx.x == 0) {
#if !RL4RS_PASS_WT
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        if (!DRAINED || !RL4RS_PASS_WT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (flags) __hip_atomic_store(bar + 64 + blockIdx.x, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    mid();
    if (flags) {
        if (threadIdx.x < 64) {
            // bounded wait (~seconds): if the workgroups are not all resident the pass gives up instead of hanging the device
            // lane l < 63 looks at the words of workgroups 2l and 2l + 1 (one 8-byte load), lane 63 at (counter, timeout flag)
            const unsigned lane = threadIdx.x;
            unsigned long long* word = reinterpret_cast<unsigned long long*>(lane == 63u ? bar : bar + 64 + 2 * lane);
            unsigned spins = 0, dead = 0;
            for (;;) {
                const unsigned long long v2 = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned lo = (unsigned)v2, hi = (unsigned)(v2 >> 32);
                const bool here = lane == 63u || ((2 * lane >= nwg || lo >= gen) && (2 * lane + 1 >= nwg || hi >= gen));
                dead = __builtin_amdgcn_readlane(lane == 63u ? hi : 0u, 63);
                if (__builtin_amdgcn_ballot_w64(here) == ~0ull || dead) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1u << 21)) {
                    if (lane == 0) {
                        __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (dead_host) __hip_atomic_store(dead_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
            if (lane == 0) s_dead = dead;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    } else if (threadIdx.x == 0) {
        const unsigned target = gen * nwg;
        // bounded wait (~seconds): if the workgroups are not all resident (a tool that serialises workgroups, a shared GPU)
        // the pass gives up instead of hanging the device - bar[1] is raised and rl4rs_policy_ppo_epoch reports it
        unsigned long long* bar2 = reinterpret_cast<unsigned long long*>(bar);
        unsigned spins = 0;
        unsigned long long w = __hip_atomic_load(bar2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((unsigned)w < target && (unsigned)(w >> 32) == 0u) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 21)) {   // ~1 us per poll: gives up after a few seconds
                __hip_atomic_store(bar + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dead_host) __hip_atomic_store(dead_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            w = __hip_atomic_load(bar2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_dead = (unsigned)(w >> 32);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return s_dead == 0u;
}

#define RL4RS_U(x) ((unsigned)(x))
// Every load of the pass is a raw buffer load: descriptor (4 SGPRs) + ONE 32-bit lane offset per operand + a scalar / immediate
// offset per element.  With flat pointers the compiler formed a 64-bit vector address per 4 KB window of every operand, hoisted
// them out of the minibatch loop and spilled them: each reload was a scratch_load + s_waitcnt vmcnt(0) in front of its load.
// Out-of-range offsets return 0 (no clamps in the lane offsets of ragged edges).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t pass_rsrc(const void* p, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bld(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}

// one K slice of [8 rows] x [64 columns] on 4x4x1 MFMAs: A rows from LDS (16-byte reads of 4 consecutive k of row lane % 4 of each
// 4-row tile), NQ quads of k, B = bw[k] (the wave's register per k: W[k][lane]); two accumulator chains per row tile
// MT = row tiles (2: rows 0 .. 7; 1: rows 0 .. 3 only - the 4-row workgroups of k_ppo_pass<true, 4>: o1 stays zero, four chains on the one tile)
template <int NQ, int MT = 2>
__device__ __forceinline__ void std_slice(const float* arow, int stride, const float (&bw)[4 * NQ], f32x4_t& o0, f32x4_t& o1) {
    f32x4_t c00 = {0.f, 0.f, 0.f, 0.f}, c01 = c00, c10 = c00, c11 = c00;
#pragma unroll
    for (int qd = 0; qd < NQ; ++qd) {
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(arow + 4 * qd);
        if constexpr (MT == 2) {
            const f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(arow + 4 * stride + 4 * qd);
#pragma unroll
            for (int jj = 0; jj < 4; jj += 2) {
                c00 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[jj], bw[4 * qd + jj], c00, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[jj], bw[4 * qd + jj], c10, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[jj + 1], bw[4 * qd + jj + 1], c01, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1[jj + 1], bw[4 * qd + jj + 1], c11, 0, 0, 0);
            }
        } else {
            c00 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[0], bw[4 * qd + 0], c00, 0, 0, 0);
            c01 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[1], bw[4 * qd + 1], c01, 0, 0, 0);
            c10 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[2], bw[4 * qd + 2], c10, 0, 0, 0);
            c11 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0[3], bw[4 * qd + 3], c11, 0, 0, 0);
        }
    }
    if constexpr (MT == 2) {
        o0 = c00 + c01;
        o1 = c10 + c11;
    } else {
        o0 = (c00 + c01) + (c10 + c11);
        o1 = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}
// register i of lane l = out[row i][column l]
template <int MT = 2>
__device__ __forceinline__ void std_put(float* dst, int ld, int lane, const f32x4_t& o0, const f32x4_t& o1) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dst[i * ld + lane] = o0[i];
        if (MT == 2) dst[(4 + i) * ld + lane] = o1[i];
    }
}


// STD: OD = 256, HID = 64, A = 284 (AE = 285, W = 9), R = 8 rows per workgroup, MB % 256 == 0 - everything constant folds.
// SR (STD only): rows per workgroup, 8 or 4.  Four rows = one 4-row MFMA tile per stage (half the matrix work of a workgroup) and one
// row loss per SIMD instead of two sharing it; the LDS layout stays that of eight rows (rows 4 .. 7 unused).
template <bool STD, int SR = 8>
__global__ __launch_bounds__(512) void k_ppo_pass(PassArgs a) {
    static_assert(SR == 8 || (STD && SR == 4), "rows per workgroup of the compile-time form");
    constexpr int MT = SR / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PolDims d = a.d;
    if (STD) { d.OD = 256; d.HID = 64; d.A = 284; d.AE = 285; d.W = 9; }
    const int OD = d.OD, HID = d.HID, AE = d.AE;
    const int MB = a.MB;
    const int SO = OD | 1, SH = HID | 1, SA = AE | 1;               // odd row strides: conflict-free column walks
    float* s_obs = reinterpret_cast<float*>(smem);                   // [32][SO]
    float* s_h = s_obs + 32 * SO;                                    // [32][SH]
    float* s_out = s_h + 32 * SH;                                    // [32][SA]  (scratch for the dH partials afterwards)
    float* s_d = s_out + 32 * SA;                                    // [32][SA]  (scratch for the layer-1 partials before)
    float* s_old = s_d + 32 * SA;                                    // [32][A]   old logits of the tile's rows
    float* s_sc = s_old + 32 * d.A;                                  // [5][32]   action (as int), adv, ret, old logp, old value
    uint32_t* s_mask = reinterpret_cast<uint32_t*>(s_sc + 5 * 32);   // [32][W]   action-mask words of the tile's rows
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: every role / task decision below is a scalar branch
    const int n_prm = OD * HID + HID + HID * AE + AE;
    const int o_b1 = OD * HID, o_w2 = o_b1 + HID, o_b2 = o_w2 + HID * AE;      // float offsets inside the parameter buffer
    const __amdgpu_buffer_rsrc_t rs_prm = pass_rsrc(a.prm, (size_t)n_prm * 4), rs_am = pass_rsrc(a.am, (size_t)n_prm * 4),
                                 rs_av = pass_rsrc(a.av, (size_t)n_prm * 4), rs_w2t = pass_rsrc(a.w2t, (size_t)HID * AE * 4);
    const int NT1 = HID / 32, NT2 = (AE + 31) / 32, parts = 8 / NT1;
    const int R = STD ? SR : a.rows;         // samples of this workgroup (8, 16 or 32; compile-time form: SR): rows R..31 of every 32x32x2 tile are idle
    const int r0 = blockIdx.x * R;
    const int t1 = wave % NT1, q1 = wave / NT1;                      // layer-1 / dH role of the wave: column tile, K part
    unsigned gen = 0;
    {   // w2t = W2e^T (kept current by phase B afterwards): four requests per thread in flight, then the stores
        const unsigned stride = gridDim.x * 512, n_w2 = RL4RS_U(HID * AE);
        for (unsigned i0 = blockIdx.x * 512 + tid; i0 < n_w2; i0 += 4 * stride) {
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = bld(rs_prm, (int)(i0 * 4), (int)((RL4RS_U(o_w2) + u * stride) * 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned i = i0 + u * stride;
                if (i < n_w2) {
                    const unsigned j = i / RL4RS_U(AE), c = i - j * RL4RS_U(AE);
                    PUB(a.w2t[c * RL4RS_U(HID) + j], x[u]);
                }
            }
        }
    }
    // The inputs of a minibatch that no workgroup writes (observation rows, old logits, mask words, per-sample scalars) are
    // requested a phase ahead (behind phase B's tile loads; the first minibatch's inside the first barrier) and go to LDS from
    // registers at the top of phase A.  Only when they fit a few registers per thread (8 rows per workgroup: 4 + 5 + 1 + 5).
    constexpr int PF_OBS = 4, PF_OLD = 5;
    const bool can_pf = STD || (R * OD <= 512 * PF_OBS && R * d.A <= 512 * PF_OLD && R * d.W <= 512);
    float pf_obs[PF_OBS], pf_old[PF_OLD], pf_sc[4];
    uint32_t pf_mask = 0;
    int pf_act = 0;
    auto prefetch = [&](int mbn) {
        if (!can_pf || mbn >= a.mb_end) return;
        const size_t ln = (size_t)mbn * MB + r0;                                     // scalar
        // descriptors over exactly this workgroup's rows: a lane past the end reads 0 (and does not store it)
        const __amdgpu_buffer_rsrc_t rs_o = pass_rsrc(a.obs + ln * OD, (size_t)R * OD * 4), rs_l = pass_rsrc(a.L.old_logits + ln * d.A, (size_t)R * d.A * 4);
#pragma unroll
        for (int u = 0; u < (STD ? PF_OBS * SR / 8 : PF_OBS); ++u) pf_obs[u] = bld(rs_o, tid * 4, 2048 * u);
#pragma unroll
        for (int u = 0; u < ((STD && SR == 4) ? 3 : PF_OLD); ++u) pf_old[u] = bld(rs_l, tid * 4, 2048 * u);
        if (a.mask) pf_mask = __builtin_amdgcn_raw_buffer_load_b32(pass_rsrc(a.mask + ln * d.W, (size_t)R * d.W * 4), tid * 4, 0, 0);
        pf_act = (int)__builtin_amdgcn_raw_buffer_load_b32(pass_rsrc(a.L.actions + ln, (size_t)R * 4), tid * 4, 0, 0);
        pf_sc[0] = bld(pass_rsrc(a.L.adv + ln, (size_t)R * 4), tid * 4, 0);
        pf_sc[1] = bld(pass_rsrc(a.L.ret + ln, (size_t)R * 4), tid * 4, 0);
        pf_sc[2] = bld(pass_rsrc(a.L.old_logp + ln, (size_t)R * 4), tid * 4, 0);
        pf_sc[3] = bld(pass_rsrc(a.L.old_value + ln, (size_t)R * 4), tid * 4, 0);
    };
    // weight slices in MFMA B-operand order (32x32x2: lane = column li, k parity half): 32 rows of k per trip
    const int v_w1 = (half * HID + t1 * 32 + li) * 4;                                // lane offsets (bytes)
    auto load_w1 = [&](int k, float (&bv)[32]) {
#pragma unroll
        for (int u = 0; u < 32; ++u) bv[u] = bld(rs_prm, v_w1, (k + 2 * u) * HID * 4);
    };
    auto load_w2 = [&](int t, int k, float (&bv)[32], float& bias) {
        const int v = (half * AE + min(t * 32 + li, AE - 1)) * 4;                   // clamped column: the loads stay unconditional
#pragma unroll
        for (int u = 0; u < 32; ++u) bv[u] = bld(rs_prm, v, (o_w2 + (k + 2 * u) * AE) * 4);
        if (k == 0) bias = bld(rs_prm, min(t * 32 + li, AE - 1) * 4, o_b2 * 4);
    };
    const int kper_h = ((AE + parts - 1) / parts + 1) & ~1;                          // dH: even number of k per part
    auto load_w2t = [&](int k, float (&wv)[36]) {                                    // (rows k >= AE are past the buffer: 0)
#pragma unroll
        for (int u = 0; u < 36; ++u) wv[u] = bld(rs_w2t, v_w1, (k + 2 * u) * HID * 4);
    };

    // ---- phase B's task map.  One task = one 32x32 gradient tile or 32 bias columns + the Adam update of its parameters, done by a
    // group of NW waves of one workgroup: each wave reduces 1 / NW of the minibatch's samples, the partials meet in LDS and are
    // summed in a fixed order, then each wave updates 1 / NW of the tile's parameters.  All of it scalar.
    //   generic: NW = 4, two groups (= two tasks) per workgroup, grid = the MB / R workgroups of phase A
    //   STD:     NW = 8, ONE task per workgroup, grid = max(MB / 8, 45): at MB = 256 thirteen extra workgroups do nothing but phase B.
    //            With two tile tasks on 17 of 32 workgroups and 15 idle, phase B's critical path was two waves per SIMD x 32 MFMAs x
    //            64 cycles (2 us) behind TWO dependent round trips of 32 requests per wave; one tile on eight waves is 16 sample pairs
    //            per wave = one round trip of 32 requests, then 16 MFMAs per wave.
    constexpr int NW = STD ? 8 : 4, NG = 8 / NW, NR = 16 / NW;            // waves per task, tasks per workgroup and trip, accumulator registers per wave
    const int n_t1 = (OD / 32) * NT1, n_t2 = NT1 * NT2, total = n_t1 + n_t2 + NT1 + NT2;
    const int grp = STD ? 0 : wave >> 2, q = STD ? wave : wave & 3;
    const int spq = MB / NW;                                              // samples per wave of a group
    // accumulator register r of a 32x32x2 tile holds row (r & 3) + 8 (r >> 2) + 4 half; wave q owns r = NR q .. NR q + NR - 1 = rows rowbase + c (+ 4 half)
    const int rowbase = STD ? 2 * (q & 1) + 8 * (q >> 1) : 8 * q;
    const int n_a = MB / R;                                               // workgroups that have rows of the minibatch (phase A)
    const bool is_a = (int)blockIdx.x < n_a;
    struct Task { bool live, is_tile, first; int tm, tn, Nc, s_e; };
    auto map_task = [&](int task) {
        Task T;
        T.live = task < total;
        T.is_tile = task < n_t1 + n_t2;
        T.first = T.is_tile ? task < n_t1 : (task - n_t1 - n_t2) < NT1;
        T.tm = 0;
        if (T.is_tile) {
            const int tl = T.first ? task : task - n_t1;
            const int tn_n = T.first ? NT1 : NT2;
            T.tm = tl / tn_n;
            T.tn = tl - T.tm * tn_n;
        } else {
            const int tl = task - n_t1 - n_t2;
            T.tn = T.first ? tl : tl - NT1;
        }
        T.Nc = T.first ? HID : AE;
        // scalar part of the parameter offset (floats); lane part: tiles (4 half) * Nc + li [+ c * Nc], bias columns li
        if (T.is_tile) T.s_e = (T.first ? 0 : o_w2) + (T.tm * 32 + rowbase) * T.Nc + T.tn * 32;
        else T.s_e = (T.first ? o_b1 : o_b2) + T.tn * 32;
        return T;
    };
    // Adam operands resident in registers when the task map has ONE trip (the wave updates the same parameters every minibatch)
    const bool resident = a.apply && total <= (int)gridDim.x * NG;
    float rp[NR], rm[NR], rv[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) rp[c] = rm[c] = rv[c] = 0.f;
    if (resident) {
        const Task T = map_task(blockIdx.x * NG + grp);
        if (T.live) {                                           // (columns past Nc are past nothing harmful: never stored)
            const int v = ((T.is_tile ? 4 * half * T.Nc : 0) + li) * 4;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const int so = (T.s_e + (T.is_tile ? c * T.Nc : 0)) * 4;
                rp[c] = bld(rs_prm, v, so); rm[c] = bld(rs_am, v, so); rv[c] = bld(rs_av, v, so);
            }
        }
    }
    if (!grid_barrier(a.bar, a.dead_host, gridDim.x, gen, [&]() { if (is_a) prefetch(a.mb_begin); })) return;
    for (int mb = a.mb_begin; mb < a.mb_end; ++mb) {
        const size_t lo = (size_t)mb * MB;
        // An opaque zero, new in every iteration: every scalar offset / descriptor below is formed from it, so none of them is
        // loop-invariant.  Left invariant, LLVM hoists all ~200 of them out of the minibatch loop, runs out of SGPRs and parks them
        // in VGPR lanes: 355 v_writelane in front of the loop and 649 v_readlane - VALU instructions, each in front of the load it
        // feeds - per minibatch and wave (the code object's "SGPR spills").  An s_add in place costs nothing.
        int z = 0;
        asm volatile("" : "+s"(z));
        const int wz = wave + z;
        const __amdgpu_buffer_rsrc_t rs_prm = pass_rsrc(a.prm + z, (size_t)n_prm * 4), rs_am = pass_rsrc(a.am + z, (size_t)n_prm * 4),
                                     rs_av = pass_rsrc(a.av + z, (size_t)n_prm * 4), rs_w2t = pass_rsrc(a.w2t + z, (size_t)HID * AE * 4);
        // ------------------------------------------------------------------ phase A
        RL4RS_PT(0);
        if (is_a) {
        if constexpr (STD) {
            // ---- the default shape on v_mfma_f32_4x4x1 (4 rows x 64 columns x 1 k per instruction): the 8 rows of the workgroup are
            // two row tiles with NO idle rows (the 32x32x2 tiles of the generic branch keep 24 of 32 rows idle, and two waves share
            // a SIMD's matrix pipe: 1.9 + 2.9 + 2.2 us of the three stages were MFMA issue).  Every stage splits K over the 8 waves
            // (B operand = one 256-byte weight row per k and lane = column: the natural layout of W1 / W2e / w2t, one register
            // per k), partial sums meet in LDS in a fixed order.  A wave never has more than 63 requests in flight (the vmcnt
            // counter's 6 bits: the 64th request stalls the wave until the first returns - the 100-load prologue of this round's first
            // version cost a whole memory latency that way): each stage's weights are requested behind the MFMAs of the stage before.
            float* s_x = reinterpret_cast<float*>(smem);      // [8][260] observation rows          (region A: 8192 floats = phase B's s_part)
            float* s_g = s_x + 2080;                           // [8][288] d loss / d [logits | value], pad columns 285..287 zero (dH's A operand)
            float* s_hh = s_x + 8192;                          // [8][68]  hidden
            float* s_p = s_hh + 544;                           // 8192 floats of partial sums: layer 1 [8 parts][8][64]; layer 2 [2][8][256] + [8][8][64]; dH [8][8][64]
            float* s_lg = s_p + 8192;                          // [8][288] masked logits | value
            float* s_ol = s_lg + 2304;                         // [8][284] old logits
            float* s_s5 = s_ol + 2272;                         // [5][32]  action (as int), adv, ret, old logp, old value
            uint32_t* s_mk = reinterpret_cast<uint32_t*>(s_s5 + 160);      // [8][9] mask words
            const int vl = lane * 4;
            float bv1[32], bv2[32], bv2t[8], wv[36], b2v[5];
#pragma unroll
            for (int u = 0; u < 32; ++u) bv1[u] = bld(rs_prm, vl, (wz * 32 + u) * 256);                       // W1 rows wave*32 .. +31
            const float b1v = bld(rs_prm, vl, (o_b1 + z) * 4);
#pragma unroll
            for (int i = 0; i < 5; ++i) b2v[i] = bld(rs_prm, vl, (o_b2 + 64 * i + z) * 4);                          // (past the buffer: 0)
#pragma unroll
            for (int u = 0; u < 8; ++u) bv2t[u] = bld(rs_prm, vl, (o_w2 + (wz * 8 + u) * 285 + 256) * 4);     // W2e rows wave*8 .. +7, columns 256 ..
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < PF_OBS * SR / 8; ++u) s_x[(u * 2 + (tid >> 8)) * 260 + (tid & 255)] = pf_obs[u];       // i = tid + 512 u -> row i / 256
#pragma unroll
            for (int u = 0; u < PF_OLD; ++u)
                if (tid + 512 * u < SR * 284) s_ol[tid + 512 * u] = pf_old[u];
            if (a.mask && tid < SR * 9) s_mk[tid] = pf_mask;
            if (tid < SR) {
                reinterpret_cast<int32_t*>(s_s5)[tid] = pf_act;
                s_s5[32 + tid] = pf_sc[0]; s_s5[64 + tid] = pf_sc[1]; s_s5[96 + tid] = pf_sc[2]; s_s5[128 + tid] = pf_sc[3];
            }
            __syncthreads();
            RL4RS_PT(1);
            f32x4_t o0, o1;
            // layer 1: wave w multiplies k in [32 w, 32 w + 32)
            std_slice<8, MT>(s_x + (lane & 3) * 260 + wave * 32, 260, bv1, o0, o1);
            // layer 2's main operand (column tile wave & 3, K half wave >> 2) flies during the partial-sum exchange and the tanh
#pragma unroll
            for (int u = 0; u < 32; ++u) bv2[u] = bld(rs_prm, vl, (o_w2 + ((wz >> 2) * 32 + u) * 285 + (wz & 3) * 64) * 4);
            __builtin_amdgcn_sched_barrier(0);
            std_put<MT>(s_p + wave * 512, 64, lane, o0, o1);
            __syncthreads();
            if (wave < SR) {   // thread = (row wave, column lane)
                float sum = b1v;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) sum += s_p[pp * 512 + wave * 64 + lane];
                const float h = tanhf(sum);
                s_hh[wave * 68 + lane] = h;
                PUB(a.H[RL4RS_U((r0 + wave) * 64 + lane)], h);
            }
            __syncthreads();
            RL4RS_PT(2);
            // layer 2: columns 0..255 as 4 tiles x 2 K halves (one per wave), columns 256..284 split over K eight ways
            std_slice<8, MT>(s_hh + (lane & 3) * 68 + (wave >> 2) * 32, 68, bv2, o0, o1);
            f32x4_t t0, t1;
            std_slice<2, MT>(s_hh + (lane & 3) * 68 + wave * 8, 68, bv2t, t0, t1);
            // dH's operand (rows 36 w .. of the transposed W2e; rows >= 285 are past the buffer: 0) flies during the row losses
#pragma unroll
            for (int u = 0; u < 36; ++u) wv[u] = bld(rs_w2t, vl, (wz * 36 + u) * 256);
            __builtin_amdgcn_sched_barrier(0);
            std_put<MT>(s_p + (wave >> 2) * 2048 + (wave & 3) * 64, 256, lane, o0, o1);
            std_put<MT>(s_p + 4096 + wave * 512, 64, lane, t0, t1);
            __syncthreads();
            RL4RS_PT(3);
            if (wave < SR) {   // row losses: wave w owns row w - it first joins the partial sums of its row (+ bias, + action mask)
                float* so = s_lg + wave * 288;
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    const int c = lane + 64 * i;
                    if (c < 285) {
                        float v = b2v[i];
                        if (i < 4) {
                            v = (v + s_p[wave * 256 + c]) + s_p[2048 + wave * 256 + c];
                        } else {
#pragma unroll
                            for (int pp = 0; pp < 8; ++pp) v += s_p[4096 + pp * 512 + wave * 64 + lane];
                        }
                        const uint32_t mw = a.mask ? s_mk[wave * 9 + (c >> 5)] : 0xffffffffu;
                        if (c < 284 && !((mw >> (c & 31)) & 1u)) v = v + (-3.4028235e38f);
                        so[c] = v;
                    }
                }
                if (lane < 3) s_g[wave * 288 + 285 + lane] = 0.f;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (wave == 0) RL4RS_PT(14);
                LossArgs L = a.L;                      // inputs from the LDS copies, indexed by the row within the tile
                L.actions = reinterpret_cast<const int32_t*>(s_s5); L.adv = s_s5 + 32; L.ret = s_s5 + 64; L.old_logp = s_s5 + 96;
                L.old_value = s_s5 + 128; L.old_logits = s_ol;
                float mx = -3.4028235e38f;
                for (int c = lane; c < 284; c += 64) mx = fmaxf(mx, so[c]);
                mx = wave_max(mx);
                float se = 0.f;
                for (int c = lane; c < 284; c += 64) se += expf(so[c] - mx);
                const float lse = mx + logf(wave_sum(se));
                if (wave == 0) RL4RS_PT(15);
                const float4 tm = policy_row_loss<RL4RS_PASS_WT != 0>(d, L, so, lse, wave, lane, s_g + wave * 288, a.dOut + (size_t)r0 * AE);
                if (lane == 0) a.terms[lo + r0 + wave] = tm;       // per-sample loss terms of the whole pass (KL mean -> kl_coeff rule)
                if (wave == 0) RL4RS_PT(11);
            }
            __syncthreads();
            RL4RS_PT(4);
            // dH = dOut W2e^T: wave w multiplies k in [36 w, 36 w + 36)
            std_slice<9, MT>(s_g + (lane & 3) * 288 + wave * 36, 288, wv, o0, o1);
            RL4RS_PT(12);
            std_put<MT>(s_p + wave * 512, 64, lane, o0, o1);
            __syncthreads();
            RL4RS_PT(13);
            if (wave < SR) {
                float sum = 0.f;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) sum += s_p[pp * 512 + wave * 64 + lane];
                const float h = s_hh[wave * 68 + lane];
                PUB(a.dHpre[RL4RS_U((r0 + wave) * 64 + lane)], sum * (1.f - h * h));
            }
        } else {
            float bv1[32], bv2[32];
            const int q1z = q1 + z;                       // (loop-variant: see `z`)
            load_w1(q1z * (OD / parts), bv1);
            const float b1v = bld(rs_prm, (tid % HID) * 4, o_b1 * 4);
            __builtin_amdgcn_sched_barrier(0);
            if (can_pf) {
    #pragma unroll
                for (int u = 0; u < PF_OBS; ++u) {
                    const int i = tid + 512 * u;
                    if (i < R * OD) s_obs[(i / OD) * SO + i % OD] = pf_obs[u];
                }
    #pragma unroll
                for (int u = 0; u < PF_OLD; ++u)
                    if (tid + 512 * u < R * d.A) s_old[tid + 512 * u] = pf_old[u];
                if (a.mask && tid < R * d.W) s_mask[tid] = pf_mask;
                if (tid < R) {
                    reinterpret_cast<int32_t*>(s_sc)[tid] = pf_act;
                    s_sc[32 + tid] = pf_sc[0]; s_sc[64 + tid] = pf_sc[1]; s_sc[96 + tid] = pf_sc[2]; s_sc[128 + tid] = pf_sc[3];
                }
            } else {
                const size_t ln = lo + r0;
                const __amdgpu_buffer_rsrc_t rs_o = pass_rsrc(a.obs + ln * OD, (size_t)R * OD * 4), rs_l = pass_rsrc(a.L.old_logits + ln * d.A, (size_t)R * d.A * 4);
                for (int i0 = 0; i0 < R * OD; i0 += 512 * 16) {                 // 16 loads in flight per thread and trip
                    float x[16];
    #pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = bld(rs_o, tid * 4, (i0 + 512 * u) * 4);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int u = 0; u < 16; ++u) {
                        const int i = i0 + tid + 512 * u;
                        if (i < R * OD) s_obs[(i / OD) * SO + i % OD] = x[u];
                    }
                }
                // the row-loss inputs do not depend on the parameters: staged here, in one round trip with the observations
                for (int i0 = 0; i0 < R * d.A; i0 += 512 * 16) {
                    float x[16];
    #pragma unroll
                    for (int u = 0; u < 16; ++u) x[u] = bld(rs_l, tid * 4, (i0 + 512 * u) * 4);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int u = 0; u < 16; ++u)
                        if (i0 + tid + 512 * u < R * d.A) s_old[i0 + tid + 512 * u] = x[u];
                }
                {   // (R * W <= 32 * 16 = 512: one request per thread, no loop to serialise)
                    uint32_t mw = 0;
                    if (a.mask) mw = __builtin_amdgcn_raw_buffer_load_b32(pass_rsrc(a.mask + ln * d.W, (size_t)R * d.W * 4), tid * 4, 0, 0);
                    const int32_t ac = (int)__builtin_amdgcn_raw_buffer_load_b32(pass_rsrc(a.L.actions + ln, (size_t)R * 4), tid * 4, 0, 0);
                    const float sc0 = bld(pass_rsrc(a.L.adv + ln, (size_t)R * 4), tid * 4, 0), sc1 = bld(pass_rsrc(a.L.ret + ln, (size_t)R * 4), tid * 4, 0);
                    const float sc2 = bld(pass_rsrc(a.L.old_logp + ln, (size_t)R * 4), tid * 4, 0), sc3 = bld(pass_rsrc(a.L.old_value + ln, (size_t)R * 4), tid * 4, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (a.mask && tid < R * d.W) s_mask[tid] = mw;
                    if (tid < R) {
                        reinterpret_cast<int32_t*>(s_sc)[tid] = ac;
                        s_sc[32 + tid] = sc0; s_sc[64 + tid] = sc1; s_sc[96 + tid] = sc2; s_sc[128 + tid] = sc3;
                    }
                }
            }
            __syncthreads();
            RL4RS_PT(1);
            {   // layer 1, split over K: wave -> (tile t1, part q1)
                const int kper = OD / parts, kb = q1z * kper, ke = kb + kper;
                f32x16 acc;
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                for (int k = kb;;) {                                        // one memory round trip per 64 k (kper % 64 == 0)
                    __builtin_amdgcn_sched_barrier(0);      // all loads of the trip in flight before the first MFMA
    #pragma unroll
                    for (int u = 0; u < 32; ++u)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_obs[li * SO + k + 2 * u + half], bv1[u], acc, 0, 0, 0);
                    k += 64;
                    if (k >= ke) break;
                    load_w1(k, bv1);
                }
                float* part = s_d + (size_t)(t1 * parts + q1) * 1024;
                for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
            }
            __syncthreads();
            for (int i = tid; i < R * HID; i += 512) {
                const int r = i / HID, j = i - r * HID, t = j >> 5, c = j & 31;
                float s = (i == tid) ? b1v : bld(rs_prm, j * 4, o_b1 * 4);  // (tid % HID == j on the first trip)
                for (int qq = 0; qq < parts; ++qq) s += s_d[(size_t)(t * parts + qq) * 1024 + r * 32 + c];
                const float h = tanhf(s);
                s_h[r * SH + j] = h;
                PUB(a.H[RL4RS_U((r0 + r) * HID + j)], h);
            }
            float wv[36];
            __syncthreads();
            RL4RS_PT(2);
            auto l2_tile = [&](int t, float (&bv)[32], float bias, bool pre) {
                // layer 2 (+ action mask) of column tile t: one memory round trip per 64 k unless preloaded
                const int col = t * 32 + li;
                const bool c_ok = col < AE;
                const int colc = c_ok ? col : AE - 1;
                f32x16 acc;
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                for (int k = 0; k < HID; k += 64) {            // HID % 64 == 0
                    if (!(pre && k == 0)) load_w2(t, k, bv, bias);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int u = 0; u < 32; ++u)
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_h[li * SH + k + 2 * u + half], bv[u], acc, 0, 0, 0);
                }
                if (c_ok)
                    for (int r = 0; r < 16; ++r) {
                        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row < R) {
                            float v = acc[r] + bias;
                            const uint32_t mw = a.mask ? s_mask[row * d.W + (colc >> 5)] : 0xffffffffu;
                            if (col < d.A && !((mw >> (col & 31)) & 1u)) v = v + (-3.4028235e38f);
                            s_out[row * SA + col] = v;
                        }
                    }
            };
            for (int t = wz; t < NT2; t += 8) l2_tile(t, bv2, 0.f, false);
            __syncthreads();
            RL4RS_PT(3);
            {   // row losses: wave w takes rows w, w + 8, ...
                LossArgs L = a.L;                      // inputs from the LDS copies, indexed by the row within the tile
                L.actions = reinterpret_cast<const int32_t*>(s_sc); L.adv = s_sc + 32; L.ret = s_sc + 64; L.old_logp = s_sc + 96;
                L.old_value = s_sc + 128; L.old_logits = s_old;
                for (int row = wave; row < R; row += 8) {
                    const float* so = s_out + row * SA;
                    float mx = -3.4028235e38f;
                    for (int c = lane; c < d.A; c += 64) mx = fmaxf(mx, so[c]);
                    mx = wave_max(mx);
                    float se = 0.f;
                    for (int c = lane; c < d.A; c += 64) se += expf(so[c] - mx);
                    const float lse = mx + logf(wave_sum(se));
                    const float4 tm = policy_row_loss<RL4RS_PASS_WT != 0>(d, L, so, lse, row, lane, s_d + row * SA, a.dOut + (size_t)r0 * AE);
                    if (lane == 0) a.terms[lo + r0 + row] = tm;       // per-sample loss terms of the whole pass (KL mean -> kl_coeff rule)
                    if (row == 0) RL4RS_PT(11);
                }
            }
            __syncthreads();
            RL4RS_PT(4);
            {   // dH = dOut W2e^T, split over K: wave -> (tile t1, part q1); the B operand comes from the transposed copy (coalesced)
                f32x16 acc;
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const float* drow = s_d + li * SA;
                const int kb = q1z * kper_h, k_hi = min(kb + kper_h, AE);
                for (int k = kb; k < k_hi; k += 72) {                       // 36 loads in flight per trip (one trip for AE <= 288)
                    load_w2t(k, wv);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int u = 0; u < 36; ++u) {
                        const int kk = k + 2 * u + half;
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk < k_hi ? drow[min(kk, AE - 1)] : 0.f, wv[u], acc, 0, 0, 0);
                    }
                }
                RL4RS_PT(12);
                float* part = s_out + (size_t)(t1 * parts + q1) * 1024;
                for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
            }
            __syncthreads();
            RL4RS_PT(13);
            for (int i = tid; i < R * HID; i += 512) {
                const int r = i / HID, j = i - r * HID, t = j >> 5, c = j & 31;
                float s = 0.f;
                for (int qq = 0; qq < parts; ++qq) s += s_out[(size_t)(t * parts + qq) * 1024 + r * 32 + c];
                const float h = s_h[r * SH + j];
                PUB(a.dHpre[RL4RS_U((r0 + r) * HID + j)], s * (1.f - h * h));
            }
        }
        }   // is_a
        RL4RS_PT(5);
        if (!grid_barrier(a.bar, a.dead_host, gridDim.x, gen)) return;
        RL4RS_PT(6);
        // ------------------------------------------------------------------ phase B
        bool tile_stored = false;          // the wave's LAST stores are the 2 NR private moment stores of a tile task (scalar: the task map is)
        {
            const double tt = (double)(a.t0 + mb + 1);
            const float lr_t = (float)((double)a.lr * sqrt(1.0 - pow((double)a.b2, tt)) / (1.0 - pow((double)a.b1, tt)));
            RL4RS_PT(10);
            float* s_part = reinterpret_cast<float*>(smem) + (size_t)grp * NW * 1024;     // [NW partials][32 x 32] per group (the observation rows' region is free here)
            constexpr int TP = 16;      // sample pairs per trip: 32 requests in flight (one trip of 64 is SLOWER - the 64th request stalls the wave, vmcnt has 6 bits)
            for (int t0 = 0; t0 < total; t0 += gridDim.x * NG) {          // uniform trip count: the barriers below are workgroup-wide
                const Task T = map_task(t0 + blockIdx.x * NG + grp);
                tile_stored = false;
                const int j = T.tn * 32 + li;
                const bool j_ok = j < T.Nc;
                if (T.live && T.is_tile) {
                    // 1 / NW of A^T B over the samples; A = obs [MB, OD] or H [MB, HID], B = dHpre or dOut (this minibatch's rows)
                    const int lda = T.first ? OD : HID, ldb = T.Nc;
                    const __amdgpu_buffer_rsrc_t rs_a = T.first ? pass_rsrc(a.obs + lo * OD, (size_t)MB * OD * 4) : pass_rsrc(a.H, (size_t)MB * HID * 4);
                    const __amdgpu_buffer_rsrc_t rs_b = T.first ? pass_rsrc(a.dHpre, (size_t)MB * HID * 4) : pass_rsrc(a.dOut, (size_t)MB * AE * 4);
                    const int v_a = (half * lda + T.tm * 32 + li) * 4, v_b = (half * ldb + (j_ok ? j : T.Nc - 1)) * 4;
                    f32x16 acc;
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    for (int n = q * spq + z; n < (q + 1) * spq; n += 2 * TP) {
                        float av[TP], bw[TP];
#pragma unroll
                        for (int u = 0; u < TP; ++u) {
                            av[u] = bld(rs_a, v_a, (n + 2 * u) * lda * 4);
                            bw[u] = bld(rs_b, v_b, (n + 2 * u) * ldb * 4);
                        }
                        if (is_a && n + 2 * TP >= (q + 1) * spq && t0 == 0) prefetch(mb + 1);     // behind the LAST trip's requests: nothing of phase B waits for it
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < TP; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bw[u], acc, 0, 0, 0);
                    }
                    for (int r = 0; r < 16; ++r) s_part[q * 1024 + r * 64 + lane] = acc[r];
                } else if (T.live) {
                    // 1 / NW of the bias column sums: lanes = 32 columns x 2 sample parities
                    const __amdgpu_buffer_rsrc_t rs_x = T.first ? pass_rsrc(a.dHpre, (size_t)MB * HID * 4) : pass_rsrc(a.dOut, (size_t)MB * AE * 4);
                    const int v_x = (half * T.Nc + (j_ok ? j : T.Nc - 1)) * 4;
                    float bsum = 0.f;
                    for (int n = q * spq + z; n < (q + 1) * spq; n += 32) {
                        float x[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) x[u] = bld(rs_x, v_x, (n + 2 * u) * T.Nc * 4);
                        if (is_a && n + 32 >= (q + 1) * spq && t0 == 0) prefetch(mb + 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 16; ++u) bsum += x[u];
                    }
                    s_part[q * 1024 + lane] = bsum;
                } else if (is_a && t0 == 0) {
                    prefetch(mb + 1);
                }
                // Adam's operands: the wave's resident copies, or (multi-trip task maps) requested here, in front of the workgroup
                // barrier that joins the partial sums
                float pp[NR], mm[NR], vv[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    if (resident) { pp[c] = rp[c]; mm[c] = rm[c]; vv[c] = rv[c]; }
                    else if (a.apply && T.live) {
                        const int v = ((T.is_tile ? 4 * half * T.Nc : 0) + li) * 4, so = (T.s_e + (T.is_tile ? c * T.Nc : 0)) * 4;
                        pp[c] = bld(rs_prm, v, so); mm[c] = bld(rs_am, v, so); vv[c] = bld(rs_av, v, so);
                    } else { pp[c] = mm[c] = vv[c] = 0.f; }
                }
                RL4RS_PT(9);
                __syncthreads();
                if (T.live && T.is_tile) {
                    if (j_ok) {
                        // wave q owns accumulator registers NR q .. NR q + NR - 1 of the tile: rows tm*32 + rowbase + c + 4 half, column j
                        const unsigned e0 = RL4RS_U(T.s_e + 4 * half * T.Nc + li);
                        const unsigned w0 = RL4RS_U(j * HID + T.tm * 32 + rowbase + 4 * half);     // the same elements in w2t: NR consecutive ones
                        float g[NR];
#pragma unroll
                        for (int c = 0; c < NR; ++c) {
                            const int r = NR * q + c;
                            float sum = s_part[r * 64 + lane];
#pragma unroll
                            for (int pw = 1; pw < NW; ++pw) sum += s_part[pw * 1024 + r * 64 + lane];
                            g[c] = sum;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        float pn[NR], mi[NR], vi[NR];
#pragma unroll
                        for (int c = 0; c < NR; ++c) {
                            mi[c] = a.b1 * mm[c] + (1.f - a.b1) * g[c];
                            vi[c] = a.b2 * vv[c] + (1.f - a.b2) * g[c] * g[c];
                            pn[c] = pp[c] - lr_t * mi[c] / (sqrtf(vi[c]) + a.eps);
                        }
                        // the gradient itself leaves the kernel in the data-parallel form and for the pass's last minibatch (the API's grad_dev)
                        if (!a.apply || mb == a.mb_end - 1) {
#pragma unroll
                            for (int c = 0; c < NR; ++c) a.grad[e0 + RL4RS_U(c * T.Nc)] = g[c];
                        }
                        if (a.apply) {
                            // what OTHER workgroups read (the parameters, W2e's transposed copy) goes first, write-through; the moments -
                            // private to this wave - go last: the barrier's drain does not wait for them
#pragma unroll
                            for (int c = 0; c < NR; ++c) PUB(a.prm[e0 + RL4RS_U(c * T.Nc)], pn[c]);
                            if (!T.first) {
                                // rows rowbase + c + 4 half of column j are NR CONSECUTIVE elements of w2t's row j: one 16- / 8-byte store
                                // per lane (was four 4-byte stores to 64 different lines each)
                                if constexpr (NR == 4) {
                                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                                    const u32x4_t pk = {__builtin_bit_cast(unsigned, pn[0]), __builtin_bit_cast(unsigned, pn[1]),
                                                        __builtin_bit_cast(unsigned, pn[2]), __builtin_bit_cast(unsigned, pn[3])};
                                    __builtin_amdgcn_raw_buffer_store_b128(pk, rs_w2t, (int)(w0 * 4), 0, RL4RS_PASS_WT ? 16 : 0);      // aux 16 = sc1
                                } else {
                                    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
                                    const u32x2_t pk = {__builtin_bit_cast(unsigned, pn[0]), __builtin_bit_cast(unsigned, pn[1])};
                                    __builtin_amdgcn_raw_buffer_store_b64(pk, rs_w2t, (int)(w0 * 4), 0, RL4RS_PASS_WT ? 16 : 0);
                                }
                            }
#pragma unroll
                            for (int c = 0; c < NR; ++c) a.am[e0 + RL4RS_U(c * T.Nc)] = mi[c];
#pragma unroll
                            for (int c = 0; c < NR; ++c) a.av[e0 + RL4RS_U(c * T.Nc)] = vi[c];
                            if (resident) {
#pragma unroll
                                for (int c = 0; c < NR; ++c) { rp[c] = pn[c]; rm[c] = mi[c]; rv[c] = vi[c]; }
                            }
                            tile_stored = true;
                        }
                    }
                } else if (T.live && q == 0 && half == 0 && j_ok) {
                    float sum = 0.f;
                    for (int qq = 0; qq < NW; ++qq) sum += s_part[qq * 1024 + li] + s_part[qq * 1024 + 32 + li];
                    const unsigned e = RL4RS_U(T.s_e + li);
                    a.grad[e] = sum;
                    if (a.apply) {
                        const float mi = a.b1 * mm[0] + (1.f - a.b1) * sum;
                        const float vi = a.b2 * vv[0] + (1.f - a.b2) * sum * sum;
                        const float pn = pp[0] - lr_t * mi / (sqrtf(vi) + a.eps);
                        PUB(a.prm[e], pn);
                        a.am[e] = mi;
                        a.av[e] = vi;
                        if (resident) { rp[0] = pn; rm[0] = mi; rv[0] = vi; }
                    }
                }
                if (t0 + (int)gridDim.x * NG < total) __syncthreads();    // s_part is rewritten by the next trip
            }
        }
        RL4RS_PT(7);
        // drain what other workgroups will read, not the 2 NR moment stores behind it (vmcnt retires in order: "at most 2 NR
        // outstanding" = everything before the moments has completed; 2 us of a minibatch were waves waiting for their own private stores)
        if (tile_stored) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NR) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!grid_barrier<NoMid, true>(a.bar, a.dead_host, gridDim.x, gen)) return;
        RL4RS_PT(8);
    }
}
