// k_din_x: DIN attention scores of the DIEN scorer in fp16x2 form, second generation (included by dien.hip).
//
// Same arithmetic as k_din_scores<*, true> (deepctr LocalActivationUnit, rl4rs/nets/utils.py:112-119: hidden (64, 16), sigmoid,
// raw scores; layer 1 as q(W1a+W1c) [GEMM] + k_t(W1b-W1c) [cache] + (q*k_t)W1d [here], operands split into fp16 hi + lo, three
// v_mfma_f32_32x32x16_f16 per product), different machine mapping.  What bounded the first form (one row per wave, 144
// registers = 3 waves per SIMD): each wave alternated a VALU-only stretch (building the split (q*k_t) operand, the sigmoids)
// with an MFMA-only stretch, every wave pulled its own copy of the 32 KB of W1d fragments through the L1 return path, and
// the AK rows arrived in the epilogue.
//
//   * W1d fragments (32 KB, both planes) are staged ONCE per workgroup in LDS and read with ds_read_b128 right behind their
//     use; a workgroup is 8 waves and takes rows_per_wg rows (obs-sized launches: 16, so that
//     two workgroups per CU = 4 waves per SIMD cover the launch in one round, no tail).
//   * one 32-step tile at a time (32 accumulator registers instead of 64) and no double buffers, so that the kernel fits 128
//     registers = 4 waves per SIMD, which overlap each other's VALU (operand split, sigmoids) and MFMA stretches; the cache
//     rows are requested RING k-blocks ahead.
//   * the accumulators START as AK_t + qa (the cached k-side term and the q-side GEMM term), loaded while the first operands
//     are built: no loads in the epilogue.
//   * fp16 split by v_cvt_pk_f16_f32 + v_fma_mix_f32 (split_h16_pair): 2.5 VALU instructions per element.
//   * the first-GRU states are read from a second, FRAGMENT-ORDER copy of the cache (k_h1_frag, once per encode):
//     h1f[slot][step tile n][k-block kb][lane][8 floats], lane (li, kg) = step n*32 + li, k = kb*16 + kg*8 + 0..7 - exactly
//     the 32 bytes a lane multiplies by q, so a wave's request is one contiguous 2 KB piece (16 full cache lines) instead of
//     32 lines of which it uses a quarter each; the rows of a reward group (same slot) hit in L1, no LDS tile needed.
#pragma once

#ifndef RL4RS_DINX_AB
#define RL4RS_DINX_AB 0         // timing ablations of k_din_x (results are WRONG when non-zero): 1 all rows on cache slot 0, 2 no layer-1 sigmoids
#endif

namespace rl4rs {

#ifdef RL4RS_DINX_TRACE    // s_memtime marks of workgroup (40, 0): [wave][tile 0..3 | 4 = workgroup marks][mark] (tools/dinx_trace.py)
#define DINX_TR(tile, k) do { if (a.trace && blockIdx.x == 40 && blockIdx.y == 0 && lane == 0) \
        a.trace[(wave * 5 + (tile)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define DINX_TR(tile, k) do { } while (0)
#endif

// h1 [slot, L, E = 128] -> fragment order [slot][NT][8][64][8]; steps >= L of the last tile are zero
__global__ __launch_bounds__(256) void k_h1_frag(const float* __restrict__ h1, float* __restrict__ h1f, int slot_base, int cnt, int L) {
    const int NT = (L + 31) / 32;
    const int64_t total = (int64_t)cnt * NT * 8 * 64 * 2;          // float4 pieces
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int q4 = (int)(i & 1), lane = (int)((i >> 1) & 63), kb = (int)((i >> 7) & 7);
        const int64_t sn = i >> 10;
        const int n = (int)(sn % NT);
        const int64_t slot = slot_base + sn / NT;
        const int t = n * 32 + (lane & 31), e = kb * 16 + (lane >> 5) * 8 + q4 * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t < L) v = *reinterpret_cast<const float4*>(h1 + (slot * L + t) * 128 + e);
        *reinterpret_cast<float4*>(h1f + (((slot * NT + n) * 8 + kb) * 64 + lane) * 8 + q4 * 4) = v;
    }
}

__global__ __launch_bounds__(512, RL4RS_DINX_WPE) void k_din_x(DinArgs a, int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 128, KB = 8, NW = 8;
    const int L = a.L;
    const int sq = blockIdx.y;
    char* s_w1 = smem;                                                  // [(m*8 + kb)*2 + plane][lane][8 halfs]: 32 KB
    char* s_w2b = smem + 32768;                                         // [(m*2 + kb2)*2 + plane][lane][8 halfs]: 8 KB
    float* s_misc = reinterpret_cast<float*>(smem + 40960);             // b2[16] w3[16] b3 pad -> 48
    float* s_wave = s_misc + 48;                                        // per wave: q[E] + qa[64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    DINX_TR(4, 0);
    {
        // staging: every request of the workgroup's 40 KB of weights goes out before the first is stored.  (As two plain loops the
        // ISA was "load, s_waitcnt vmcnt(0), store" per iteration: eight serialised memory round trips = 11 k of a workgroup's
        // ~88 k cycles, tools/dinx_trace.py.)  W2 entries beyond the 16 real output units read a clamped address and are zeroed.
        // (ext_vector_type, not HIP's uint4 struct: a struct copied out of global memory is a memcpy the optimiser leaves in scratch)
        typedef float stage4_t __attribute__((ext_vector_type(4)));
        const stage4_t* src = reinterpret_cast<const stage4_t*>(a.w1d16[sq]);
        stage4_t* dst = reinterpret_cast<stage4_t*>(s_w1);
        stage4_t wv[4];
        float w2v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) wv[p] = src[tid + p * 512];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = tid + p * 512, e = i & 7, ln = (i >> 3) & 63, f = i >> 9, o = ln & 31;
            w2v[p] = a.w2[sq][((f >> 1) * 32 + crow((f & 1) * 8 + e, ln >> 5)) * ATT_H2 + min(o, ATT_H2 - 1)];
        }
        float m0 = 0.f, m1 = 0.f, m2 = 0.f;
        if (tid < ATT_H2) { m0 = a.b2[sq][tid]; m1 = a.w3[sq][tid]; }
        if (tid == 0) m2 = a.b3[sq][0];
#pragma unroll
        for (int p = 0; p < 4; ++p) dst[tid + p * 512] = wv[p];
        _Float16* s_w2h = reinterpret_cast<_Float16*>(s_w2b);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int i = tid + p * 512, e = i & 7, ln = (i >> 3) & 63, f = i >> 9, o = ln & 31;
            const float v = o < ATT_H2 ? w2v[p] : 0.f;
            const _Float16 hi = (_Float16)v;
            s_w2h[(f * 2) * 512 + ln * 8 + e] = hi;
            s_w2h[(f * 2 + 1) * 512 + ln * 8 + e] = (_Float16)(v - (float)hi);
        }
        if (tid < ATT_H2) { s_misc[tid] = m0; s_misc[16 + tid] = m1; }
        if (tid == 0) s_misc[32] = m2;
    }
    __syncthreads();
    DINX_TR(4, 1);
    float* s_q = s_wave + (size_t)wave * (E + ATT_H1);
    float* s_qa = s_q + E;
    const int ntile = (L + 31) / 32;
    const int grp = a.group;

    for (int j = wave; j < rows_per_wg; j += NW) {
        // row groups (rows that share a cache slot) in the caller's processing order, rows of a group consecutive
        const int idx = blockIdx.x * rows_per_wg + j;
        if (idx >= a.R) break;
        const int g = idx / grp;
        const int gs = a.order ? a.order[g] : g;
        const int row = gs * grp + (idx - g * grp);
#if RL4RS_DINX_AB & 1       // timing ablation (results WRONG): every row reads cache slot 0 - the cache traffic becomes L1 / L2 hits
        const int slot = 0 * a.slots[(size_t)sq * a.slots_stride + gs];
#else
        const int slot = a.slots[(size_t)sq * a.slots_stride + gs];
#endif
        const int lead = a.lead[sq] ? a.lead[sq][slot] : 0;
        __builtin_amdgcn_wave_barrier();
        // the row's q and q-side term are requested here and staged in LDS inside the first tile, behind that tile's own
        // requests: one memory round trip at the start of a row instead of two
        const float q_lo = a.q[(size_t)row * E + lane], q_hi = a.q[(size_t)row * E + lane + 64];
        const float qa_v = a.qa[(size_t)sq * a.qa_stride + (size_t)row * a.qa_ld + lane];
        const float* qp = s_q + half * 8;

        for (int n = 0; n < ntile; ++n) {
            DINX_TR((j / NW) * 2 + n, 0);
            const int t = n * 32 + li;
            const int tc = min(t, L - 1);                  // steps >= L re-read the last row (results never stored)
            const int slot_t = t < lead ? a.pad_slot : slot;      // (a lane = a step of the tile: front padding comes from the pad slot)
            const float* hp = a.h1f[sq] + (((size_t)slot_t * ntile + n) * KB * 64 + lane) * 8;
            const float* akp = a.proj[sq] + ((size_t)slot_t * L + tc) * a.pld;
            float4 hr[RL4RS_DINX_RING][2];
            auto ldh = [&](int s, int kb) {
                hr[s][0] = *reinterpret_cast<const float4*>(hp + kb * 512);
                hr[s][1] = *reinterpret_cast<const float4*>(hp + kb * 512 + 4);
            };
            half8_t wh[2], wl[2];
            auto ldw = [&](int m, int kb) {
                wh[m] = *reinterpret_cast<const half8_t*>(s_w1 + ((m * KB + kb) * 2) * 1024 + lane * 16);
                wl[m] = *reinterpret_cast<const half8_t*>(s_w1 + ((m * KB + kb) * 2 + 1) * 1024 + lane * 16);
            };
            half8_t bh, bl;
            auto mkb = [&](int s, int kb) {
                const float4 qa4 = *reinterpret_cast<const float4*>(qp + kb * 16);
                const float4 qb4 = *reinterpret_cast<const float4*>(qp + kb * 16 + 4);
                const float pr[8] = {hr[s][0].x * qa4.x, hr[s][0].y * qa4.y, hr[s][0].z * qa4.z, hr[s][0].w * qa4.w,
                                     hr[s][1].x * qb4.x, hr[s][1].y * qb4.y, hr[s][1].z * qb4.z, hr[s][1].w * qb4.w};
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    half2_t h2, l2;
                    split_h16_pair(pr[e], pr[e + 1], h2, l2);
                    bh[e] = h2[0]; bh[e + 1] = h2[1];
                    bl[e] = l2[0]; bl[e + 1] = l2[1];
                }
            };
            // accumulators start as AK_t + qa: register r of tile m = hidden unit m*32 + crow(r, half)
            f32x16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 ak = *reinterpret_cast<const float4*>(akp + m * 32 + 8 * r4 + 4 * half);
                    acc[m][r4 * 4 + 0] = ak.x; acc[m][r4 * 4 + 1] = ak.y; acc[m][r4 * 4 + 2] = ak.z; acc[m][r4 * 4 + 3] = ak.w;
                }
#pragma unroll
            for (int kb = 0; kb < RL4RS_DINX_RING; ++kb) ldh(kb, kb);
            __builtin_amdgcn_sched_barrier(0);
            if (n == 0) {
                s_q[lane] = q_lo;
                s_q[lane + 64] = q_hi;
                s_qa[lane] = qa_v;
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const float4 qv = *reinterpret_cast<const float4*>(s_qa + m * 32 + 8 * r4 + 4 * half);
                    acc[m][r4 * 4 + 0] += qv.x; acc[m][r4 * 4 + 1] += qv.y; acc[m][r4 * 4 + 2] += qv.z; acc[m][r4 * 4 + 3] += qv.w;
                }
            __builtin_amdgcn_sched_barrier(0);
            // per k-block: weight fragments (LDS) requested first, the split operand built behind them, then the six MFMAs; the
            // cache rows of k-block kb + RING are requested as soon as kb's have been consumed.  A wave alternates a VALU and an
            // MFMA stretch; the four waves of a SIMD fill each other's gaps (registers kept under 128 for that).
            DINX_TR((j / NW) * 2 + n, 1);
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                ldw(0, kb);
                ldw(1, kb);
                mkb(kb % RL4RS_DINX_RING, kb);
                __builtin_amdgcn_sched_barrier(0);
                if (kb == 0) DINX_TR((j / NW) * 2 + n, 2);
                if (kb == 4) DINX_TR((j / NW) * 2 + n, 3);
                if (kb + RL4RS_DINX_RING < KB) ldh(kb % RL4RS_DINX_RING, kb + RL4RS_DINX_RING);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], bh, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0], bh, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[1], bh, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0], bl, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[1], bl, acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            DINX_TR((j / NW) * 2 + n, 4);
            // epilogue: hid1 = sigmoid(acc); layer 2 on the matrix pipe in the same split form; layer 3 in registers
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2) {
                    half8_t bh2, bl2;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        half2_t h2, l2;
#if RL4RS_DINX_AB & 2       // timing ablation (results WRONG): no layer-1 sigmoids
                        split_h16_pair(acc[m][kb2 * 8 + e], acc[m][kb2 * 8 + e + 1], h2, l2);
#else
                        split_h16_pair(gate_sigmoid(acc[m][kb2 * 8 + e]), gate_sigmoid(acc[m][kb2 * 8 + e + 1]), h2, l2);
#endif
                        bh2[e] = h2[0]; bh2[e + 1] = h2[1];
                        bl2[e] = l2[0]; bl2[e + 1] = l2[1];
                    }
                    const half8_t ah2 = *reinterpret_cast<const half8_t*>(s_w2b + ((m * 2 + kb2) * 2) * 1024 + lane * 16);
                    const half8_t al2 = *reinterpret_cast<const half8_t*>(s_w2b + ((m * 2 + kb2) * 2 + 1) * 1024 + lane * 16);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah2, bh2, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al2, bh2, acc2, 0, 0, 0);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah2, bl2, acc2, 0, 0, 0);
                }
            // acc2 register r < 8 of this lane = hid2 pre-activation of output unit crow(r, half) at this lane's step
            float sc = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int o = crow(r, half);
                const float h2 = gate_sigmoid(acc2[r] + s_misc[o]);
                sc = fmaf(h2, s_misc[16 + o], sc);
            }
            sc += __shfl_xor(sc, 32);
            sc += s_misc[32];
            if (half == 0 && t < L) a.scores[(size_t)sq * a.scores_stride + (size_t)row * L + t] = sc;
            DINX_TR((j / NW) * 2 + n, 5);
        }
    }
    DINX_TR(4, 2);
}


// Round 4, measured and dropped - "k_din_x2": both 32-step tiles of a row in one wave (one W1d fragment read feeding two tiles,
// twelve MFMAs per k-block on four accumulators, next k-block's fragments / split operands requested and built a k-block
// ahead, 172 registers = two waves per SIMD).  Bit-identical scores, but 1.15 ms per episode-batch against 0.925 ms for this
// kernel (same box): what bounds the DIN scores is VALU issue, not the LDS / cache waits the wave-level counters suggested
// (SQ_ACTIVE_INST_VALU = 42 k cycles per SIMD and obs-sized launch against 30.7 k cycles of MFMA; timing ablations: all rows on
// one cache slot -9 %, no layer-1 sigmoids -4 %), and eight waves per CU overlap the two pipes worse than sixteen.
// A finding worth keeping from that attempt: split_h16_pair is INLINE ASM, invisible to the compiler's MFMA hazard recogniser.
// With double-buffered operands the register allocator gave a just-dead MFMA B-operand register to the next asm conversion
// issued right behind that MFMA, and the 8-pass v_mfma_f32_32x32x16_f16 was still reading it: wrong scores in 4-lane groups
// (tests/test_gpu_dien.py::test_dien_rowwise_matches_oracle caught it).  In this kernel every asm definition sits behind an LDS
// or cache wait; keep it that way, or write the split with compiler-visible conversions where a definition can follow an MFMA.
//
// Round 5, measured and dropped (same-box A/Bs of the DIN ms per episode-batch; head = this kernel, 0.90 ms):
//   * the ring of first-GRU states 3 / 4 / 8 k-blocks deep instead of 2: 0.909 / 0.919 / (18 spills); the ring running ACROSS tiles and
//     rows (the slots freed by a tile's last k-blocks take the next tile's first ones): 0.908.  The s_memtime marks
//     (tools/dinx_trace.py, -DRL4RS_DINX_TRACE) do show 1 - 4 k cycles of waiting at every tile start on a workload with 4096
//     distinct histories (HBM-bound there: 537 MB per launch), but on the bench's sharing (1 771 distinct, duplicates adjacent) the
//     wait only moves into the first k-blocks: a tile is ~4 k cycles of requests + first operand, ~10.5 k of k loop (1.3 k per
//     k-block), 2.3 k of epilogue whichever way the requests are arranged; workgroup life 85 - 95 k cycles of which 11 k staging.
//   * W1d's hi planes (or both planes) resident in registers, two waves per SIMD, 32 rows per workgroup (4 instead of 6 / 2
//     instead of 6 ds_read_b128 per k-block): 1.01 / 0.99 ms - fewer LDS reads do not pay for half the waves.
//   So the k loop runs at ~325 cycles per wave and k-block on a SIMD whose matrix pipe needs 192, whose VALU ~160 and whose LDS
//   issue ~120 for it: three comparably loaded resources overlapping at ~60 %, four waves per SIMD being what makes them overlap
//   at all.  Neither more bytes in flight nor fewer LDS reads nor fewer, fatter waves moves it.

inline size_t din_x_smem() { return 40960 + 48 * 4 + (size_t)8 * (128 + ATT_H1) * 4; }

}  // namespace rl4rs
