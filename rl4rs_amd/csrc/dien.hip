// DIEN simulator-net forward on gfx950 (rl4rs/nets/dien.py:8-45, rl4rs/nets/utils.py:16-25,48-54,100-129;
// cell equations of TF1.15 GRUCell / deepctr-0.9.0 VecAttGRUCell + LocalActivationUnit).
//
// Design: exact fp32 on v_mfma_f32_32x32x2_f32.
//   * Everything that depends only on a sequence input (embedding lookup, the first GRU, and the
//     input-side halves of the attention MLP and of the AUGRU matmuls) is computed ONCE per sequence
//     ("encode") into a per-slot cache in HBM: h1 [slot,L,E] and proj [slot,L, 64 | 4E | 2E].  Rows of one
//     env share their sequences, so a forward over R rows only indexes the cache by slot.
//         [x,h] @ W  =  x @ W[:E]  (cached, incl. bias)  +  h @ W[E:]   (per step, on the matrix cores)
//         [q,k,q-k,q*k] @ W1 = q @ (W1a+W1c) (per row) + k @ (W1b-W1c) + b1 (cached) + (q*k) @ W1d (MFMA)
//   * The first-GRU input projection is a table: embw1 = seq_emb @ Wx + b, [H, 3E], built at create time
//     (HBM is 288 GB; 150 MB per sequence input buys the whole x-side GEMM of the first GRU).
//   * Recurrences run as ONE launch for all L steps: a workgroup owns 32 rows, keeps h in LDS/registers and
//     streams the (pre-packed, L2-resident) recurrent weights every step; no inter-workgroup traffic.
#include <cstdlib>
#include <vector>

#include "common.hpp"
#include "recur_args.hpp"

namespace rl4rs {


__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float eluf_(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ int crow(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

constexpr int ATT_H1 = 64, ATT_H2 = 16, OBS_DIM = 256;
#ifndef RL4RS_AUGRU_U
#define RL4RS_AUGRU_U 2
#endif
constexpr int AUGRU_U = RL4RS_AUGRU_U, GRU_U = 2;   // k-blocks per register-ring slot
#ifndef RL4RS_H16_NLDS
#define RL4RS_H16_NLDS 5         // further weight items of a step kept in the LDS the planes leave free (16 KB per item)
#endif
#ifndef RL4RS_H16_NRES
#define RL4RS_H16_NRES 14        // weight items of a step kept resident in registers (k_augru_h16)
#endif
#ifndef RL4RS_H16_RING1
#define RL4RS_H16_RING1 4        // weight ring depth (items) of k_augru_h16: 3 items = 9 MFMAs ahead;
                                 // measured (ring, resident): (8,10) 60.6 ms, (6,12) 59.5, (4,14) 59.0 per 5 episodes
#endif



#ifndef RL4RS_CAT_FAST_EXP
#define RL4RS_CAT_FAST_EXP 0     // 1: the softmax of the category self-attention on the hardware exp2 (timing A/B, round 4)
#endif
#if RL4RS_CAT_FAST_EXP
#define CAT_EXP(x) __builtin_amdgcn_exp2f(1.4426950408889634f * (x))
#else
#define CAT_EXP(x) expf(x)
#endif
// -------------------------------------------------------------------------------------------------
// Category branch (utils.py:16-25) + attention query (dien.py:29-30, utils.py:114-115).
// One wave per row; 4 rows per block.  Writes allf[row, off_c : off_c + E] = mean_i(softmax(E E^T) E),
// (optionally) allf[row, off_c+E : off_c+E+Cn*E] = flatten(E), q[row] = mean(seq_emb[cat[-10:]]).
// S = E E^T (Cn <= 32 rows, K = E) is ONE 32x32 MFMA tile whose A and B fragments are the same registers.
// S is symmetric, so the softmax over the keys of query i equals the softmax down column i of the tile: lane
// (column) i normalises its own 16+16 register values (two half-waves), no cross-lane traffic for max / sum.
__global__ __launch_bounds__(256) void k_cat_attn(const int32_t* __restrict__ cat, int R, int Cn, int E, int H,
                                                  const float* __restrict__ cat_emb,
                                                  const float* __restrict__ seq_emb, float* __restrict__ allf,
                                                  int ldf, int off_c, float* __restrict__ q, int write_flat, int h16,
                                                  const float* __restrict__ ptab, const float* __restrict__ obs_b,
                                                  float* __restrict__ tsum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31;
    const int LE = E + 4;                                 // 16-byte aligned rows, conflict-free ds_read_b128
    float* sE = reinterpret_cast<float*>(smem) + (size_t)wave * (Cn * LE + 32);
    float* sW = sE + Cn * LE;                             // [32] column weights
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    const int32_t* crowp = cat + (size_t)row * Cn;
    float* frow = allf + (size_t)row * ldf + off_c;
    // embedding gather: ids first (one coalesced load, broadcast by shuffle), then 8 rows' loads in flight at a time
    // (a load-store-load chain per row would expose the full memory latency Cn times)
    const int myid = (lane < Cn) ? min(max(crowp[lane], 0), H - 1) : 0;
    // the query's own gather (seq_emb rows of the last 10 ids, utils.py:114-115) is requested FIRST and consumed last: its
    // latency hides behind everything else
    const int nq = min(10, Cn);
    float qv0[10], qv1[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        const float* src = seq_emb + (size_t)__shfl(myid, max(Cn - 10 + u, 0)) * E;
        qv0[u] = lane < E ? src[lane] : 0.f;
        qv1[u] = lane + 64 < E ? src[lane + 64] : 0.f;
    }
    if (Cn <= 24 && E == 128) {
        // all category rows in flight at once (48 registers): one memory round trip instead of three
        float v0[24], v1[24];
#pragma unroll
        for (int u = 0; u < 24; ++u) {
            const float* src = cat_emb + (size_t)__shfl(myid, min(u, Cn - 1)) * E;
            v0[u] = src[lane];
            v1[u] = src[lane + 64];
        }
#pragma unroll
        for (int u = 0; u < 24; ++u)
            if (u < Cn) {
                sE[u * LE + lane] = v0[u];
                sE[u * LE + lane + 64] = v1[u];
                if (write_flat) {
                    frow[E + u * E + lane] = v0[u];
                    frow[E + u * E + lane + 64] = v1[u];
                }
            }
    } else
    for (int c0 = 0; c0 < Cn; c0 += 8) {
        float v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int id = __shfl(myid, min(c0 + u, Cn - 1));
            const float* src = cat_emb + (size_t)id * E;
            v0[u] = src[lane];
            v1[u] = src[lane + 64];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int c = c0 + u;
            if (c < Cn) {
                sE[c * LE + lane] = v0[u];
                sE[c * LE + lane + 64] = v1[u];
                if (write_flat) {                          // Flatten()(category_emb)
                    frow[E + c * E + lane] = v0[u];
                    frow[E + c * E + lane + 64] = v1[u];
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // table half of the head (tsum != NULL): tsum[row] = obs_b + sum_j P_j[cat[row, j]], P_j = cat_emb @ W_obs[flatten slot j]
    // ([H, 256] per slot, built at load), i.e. the Flatten(category_emb) part of the head as Cn row gathers of 1 KB.  It
    // depends on the ids only, so it rides in this kernel's latency shadow: 12 rows requested here and summed behind the
    // Gram matrix, the rest requested there and summed at the end; lane owns 4 consecutive outputs.  The head GEMM adds
    // tsum in its epilogue.
    constexpr int TCH = 12;
    float4 tacc = make_float4(0.f, 0.f, 0.f, 0.f), tv[TCH];
    const bool do_t = tsum != nullptr && Cn <= 2 * TCH;
    if (do_t) {
        tacc = reinterpret_cast<const float4*>(obs_b)[lane];
#pragma unroll
        for (int u = 0; u < TCH; ++u)
            tv[u] = reinterpret_cast<const float4*>(ptab + ((size_t)min(u, Cn - 1) * H + __shfl(myid, min(u, Cn - 1))) * OBS_DIM)[lane];
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool row_ok = li < Cn;
    const float* erow = sE + (row_ok ? li : 0) * LE + half * 4;
    if (h16 && E % 16 == 0) {
        // fp16x2 form of the Gram matrix E E^T (scorer_mode FP16X2; embedding tables range-checked at load): the same
        // fragment is the A and the B operand, 3 x E/16 f16 MFMAs instead of E/2 fp32 ones
        const float* er16 = sE + (row_ok ? li : 0) * LE + half * 8;
        for (int kb = 0; kb < E / 16; ++kb) {
            float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
            if (row_ok) { f0 = *reinterpret_cast<const float4*>(er16 + kb * 16); f1 = *reinterpret_cast<const float4*>(er16 + kb * 16 + 4); }
            const float x[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
            half8_t fh, fl;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                half2_t h2, l2;
                split_h16_pair(x[e], x[e + 1], h2, l2);
                fh[e] = h2[0]; fh[e + 1] = h2[1];
                fl[e] = l2[0]; fl[e + 1] = l2[1];
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, fh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fl, acc, 0, 0, 0);
        }
    } else
    for (int kb = 0; kb < E / 8; ++kb) {
        float4 f = row_ok ? *reinterpret_cast<const float4*>(erow + kb * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, f.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, f.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, f.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, f.w, acc, 0, 0, 0);
    }
    if (do_t) {
#pragma unroll
        for (int u = 0; u < TCH; ++u)
            if (u < Cn) { tacc.x += tv[u].x; tacc.y += tv[u].y; tacc.z += tv[u].z; tacc.w += tv[u].w; }
#pragma unroll
        for (int u = 0; u < TCH; ++u)
            tv[u] = reinterpret_cast<const float4*>(ptab + ((size_t)min(TCH + u, Cn - 1) * H + __shfl(myid, min(TCH + u, Cn - 1))) * OBS_DIM)[lane];
    }
    // lane = query i (column li); register r of half h = key j = crow(r, h).  keras Attention: no scale, no mask.
    float m = -3.4e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (crow(r, half) < Cn) m = fmaxf(m, acc[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float z = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float ev = (crow(r, half) < Cn) ? CAT_EXP(acc[r] - m) : 0.f;
        acc[r] = ev;
        z += ev;
    }
    z += __shfl_xor(z, 32);
    const float inv = row_ok ? 1.f / z : 0.f;             // queries beyond Cn contribute nothing
    // GlobalAveragePooling1D over the Cn attended rows: column weight of key j = sum over queries i of w_ij
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] * inv;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        if (li == 0) sW[crow(r, half)] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float invc = 1.f / (float)Cn;
    for (int k = lane; k < E; k += 64) {
        float s = 0.f;
        for (int j = 0; j < Cn; ++j) s = fmaf(sW[j], sE[j * LE + k], s);
        frow[k] = s * invc;
    }
    if (do_t) {
#pragma unroll
        for (int u = 0; u < TCH; ++u)
            if (TCH + u < Cn) { tacc.x += tv[u].x; tacc.y += tv[u].y; tacc.z += tv[u].z; tacc.w += tv[u].w; }
        reinterpret_cast<float4*>(tsum + (size_t)row * OBS_DIM)[lane] = tacc;
    }
    // query = reduce_mean(seq_emb[cat[-10:]]) (utils.py:114-115): rows requested at the top, summed in id order
    const float invq = 1.f / (float)nq;
    if (E <= 128) {
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int u = 0; u < 10; ++u)
            if (u >= 10 - nq) { s0 += qv0[u]; s1 += qv1[u]; }
        if (lane < E) q[(size_t)row * E + lane] = s0 * invq;
        if (lane + 64 < E) q[(size_t)row * E + lane + 64] = s1 * invq;
    } else
    for (int k = lane; k < E; k += 64) {
        float s = 0.f;
        for (int c = Cn - nq; c < Cn; ++c) s += seq_emb[(size_t)__shfl(myid, c) * E + k];
        q[(size_t)row * E + k] = s * invq;
    }
}

// -------------------------------------------------------------------------------------------------
// k_cat_attn2: the same category branch for the default shape (E = 128, Cn <= 24) with HALF the LDS per row.
// k_cat_attn keeps a [Cn, E + 4] fp32 image of the row's embedding block in LDS (11.2 KB for Cn = 21): 4-row workgroups of
// 44.8 KB fit three to a CU = 12 rows in flight per CU, but an obs-sized launch at B = 4096 has 16 rows per CU, so it ran
// as one full round plus a one-third-full second round of the same dependent chain (ids -> gathers -> LDS -> Gram MFMAs ->
// softmax -> pooled row), and the reward-sized launch had 12 chains per CU in flight where the register file allows 16+.
// Here the image holds one 64-column half of the block at a time (5.8 KB per row): the Gram matrix S = E E^T is accumulated
// over the two halves (the k order of an MFMA sum is free), and the pooled row comes from the gathered registers (lane e
// still holds E[j][e], E[j][e + 64] of every row j) instead of re-reading the image.  22.9 KB per workgroup: seven fit a CU,
// the register file (<= 128 VGPRs) allows 16 waves = 16 rows: an obs-sized launch is ONE round.  Bit-identical arithmetic to
// k_cat_attn except the pooled row (register FMAs in the same j order: identical too).
template <int MAXC, bool EXACT>
__device__ __forceinline__ void cat_attn2_row(float* sE, int row, int lane, const int32_t* __restrict__ cat, int Cn_arg, int H,
                                              const float* __restrict__ cat_emb, const float* __restrict__ seq_emb,
                                              float* __restrict__ allf, int ldf, int off_c, float* __restrict__ q, int write_flat,
                                              int h16, const float* __restrict__ ptab, const float* __restrict__ obs_b,
                                              float* __restrict__ tsum) {
    constexpr int E = 128, HK = 64, LE = HK + 4, TCH = 4;       // MAXC: rows held in registers; EXACT: Cn == MAXC at compile time
    const int Cn = EXACT ? MAXC : Cn_arg;                       // (the default shape, Cn = 21: no clamps, no guards, fewer registers)
    const int half = lane >> 5, li = lane & 31;
    float* sW = sE + Cn * LE;
    const int32_t* crowp = cat + (size_t)row * Cn;
    float* frow = allf + (size_t)row * ldf + off_c;
    const int myid = (lane < Cn) ? min(max(crowp[lane], 0), H - 1) : 0;
    const int nq = min(10, Cn);
    // register budget (<= 128 for four waves per SIMD): the query rows are requested first and folded into two sums as soon as
    // they are there (they return in request order, ahead of the 48 category-row requests behind them); the head-table rows
    // come as two chunks of 4 (requested at the top and at the half-K boundary) and the rest in one go behind the MFMAs
    float qv0[10], qv1[10];
#pragma unroll
    for (int u = 0; u < 10; ++u) {
        const float* src = seq_emb + (size_t)__builtin_amdgcn_readlane(myid, max(Cn - 10 + u, 0)) * E;      // wave-uniform: scalar base
        qv0[u] = src[lane];
        qv1[u] = src[lane + 64];
    }
    float v0[MAXC], v1[MAXC];
#pragma unroll
    for (int u = 0; u < MAXC; ++u) {
        const float* src = cat_emb + (size_t)__builtin_amdgcn_readlane(myid, min(u, Cn - 1)) * E;
        v0[u] = src[lane];
        v1[u] = src[lane + 64];
    }
    __builtin_amdgcn_sched_barrier(0);
    float q0 = 0.f, q1 = 0.f;
#pragma unroll
    for (int u = 0; u < 10; ++u)
        if (u >= 10 - nq) { q0 += qv0[u]; q1 += qv1[u]; }
    // pinned HERE: the sums are only stored at the very end, and LLVM's sinking pass otherwise moves the twenty adds down there -
    // which kept the twenty loaded values alive across the whole kernel and made the allocator spill eight of them, each spill a
    // "s_waitcnt vmcnt(0); scratch_store" right behind its load: eight serialised memory round trips at the top of every row
    asm volatile("" : "+v"(q0), "+v"(q1));
    __builtin_amdgcn_sched_barrier(0);
    const bool do_t = tsum != nullptr;
    float4 tacc = make_float4(0.f, 0.f, 0.f, 0.f), tv[TCH];
    auto t_request = [&](int c) {
#pragma unroll
        for (int u = 0; u < TCH; ++u) {
            const int j = min(c * TCH + u, Cn - 1);
            tv[u] = reinterpret_cast<const float4*>(ptab + ((size_t)j * H + __builtin_amdgcn_readlane(myid, j)) * OBS_DIM)[lane];
        }
    };
    auto t_add = [&](int c) {
#pragma unroll
        for (int u = 0; u < TCH; ++u)
            if (c * TCH + u < Cn) { tacc.x += tv[u].x; tacc.y += tv[u].y; tacc.z += tv[u].z; tacc.w += tv[u].w; }
    };
    // stage c: sum chunk c - 1, request chunk c (same summation order as k_cat_attn: bias, then rows 0, 1, 2, ...)
#define CAT2_T_STAGE(c) do { if (do_t) { if ((c) > 0) t_add((c) - 1); if ((c) * TCH < Cn && (c) * TCH < MAXC) t_request(c); } } while (0)
    if (do_t) tacc = reinterpret_cast<const float4*>(obs_b)[lane];
    CAT2_T_STAGE(0);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool row_ok = li < Cn;
    const float* er32 = sE + (row_ok ? li : 0) * LE + half * 4;
    const float* er16 = sE + (row_ok ? li : 0) * LE + half * 8;
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        if (ph) {                                   // every lane has finished reading the first half
            CAT2_T_STAGE(1);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
#pragma unroll
        for (int u = 0; u < MAXC; ++u)
            if (u < Cn) {
                sE[u * LE + lane] = ph ? v1[u] : v0[u];
                if (write_flat) frow[E + u * E + ph * 64 + lane] = ph ? v1[u] : v0[u];       // Flatten()(category_emb)
            }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (h16) {
#pragma unroll
            for (int kb = 0; kb < HK / 16; ++kb) {
                float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
                if (row_ok) { f0 = *reinterpret_cast<const float4*>(er16 + kb * 16); f1 = *reinterpret_cast<const float4*>(er16 + kb * 16 + 4); }
                const float x[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                half8_t fh, fl;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    half2_t h2, l2;
                    split_h16_pair(x[e], x[e + 1], h2, l2);
                    fh[e] = h2[0]; fh[e + 1] = h2[1];
                    fl[e] = l2[0]; fl[e + 1] = l2[1];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, fh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fl, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < HK / 8; ++kb) {
                float4 f = row_ok ? *reinterpret_cast<const float4*>(er32 + kb * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, f.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, f.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, f.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, f.w, acc, 0, 0, 0);
            }
        }
    }
    // (the second half of the block stays in the LDS image: the pooled row reads its columns 64..127 from there, columns
    // 0..63 from the registers - the v1 registers are free from here on)
    // the rest of the head-table rows (8 .. Cn-1) are requested HERE - the second-half registers and the MFMA operands are dead -
    // instead of one chunk of 4 per later stage, each of which exposed a memory round trip in front of the next.  Default shape
    // (EXACT): all 13 in one go, summed at the end; other shapes: two batches of 8 (the second behind the softmax).
    constexpr int TW = MAXC - 2 * TCH, TWB = EXACT ? TW : 8;
    float4 tw[TWB];
    auto rest_request = [&](int b0) {
#pragma unroll
        for (int u = 0; u < TWB; ++u) {
            const int j = min(2 * TCH + b0 + u, Cn - 1);
            tw[u] = reinterpret_cast<const float4*>(ptab + ((size_t)j * H + __builtin_amdgcn_readlane(myid, j)) * OBS_DIM)[lane];
        }
    };
    auto rest_add = [&](int b0) {
#pragma unroll
        for (int u = 0; u < TWB; ++u)
            if (2 * TCH + b0 + u < Cn) { tacc.x += tw[u].x; tacc.y += tw[u].y; tacc.z += tw[u].z; tacc.w += tw[u].w; }
    };
    if (do_t) {
        t_add(1);
        rest_request(0);
    }
    float m = -3.4e38f;
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (crow(r, half) < Cn) m = fmaxf(m, acc[r]);
    m = fmaxf(m, __shfl_xor(m, 32));
    float z = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float ev = (crow(r, half) < Cn) ? CAT_EXP(acc[r] - m) : 0.f;
        acc[r] = ev;
        z += ev;
    }
    z += __shfl_xor(z, 32);
    const float inv = row_ok ? 1.f / z : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = acc[r] * inv;
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 16);
        if (li == 0) sW[crow(r, half)] = v;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (do_t && TWB < TW) {
        rest_add(0);
        if (2 * TCH + TWB < Cn) rest_request(TWB);
    }
    {
        const float invc = 1.f / (float)Cn;
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
            if (j < Cn) {
                const float w = sW[j];
                s0 = fmaf(w, v0[j], s0);
                s1 = fmaf(w, sE[j * LE + lane], s1);
            }
        frow[lane] = s0 * invc;
        frow[lane + 64] = s1 * invc;
    }
    const float invq = 1.f / (float)nq;
    q[(size_t)row * E + lane] = q0 * invq;
    q[(size_t)row * E + lane + 64] = q1 * invq;
    if (do_t) {
        rest_add(TWB < TW ? TWB : 0);
        reinterpret_cast<float4*>(tsum + (size_t)row * OBS_DIM)[lane] = tacc;
    }
#undef CAT2_T_STAGE
}

template <int MAXC, bool EXACT>
__global__ __launch_bounds__(256, 4) void k_cat_attn2(const int32_t* __restrict__ cat, int R, int Cn, int H,
                                                      const float* __restrict__ cat_emb, const float* __restrict__ seq_emb,
                                                      float* __restrict__ allf, int ldf, int off_c, float* __restrict__ q, int write_flat,
                                                      int h16, const float* __restrict__ ptab, const float* __restrict__ obs_b,
                                                      float* __restrict__ tsum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    cat_attn2_row<MAXC, EXACT>(reinterpret_cast<float*>(smem) + (size_t)wave * (Cn * (64 + 4) + 32), row, lane, cat, Cn, H, cat_emb, seq_emb, allf, ldf,
                  off_c, q, write_flat, h16, ptab, obs_b, tsum);
}

// -------------------------------------------------------------------------------------------------
// k_cat_attn2g<G>: the category branch for a launch whose rows come in groups of G that share every category id but the last.
// That is what the reward forward is (rl4rs/env/slate.py:117-131, get_complete_states): the category row of complete state j of
// an env is [user portrait (10), sequence id, ALL nine chosen items, item j] - 20 of the 21 ids are the env's, only the last
// is the row's - and the per-row kernel gathers the same 20 embedding rows (512 B each), the same 20 head-table rows (1 KB
// each) and nine of the ten query rows once PER ROW: 37 KB of gathers per row, 83 vector loads per wave in a six-trip
// dependent chain, which is what bounded the reward-sized launch (166 us for 32 768 rows = the L2 at ~8 TB/s).
// Here a workgroup is one group, wave w = row w of it:
//   * the Cn - 1 shared embedding rows, their head-table rows and the nine shared query rows are gathered ONCE per group into LDS
//     (wave w takes rows w, w + G, ..: at most 3 + 3 + 2 requests per wave), each wave gathers only its own last id's three
//     rows beside them: ~14 vector loads per wave in ONE round trip instead of 83 in six;
//   * one barrier, then every wave runs its row off the shared image: the Gram tile / softmax / pooled row (lane li = Cn - 1
//     reads the wave's private row; all k ascending as in k_cat_attn), the query sum and the head addend - each summed in the
//     per-row kernel's own order (shared rows in id order, the row's own term last), so every output is BIT-IDENTICAL to
//     k_cat_attn2 (tests/test_gpu_dien.py::test_group_category_kernel_is_bit_identical).
// Nothing is assumed about the ids: each wave compares its row's with the staged ("lead") row's while the loads are in flight,
// and rows that differ are served by further passes of the same loop with the next open row as the lead.
template <int G>
__global__ __launch_bounds__(64 * G, 4) void k_cat_attn2g(const int32_t* __restrict__ cat, int Cn, int H,
                                                          const float* __restrict__ cat_emb, const float* __restrict__ seq_emb,
                                                          float* __restrict__ allf, int ldf, int off_c, float* __restrict__ q, int write_flat,
                                                          int h16, const float* __restrict__ ptab, const float* __restrict__ obs_b,
                                                          float* __restrict__ tsum) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 128, LE = E + 4, MAXS = 23;              // MAXS: shared rows at most (Cn <= 24)
    constexpr int RJ = (MAXS + G - 1) / G, RQ = (9 + G - 1) / G;   // shared rows / shared query rows per wave
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31;
    const int CS = Cn - 1;                                     // shared rows
    float* fs = reinterpret_cast<float*>(smem);
    int* s_flag = reinterpret_cast<int*>(fs);                  // [16] per-wave "my ids are row 0's"
    float* sQ = fs + 16;                                       // [9][E] the nine shared query rows
    float* sT = sQ + 9 * E;                                    // [CS][OBS_DIM] shared head-table rows
    float* sE = sT + (size_t)CS * OBS_DIM;                     // [CS][LE] shared embedding rows
    float* sP = sE + (size_t)CS * LE + (size_t)wave * (LE + 32);   // per wave: own last row [LE] + column weights [32]
    float* sW = sP + LE;
    const int row0 = blockIdx.x * G, row = row0 + wave;
    const int myid = (lane < Cn) ? min(max(cat[(size_t)row * Cn + lane], 0), H - 1) : 0;
    const bool do_t = tsum != nullptr;
    // this row's own last id: embedding row, query row, head-table row (requested once, kept across passes)
    const int last = __builtin_amdgcn_readlane(myid, Cn - 1);
    const float pc0 = cat_emb[(size_t)last * E + lane], pc1 = cat_emb[(size_t)last * E + lane + 64];
    const float pq0 = seq_emb[(size_t)last * E + lane], pq1 = seq_emb[(size_t)last * E + lane + 64];
    float4 pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (do_t) pt = reinterpret_cast<const float4*>(ptab + ((size_t)CS * H + last) * OBS_DIM)[lane];
    // Pass p stages the shared rows of the first row not served yet (the "lead"; pass 0: row 0) and serves every row whose
    // shared ids equal the lead's.  A group of complete states is one pass; a group that shares nothing is G passes - the same
    // results either way (each row is always computed from exactly its own ids), the sharing only decides the speed.
    uint32_t done = 0;                                         // bit w: row w served (kept identically by every wave)
    int lead = 0;
    while (true) {
        const int id0 = (lane < Cn) ? min(max(cat[(size_t)(row0 + lead) * Cn + lane], 0), H - 1) : 0;
        // every wave gathers its share of the lead's shared rows into LDS: embedding rows and head-table rows j = wave, wave + G, ..,
        // query rows u = wave, wave + G (all requested before the first is stored)
        {
            float v0[RJ], v1[RJ];
            float4 tv[RJ];
            float qv0[RQ], qv1[RQ];
#pragma unroll
            for (int u = 0; u < RJ; ++u) {
                const int j = wave + u * G;                    // wave-uniform
                v0[u] = 0.f;
                v1[u] = 0.f;
                tv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (j < CS) {
                    const int id = __builtin_amdgcn_readlane(id0, j);
                    v0[u] = cat_emb[(size_t)id * E + lane];
                    v1[u] = cat_emb[(size_t)id * E + lane + 64];
                    if (do_t) tv[u] = reinterpret_cast<const float4*>(ptab + ((size_t)j * H + id) * OBS_DIM)[lane];
                }
            }
#pragma unroll
            for (int u = 0; u < RQ; ++u) {
                const int k = wave + u * G;
                qv0[u] = 0.f;
                qv1[u] = 0.f;
                if (k < 9) {
                    const float* src = seq_emb + (size_t)__builtin_amdgcn_readlane(id0, Cn - 10 + k) * E;
                    qv0[u] = src[lane];
                    qv1[u] = src[lane + 64];
                }
            }
#pragma unroll
            for (int u = 0; u < RJ; ++u) {
                const int j = wave + u * G;
                if (j < CS) {
                    sE[j * LE + lane] = v0[u];
                    sE[j * LE + lane + 64] = v1[u];
                    if (do_t) reinterpret_cast<float4*>(sT + (size_t)j * OBS_DIM)[lane] = tv[u];
                }
            }
#pragma unroll
            for (int u = 0; u < RQ; ++u) {
                const int k = wave + u * G;
                if (k < 9) { sQ[k * E + lane] = qv0[u]; sQ[k * E + lane + 64] = qv1[u]; }
            }
        }
        sP[lane] = pc0;                                        // (behind the pass's requests; the same values every pass)
        sP[lane + 64] = pc1;
        {
            const bool same = __all((lane >= CS) || (myid == id0));
            if (lane == 0) s_flag[wave] = same ? 1 : 0;
        }
        __syncthreads();
        uint32_t hit = 0;
#pragma unroll
        for (int w = 0; w < G; ++w) hit |= (s_flag[w] != 0 ? 1u : 0u) << w;
        hit &= ~done;
        if ((hit >> wave) & 1u) {                              // this row is served by this pass
            float* frow = allf + (size_t)row * ldf + off_c;
            if (write_flat) {                                          // Flatten()(category_emb) of the GEMM-form head
                for (int u = 0; u < Cn; ++u) {
                    const float* er = (u < CS) ? sE + u * LE : sP;
                    frow[E + u * E + lane] = er[lane];
                    frow[E + u * E + lane + 64] = er[lane + 64];
                }
            }
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bool row_ok = li < Cn;
            const float* erow = (li < CS) ? sE + li * LE : sP;         // lanes beyond Cn read the private row, results unused
            if (h16) {
                const float* er16 = erow + half * 8;
#pragma unroll
                for (int kb = 0; kb < E / 16; ++kb) {
                    float4 f0 = make_float4(0.f, 0.f, 0.f, 0.f), f1 = f0;
                    if (row_ok) { f0 = *reinterpret_cast<const float4*>(er16 + kb * 16); f1 = *reinterpret_cast<const float4*>(er16 + kb * 16 + 4); }
                    const float x[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
                    half8_t fh, fl;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        half2_t h2, l2;
                        split_h16_pair(x[e], x[e + 1], h2, l2);
                        fh[e] = h2[0]; fh[e + 1] = h2[1];
                        fl[e] = l2[0]; fl[e + 1] = l2[1];
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, fh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, fl, acc, 0, 0, 0);
                }
            } else {
                const float* er32 = erow + half * 4;
#pragma unroll
                for (int kb = 0; kb < E / 8; ++kb) {
                    float4 f = row_ok ? *reinterpret_cast<const float4*>(er32 + kb * 8) : make_float4(0.f, 0.f, 0.f, 0.f);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.x, f.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.y, f.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.z, f.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.w, f.w, acc, 0, 0, 0);
                }
            }
            float m = -3.4e38f;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (crow(r, half) < Cn) m = fmaxf(m, acc[r]);
            m = fmaxf(m, __shfl_xor(m, 32));
            float z = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float ev = (crow(r, half) < Cn) ? CAT_EXP(acc[r] - m) : 0.f;
                acc[r] = ev;
                z += ev;
            }
            z += __shfl_xor(z, 32);
            const float inv = row_ok ? 1.f / z : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[r] * inv;
                v += __shfl_xor(v, 1);
                v += __shfl_xor(v, 2);
                v += __shfl_xor(v, 4);
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                if (li == 0) sW[crow(r, half)] = v;
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            {
                const float invc = 1.f / (float)Cn;
                float s0 = 0.f, s1 = 0.f;
                for (int j = 0; j < CS; ++j) {
                    const float w = sW[j];
                    s0 = fmaf(w, sE[j * LE + lane], s0);
                    s1 = fmaf(w, sE[j * LE + lane + 64], s1);
                }
                s0 = fmaf(sW[CS], pc0, s0);
                s1 = fmaf(sW[CS], pc1, s1);
                frow[lane] = s0 * invc;
                frow[lane + 64] = s1 * invc;
            }
            // query = mean of the last ten ids' rows, head addend = bias + table rows: both summed in the per-row kernel's order
            // (shared rows in id order, this row's own one last)
            const float invq = 1.f / (float)min(10, Cn);
            float q0 = 0.f, q1 = 0.f;
#pragma unroll
            for (int u = 0; u < 9; ++u) { q0 += sQ[u * E + lane]; q1 += sQ[u * E + lane + 64]; }
            q[(size_t)row * E + lane] = (q0 + pq0) * invq;
            q[(size_t)row * E + lane + 64] = (q1 + pq1) * invq;
            if (do_t) {
                float4 t = reinterpret_cast<const float4*>(obs_b)[lane];
                for (int j = 0; j < CS; ++j) {
                    const float4 v = reinterpret_cast<const float4*>(sT + (size_t)j * OBS_DIM)[lane];
                    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
                }
                t.x += pt.x; t.y += pt.y; t.z += pt.z; t.w += pt.w;
                reinterpret_cast<float4*>(tsum + (size_t)row * OBS_DIM)[lane] = t;
            }
        }
        done |= hit;
        if (done == (1u << G) - 1u) break;
        __syncthreads();                                       // the served rows have read the image; the next pass overwrites it
        lead = __ffs(~done) - 1;
    }
}

inline size_t cat_attn2g_smem(int G, int Cn) {
    const size_t grp = 16 + 9 * 128 + (size_t)(Cn - 1) * (OBS_DIM + 132) + (size_t)G * (132 + 32);
    return grp * 4;
}

// -------------------------------------------------------------------------------------------------
// Recurrent kernel: GRU (NH = E, first layer, writes every state) or AUGRU (NH = 2E, writes the final state).
//   gates  [r|u] = sigmoid(xg_t + h @ Wg)      Wg packed [2*NW tiles][NH/8][64 lanes][4]
//   cand   c     = tanh(xc_t + (r*h) @ Wc)     Wc packed [NW tiles][NH/8][64][4]
//   GRU:   h' = u*h + (1-u)*c        AUGRU: u <- (1 - a_t) * u first (deepctr VecAttGRUCell)
// xg_t / xc_t are the cached input projections (bias folded in):
//   GRU  : row = table[ids[row,t]]            (xld = 3*NH, layout [2NH gates | NH cand])
//   AUGRU: row = proj[slot(row)*L + t] + xoff (xld = proj ld)
// Workgroup = NH/32 waves, 32 rows; wave w owns hidden columns [32w, 32w+32) of r, u, c and h.
// grid = (row tiles, sequence inputs): all sequence inputs of a forward share ONE launch.
//
// Schedule (per step, two phases separated by one barrier each):
//   * the h-side weight fragments stream from L2 through a 2-deep register ring, U k-blocks (8 k each) per
//     slot: the loads of group g+1 are issued before the MFMAs of group g, and the first group of the NEXT
//     phase is issued during the last group of the current one (weights do not depend on the recurrence);
//   * the cached input projections of a phase are requested right after its first weight group and added in
//     the phase's epilogue (thousands of MFMA cycles later), so their HBM latency never sits in front of an MFMA.

#ifndef RL4RS_FAST_ACT
#define RL4RS_FAST_ACT 1
#endif
// sigmoid / tanh on the hardware exp2 + rcp (abs error ~1e-7; the gates are convex-combined into h, so
// absolute error is what propagates).  -DRL4RS_FAST_ACT=0 selects the libm-accurate forms.
__device__ __forceinline__ float gate_sigmoid(float x) {
#if RL4RS_FAST_ACT
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
#else
    return 1.f / (1.f + expf(-x));
#endif
}
__device__ __forceinline__ float gate_tanh(float x) {
#if RL4RS_FAST_ACT
    // 1 - 2/(exp(2x)+1); exp2 saturates cleanly to 0 / inf
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
#else
    return tanhf(x);
#endif
}

// AB = ablation bits for timing experiments only (results are wrong when non-zero):
//   1 no activation math, 2 weight fragments loaded once (no streaming), 4 no x-projection loads, 8 no barriers
struct f4bits { float x, y, z, w; };
// NOTE: the b128 builtin's result type is a 128-bit value that does NOT convert element-wise to an
// ext_vector typedef on this compiler (it splats the low dword): bit_cast the `auto` result instead.
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    static_assert(sizeof(v) == 16, "b128");
    f4bits f = __builtin_bit_cast(f4bits, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, soff, 0));
}

template <int NH, bool AUGRU, int U, int AB = 0, bool SAVE = false>
__global__ __launch_bounds__(NH * 2) void k_recur(RecurArgs a) {
    constexpr int U2 = 2 * U;
    constexpr int NW = NH / 32, KB = NH / 8, LDH = NH + 4, NG = KB / U, NG2 = KB / U2;
    static_assert(KB % U2 == 0 && NG >= 2 && NG2 >= 2, "k-block groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hb = reinterpret_cast<float*>(smem);          // [32][LDH]
    float* rhb = hb + 32 * LDH;                          // [32][LDH]
    float* s_att = rhb + 32 * LDH;                       // AUGRU: [32][L+1]
    int32_t* s_ids = reinterpret_cast<int32_t*>(rhb + 32 * LDH);   // GRU: [32][L+1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * 32;
    const int sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;
    const int col = wave * 32 + li;
    const int xld4 = (int)a.xld * 4;

    // buffer descriptors: all streamed operands are addressed as (SGPR base, 32-bit lane offset, scalar offset),
    // so no load costs address VGPRs and the unrolled ring below stays inside the register budget
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xbase[sq]), 0, (int)a.xbytes, 0x00020000);
    const int vl16 = lane * 16;
    const int so_r = wave * KB * 1024, so_u = (NW + wave) * KB * 1024, so_c = wave * KB * 1024;   // + kb * 1024

    for (int i = tid; i < 32 * LDH; i += NH * 2) hb[i] = 0.f;
    for (int i = tid; i < 32 * L; i += NH * 2) {
        int r = i / L, t = i - r * L;
        int gr = min(row0 + r, a.n_rows - 1);
        if (SAVE) s_att[r * LDT + t] = a.sv_att[sq] ? a.sv_att[sq][(size_t)gr * L + t] : 0.f;
        else if (AUGRU) s_att[r * LDT + t] = a.att[(size_t)sq * a.att_stride + (size_t)gr * L + t];
        else s_ids[r * LDT + t] = a.ids[(size_t)gr * L + t];
    }
    // byte offset of each row's x-projection at t = 0 (AUGRU: slot * L * xld)
    uint32_t* s_xoff = reinterpret_cast<uint32_t*>(s_att + 32 * LDT);
    if (tid < 32) {
        int gr = min(row0 + tid, a.n_rows - 1);
        s_xoff[tid] = AUGRU ? (uint32_t)a.slots[(size_t)sq * a.slots_stride + gr / a.group] * (uint32_t)L * (uint32_t)xld4 : 0u;
    }
    f32x16 h_own;
#pragma unroll
    for (int r = 0; r < 16; ++r) h_own[r] = 0.f;
    const float* arow = hb + li * LDH + half * 4;
    const float* rrow = rhb + li * LDH + half * 4;
    const int xcol4 = (a.xoff + col) * 4;
    __syncthreads();

    // x-projection fetch of one gate block (0 = r, 1 = u, 2 = c) of step t into dst
    auto load_x = [&](f32x16& dst, int t, int block) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int voff = AUGRU ? (int)s_xoff[crow(r, half)] + xcol4
                             : s_ids[crow(r, half) * LDT + t] * xld4 + xcol4;
            dst[r] = buf_load1(rs_x, voff, (AUGRU ? t * xld4 : 0) + (SAVE ? a.sv_blk[block] : block) * NH * 4);
        }
    };
    f32x16 xr_, xu_, xc_;
    float4 wr[2][U], wu[2][U], a1[2][U], wc[2][U2], a2[2][U2];
#pragma unroll
    for (int i = 0; i < U; ++i) {
        wr[0][i] = buf_load4(rs_wg, vl16, so_r + i * 1024);
        wu[0][i] = buf_load4(rs_wg, vl16, so_u + i * 1024);
    }
    load_x(xr_, 0, 0);
    load_x(xu_, 0, 1);

    for (int t = 0; t < L; ++t) {
        f32x16 acc_r, acc_u;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_r[r] = 0.f; acc_u[r] = 0.f; }
        // ---- phase 1: gates.  Fully unrolled 2-deep ring: group g+1 is requested before group g is consumed.
#pragma unroll
        for (int i = 0; i < U; ++i) a1[0][i] = *reinterpret_cast<const float4*>(arow + i * 8);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int cb = g & 1, nb = cb ^ 1;
            if (g + 1 < NG) {
#pragma unroll
                for (int i = 0; i < U; ++i) {
                    const int kb = (g + 1) * U + i;
                    if (!(AB & 2)) {
                        wr[nb][i] = buf_load4(rs_wg, vl16, so_r + kb * 1024);
                        wu[nb][i] = buf_load4(rs_wg, vl16, so_u + kb * 1024);
                    }
                    a1[nb][i] = *reinterpret_cast<const float4*>(arow + kb * 8);
                }
            } else {
#pragma unroll
                for (int i = 0; i < U2; ++i)
                    if (!(AB & 2) || t == 0) wc[0][i] = buf_load4(rs_wc, vl16, so_c + i * 1024);   // first group of phase 2
                // candidate x-projection: consumed after phase 2.  vmcnt retires in order; the next wait that
                // covers these loads is a full epilogue + barrier + one MFMA group away.
                if (!(AB & 4) || t == 0) load_x(xc_, t, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < U; ++i) {
                acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].x, wr[cb][i].x, acc_r, 0, 0, 0);
                acc_u = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].x, wu[cb][i].x, acc_u, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].y, wr[cb][i].y, acc_r, 0, 0, 0);
                acc_u = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].y, wu[cb][i].y, acc_u, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].z, wr[cb][i].z, acc_r, 0, 0, 0);
                acc_u = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].z, wu[cb][i].z, acc_u, 0, 0, 0);
                acc_r = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].w, wr[cb][i].w, acc_r, 0, 0, 0);
                acc_u = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[cb][i].w, wu[cb][i].w, acc_u, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float rg, ug;
            if ((!AUGRU || SAVE) && a.hard_gates) {          // keras recurrent_activation = hard_sigmoid (uniform branch)
                rg = fminf(fmaxf(0.2f * (acc_r[r] + xr_[r]) + 0.5f, 0.f), 1.f);
                ug = fminf(fmaxf(0.2f * (acc_u[r] + xu_[r]) + 0.5f, 0.f), 1.f);
            } else {
                rg = (AB & 1) ? acc_r[r] + xr_[r] : gate_sigmoid(acc_r[r] + xr_[r]);
                ug = (AB & 1) ? acc_u[r] + xu_[r] : gate_sigmoid(acc_u[r] + xu_[r]);
            }
            acc_u[r] = ug;
            rhb[crow(r, half) * LDH + col] = rg * h_own[r];
            if (SAVE && row0 + crow(r, half) < a.n_rows) {
                const size_t si = ((size_t)(row0 + crow(r, half)) * L + t) * NH + col;
                a.sv_r[sq][si] = rg;
                a.sv_u[sq][si] = ug;
                a.sv_rh[sq][si] = rg * h_own[r];
            }
        }
        if (!(AB & 8)) __syncthreads();
        // ---- phase 2: candidate + state update
        f32x16 acc_c;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_c[r] = 0.f;
#pragma unroll
        for (int i = 0; i < U2; ++i) a2[0][i] = *reinterpret_cast<const float4*>(rrow + i * 8);
#pragma unroll
        for (int g = 0; g < NG2; ++g) {
            const int cb = g & 1, nb = cb ^ 1;
            if (g + 1 < NG2) {
#pragma unroll
                for (int i = 0; i < U2; ++i) {
                    const int kb = (g + 1) * U2 + i;
                    if (!(AB & 2)) wc[nb][i] = buf_load4(rs_wc, vl16, so_c + kb * 1024);
                    a2[nb][i] = *reinterpret_cast<const float4*>(rrow + kb * 8);
                }
            } else {
#pragma unroll
                for (int i = 0; i < U; ++i)
                    if (!(AB & 2)) {                                              // first group of the next phase 1
                        wr[0][i] = buf_load4(rs_wg, vl16, so_r + i * 1024);
                        wu[0][i] = buf_load4(rs_wg, vl16, so_u + i * 1024);
                    }
                if (t + 1 < L && !(AB & 4)) {                                     // consumed after the next phase 1
                    load_x(xr_, t + 1, 0);
                    load_x(xu_, t + 1, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < U2; ++i) {
                acc_c = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[cb][i].x, wc[cb][i].x, acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[cb][i].y, wc[cb][i].y, acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[cb][i].z, wc[cb][i].z, acc_c, 0, 0, 0);
                acc_c = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[cb][i].w, wc[cb][i].w, acc_c, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float c = (AB & 1) ? acc_c[r] + xc_[r] : gate_tanh(acc_c[r] + xc_[r]);
            float u = acc_u[r];
            if (AUGRU) u = (1.0f - s_att[crow(r, half) * LDT + t]) * u;
            float hn = u * h_own[r] + (1.0f - u) * c;
            h_own[r] = hn;
            hb[crow(r, half) * LDH + col] = hn;
            if (SAVE && row0 + crow(r, half) < a.n_rows) {
                const size_t si = ((size_t)(row0 + crow(r, half)) * L + t) * NH + col;
                a.sv_c[sq][si] = c;
                a.sv_h[sq][si] = hn;
            }
            if (!AUGRU && row0 + crow(r, half) < a.n_rows) {
                if (!a.final_only) {
                    int64_t orow = ((int64_t)a.slot_base + row0 + crow(r, half)) * L + t;
                    a.out[orow * a.out_ld + a.out_off + col] = hn;
                } else if (t == L - 1) {
                    a.out[((int64_t)a.slot_base + row0 + crow(r, half)) * a.out_ld + a.out_off + col] = hn;
                }
            }
        }
        if (!(AB & 8)) __syncthreads();
    }
    if (AUGRU && !SAVE) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row0 + crow(r, half) < a.n_rows)
                a.out[(int64_t)(row0 + crow(r, half)) * a.out_ld + a.out_off + sq * a.out_seq_off + col] = h_own[r];
    }
}

// -------------------------------------------------------------------------------------------------
// AUGRU with fp16x2 operand splitting (scorer_mode RL4RS_SCORER_FP16X2, the AUTO default when the weights fit fp16 range).
//   a = a_hi + a_lo,  a_hi = fp16(a), a_lo = fp16(a - a_hi)      (both operands; weights are split at load time)
//   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi                       (3 v_mfma_f32_32x32x16_f16 per 16-wide k-block)
// The dropped a_lo*b_lo term and the fp16 rounding of the lo parts leave a relative error ~2^-22 per product
// (fp32 keeps 2^-24); products are exact in the fp32 accumulator.  h and r*h live in (-1,1), so the un-scaled lo parts
// only reach fp16 subnormals (absolute error <= 2^-25) - no scaling needed; the weights are range-checked at load.
// The matrix pipe runs this 16/3 = 5.3x faster than the fp32 form at the SAME weight bytes (2 planes x 2 B).

// Schedule ("gate-sliced", MT row tiles of 32 rows per workgroup, wave w owns hidden columns [32w, 32w+32)):
//   one step = 3*KB weight items (gate g = r,u,c ; k-block kb), each item = 3*MT MFMAs on ONE weight fragment pair
//     slot R  (items  0..KB-1)  : acc_r += h  Wr                    - matrix pipe only
//     slot U  (items KB..2KB-1) : acc_u += h  Wu    || VALU: r = sigmoid(acc_r), r*h -> fp16 hi/lo planes (item kb
//                                                      retires accumulator register kb of every tile)
//     barrier
//     slot C  (items 2KB..3KB-1): acc_c += (r*h) Wc || VALU: u = (1 - a_t) sigmoid(acc_u)
//     exposed: c = tanh(acc_c), h' = u h + (1-u) c -> fp16 hi/lo planes ; barrier
//   Two of the three epilogues execute in the shadow of the MFMAs.  The cached input projections are loaded straight
//   INTO the accumulators (MFMA C-in), so they cost no registers; they are requested at points that are followed by a
//   long stretch without weight waits (the in-order vmcnt makes every later weight wait also wait for them: ~5K
//   cycles when they were issued inside slot R).  The weight fragments stream through a RING-deep register ring, LA =
//   RING-1 items ahead, across slots and steps; the first NRES items of a step stay resident in registers and the
//   next NLDS items in the LDS the operand planes leave free (each wave keeps and re-reads only its own fragments,
//   ds_read_b128 one item ahead: 10 of 34 streamed items leave the L1 / texture-address path, measured -1.5 %).
//   What bounds it (s_memtime marks of one workgroup, tools/h16_trace.py; PMC: matrix pipe ~41 % busy): the L1 /
//   texture-address path.  8 waves x 2 KB of weight fragments per item at 64 B/clk = 256 cycles against 192 cycles
//   of MFMA, plus ~3.3K cycles per step for the 384 dword-per-lane projection loads; the two waves of a SIMD
//   therefore take ~300 cycles per item pair, the exposed epilogue ~6K cycles per step (of ~23K).  Tried and measured
//   flat: separate accumulators for the hi*lo / lo*hi products (no dependent-MFMA stall to remove), ping-pong u/c
//   accumulator sets with all projections requested at the start of the epilogue, 64-row workgroups (MT = 2: halves
//   the weight bytes per row but spills at 256 registers; +2 % on the reward-sized launch), spreading the projection
//   loads over the items of slot C / the elements of the epilogue (2x SLOWER: every weight wait then sits behind an
//   HBM-latency load), staging the next step's projections in 48 extra registers requested at the start of the
//   epilogue (slot C drops from ~6K to ~3K cycles, but the 48 narrow loads of a wave take ~3.3K cycles of
//   texture-address time to ISSUE and that lands in the exposed epilogue: net 4-6 % slower).  The projection loads take
//   ~4.5K cycles to land - all CUs burst theirs at the same phase.  Resident weights are the
//   lever that worked (-6 % at 14 of 48 items).  Next: a 64-row form that fits, or LDS-DMA staging of the ring.
#ifdef RL4RS_H16_TRACE     // s_memtime marks of workgroup (0,0), steps 8..11: [wave][step][mark]
#define RL4RS_TR(k) do { if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && t >= 8 && t < 12) \
        a.trace[(wave * 4 + (t - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RL4RS_TR(k) do { } while (0)
#endif
template <int MT, int RING, int NRES>
__global__ __launch_bounds__(512) void k_augru_h16(RecurArgs a) {
    constexpr int NH = 256, NW = 8, KB = NH / 16, LDP = NH + 8, MR = MT * 32, NI = 3 * KB, LA = RING - 1;
    static_assert(NI % RING == 0 && RING >= 2 && RING <= KB && NRES >= 0 && NRES <= KB, "weight ring");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* hp_hi = reinterpret_cast<_Float16*>(smem);     // [MR][LDP] each
    _Float16* hp_lo = hp_hi + MR * LDP;
    _Float16* rp_hi = hp_lo + MR * LDP;
    _Float16* rp_lo = rp_hi + MR * LDP;
    float* s_att = reinterpret_cast<float*>(rp_lo + MR * LDP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * MR;
    const int sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;
    const int col = wave * 32 + li;
    const int xld4 = (int)a.xld * 4;
    // packed fp16 planes: [ntile][KB][plane hi/lo][64 lanes][8 halfs] -> 1 KB per (ntile, kb, plane)
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xbase[sq]), 0, (int)a.xbytes, 0x00020000);
    const int vl16 = lane * 16;
    const int so_r = wave * KB * 2048, so_u = (NW + wave) * KB * 2048, so_c = wave * KB * 2048;   // + kb*2048 + plane*1024

    for (int i = tid; i < 2 * MR * LDP; i += 512) hp_hi[i] = (_Float16)0.f;     // hi and lo planes of h
    for (int i = tid; i < MR * L; i += 512) {
        int r = i / L, t = i - r * L;
        int gr = min(row0 + r, a.n_rows - 1);
        s_att[r * LDT + t] = a.att[(size_t)sq * a.att_stride + (size_t)gr * L + t];
    }
    uint32_t* s_xoff = reinterpret_cast<uint32_t*>(s_att + MR * LDT);
    if (tid < MR) {
        int gr = min(row0 + tid, a.n_rows - 1);
        s_xoff[tid] = (uint32_t)a.slots[(size_t)sq * a.slots_stride + gr / a.group] * (uint32_t)L * (uint32_t)xld4;
    }
    const int aoff = li * LDP + half * 8;
    const int xcol4 = (a.xoff + col) * 4;
    // items NRES .. NRES+NLDS-1 of this wave, hi and lo planes: [item][plane][lane][8 halfs] at 1 KB per (item, plane)
    constexpr int NLDS = RL4RS_H16_NLDS;
    char* lds_w = reinterpret_cast<char*>(s_xoff + MR) + (size_t)wave * NLDS * 2048;
    __syncthreads();

    // weight item i of a step: gate i / KB, k-block i % KB (i is a compile-time constant after unrolling).
    // The three scalar bases are made opaque once per step so that the 96 per-item scalar offsets are re-derived
    // with one s_add each instead of being hoisted out of the step loop (which spills SGPRs).
    int sb_r = so_r, sb_u = so_u, sb_c = so_c;
    auto wload = [&](int i, half8_t& hi, half8_t& lo) {
        const int g = i / KB, kb = i % KB;
        if (g == 0) {
            hi = buf_load_h8(rs_wg, vl16, sb_r + kb * 2048); lo = buf_load_h8(rs_wg, vl16, sb_r + kb * 2048 + 1024);
        } else if (g == 1) {
            hi = buf_load_h8(rs_wg, vl16, sb_u + kb * 2048); lo = buf_load_h8(rs_wg, vl16, sb_u + kb * 2048 + 1024);
        } else {
            hi = buf_load_h8(rs_wc, vl16, sb_c + kb * 2048); lo = buf_load_h8(rs_wc, vl16, sb_c + kb * 2048 + 1024);
        }
    };
    // cached input projection of (tile m, step t, gate block) -> accumulator (C-in of the gate's MFMA chain)
    // (the per-row cache offsets are re-read from LDS at every use: keeping them live would cost 16*MT registers)
    auto load_x = [&](f32x16& dst, int m, int t, int block) {
        const uint32_t* px = s_xoff + m * 32 + 4 * half;
        asm volatile("" : "+v"(px));
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dst[r] = buf_load1(rs_x, (int)px[crow(r, 0)] + xcol4, t * xld4 + block * NH * 4);
    };
    f32x16 acc_r[MT], acc_u[MT], acc_c[MT], h_own[MT];
    half8_t wh[RING], wl[RING], ah[2][MT], al[2][MT];
    // The kernel is bound by the L1 / texture-address path, not by the matrix pipe: 8 waves x 2 KB of weight
    // fragments per item at 64 B/clk = 256 cycles against 192 cycles of MFMA (s_memtime marks, tools/h16_trace.py).
    // The first NRES items of a step therefore stay in registers for the whole kernel (NRES * 8 registers per wave).
    half8_t res_h[NRES > 0 ? NRES : 1], res_l[NRES > 0 ? NRES : 1];
#pragma unroll
    for (int i = 0; i < NRES; ++i) wload(i, res_h[i], res_l[i]);
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) h_own[m][r] = 0.f;
        load_x(acc_r[m], m, 0, 0);
        load_x(acc_u[m], m, 0, 1);
        load_x(acc_c[m], m, 0, 2);
    }
#pragma unroll
    for (int i = 0; i < NLDS; ++i) {
        half8_t hi, lo;
        wload(NRES + i, hi, lo);
        *reinterpret_cast<half8_t*>(lds_w + i * 2048 + vl16) = hi;
        *reinterpret_cast<half8_t*>(lds_w + i * 2048 + 1024 + vl16) = lo;
    }
#pragma unroll
    for (int i = NRES + NLDS; i < NRES + NLDS + LA; ++i) wload(i % NI, wh[i % RING], wl[i % RING]);
    // (the LDS-resident items are read back by the owning wave only: no barrier needed, the compiler orders the accesses)
    if (NLDS > 0) {
        wh[NRES % RING] = *reinterpret_cast<const half8_t*>(lds_w + vl16);
        wl[NRES % RING] = *reinterpret_cast<const half8_t*>(lds_w + 1024 + vl16);
    }

    bool out_of_range = false;
    const float k_r = a.k_r[sq][wave], k_u = a.k_u[sq][wave], k_c = a.k_c[sq][wave];   // activations of prescaled pre-activations (RecurArgs)
    const int TL = a.steps > 0 ? a.steps : L;
#pragma unroll 1
    for (int t = 0; t < TL; ++t) {
        asm volatile("" : "+s"(sb_r), "+s"(sb_u), "+s"(sb_c));
        RL4RS_TR(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            ah[0][m] = *reinterpret_cast<const half8_t*>(hp_hi + m * 32 * LDP + aoff);
            al[0][m] = *reinterpret_cast<const half8_t*>(hp_lo + m * 32 * LDP + aoff);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = i / KB, kb = i % KB, cb = i & 1, nb = cb ^ 1;
            if (i == KB) RL4RS_TR(1);
            if (i == 2 * KB) {
                RL4RS_TR(2);
                // r*h planes complete; the operand fragments of item 2KB are read here (never across the barrier)
                __syncthreads();
                RL4RS_TR(3);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[cb][m] = *reinterpret_cast<const half8_t*>(rp_hi + m * 32 * LDP + aoff);
                    al[cb][m] = *reinterpret_cast<const half8_t*>(rp_lo + m * 32 * LDP + aoff);
                }
            }
            // streamed items run LA ahead in the ring (the resident ones are skipped)
            if ((i + LA) % NI >= NRES + NLDS) wload((i + LA) % NI, wh[(i + LA) % RING], wl[(i + LA) % RING]);
            if ((i + 1) % NI >= NRES && (i + 1) % NI < NRES + NLDS) {       // next item lives in LDS: read it one item ahead
                wh[(i + 1) % RING] = *reinterpret_cast<const half8_t*>(lds_w + ((i + 1) % NI - NRES) * 2048 + vl16);
                wl[(i + 1) % RING] = *reinterpret_cast<const half8_t*>(lds_w + ((i + 1) % NI - NRES) * 2048 + 1024 + vl16);
            }
            if (i == 2 * KB && t + 1 < L) {
                // next step's r-gate projection goes into the (retired) r accumulators
#pragma unroll
                for (int m = 0; m < MT; ++m) load_x(acc_r[m], m, t + 1, 0);
            }
            if (i + 1 < NI && i + 1 != 2 * KB) {
                const int kn = (i + 1) % KB;
                const _Float16* ph = (i + 1 < 2 * KB) ? hp_hi : rp_hi;
                const _Float16* pl = (i + 1 < 2 * KB) ? hp_lo : rp_lo;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[nb][m] = *reinterpret_cast<const half8_t*>(ph + m * 32 * LDP + aoff + kn * 16);
                    al[nb][m] = *reinterpret_cast<const half8_t*>(pl + m * 32 * LDP + aoff + kn * 16);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            const half8_t bh = i < NRES ? res_h[i < NRES ? i : 0] : wh[i % RING];
            const half8_t bl = i < NRES ? res_l[i < NRES ? i : 0] : wl[i % RING];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x16& acc = g == 0 ? acc_r[m] : (g == 1 ? acc_u[m] : acc_c[m]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][m], bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][m], bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][m], bh, acc, 0, 0, 0);
            }
            if (g == 1) {            // shadow: reset gate, accumulator register kb of every tile
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float rg = gate_sigmoid_k(acc_r[m][kb], k_r);
                    const float v = rg * h_own[m][kb];
                    const _Float16 vh = (_Float16)v;
                    rp_hi[(m * 32 + crow(kb, half)) * LDP + col] = vh;
                    rp_lo[(m * 32 + crow(kb, half)) * LDP + col] = (_Float16)(v - (float)vh);
                }
            } else if (g == 2) {     // shadow: attentional update gate
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float pre = acc_u[m][kb];
                    asm volatile("" : "+v"(pre));      // keeps element kb's chain in item kb (else all 16 cluster up front)
                    acc_u[m][kb] = (1.0f - s_att[(m * 32 + crow(kb, half)) * LDT + t]) * gate_sigmoid_k(pre, k_u);
                }
            }
            if (g != 0) {
                // a wave issues in order: without this the VALU chunk only overlaps the last MFMA of the item
#pragma unroll
                for (int q = 0; q < 3 * MT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    if (g == 1) __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);    // VALU in its shadow
                    else __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                if (g == 1) __builtin_amdgcn_sched_group_barrier(0x200, 2 * MT, 0);   // DS writes
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RL4RS_TR(4);
        // exposed epilogue: candidate + state update, then the next step's u / c projections into the accumulators
        // (requested here, the barrier and slot R cover their HBM latency before slot U needs them)
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float c = gate_tanh_k(acc_c[m][r], k_c);
                const float hn = __builtin_fmaf(acc_u[m][r], h_own[m][r] - c, c);     // u h + (1-u) c
                out_of_range |= !(fabsf(hn) < 6.0e4f);      // fp16 planes cannot carry it (also catches NaN)
                h_own[m][r] = hn;
                const _Float16 vh = (_Float16)hn;
                hp_hi[(m * 32 + crow(r, half)) * LDP + col] = vh;
                hp_lo[(m * 32 + crow(r, half)) * LDP + col] = (_Float16)(hn - (float)vh);
            }
        }
        if (t + 1 < L) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                load_x(acc_u[m], m, t + 1, 1);
                load_x(acc_c[m], m, t + 1, 2);
            }
        }
        RL4RS_TR(5);
        __syncthreads();
        RL4RS_TR(6);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (row0 + m * 32 + crow(r, half) < a.n_rows)
                a.out[(int64_t)(row0 + m * 32 + crow(r, half)) * a.out_ld + a.out_off + sq * a.out_seq_off + col] = h_own[m][r];
    if (out_of_range && a.range_flag) atomicOr(a.range_flag, 1);
}

// k_augru_x (augru_x.hpp) lives in its own translation unit, augru_x.hip (built WITH SLP vectorisation, see there)
int augru_x_prepare();
void augru_x_launch(int rows_per_wg, int n_seq, hipStream_t st, const RecurArgs& a);

// -------------------------------------------------------------------------------------------------
// First-layer GRU with the same fp16x2 operand splitting (scorer_mode fp16x2): NH = E = 128, 4 waves, 32 rows per
// workgroup, wave w owns hidden columns [32w, 32w+32) of r, u, c and h.  At this size nothing has to stream: the gate
// weights of a wave (2 gates x 8 k-blocks x hi/lo planes = 128 registers) stay in registers for the whole kernel, its
// candidate weights (16 KB) in LDS, and the input projections come from the per-item table as MFMA C-in (requested a
// slot or more ahead).  Per step 24 items x 3 v_mfma_f32_32x32x16_f16 = 2.3K cycles of matrix pipe against 12.3K for the
// exact-fp32 form (k_recur<128, false>), with the same schedule of epilogues: reset gate in the shadow of slot U,
// update gate in the shadow of slot C, candidate + blend exposed.  h stays in (-1, 1): no range flag needed.
__global__ __launch_bounds__(256) void k_gru_h16(RecurArgs a) {
    constexpr int NH = 128, NW = 4, KB = NH / 16, LDP = NH + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* hp_hi = reinterpret_cast<_Float16*>(smem);     // [32][LDP] each
    _Float16* hp_lo = hp_hi + 32 * LDP;
    _Float16* rp_hi = hp_lo + 32 * LDP;
    _Float16* rp_lo = rp_hi + 32 * LDP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * 32;
    const int L = a.L, LDT = L + 1;
    const int col = wave * 32 + li;
    const int xld4 = (int)a.xld * 4;
    int32_t* s_ids = reinterpret_cast<int32_t*>(rp_lo + 32 * LDP);                  // [32][LDT]
    char* lds_wc = smem + ((4 * 32 * LDP * 2 + 32 * LDT * 4 + 15) & ~15) + wave * KB * 2048;     // this wave's [kb][plane][lane][8]
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[0]), 0, 2 * NW * KB * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[0]), 0, NW * KB * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xbase[0]), 0, (int)a.xbytes, 0x00020000);
    const int vl16 = lane * 16;

    for (int i = tid; i < 2 * 32 * LDP; i += 256) hp_hi[i] = (_Float16)0.f;          // hi and lo planes of h
    for (int i = tid; i < 32 * L; i += 256) {
        const int r = i / L, t = i - r * L;
        s_ids[r * LDT + t] = a.ids[(size_t)min(row0 + r, a.n_rows - 1) * L + t];
    }
    half8_t wr_h[KB], wr_l[KB], wu_h[KB], wu_l[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        wr_h[kb] = buf_load_h8(rs_wg, vl16, (wave * KB + kb) * 2048);
        wr_l[kb] = buf_load_h8(rs_wg, vl16, (wave * KB + kb) * 2048 + 1024);
        wu_h[kb] = buf_load_h8(rs_wg, vl16, ((NW + wave) * KB + kb) * 2048);
        wu_l[kb] = buf_load_h8(rs_wg, vl16, ((NW + wave) * KB + kb) * 2048 + 1024);
        *reinterpret_cast<half8_t*>(lds_wc + kb * 2048 + vl16) = buf_load_h8(rs_wc, vl16, (wave * KB + kb) * 2048);
        *reinterpret_cast<half8_t*>(lds_wc + kb * 2048 + 1024 + vl16) = buf_load_h8(rs_wc, vl16, (wave * KB + kb) * 2048 + 1024);
    }
    __syncthreads();

    const int xcol4 = (a.xoff + col) * 4;
    auto load_x = [&](f32x16& dst, int t, int block) {       // table row of each of the lane's 16 rows -> accumulator (C-in)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            dst[r] = buf_load1(rs_x, s_ids[crow(r, half) * LDT + t] * xld4 + xcol4, block * NH * 4);
    };
    // output [slot_base + row, t, :] (every state): descriptor over rows [row0, row0 + rows_here) of it (<= 32 x L x out_ld x 4 bytes)
    const int rows_here = min(32, a.n_rows - row0), old4 = (int)a.out_ld * 4;
    // (the 64-bit product of the base address runs on the vector ALU: without the readfirstlanes the descriptor counts as
    // divergent and every store becomes a waterfall loop)
    const uint64_t ob = reinterpret_cast<uint64_t>(a.out + ((int64_t)a.slot_base + row0) * L * a.out_ld + a.out_off);
    float* const obase = reinterpret_cast<float*>(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) |
                                                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ob));
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        obase, 0, __builtin_amdgcn_readfirstlane((int)((((int64_t)rows_here * L - 1) * a.out_ld + NH) * 4)), 0x00020000);
    const int o_voff = 4 * half * L * old4 + col * 4;
    f32x16 acc_r, acc_u, acc_c, h_own;
    float ug[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) h_own[r] = 0.f;
    // Leading zero ids (front padding): the state after k steps on id 0 from h = 0 does not depend on the row, so a workgroup whose
    // rows ALL start with t0 zeros takes those t0 states from the handle's table (computed by this kernel on one all-zero row when
    // the handle was created: the kernels are batch-position invariant, so the table holds the very bits each row would compute)
    // and runs steps t0 .. L - 1 only.  SeqSlate's second sequence input - the items exposed on earlier pages, 9 / 18 / 27 ids behind
    // 55 / 46 / 37 zeros, re-encoded for all envs at every page end - is the case this is for.
    int t0 = 0;
    if (a.pad) {                                                          // (uniform)
        __shared__ int s_lead[32];
        __shared__ int s_t0;
        if (tid < 32) s_lead[tid] = L;
        if (tid == 0) s_t0 = L;
        __syncthreads();
        {
            const int r = tid >> 3, c = tid & 7, cl = (L + 7) >> 3;       // 8 threads per row, a chunk of the row's ids each
            int first = L;
            for (int t = min(L, (c + 1) * cl) - 1; t >= c * cl; --t)
                if (s_ids[r * LDT + t] != 0) first = t;
            if (first < L) atomicMin(&s_lead[r], first);
        }
        __syncthreads();
        if (tid < 32) {
            atomicMin(&s_t0, s_lead[tid]);                                // (rows past n_rows repeat the last valid row)
            if (a.lead_out && row0 + tid < a.n_rows) a.lead_out[a.slot_base + row0 + tid] = s_lead[tid];
        }
        __syncthreads();
        t0 = __builtin_amdgcn_readfirstlane(s_t0);
        if (t0 > 0) {
            const float hv = a.pad[(size_t)(t0 - 1) * NH + col];
            const _Float16 vh = (_Float16)hv, vl = (_Float16)(hv - (float)vh);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                h_own[r] = hv;
                hp_hi[crow(r, half) * LDP + col] = vh;
                hp_lo[crow(r, half) * LDP + col] = vl;
            }
#pragma unroll 1
            for (int t = 0; t < t0; ++t) {
                const unsigned v = __builtin_bit_cast(unsigned, a.pad[(size_t)t * NH + col]);
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(v, rs_o, o_voff, (((r & 3) + 8 * (r >> 2)) * L + t) * old4, 0);
            }
            __syncthreads();
        }
    }
    if (t0 < L) {
        load_x(acc_r, t0, 0);
        load_x(acc_u, t0, 1);
        load_x(acc_c, t0, 2);
    }
    const int aoff = li * LDP + half * 8;

#pragma unroll 1
    for (int t = t0; t < L; ++t) {
        // ---- slot R: acc_r += h Wr
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const half8_t ah = *reinterpret_cast<const half8_t*>(hp_hi + aoff + kb * 16);
            const half8_t al = *reinterpret_cast<const half8_t*>(hp_lo + aoff + kb * 16);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wr_h[kb], acc_r, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wr_l[kb], acc_r, 0, 0, 0);
            acc_r = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wr_h[kb], acc_r, 0, 0, 0);
        }
        // ---- slot U: acc_u += h Wu   ||   reset gate: item kb retires accumulator registers 2kb, 2kb + 1
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const half8_t ah = *reinterpret_cast<const half8_t*>(hp_hi + aoff + kb * 16);
            const half8_t al = *reinterpret_cast<const half8_t*>(hp_lo + aoff + kb * 16);
            acc_u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wu_h[kb], acc_u, 0, 0, 0);
            acc_u = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wu_l[kb], acc_u, 0, 0, 0);
            acc_u = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wu_h[kb], acc_u, 0, 0, 0);
#pragma unroll
            for (int r = 2 * kb; r < 2 * kb + 2; ++r) {
                const float v = gate_sigmoid(acc_r[r]) * h_own[r];
                const _Float16 vh = (_Float16)v;
                rp_hi[crow(r, half) * LDP + col] = vh;
                rp_lo[crow(r, half) * LDP + col] = (_Float16)(v - (float)vh);
            }
        }
        if (t + 1 < L) load_x(acc_r, t + 1, 0);              // next step's r projection into the retired accumulators
        __syncthreads();
        // ---- slot C: acc_c += (r*h) Wc (LDS, one item ahead)   ||   update gate
        half8_t wc_h[2], wc_l[2];
        wc_h[0] = *reinterpret_cast<const half8_t*>(lds_wc + vl16);
        wc_l[0] = *reinterpret_cast<const half8_t*>(lds_wc + 1024 + vl16);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const int cb = kb & 1, nb = cb ^ 1;
            if (kb + 1 < KB) {
                wc_h[nb] = *reinterpret_cast<const half8_t*>(lds_wc + (kb + 1) * 2048 + vl16);
                wc_l[nb] = *reinterpret_cast<const half8_t*>(lds_wc + (kb + 1) * 2048 + 1024 + vl16);
            }
            const half8_t ah = *reinterpret_cast<const half8_t*>(rp_hi + aoff + kb * 16);
            const half8_t al = *reinterpret_cast<const half8_t*>(rp_lo + aoff + kb * 16);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wc_h[cb], acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, wc_l[cb], acc_c, 0, 0, 0);
            acc_c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, wc_h[cb], acc_c, 0, 0, 0);
            ug[2 * kb] = gate_sigmoid(acc_u[2 * kb]);
            ug[2 * kb + 1] = gate_sigmoid(acc_u[2 * kb + 1]);
        }
        if (t + 1 < L) load_x(acc_u, t + 1, 1);
        // ---- exposed: candidate, blend, new state planes, h1 cache row
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float c = gate_tanh(acc_c[r]);
            float hn = __builtin_fmaf(ug[r], h_own[r] - c, c);           // u h + (1-u) c
            // the planes are a function of the ROUNDED fp32 state: left alone, the compiler fuses the fma with the conversion
            // (v_fma_mixlo_f16: one rounding of the exact result to fp16), and the hi plane then differs from (_Float16)hn in the
            // rare double-rounding cases - which made a state taken from the pad table (split from its fp32 value above) differ
            // from the same state computed here, one fp32 ulp a step later, for one element in ~8000
            asm volatile("" : "+v"(hn));
            h_own[r] = hn;
            const _Float16 vh = (_Float16)hn;
            hp_hi[crow(r, half) * LDP + col] = vh;
            hp_lo[crow(r, half) * LDP + col] = (_Float16)(hn - (float)vh);
            // h1 cache row through the descriptor over this workgroup's valid rows: the row / step part of the address is a scalar
            // offset, rows beyond n_rows fall outside the descriptor and are dropped by the hardware - one store per element
            // instead of a compare, an exec mask and a 64-bit address (16 x ~20 instructions per step in the exposed stretch)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hn), rs_o, o_voff, (((r & 3) + 8 * (r >> 2)) * L + t) * old4, 0);
        }
        if (t + 1 < L) load_x(acc_c, t + 1, 2);
        __syncthreads();
    }
}

// -------------------------------------------------------------------------------------------------
// DIN attention scores (deepctr LocalActivationUnit, att_hidden_units=(64,16), sigmoid, raw scores).
//   hid1^T [64 units x L steps] = W1d^T (q*h1_t) + qa + AK_t    as 2x2 32x32 MFMA tiles per row
//   hid2 = sigmoid(hid1 W2 + b2), score = hid2 w3 + b3           in registers (+ one half-wave swap)
// grid.y = sequence input.  STAGE = true: a workgroup owns one group of rows that share a cache slot (the 9
// reward rows of an env), stages the slot's h1 tile [L,E] in LDS once and its waves take the rows round-robin.
// STAGE = false (group == 1, obs rows): one row per wave, 4 rows per workgroup, the (q*h1_t) operand is read
// straight from the cache (every byte is used exactly once, LDS staging would only cost occupancy).
// Weight / operand fragments run through a 2-deep register ring like k_recur.
struct DinArgs {
    int R, L, E, group, n_groups;
    const int32_t* slots; int64_t slots_stride;     // [n_seq][n_groups]
    const float* h1[4];          // [slot, L, E]
    const float* h1f[4];         // k_din_x: the same states in fragment order (k_h1_frag)
    const float* proj[4]; int64_t pld;   // [slot*L, pld], AK at column 0
    const float* q;              // [R, E]
    const float* w1ac[4];        // [E, 64]   (W1a + W1c)
    const float* qa; int64_t qa_stride; int qa_ld;   // q @ (W1a + W1c) of input s, row r, unit j at qa[s*qa_stride + r*qa_ld + j] (one GEMM over all inputs)
    const float* w1d[4];         // packed [2][E/8][64][4]
    const float* w1d16[4];       // H16: fp16 hi/lo planes [2][E/16][2][64][8 halfs] (pack_frag_h16)
    const float* w2[4]; const float* b2[4]; const float* w3[4]; const float* b3[4];
    float* scores; int64_t scores_stride;           // [n_seq][scores_stride] rows of L
    const int32_t* order;        // processing order of the row groups (NULL = identity): rl4rs_dien_set_row_order
    unsigned long long* trace;   // -DRL4RS_DINX_TRACE timing experiments only
    // k_din_x: leading zero ids of the sequence in every cache slot and the slot of the all-zero sequence (RecurArgs::lead / pad_slot):
    // the steps of a row's front padding read the pad slot's states and projections - the same bytes, shared by all rows.  NULL = off
    const int32_t* lead[4]; int pad_slot;
};

template <bool STAGE, bool H16>
__global__ __launch_bounds__(256) void k_din_scores(DinArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int L = a.L, E = a.E, LDK = E + 4;
    const int sq = blockIdx.y;
    // layer 2 runs on the matrix pipe: hid2^T [16 x steps] = W2^T [16 x 64 units] hid1^T, two hidden units per
    // v_mfma_f32_32x32x2_f32.  The layer-1 accumulator layout IS a B operand of that MFMA (lane = step, register r of
    // half-wave h = unit crow(r, h): the two half-waves supply k = 0 and k = 1), so only W2 needs arranging:
    // s_w2pk[(m*16 + r)*64 + lane] = W2[m*32 + crow(r, half)][li] for li < 16, else 0 (the A operand, output unit = li).
    float* s_w2 = reinterpret_cast<float*>(smem);        // [2*16][64]
    float* s_misc = s_w2 + 2 * 16 * 64;                  // b2[16] w3[16] b3[1] pad -> 48
    float* s_wave = s_misc + 48;                         // per wave: q[E] + qa[64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, nw = blockDim.x >> 6;
    float* s_h1 = s_wave + (size_t)nw * (E + ATT_H1);    // STAGE: [L][LDK]
    const int half = lane >> 5, li = lane & 31;
    if (H16) {
        // fp16x2 layer 2 on v_mfma_f32_32x32x16_f16: A fragment (m, kb2) of lane (o, kg) = W2[m*32 + crow(kb2*8 + i, kg)][o], i = 0..7
        // (the 8 layer-1 accumulator registers kb2*8.. of a half-wave ARE the 8 k values of its B fragment); hi / lo planes,
        // [m*2 + kb2][plane][lane][8 halfs] = 8 KB, the same bytes as the fp32 table
        _Float16* s_w2h = reinterpret_cast<_Float16*>(smem);
        for (int i = tid; i < 4 * 64 * 8; i += blockDim.x) {
            const int e = i & 7, ln = (i >> 3) & 63, f = i >> 9, o = ln & 31;
            const float v = o < ATT_H2 ? a.w2[sq][((f >> 1) * 32 + crow((f & 1) * 8 + e, ln >> 5)) * ATT_H2 + o] : 0.f;
            const _Float16 hi = (_Float16)v;
            s_w2h[(f * 2) * 512 + ln * 8 + e] = hi;
            s_w2h[(f * 2 + 1) * 512 + ln * 8 + e] = (_Float16)(v - (float)hi);
        }
    } else
    for (int i = tid; i < 2 * 16 * 64; i += blockDim.x) {
        const int ln = i & 63, mr = i >> 6, o = ln & 31;
        s_w2[i] = o < ATT_H2 ? a.w2[sq][((mr >> 4) * 32 + crow(mr & 15, ln >> 5)) * ATT_H2 + o] : 0.f;
    }
    if (tid < ATT_H2) { s_misc[tid] = a.b2[sq][tid]; s_misc[16 + tid] = a.w3[sq][tid]; }
    if (tid == 0) s_misc[32] = a.b3[sq][0];
    int g0, slot0 = 0;
    if (STAGE) {
        g0 = a.order ? a.order[blockIdx.x] : blockIdx.x;
        slot0 = a.slots[(size_t)sq * a.slots_stride + g0];
        const float* h1g = a.h1[sq] + (size_t)slot0 * L * E;
        for (int i = tid; i < L * (E / 4); i += blockDim.x) {
            int t = i / (E / 4), k4 = i - t * (E / 4);
            *reinterpret_cast<float4*>(s_h1 + t * LDK + k4 * 4) = *reinterpret_cast<const float4*>(h1g + (size_t)t * E + k4 * 4);
        }
    } else {
        g0 = blockIdx.x * nw;
    }
    __syncthreads();
    float* s_q = s_wave + (size_t)wave * (E + ATT_H1);
    float* s_qa = s_q + E;
    const int KB = E / 8;
    const __amdgpu_buffer_rsrc_t rs_w1d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1d[sq]), 0, 2 * KB * 1024, 0x00020000);
    const int vl16 = lane * 16;
    const float* __restrict__ w1ac = a.w1ac[sq];
    const int ntile = (L + 31) / 32;    // L <= 64 -> 1 or 2 N tiles
    const int t0 = min(li, L - 1), t1 = min(32 + li, L - 1);     // clamped: steps >= L re-read the last row (results never stored)
    const int nrows = STAGE ? a.group : 1;
    const int jstep = STAGE ? nw : 1;

    for (int j = STAGE ? wave : 0; j < nrows; j += jstep) {
        if (!STAGE && g0 + wave >= a.R) break;
        const int row = STAGE ? g0 * a.group + j : (a.order ? a.order[g0 + wave] : g0 + wave);
        if (row >= a.R) break;
        const int slot = STAGE ? slot0 : a.slots[(size_t)sq * a.slots_stride + row];
        for (int k = lane; k < E; k += 64) s_q[k] = a.q[(size_t)row * E + k];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (a.qa) {   // qa = q @ (W1a + W1c) comes from a GEMM over all rows (a 128-step serial loop per wave otherwise)
            s_qa[lane] = a.qa[(size_t)sq * a.qa_stride + (size_t)row * a.qa_ld + lane];
        } else {      // lane = hidden unit
            float s = 0.f;
#pragma unroll 8
            for (int k = 0; k < E; ++k) s = fmaf(s_q[k], w1ac[k * ATT_H1 + lane], s);
            s_qa[lane] = s;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // B-operand source rows for this lane's two steps
        const float* hsrc0 = STAGE ? s_h1 + t0 * LDK + half * 4 : a.h1[sq] + ((size_t)slot * L + t0) * E + half * 4;
        const float* hsrc1 = STAGE ? s_h1 + t1 * LDK + half * 4 : a.h1[sq] + ((size_t)slot * L + t1) * E + half * 4;
        const float* qsrc = s_q + half * 4;
        f32x16 acc00, acc10, acc01, acc11;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc10[r] = 0.f; acc01[r] = 0.f; acc11[r] = 0.f; }
        if (H16) {
            // fp16x2 layer 1 (scorer_mode FP16X2): same hi/lo operand split as k_augru_h16, 3 v_mfma_f32_32x32x16_f16 per
            // product.  Lane (li, half) supplies k = kb*16 + half*8 + 0..7 of its step's (q * h1_t) row; q*h1 is bounded by
            // the embedding table (range-checked at create), so the lo parts need no scaling.
            const __amdgpu_buffer_rsrc_t rs16 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1d16[sq]), 0, 2 * (E / 16) * 2048, 0x00020000);
            const float* h0 = hsrc0 - half * 4 + half * 8;      // rows t0 / t1 at k = half*8
            const float* h1p = hsrc1 - half * 4 + half * 8;
            const float* qp = s_q + half * 8;
            constexpr int KB16 = 8;
            half8_t wa[2][2][2];            // [ring][m tile][plane]
            float4 hr[2][2][2];             // [ring][step tile][lo/hi half of the 8 k]
            auto ld = [&](int slot, int kb) {
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    wa[slot][m][0] = buf_load_h8(rs16, vl16, (m * KB16 + kb) * 2048);
                    wa[slot][m][1] = buf_load_h8(rs16, vl16, (m * KB16 + kb) * 2048 + 1024);
                }
                hr[slot][0][0] = *reinterpret_cast<const float4*>(h0 + kb * 16);
                hr[slot][0][1] = *reinterpret_cast<const float4*>(h0 + kb * 16 + 4);
                hr[slot][1][0] = *reinterpret_cast<const float4*>(h1p + kb * 16);
                hr[slot][1][1] = *reinterpret_cast<const float4*>(h1p + kb * 16 + 4);
            };
            ld(0, 0);
#pragma unroll
            for (int kb = 0; kb < KB16; ++kb) {
                const int cb = kb & 1, nb = cb ^ 1;
                if (kb + 1 < KB16) ld(nb, kb + 1);
                const float4 qa4 = *reinterpret_cast<const float4*>(qp + kb * 16);
                const float4 qb4 = *reinterpret_cast<const float4*>(qp + kb * 16 + 4);
                __builtin_amdgcn_sched_barrier(0);
                half8_t bh[2], bl[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const float pr[8] = {hr[cb][n][0].x * qa4.x, hr[cb][n][0].y * qa4.y, hr[cb][n][0].z * qa4.z, hr[cb][n][0].w * qa4.w,
                                         hr[cb][n][1].x * qb4.x, hr[cb][n][1].y * qb4.y, hr[cb][n][1].z * qb4.z, hr[cb][n][1].w * qb4.w};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        half2_t h2, l2;
                        split_h16_pair(pr[e], pr[e + 1], h2, l2);
                        bh[n][e] = h2[0]; bh[n][e + 1] = h2[1];
                        bl[n][e] = l2[0]; bl[n][e + 1] = l2[1];
                    }
                }
#define RL4RS_DIN3(acc, m, n)                                                                         \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cb][m][0], bh[n], acc, 0, 0, 0);      \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cb][m][1], bh[n], acc, 0, 0, 0);      \
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wa[cb][m][0], bl[n], acc, 0, 0, 0);
                RL4RS_DIN3(acc00, 0, 0) RL4RS_DIN3(acc10, 1, 0) RL4RS_DIN3(acc01, 0, 1) RL4RS_DIN3(acc11, 1, 1)
#undef RL4RS_DIN3
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
        // fully unrolled 2-deep ring over the E/8 k-blocks: weights through a buffer descriptor, (q*h1_t) operand
        // from LDS (STAGE) or straight from the cache (immediate offsets off two per-lane row pointers)
        float4 aw0[2], aw1[2], hv0[2], hv1[2], qv[2];
        aw0[0] = buf_load4(rs_w1d, vl16, 0);
        aw1[0] = buf_load4(rs_w1d, vl16, KB * 1024);
        hv0[0] = *reinterpret_cast<const float4*>(hsrc0);
        hv1[0] = *reinterpret_cast<const float4*>(hsrc1);
        qv[0] = *reinterpret_cast<const float4*>(qsrc);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const int cb = kb & 1, nb = cb ^ 1;
            if (kb + 1 < 16) {
                aw0[nb] = buf_load4(rs_w1d, vl16, (kb + 1) * 1024);
                aw1[nb] = buf_load4(rs_w1d, vl16, (KB + kb + 1) * 1024);
                hv0[nb] = *reinterpret_cast<const float4*>(hsrc0 + (kb + 1) * 8);
                hv1[nb] = *reinterpret_cast<const float4*>(hsrc1 + (kb + 1) * 8);
                qv[nb] = *reinterpret_cast<const float4*>(qsrc + (kb + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const float b0[4] = {hv0[cb].x * qv[cb].x, hv0[cb].y * qv[cb].y, hv0[cb].z * qv[cb].z, hv0[cb].w * qv[cb].w};
                const float b1[4] = {hv1[cb].x * qv[cb].x, hv1[cb].y * qv[cb].y, hv1[cb].z * qv[cb].z, hv1[cb].w * qv[cb].w};
                const float a0[4] = {aw0[cb].x, aw0[cb].y, aw0[cb].z, aw0[cb].w};
                const float a1[4] = {aw1[cb].x, aw1[cb].y, aw1[cb].z, aw1[cb].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    // both 32-step tiles are always computed (for maxlen <= 32 the second one re-reads clamped rows and
                    // its results are never stored): no branch inside the MFMA stream
                    acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b0[i], acc00, 0, 0, 0);
                    acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b0[i], acc10, 0, 0, 0);
                    acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i], b1[i], acc01, 0, 0, 0);
                    acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i], b1[i], acc11, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        // epilogue: lane holds, for step t (= N index), 16 of the 32 hidden units of each M tile
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            if (n >= ntile) break;
            const int t = n * 32 + li;
            const int tc = min(t, L - 1);
            const float* akp = a.proj[sq] + ((size_t)slot * L + tc) * a.pld;
            f32x16 acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                float hvv[16];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int jr = m * 32 + 8 * r4 + 4 * half;       // hidden units jr..jr+3
                    float4 ak = *reinterpret_cast<const float4*>(akp + jr);
                    float akv[4] = {ak.x, ak.y, ak.z, ak.w};
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const float accv = (n == 0) ? (m == 0 ? acc00[r4 * 4 + rr] : acc10[r4 * 4 + rr])
                                                    : (m == 0 ? acc01[r4 * 4 + rr] : acc11[r4 * 4 + rr]);
                        const float hv = gate_sigmoid(accv + s_qa[jr + rr] + akv[rr]);
                        if (H16) hvv[r4 * 4 + rr] = hv;
                        else acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(s_w2[(m * 16 + r4 * 4 + rr) * 64 + lane], hv, acc2, 0, 0, 0);
                    }
                }
                if (H16) {      // hidden activations in (0, 1): fp16 hi + lo, 3 MFMAs per product like layer 1
                    const char* s_w2b = smem;
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2) {
                        half8_t bh2, bl2;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) {
                            half2_t h2, l2;
                            split_h16_pair(hvv[kb2 * 8 + e], hvv[kb2 * 8 + e + 1], h2, l2);
                            bh2[e] = h2[0]; bh2[e + 1] = h2[1];
                            bl2[e] = l2[0]; bl2[e + 1] = l2[1];
                        }
                        const half8_t ah2 = *reinterpret_cast<const half8_t*>(s_w2b + ((m * 2 + kb2) * 2) * 1024 + lane * 16);
                        const half8_t al2 = *reinterpret_cast<const half8_t*>(s_w2b + ((m * 2 + kb2) * 2 + 1) * 1024 + lane * 16);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah2, bh2, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al2, bh2, acc2, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah2, bl2, acc2, 0, 0, 0);
                    }
                }
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
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace rl4rs
#ifndef RL4RS_DINX_RING
#define RL4RS_DINX_RING 2       // cache rows requested this many k-blocks ahead
#endif
#ifndef RL4RS_DINX_WPE
#define RL4RS_DINX_WPE 4        // waves per SIMD the register allocation aims at
#endif
#include "din_x.hpp"
namespace rl4rs {

// softmax(obs @ out_w + out_b)[:, 1]  (dien.py:36, slate.py:298).  One wave per row.
__global__ __launch_bounds__(256) void k_head_prob(const float* __restrict__ obs, int R, int D, int K,
                                                   const float* __restrict__ w, const float* __restrict__ b,
                                                   float* __restrict__ prob) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    float logit[8];
    for (int c = 0; c < K; ++c) {
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s = fmaf(obs[(size_t)row * D + k], w[(size_t)k * K + c], s);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        logit[c] = s + b[c];
    }
    float m = logit[0];
    for (int c = 1; c < K; ++c) m = fmaxf(m, logit[c]);
    float sum = 0.f;
    for (int c = 0; c < K; ++c) sum += expf(logit[c] - m);
    if (lane == 0) prob[row] = expf(logit[K > 1 ? 1 : 0] - m) / sum;
}

// 'simulator_obs' epilogue for the table form of the head (dien.py:35):
//   obs[row] = ELU( pre[row] + b + sum_j P_j[cat[row, j]] ),   P_j = cat_emb @ W_obs[flatten slot j]  ([H, 256], built at load)
// i.e. the Flatten(category_emb) slice of the concat (Cn*E of the 3456 inputs) never goes through the GEMM: its
// contribution is Cn row gathers of 1 KB.  One wave per row, each lane owns 4 consecutive outputs (float4).
__global__ __launch_bounds__(256) void k_head_finish(float* __restrict__ obs, int R, const int32_t* __restrict__ cat, int Cn,
                                                     int H, const float* __restrict__ ptab, const float* __restrict__ bias) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= R) return;
    float4* o = reinterpret_cast<float4*>(obs + (size_t)row * OBS_DIM) + lane;
    float4 acc = *o;
    const float4 b = reinterpret_cast<const float4*>(bias)[lane];
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
    const int32_t* crowp = cat + (size_t)row * Cn;
    for (int j = 0; j < Cn; ++j) {
        int id = min(max(crowp[j], 0), H - 1);
        const float4 v = reinterpret_cast<const float4*>(ptab + ((size_t)j * H + id) * OBS_DIM)[lane];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    acc.x = eluf_(acc.x); acc.y = eluf_(acc.y); acc.z = eluf_(acc.z); acc.w = eluf_(acc.w);
    *o = acc;
}

}  // namespace rl4rs

// =================================================================================================
using namespace rl4rs;

namespace {
enum { KID_CAT = 0, KID_DENSE, KID_DIN, KID_AUGRU, KID_HEAD, KID_PROB, KID_GRU1, KID_PROJ, KID_COUNT };
const char* kKernelNames[KID_COUNT] = {"k_cat_attn", "k_gemm_f32(dense tower)", "k_din_scores",
                                       "augru", "k_gemm_f32(simulator_obs)", "k_head_prob",
                                       "k_recur<128,gru>", "k_gemm_f32(seq projections)"};
struct EvPair { int id; hipEvent_t a, b; };
}  // namespace

static size_t augru_h16_smem(int mt, int nh2, int L) {
    return (size_t)4 * mt * 32 * (nh2 + 8) * 2 + (size_t)(mt * 32 * (L + 1) + mt * 32) * 4 + (size_t)8 * RL4RS_H16_NLDS * 2048;
}

struct rl4rs_dien {
    rl4rs_dien_cfg c;
    int n_cu;
    int E, U, L, S, Cn, Dn, H, K, F, PLD, NH2;
    int Fld;               // row stride of allf: F, or only the Kh columns that exist in the table form of the head
    // weights (device)
    float *cat_emb, *seq_emb, *dense_w1, *dense_b1, *dense_w2, *dense_b2, *obs_w, *obs_b, *out_w, *out_b;
    float* ptab;           // [Cn, H, 256] head tables of the flattened category embeddings (NULL = GEMM form)
    float* embw1[4];       // [H, 3E]   first-GRU input projection table (bias folded)
    float* gru_wg[4];      // packed h-side gate weights     [2*E/32][E/8][64][4]
    float* gru_wc[4];      // packed h-side candidate weights
    float* wproj[4];       // [E, PLD]  = [W1b - W1c | augru gate x-side | augru cand x-side]
    float* bproj[4];       // [PLD]
    float* w1ac[4];        // [E, 64]
    float* w1ac_all;       // [E, S*64]: the q-side matrices of all sequence inputs side by side, packed (one GEMM per forward)
    float* qa;             // [S, max_rows, 64]
    float* w1d[4];         // packed [2][E/8][64][4]
    float* w1d16[4];       // fp16 hi/lo planes of the same fragments (fp16x2 mode)
    float *att_w2[4], *att_b2[4], *att_w3[4], *att_b3[4];
    float* augru_wg16[4];  // fp16 hi/lo planes of the same fragments (optional fp16x2 mode)
    float* augru_wc16[4];
    float* gru_wg16[4];    // first GRU, fp16 hi/lo planes (k_gru_h16)
    float* gru_pad[4];     // k_gru_h16: [L][E] states after 1 .. L leading zero ids (RecurArgs::pad), NULL = off (RL4RS_DIEN_OPT_NO_GRU_PAD)
    int32_t* lead[4];      // [max_slots + 1] leading zero ids of the sequence in every cache slot (k_gru_h16 writes, k_augru_x<.., PAD> reads); slot max_slots = the all-zero sequence
    float* gru_wc16[4];
    bool gru16, gru16_attr;
    bool fp16x2;
    bool augru_x;          // fp16x2 mode: k_augru_x (default) or the first-generation k_augru_h16 (RL4RS_DIEN_OPT_AUGRU_H16)
    int augru_rows;        // k_augru_x row-tile form: 0 automatic, 32, 64 (rl4rs_dien_set_augru_rows)
    float augru_s[4][3][8];   // fp16x2: power-of-two prescale of the AUGRU's reset / update / candidate column tiles (1 in fp32 mode)
    bool din_x;            // fp16x2 DIN scores through k_din_x (RL4RS_DIN=v1 keeps k_din_scores<*, true>)
    bool dense_chain;      // fp16x2 mode: both dense-tower layers in one launch (RL4RS_DENSE_FUSED=0 at create: two GEMMs)
    float* tsum;           // [max_rows, 256]: obs_b + the per-slot head tables' rows, built by k_cat_attn (table form, Cn <= 24)
    bool dense_fork;       // RL4RS_DIEN_OPT_DENSE_FORK: the dense tower (depends on nothing before the head) on side_stream, beside the category / DIN / AUGRU launches
    hipStream_t side_stream; hipEvent_t ev_fork, ev_join;
    bool cat_group;        // reward-sized launches (rows in groups of 8 / 9): k_cat_attn2g, one workgroup per group (RL4RS_DIEN_OPT_NO_CAT_GROUP: per row)
    bool cat_v2;           // category branch through k_cat_attn2 (half-K LDS image) when the shape allows (RL4RS_DIEN_OPT_CAT_V1: first form)
    bool cat16;            // fp16x2 mode: the Gram matrix of k_cat_attn in the split form (cat_emb inside the fp16 range)
    bool gemm16;           // fp16x2 mode: the plain GEMMs (dense tower, q-side DIN term, cache projections, head) through k_gemm_h16
    bool din16;            // fp16x2 mode: the DIN layer-1 operands (q*h1 bounded by the embedding table, W1d) fit fp16 too
    int* range_flag;       // device int: a k_augru_h16 state left the fp16 range (sticky until read)
    const int32_t* row_order; int row_order_n;    // processing order of the row groups of a forward (caller-owned), or NULL
    float* augru_wg[4];    // packed [2*NH2/32][NH2/8][64][4]
    float* augru_wc[4];
    // caches
    float* h1[4];          // [max_slots, L, E]
    float* h1f[4];         // fragment-order copy for k_din_x [max_slots, ceil(L/32), 8, 64, 8] (fp16x2 mode, E = 128)
    float* proj[4];        // [max_slots*L, PLD]
    // scratch
    float *allf, *dh, *q, *scores, *obs_tmp;
    float* obs_mirror;     // rl4rs_dien_set_obs_mirror: device-visible host memory [R, OBS_DIM] for the NEXT forward's observation
    bool obs_mirror_used;  // ... whether that forward's head kernel took it (only the fp16x2 GEMM with the table addend writes mirrors)
    std::vector<void*> owned;
    // profiling
    int profiling;         // 0 off, 1 every kernel class, 2 only the AUGRU recurrence (two event records per forward)
    std::vector<EvPair> pending;
    std::vector<hipEvent_t> pool;
    double ms_total[KID_COUNT];
    long launches[KID_COUNT];
};

namespace {

int upload(rl4rs_dien* n, float** dst, const float* src, size_t count, hipStream_t st) {
    int rc = dev_alloc(dst, count);
    if (rc) return rc;
    n->owned.push_back(*dst);
    RL4RS_HIP_TRY(hipMemcpyAsync(*dst, src, count * 4, hipMemcpyHostToDevice, st));
    return RL4RS_OK;
}
int alloc_f(rl4rs_dien* n, float** dst, size_t count) {
    int rc = dev_alloc(dst, count);
    if (rc) return rc;
    n->owned.push_back(*dst);
    return RL4RS_OK;
}

// pack Wh [K, N] (row-major, leading dim ld, starting at row k_off) into MFMA-B fragment order
std::vector<float> pack_frag(const float* w, int ld, int k_off, int K, int N) {
    const int KB = K / 8, NT = N / 32;
    std::vector<float> out((size_t)K * N);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    int k = kb * 8 + (lane >> 5) * 4 + i;
                    int j = nt * 32 + (lane & 31);
                    out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + i] = w[(size_t)(k_off + k) * ld + j];
                }
    return out;
}

// pack Wh [K, N] (rows k_off..) into fp16 hi/lo B-fragment planes for v_mfma_f32_32x32x16_f16:
// [ntile][kb16][plane][lane][8]: element i of lane (j = lane&31, kg = lane>>5) is W[kb16*16 + kg*8 + i][nt*32 + j]
std::vector<float> pack_frag_h16(const float* w, int ld, int k_off, int K, int N) {
    const int KB = K / 16, NT = N / 32;
    std::vector<uint16_t> out((size_t)NT * KB * 2 * 64 * 8);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const float v = w[(size_t)(k_off + kb * 16 + (lane >> 5) * 8 + i) * ld + nt * 32 + (lane & 31)];
                    const uint16_t hi = f32_to_f16(v);
                    const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                    const size_t base = (((size_t)nt * KB + kb) * 2) * 512 + (size_t)lane * 8 + i;
                    out[base] = hi;
                    out[base + 512] = lo;
                }
    std::vector<float> f(out.size() / 2);
    memcpy(f.data(), out.data(), out.size() * 2);
    return f;
}

static int scorer_gemm(rl4rs_dien* n, const float* a, int64_t lda, const float* wp, const float* bias, float* c, int64_t ldc,
                       int M, int N, int K, int act, hipStream_t st) {
    return n->gemm16 ? launch_gemm_h16(a, lda, wp, bias, c, ldc, M, N, K, act, st)
                     : launch_gemm_packed(a, lda, wp, bias, c, ldc, M, N, K, act, st);
}

struct Prof {
    rl4rs_dien* n; int id; hipStream_t st; hipEvent_t a, b; bool on;
    Prof(rl4rs_dien* n_, int id_, hipStream_t st_) : n(n_), id(id_), st(st_), on(n_->profiling == 1 || (n_->profiling == 2 && id_ == KID_AUGRU)) {
        if (!on) return;
        auto get = [&]() {
            hipEvent_t e;
            if (!n->pool.empty()) { e = n->pool.back(); n->pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) { on = false; e = nullptr; }
            return e;
        };
        a = get(); b = get();
        if (on) (void)hipEventRecord(a, st);
    }
    ~Prof() {
        if (!on) return;
        (void)hipEventRecord(b, st);
        n->pending.push_back({id, a, b});
    }
};

}  // namespace

extern "C" {

int rl4rs_dien_create(const rl4rs_dien_cfg* c, const rl4rs_dien_weights* w, void* stream, rl4rs_dien** out) {
    RL4RS_REQUIRE(c && w && out, "dien_create: null argument");
    RL4RS_REQUIRE(c->emb_size == 128 && c->hidden_units > 0 && c->hidden_units % 32 == 0,
                  "dien: emb_size must be 128 (got %d), hidden_units a multiple of 32", c->emb_size);
    RL4RS_REQUIRE(c->maxlen >= 1 && c->maxlen <= 64, "dien: maxlen must be in 1..64 (got %d)", c->maxlen);
    RL4RS_REQUIRE(c->seq_num >= 1 && c->seq_num <= 4, "dien: seq_num must be in 1..4");
    RL4RS_REQUIRE(c->class_num >= 1 && c->class_num <= 8, "dien: class_num must be in 1..8");
    RL4RS_REQUIRE(c->category_feature_num >= 1 && c->category_feature_num <= 32, "dien: category_feature_num must be in 1..32");
    RL4RS_REQUIRE(c->max_rows > 0 && c->max_slots > 0 && c->category_hash_size > 0 && c->dense_feature_num > 0,
                  "dien: bad sizes");
    RL4RS_REQUIRE(((int64_t)c->max_slots + 1) * c->maxlen * (ATT_H1 + 6 * c->emb_size) * 4 < (int64_t)0x7fffffff * 2,      // (+ 1: the pad slot)
                  "dien: max_slots=%d too large: the per-input sequence cache must stay below 4 GB (32-bit buffer offsets)",
                  c->max_slots);
    RL4RS_REQUIRE((int64_t)c->category_hash_size * 3 * c->emb_size * 4 < (int64_t)0x7fffffff * 2,
                  "dien: category_hash_size too large for 32-bit buffer offsets");
    bool want_fp16x2 = false;
    {   // arithmetic of the AUGRU recurrence (include/rl4rs_hip.h RL4RS_SCORER_*)
        RL4RS_REQUIRE(c->scorer_mode >= RL4RS_SCORER_AUTO && c->scorer_mode <= RL4RS_SCORER_FP16X2,
                      "dien: scorer_mode must be 0 (auto), 1 (fp32) or 2 (fp16x2), got %d", c->scorer_mode);
        const size_t e = c->emb_size, nh2 = 2 * e;
        int mode = c->scorer_mode;
        float wmax = 0.f;
        bool finite = true;
        for (int s = 0; s < c->seq_num; ++s) {
            RL4RS_REQUIRE(w->augru_gate_w[s] && w->augru_cand_w[s], "dien_create: null AUGRU weights for sequence input %d", s);
            for (size_t i = 0; i < (e + nh2) * 2 * nh2; ++i) {
                float v = fabsf(w->augru_gate_w[s][i]);
                finite = finite && v < 3.0e38f; wmax = fmaxf(wmax, v);
            }
            for (size_t i = 0; i < (e + nh2) * nh2; ++i) {
                float v = fabsf(w->augru_cand_w[s][i]);
                finite = finite && v < 3.0e38f; wmax = fmaxf(wmax, v);
            }
        }
        // every fp16x2 matrix is stored multiplied by its own power of two (pow2_prescale below), so the split form has no
        // weight-range condition left: any FINITE checkpoint runs in fp16x2
        const bool fits = finite;
        if (mode == RL4RS_SCORER_AUTO) mode = fits ? RL4RS_SCORER_FP16X2 : RL4RS_SCORER_FP32;
        RL4RS_REQUIRE(mode != RL4RS_SCORER_FP16X2 || fits, "dien: the fp16x2 scorer needs finite AUGRU weights (max |w| = %g)", (double)wmax);
        want_fp16x2 = mode == RL4RS_SCORER_FP16X2;
    }
    int ndev = rl4rs_device_count();
    if (ndev <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    rl4rs_dien* n = new rl4rs_dien();
    n->c = *c;
    const int E = c->emb_size, U = c->hidden_units, L = c->maxlen, S = c->seq_num, Cn = c->category_feature_num;
    const int Dn = c->dense_feature_num, H = c->category_hash_size, K = c->class_num;
    const int NH2 = 2 * E, PLD = ATT_H1 + 3 * NH2;
    const int F = S * NH2 + U + (Cn + 1) * E;
    n->E = E; n->U = U; n->L = L; n->S = S; n->Cn = Cn; n->Dn = Dn; n->H = H; n->K = K; n->F = F;
    n->PLD = PLD; n->NH2 = NH2;
    n->profiling = 0;
    n->fp16x2 = want_fp16x2;
    n->row_order = nullptr;
    n->row_order_n = 0;
    // kernel-path selection: fields of the configuration (RL4RS_DIEN_OPT_*), never the process environment - a test or an A/B
    // run holds handles with different paths side by side
    const uint32_t opts = c->kernel_opts;
    RL4RS_REQUIRE((opts & ~(uint32_t)RL4RS_DIEN_OPT_ALL) == 0, "dien: unknown kernel_opts bits 0x%x", opts & ~(uint32_t)RL4RS_DIEN_OPT_ALL);
    RL4RS_REQUIRE(!((opts & RL4RS_DIEN_OPT_AUGRU_ROWS32) && (opts & RL4RS_DIEN_OPT_AUGRU_ROWS64)),
                  "dien: kernel_opts asks for both the 32-row and the 64-row AUGRU form");
    n->augru_x = !(opts & RL4RS_DIEN_OPT_AUGRU_H16);
    n->augru_rows = (opts & RL4RS_DIEN_OPT_AUGRU_ROWS32) ? 32 : ((opts & RL4RS_DIEN_OPT_AUGRU_ROWS64) ? 64 : 0);
    n->din16 = false;
    n->din_x = !(opts & RL4RS_DIEN_OPT_DIN_V1);
    n->gemm16 = false;
    n->cat16 = false;
    n->cat_v2 = !(opts & RL4RS_DIEN_OPT_CAT_V1);
    n->cat_group = !(opts & RL4RS_DIEN_OPT_NO_CAT_GROUP);
    n->dense_fork = (opts & RL4RS_DIEN_OPT_DENSE_FORK) != 0;
    n->side_stream = nullptr; n->ev_fork = nullptr; n->ev_join = nullptr;
    n->dense_chain = !(opts & RL4RS_DIEN_OPT_NO_DENSE_CHAIN);
    n->gru16 = false;
    n->gru16_attr = false;
    if (want_fp16x2) {      // the DIN layer-1 split needs |q * h1| <= max |seq_emb| and the q*k rows of att_w1 inside fp16 range
        float mx = 0.f;
        bool fin = true;
        for (size_t i = 0; i < (size_t)c->category_hash_size * c->emb_size; ++i) {
            const float v = fabsf(w->seq_emb[i]);
            fin = fin && v == v; mx = fmaxf(mx, v);
        }
        for (int s = 0; s < c->seq_num && w->att_w1[s]; ++s)
            for (size_t i = (size_t)3 * c->emb_size * ATT_H1; i < (size_t)4 * c->emb_size * ATT_H1; ++i) {
                const float v = fabsf(w->att_w1[s][i]);
                fin = fin && v == v; mx = fmaxf(mx, v);
            }
        for (int s = 0; s < c->seq_num && w->att_w2[s]; ++s)         // layer 2 runs in the same split form
            for (size_t i = 0; i < (size_t)ATT_H1 * ATT_H2; ++i) {
                const float v = fabsf(w->att_w2[s][i]);
                fin = fin && v == v; mx = fmaxf(mx, v);
            }
        n->din16 = fin && mx < 6.0e4f && !(opts & RL4RS_DIEN_OPT_NO_DIN16);
        // the first GRU in the same split form: E = 128 only, h-side weights inside fp16 range
        float gmx = 0.f;
        bool gfin = true;
        for (int s = 0; s < c->seq_num && w->gru_gate_w[s] && w->gru_cand_w[s]; ++s) {
            for (size_t i = 0; i < (size_t)2 * c->emb_size * 2 * c->emb_size; ++i) {
                const float v = fabsf(w->gru_gate_w[s][i]);
                gfin = gfin && v == v; gmx = fmaxf(gmx, v);
            }
            for (size_t i = 0; i < (size_t)2 * c->emb_size * c->emb_size; ++i) {
                const float v = fabsf(w->gru_cand_w[s][i]);
                gfin = gfin && v == v; gmx = fmaxf(gmx, v);
            }
        }
        n->gru16 = c->emb_size == 128 && gfin && gmx < 6.0e4f && !(opts & RL4RS_DIEN_OPT_NO_GRU16);
        // the plain GEMMs in the same split form: every weight finite (each packed matrix carries its own power-of-two
        // prescale, pack_gemm_weight_h16: no range condition).  Activations are split on the fly; one that leaves the fp16
        // range turns its output row into NaN (gemm.hip).
        bool wfin = true;
        auto chk = [&](const float* p, size_t cnt) {
            for (size_t i = 0; p && i < cnt; ++i) wfin = wfin && fabsf(p[i]) < 1.0e30f;      // false for NaN / inf too
        };
        chk(w->dense_w1, (size_t)c->dense_feature_num * c->hidden_units);
        chk(w->dense_w2, (size_t)c->hidden_units * c->hidden_units);
        chk(w->obs_w, (size_t)(c->seq_num * 2 * c->emb_size + c->hidden_units + (c->category_feature_num + 1) * c->emb_size) * OBS_DIM);
        for (int s = 0; s < c->seq_num; ++s) {
            chk(w->att_w1[s], (size_t)4 * c->emb_size * ATT_H1);
            chk(w->augru_gate_w[s], (size_t)c->emb_size * 4 * c->emb_size);
            chk(w->augru_cand_w[s], (size_t)c->emb_size * 2 * c->emb_size);
        }
        n->gemm16 = wfin && !(opts & RL4RS_DIEN_OPT_NO_GEMM16);
        bool cfin = true;
        for (size_t i = 0; i < (size_t)c->category_hash_size * c->emb_size; ++i) cfin = cfin && fabsf(w->cat_emb[i]) < 6.0e4f;
        n->cat16 = cfin && !(opts & RL4RS_DIEN_OPT_NO_CAT16);
    }
    {
        float* f = nullptr;
        int rc0 = alloc_f(n, &f, 1);
        if (rc0) return rc0;
        n->range_flag = reinterpret_cast<int*>(f);
        RL4RS_HIP_TRY(hipMemsetAsync(n->range_flag, 0, 4, (hipStream_t)stream));
    }
    {
        int dev = 0, cus = 0;
        RL4RS_HIP_TRY(hipGetDevice(&dev));
        RL4RS_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        n->n_cu = cus > 0 ? cus : 256;
    }
    for (int i = 0; i < KID_COUNT; ++i) { n->ms_total[i] = 0; n->launches[i] = 0; }
    int rc;
#define UP(dst, src, cnt) if ((rc = upload(n, &n->dst, (src), (size_t)(cnt), st)) != RL4RS_OK) return rc
#define AL(dst, cnt) if ((rc = alloc_f(n, &n->dst, (size_t)(cnt))) != RL4RS_OK) return rc
    std::vector<std::vector<float>> keep;    // host staging must outlive the async copies
    keep.reserve(64);
    UP(cat_emb, w->cat_emb, (size_t)H * E);
    UP(seq_emb, w->seq_emb, (size_t)H * E);
    // GEMM weights in the fragment order of the form that will run them (k_gemm_h16 in fp16x2 mode, else k_gemm_pk)
    auto pack_w = [&](const float* src, int64_t ldw, int kk, int nn) {
        return n->gemm16 ? pack_gemm_weight_h16(src, ldw, kk, nn) : pack_gemm_weight(src, ldw, kk, nn);
    };
    { auto pk = pack_w(w->dense_w1, U, Dn, U); keep.push_back(std::move(pk)); UP(dense_w1, keep.back().data(), keep.back().size()); }
    UP(dense_b1, w->dense_b1, U);
    { auto pk = pack_w(w->dense_w2, U, U, U); keep.push_back(std::move(pk)); UP(dense_w2, keep.back().data(), keep.back().size()); }
    UP(dense_b2, w->dense_b2, U);
    // head: table form unless disabled (RL4RS_DIEN_OPT_NO_HEAD_TABLES) or the tables would not fit a sane budget (8 GB)
    const int Kh = S * NH2 + U + E;                       // [sequence finals | dense | pooled attention]
    const bool use_tables = !(opts & RL4RS_DIEN_OPT_NO_HEAD_TABLES) && ((int64_t)Cn * H * OBS_DIM * 4 <= ((int64_t)8 << 30));
    n->ptab = nullptr;
    { auto pk = pack_w(w->obs_w, OBS_DIM, use_tables ? Kh : F, OBS_DIM); keep.push_back(std::move(pk)); UP(obs_w, keep.back().data(), keep.back().size()); }
    if (use_tables) {
        float* d_wflat;      // raw rows [Kh, F) of obs_w = the Cn blocks of E rows each
        if ((rc = upload(n, &d_wflat, w->obs_w + (size_t)Kh * OBS_DIM, (size_t)Cn * E * OBS_DIM, st))) return rc;
        AL(ptab, (size_t)Cn * H * OBS_DIM);
        for (int j = 0; j < Cn; ++j)
            if ((rc = launch_gemm_f32(n->cat_emb, E, d_wflat + (size_t)j * E * OBS_DIM, OBS_DIM, nullptr,
                                      n->ptab + (size_t)j * H * OBS_DIM, OBS_DIM, H, OBS_DIM, E, 0, st))) return rc;
    }
    UP(obs_b, w->obs_b, OBS_DIM);
    UP(out_w, w->out_w, (size_t)OBS_DIM * K);
    UP(out_b, w->out_b, K);
    std::vector<float> wac_all((size_t)E * S * ATT_H1);
    for (int s = 0; s < S; ++s) {
        RL4RS_REQUIRE(w->gru_gate_w[s] && w->gru_cand_w[s] && w->att_w1[s] && w->augru_gate_w[s] && w->augru_cand_w[s],
                      "dien_create: weights of sequence input %d missing", s);
        // ---- first GRU: x-side table weights [E, 3E] + bias, h-side packed
        std::vector<float> wx((size_t)E * 3 * E), bx(3 * E);
        for (int k = 0; k < E; ++k) {
            for (int j = 0; j < 2 * E; ++j) wx[(size_t)k * 3 * E + j] = w->gru_gate_w[s][(size_t)k * 2 * E + j];
            for (int j = 0; j < E; ++j) wx[(size_t)k * 3 * E + 2 * E + j] = w->gru_cand_w[s][(size_t)k * E + j];
        }
        for (int j = 0; j < 2 * E; ++j) bx[j] = w->gru_gate_b[s][j];
        for (int j = 0; j < E; ++j) bx[2 * E + j] = w->gru_cand_b[s][j];
        float *d_wx, *d_bx;
        if ((rc = upload(n, &d_wx, wx.data(), wx.size(), st))) return rc;
        if ((rc = upload(n, &d_bx, bx.data(), bx.size(), st))) return rc;
        AL(embw1[s], (size_t)H * 3 * E);
        if ((rc = launch_gemm_f32(n->seq_emb, E, d_wx, 3 * E, d_bx, n->embw1[s], 3 * E, H, 3 * E, E, 0, st))) return rc;
        keep.push_back(std::move(wx)); keep.push_back(std::move(bx));
        keep.push_back(pack_frag(w->gru_gate_w[s], 2 * E, E, E, 2 * E));
        UP(gru_wg[s], keep.back().data(), keep.back().size());
        keep.push_back(pack_frag(w->gru_cand_w[s], E, E, E, E));
        UP(gru_wc[s], keep.back().data(), keep.back().size());
        n->gru_wg16[s] = n->gru_wc16[s] = nullptr;
        if (n->gru16) {
            keep.push_back(pack_frag_h16(w->gru_gate_w[s], 2 * E, E, E, 2 * E));
            UP(gru_wg16[s], keep.back().data(), keep.back().size());
            keep.push_back(pack_frag_h16(w->gru_cand_w[s], E, E, E, E));
            UP(gru_wc16[s], keep.back().data(), keep.back().size());
        }
        // ---- power-of-two prescale of the AUGRU's recurrent (h-side) matrices in fp16x2 mode (pow2_prescale)
        // one scale per 32-column tile of the reset / update / candidate matrices = per (wave, gate) of the recurrence kernels
        std::vector<float> sgc(2 * NH2, 1.f), scc(NH2, 1.f);          // per-column view of the tile scales
        for (int g = 0; g < 3; ++g)
            for (int t = 0; t < 8; ++t) n->augru_s[s][g][t] = 1.f;
        if (n->fp16x2) {
            RL4RS_REQUIRE(NH2 == 256, "dien: the fp16x2 recurrence is built for 2 * emb_size = 256 hidden units");
            for (int g = 0; g < 3; ++g)
                for (int t = 0; t < NH2 / 32; ++t) {
                    float mx = 0.f;
                    for (int k = E; k < E + NH2; ++k)
                        for (int j = 32 * t; j < 32 * t + 32; ++j)
                            mx = fmaxf(mx, fabsf(g < 2 ? w->augru_gate_w[s][(size_t)k * 2 * NH2 + g * NH2 + j] : w->augru_cand_w[s][(size_t)k * NH2 + j]));
                    const float sc_t = pow2_prescale(mx);
                    n->augru_s[s][g][t] = sc_t;
                    for (int j = 32 * t; j < 32 * t + 32; ++j) { if (g < 2) sgc[g * NH2 + j] = sc_t; else scc[j] = sc_t; }
                }
        }
        // ---- projections of h1: [W1b - W1c | augru gate x-side | augru cand x-side], bias [b1 | bg | bc]
        std::vector<float> wp((size_t)E * PLD), bp(PLD), wac((size_t)E * ATT_H1);
        const float* w1 = w->att_w1[s];    // rows: q [0,E) | k [E,2E) | q-k [2E,3E) | q*k [3E,4E)
        for (int k = 0; k < E; ++k) {
            for (int j = 0; j < ATT_H1; ++j) {
                wp[(size_t)k * PLD + j] = w1[(size_t)(E + k) * ATT_H1 + j] - w1[(size_t)(2 * E + k) * ATT_H1 + j];
                wac[(size_t)k * ATT_H1 + j] = w1[(size_t)k * ATT_H1 + j] + w1[(size_t)(2 * E + k) * ATT_H1 + j];
            }
            // (fp16x2: the AUGRU sections arrive at the recurrence as MFMA C-in next to products of PRESCALED weights: same scale)
            for (int j = 0; j < 2 * NH2; ++j) wp[(size_t)k * PLD + ATT_H1 + j] = w->augru_gate_w[s][(size_t)k * 2 * NH2 + j] * sgc[j];
            for (int j = 0; j < NH2; ++j) wp[(size_t)k * PLD + ATT_H1 + 2 * NH2 + j] = w->augru_cand_w[s][(size_t)k * NH2 + j] * scc[j];
        }
        for (int j = 0; j < ATT_H1; ++j) bp[j] = w->att_b1[s][j];
        for (int j = 0; j < 2 * NH2; ++j) bp[ATT_H1 + j] = w->augru_gate_b[s][j] * sgc[j];
        for (int j = 0; j < NH2; ++j) bp[ATT_H1 + 2 * NH2 + j] = w->augru_cand_b[s][j] * scc[j];
        keep.push_back(pack_w(wp.data(), PLD, E, PLD)); UP(wproj[s], keep.back().data(), keep.back().size());
        keep.push_back(std::move(bp)); UP(bproj[s], keep.back().data(), keep.back().size());
        for (int k = 0; k < E; ++k)
            for (int j = 0; j < ATT_H1; ++j) wac_all[(size_t)k * S * ATT_H1 + s * ATT_H1 + j] = wac[(size_t)k * ATT_H1 + j];
        keep.push_back(std::move(wac)); UP(w1ac[s], keep.back().data(), keep.back().size());
        keep.push_back(pack_frag(w1, ATT_H1, 3 * E, E, ATT_H1));
        UP(w1d[s], keep.back().data(), keep.back().size());
        n->w1d16[s] = nullptr;
        if (n->fp16x2 && n->din16) {
            keep.push_back(pack_frag_h16(w1, ATT_H1, 3 * E, E, ATT_H1));
            UP(w1d16[s], keep.back().data(), keep.back().size());
        }
        UP(att_w2[s], w->att_w2[s], ATT_H1 * ATT_H2);
        UP(att_b2[s], w->att_b2[s], ATT_H2);
        UP(att_w3[s], w->att_w3[s], ATT_H2);
        UP(att_b3[s], w->att_b3[s], 1);
        keep.push_back(pack_frag(w->augru_gate_w[s], 2 * NH2, E, NH2, 2 * NH2));
        UP(augru_wg[s], keep.back().data(), keep.back().size());
        keep.push_back(pack_frag(w->augru_cand_w[s], NH2, E, NH2, NH2));
        UP(augru_wc[s], keep.back().data(), keep.back().size());
        n->augru_wg16[s] = n->augru_wc16[s] = nullptr;
        if (n->fp16x2) {
            std::vector<float> hs((size_t)NH2 * 2 * NH2);
            for (size_t i = 0; i < hs.size(); ++i) hs[i] = w->augru_gate_w[s][(size_t)E * 2 * NH2 + i] * sgc[i % (2 * NH2)];
            keep.push_back(pack_frag_h16(hs.data(), 2 * NH2, 0, NH2, 2 * NH2));
            UP(augru_wg16[s], keep.back().data(), keep.back().size());
            hs.resize((size_t)NH2 * NH2);
            for (size_t i = 0; i < hs.size(); ++i) hs[i] = w->augru_cand_w[s][(size_t)E * NH2 + i] * scc[i % NH2];
            keep.push_back(pack_frag_h16(hs.data(), NH2, 0, NH2, NH2));
            UP(augru_wc16[s], keep.back().data(), keep.back().size());
        }
        AL(h1[s], ((size_t)c->max_slots + 1) * L * E);                  // (+ 1 everywhere: slot max_slots holds the all-zero sequence)
        n->h1f[s] = nullptr;
        if (n->fp16x2 && n->din16 && n->din_x && E == 128 && L <= 64) AL(h1f[s], ((size_t)c->max_slots + 1) * ((L + 31) / 32) * 32 * E);
        AL(proj[s], ((size_t)c->max_slots + 1) * L * PLD);
    }
    keep.push_back(pack_w(wac_all.data(), S * ATT_H1, E, S * ATT_H1));
    UP(w1ac_all, keep.back().data(), keep.back().size());
    // table form: the Flatten(category_emb) columns are never materialised, so the rows are only Kh wide (contiguous
    // 3 KB rows for the head GEMM instead of 3 KB out of every 13.8 KB)
    n->Fld = n->ptab ? Kh : F;
    AL(allf, (size_t)c->max_rows * n->Fld);
    AL(dh, (size_t)c->max_rows * U);
    AL(q, (size_t)c->max_rows * E);
    AL(qa, (size_t)S * c->max_rows * ATT_H1);
    AL(scores, (size_t)S * c->max_rows * L);
    AL(obs_tmp, (size_t)c->max_rows * OBS_DIM);
    n->tsum = nullptr;
    if (n->ptab && Cn <= 24 && !(opts & RL4RS_DIEN_OPT_NO_HEAD_FUSED)) AL(tsum, (size_t)c->max_rows * OBS_DIM);
    // first GRU: the states after 1 .. L leading zero ids, per sequence input (RecurArgs::pad) - the kernel itself on ONE all-zero row
    for (int s = 0; s < 4; ++s) { n->gru_pad[s] = nullptr; n->lead[s] = nullptr; }
    if (n->gru16 && !(opts & RL4RS_DIEN_OPT_NO_GRU_PAD)) {
        float* zero_ids = nullptr;                       // L zero int32 ids (all-zero bits either way)
        if ((rc = alloc_f(n, &zero_ids, (size_t)L)) != RL4RS_OK) return rc;
        RL4RS_HIP_TRY(hipMemsetAsync(zero_ids, 0, (size_t)L * 4, st));
        const size_t smem16 = (((size_t)4 * 32 * (128 + 8) * 2 + (size_t)32 * (L + 1) * 4 + 15) & ~(size_t)15) + (size_t)4 * 8 * 2048;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_gru_h16), smem16))) return rc;
        n->gru16_attr = true;
        for (int s = 0; s < S; ++s) {
            AL(gru_pad[s], (size_t)L * E);
            RecurArgs a;
            memset(&a, 0, sizeof(a));
            a.n_rows = 1; a.L = L; a.group = 1;
            a.xbase[0] = n->embw1[s]; a.xld = 3 * E; a.xoff = 0; a.xbytes = (int64_t)H * 3 * E * 4;
            a.ids = reinterpret_cast<const int32_t*>(zero_ids);
            a.wg[0] = n->gru_wg16[s]; a.wc[0] = n->gru_wc16[s];
            a.out = n->gru_pad[s]; a.out_ld = E; a.slot_base = 0;
            hipLaunchKernelGGL(k_gru_h16, dim3(1, 1), dim3(256), smem16, st, a);
            RL4RS_LAUNCH_CHECK();
            // ... and the cache slot of the all-zero sequence (slot max_slots): states, fragment copy, projections - what encode does
            const int P = c->max_slots;
            RL4RS_HIP_TRY(hipMemcpyAsync(n->h1[s] + (size_t)P * L * E, n->gru_pad[s], (size_t)L * E * 4, hipMemcpyDeviceToDevice, st));
            if (n->h1f[s]) {
                const int64_t pieces = (int64_t)((L + 31) / 32) * 1024;
                hipLaunchKernelGGL(k_h1_frag, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, n->h1[s], n->h1f[s], P, 1, L);
                RL4RS_LAUNCH_CHECK();
            }
            if ((rc = scorer_gemm(n, n->h1[s] + (size_t)P * L * E, E, n->wproj[s], n->bproj[s], n->proj[s] + (size_t)P * L * n->PLD, n->PLD,
                                  L, n->PLD, E, 0, st))) return rc;
            float* lead_f = nullptr;
            if ((rc = alloc_f(n, &lead_f, (size_t)c->max_slots + 1)) != RL4RS_OK) return rc;
            RL4RS_HIP_TRY(hipMemsetAsync(lead_f, 0, ((size_t)c->max_slots + 1) * 4, st));
            n->lead[s] = reinterpret_cast<int32_t*>(lead_f);
        }
    }
#undef UP
#undef AL
    // LDS opt-in above the 64 KB default where needed
    {
        size_t sm_aug = (size_t)(2 * 32 * (NH2 + 4) + 32 * (L + 1) + 32) * 4;
        size_t sm_gru = (size_t)(2 * 32 * (E + 4) + 32 * (L + 1) + 32) * 4;
        const void* augru_variants[] = {
            reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 0>),
#ifdef RL4RS_ABLATE
            reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 1>), reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 2>),
            reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 4>), reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 8>),
            reinterpret_cast<const void*>(&k_recur<256, true, AUGRU_U, 15>),
#endif
        };
        // (per-function limits, only ever raised: handles with different maxlen share these kernels)
        for (const void* f : augru_variants)
            if ((rc = raise_dyn_smem(f, sm_aug))) return rc;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_augru_h16<1, RL4RS_H16_RING1, RL4RS_H16_NRES>), augru_h16_smem(1, NH2, L)))) return rc;
        if ((rc = augru_x_prepare())) return rc;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur<128, false, GRU_U>), sm_gru))) return rc;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_din_x), din_x_smem()))) return rc;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_din_scores<true, false>), 96 * 1024))) return rc;
        if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_din_scores<true, true>), 96 * 1024))) return rc;
    }
    RL4RS_HIP_TRY(hipStreamSynchronize(st));   // host staging (keep) may now be released
    *out = n;
    return RL4RS_OK;
}

int rl4rs_dien_destroy(rl4rs_dien* n) {
    if (!n) return RL4RS_OK;
    for (void* p : n->owned) (void)hipFree(p);
    for (auto& e : n->pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (auto e : n->pool) (void)hipEventDestroy(e);
    if (n->side_stream) { (void)hipStreamDestroy(n->side_stream); (void)hipEventDestroy(n->ev_fork); (void)hipEventDestroy(n->ev_join); }
    delete n;
    return RL4RS_OK;
}

int rl4rs_dien_encode(rl4rs_dien* n, int32_t s, const int32_t* ids, int32_t cnt, int32_t slot_base, void* stream) {
    RL4RS_REQUIRE(n && ids, "dien_encode: null argument");
    RL4RS_REQUIRE(s >= 0 && s < n->S, "dien_encode: sequence input %d out of range", s);
    RL4RS_REQUIRE(cnt > 0 && slot_base >= 0 && slot_base + cnt <= n->c.max_slots,
                  "dien_encode: slots [%d,%d) exceed max_slots=%d", slot_base, slot_base + cnt, n->c.max_slots);
    hipStream_t st = (hipStream_t)stream;
    const int E = n->E, L = n->L;
    {
        Prof p(n, KID_GRU1, st);
        RecurArgs a;
        memset(&a, 0, sizeof(a));
        a.n_rows = cnt; a.L = L; a.group = 1;
        a.xbase[0] = n->embw1[s]; a.xld = 3 * E; a.xoff = 0; a.xbytes = (int64_t)n->H * 3 * E * 4;
        a.ids = ids; a.slots = nullptr; a.slots_stride = 0;
        a.wg[0] = n->gru_wg[s]; a.wc[0] = n->gru_wc[s]; a.att = nullptr; a.att_stride = 0;
        a.out = n->h1[s]; a.out_ld = E; a.out_off = 0; a.out_seq_off = 0; a.slot_base = slot_base;
        if (n->gru16) {
            a.wg[0] = n->gru_wg16[s]; a.wc[0] = n->gru_wc16[s];
            a.pad = n->gru_pad[s]; a.lead_out = n->lead[s];
            const size_t smem16 = (((size_t)4 * 32 * (128 + 8) * 2 + (size_t)32 * (L + 1) * 4 + 15) & ~(size_t)15) + (size_t)4 * 8 * 2048;
            if (!n->gru16_attr) {
                int rc16 = raise_dyn_smem(reinterpret_cast<const void*>(&k_gru_h16), smem16);
                if (rc16) return rc16;
                n->gru16_attr = true;
            }
            hipLaunchKernelGGL(k_gru_h16, dim3((cnt + 31) / 32, 1), dim3(256), smem16, st, a);
        } else {
            size_t smem = (size_t)(2 * 32 * (E + 4) + 32 * (L + 1) + 32) * 4;
            hipLaunchKernelGGL((k_recur<128, false, GRU_U>), dim3((cnt + 31) / 32, 1), dim3(256), smem, st, a);
        }
        if (n->h1f[s]) {
            const int64_t pieces = (int64_t)cnt * ((L + 31) / 32) * 1024;
            hipLaunchKernelGGL(k_h1_frag, dim3((unsigned)((pieces + 255) / 256 < 8192 ? (pieces + 255) / 256 : 8192)), dim3(256), 0, st,
                               n->h1[s], n->h1f[s], slot_base, cnt, L);
        }
        RL4RS_LAUNCH_CHECK();
    }
    {
        Prof p(n, KID_PROJ, st);
        int rc = scorer_gemm(n, n->h1[s] + (size_t)slot_base * L * E, E, n->wproj[s], n->bproj[s],
                                    n->proj[s] + (size_t)slot_base * L * n->PLD, n->PLD, cnt * L, n->PLD, E, 0, st);
        if (rc) return rc;
    }
    return RL4RS_OK;
}

int rl4rs_dien_forward(rl4rs_dien* n, int32_t R, int32_t group, const float* dense, const int32_t* cat,
                       const int32_t* slots, float* obs, float* prob, void* stream) {
    RL4RS_REQUIRE(n && dense && cat && slots, "dien_forward: null argument");
    RL4RS_REQUIRE(R > 0 && R <= n->c.max_rows, "dien_forward: R=%d exceeds max_rows=%d", R, n->c.max_rows);
    RL4RS_REQUIRE(group >= 1 && R % group == 0, "dien_forward: R=%d is not a multiple of group=%d", R, group);
    hipStream_t st = (hipStream_t)stream;
    const int E = n->E, U = n->U, L = n->L, S = n->S, Cn = n->Cn, F = n->Fld, NH2 = n->NH2;
    const int off_d = S * NH2, off_c = off_d + U;
    const int ngroups = R / group;
    int rc;
    n->obs_mirror_used = false;
    struct MirrorOnce { rl4rs_dien* n; ~MirrorOnce() { n->obs_mirror = nullptr; } } mirror_once{n};       // one forward only
    auto dense_tower = [&](hipStream_t s2) -> int {
        int r2;
        if (n->gemm16 && n->dense_chain && U <= 128 && U % 16 == 0) {      // both layers in one launch, the hidden tile stays in LDS
            if ((r2 = launch_gemm_h16_chain(dense, n->Dn, n->dense_w1, n->dense_b1, U, n->Dn, 1, n->dense_w2, n->dense_b2,
                                            n->allf + off_d, F, U, 1, R, s2))) return r2;
        } else {
            if ((r2 = scorer_gemm(n, dense, n->Dn, n->dense_w1, n->dense_b1, n->dh, U, R, U, n->Dn, 1, s2))) return r2;
            if ((r2 = scorer_gemm(n, n->dh, U, n->dense_w2, n->dense_b2, n->allf + off_d, F, R, U, U, 1, s2))) return r2;
        }
        return RL4RS_OK;
    };
    // RL4RS_DIEN_OPT_DENSE_FORK: the dense tower reads only `dense` and writes its own columns of allf - it runs on a second stream
    // of the handle from the top of the forward and is joined in front of the head GEMM (event pair owned by the handle)
    bool forked = false;
    if (n->dense_fork) {
        if (!n->side_stream) {
            RL4RS_HIP_TRY(hipStreamCreateWithFlags(&n->side_stream, hipStreamNonBlocking));
            RL4RS_HIP_TRY(hipEventCreateWithFlags(&n->ev_fork, hipEventDisableTiming));
            RL4RS_HIP_TRY(hipEventCreateWithFlags(&n->ev_join, hipEventDisableTiming));
        }
        RL4RS_HIP_TRY(hipEventRecord(n->ev_fork, st));                       // (the inputs - and the previous forward's readers of allf - are ordered before this point)
        RL4RS_HIP_TRY(hipStreamWaitEvent(n->side_stream, n->ev_fork, 0));
        if ((rc = dense_tower(n->side_stream))) return rc;
        RL4RS_HIP_TRY(hipEventRecord(n->ev_join, n->side_stream));
        forked = true;
    }
    {
        Prof p(n, KID_CAT, st);
        if (n->cat_v2 && n->cat_group && E == 128 && Cn >= 11 && Cn <= 24 && (group == 8 || group == 9)) {
            // rows in groups that (normally) share all but the last category id: one workgroup per group, shared gathers once
            const size_t smem = cat_attn2g_smem(group, Cn);
            if (group == 8)
                hipLaunchKernelGGL(k_cat_attn2g<8>, dim3(ngroups), dim3(512), smem, st, cat, Cn, n->H, n->cat_emb, n->seq_emb, n->allf, F,
                                   off_c, n->q, n->ptab ? 0 : 1, (n->fp16x2 && n->cat16) ? 1 : 0, n->ptab, n->obs_b, n->tsum);
            else
                hipLaunchKernelGGL(k_cat_attn2g<9>, dim3(ngroups), dim3(576), smem, st, cat, Cn, n->H, n->cat_emb, n->seq_emb, n->allf, F,
                                   off_c, n->q, n->ptab ? 0 : 1, (n->fp16x2 && n->cat16) ? 1 : 0, n->ptab, n->obs_b, n->tsum);
        } else if (n->cat_v2 && E == 128 && Cn <= 24) {
            size_t smem = (size_t)4 * (Cn * (64 + 4) + 32) * 4;
            if (Cn == 21)
                hipLaunchKernelGGL((k_cat_attn2<21, true>), dim3((R + 3) / 4), dim3(256), smem, st, cat, R, Cn, n->H, n->cat_emb,
                                   n->seq_emb, n->allf, F, off_c, n->q, n->ptab ? 0 : 1, (n->fp16x2 && n->cat16) ? 1 : 0,
                                   n->ptab, n->obs_b, n->tsum);
            else
                hipLaunchKernelGGL((k_cat_attn2<24, false>), dim3((R + 3) / 4), dim3(256), smem, st, cat, R, Cn, n->H, n->cat_emb,
                                   n->seq_emb, n->allf, F, off_c, n->q, n->ptab ? 0 : 1, (n->fp16x2 && n->cat16) ? 1 : 0,
                                   n->ptab, n->obs_b, n->tsum);
        } else {
            size_t smem = (size_t)4 * (Cn * (E + 4) + 32) * 4;
            hipLaunchKernelGGL(k_cat_attn, dim3((R + 3) / 4), dim3(256), smem, st, cat, R, Cn, E, n->H, n->cat_emb,
                               n->seq_emb, n->allf, F, off_c, n->q, n->ptab ? 0 : 1, (n->fp16x2 && n->cat16) ? 1 : 0,
                               n->ptab, n->obs_b, n->tsum);
        }
        RL4RS_LAUNCH_CHECK();
    }
    if (!forked) {
        Prof p(n, KID_DENSE, st);
        if ((rc = dense_tower(st))) return rc;
    }
    {
        Prof p(n, KID_DIN, st);
        DinArgs a;
        memset(&a, 0, sizeof(a));
        a.R = R; a.L = L; a.E = E; a.group = group; a.n_groups = ngroups;
        a.slots = slots; a.slots_stride = ngroups; a.pld = n->PLD; a.q = n->q;
        a.qa = n->qa; a.qa_stride = ATT_H1; a.qa_ld = S * ATT_H1;       // one GEMM for the q-side term of every input: [R, S*64]
        const bool h16 = n->fp16x2 && n->din16;
        if ((rc = scorer_gemm(n, n->q, E, n->w1ac_all, nullptr, n->qa, S * ATT_H1, R, S * ATT_H1, E, 0, st))) return rc;
        for (int s = 0; s < S; ++s) {
            a.h1[s] = n->h1[s]; a.h1f[s] = n->h1f[s]; a.proj[s] = n->proj[s]; a.w1ac[s] = n->w1ac[s]; a.w1d[s] = n->w1d[s]; a.w1d16[s] = n->w1d16[s];
            a.w2[s] = n->att_w2[s]; a.b2[s] = n->att_b2[s]; a.w3[s] = n->att_w3[s]; a.b3[s] = n->att_b3[s];
        }
        a.scores = n->scores; a.scores_stride = (int64_t)n->c.max_rows * L;
        a.order = (n->row_order && n->row_order_n == ngroups) ? n->row_order : nullptr;
        for (int s = 0; s < S; ++s) a.lead[s] = n->lead[s];
        a.pad_slot = n->c.max_slots;
#ifdef RL4RS_DINX_TRACE      // timing experiments only (tools/dinx_trace.py)
        {
            static unsigned long long* din_trace = nullptr;
            if (!din_trace) { (void)hipMalloc((void**)&din_trace, 8 * 5 * 8 * 8); (void)hipMemset(din_trace, 0, 8 * 5 * 8 * 8); }
            a.trace = din_trace;
            if (getenv("RL4RS_DINX_TRACE_DUMP")) {       // the marks of the PREVIOUS launch
                unsigned long long host[8 * 5 * 8];
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(host, din_trace, sizeof(host), hipMemcpyDeviceToHost);
                FILE* f = fopen(getenv("RL4RS_DINX_TRACE_DUMP"), "wb");
                if (f) { fwrite(host, 1, sizeof(host), f); fclose(f); }
            }
        }
#endif
        if (h16 && n->h1f[0]) {
            // 16 rows per 8-wave workgroup: an obs-sized launch (R = 4096, two inputs) is two workgroups per CU, one round
            hipLaunchKernelGGL(k_din_x, dim3((R + 15) / 16, S), dim3(512), din_x_smem(), st, a, 16);
        } else if (group == 1) {
            const int nw = 4;
            size_t smem = ((size_t)2 * 16 * 64 + 48 + (size_t)nw * (E + ATT_H1)) * 4;
            if (h16) hipLaunchKernelGGL((k_din_scores<false, true>), dim3((R + nw - 1) / nw, S), dim3(64 * nw), smem, st, a);
            else hipLaunchKernelGGL((k_din_scores<false, false>), dim3((R + nw - 1) / nw, S), dim3(64 * nw), smem, st, a);
        } else {
            int nw = group % 4 == 0 ? 4 : (group % 3 == 0 ? 3 : (group % 2 == 0 ? 2 : (group < 4 ? group : 4)));
            size_t smem = ((size_t)L * (E + 4) + 2 * 16 * 64 + 48 + (size_t)nw * (E + ATT_H1)) * 4;
            if (h16) hipLaunchKernelGGL((k_din_scores<true, true>), dim3(ngroups, S), dim3(64 * nw), smem, st, a);
            else hipLaunchKernelGGL((k_din_scores<true, false>), dim3(ngroups, S), dim3(64 * nw), smem, st, a);
        }
        RL4RS_LAUNCH_CHECK();
    }
    {
        Prof p(n, KID_AUGRU, st);
        size_t smem = (size_t)(2 * 32 * (NH2 + 4) + 32 * (L + 1) + 32) * 4;
        RecurArgs a;
        memset(&a, 0, sizeof(a));
        a.n_rows = R; a.L = L; a.group = group;
        a.xld = n->PLD; a.xoff = ATT_H1; a.xbytes = ((int64_t)n->c.max_slots + 1) * L * n->PLD * 4;
        a.ids = nullptr; a.slots = slots; a.slots_stride = ngroups;
        for (int s = 0; s < S; ++s) a.lead[s] = n->lead[s];
        a.pad_slot = n->c.max_slots;
        for (int s = 0; s < S; ++s) { a.xbase[s] = n->proj[s]; a.wg[s] = n->augru_wg[s]; a.wc[s] = n->augru_wc[s]; }
        a.att = n->scores; a.att_stride = (int64_t)n->c.max_rows * L;
        a.out = n->allf; a.out_ld = F; a.out_off = 0; a.out_seq_off = NH2; a.slot_base = 0;
        dim3 grid((R + 31) / 32, S), block(512);
        if (n->fp16x2) {
            for (int s = 0; s < S; ++s) { a.wg[s] = n->augru_wg16[s]; a.wc[s] = n->augru_wc16[s]; }
            a.range_flag = n->range_flag;
            for (int s = 0; s < S; ++s)
                for (int t = 0; t < 8; ++t) {
                    a.k_r[s][t] = -1.4426950408889634f / n->augru_s[s][0][t];
                    a.k_u[s][t] = -1.4426950408889634f / n->augru_s[s][1][t];
                    a.k_c[s][t] = 2.8853900817779268f / n->augru_s[s][2][t];
                }
            a.order = (n->augru_x && n->row_order && n->row_order_n == ngroups) ? n->row_order : nullptr;
            a.steps = 0;
#if defined(RL4RS_H16_TRACE) || defined(RL4RS_X_TRACE)
            static unsigned long long* trace_buf = nullptr;
            if (!trace_buf) { (void)hipMalloc((void**)&trace_buf, 8 * 4 * 8 * 8); (void)hipMemset(trace_buf, 0, 8 * 4 * 8 * 8); }
            a.trace = trace_buf;
            if (getenv("RL4RS_H16_TRACE_DUMP")) {
                unsigned long long host[8 * 4 * 8];
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(host, trace_buf, sizeof(host), hipMemcpyDeviceToHost);
                FILE* f = fopen(getenv("RL4RS_H16_TRACE_DUMP"), "wb");
                if (f) { fwrite(host, 1, sizeof(host), f); fclose(f); }
            }
#endif
            // 64-row workgroups when the rows come in whole groups of 8 per cache slot (the reward forward) and there are enough
            // of them to keep every CU busy; 32-row workgroups otherwise (obs-sized launches: one 32-row tile per CU)
            // (n->augru_rows = 32 / 64 pins the form: rl4rs_dien_cfg.kernel_opts, rl4rs_dien_set_augru_rows)
            const bool mt2 = n->augru_rows != 32 && group % 8 == 0 && R % 64 == 0 &&
                             (n->augru_rows == 64 || (int64_t)(R / 64) * S >= 2 * (int64_t)n->n_cu);
            if (n->augru_x)
                augru_x_launch(mt2 ? 64 : 32, S, st, a);
            else
                hipLaunchKernelGGL((k_augru_h16<1, RL4RS_H16_RING1, RL4RS_H16_NRES>), grid, block, augru_h16_smem(1, NH2, L), st, a);
        } else
#ifdef RL4RS_ABLATE      // timing experiments only (tools/ablate_augru.sh builds with -DRL4RS_ABLATE)
        static const int ablate = getenv("RL4RS_AUGRU_ABLATE") ? atoi(getenv("RL4RS_AUGRU_ABLATE")) : 0;
        switch (ablate) {
            case 1: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 1>), grid, block, smem, st, a); break;
            case 2: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 2>), grid, block, smem, st, a); break;
            case 4: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 4>), grid, block, smem, st, a); break;
            case 8: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 8>), grid, block, smem, st, a); break;
            case 15: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 15>), grid, block, smem, st, a); break;
            default: hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 0>), grid, block, smem, st, a); break;
        }
#else
        hipLaunchKernelGGL((k_recur<256, true, AUGRU_U, 0>), grid, block, smem, st, a);
#endif
        ;
        RL4RS_LAUNCH_CHECK();
    }
    float* obs_out = obs ? obs : n->obs_tmp;
    if (forked) RL4RS_HIP_TRY(hipStreamWaitEvent(st, n->ev_join, 0));
    {
        Prof p(n, KID_HEAD, st);
        if (n->ptab && n->tsum) {     // obs = ELU(allf W + [b + table rows]): the addend was built inside k_cat_attn
            const int Kh = S * NH2 + U + E;
            if (n->gemm16) {
                float* mirror = (obs && n->obs_mirror) ? n->obs_mirror : nullptr;
                rc = launch_gemm_h16(n->allf, F, n->obs_w, nullptr, obs_out, OBS_DIM, R, OBS_DIM, Kh, 1, st, n->tsum, OBS_DIM, mirror, OBS_DIM);
                n->obs_mirror_used = mirror != nullptr;
            }
            else rc = launch_gemm_packed(n->allf, F, n->obs_w, nullptr, obs_out, OBS_DIM, R, OBS_DIM, Kh, 1, st, n->tsum, OBS_DIM);
            if (rc) return rc;
        } else if (n->ptab) {
            const int Kh = S * NH2 + U + E;
            if ((rc = scorer_gemm(n, n->allf, F, n->obs_w, nullptr, obs_out, OBS_DIM, R, OBS_DIM, Kh, 0, st))) return rc;
            hipLaunchKernelGGL(k_head_finish, dim3((R + 3) / 4), dim3(256), 0, st, obs_out, R, cat, Cn, n->H, n->ptab, n->obs_b);
            RL4RS_LAUNCH_CHECK();
        } else {
            if ((rc = scorer_gemm(n, n->allf, F, n->obs_w, n->obs_b, obs_out, OBS_DIM, R, OBS_DIM, n->F, 1, st))) return rc;
        }
    }
    if (prob) {
        Prof p(n, KID_PROB, st);
        hipLaunchKernelGGL(k_head_prob, dim3((R + 3) / 4), dim3(256), 0, st, obs_out, R, OBS_DIM, n->K, n->out_w,
                           n->out_b, prob);
        RL4RS_LAUNCH_CHECK();
    }
    return RL4RS_OK;
}

int rl4rs_dien_head_prob(rl4rs_dien* n, int32_t R, const float* obs, float* prob, void* stream) {
    RL4RS_REQUIRE(n && obs && prob && R > 0, "dien_head_prob: bad argument");
    hipStream_t st = (hipStream_t)stream;
    Prof p(n, KID_PROB, st);
    hipLaunchKernelGGL(k_head_prob, dim3((R + 3) / 4), dim3(256), 0, st, obs, R, OBS_DIM, n->K, n->out_w, n->out_b, prob);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_dien_buffer(rl4rs_dien* n, int which, void** p, int64_t* bytes) {
    RL4RS_REQUIRE(n && p, "dien_buffer: null argument");
    int64_t b = 0;
    void* ptr = nullptr;
    switch (which) {
        case RL4RS_DIEN_ALL_FEATURE: ptr = n->allf; b = (int64_t)n->c.max_rows * n->Fld * 4; break;
        case RL4RS_DIEN_SCORES: ptr = n->scores; b = (int64_t)n->S * n->c.max_rows * n->L * 4; break;
        case RL4RS_DIEN_QUERY: ptr = n->q; b = (int64_t)n->c.max_rows * n->E * 4; break;
        case RL4RS_DIEN_H1: ptr = n->h1[0]; b = (int64_t)n->c.max_slots * n->L * n->E * 4; break;
        default: set_error("dien_buffer: unknown buffer id %d", which); return RL4RS_EINVAL;
    }
    *p = ptr;
    if (bytes) *bytes = b;
    return RL4RS_OK;
}

// Processing order of the row groups of the following forwards: order_dev[i] = the group (env) handled at position i, a
// permutation of 0 .. n_groups-1 (caller-owned device memory, NULL = identity).  A pure locality hint - every row's result is
// unchanged: with the envs sorted by the cache slot of their history, rows that read the same first-GRU states / cached
// projections sit in the same or in neighbouring tiles, so duplicates hit in L2 instead of going to HBM again.  Used when
// n_groups matches the forward's R / group.
int rl4rs_dien_set_row_order(rl4rs_dien* n, const int32_t* order_dev, int32_t n_groups) {
    RL4RS_REQUIRE(n && n_groups >= 0, "dien_set_row_order: bad argument");
    n->row_order = order_dev;
    n->row_order_n = order_dev ? n_groups : 0;
    return RL4RS_OK;
}

// Row-tile form of k_augru_x for the following forwards: 0 = automatic (64-row workgroups for reward-sized launches, see
// rl4rs_dien_forward), 32 = always 32-row workgroups, 64 = 64-row workgroups whenever the launch shape admits them (rows in
// whole groups of 8 per cache slot, R % 64 == 0).  Both forms run the same MFMA sequence per row: results are bit-identical.
int rl4rs_dien_set_augru_rows(rl4rs_dien* n, int32_t rows) {
    RL4RS_REQUIRE(n && (rows == 0 || rows == 32 || rows == 64), "dien_set_augru_rows: rows must be 0, 32 or 64");
    n->augru_rows = rows;
    return RL4RS_OK;
}

int rl4rs_dien_set_profiling(rl4rs_dien* n, int enable) {
    RL4RS_REQUIRE(n, "dien_set_profiling: null handle");
    n->profiling = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
    return RL4RS_OK;
}
int rl4rs_dien_scorer_mode(rl4rs_dien* n, int32_t* mode) {
    RL4RS_REQUIRE(n && mode, "dien_scorer_mode: null argument");
    *mode = n->fp16x2 ? RL4RS_SCORER_FP16X2 : RL4RS_SCORER_FP32;
    return RL4RS_OK;
}
// Synchronises the stream, returns and clears the handle's status bits (include/rl4rs_hip.h RL4RS_DIEN_STATUS_*).
int rl4rs_dien_status(rl4rs_dien* n, int32_t* flags, void* stream) {
    RL4RS_REQUIRE(n && flags, "dien_status: null argument");
    hipStream_t st = (hipStream_t)stream;
    int v = 0;
    RL4RS_HIP_TRY(hipMemcpyAsync(&v, n->range_flag, 4, hipMemcpyDeviceToHost, st));
    RL4RS_HIP_TRY(hipStreamSynchronize(st));
    if (v) RL4RS_HIP_TRY(hipMemsetAsync(n->range_flag, 0, 4, st));
    *flags = v ? RL4RS_DIEN_STATUS_FP16_RANGE : 0;
    return RL4RS_OK;
}
}  // extern "C"
namespace rl4rs {
void dien_set_obs_mirror(rl4rs_dien* n, float* host_visible) { if (n) n->obs_mirror = host_visible; }
bool dien_obs_mirror_used(const rl4rs_dien* n) { return n && n->obs_mirror_used; }
}  // namespace rl4rs
extern "C" {
// The status word itself (device pointer owned by the handle, != 0 <=> RL4RS_DIEN_STATUS_FP16_RANGE pending): a caller that
// already copies a record to the host every step reads it there instead of paying rl4rs_dien_status's synchronisation.
int rl4rs_dien_status_word(rl4rs_dien* n, int32_t** word_dev) {
    RL4RS_REQUIRE(n && word_dev, "dien_status_word: null argument");
    *word_dev = n->range_flag;
    return RL4RS_OK;
}
int rl4rs_dien_kernel_count(void) { return KID_COUNT; }
// The kernels THIS handle launches for kernel class `which` (they depend on the scorer mode and the handle's options), as they
// appear in a rocprofv3 kernel trace - the static class names of rl4rs_dien_kernel_name are those of the fp32 build.
int rl4rs_dien_kernel_label(rl4rs_dien* n, int which, char* buf, int32_t cap) {
    RL4RS_REQUIRE(n && buf && cap > 0 && which >= 0 && which < KID_COUNT, "dien_kernel_label: bad argument");
    const char* gemm = n->gemm16 ? "k_gemm_h16" : "k_gemm_pk";
    const bool din_x = n->fp16x2 && n->din16 && n->h1f[0];
    std::string s;
    switch (which) {
        case KID_CAT: s = (n->cat_v2 && n->E == 128 && n->Cn <= 24) ? (n->cat_group ? "k_cat_attn2 / k_cat_attn2g (grouped rows)" : "k_cat_attn2") : "k_cat_attn"; break;
        case KID_DENSE:
            s = (n->gemm16 && n->dense_chain && n->U <= 128 && n->U % 16 == 0) ? "k_gemm_h16<chain>(dense tower, both layers)"
                                                                                : std::string(gemm) + " x2 (dense tower)";
            break;
        case KID_DIN:
            s = std::string(din_x ? "k_din_x" : (n->fp16x2 && n->din16 ? "k_din_scores<h16>" : "k_din_scores")) + " + " + gemm + "(q-side term)";
            break;
        case KID_AUGRU:
            s = !n->fp16x2 ? "k_recur<256,augru>" : (n->augru_x ? "k_augru_x" : "k_augru_h16");
            break;
        case KID_HEAD:
            s = std::string(gemm) + "(simulator_obs)" + ((n->ptab && !n->tsum) ? " + k_head_finish" : "");
            break;
        case KID_PROB: s = "k_head_prob"; break;
        case KID_GRU1:
            s = n->gru16 ? "k_gru_h16" : "k_recur<128,gru>";
            if (n->h1f[0]) s += " + k_h1_frag";
            break;
        case KID_PROJ: s = n->gemm16 ? "k_gemm_h16 / k_gemm_h16_wres(seq projections)" : "k_gemm_pk(seq projections)"; break;
    }
    snprintf(buf, (size_t)cap, "%s", s.c_str());
    return RL4RS_OK;
}
const char* rl4rs_dien_kernel_name(int which) {
    return (which >= 0 && which < KID_COUNT) ? kKernelNames[which] : "";
}
// Drains the recorded event pairs (synchronises on them) and returns the cumulative time / launch count
// of kernel class `which` since the last rl4rs_dien_profile_reset.
int rl4rs_dien_profile_read(rl4rs_dien* n, int which, double* ms_total, int64_t* launches) {
    RL4RS_REQUIRE(n && which >= 0 && which < KID_COUNT, "dien_profile_read: bad argument");
    for (auto& e : n->pending) {
        RL4RS_HIP_TRY(hipEventSynchronize(e.b));
        float ms = 0.f;
        RL4RS_HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
        n->ms_total[e.id] += ms;
        n->launches[e.id] += 1;
        n->pool.push_back(e.a);
        n->pool.push_back(e.b);
    }
    n->pending.clear();
    if (ms_total) *ms_total = n->ms_total[which];
    if (launches) *launches = n->launches[which];
    return RL4RS_OK;
}
int rl4rs_dien_profile_reset(rl4rs_dien* n) {
    RL4RS_REQUIRE(n, "dien_profile_reset: null handle");
    double d; int64_t l;
    int rc = rl4rs_dien_profile_read(n, 0, &d, &l);
    for (int i = 0; i < KID_COUNT; ++i) { n->ms_total[i] = 0; n->launches[i] = 0; }
    return rc;
}

}  // extern "C"

#include "recur_train.hpp"
#include "simnet.hpp"
