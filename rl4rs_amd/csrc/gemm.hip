// Plain fp32 GEMM on the gfx950 matrix cores: C[M,N] = act(A[M,K] @ W[K,N] + bias).
//
// Used for the non-recurrent layers of the DIEN scorer (dense tower utils.py:48-54, simulator_obs head
// dien.py:35, the input-side projections hoisted out of the recurrences) and the policy net.
//
// v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD, MI355X_MICROARCH.md): a 32x32 output tile per
// wave, lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31].  The K order inside an 8-wide block is
// permuted (lower half-wave takes k0..k0+3, upper half k0+4..k0+7) so each lane fetches its four A
// values with ONE 16-byte LDS read; the sum over k is order-independent as long as A and B agree.
//
// Block = 256 threads = 2x2 waves, tile 128(M) x 64(N), BK = 32.  A is staged global -> registers ->
// LDS (double buffered, rows padded to 36 floats: conflict-free for ds_read_b128); W is small and
// L1/L2 resident, so B fragments come straight from global memory (128-byte coalesced per half-wave).
#include <cstring>
#include <vector>

#include "common.hpp"

namespace rl4rs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_ELU: return x > 0.f ? x : expm1f(x);
        case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case ACT_TANH: return tanhf(x);
        case ACT_RELU: return fmaxf(x, 0.f);
        default: return x;
    }
}

constexpr int GBM = 128, GBN = 64, GBK = 32, GLD = GBK + 4;

__global__ __launch_bounds__(256) void k_gemm_f32(const float* __restrict__ A, int64_t lda,
                                                  const float* __restrict__ W, int64_t ldw,
                                                  const float* __restrict__ bias, float* __restrict__ C,
                                                  int64_t ldc, int M, int N, int K, int act,
                                                  const float* __restrict__ addend, int64_t ldadd, int add_div) {
    __shared__ __attribute__((aligned(16))) float As[2][GBM][GLD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
    const int col = n0 + wn * 32 + li;
    const bool col_ok = col < N;
    const bool vec_ok = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);

    f32x16 acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

    // staging map: thread -> (row, 4-float chunk); 128 rows x 8 chunks = 1024 chunks, 4 per thread.
    // (Measured and reverted: unconditional clamped loads with the B operand prefetched one k-tile ahead in 16 registers -
    // the loads then issue in batches, but the create-time 100000 x 128 x 384 products ran 8 % SLOWER (144 vs 134 us) and
    // the training steps that use this kernel did not move; the scorer's GEMMs go through k_gemm_pk.)
    float4 stage[4];
    auto load_tile = [&](int kt) {
        for (int p = 0; p < 4; ++p) {
            int c = tid + p * 256;
            int r = c >> 3, kq = (c & 7) << 2;
            int gr = m0 + r, gk = kt * GBK + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < M) {
                const float* src = A + (size_t)gr * lda + gk;
                if (vec_ok && gk + 3 < K) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    if (gk + 0 < K) v.x = src[0];
                    if (gk + 1 < K) v.y = src[1];
                    if (gk + 2 < K) v.z = src[2];
                    if (gk + 3 < K) v.w = src[3];
                }
            }
            stage[p] = v;
        }
    };
    auto store_tile = [&](int buf) {
        for (int p = 0; p < 4; ++p) {
            int c = tid + p * 256;
            int r = c >> 3, kq = (c & 7) << 2;
            *reinterpret_cast<float4*>(&As[buf][r][kq]) = stage[p];
        }
    };

    const int nkt = (K + GBK - 1) / GBK;
    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const int arow = wm * 64 + li;
#pragma unroll
        for (int kb = 0; kb < GBK / 8; ++kb) {
            const int kk = kb * 8 + half * 4;
            float4 a0 = *reinterpret_cast<const float4*>(&As[cur][arow][kk]);
            float4 a1 = *reinterpret_cast<const float4*>(&As[cur][arow + 32][kk]);
            const int gk = kt * GBK + kk;
            float b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                b[i] = (col_ok && gk + i < K) ? W[(size_t)(gk + i) * ldw + col] : 0.f;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b[0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b[0], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b[1], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b[1], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b[2], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b[2], acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b[3], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b[3], acc1, 0, 0, 0);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1);
        __syncthreads();
    }
    // epilogue: C/D layout of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if (col_ok) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            int g0 = m0 + wm * 64 + row;
            int g1 = g0 + 32;
            // optional addend row g / add_div: a term shared by add_div consecutive rows (contirl.hpp: the observation side of a
            // first layer whose observation is repeated for add_div sampled actions)
            if (g0 < M) C[(size_t)g0 * ldc + col] = apply_act(acc0[r] + bv + (addend ? addend[(size_t)(g0 / add_div) * ldadd + col] : 0.f), act);
            if (g1 < M) C[(size_t)g1 * ldc + col] = apply_act(acc1[r] + bv + (addend ? addend[(size_t)(g1 / add_div) * ldadd + col] : 0.f), act);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Small-M forms of the same fp32 GEMM for the training paths (minibatches of 256 rows: policy / offline-RL / simulator
// training).  With 128 x 64 block tiles a 256 x 256 output is 8 workgroups on a 256-CU chip and one launch took ~32 us
// (profiles/r04a_bcq_kernel_stats.md: 37 such launches = half of a continuous-BCQ update).  Here a workgroup owns ONE 32 x 32
// output tile and its four waves split K (interleaved 8-wide k-blocks), so the same output is 64 workgroups x 4 waves and each
// wave runs K / 8 MFMAs; the partial tiles meet in LDS.  Operands come straight from global memory (they are L2-resident
// and read once per tile): A as one 16-byte load per lane per k-block (the k order inside an 8-block is permuted like
// k_gemm_f32's), B as four coalesced dwords (NN) or one 16-byte load (NT).
//   k_gemm_small  C[M,N] = act(A[M,K] W[K,N] (+ A2[M,K2] W2[K2,N]) + bias + addend[row / add_div])     two operand pairs: the
//                 first layer of an action-conditioned MLP on cat([x, a]) without materialising the concatenation
//   k_gemm_nt     C[M,Kin] = dY[M,Nout] W[Kin,Nout]^T (* [relu_of > 0])      the input gradient of a Linear layer without the
//                 transposed weight copy, with the ReLU derivative of the layer below folded into the epilogue
#ifndef RL4RS_SMALL_U
#define RL4RS_SMALL_U 4       // k-blocks of a wave's K slice whose operand loads are in flight together (2: 8.6 us, 4: see DESIGN 9)
#endif
constexpr int SMALL_U = RL4RS_SMALL_U;
__device__ __forceinline__ float4 ld4_guard(const float* __restrict__ p, bool row_ok, int left, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row_ok && left > 0) {
        if (vec && left >= 4) {
            v = *reinterpret_cast<const float4*>(p);
        } else {
            v.x = p[0];
            if (left > 1) v.y = p[1];
            if (left > 2) v.z = p[2];
            if (left > 3) v.w = p[3];
        }
    }
    return v;
}

// split-K partial tiles -> one tile; thread (wave w, lane l) finishes accumulator registers 4w .. 4w+3 of lane l
#define SMALL_REDUCE_PROLOGUE()                                                            \
    __shared__ float red[4][16][64];                                                       \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];           \
    __syncthreads();

__global__ __launch_bounds__(256) void k_gemm_small(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw, int K,
                                                    const float* __restrict__ A2, int64_t lda2, const float* __restrict__ W2, int64_t ldw2, int K2,
                                                    const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M, int N, int act,
                                                    const float* __restrict__ addend, int64_t ldadd, int add_div) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int row = m0 + li, col = n0 + li;
    const bool row_ok = row < M, col_ok = col < N;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    auto run = [&](const float* __restrict__ a, int64_t la, const float* __restrict__ w, int64_t lw, int k_len) {
        const bool vec = ((la & 3) == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
        const int nkb = (k_len + 7) / 8;
        const float* arow = a + (size_t)(row_ok ? row : 0) * la;
        for (int kb = wave; kb < nkb; kb += 4 * SMALL_U) {           // SMALL_U k-blocks per trip: all their loads in flight before the MFMAs
            float4 av[SMALL_U];
            float bv[SMALL_U][4];
#pragma unroll
            for (int u = 0; u < SMALL_U; ++u) {
                const int k = (kb + 4 * u) * 8 + half * 4;
                av[u] = ld4_guard(arow + k, row_ok, k_len - k, vec);
#pragma unroll
                for (int i = 0; i < 4; ++i) bv[u][i] = (col_ok && k + i < k_len) ? w[(size_t)(k + i) * lw + col] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < SMALL_U; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u][3], acc, 0, 0, 0);
            }
        }
    };
    run(A, lda, W, ldw, K);
    if (A2) run(A2, lda2, W2, ldw2, K2);
    SMALL_REDUCE_PROLOGUE()
    if (col_ok) {
        const float bvl = bias ? bias[col] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const int g = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (g < M) {
                float v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane] + bvl;
                if (addend) v += addend[(size_t)(g / add_div) * ldadd + col];
                C[(size_t)g * ldc + col] = apply_act(v, act);
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gemm_nt(const float* __restrict__ dY, int64_t ldy, const float* __restrict__ W, int64_t ldw,
                                                 float* __restrict__ dX, int64_t ldx, int M, int Kin, int Nout,
                                                 const float* __restrict__ relu_of, int64_t ldr) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int row = m0 + li, col = n0 + li;
    const bool row_ok = row < M, col_ok = col < Kin;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool vec_a = ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(dY) & 15) == 0);
    const bool vec_b = ((ldw & 3) == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);
    const float* arow = dY + (size_t)(row_ok ? row : 0) * ldy;
    const float* brow = W + (size_t)(col_ok ? col : 0) * ldw;
    const int nkb = (Nout + 7) / 8;
    for (int kb = wave; kb < nkb; kb += 4 * SMALL_U) {
        float4 av[SMALL_U], bv[SMALL_U];
#pragma unroll
        for (int u = 0; u < SMALL_U; ++u) {
            const int k = (kb + 4 * u) * 8 + half * 4;
            av[u] = ld4_guard(arow + k, row_ok, Nout - k, vec_a);
            bv[u] = ld4_guard(brow + k, col_ok, Nout - k, vec_b);
        }
#pragma unroll
        for (int u = 0; u < SMALL_U; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
    }
    SMALL_REDUCE_PROLOGUE()
    if (col_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const int g = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (g < M) {
                float v = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
                if (relu_of && !(relu_of[(size_t)g * ldr + col] > 0.f)) v = 0.f;
                dX[(size_t)g * ldx + col] = v;
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Large form of the same GEMM (round 4): 128 x 128 block tiles, BOTH operands staged in LDS, 2 x 2 accumulator tiles per wave.
// k_gemm_f32 fetches its B fragments from global memory right in front of their MFMAs (four dword loads per eight MFMAs, an
// L1 / L2 round trip in every k-block): 0.31 - 0.34 of the fp32 MFMA peak on the 25 600 / 409 600-row x 256 x 256 forwards of
// the continuous learners (tools/gemm_rate.py).  Here a wave owns a 64 x 64 output tile: per 8-wide k-block two ds_read_b128
// (A, the k order inside the block permuted as in k_gemm_f32) and eight conflict-free ds_read_b32 (B, staged untransposed
// [k][n], row stride 132) feed sixteen MFMAs; the next k-tile's global loads are in flight while the current one is multiplied.
// 70.6 KB of LDS per workgroup: two workgroups per CU.
constexpr int TBM = 128, TBN = 128, TBK = 32, TLDA = TBK + 4, TLDB = TBN + 4;
constexpr size_t T128_SMEM = (size_t)2 * (TBM * TLDA + TBK * TLDB) * 4;

__global__ __launch_bounds__(256, 2) void k_gemm_f32_t128(const float* __restrict__ A, int64_t lda, const float* __restrict__ W, int64_t ldw,
                                                          const float* __restrict__ bias, float* __restrict__ C, int64_t ldc, int M, int N, int K,
                                                          int act, const float* __restrict__ addend, int64_t ldadd, int add_div) {
    extern __shared__ __attribute__((aligned(16))) float t128_sm[];
    float* As = t128_sm;                                  // [2][TBM][TLDA]
    float* Bs = t128_sm + 2 * TBM * TLDA;                 // [2][TBK][TLDB]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * TBM, n0 = blockIdx.y * TBN;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // staging loads are UNCONDITIONAL 16-byte loads from clamped addresses (the launcher takes this kernel only for 16-byte
    // aligned operands with K and N multiples of 4): out-of-range k / n chunks are zeroed by a select afterwards, out-of-range
    // rows are never stored.  (Guarded loads compile to a branch per load and a wait behind each: the tile's eight requests
    // then go out one memory round trip at a time.)
    float4 sa[4], sb[4];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = tid + p * 256;
            {   // A: 128 rows x 8 four-float chunks
                const int r = c >> 3, kq = (c & 7) << 2, gr = min(m0 + r, M - 1), gk = kt * TBK + kq;
                sa[p] = *reinterpret_cast<const float4*>(A + (size_t)gr * lda + min(gk, K - 4));
            }
            {   // B: 32 k-rows x 32 four-float chunks
                const int kr = c >> 5, nq = (c & 31) << 2, gk = kt * TBK + kr, gn = n0 + nq;
                sb[p] = *reinterpret_cast<const float4*>(W + (size_t)min(gk, K - 1) * ldw + min(gn, N - 4));
            }
        }
    };
    // (the zeroing of out-of-range chunks happens HERE, behind the multiplication of the current tile: a select right behind the
    // loads makes the wave wait for them before it starts multiplying)
    auto store_tile = [&](int buf, int kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = tid + p * 256;
            const bool a_ok = kt * TBK + ((c & 7) << 2) < K;
            const bool b_ok = kt * TBK + (c >> 5) < K && n0 + ((c & 31) << 2) < N;
            // (component-wise selects: a select of the whole float4 sends the staging arrays through scratch memory)
            const float4 va = make_float4(a_ok ? sa[p].x : 0.f, a_ok ? sa[p].y : 0.f, a_ok ? sa[p].z : 0.f, a_ok ? sa[p].w : 0.f);
            const float4 vb = make_float4(b_ok ? sb[p].x : 0.f, b_ok ? sb[p].y : 0.f, b_ok ? sb[p].z : 0.f, b_ok ? sb[p].w : 0.f);
            *reinterpret_cast<float4*>(&As[(buf * TBM + (c >> 3)) * TLDA + ((c & 7) << 2)]) = va;
            *reinterpret_cast<float4*>(&Bs[(buf * TBK + (c >> 5)) * TLDB + ((c & 31) << 2)]) = vb;
        }
    };
    const int nkt = (K + TBK - 1) / TBK;
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);
        const float* a_base = As + (cur * TBM + wm * 64 + li) * TLDA + half * 4;
        const float* b_base = Bs + (cur * TBK + half * 4) * TLDB + wn * 64 + li;
        // operands of k-block kb + 1 are read while k-block kb is multiplied (two register sets; the schedule is pinned: left
        // alone the compiler reads each B pair into one register pair right in front of its four MFMAs, an LDS round trip in
        // front of every group)
        float4 a0[2], a1[2];
        float b0[2][4], b1[2][4];
        auto lds_read = [&](int s_, int kb) {
            a0[s_] = *reinterpret_cast<const float4*>(a_base + kb * 8);
            a1[s_] = *reinterpret_cast<const float4*>(a_base + 32 * TLDA + kb * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                b0[s_][i] = b_base[(kb * 8 + i) * TLDB];
                b1[s_][i] = b_base[(kb * 8 + i) * TLDB + 32];
            }
        };
        lds_read(0, 0);
#pragma unroll
        for (int kb = 0; kb < TBK / 8; ++kb) {
            const int c_ = kb & 1;
            if (kb + 1 < TBK / 8) lds_read(c_ ^ 1, kb + 1);
            const float av0[4] = {a0[c_].x, a0[c_].y, a0[c_].z, a0[c_].w}, av1[4] = {a1[c_].x, a1[c_].y, a1[c_].z, a1[c_].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[i], b0[c_][i], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0[i], b1[c_][i], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[i], b0[c_][i], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1[i], b1[c_][i], acc[1][1], 0, 0, 0);
            }
            // this k-block: its LDS reads (of the NEXT block's operands) first, then the sixteen MFMAs
            __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
        }
        if (kt + 1 < nkt) store_tile(cur ^ 1, kt + 1);
        __syncthreads();
    }
    // C (and the row-shared addend) through a buffer descriptor over this workgroup's valid rows: rows >= M fall outside the
    // descriptor and are dropped by the hardware, the row part of an address is a scalar offset - one instruction per element
    // instead of a compare, an exec mask and a 64-bit address (the lesson of k_gemm_h16_wres)
    const int rows_here = min(TBM, M - m0);
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)m0 * ldc, 0, (int)((((int64_t)rows_here - 1) * ldc + N) * 4), 0x00020000);
    const int ldc4 = (int)ldc * 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + li;
        const bool col_ok = col < N;
        const float bv = (bias && col_ok) ? bias[col] : 0.f;
        // a column beyond N gets an offset outside the descriptor: its stores are dropped like the rows beyond M
        const int c_voff = col_ok ? (4 * half * (int)ldc + col) * 4 : 0x7ffffff0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // the sixteen addend values of an accumulator tile are requested together (one load + wait + activation + store per
            // element was sixteen serialised round trips per tile: the ISA showed `s_waitcnt vmcnt(0)` behind every one)
            float ad[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                ad[r] = 0.f;
                if (addend) {
                    const int g = min(m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, M - 1);
                    ad[r] = addend[(size_t)(g / add_div) * ldadd + min(col, N - 1)];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2);           // local row without the half part
                float v = acc[i][j][r] + bv;
                if (addend) v += ad[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(v, act)), rs_c, c_voff, lr * ldc4, 0);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Same GEMM with the weight matrix PRE-PACKED into MFMA B-fragment order (done once when a model is
// loaded): Wp[ntile][kb][lane] is the float4 {W[kb*8 + (lane>>5)*4 + i][ntile*32 + (lane&31)], i=0..3},
// K zero-padded to a multiple of 8, N to a multiple of 32.  One 16-byte load feeds four MFMAs; the next
// k-block's fragment is requested before the current block's MFMAs (register ring).
// WM = 32-row tiles per wave: block tile = (64*WM) x 64, 2x2 waves.
struct f4bits_g { float x, y, z, w; };
__device__ __forceinline__ float4 gbuf_load4(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);   // result must be bit_cast (see dien.hip)
    f4bits_g f = __builtin_bit_cast(f4bits_g, v);
    return make_float4(f.x, f.y, f.z, f.w);
}

template <int WM>
__global__ __launch_bounds__(256) void k_gemm_pk(const float* __restrict__ A, int64_t lda,
                                                 const float4* __restrict__ Wp, int KB,
                                                 const float* __restrict__ bias, float* __restrict__ C,
                                                 int64_t ldc, int M, int N, int K, int act,
                                                 const float* __restrict__ addend, int64_t ldadd) {
    constexpr int BM = 64 * WM;
    constexpr int KPT = GBK / 8;          // k-blocks per k-tile
    __shared__ __attribute__((aligned(16))) float As[2][BM][GLD];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int nt = blockIdx.y * 2 + wn;
    const int NT = (N + 31) / 32;
    const int col = nt * 32 + li;
    const bool tile_ok = nt < NT;
    const bool vec_ok = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    // this wave's B-fragment stream: [KB][64 lanes] float4, addressed through a buffer descriptor so that reads
    // past the last k-block return zeros (K tail) and no load costs address VGPRs
    const float4* wtile = Wp + (size_t)(tile_ok ? nt : 0) * KB * 64;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(wtile), 0, KB * 1024, 0x00020000);
    const int vl16 = lane * 16;

    f32x16 acc[WM];
#pragma unroll
    for (int w = 0; w < WM; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[w][i] = 0.f;

    constexpr int NST = BM * 8 / 256;     // float4 chunks per thread per A tile
    float4 stage[3][NST];                 // A tiles kt+1 .. kt+3 in flight (HBM / Infinity-Cache latency >> one tile)
    auto load_tile = [&](float4 (&st)[NST], int kt) {
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            int c = tid + p * 256;
            int r = c >> 3, kq = (c & 7) << 2;
            int gr = m0 + r, gk = kt * GBK + kq;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gr < M && gk < K) {
                const float* src = A + (size_t)gr * lda + gk;
                if (vec_ok && gk + 3 < K) {
                    v = *reinterpret_cast<const float4*>(src);
                } else {
                    v.x = src[0];
                    if (gk + 1 < K) v.y = src[1];
                    if (gk + 2 < K) v.z = src[2];
                    if (gk + 3 < K) v.w = src[3];
                }
            }
            st[p] = v;
        }
    };
    auto store_tile = [&](const float4 (&st)[NST], int buf) {
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            int c = tid + p * 256;
            int r = c >> 3, kq = (c & 7) << 2;
            *reinterpret_cast<float4*>(&As[buf][r][kq]) = st[p];
        }
    };
    float4 bcur[KPT], bnext[KPT];
    auto load_b = [&](float4 (&b)[KPT], int kt) {
        const int voff = vl16 + kt * (KPT * 1024);
#pragma unroll
        for (int j = 0; j < KPT; ++j) b[j] = gbuf_load4(rs_w, voff + j * 1024, 0);
    };
    const int arow = wm * 32 * WM + li;
    auto compute = [&](const float4 (&b)[KPT], int buf) {
#pragma unroll
        for (int j = 0; j < KPT; ++j) {
            float4 av[WM];
#pragma unroll
            for (int w = 0; w < WM; ++w)
                av[w] = *reinterpret_cast<const float4*>(&As[buf][arow + 32 * w][j * 8 + half * 4]);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[w].x, b[j].x, acc[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[w].y, b[j].y, acc[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[w].z, b[j].z, acc[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[w].w, b[j].w, acc[w], 0, 0, 0);
        }
    };

    const int nkt = (K + GBK - 1) / GBK;
    // prologue: tile 0 -> LDS[0]; tiles 1..3 in flight; B fragments of tile 0
    load_tile(stage[0], 0);
    load_b(bcur, 0);
    load_tile(stage[1], 1);
    load_tile(stage[2], 2);
    store_tile(stage[0], 0);
    load_tile(stage[0], 3);
    __syncthreads();
    // steady state, unrolled by 6 so the LDS buffer (kt & 1) and the register stage ((kt + 1) % 3) are static:
    // at the top of iteration kt: LDS[kt&1] = tile kt; stage[(kt+1)%3] = tile kt+1, stage[(kt+2)%3] = tile kt+2,
    // stage[kt%3] = tile kt+3 (all possibly still in flight); bcur = B fragments of tile kt.
#define RL4RS_GEMM_STEP(KT, S1, S0, BC, BN)                                        \
    if ((KT) < nkt) {                                                              \
        load_b(BN, (KT) + 1);                                                      \
        compute(BC, (KT) & 1);                                                     \
        store_tile(stage[S1], ((KT) + 1) & 1);     /* tile kt+1 (issued >= 2 tiles ago) */ \
        load_tile(stage[S1], (KT) + 4);            /* refill the freed stage */     \
        __syncthreads();                                                           \
    }
    for (int kt = 0; kt < nkt; kt += 6) {
        RL4RS_GEMM_STEP(kt + 0, 1, 0, bcur, bnext)
        RL4RS_GEMM_STEP(kt + 1, 2, 1, bnext, bcur)
        RL4RS_GEMM_STEP(kt + 2, 0, 2, bcur, bnext)
        RL4RS_GEMM_STEP(kt + 3, 1, 0, bnext, bcur)
        RL4RS_GEMM_STEP(kt + 4, 2, 1, bcur, bnext)
        RL4RS_GEMM_STEP(kt + 5, 0, 2, bnext, bcur)
    }
#undef RL4RS_GEMM_STEP
    if (tile_ok && col < N) {
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row = m0 + wm * 32 * WM + 32 * w + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M) C[(size_t)row * ldc + col] = apply_act(acc[w][r] + bv + (addend ? addend[(size_t)row * ldadd + col] : 0.f), act);
            }
    }
}

// -------------------------------------------------------------------------------------------------
// fp16x2 form of the packed GEMM (scorer_mode FP16X2): every fp32 operand as an fp16 hi + lo pair, every product as
// A_hi*W_hi + A_lo*W_hi + A_hi*W_lo on v_mfma_f32_32x32x16_f16 with fp32 accumulation (the arithmetic of k_augru_x; 16x the
// MAC rate of v_mfma_f32_32x32x2_f32 for 3x the MACs).  The weights are split and packed once at load
// (pack_gemm_weight_h16: [ntile][kb16][hi/lo][lane][8 halfs], lane (j, kg) holds W[kb*16 + kg*8 + 0..7][nt*32 + j]); the
// activations are split while they are staged: a thread converts 8 consecutive k of one row and writes one 16-byte entry
// into the hi and the lo plane, planes in slab order [kb][k-half][row][8 halfs] so a fragment read is one conflict-free
// ds_read_b128 (slabs padded by 16 bytes: the 8 entries of one row, written by 8 neighbouring lanes, spread over the banks).
// A value outside the fp16 range becomes inf in the hi plane and -inf in the lo plane, i.e. NaN in its whole output row:
// out-of-range inputs can never come back as plausible numbers.
// Block = 4 waves, wave w owns n-tile blockIdx.y*4 + w for all BM = 32*WM rows; k-tile = 64.
typedef _Float16 ghalf8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ ghalf8_t gbuf_load_h8(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return __builtin_bit_cast(ghalf8_t, v);
}

// Optional second layer chained onto the first (the dense tower: Dense+ELU twice, utils.py:48-54): when the first layer's N
// fits one workgroup (N <= 128, a multiple of 16) its activated output tile never leaves the CU - it is split into the LDS
// planes the main loop has finished with and multiplied by the second weight matrix (K2 = N, N2 <= 128).  Same values, same
// k-blocks and the same MFMA sequence as two launches with the intermediate in HBM: bit-identical results, one launch less.
struct G16Chain {
    const char* wp2; int kb2; const float* bias2; float* c2; int64_t ldc2; int n2; int act2;
    // (unchained launches) second destination of the SAME output elements, row stride ldm: device-visible pinned HOST memory - the
    // observation of a reference-shaped step leaves for the host from the head GEMM's epilogue, as its tiles finish, instead of
    // through a device-to-host copy that can only start when the whole GEMM has ended
    float* mirror; int64_t ldm;
};

// VEC: rows of A are 16-byte aligned (decided by the host: lda % 4 == 0 and A aligned) -> two 16-byte loads per chunk
template <int WM, bool VEC>
__global__ __launch_bounds__(256) void k_gemm_h16(const float* __restrict__ A, int64_t lda,
                                                  const char* __restrict__ Wp, int KB,
                                                  const float* __restrict__ bias, float* __restrict__ C,
                                                  int64_t ldc, int M, int N, int K, int act,
                                                  const float* __restrict__ addend, int64_t ldadd, G16Chain chain) {
    constexpr int BM = 32 * WM, BK = 64, KBT = BK / 16;
    constexpr int SLAB = BM * 16 + 16, PLANE = 2 * KBT * SLAB;
    __shared__ __attribute__((aligned(16))) char As[2][2][PLANE];          // [buffer][hi / lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * BM;
    const int nt = blockIdx.y * 4 + wave;
    const int NT = (N + 31) / 32;
    const bool tile_ok = nt < NT;
    const int col = nt * 32 + li;
    const char* wtile = Wp + (size_t)(tile_ok ? nt : 0) * KB * 2048;
    const float inv_s = reinterpret_cast<const float*>(Wp)[(size_t)NT * KB * 512 + (tile_ok ? col : 0)];     // 1 / (this column's power-of-two prescale)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wtile), 0, KB * 2048, 0x00020000);
    const int vl16 = lane * 16;

    f32x16 acc[WM];
#pragma unroll
    for (int w = 0; w < WM; ++w)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[w][i] = 0.f;

    // A tiles in flight: 2 register stages of 8 consecutive k of one row per chunk.  WEIGHT fragments: a ring of NB k-tiles
    // (measured: with one tile of lookahead a 32-row launch waited ~1 us of L2 latency per k-tile - 17 us for the head's 12
    // k-tiles of 2.4 us of MFMA work; requesting the A tiles deeper changed nothing).  32-row tiles (small launches, one or
    // two workgroups per CU) keep 4 k-tiles of fragments in flight, 64-row tiles 2.
    constexpr int NS = 2, NB = WM == 1 ? 4 : 2;
    float4 stage[NS][WM][2];
    // A through a buffer descriptor over this workgroup's rows: rows >= M and everything past the last element read as
    // zero, so the loads are UNCONDITIONAL (no divergent branches: the compiler keeps every requested tile in flight and
    // waits with exact counts; guarded loads had forced a full drain of the weight ring at every k-tile); columns >= K of
    // a row (they exist when lda > K) are cleared with selects
    const int rows_here = min(BM, M - m0);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(A + (size_t)m0 * lda), 0, (int)((((int64_t)rows_here - 1) * lda + K) * 4), 0x00020000);
    auto gload = [&](float4 (&st)[WM][2], int kt) {
#pragma unroll
        for (int p = 0; p < WM; ++p) {
            const int c = tid + p * 256;
            const int r = c >> 3, gk = kt * BK + (c & 7) * 8;
            const int voff = (int)(((int64_t)r * lda + gk) * 4);
            float x[8];
            if (VEC) {
                const float4 v0 = gbuf_load4(rs_a, voff, 0), v1 = gbuf_load4(rs_a, voff + 16, 0);
                x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, voff + e * 4, 0, 0));
            }
            st[p][0] = make_float4(x[0], x[1], x[2], x[3]);        // raw: the K-tail selects wait for the data, so they run in lstore
            st[p][1] = make_float4(x[4], x[5], x[6], x[7]);
        }
    };
    auto lstore = [&](const float4 (&st)[WM][2], int buf, int kt) {
#pragma unroll
        for (int p = 0; p < WM; ++p) {
            const int c = tid + p * 256;
            const int off = (c & 7) * SLAB + (c >> 3) * 16;
            const int gk = kt * BK + (c & 7) * 8;
            float x[8] = {st[p][0].x, st[p][0].y, st[p][0].z, st[p][0].w, st[p][1].x, st[p][1].y, st[p][1].z, st[p][1].w};
            if (gk + 7 >= K) {                   // only the last k-tile of a K that is not a multiple of 8 has such a chunk
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = (gk + e < K) ? x[e] : 0.f;
            }
            ghalf8_t hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];
                hi[e] = h;
                lo[e] = (_Float16)(x[e] - (float)h);
            }
            *reinterpret_cast<ghalf8_t*>(&As[buf][0][off]) = hi;
            *reinterpret_cast<ghalf8_t*>(&As[buf][1][off]) = lo;
        }
    };
    ghalf8_t bh[NB][KBT], bl[NB][KBT];
    auto load_b = [&](ghalf8_t (&h)[KBT], ghalf8_t (&l)[KBT], int kt) {
#pragma unroll
        for (int j = 0; j < KBT; ++j) {
            h[j] = gbuf_load_h8(rs_w, vl16, (kt * KBT + j) * 2048);
            l[j] = gbuf_load_h8(rs_w, vl16 + 1024, (kt * KBT + j) * 2048);
        }
    };
    auto compute = [&](const ghalf8_t (&h)[KBT], const ghalf8_t (&l)[KBT], int buf) {
#pragma unroll
        for (int j = 0; j < KBT; ++j) {
            ghalf8_t ah[WM], al[WM];
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                const int off = (j * 2 + half) * SLAB + (w * 32 + li) * 16;
                ah[w] = *reinterpret_cast<const ghalf8_t*>(&As[buf][0][off]);
                al[w] = *reinterpret_cast<const ghalf8_t*>(&As[buf][1][off]);
            }
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[w], h[j], acc[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[w], h[j], acc[w], 0, 0, 0);
#pragma unroll
            for (int w = 0; w < WM; ++w) acc[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[w], l[j], acc[w], 0, 0, 0);
        }
    };

    const int nkt = (K + BK - 1) / BK;
#pragma unroll
    for (int i = 0; i < NS; ++i) gload(stage[i], i);
#pragma unroll
    for (int i = 0; i < NB - 1; ++i) load_b(bh[i], bl[i], i);
    lstore(stage[0], 0, 0);
    __syncthreads();
    // at the top of step kt: LDS[kt&1] = tile kt, stage[(kt+1)&1] = tile kt+1 (in flight), stage[kt&1] free,
    // bh/bl[kt%NB .. (kt+NB-2)%NB] = fragments of tiles kt .. kt+NB-2 (in flight)
    for (int kt0 = 0; kt0 < nkt; kt0 += NB) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int kt = kt0 + i;
            if (kt < nkt) {
                gload(stage[i & 1], kt + NS);           // A first: its wait (in-order counter) must not cover the newest B tile
                load_b(bh[(i + NB - 1) % NB], bl[(i + NB - 1) % NB], kt + NB - 1);
                compute(bh[i], bl[i], i & 1);
                lstore(stage[(i & 1) ^ 1], (i & 1) ^ 1, kt + 1);
                __syncthreads();
            }
        }
    }
    if (chain.wp2) {        // uniform: chained second layer (gridDim.y == 1, N <= 128, N % 16 == 0: checked by the launcher)
        const int NT2 = (chain.n2 + 31) / 32;
        const bool tile2_ok = wave < NT2;
        const float inv_s2 = reinterpret_cast<const float*>(chain.wp2)[(size_t)NT2 * chain.kb2 * 512 + (tile2_ok ? wave * 32 + li : 0)];
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(chain.wp2 + (size_t)(tile2_ok ? wave : 0) * chain.kb2 * 2048), 0, chain.kb2 * 2048, 0x00020000);
        ghalf8_t b2h[8], b2l[8];                       // K2 <= 128: every fragment of the second layer requested up front
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            b2h[j] = gbuf_load_h8(rs_w2, vl16, j * 2048);
            b2l[j] = gbuf_load_h8(rs_w2, vl16 + 1024, j * 2048);
        }
        char* p_hi = &As[0][0][0];                     // the main loop's last barrier has passed: its tiles are dead
        char* p_lo = p_hi + 16 * SLAB;                 // 16 slabs (k / 8) per plane = exactly the two tile buffers
        if (tile_ok) {
            const float bv = (bias && col < N) ? bias[col] : 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const float v = (col < N && m0 + rl < M) ? apply_act(acc[w][r] * inv_s + bv, act) : 0.f;
                    const _Float16 hi = (_Float16)v;
                    const int off = (col >> 3) * SLAB + rl * 16 + (col & 7) * 2;
                    *reinterpret_cast<_Float16*>(p_hi + off) = hi;
                    *reinterpret_cast<_Float16*>(p_lo + off) = (_Float16)(v - (float)hi);
                }
        }
        __syncthreads();
        f32x16 acc2[WM];
#pragma unroll
        for (int w = 0; w < WM; ++w)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[w][i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < chain.kb2) {
#pragma unroll
                for (int w = 0; w < WM; ++w) {
                    const int off = (j * 2 + half) * SLAB + (w * 32 + li) * 16;
                    const ghalf8_t ah = *reinterpret_cast<const ghalf8_t*>(p_hi + off);
                    const ghalf8_t al = *reinterpret_cast<const ghalf8_t*>(p_lo + off);
                    acc2[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2h[j], acc2[w], 0, 0, 0);
                    acc2[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, b2h[j], acc2[w], 0, 0, 0);
                    acc2[w] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, b2l[j], acc2[w], 0, 0, 0);
                }
            }
        }
        const int col2 = wave * 32 + li;
        if (tile2_ok && col2 < chain.n2) {
            const float bv2 = chain.bias2 ? chain.bias2[col2] : 0.f;
            const __amdgpu_buffer_rsrc_t rs_c2 = __builtin_amdgcn_make_buffer_rsrc(chain.c2 + (size_t)m0 * chain.ldc2, 0,
                                                                                   (int)((((int64_t)rows_here - 1) * chain.ldc2 + chain.n2) * 4), 0x00020000);
            const int l24 = (int)chain.ldc2 * 4, v2 = (4 * half * (int)chain.ldc2 + col2) * 4;
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc2[w][r] * inv_s2 + bv2, chain.act2)), rs_c2, v2,
                                                          (32 * w + (r & 3) + 8 * (r >> 2)) * l24, 0);
        }
        return;
    }
    if (tile_ok && col < N) {
        // C (and the addend) through buffer descriptors over this workgroup's valid rows: rows >= M are dropped (read as zero) by
        // the hardware and the row part of an address is a scalar - one instruction per element instead of a compare, an exec
        // mask and a 64-bit address (see k_gemm_h16_wres: that VALU work was comparable to the tile's MFMA time)
        const float bv = bias ? bias[col] : 0.f;
        const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(C + (size_t)m0 * ldc, 0, (int)((((int64_t)rows_here - 1) * ldc + N) * 4), 0x00020000);
        const int ldc4 = (int)ldc * 4, c_voff = (4 * half * (int)ldc + col) * 4;
        if (addend) {
            const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(addend + (size_t)m0 * ldadd), 0,
                                                                                  (int)((((int64_t)rows_here - 1) * ldadd + N) * 4), 0x00020000);
            const int ldd4 = (int)ldadd * 4, d_voff = (4 * half * (int)ldadd + col) * 4;
            float add[WM][16];
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    add[w][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, d_voff, (32 * w + (r & 3) + 8 * (r >> 2)) * ldd4, 0));
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[w][r] * inv_s + bv + add[w][r], act)), rs_c, c_voff,
                                                          (32 * w + (r & 3) + 8 * (r >> 2)) * ldc4, 0);
        } else {
#pragma unroll
            for (int w = 0; w < WM; ++w)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[w][r] * inv_s + bv + 0.f, act)), rs_c, c_voff,
                                                          (32 * w + (r & 3) + 8 * (r >> 2)) * ldc4, 0);
        }
        if (chain.mirror) {
            // the same values (recomputed from the registers: same expression, same bits) to the host block
            const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(chain.mirror + (size_t)m0 * chain.ldm, 0,
                                                                                  (int)((((int64_t)rows_here - 1) * chain.ldm + N) * 4), 0x00020000);
            const int ldm4 = (int)chain.ldm * 4, m_voff = (4 * half * (int)chain.ldm + col) * 4;
            if (addend) {
                const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(addend + (size_t)m0 * ldadd), 0,
                                                                                      (int)((((int64_t)rows_here - 1) * ldadd + N) * 4), 0x00020000);
                const int ldd4 = (int)ldadd * 4, d_voff = (4 * half * (int)ldadd + col) * 4;
#pragma unroll
                for (int w = 0; w < WM; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float ad = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_d, d_voff, (32 * w + (r & 3) + 8 * (r >> 2)) * ldd4, 0));
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[w][r] * inv_s + bv + ad, act)), rs_m, m_voff,
                                                              (32 * w + (r & 3) + 8 * (r >> 2)) * ldm4, 0);
                    }
            } else {
#pragma unroll
                for (int w = 0; w < WM; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[w][r] * inv_s + bv + 0.f, act)), rs_m, m_voff,
                                                              (32 * w + (r & 3) + 8 * (r >> 2)) * ldm4, 0);
            }
        }
    }
}

// -------------------------------------------------------------------------------------------------
// k_gemm_h16_wres: the same fp16x2 GEMM for a SHORT K and MANY rows (the scorer's cache projections: K = 128, N = 832, one call
// per encode, M = distinct histories x maxlen ~ 10^5), WEIGHTS RESIDENT.  k_gemm_h16 tiles that shape as (M / 64) x 7
// short-lived workgroups: 0.27 ms per call.  What the way here measured (rocprofv3, 113 000 x 832 x 128, one GPU box):
//   one workgroup per 32 rows and all columns, C staged in LDS and written as one contiguous piece      264 us
//     - without the C stores 254: the 377 MB output stream is NOT the cost (staging + a stream of C alone: 75 us)
//     - without the column-tile loop 75: the loop waits for weight fragments - every 32-row block pulls the whole 426 KB
//       weight set through L2 -> L1 again (1.5 GB per call)
//   the same with four workgroups per CU (fragments single-buffered, half a tile of lookahead)             187 us  (8 TB/s of L2)
//   operands trade places - a wave keeps ONE column tile's fragments for the whole launch (below)          134 us
//   rows requested two blocks ahead / one barrier per 64 rows                                              128 / 122 us
//   C through a buffer descriptor with scalar row offsets                                                    84 - 98 us
// The last step is the general lesson: per 32 x 32 tile the epilogue `if (row < M) C[row * ldc + col] = ...` is a compare, an exec
// mask, a 64-bit address and a store per element - ~400 VALU issue slots next to 24 MFMAs; with four waves on a SIMD that
// was 6 us of VALU per 64-row block against 3 us of matrix pipe.
// Form: a wave keeps its column tile's fragments (all k-blocks, both planes: 64 registers); a workgroup of 13 waves covers 416
// columns; the workgroups of a column group walk the 64-row blocks of A - staged and split into the fp16 planes once per
// (block, column group), double-buffered in LDS, the next block's rows requested before this block's MFMAs, ONE barrier per
// block.  Weight traffic: 426 KB per workgroup per launch.  Per output element the MFMA sequence is that of k_gemm_h16
// (k-blocks in order, hi*hi, lo*hi, hi*lo): bit-identical results (tests/test_gpu_gemm.py compares the two paths).
constexpr int WRES_WAVES = 13;
template <int KB>
__global__ __launch_bounds__(WRES_WAVES * 64) void k_gemm_h16_wres(const float* __restrict__ A, int64_t lda, const char* __restrict__ Wp,
                                                                   const float* __restrict__ bias, float* __restrict__ C, int64_t ldc,
                                                                   int M, int N, int act) {
    // a row block = SUB sub-blocks of 32 rows: one barrier per 64 rows (with 32 an iteration was 4.6 us for 1.1 us of MFMAs)
    constexpr int SUB = 2, BM = 32 * SUB, SLAB = 32 * 16 + 16, PLANE = 2 * KB * SLAB, K = KB * 16, CHUNKS = 32 * K / 8;
    static_assert(CHUNKS <= WRES_WAVES * 64, "one chunk per thread and sub-block");
    __shared__ __attribute__((aligned(16))) char s_pl[2][SUB][2][PLANE];    // [buffer][sub-block][hi / lo]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int NT = (N + 31) / 32;
    const int nt = blockIdx.y * WRES_WAVES + wave;
    const bool tile_ok = nt < NT;
    const int col = nt * 32 + li;
    const int nrb = (M + BM - 1) / BM;
    // this wave's column tile: all k-blocks, both planes, for the whole launch
    ghalf8_t bh[KB], bl[KB];
    {
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(Wp + (size_t)(tile_ok ? nt : 0) * KB * 2048), 0, KB * 2048, 0x00020000);
#pragma unroll
        for (int j = 0; j < KB; ++j) {
            bh[j] = gbuf_load_h8(rs_w, lane * 16, j * 2048);
            bl[j] = gbuf_load_h8(rs_w, lane * 16 + 1024, j * 2048);
        }
    }
    const bool col_ok = tile_ok && col < N;
    const float inv_s = reinterpret_cast<const float*>(Wp)[(size_t)NT * KB * 512 + (col_ok ? col : 0)];
    const float bv = (bias && col_ok) ? bias[col] : 0.f;
    const int ldc4 = (int)ldc * 4;
    const int c_voff = col_ok ? (4 * half * (int)ldc + col) * 4 : 0x7ffffff0;      // this lane's element of row 4 * half; other rows: + scalar
    // A staging: thread c < CHUNKS owns chunk c (8 consecutive k of one row) of every 32-row sub-block
    const bool stager = tid < CHUNKS;
    const int cr = tid / (K / 8), ckc = tid % (K / 8);
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)((((int64_t)M - 1) * lda + K) * 4), 0x00020000);
    auto request = [&](int rb, float4 (&v)[SUB][2]) {       // rows >= M read as zero (past the descriptor's end)
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const int64_t row = (int64_t)rb * BM + u * 32 + cr;
            const int voff = (int)((row * lda + ckc * 8) * 4);
            v[u][0] = gbuf_load4(rs_a, row < M ? voff : 0x7ffffff0, 0);
            v[u][1] = gbuf_load4(rs_a, row < M ? voff + 16 : 0x7ffffff0, 0);
        }
    };
    auto stage = [&](int buf, const float4 (&v)[SUB][2]) {
#pragma unroll
        for (int u = 0; u < SUB; ++u) {
            const float x[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
            ghalf8_t hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];
                hi[e] = h;
                lo[e] = (_Float16)(x[e] - (float)h);
            }
            const int off = ckc * SLAB + cr * 16;            // slab (kb, k-half) = chunk index within the row
            *reinterpret_cast<ghalf8_t*>(&s_pl[buf][u][0][off]) = hi;
            *reinterpret_cast<ghalf8_t*>(&s_pl[buf][u][1][off]) = lo;
        }
    };
    float4 pf[SUB][2];
#pragma unroll
    for (int u = 0; u < SUB; ++u) pf[u][0] = pf[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int G = gridDim.x;
    int rb = blockIdx.x;
    if (rb < nrb && stager) { request(rb, pf); stage(0, pf); }
    int buf = 0;
    for (; rb < nrb; rb += G, buf ^= 1) {
        const bool more = rb + G < nrb;
        if (more && stager) request(rb + G, pf);             // the next block's rows fly under this block's MFMAs
        __syncthreads();                                     // planes[buf] complete; everybody is done with planes[buf ^ 1]
        if (tile_ok) {
#pragma unroll
            for (int u = 0; u < SUB; ++u) {
                f32x16 acc;
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                for (int j = 0; j < KB; ++j) {
                    const int off = (j * 2 + half) * SLAB + li * 16;
                    const ghalf8_t ah = *reinterpret_cast<const ghalf8_t*>(&s_pl[buf][u][0][off]);
                    const ghalf8_t al = *reinterpret_cast<const ghalf8_t*>(&s_pl[buf][u][1][off]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc, 0, 0, 0);
                }
                // C through a buffer descriptor over the sub-block's valid rows: rows >= M and columns >= N (their lanes carry
                // an out-of-range offset) are dropped by the hardware, the row part of the address is a scalar - one
                // instruction per element instead of a compare, an exec mask and a 64-bit address (the epilogue's VALU was
                // the kernel's bound: ~400 instructions per wave and sub-block next to 24 MFMAs)
                const int m0 = rb * BM + u * 32;
                if (m0 < M) {
                    const int rows_here = min(32, M - m0);
                    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(
                        C + (size_t)m0 * ldc, 0, (int)((((int64_t)rows_here - 1) * ldc + N) * 4), 0x00020000);
                    if (act == ACT_NONE) {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[r] * inv_s + bv), rs_c, c_voff,
                                                                  ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
                    } else {
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, apply_act(acc[r] * inv_s + bv, act)), rs_c, c_voff,
                                                                  ((r & 3) + 8 * (r >> 2)) * ldc4, 0);
                    }
                }
            }
        }
        if (more && stager) stage(buf ^ 1, pf);
    }
}

// host: split W [K,N] into fp16 hi / lo and pack it into B-fragment order for k_gemm_h16 (K padded to 16, N to 32);
// returned as floats (NT * KB16 * 512)
// Every COLUMN j is stored multiplied by its own power of two s_j (pow2_prescale: max_k |w[k][j]| s_j in [2^13, 2^14), exact), so
// there is no weight-range condition, small weights keep a normal lo part, and an outlier costs precision in its own column
// only (the scorer's projection matrix is a concatenation of sections of very different scales).  The 1 / s_j follow the
// fragments as NT * 32 floats; the kernel's epilogue multiplies a lane's accumulators by its column's value - exact - before
// bias / addend / activation.
std::vector<float> pack_gemm_weight_h16(const float* w, int64_t ldw, int K, int N) {
    const int KB = (K + 15) / 16, NT = (N + 31) / 32;
    std::vector<float> scale((size_t)NT * 32, 1.f);
    for (int j = 0; j < N; ++j) {
        float mx = 0.f;
        for (int k = 0; k < K; ++k) mx = fmaxf(mx, fabsf(w[(size_t)k * ldw + j]));
        scale[j] = pow2_prescale(mx);
    }
    std::vector<uint16_t> out(((size_t)NT * KB * 512 + (size_t)NT * 32) * 2, 0);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int k = kb * 16 + (lane >> 5) * 8 + i, j = nt * 32 + (lane & 31);
                    if (k >= K || j >= N) continue;
                    const float v = w[(size_t)k * ldw + j] * scale[j];
                    const uint16_t hi = f32_to_f16(v);
                    const uint16_t lo = f32_to_f16(v - f16_to_f32(hi));
                    const size_t base = (((size_t)nt * KB + kb) * 2) * 512 + (size_t)lane * 8 + i;
                    out[base] = hi;
                    out[base + 512] = lo;
                }
    std::vector<float> f(out.size() / 2);
    memcpy(f.data(), out.data(), out.size() * 2);
    for (int j = 0; j < NT * 32; ++j) f[(size_t)NT * KB * 512 + j] = 1.0f / scale[j];     // trailer: Wp_f32[NT * KB * 512 + column]
    return f;
}

static int launch_gemm_h16_impl(const float* a, int64_t lda, const float* wp16, const float* bias, float* c, int64_t ldc,
                                int M, int N, int K, int act, hipStream_t st, const float* addend, int64_t ldadd, G16Chain chain);

int launch_gemm_h16(const float* a, int64_t lda, const float* wp16, const float* bias, float* c, int64_t ldc,
                    int M, int N, int K, int act, hipStream_t st, const float* addend, int64_t ldadd, float* mirror, int64_t ldm) {
    G16Chain none = {};
    none.mirror = mirror; none.ldm = ldm;
    return launch_gemm_h16_impl(a, lda, wp16, bias, c, ldc, M, N, K, act, st, addend, ldadd, none);
}

// c2 = act2(act1(a W1 + b1) W2 + b2) in one launch; N1 <= 128 and a multiple of 16, N2 <= 128 (else RL4RS_EINVAL)
int launch_gemm_h16_chain(const float* a, int64_t lda, const float* wp1, const float* bias1, int N1, int K1, int act1,
                          const float* wp2, const float* bias2, float* c2, int64_t ldc2, int N2, int act2, int M, hipStream_t st) {
    if (N1 > 128 || (N1 & 15) || N2 > 128 || N1 <= 0 || N2 <= 0) {
        set_error("gemm_h16_chain: unsupported widths %d -> %d", N1, N2);
        return RL4RS_EINVAL;
    }
    G16Chain ch = {reinterpret_cast<const char*>(wp2), (N1 + 15) / 16, bias2, c2, ldc2, N2, act2, nullptr, 0};
    return launch_gemm_h16_impl(a, lda, wp1, bias1, nullptr, 0, M, N1, K1, act1, st, nullptr, 0, ch);
}

static int device_cus() {
    static int cus[64] = {0};               // per device ordinal; a benign race: every thread computes the same value
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0)
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    return cus[dev];
}

static int launch_gemm_h16_impl(const float* a, int64_t lda, const float* wp16, const float* bias, float* c, int64_t ldc,
                                int M, int N, int K, int act, hipStream_t st, const float* addend, int64_t ldadd, G16Chain chain) {
    if (M <= 0 || N <= 0 || K <= 0) return RL4RS_OK;
    const int KB = (K + 15) / 16;
    const int ny = ((N + 31) / 32 + 3) / 4;
    const char* wp = reinterpret_cast<const char*>(wp16);
    const bool vec = ((lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(a) & 15) == 0);
#ifndef RL4RS_G16_NO_ROWS
    // short K, many rows, wide C (the cache projections): weights resident in registers, the workgroups walk the rows
    if (K == 128 && vec && !addend && !chain.wp2 && N >= 256 && M >= 8192 && (int64_t)M * lda * 4 < (int64_t)0x7fffff00) {
        const int groups = ((N + 31) / 32 + WRES_WAVES - 1) / WRES_WAVES;
        int workers = device_cus() / groups;
        if (workers < 1) workers = 1;
        hipLaunchKernelGGL((k_gemm_h16_wres<8>), dim3(workers, groups), dim3(WRES_WAVES * 64), 0, st, a, lda, wp, bias, c, ldc, M, N, act);
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
#endif
    const bool small = (int64_t)((M + 63) / 64) * ny < 512;       // small problems: 32-row tiles so that the grid covers the CUs
    const dim3 grid(small ? (M + 31) / 32 : (M + 63) / 64, ny);
#define RL4RS_G16_LAUNCH(WM_, VEC_) hipLaunchKernelGGL((k_gemm_h16<WM_, VEC_>), grid, dim3(256), 0, st, a, lda, wp, KB, bias, c, ldc, M, N, K, act, addend, ldadd, chain)
    if (small) { if (vec) RL4RS_G16_LAUNCH(1, true); else RL4RS_G16_LAUNCH(1, false); }
    else { if (vec) RL4RS_G16_LAUNCH(2, true); else RL4RS_G16_LAUNCH(2, false); }
#undef RL4RS_G16_LAUNCH
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// -------------------------------------------------------------------------------------------------
// Device-side pack_gemm_weight_h16: the continuous learners' weights change every update, so the fp16 hi / lo fragment planes of
// their no-grad forwards (k_amlp_fwd_h16) are rebuilt on the device in front of each use - same per-column power-of-two prescale,
// same layout and trailer, bit-identical to the host function.  One workgroup per 32-column tile, grid.y = matrix (<= 4 per launch).
__device__ __forceinline__ float pow2_prescale_dev(float maxabs) {
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.f;
    int e = 0;
    (void)frexpf(maxabs, &e);
    int k = 14 - e;
#ifndef RL4RS_PRESCALE_UP
    if (e <= 14 && e > -6) k = 0;
#endif
    if (k > 100) k = 100;
    if (k < -100) k = -100;
    return ldexpf(1.f, k);
}

__global__ __launch_bounds__(1024) void k_pack_h16_dev(PackH16Args a) {
    const PackH16Desc d = a.d[blockIdx.y];
    if (!d.w) return;
    const int KB = (d.K + 15) / 16, NT = (d.N + 31) / 32, nt = blockIdx.x;
    if (nt >= NT) return;
    __shared__ float s_mx[32][33];
    __shared__ float s_scale[32];
    const int tid = threadIdx.x, j = tid & 31, g = tid >> 5, col = nt * 32 + j;
    // column maxima: 32 row groups, four independent loads in flight per thread
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    if (col < d.N) {
        int k = g;
        for (; k + 96 < d.K; k += 128) {
            m0 = fmaxf(m0, fabsf(d.w[(size_t)k * d.ldw + col]));
            m1 = fmaxf(m1, fabsf(d.w[(size_t)(k + 32) * d.ldw + col]));
            m2 = fmaxf(m2, fabsf(d.w[(size_t)(k + 64) * d.ldw + col]));
            m3 = fmaxf(m3, fabsf(d.w[(size_t)(k + 96) * d.ldw + col]));
        }
        for (; k < d.K; k += 32) m0 = fmaxf(m0, fabsf(d.w[(size_t)k * d.ldw + col]));
    }
    s_mx[g][j] = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    __syncthreads();
    if (tid < 32) {
        float m = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) m = fmaxf(m, s_mx[i][tid]);
        const float s = (nt * 32 + tid < d.N) ? pow2_prescale_dev(m) : 1.f;
        s_scale[tid] = s;
        d.out[(size_t)NT * KB * 512 + nt * 32 + tid] = 1.0f / s;
    }
    __syncthreads();
    _Float16* oh = reinterpret_cast<_Float16*>(d.out);
    for (int idx = tid; idx < KB * 64; idx += 1024) {
        const int kb = idx >> 6, lane = idx & 63, jj = lane & 31, c = nt * 32 + jj;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int k = kb * 16 + (lane >> 5) * 8 + i;
            v[i] = (k < d.K && c < d.N) ? d.w[(size_t)k * d.ldw + c] : 0.f;
        }
        ghalf8_t hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float x = v[i] * s_scale[jj];
            asm volatile("" : "+v"(x));                // ONE rounded fp32 value for both parts (see plane_store in augru_x.hpp)
            const _Float16 h = (_Float16)x;
            hi[i] = h;
            lo[i] = (_Float16)(x - (float)h);
        }
        const size_t base = (((size_t)nt * KB + kb) * 2) * 512 + (size_t)lane * 8;
        *reinterpret_cast<ghalf8_t*>(oh + base) = hi;
        *reinterpret_cast<ghalf8_t*>(oh + base + 512) = lo;
    }
}

int launch_pack_h16_dev(const PackH16Desc* d, int n, hipStream_t st) {
    if (n <= 0 || n > 4) { set_error("pack_h16_dev: 1..4 matrices per launch"); return RL4RS_EINVAL; }
    PackH16Args a = {};
    int nt = 1;
    for (int i = 0; i < n; ++i) { a.d[i] = d[i]; nt = std::max(nt, (d[i].N + 31) / 32); }
    hipLaunchKernelGGL(k_pack_h16_dev, dim3(nt, n), dim3(1024), 0, st, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// -------------------------------------------------------------------------------------------------
// k_amlp_fwd_h16: the WHOLE no-grad forward of an "amlp" (contirl.hpp: relu([x | a] W1 + b1) -> relu(. W2 + b2) -> act(. W3 + b3),
// hidden 256 x 256) for many rows in ONE launch, fp16x2 arithmetic (operands as fp16 hi + lo, three v_mfma_f32_32x32x16_f16 per
// product, fp32 accumulation: the scorer's form).  What it replaces: three launches per network through HBM - the BCQ target and
// the evaluation rollout push 25 600 / 409 600 sampled rows through three such networks per update / env step, and the middle
// 256 x 256 layer alone wrote and re-read 2 x 419 MB per network at 409 600 rows (k_gemm_f32_t128: 645 us; the unfused fp16x2
// GEMM: 340 us, HBM-bound).  Here a workgroup takes 64 rows through all three layers and the activations never leave the CU:
//   * TRANSPOSED tiles like k_augru_x (A = weight fragment, B = activation fragment): a lane owns one row and 16 hidden columns
//     in runs of 4, so an activated tile goes back to the LDS planes ([k-block][k-half][row][8 halfs], the next layer's B
//     fragments are single conflict-free ds_read_b128) as packed 8-byte writes;
//   * layer 1 multiplies only the ACTION side (K = act_dim <= 64); the observation side x W1[:D] + b1 arrives as the row-shared
//     addend proj[row / rep] (one small GEMM over the distinct observations, contirl.hpp);
//   * 4 waves, wave w owns hidden columns [64w, 64w + 64) (two 32-column tiles x two 32-row tiles = 64 accumulator registers),
//     weight fragments stream from L2 through a 4-deep register ring (256 KB per workgroup for the middle layer);
//   * the head (out_dim <= 64) runs as (column tile, row tile) units on the waves, k-blocks split over the idle waves and
//     summed through LDS when there are only two units;
//   * LDS 80 KB -> two workgroups per CU, whose MFMA and epilogue phases overlap.
// Rows whose activations leave the fp16 range turn NaN (inf in the hi plane, -inf in the lo plane), as in k_gemm_h16.
typedef _Float16 ghalf4_t __attribute__((ext_vector_type(4)));
typedef _Float16 ghalf2_t __attribute__((ext_vector_type(2)));
typedef float gfloat2_t __attribute__((ext_vector_type(2)));
#ifdef RL4RS_AMLP_TRACE        // s_memtime marks of two workgroups (first, middle of the grid), per wave: tools/amlp_trace.py
__device__ unsigned long long g_amlp_trace[2 * 8 * 16];
#define RL4RS_AT(k) do { if (lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2)) \
        g_amlp_trace[((blockIdx.x == 0 ? 0 : 1) * 8 + wave) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RL4RS_AT(k) do { } while (0)
#endif
#ifndef RL4RS_AMLP_AB
#define RL4RS_AMLP_AB 0           // timing ablations (results WRONG): 1 every middle-layer fragment load hits one L1-resident line set,
#endif                            // 2 one MFMA per product instead of three, 4 no fp16 conversions in the plane stores
#ifndef RL4RS_AMLP_MT
#define RL4RS_AMLP_MT 2           // 32-row tiles per workgroup (2: 64 rows; 1: 32 rows, half the LDS but twice the weight bytes per row: measured slower, 0.76 -> 0.91 ms predict)
#endif
#ifndef RL4RS_AMLP_NW
#define RL4RS_AMLP_NW 8           // waves per workgroup (4 or 8; compile-time only: the two forms sum the head's k-blocks in different groupings)
#endif

// ReLU that lets NaN through (fmaxf(NaN, 0) = 0 would hide an out-of-range row behind the first activation)
__device__ __forceinline__ float relu_nan(float x) { return x < 0.f ? 0.f : x; }

// head activation of the fused forward: tanh through the hardware exp2 / rcp (3 ulp; libm's tanhf was a third of the kernel's
// finalisation time), everything else as apply_act
__device__ __forceinline__ float head_act_fast(float x, int act) {
    if (act == ACT_TANH) {
        const float t = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x));        // exp(-2|x|)
        const float r = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
        return x != x ? x : copysignf(r, x);
    }
    if (act == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
    if (act == ACT_RELU) return relu_nan(x);
    return x;                                          // ACT_NONE (the launcher refuses anything else)
}

// NW: waves per workgroup (4: two 32-column tiles per wave, 8: one); KBX: k-blocks of the action-side input the LDS image has
// room for (2: act_dim <= 32, 74.5 KB; 4: act_dim <= 64, 82.5 KB: one workgroup per CU).  64 rows per workgroup; two workgroups per CU
// for the learners' act_dim = 32.
template <int NW, int KBX, int MT>
__global__ __launch_bounds__(64 * NW, MT == 1 ? 3 : (KBX == 2 ? NW / 2 : NW / 4)) void k_amlp_fwd_h16(AmlpFwdH16 a) {
    static_assert((NW == 4 || NW == 8) && (MT == 1 || MT == 2) && NW <= 4 * MT, "the head splits its k-blocks over at most 4 waves per unit");
    constexpr int CT = 8 / NW, ROWS = 32 * MT, SLAB = ROWS * 16, HPLANE = 32 * SLAB, XPLANE = 2 * KBX * SLAB;
    constexpr int RING = NW == 8 ? 3 : 4;              // k-blocks of weight fragments in flight per wave (middle layer and head)
    __shared__ __attribute__((aligned(16))) char smem[2 * HPLANE + 2 * XPLANE];
    // per-column constants of the later phases: [0,256) 1 / prescale of W2's columns, [256,512) b2, [512,576) 1 / prescale of W3's
    // columns, [576,640) b3 - fetched ONCE per workgroup at its start (a global load in front of a phase cost that phase ~2K cycles
    // of exposed latency: a workgroup lives ~30K cycles and two per CU hide little)
    __shared__ __attribute__((aligned(16))) float s_c[640];
    char* h_hi = smem;
    char* h_lo = smem + HPLANE;
    char* x_hi = smem + 2 * HPLANE;
    char* x_lo = x_hi + XPLANE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int m0 = blockIdx.x * ROWS;
    const int KB1 = (a.E + 15) >> 4;
    const int vl16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w1p), 0, 8 * KB1 * 2048, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w2p), 0, 8 * 16 * 2048, 0x00020000);
    const int NT3 = (a.K3 + 31) >> 5;
    const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.w3p), 0, NT3 * 16 * 2048, 0x00020000);

    RL4RS_AT(0);
    // ---- layer-1 weight fragments of this wave's column tiles (requested first: their L2 latency hides behind the staging)
    ghalf8_t w1h[CT][KBX], w1l[CT][KBX];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
        for (int kb = 0; kb < KBX; ++kb)
            if (kb < KB1) {
                const int so = ((CT * wave + t) * KB1 + kb) * 2048;
                w1h[t][kb] = gbuf_load_h8(rs_w1, vl16, so);
                w1l[t][kb] = gbuf_load_h8(rs_w1, vl16 + 1024, so);
            }
    // ---- everything else layer 1 needs from global memory is requested NOW, in front of the staging loop: a workgroup has one
    // exposed memory round trip at its start instead of one per phase (with the loads where they are used the first form of this
    // kernel ran at a third of its MFMA time's pace even with a third of the MFMAs removed)
    float4 pj[MT][CT][4], is1[CT][4];
    {
        const float* tr1 = reinterpret_cast<const float*>(a.w1p) + (size_t)8 * KB1 * 512;
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (CT * wave + t) * 32 + 8 * q + 4 * half;
                is1[t][q] = *reinterpret_cast<const float4*>(tr1 + c0);
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = min(m0 + m * 32 + li, a.N - 1);
                    pj[m][t][q] = *reinterpret_cast<const float4*>(a.proj + (size_t)(row / a.rep) * 256 + c0);
                }
            }
    }
    {
        const float* tr2 = reinterpret_cast<const float*>(a.w2p) + (size_t)8 * 16 * 512;
        const float* tr3 = reinterpret_cast<const float*>(a.w3p) + (size_t)NT3 * 16 * 512;
        for (int c = tid; c < 640; c += 64 * NW) {
            float v;
            if (c < 256) v = tr2[c];
            else if (c < 512) v = a.b2[c - 256];
            else if (c < 576) v = (c - 512 < NT3 * 32) ? tr3[c - 512] : 1.f;
            else v = (c - 576 < a.K3) ? a.b3[c - 576] : 0.f;
            s_c[c] = v;
        }
    }
    // ---- stage the action-side input rows as fp16 hi / lo planes
    {
        const int nch = KB1 * 2;                       // chunks of 8 consecutive k per row
        for (int c = tid; c < ROWS * nch; c += 64 * NW) {
            const int r = c & (ROWS - 1), ck = c / ROWS;
            const int row = m0 + r, gk = ck * 8;
            float x[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = 0.f;
            if (row < a.N && gk < a.E) {               // E % 8 == 0 (launcher)
                const float4 v0 = *reinterpret_cast<const float4*>(a.act + (size_t)row * a.E + gk);
                const float4 v1 = *reinterpret_cast<const float4*>(a.act + (size_t)row * a.E + gk + 4);
                x[0] = v0.x; x[1] = v0.y; x[2] = v0.z; x[3] = v0.w; x[4] = v1.x; x[5] = v1.y; x[6] = v1.z; x[7] = v1.w;
            }
            ghalf8_t hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];
                hi[e] = h;
                lo[e] = (_Float16)(x[e] - (float)h);
            }
            *reinterpret_cast<ghalf8_t*>(x_hi + ck * SLAB + r * 16) = hi;
            *reinterpret_cast<ghalf8_t*>(x_lo + ck * SLAB + r * 16) = lo;
        }
    }
    RL4RS_AT(1);
    __syncthreads();
    RL4RS_AT(2);

    f32x16 acc[CT][MT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][m][i] = 0.f;
    };
    // the three products of one (weight fragment, activation fragment) pair for every tile of the wave, product terms outermost:
    // consecutive MFMAs go to different accumulators
    auto mfma_kb = [&](const ghalf8_t (&wh)[CT], const ghalf8_t (&wl)[CT], const ghalf8_t (&bh)[MT], const ghalf8_t (&bl)[MT]) {
#pragma unroll
        for (int term = 0; term < ((RL4RS_AMLP_AB & 2) ? 1 : 3); ++term)
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[t][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 1 ? wl[t] : wh[t], term == 2 ? bl[m] : bh[m], acc[t][m], 0, 0, 0);
        if (RL4RS_AMLP_AB & 2) asm volatile("" :: "v"(wl[0]), "v"(bl[0]));
    };
    // activation fragments (B operand) of k-block kb, row tile m: lane (row li, k-half `half`)
    auto bfrag = [&](const char* p_hi, const char* p_lo, int kb, ghalf8_t (&bh)[MT], ghalf8_t (&bl)[MT]) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int off = (kb * 2 + half) * SLAB + (m * 32 + li) * 16;
            bh[m] = *reinterpret_cast<const ghalf8_t*>(p_hi + off);
            bl[m] = *reinterpret_cast<const ghalf8_t*>(p_lo + off);
        }
    };
    // four consecutive hidden columns (tile nt, run q) of this lane's row (row tile m) -> the activation planes
    auto plane_store = [&](int nt, int m, int q, const float (&v)[4]) {
        ghalf4_t vh, vl;
        const int o = ((2 * nt + (q >> 1)) * 2 + (q & 1)) * SLAB + (m * 32 + li) * 16 + half * 8;
        if (RL4RS_AMLP_AB & 4) {                       // timing only: no conversions
            *reinterpret_cast<float2*>(h_hi + o) = make_float2(v[0], v[1]);
            *reinterpret_cast<float2*>(h_lo + o) = make_float2(v[2], v[3]);
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; j += 2) {               // as 2-vectors: packed conversions (v_cvt_pk_f16_f32)
            gfloat2_t x2 = {v[j], v[j + 1]};
            asm volatile("" : "+v"(x2));               // one rounded fp32 value per element for both parts
            const ghalf2_t h2 = __builtin_convertvector(x2, ghalf2_t);
            const gfloat2_t d2 = {x2[0] - (float)h2[0], x2[1] - (float)h2[1]};       // scalar subtractions: packed fp32 does not run beside MFMAs
            const ghalf2_t l2 = __builtin_convertvector(d2, ghalf2_t);
            vh[j] = h2[0]; vh[j + 1] = h2[1];
            vl[j] = l2[0]; vl[j + 1] = l2[1];
        }
        *reinterpret_cast<ghalf4_t*>(h_hi + o) = vh;
        *reinterpret_cast<ghalf4_t*>(h_lo + o) = vl;
    };
    // acc * (1 / column prescale) + addend -> ReLU -> planes.  1 / prescale is a power of two, so the fma rounds once, like a + b
    auto epilogue = [&](const float4 (&is)[CT][4], const float4* add /* [MT][CT][4] or [CT][4] */, bool per_row) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int t = 0; t < CT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 s4 = is[t][q], p4 = add[((per_row ? m * CT : 0) + t) * 4 + q];
                    const float v[4] = {relu_nan(__builtin_fmaf(acc[t][m][4 * q + 0], s4.x, p4.x)), relu_nan(__builtin_fmaf(acc[t][m][4 * q + 1], s4.y, p4.y)),
                                        relu_nan(__builtin_fmaf(acc[t][m][4 * q + 2], s4.z, p4.z)), relu_nan(__builtin_fmaf(acc[t][m][4 * q + 3], s4.w, p4.w))};
                    plane_store(CT * wave + t, m, q, v);
                }
    };

    // ---- layer 1: action side, K = E
    zero_acc();
#pragma unroll
    for (int kb = 0; kb < KBX; ++kb)
        if (kb < KB1) {
            ghalf8_t bh[MT], bl[MT], wh[CT], wl[CT];
            bfrag(x_hi, x_lo, kb, bh, bl);
#pragma unroll
            for (int t = 0; t < CT; ++t) { wh[t] = w1h[t][kb]; wl[t] = w1l[t][kb]; }
            mfma_kb(wh, wl, bh, bl);
        }
    RL4RS_AT(3);
    // the middle layer's first fragments are requested before the epilogue: ring slot s holds k-block kb with kb % RING == s
    ghalf8_t w2h[RING][CT], w2l[RING][CT];
    auto w2load = [&](int kb) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int so = ((CT * wave + t) * 16 + ((RL4RS_AMLP_AB & 1) ? 0 : kb)) * 2048;
            w2h[kb % RING][t] = gbuf_load_h8(rs_w2, vl16, so);
            w2l[kb % RING][t] = gbuf_load_h8(rs_w2, vl16 + 1024, so);
        }
    };
#pragma unroll
    for (int kb = 0; kb < RING - 1; ++kb) w2load(kb);
    epilogue(is1, &pj[0][0][0], true);
    RL4RS_AT(4);
    __syncthreads();
    RL4RS_AT(5);

    // ---- layer 2: 256 x 256
    zero_acc();
    float4 is2[CT][4], bb2[CT][4];
    auto load_c2 = [&]() {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (CT * wave + t) * 32 + 8 * q + 4 * half;
                is2[t][q] = *reinterpret_cast<const float4*>(s_c + c0);
                bb2[t][q] = *reinterpret_cast<const float4*>(s_c + 256 + c0);
            }
    };
    {
        ghalf8_t bh[2][MT], bl[2][MT];
        bfrag(h_hi, h_lo, 0, bh[0], bl[0]);
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            if (kb + RING - 1 < 16) w2load(kb + RING - 1);
            if (kb + 1 < 16) bfrag(h_hi, h_lo, kb + 1, bh[(kb + 1) & 1], bl[(kb + 1) & 1]);
            mfma_kb(w2h[kb % RING], w2l[kb % RING], bh[kb & 1], bl[kb & 1]);
        }
    }
    RL4RS_AT(6);
    // head: unit = (column tile, row tile); the KS = NW / units waves of a unit split its 16 k-blocks
    const int units = NT3 * MT;                        // 2 or 4
    const int KS = NW / units, unit = wave % units, ks = wave / units;
    const int nt3 = unit / MT, m3 = unit % MT;
    const int kb_n = 16 / KS, kb_lo = ks * kb_n;
    ghalf8_t w3h[RING], w3l[RING];
    auto w3load = [&](int i) {
        const int so = (nt3 * 16 + kb_lo + i) * 2048;
        w3h[i % RING] = gbuf_load_h8(rs_w3, vl16, so);
        w3l[i % RING] = gbuf_load_h8(rs_w3, vl16 + 1024, so);
    };
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) w3load(i);
    __syncthreads();                                   // every wave has read the layer-1 planes
    RL4RS_AT(7);
    load_c2();
    epilogue(is2, &bb2[0][0], false);
    RL4RS_AT(8);
    __syncthreads();
    RL4RS_AT(9);

    // ---- head
    f32x16 acc3;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc3[i] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < kb_n) {
            if (i + RING - 1 < kb_n) w3load(i + RING - 1);
            const int off = ((kb_lo + i) * 2 + half) * SLAB + (m3 * 32 + li) * 16;
            const ghalf8_t bh = *reinterpret_cast<const ghalf8_t*>(h_hi + off);
            const ghalf8_t bl = *reinterpret_cast<const ghalf8_t*>(h_lo + off);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[i % RING], bh, acc3, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3l[i % RING], bh, acc3, 0, 0, 0);
            acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w3h[i % RING], bl, acc3, 0, 0, 0);
        }
    }
    RL4RS_AT(10);
    // finalisation: the KS partial tiles of a unit meet in the (now dead) activation planes, [unit][ks][16 elements][64 lanes]; wave ks
    // of the unit then owns the column runs q in [ks * 4 / KS, (ks + 1) * 4 / KS) and sums their partials in k order (fixed order)
    const int q_n = 4 / KS, q_lo = ks * q_n;
    if (KS > 1) {                                      // uniform
        float* P = reinterpret_cast<float*>(smem);
        __syncthreads();                               // every head MFMA has read its fragments
        float* mine = P + (size_t)(unit * KS + ks) * 1024;
#pragma unroll
        for (int i = 0; i < 16; ++i) mine[i * 64 + lane] = acc3[i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < q_lo || q >= q_lo + q_n) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float sum = 0.f;
                for (int k2 = 0; k2 < KS; ++k2) sum += P[(size_t)(unit * KS + k2) * 1024 + (4 * q + j) * 64 + lane];
                acc3[4 * q + j] = sum;
            }
        }
    }
    {
        const int row = m0 + m3 * 32 + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < q_lo || q >= q_lo + q_n) continue;
            const int c0 = nt3 * 32 + 8 * q + 4 * half;
            const float4 s4 = *reinterpret_cast<const float4*>(s_c + 512 + c0), b4 = *reinterpret_cast<const float4*>(s_c + 576 + c0);
            const float sc[4] = {s4.x, s4.y, s4.z, s4.w}, bs[4] = {b4.x, b4.y, b4.z, b4.w};
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = head_act_fast(__builtin_fmaf(acc3[4 * q + j], sc[j], bs[j]), a.head_act);
            if (row < a.N) {
                float* dst = a.out + (size_t)row * a.K3 + c0;
                if ((a.K3 & 3) == 0 && c0 + 3 < a.K3) {
                    *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j < a.K3) dst[j] = v[j];
                }
            }
        }
    }
    RL4RS_AT(11);
}

int launch_amlp_fwd_h16(const AmlpFwdH16& a, hipStream_t st) {
    if (a.N <= 0) return RL4RS_OK;
    if (a.E <= 0 || a.E > 64 || (a.E & 7) || a.K3 <= 0 || a.K3 > 64 || a.rep <= 0 || a.head_act == ACT_ELU || (reinterpret_cast<uintptr_t>(a.act) & 15) ||
        (reinterpret_cast<uintptr_t>(a.proj) & 15) || (reinterpret_cast<uintptr_t>(a.b2) & 15)) {
        set_error("amlp_fwd_h16: unsupported shape (act_dim %d, out_dim %d) or unaligned operand", a.E, a.K3);
        return RL4RS_EINVAL;
    }
    const dim3 grid((a.N + 32 * RL4RS_AMLP_MT - 1) / (32 * RL4RS_AMLP_MT));
    if (a.E <= 32) hipLaunchKernelGGL((k_amlp_fwd_h16<RL4RS_AMLP_NW, 2, RL4RS_AMLP_MT>), grid, dim3(64 * RL4RS_AMLP_NW), 0, st, a);
    else hipLaunchKernelGGL((k_amlp_fwd_h16<RL4RS_AMLP_NW, 4, RL4RS_AMLP_MT>), grid, dim3(64 * RL4RS_AMLP_NW), 0, st, a);
    RL4RS_LAUNCH_CHECK();
#ifdef RL4RS_AMLP_TRACE
    if (const char* path = getenv("RL4RS_AMLP_TRACE_DUMP")) {
        unsigned long long h[2 * 8 * 16];
        (void)hipStreamSynchronize(st);
        if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_amlp_trace), sizeof(h)) == hipSuccess)
            if (FILE* f = fopen(path, "wb")) { fwrite(h, 1, sizeof(h), f); fclose(f); }
    }
#endif
    return RL4RS_OK;
}

// host: pack W [K,N] (leading dim ldw) into fragment order; returns floats ( NT * KB * 64 * 4 )
std::vector<float> pack_gemm_weight(const float* w, int64_t ldw, int K, int N) {
    const int KB = (K + 7) / 8, NT = (N + 31) / 32;
    std::vector<float> out((size_t)NT * KB * 256, 0.f);
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i) {
                    int k = kb * 8 + (lane >> 5) * 4 + i;
                    int j = nt * 32 + (lane & 31);
                    if (k < K && j < N) out[(((size_t)nt * KB + kb) * 64 + lane) * 4 + i] = w[(size_t)k * ldw + j];
                }
    return out;
}

int launch_gemm_packed(const float* a, int64_t lda, const float* wp, const float* bias, float* c, int64_t ldc,
                       int M, int N, int K, int act, hipStream_t st, const float* addend, int64_t ldadd) {
    if (M <= 0 || N <= 0 || K <= 0) return RL4RS_OK;
    const int KB = (K + 7) / 8;
    const int ny = ((N + 31) / 32 + 1) / 2;
    // small problems: 64-row block tiles so that the grid still covers the 256 CUs
    const bool small = ((M + 127) / 128) * ny < 512;
    if (small) {
        dim3 grid((M + 63) / 64, ny);
        hipLaunchKernelGGL(k_gemm_pk<1>, grid, dim3(256), 0, st, a, lda, reinterpret_cast<const float4*>(wp), KB, bias, c,
                           ldc, M, N, K, act, addend, ldadd);
    } else {
        dim3 grid((M + 127) / 128, ny);
        hipLaunchKernelGGL(k_gemm_pk<2>, grid, dim3(256), 0, st, a, lda, reinterpret_cast<const float4*>(wp), KB, bias, c,
                           ldc, M, N, K, act, addend, ldadd);
    }
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

constexpr unsigned SMALL_GEMM_MAX_BIG_BLOCKS = 96;

int launch_gemm_small(const float* a, int64_t lda, const float* w, int64_t ldw, int K, const float* a2, int64_t lda2, const float* w2,
                      int64_t ldw2, int K2, const float* bias, float* c, int64_t ldc, int M, int N, int act, hipStream_t st,
                      const float* addend, int64_t ldadd, int add_div) {
    if (M <= 0 || N <= 0 || K <= 0) return RL4RS_OK;
    dim3 grid((M + 31) / 32, (N + 31) / 32);
    if (grid.y > 65535u) {
        set_error("gemm: N=%d too large", N);
        return RL4RS_EINVAL;
    }
    hipLaunchKernelGGL(k_gemm_small, grid, dim3(256), 0, st, a, lda, w, ldw, K, a2, lda2, w2, ldw2, K2, bias, c, ldc, M, N, act, addend,
                       ldadd, add_div > 0 ? add_div : 1);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int launch_gemm_nt(const float* dy, int64_t ldy, const float* w, int64_t ldw, float* dx, int64_t ldx, int M, int Kin, int Nout,
                   hipStream_t st, const float* relu_of, int64_t ldr) {
    if (M <= 0 || Kin <= 0 || Nout <= 0) return RL4RS_OK;
    dim3 grid((M + 31) / 32, (Kin + 31) / 32);
    if (grid.y > 65535u) {
        set_error("gemm_nt: Kin=%d too large", Kin);
        return RL4RS_EINVAL;
    }
    hipLaunchKernelGGL(k_gemm_nt, grid, dim3(256), 0, st, dy, ldy, w, ldw, dx, ldx, M, Kin, Nout, relu_of, ldr);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int launch_gemm_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias, float* c,
                    int64_t ldc, int M, int N, int K, int act, hipStream_t st, const float* addend, int64_t ldadd, int add_div) {
    if (M <= 0 || N <= 0 || K <= 0) return RL4RS_OK;
    dim3 grid((M + GBM - 1) / GBM, (N + GBN - 1) / GBN);
    if (grid.y > 65535u) {
        set_error("gemm: N=%d too large", N);
        return RL4RS_EINVAL;
    }
    if (grid.x * grid.y < SMALL_GEMM_MAX_BIG_BLOCKS || N <= 32)       // the 128 x 64 tiling would leave most of the 256 CUs idle (or half of every tile: N <= 32)
        return launch_gemm_small(a, lda, w, ldw, K, nullptr, 0, nullptr, 0, 0, bias, c, ldc, M, N, act, st, addend, ldadd, add_div);
    if (M >= 2048 && N >= 96 && K >= 32 && (K & 3) == 0 && (N & 3) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(w)) & 15) == 0 && (int64_t)TBM * ldc * 4 < 0x7fff0000) {
        // many rows x a wide output: both operands through LDS
        int rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_gemm_f32_t128), T128_SMEM);
        if (rc) return rc;
        dim3 g2((M + TBM - 1) / TBM, (N + TBN - 1) / TBN);
        hipLaunchKernelGGL(k_gemm_f32_t128, g2, dim3(256), T128_SMEM, st, a, lda, w, ldw, bias, c, ldc, M, N, K, act, addend, ldadd,
                           add_div > 0 ? add_div : 1);
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, st, a, lda, w, ldw, bias, c, ldc, M, N, K, act, addend, ldadd, add_div > 0 ? add_div : 1);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace rl4rs

extern "C" int rl4rs_gemm_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                              float* c, int64_t ldc, int32_t M, int32_t N, int32_t K, int act, void* stream) {
    RL4RS_REQUIRE(a && w && c && M > 0 && N > 0 && K > 0, "rl4rs_gemm_f32: bad argument");
    RL4RS_REQUIRE(lda >= K && ldw >= N && ldc >= N, "rl4rs_gemm_f32: leading dimension too small");
    return rl4rs::launch_gemm_f32(a, lda, w, ldw, bias, c, ldc, M, N, K, act, (hipStream_t)stream);
}

// Packed-weight variant exposed for tests: packs W on the host, uploads, runs, frees (synchronous).
extern "C" int rl4rs_gemm_f32_packed(const float* a_dev, int64_t lda, const float* w_host, int64_t ldw,
                                     const float* bias_dev, float* c_dev, int64_t ldc, int32_t M, int32_t N,
                                     int32_t K, int act, void* stream) {
    RL4RS_REQUIRE(a_dev && w_host && c_dev && M > 0 && N > 0 && K > 0, "rl4rs_gemm_f32_packed: bad argument");
    std::vector<float> pk = rl4rs::pack_gemm_weight(w_host, ldw, K, N);
    float* d = nullptr;
    int rc = rl4rs::dev_alloc(&d, pk.size());
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    RL4RS_HIP_TRY(hipMemcpyAsync(d, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, st));
    rc = rl4rs::launch_gemm_packed(a_dev, lda, d, bias_dev, c_dev, ldc, M, N, K, act, st);
    RL4RS_HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(d);
    return rc;
}

// fp16x2 variant exposed for tests: splits + packs W on the host, uploads, runs, frees (synchronous).
extern "C" int rl4rs_gemm_h16_packed(const float* a_dev, int64_t lda, const float* w_host, int64_t ldw,
                                     const float* bias_dev, float* c_dev, int64_t ldc, int32_t M, int32_t N,
                                     int32_t K, int act, void* stream) {
    RL4RS_REQUIRE(a_dev && w_host && c_dev && M > 0 && N > 0 && K > 0, "rl4rs_gemm_h16_packed: bad argument");
    std::vector<float> pk = rl4rs::pack_gemm_weight_h16(w_host, ldw, K, N);
    float* d = nullptr;
    int rc = rl4rs::dev_alloc(&d, pk.size());
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    RL4RS_HIP_TRY(hipMemcpyAsync(d, pk.data(), pk.size() * 4, hipMemcpyHostToDevice, st));
    rc = rl4rs::launch_gemm_h16(a_dev, lda, d, bias_dev, c_dev, ldc, M, N, K, act, st);
    RL4RS_HIP_TRY(hipStreamSynchronize(st));
    (void)hipFree(d);
    return rc;
}

// test hook: pack W [K,N] on the host (pack_gemm_weight_h16) and on the device (k_pack_h16_dev) and count the 32-bit words that
// differ (0 = bit-identical planes and trailer).  Synchronous.
extern "C" int rl4rs_pack_h16_selftest(const float* w_host, int64_t ldw, int32_t K, int32_t N, int64_t* mismatches, void* stream) {
    RL4RS_REQUIRE(w_host && mismatches && K > 0 && N > 0 && ldw >= N, "rl4rs_pack_h16_selftest: bad argument");
    std::vector<float> pk = rl4rs::pack_gemm_weight_h16(w_host, ldw, K, N);
    float *dw = nullptr, *dp = nullptr;
    int rc = rl4rs::dev_alloc(&dw, (size_t)K * ldw);
    if (rc) return rc;
    if ((rc = rl4rs::dev_alloc(&dp, pk.size()))) { (void)hipFree(dw); return rc; }
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> back(pk.size());
    hipError_t e = hipMemcpyAsync(dw, w_host, (size_t)K * ldw * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemsetAsync(dp, 0, pk.size() * 4, st);
    rl4rs::PackH16Desc d = {dw, dp, (int)ldw, K, N};
    if (e == hipSuccess) rc = rl4rs::launch_pack_h16_dev(&d, 1, st);
    if (e == hipSuccess && rc == RL4RS_OK) e = hipMemcpyAsync(back.data(), dp, pk.size() * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(dw);
    (void)hipFree(dp);
    if (e != hipSuccess) { rl4rs::set_error("rl4rs_pack_h16_selftest: %s", hipGetErrorString(e)); return RL4RS_EHIP; }
    if (rc) return rc;
    int64_t bad = 0;
    for (size_t i = 0; i < pk.size(); ++i) bad += memcmp(&pk[i], &back[i], 4) != 0;
    *mismatches = bad;
    return RL4RS_OK;
}
