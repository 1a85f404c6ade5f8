// Training-mode recurrences of the DIEN simulator as PERSISTENT kernels (included by dien.hip; used by dientrain.hpp through the
// launchers declared in common.hpp): script/supervised_train.py:37-42 trains the cells of rl4rs/nets/utils.py:100-129 with
// model.fit; the step-by-step form (two small GEMMs + two element-wise kernels per step forwards, five launches per step
// backwards: 2 300 dependent launches per 256-sample training step) was bound by launch latency, not by arithmetic.
//
//   forward   k_recur<NH, true, U, 0, SAVE = true> - the inference recurrence kernel itself (fp32 MFMA, h in LDS, weight
//             fragments streamed through a register ring), additionally storing what BPTT needs per (row, step): r, u, c, h, r*h.
//   backward  k_recur_bwd<NH> below: one workgroup per 32 rows walks t = L-1 .. 0 with the state gradient in registers,
//             no grid-level synchronisation (rows are independent).  Per step
//                 d      = dh + upstream_t                                 u' = (1 - a_t) u
//                 dAc    = d (1 - u') (1 - c^2)                            dup = d (h_prev - c)
//                 dAg_u  = dup (1 - a_t) u (1 - u)                         d a_t = - sum_cols dup u        (AUGRU)
//                 d(rh)  = dAc  Wc_h^T                      [32 x NH] x [NH x NH]   on v_mfma_f32_32x32x2_f32
//                 dAg_r  = d(rh) h_prev r (1 - r)                          dhp = d u' + d(rh) r
//                 dh     = dhp + [dAg_r | dAg_u] Wg_h^T     [32 x 2NH] x [2NH x NH]
//             dAg / dAc of every step are written out: the parameter gradients and the gradient of the layer input stay the
//             sample-axis GEMM reductions over all (row, step) pairs they were (dientrain.hpp).
// Same equations and the same saved quantities as the step-by-step form; checked against float64 autograd
// (tests/test_gpu_simtrain.py).
#pragma once

namespace rl4rs {

// W [*, ld] rows k_off .. k_off + K (transpose = 0: B[k][n] = W[k_off + k][n]; 1: B[k][n] = W[k_off + n][k], N = rows, K = columns)
// -> MFMA-B fragment order of pack_frag: out[((nt * KB + kb) * 64 + lane) * 4 + i] = B[kb * 8 + (lane >> 5) * 4 + i][nt * 32 + (lane & 31)]
// (K_total, k_dst: the slice is rows [k_dst, k_dst + K) of a [K_total x N] operand's fragment buffer)
__global__ void k_pack_frag_dev(const float* __restrict__ w, int64_t ld, int k_off, int K, int N, int transpose, float* __restrict__ out,
                                int K_total, int k_dst) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= K * N) return;
    const int i = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
    const int KB = K / 8, kb = rest % KB, nt = rest / KB;
    const int k = kb * 8 + (lane >> 5) * 4 + i, n = nt * 32 + (lane & 31);
    const float v = transpose ? w[(size_t)(k_off + n) * ld + k] : w[(size_t)(k_off + k) * ld + n];
    out[(((size_t)nt * (K_total / 8) + (k_dst / 8 + kb)) * 64 + lane) * 4 + i] = v;
}

struct RecurBwdArgs {
    int n_rows, L;
    const float *R[4], *U[4], *C[4], *H[4];        // saved forward, [n_rows * L, NH] each
    const float* att[4];                           // [n_rows, L] or NULL (plain GRU)
    const float* up_last[4]; int64_t ld_up;        // gradient of the final state [n_rows, NH] (row stride ld_up) or NULL
    const float* up_all[4];                        // gradient of every state [n_rows * L, NH] or NULL
    const float* wcT[4];                           // Wc_h^T [NH x NH] in fragment order
    const float* wgT[4];                           // Wg_h^T [2NH x NH] in fragment order
    float *dr[4], *du[4], *dc[4];                  // out: gate / candidate pre-activation gradients, one row per (row, step)
    int64_t ld_g, ld_c;                            //      row strides of dr / du and of dc
    float* d_score[4];                             // out (AUGRU): d a_t [n_rows, L] or NULL
    int hard;                                      // hard_sigmoid gates (keras GRU)
};

template <int NH>
__global__ __launch_bounds__(NH * 2) void k_recur_bwd(RecurBwdArgs a) {
    constexpr int NW = NH / 32, LDA = NH + 4, LDG = 2 * NH + 4, KB1 = NH / 8, KB2 = 2 * NH / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tA = reinterpret_cast<float*>(smem);          // [32][LDA]   dAc tile (A operand of the first product)
    float* tG = tA + 32 * LDA;                           // [32][LDG]   [dAg_r | dAg_u] tile (A operand of the second product)
    float* s_att = tG + 32 * LDG;                        // [32][L + 1]
    float* s_red = s_att + 32 * (a.L + 1);               // [NW][32]    per-wave partial row sums of -dup u
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * 32, sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;
    const int col = wave * 32 + li;
    const bool aug = a.att[sq] != nullptr;
    for (int i = tid; i < 32 * L; i += NH * 2) {
        const int r = i / L, t = i - r * L;
        const int gr = min(row0 + r, a.n_rows - 1);
        s_att[r * LDT + t] = aug ? a.att[sq][(size_t)gr * L + t] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wcT[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wgT[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const int vl16 = lane * 16;
    const int so_c = wave * KB1 * 1024, so_g = wave * KB2 * 1024;      // this wave's column tile of each transposed matrix
    const float* arow = tA + li * LDA + half * 4;
    const float* grow = tG + li * LDG + half * 4;
    f32x16 dh;
#pragma unroll
    for (int r = 0; r < 16; ++r) dh[r] = 0.f;
    __syncthreads();

    for (int t = L - 1; t >= 0; --t) {
        float d[16], up[16], rg[16], hp[16], dagu[16];
        float red[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = crow(r, half);
            const int gr = min(row0 + row, a.n_rows - 1);
            const size_t si = ((size_t)gr * L + t) * NH + col;
            const float u = a.U[sq][si], c = a.C[sq][si];
            rg[r] = a.R[sq][si];
            hp[r] = t > 0 ? a.H[sq][si - NH] : 0.f;
            const float at = s_att[row * LDT + t];
            float dd = dh[r];
            if (a.up_all[sq]) dd += a.up_all[sq][si];
            if (t == L - 1 && a.up_last[sq]) dd += a.up_last[sq][(size_t)gr * a.ld_up + col];
            d[r] = dd;
            up[r] = (1.0f - at) * u;
            const float dac = dd * (1.0f - up[r]) * (1.0f - c * c);
            const float dup = dd * (hp[r] - c);
            dagu[r] = dup * (1.0f - at) * (a.hard ? ((u > 0.f && u < 1.f) ? 0.2f : 0.f) : u * (1.0f - u));
            red[r] = -dup * u;
            tA[row * LDA + col] = dac;
            if (row0 + row < a.n_rows) a.dc[sq][((size_t)gr * L + t) * a.ld_c + col] = dac;
        }
        if (aug) {
            // d a_t of a row = sum over ALL hidden columns of -dup u: 32 lanes of a half hold 32 columns of one row -> shuffle sum,
            // then one partial per (wave, row) in LDS, summed in wave order after the barrier (fixed order: deterministic)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = red[r];
#pragma unroll
                for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
                if (li == 0) s_red[wave * 32 + crow(r, half)] = v;
            }
        }
        __syncthreads();
        if (aug && tid < 32 && row0 + tid < a.n_rows && a.d_score[sq]) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += s_red[w * 32 + tid];
            a.d_score[sq][(size_t)(row0 + tid) * L + t] = v;
        }
        // ---- d(r h) = dAc Wc_h^T
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        {
            float4 w4[2], a4[2];
            w4[0] = buf_load4(rs_c, vl16, so_c);
            a4[0] = *reinterpret_cast<const float4*>(arow);
#pragma unroll 4
            for (int kb = 0; kb < KB1; ++kb) {
                const int cb = kb & 1, nb = cb ^ 1;
                if (kb + 1 < KB1) {
                    w4[nb] = buf_load4(rs_c, vl16, so_c + (kb + 1) * 1024);
                    a4[nb] = *reinterpret_cast<const float4*>(arow + (kb + 1) * 8);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].x, w4[cb].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].y, w4[cb].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].z, w4[cb].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].w, w4[cb].w, acc, 0, 0, 0);
            }
        }
        float dhp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = crow(r, half);
            const int gr = min(row0 + row, a.n_rows - 1);
            const float drh = acc[r];
            const float dagr = drh * hp[r] * (a.hard ? ((rg[r] > 0.f && rg[r] < 1.f) ? 0.2f : 0.f) : rg[r] * (1.0f - rg[r]));
            dhp[r] = d[r] * up[r] + drh * rg[r];
            tG[row * LDG + col] = dagr;
            tG[row * LDG + NH + col] = dagu[r];
            if (row0 + row < a.n_rows) {
                const size_t gi = ((size_t)gr * L + t) * a.ld_g + col;
                a.dr[sq][gi] = dagr;
                a.du[sq][gi] = dagu[r];
            }
        }
        __syncthreads();
        // ---- dh_{t-1} = dhp + [dAg_r | dAg_u] Wg_h^T
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (t > 0) {
            float4 w4[2], a4[2];
            w4[0] = buf_load4(rs_g, vl16, so_g);
            a4[0] = *reinterpret_cast<const float4*>(grow);
#pragma unroll 4
            for (int kb = 0; kb < KB2; ++kb) {
                const int cb = kb & 1, nb = cb ^ 1;
                if (kb + 1 < KB2) {
                    w4[nb] = buf_load4(rs_g, vl16, so_g + (kb + 1) * 1024);
                    a4[nb] = *reinterpret_cast<const float4*>(grow + (kb + 1) * 8);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].x, w4[cb].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].y, w4[cb].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].z, w4[cb].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[cb].w, w4[cb].w, acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = dhp[r] + acc[r];
        // (the next step's first barrier separates its tile writes from this step's reads: tA is read before this step's second
        // barrier, tG and s_red before the next step's first one)
    }
}

inline size_t recur_bwd_smem(int NH, int L) { return (size_t)(32 * (NH + 4) + 32 * (2 * NH + 4) + 32 * (L + 1) + (NH / 32) * 32) * 4; }

}  // namespace rl4rs
#include "recur8.hpp"
namespace rl4rs {

// Which tile form a training recurrence takes: 0 = by the grid it would fill (8-row workgroups while 32-row ones would occupy
// fewer than half of the CUs), 8 / 32 = pinned (rl4rs_recur_train_set_rows: tests and A/B runs).
static int g_recur_train_rows = 0;
static int recur_train_n_cu() {
    static int n_cu = 0;
    if (!n_cu) {
        hipDeviceProp_t pr;
        n_cu = (hipGetDeviceProperties(&pr, current_device()) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    }
    return n_cu;
}
static bool recur_train_small(int N, int S) {
    if (g_recur_train_rows == 8 || g_recur_train_rows == 4) return true;
    if (g_recur_train_rows == 32) return false;
    return (int64_t)((N + 31) / 32) * S * 2 <= recur_train_n_cu();
}
// ... and, among the small forms, 4 rows per workgroup (NH = 256 only: one 4-row tile per wave, half the MFMAs of a step against the
// same weight stream) while 8-row workgroups would occupy fewer than half of the CUs; rl4rs_recur_train_set_rows(4 / 8) pins it
static bool recur_train_rows4(int NH, int N, int S) {
    if (NH != 256 || g_recur_train_rows == 8) return false;
    if (g_recur_train_rows == 4) return true;
    return (int64_t)((N + 7) / 8) * S * 2 <= recur_train_n_cu();
}

// ---------------------------------------------------------------------------------------------------------------------------
// launchers (declared in common.hpp; dientrain.hpp lives in another translation unit)
int launch_pack_frag(const float* w, int64_t ld, int k_off, int K, int N, int transpose, float* out, hipStream_t st) {
    if (K % 8 || N % 32) { set_error("pack_frag: K=%d must be a multiple of 8 and N=%d of 32", K, N); return RL4RS_EINVAL; }
    return launch_pack_frag_slice(w, ld, k_off, K, N, transpose, out, K, 0, st);
}
int launch_pack_frag_slice(const float* w, int64_t ld, int k_off, int K, int N, int transpose, float* out, int K_total, int k_dst,
                           hipStream_t st) {
    if (K % 8 || N % 32 || K_total % 8 || k_dst % 8 || k_dst + K > K_total) {
        set_error("pack_frag: K=%d (of %d at %d) must be multiples of 8 and N=%d of 32", K, K_total, k_dst, N);
        return RL4RS_EINVAL;
    }
    hipLaunchKernelGGL(k_pack_frag_dev, dim3((K * N + 255) / 256), dim3(256), 0, st, w, ld, k_off, K, N, transpose, out, K_total, k_dst);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

template <int NH>
static int recur_train_fwd_t(const RecurTrainFwd& f, hipStream_t st) {
    const size_t smem = (size_t)(2 * 32 * (NH + 4) + 32 * (f.L + 1) + 32) * 4;
    {
        // the opt-in belongs to the FUNCTION (and the device), not to a call: raised to the most any admitted shape needs (L = 64), so
        // a short sequence launched first cannot leave the limit below what a longer one asks for later.  Asked on every launch:
        // raise_dyn_smem is a cached lookup keyed by (device, function) - a process-wide `static bool` here skipped the second GPU
        // of a process, whose launch then failed above the 64 KB default
        int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur<NH, true, 2, 0, true>), (size_t)(2 * 32 * (NH + 4) + 32 * 65 + 32) * 4);
        if (rca) return rca;
    }
    RecurArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = f.N; a.L = f.L; a.group = 1;
    a.xld = 3 * NH; a.xoff = 0; a.xbytes = (int64_t)f.N * f.L * 3 * NH * 4;
    a.slots = f.iota; a.slots_stride = 0;                      // row n reads "slot" n of its own input's table
    for (int s = 0; s < f.S; ++s) {
        a.xbase[s] = f.a1[s]; a.wg[s] = f.wg[s]; a.wc[s] = f.wc[s];
        a.sv_att[s] = f.att[s];
        a.sv_r[s] = f.R[s]; a.sv_u[s] = f.U[s]; a.sv_c[s] = f.C[s]; a.sv_h[s] = f.H[s]; a.sv_rh[s] = f.RH[s];
    }
    a.hard_gates = f.hard;
    for (int b = 0; b < 3; ++b) a.sv_blk[b] = f.xblk[b];
    if (recur_train_small(f.N, f.S)) {
        for (int s = 0; s < f.S; ++s)
            if (!a.sv_r[s] || !a.sv_u[s] || !a.sv_c[s] || !a.sv_h[s] || !a.sv_rh[s]) { set_error("recur_train_fwd: a saved-tensor pointer is NULL"); return RL4RS_EINVAL; }
        if constexpr (NH == 256) {
            if (recur_train_rows4(NH, f.N, f.S)) {
                int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur8_fwd<NH, 4>), recur8_fwd_smem(NH, 64, 4));
                if (rca) return rca;
                hipLaunchKernelGGL((k_recur8_fwd<NH, 4>), dim3((f.N + 3) / 4, f.S), dim3(320), recur8_fwd_smem(NH, f.L, 4), st, a);
                RL4RS_LAUNCH_CHECK();
                return RL4RS_OK;
            }
        }
        {
            int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur8_fwd<NH>), recur8_fwd_smem(NH, 64));
            if (rca) return rca;
        }
        hipLaunchKernelGGL((k_recur8_fwd<NH>), dim3((f.N + 7) / 8, f.S), dim3(320), recur8_fwd_smem(NH, f.L), st, a);
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
    hipLaunchKernelGGL((k_recur<NH, true, 2, 0, true>), dim3((f.N + 31) / 32, f.S), dim3(NH * 2), smem, st, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int launch_recur_train_fwd(const RecurTrainFwd& f, hipStream_t st) {
    if (f.S < 1 || f.S > 4 || f.L < 1 || f.L > 64 || (int64_t)f.N * f.L * 3 * f.Hd * 4 >= (int64_t)0x7fffffff) {
        set_error("recur_train_fwd: unsupported shape (N=%d, L=%d, S=%d, Hd=%d)", f.N, f.L, f.S, f.Hd);
        return RL4RS_EINVAL;
    }
    if (f.Hd == 128) return recur_train_fwd_t<128>(f, st);
    if (f.Hd == 256) return recur_train_fwd_t<256>(f, st);
    set_error("recur_train_fwd: hidden width %d (the persistent training kernels are built for 128 and 256)", f.Hd);
    return RL4RS_EINVAL;
}

template <int NH>
static int recur_train_bwd_t(const RecurTrainBwd& b, hipStream_t st) {
    const size_t smem = recur_bwd_smem(NH, b.L);
    {                     // per (device, function), for the longest admitted sequence (see recur_train_fwd_t)
        int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur_bwd<NH>), recur_bwd_smem(NH, 64));
        if (rca) return rca;
    }
    RecurBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = b.N; a.L = b.L; a.ld_up = b.ld_up;
    for (int s = 0; s < b.S; ++s) {
        a.R[s] = b.R[s]; a.U[s] = b.U[s]; a.C[s] = b.C[s]; a.H[s] = b.H[s]; a.att[s] = b.att[s];
        a.up_last[s] = b.up_last[s]; a.up_all[s] = b.up_all[s]; a.wcT[s] = b.wcT[s]; a.wgT[s] = b.wgT[s];
        a.dr[s] = b.dr[s]; a.du[s] = b.du[s]; a.dc[s] = b.dc[s]; a.d_score[s] = b.d_score[s];
    }
    a.ld_g = b.ld_g; a.ld_c = b.ld_c; a.hard = b.hard;
    if (recur_train_small(b.N, b.S)) {
        if constexpr (NH == 256) {
            if (recur_train_rows4(NH, b.N, b.S)) {
                int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur8_bwd<NH, 4>), recur8_bwd_smem(NH, 64, 4));
                if (rca) return rca;
                hipLaunchKernelGGL((k_recur8_bwd<NH, 4>), dim3((b.N + 3) / 4, b.S), dim3(320), recur8_bwd_smem(NH, b.L, 4), st, a);
                RL4RS_LAUNCH_CHECK();
                return RL4RS_OK;
            }
        }
        {
            int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_recur8_bwd<NH>), recur8_bwd_smem(NH, 64));
            if (rca) return rca;
        }
        hipLaunchKernelGGL((k_recur8_bwd<NH>), dim3((b.N + 7) / 8, b.S), dim3(320), recur8_bwd_smem(NH, b.L), st, a);
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
    hipLaunchKernelGGL((k_recur_bwd<NH>), dim3((b.N + 31) / 32, b.S), dim3(NH * 2), smem, st, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace rl4rs

// include/rl4rs_hip.h: pin the row-tile form of the persistent training recurrences (0 automatic, 8, 32); returns the previous value
extern "C" int rl4rs_recur_train_set_rows(int32_t rows) {
    if (rows != 0 && rows != 4 && rows != 8 && rows != 32) { rl4rs::set_error("rl4rs_recur_train_set_rows: %d (0 = automatic, 4, 8, 32)", rows); return RL4RS_EINVAL; }
    rl4rs::g_recur_train_rows = rows;
    return RL4RS_OK;
}

namespace rl4rs {

int launch_recur_train_bwd(const RecurTrainBwd& b, hipStream_t st) {
    if (b.S < 1 || b.S > 4 || b.L < 1 || b.L > 64) {
        set_error("recur_train_bwd: unsupported shape (N=%d, L=%d, S=%d, Hd=%d)", b.N, b.L, b.S, b.Hd);
        return RL4RS_EINVAL;
    }
    if (b.Hd == 128) return recur_train_bwd_t<128>(b, st);
    if (b.Hd == 256) return recur_train_bwd_t<256>(b, st);
    set_error("recur_train_bwd: hidden width %d (the persistent training kernels are built for 128 and 256)", b.Hd);
    return RL4RS_EINVAL;
}

}  // namespace rl4rs
