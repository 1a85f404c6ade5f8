// Supervised training of the DIEN simulator (rl4rs/nets/dien.py, script/supervised_train.py with model_type='dien') on the
// device: training-mode forward, keras binary_crossentropy, hand-written backward through the head, the category
// self-attention, the dense tower (with its Dropout), and per sequence input the DIN attention MLP, the AUGRU and the first
// GRU (explicit BPTT), Adam.  Same style as simtrain.hpp (included before this file): every x-side product and every
// parameter gradient is one GEMM / one sample-axis reduction over all (row, step) pairs; only the h-side recurrences run
// step by step (two small GEMMs + two element-wise kernels per step and direction).  Launch-bound at the reference's batch
// of 256 - this is the functional, gradient-checked form, not a tuned one.
//
// Cells (TF 1.15 GRUCell / deepctr VecAttGRUCell, as restated for the scorer in dien.hip):
//   [r, u] = sigmoid([x, h] Wg + bg);  c = tanh([x, r*h] Wc + bc);  AUGRU: u <- (1 - a_t) u;  h' = u h + (1 - u) c
// Flat parameter layout: [cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | obs_w | obs_b | out_w | out_b |
//   per sequence input i: gru_gate_w, gru_gate_b, gru_cand_w, gru_cand_b, att_w1, att_b1, att_w2, att_b2, att_w3, att_b3,
//   augru_gate_w, augru_gate_b, augru_cand_w, augru_cand_b ]
#pragma once

namespace rl4rs {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void k_tf_gates(const float* __restrict__ a1g, const float* __restrict__ g, const float* __restrict__ hprev, int64_t ldh,
                           float* __restrict__ R, float* __restrict__ Ug, float* __restrict__ RH, int N, int Hd, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, c = i - n * Hd;
    const size_t s2 = ((size_t)n * len + t) * 2 * Hd, s1 = ((size_t)n * len + t) * Hd;
    const float r = sigm(a1g[s2 + c] + g[(size_t)n * 2 * Hd + c]);
    const float u = sigm(a1g[s2 + Hd + c] + g[(size_t)n * 2 * Hd + Hd + c]);
    R[s1 + c] = r; Ug[s1 + c] = u;
    RH[s1 + c] = r * hprev[(size_t)n * ldh + c];
}

__global__ void k_tf_update(const float* __restrict__ a1c, const float* __restrict__ gc, const float* __restrict__ hprev, int64_t ldh,
                            const float* __restrict__ Ug, const float* __restrict__ att, float* __restrict__ C, float* __restrict__ Hs,
                            int N, int Hd, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, c = i - n * Hd;
    const size_t s1 = ((size_t)n * len + t) * Hd;
    const float cc = tanhf(a1c[s1 + c] + gc[i]);
    float u = Ug[s1 + c];
    if (att) u = (1.0f - att[(size_t)n * len + t]) * u;
    C[s1 + c] = cc;
    Hs[s1 + c] = u * hprev[(size_t)n * ldh + c] + (1.0f - u) * cc;
}

// BPTT step, part 1.  dh = dh_a + dh_b + upstream(last step) + upstream(every step);  writes the candidate pre-activation
// gradient, the update-gate pre-activation gradient, and -d(u') * u (whose row sum is d a_t for the AUGRU)
__global__ void k_tf_bwd_pre(const float* __restrict__ dh_a, const float* __restrict__ dh_b, const float* __restrict__ up_last,
                             int64_t ld_up, const float* __restrict__ up_all, float* __restrict__ dh,
                             const float* __restrict__ hprev, int64_t ldh, const float* __restrict__ Ug, const float* __restrict__ C,
                             const float* __restrict__ att, float* __restrict__ dAg, float* __restrict__ dAc,
                             float* __restrict__ du_neg, int N, int Hd, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, c = i - n * Hd;
    const size_t s2 = ((size_t)n * len + t) * 2 * Hd, s1 = ((size_t)n * len + t) * Hd;
    float d = (dh_a ? dh_a[i] : 0.f) + (dh_b ? dh_b[i] : 0.f);
    if (up_last) d += up_last[(size_t)n * ld_up + c];
    if (up_all) d += up_all[s1 + c];
    dh[i] = d;
    const float u = Ug[s1 + c], cc = C[s1 + c], a = att ? att[(size_t)n * len + t] : 0.f;
    const float up = (1.0f - a) * u;
    dAc[s1 + c] = d * (1.0f - up) * (1.0f - cc * cc);
    const float dup = d * (hprev[(size_t)n * ldh + c] - cc);
    dAg[s2 + Hd + c] = dup * (1.0f - a) * u * (1.0f - u);
    if (du_neg) du_neg[i] = -dup * u;
}

__global__ __launch_bounds__(256) void k_rowsum_to(const float* __restrict__ x, int N, int W, float* __restrict__ out, int len, int t) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float s = 0.f;
    for (int c = lane; c < W; c += 64) s += x[(size_t)n * W + c];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)n * len + t] = s;
}

__global__ void k_tf_bwd_mid(const float* __restrict__ dh, const float* __restrict__ d_rh, const float* __restrict__ hprev,
                             int64_t ldh, const float* __restrict__ R, const float* __restrict__ Ug, const float* __restrict__ att,
                             float* __restrict__ dAg, float* __restrict__ dh_part, int N, int Hd, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, c = i - n * Hd;
    const size_t s2 = ((size_t)n * len + t) * 2 * Hd, s1 = ((size_t)n * len + t) * Hd;
    const float r = R[s1 + c], q = d_rh[i], a = att ? att[(size_t)n * len + t] : 0.f;
    dAg[s2 + c] = q * hprev[(size_t)n * ldh + c] * r * (1.0f - r);
    dh_part[i] = dh[i] * (1.0f - a) * Ug[s1 + c] + q * r;
}

__global__ void k_shift_prev_w(const float* __restrict__ h, float* __restrict__ hprev, int N, int W, int len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * len * W) return;
    const int t = (i / W) % len;
    hprev[i] = t == 0 ? 0.f : h[i - W];
}

// DIN attention input [q, k, q - k, q * k] for every (row, step)
__global__ void k_att_inp(const float* __restrict__ q, const float* __restrict__ K, float* __restrict__ inp, int N, int len, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * len * E) return;
    const int e = i % E, nt = i / E, n = nt / len;
    const float qq = q[(size_t)n * E + e], kk = K[i];
    float* o = inp + (size_t)nt * 4 * E;
    o[e] = qq; o[E + e] = kk; o[2 * E + e] = qq - kk; o[3 * E + e] = qq * kk;
}
// its backward: dK += d1 - d2 + d3 q ;  dq[n] += sum_t (d0 + d2 + d3 k)
__global__ void k_att_inp_bwd(const float* __restrict__ dinp, const float* __restrict__ q, const float* __restrict__ K,
                              float* __restrict__ dK, float* __restrict__ dq, int N, int len, int E) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * E) return;
    const int n = i / E, e = i - n * E;
    const float qq = q[i];
    float acc = 0.f;
    for (int t = 0; t < len; ++t) {
        const size_t nt = (size_t)n * len + t;
        const float* d = dinp + nt * 4 * E;
        const float kk = K[nt * E + e];
        dK[nt * E + e] += d[E + e] - d[2 * E + e] + d[3 * E + e] * qq;
        acc += d[e] + d[2 * E + e] + d[3 * E + e] * kk;
    }
    dq[i] += acc;
}
__global__ void k_sig_bwd(float* __restrict__ d, const float* __restrict__ y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = d[i] * y[i] * (1.0f - y[i]);
}
__global__ void k_add_inplace(float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] += b[i];
}

// category self-attention (keras Attention() on [emb, emb] + GlobalAveragePooling1D, utils.py:16-25): one workgroup per row.
//   P = softmax(C C^T) (rows), pooled = mean_i (P C)_i.  BWD = false: pooled -> out[row, 0..E).
//   BWD = true: dC [Cn, E] from d pooled (+ the Flatten(C) gradient d_flat) -> dC_out[row].
template <bool BWD>
__global__ __launch_bounds__(256) void k_catatt_train(const int32_t* __restrict__ ids, int N, int Cn, int H, int E,
                                                      const float* __restrict__ table, float* __restrict__ out, int64_t ld_out,
                                                      const float* __restrict__ dpool, const float* __restrict__ dflat, int64_t ld_d,
                                                      float* __restrict__ dC_out) {
    extern __shared__ float sm[];
    float* C = sm;                       // [Cn][E]
    float* P = C + Cn * E;               // [Cn][Cn]
    float* v = P + Cn * Cn;              // [Cn]
    float* w = v + Cn;                   // [Cn]
    float* dS = w + Cn;                  // [Cn][Cn]
    const int row = blockIdx.x, tid = threadIdx.x;
    if (row >= N) return;
    for (int i = tid; i < Cn * E; i += 256) {
        const int j = i / E, e = i - j * E;
        const int id = min(max(ids[(size_t)row * Cn + j], 0), H - 1);
        C[i] = table[(size_t)id * E + e];
    }
    __syncthreads();
    for (int p = tid; p < Cn * Cn; p += 256) {
        const int i = p / Cn, j = p - i * Cn;
        float s = 0.f;
        for (int e = 0; e < E; ++e) s += C[i * E + e] * C[j * E + e];
        P[p] = s;
    }
    __syncthreads();
    if (tid < Cn) {
        float m = -3.4028235e38f;
        for (int j = 0; j < Cn; ++j) m = fmaxf(m, P[tid * Cn + j]);
        float se = 0.f;
        for (int j = 0; j < Cn; ++j) { const float x = expf(P[tid * Cn + j] - m); P[tid * Cn + j] = x; se += x; }
        for (int j = 0; j < Cn; ++j) P[tid * Cn + j] /= se;
    }
    __syncthreads();
    if (tid < Cn) {
        float s = 0.f;
        for (int i = 0; i < Cn; ++i) s += P[i * Cn + tid];
        w[tid] = s;                      // column sums of P
    }
    __syncthreads();
    if (!BWD) {
        for (int e = tid; e < E; e += 256) {
            float s = 0.f;
            for (int j = 0; j < Cn; ++j) s += w[j] * C[j * E + e];
            out[(size_t)row * ld_out + e] = s / (float)Cn;
        }
        return;
    }
    const float* g = dpool + (size_t)row * ld_d;          // d pooled; dO_i = g / Cn for every i
    if (tid < Cn) {
        float s = 0.f;
        for (int e = 0; e < E; ++e) s += g[e] * C[tid * E + e];
        v[tid] = s / (float)Cn;                           // dP[i][j] = v[j]
    }
    __syncthreads();
    for (int p = tid; p < Cn * Cn; p += 256) {
        const int i = p / Cn;
        float dot = 0.f;
        for (int k = 0; k < Cn; ++k) dot += P[i * Cn + k] * v[k];
        dS[p] = P[p] * (v[p - i * Cn] - dot);
    }
    __syncthreads();
    for (int i2 = tid; i2 < Cn * E; i2 += 256) {
        const int j = i2 / E, e = i2 - j * E;
        float s = w[j] * g[e] / (float)Cn;                // through O = P C
        for (int k = 0; k < Cn; ++k) s += (dS[j * Cn + k] + dS[k * Cn + j]) * C[k * E + e];     // through S = C C^T
        if (dflat) s += dflat[(size_t)row * ld_d + (size_t)j * E + e];
        dC_out[((size_t)row * Cn + j) * E + e] = s;
    }
}

}  // namespace rl4rs

enum { DP_CAT_EMB = 0, DP_SEQ_EMB, DP_DW1, DP_DB1, DP_DW2, DP_DB2, DP_OBS_W, DP_OBS_B, DP_OUT_W, DP_OUT_B, DP_SEQ0, DP_PER_SEQ = 14,
       DP_COUNT = DP_SEQ0 + 4 * DP_PER_SEQ };
// per-sequence slots, relative to DP_SEQ0 + s * DP_PER_SEQ
enum { DQ_GRU_GW = 0, DQ_GRU_GB, DQ_GRU_CW, DQ_GRU_CB, DQ_ATT_W1, DQ_ATT_B1, DQ_ATT_W2, DQ_ATT_B2, DQ_ATT_W3, DQ_ATT_B3,
       DQ_AUG_GW, DQ_AUG_GB, DQ_AUG_CW, DQ_AUG_CB };

struct CellSave {      // one recurrent layer of one sequence input, training mode
    float *A1g, *A1c, *R, *Ug, *C, *Hs, *RH;
    int Hd, pgw;       // hidden width, flat-parameter slot of its gate_w (gate_b, cand_w, cand_b follow)
};

struct rl4rs_dientrain {
    rl4rs_dien_cfg c;
    int64_t n_params, off[DP_COUNT], size[DP_COUNT];
    int max_batch, F;
    TrainCtx cx;
    float *params, *grad, *adam_m, *adam_v;
    // saved forward
    float *X[4], *inp[4], *hid1[4], *hid2[4], *score[4];
    CellSave gru[4], aug[4];
    float *q, *allf, *h1, *h1d, *h2, *obs, *logits, *dC;
    int32_t* ids10;
    uint8_t *mask1, *mask2;
    // backward scratch
    float *d_logits, *d_obs, *d_allf, *d_h1, *d_h2, *d_score, *d_hid2, *d_hid1, *d_inp, *dK, *dq, *dAg, *dAc, *dX, *hprev;
    float *s_G, *s_Gc, *s_dh, *s_dhp, *s_dhg, *s_drh, *s_du, *s_zero, *s_wgT, *s_wcT, *s_tmpw, *loss_rows, *lr_dummy;
    int64_t adam_t;
    std::vector<void*> owned;
};

namespace {

int cell_forward(rl4rs_dientrain* t, int N, const CellSave& cl, const float* Xin, const float* att, hipStream_t st) {
    const int E = t->c.emb_size, L = t->c.maxlen, Hd = cl.Hd;
    const float* Wg = t->params + t->off[cl.pgw];
    const float* bg = t->params + t->off[cl.pgw + 1];
    const float* Wc = t->params + t->off[cl.pgw + 2];
    const float* bc = t->params + t->off[cl.pgw + 3];
    int rc;
    if ((rc = launch_gemm_f32(Xin, E, Wg, 2 * Hd, bg, cl.A1g, 2 * Hd, N * L, 2 * Hd, E, 0, st))) return rc;
    if ((rc = launch_gemm_f32(Xin, E, Wc, Hd, bc, cl.A1c, Hd, N * L, Hd, E, 0, st))) return rc;
    const dim3 ew((N * Hd + 255) / 256), b256(256);
    for (int ts = 0; ts < L; ++ts) {
        const float* hprev = ts == 0 ? t->s_zero : cl.Hs + (size_t)(ts - 1) * Hd;
        const int64_t ldh = ts == 0 ? Hd : (int64_t)L * Hd;
        if ((rc = launch_gemm_f32(hprev, ldh, Wg + (size_t)E * 2 * Hd, 2 * Hd, nullptr, t->s_G, 2 * Hd, N, 2 * Hd, Hd, 0, st))) return rc;
        hipLaunchKernelGGL(k_tf_gates, ew, b256, 0, st, cl.A1g, t->s_G, hprev, ldh, cl.R, cl.Ug, cl.RH, N, Hd, L, ts);
        if ((rc = launch_gemm_f32(cl.RH + (size_t)ts * Hd, (int64_t)L * Hd, Wc + (size_t)E * Hd, Hd, nullptr, t->s_Gc, Hd, N, Hd, Hd, 0, st)))
            return rc;
        hipLaunchKernelGGL(k_tf_update, ew, b256, 0, st, cl.A1c, t->s_Gc, hprev, ldh, cl.Ug, att, cl.C, cl.Hs, N, Hd, L, ts);
    }
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// BPTT of one layer.  up_last [N, Hd] (row stride ld_up) = gradient of the final state, up_all [N*L, Hd] = gradient of every
// state (either may be NULL).  Accumulates dXin [N*L, E] INTO dXin_acc, writes d a_t into d_score (AUGRU: att != NULL).
int cell_backward(rl4rs_dientrain* t, int N, const CellSave& cl, const float* Xin, const float* up_last, int64_t ld_up,
                  const float* up_all, const float* att, float* d_score, float* dXin_acc, bool accumulate, hipStream_t st) {
    const int E = t->c.emb_size, L = t->c.maxlen, Hd = cl.Hd;
    const float* Wg = t->params + t->off[cl.pgw];
    const float* Wc = t->params + t->off[cl.pgw + 2];
    float* gWg = t->grad + t->off[cl.pgw];
    float* gbg = t->grad + t->off[cl.pgw + 1];
    float* gWc = t->grad + t->off[cl.pgw + 2];
    float* gbc = t->grad + t->off[cl.pgw + 3];
    int rc;
    const dim3 ew((N * Hd + 255) / 256), b256(256);
    // h-side weights transposed once: Wg[E:, :]^T -> [2Hd, Hd], Wc[E:, :]^T -> [Hd, Hd]
    hipLaunchKernelGGL(k_transpose, dim3((Hd * 2 * Hd + 255) / 256), b256, 0, st, Wg + (size_t)E * 2 * Hd, (int64_t)2 * Hd, Hd, 2 * Hd, t->s_wgT);
    hipLaunchKernelGGL(k_transpose, dim3((Hd * Hd + 255) / 256), b256, 0, st, Wc + (size_t)E * Hd, (int64_t)Hd, Hd, Hd, t->s_wcT);
    const float* dh_a = nullptr;
    const float* dh_b = nullptr;
    for (int ts = L - 1; ts >= 0; --ts) {
        const float* hprev = ts == 0 ? t->s_zero : cl.Hs + (size_t)(ts - 1) * Hd;
        const int64_t ldh = ts == 0 ? Hd : (int64_t)L * Hd;
        hipLaunchKernelGGL(k_tf_bwd_pre, ew, b256, 0, st, dh_a, dh_b, ts == L - 1 ? up_last : (const float*)nullptr, ld_up, up_all, t->s_dh,
                           hprev, ldh, cl.Ug, cl.C, att, t->dAg, t->dAc, att ? t->s_du : (float*)nullptr, N, Hd, L, ts);
        if (att) hipLaunchKernelGGL(k_rowsum_to, dim3((N + 3) / 4), b256, 0, st, t->s_du, N, Hd, d_score, L, ts);
        if ((rc = launch_gemm_f32(t->dAc + (size_t)ts * Hd, (int64_t)L * Hd, t->s_wcT, Hd, nullptr, t->s_drh, Hd, N, Hd, Hd, 0, st))) return rc;
        hipLaunchKernelGGL(k_tf_bwd_mid, ew, b256, 0, st, t->s_dh, t->s_drh, hprev, ldh, cl.R, cl.Ug, att, t->dAg, t->s_dhp, N, Hd, L, ts);
        if (ts > 0)
            if ((rc = launch_gemm_f32(t->dAg + (size_t)ts * 2 * Hd, (int64_t)L * 2 * Hd, t->s_wgT, Hd, nullptr, t->s_dhg, Hd, N, Hd, 2 * Hd, 0, st)))
                return rc;
        dh_a = t->s_dhp;
        dh_b = t->s_dhg;
    }
    const int Ns = N * L;
    hipLaunchKernelGGL(k_shift_prev_w, dim3((Ns * Hd + 255) / 256), b256, 0, st, cl.Hs, t->hprev, N, Hd, L);
    // gate_w = [x rows ; h rows] x 2Hd columns, cand_w likewise x Hd columns
    st_tn(t->cx, st, Xin, E, E, t->dAg, 2 * Hd, 2 * Hd, Ns, gWg);
    st_tn(t->cx, st, t->hprev, Hd, Hd, t->dAg, 2 * Hd, 2 * Hd, Ns, gWg + (size_t)E * 2 * Hd);
    st_cs(t->cx, st, t->dAg, 2 * Hd, 2 * Hd, Ns, gbg);
    st_tn(t->cx, st, Xin, E, E, t->dAc, Hd, Hd, Ns, gWc);
    st_tn(t->cx, st, cl.RH, Hd, Hd, t->dAc, Hd, Hd, Ns, gWc + (size_t)E * Hd);
    st_cs(t->cx, st, t->dAc, Hd, Hd, Ns, gbc);
    // gradient of the layer input: dAg Wg[:E]^T + dAc Wc[:E]^T
    float* dst = accumulate ? t->dX : dXin_acc;
    if ((rc = st_back(t->cx, st, t->dAg, 2 * Hd, 2 * Hd, Wg, 2 * Hd, E, dst, E, Ns))) return rc;
    if (accumulate) hipLaunchKernelGGL(k_add_inplace, dim3((Ns * E + 255) / 256), b256, 0, st, dXin_acc, t->dX, Ns * E);
    if ((rc = st_back(t->cx, st, t->dAc, Hd, Hd, Wc, Hd, E, t->dX, E, Ns))) return rc;
    hipLaunchKernelGGL(k_add_inplace, dim3((Ns * E + 255) / 256), b256, 0, st, dXin_acc, t->dX, Ns * E);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_dientrain_destroy(rl4rs_dientrain* t) {
    if (!t) return RL4RS_OK;
    for (void* q : t->owned) (void)hipFree(q);
    delete t;
    return RL4RS_OK;
}

int rl4rs_dientrain_create(const rl4rs_dien_cfg* c, const rl4rs_dien_weights* w, int32_t max_batch, void* stream,
                           rl4rs_dientrain** out) {
    RL4RS_REQUIRE(c && w && out && max_batch > 0, "dientrain_create: bad argument");
    RL4RS_REQUIRE(c->emb_size > 0 && c->emb_size % 2 == 0 && c->hidden_units > 0 && c->maxlen >= 1 && c->seq_num >= 1 && c->seq_num <= 4 &&
                  c->category_feature_num >= 10 && c->category_feature_num <= 32 && c->category_hash_size > 0 &&
                  c->dense_feature_num > 0 && c->class_num >= 2 && c->class_num <= 8, "dientrain: bad sizes");
    RL4RS_REQUIRE(w->cat_emb && w->seq_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->obs_w && w->obs_b &&
                  w->out_w && w->out_b, "dientrain_create: weights missing");
    for (int s = 0; s < c->seq_num; ++s)
        RL4RS_REQUIRE(w->gru_gate_w[s] && w->gru_gate_b[s] && w->gru_cand_w[s] && w->gru_cand_b[s] && w->att_w1[s] && w->att_b1[s] &&
                      w->att_w2[s] && w->att_b2[s] && w->att_w3[s] && w->att_b3[s] && w->augru_gate_w[s] && w->augru_gate_b[s] &&
                      w->augru_cand_w[s] && w->augru_cand_b[s], "dientrain_create: weights of sequence input %d missing", s);
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t E = c->emb_size, U = c->hidden_units, H = c->category_hash_size, Dn = c->dense_feature_num, K = c->class_num;
    const int64_t S = c->seq_num, Cn = c->category_feature_num, L = c->maxlen, NH2 = 2 * E;
    rl4rs_dientrain* t = new rl4rs_dientrain();
    t->c = *c;
    t->max_batch = max_batch;
    t->adam_t = 0;
    t->F = (int)(S * NH2 + U + (Cn + 1) * E);
    int64_t sizes[DP_COUNT];
    const float* src[DP_COUNT];
    for (int i = 0; i < DP_COUNT; ++i) { sizes[i] = 0; src[i] = nullptr; }
    const int64_t base_sz[DP_SEQ0] = {H * E, H * E, Dn * U, U, U * U, U, (int64_t)t->F * 256, 256, 256 * K, K};
    const float* base_src[DP_SEQ0] = {w->cat_emb, w->seq_emb, w->dense_w1, w->dense_b1, w->dense_w2, w->dense_b2, w->obs_w, w->obs_b,
                                      w->out_w, w->out_b};
    for (int i = 0; i < DP_SEQ0; ++i) { sizes[i] = base_sz[i]; src[i] = base_src[i]; }
    for (int s = 0; s < S; ++s) {
        const int b = DP_SEQ0 + s * DP_PER_SEQ;
        const int64_t sz[DP_PER_SEQ] = {2 * E * 2 * E, 2 * E, 2 * E * E, E, 4 * E * 64, 64, 64 * 16, 16, 16, 1,
                                        3 * E * 4 * E, 4 * E, 3 * E * 2 * E, 2 * E};
        const float* sp[DP_PER_SEQ] = {w->gru_gate_w[s], w->gru_gate_b[s], w->gru_cand_w[s], w->gru_cand_b[s], w->att_w1[s], w->att_b1[s],
                                       w->att_w2[s], w->att_b2[s], w->att_w3[s], w->att_b3[s], w->augru_gate_w[s], w->augru_gate_b[s],
                                       w->augru_cand_w[s], w->augru_cand_b[s]};
        for (int i = 0; i < DP_PER_SEQ; ++i) { sizes[b + i] = sz[i]; src[b + i] = sp[i]; }
    }
    int64_t o = 0;
    for (int i = 0; i < DP_COUNT; ++i) { t->off[i] = o; t->size[i] = sizes[i]; o += sizes[i]; }
    t->n_params = o;
    int rc = RL4RS_OK;
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) t->owned.push_back(*dst);
        return r;
    };
#define DT_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_dientrain_destroy(t); return rc; } } while (0)
#define DT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
        rl4rs_dientrain_destroy(t); return RL4RS_EHIP; } } while (0)
    DT_FAIL(al(&t->params, t->n_params));
    DT_FAIL(al(&t->grad, t->n_params));
    DT_FAIL(al(&t->adam_m, t->n_params));
    DT_FAIL(al(&t->adam_v, t->n_params));
    for (int i = 0; i < DP_COUNT; ++i)
        if (sizes[i]) DT_HIP(hipMemcpyAsync(t->params + t->off[i], src[i], (size_t)sizes[i] * 4, hipMemcpyHostToDevice, st));
    DT_HIP(hipMemsetAsync(t->adam_m, 0, (size_t)t->n_params * 4, st));
    DT_HIP(hipMemsetAsync(t->adam_v, 0, (size_t)t->n_params * 4, st));
    const size_t B = max_batch, Ns = B * L;
    for (int s = 0; s < S; ++s) {
        DT_FAIL(al(&t->X[s], Ns * E)); DT_FAIL(al(&t->inp[s], Ns * 4 * E)); DT_FAIL(al(&t->hid1[s], Ns * 64));
        DT_FAIL(al(&t->hid2[s], Ns * 16)); DT_FAIL(al(&t->score[s], Ns));
        CellSave* cells[2] = {&t->gru[s], &t->aug[s]};
        for (int k = 0; k < 2; ++k) {
            CellSave& cl = *cells[k];
            cl.Hd = k == 0 ? (int)E : (int)NH2;
            cl.pgw = DP_SEQ0 + s * DP_PER_SEQ + (k == 0 ? DQ_GRU_GW : DQ_AUG_GW);
            DT_FAIL(al(&cl.A1g, Ns * 2 * cl.Hd)); DT_FAIL(al(&cl.A1c, Ns * cl.Hd)); DT_FAIL(al(&cl.R, Ns * cl.Hd));
            DT_FAIL(al(&cl.Ug, Ns * cl.Hd)); DT_FAIL(al(&cl.C, Ns * cl.Hd)); DT_FAIL(al(&cl.Hs, Ns * cl.Hd)); DT_FAIL(al(&cl.RH, Ns * cl.Hd));
        }
    }
    DT_FAIL(al(&t->q, B * E)); DT_FAIL(al(&t->allf, B * t->F)); DT_FAIL(al(&t->h1, B * U)); DT_FAIL(al(&t->h1d, B * U));
    DT_FAIL(al(&t->h2, B * U)); DT_FAIL(al(&t->obs, B * 256)); DT_FAIL(al(&t->logits, B * K)); DT_FAIL(al(&t->dC, B * Cn * E));
    { float* p; DT_FAIL(al(&p, B * 10)); t->ids10 = reinterpret_cast<int32_t*>(p); }
    { float* p; DT_FAIL(al(&p, (B * U + 3) / 4 + 1)); t->mask1 = reinterpret_cast<uint8_t*>(p);
      DT_FAIL(al(&p, (B * U + 3) / 4 + 1)); t->mask2 = reinterpret_cast<uint8_t*>(p); }
    DT_FAIL(al(&t->d_logits, B * K)); DT_FAIL(al(&t->d_obs, B * 256)); DT_FAIL(al(&t->d_allf, B * t->F)); DT_FAIL(al(&t->d_h1, B * U));
    DT_FAIL(al(&t->d_h2, B * U)); DT_FAIL(al(&t->d_score, Ns)); DT_FAIL(al(&t->d_hid2, Ns * 16)); DT_FAIL(al(&t->d_hid1, Ns * 64));
    DT_FAIL(al(&t->d_inp, Ns * 4 * E)); DT_FAIL(al(&t->dK, Ns * E)); DT_FAIL(al(&t->dq, B * E)); DT_FAIL(al(&t->dAg, Ns * 2 * NH2));
    DT_FAIL(al(&t->dAc, Ns * NH2)); DT_FAIL(al(&t->dX, Ns * E)); DT_FAIL(al(&t->hprev, Ns * NH2));
    DT_FAIL(al(&t->s_G, B * 2 * NH2)); DT_FAIL(al(&t->s_Gc, B * NH2)); DT_FAIL(al(&t->s_dh, B * NH2)); DT_FAIL(al(&t->s_dhp, B * NH2));
    DT_FAIL(al(&t->s_dhg, B * NH2)); DT_FAIL(al(&t->s_drh, B * NH2)); DT_FAIL(al(&t->s_du, B * NH2)); DT_FAIL(al(&t->s_zero, B * NH2));
    DT_FAIL(al(&t->s_wgT, 2 * NH2 * NH2)); DT_FAIL(al(&t->s_wcT, NH2 * NH2)); DT_FAIL(al(&t->s_tmpw, 4)); DT_FAIL(al(&t->loss_rows, B));
    DT_FAIL(al(&t->lr_dummy, 4));
    DT_HIP(hipMemsetAsync(t->s_zero, 0, B * NH2 * 4, st));
    // reduction scratch: the largest M x Nc of any weight gradient, times the number of 512-sample chunks of N * L
    int64_t wmax = (int64_t)t->F * 256;
    if (Dn * U > wmax) wmax = Dn * U;
    if (3 * E * 4 * E > wmax) wmax = 3 * E * 4 * E;
    t->cx.chunk = 512;
    const int nz_all = (int)((Ns + 511) / 512);
    DT_FAIL(al(&t->cx.wt, wmax));
    DT_FAIL(al(&t->cx.part, (size_t)nz_all * wmax));
    DT_HIP(hipStreamSynchronize(st));
#undef DT_HIP
#undef DT_FAIL
    *out = t;
    return RL4RS_OK;
}

int rl4rs_dientrain_params(rl4rs_dientrain* t, float** params_dev, float** grad_dev, int64_t* count) {
    RL4RS_REQUIRE(t, "dientrain_params: null handle");
    if (params_dev) *params_dev = t->params;
    if (grad_dev) *grad_dev = t->grad;
    if (count) *count = t->n_params;
    return RL4RS_OK;
}

int rl4rs_dientrain_masks(rl4rs_dientrain* t, uint8_t** mask1_dev, uint8_t** mask2_dev) {
    RL4RS_REQUIRE(t && mask1_dev && mask2_dev, "dientrain_masks: null argument");
    *mask1_dev = t->mask1;
    *mask2_dev = t->mask2;
    return RL4RS_OK;
}

int rl4rs_dientrain_grad(rl4rs_dientrain* t, int32_t N, const float* dense, const int32_t* cat, const int32_t* const* seq,
                         const int32_t* labels, float dropout_rate, uint32_t seed, uint32_t step, float* loss_dev, void* stream) {
    RL4RS_REQUIRE(t && dense && cat && seq && labels && N > 0 && N <= t->max_batch, "dientrain_grad: bad argument (N=%d, max_batch=%d)", N,
                  t ? t->max_batch : -1);
    RL4RS_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "dientrain_grad: dropout_rate must be in [0, 1)");
    for (int s = 0; s < t->c.seq_num; ++s) RL4RS_REQUIRE(seq[s], "dientrain_grad: sequence input %d is NULL", s);
    hipStream_t st = (hipStream_t)stream;
    const int E = t->c.emb_size, U = t->c.hidden_units, H = t->c.category_hash_size, Dn = t->c.dense_feature_num;
    const int Cn = t->c.category_feature_num, K = t->c.class_num, S = t->c.seq_num, L = t->c.maxlen, NH2 = 2 * E, F = t->F;
    const int Ns = N * L;
    float* P = t->params;
    float* G = t->grad;
    const int64_t* o = t->off;
    const int off_d = S * NH2, off_c = S * NH2 + U, off_f = S * NH2 + U + E;
    int rc;
    auto ew = [](int n) { return dim3((n + 255) / 256); };
    const dim3 g4((N + 3) / 4), b256(256);
    const size_t sm_cat = (size_t)(Cn * E + 2 * Cn * Cn + 2 * Cn) * 4;
    // ---------------------------------------------------------------- forward
    // query = mean of the sequence-table embeddings of the last 10 category ids (dien.py:29-30, utils.py:114-115)
    RL4RS_HIP_TRY(hipMemcpy2DAsync(t->ids10, 10 * 4, cat + (Cn - 10), (size_t)Cn * 4, 10 * 4, N, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, t->ids10, N, 10, H, E, P + o[DP_SEQ_EMB], t->q, (int64_t)E, 0);
    for (int s = 0; s < S; ++s) {
        const int pb = DP_SEQ0 + s * DP_PER_SEQ;
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, seq[s], N, L, H, E, P + o[DP_SEQ_EMB], t->X[s], (int64_t)L * E, 0);
        if ((rc = cell_forward(t, N, t->gru[s], t->X[s], nullptr, st))) return rc;
        const float* Kk = t->gru[s].Hs;                                     // keys = first-GRU states [N*L, E]
        hipLaunchKernelGGL(k_att_inp, ew(Ns * E), b256, 0, st, t->q, Kk, t->inp[s], N, L, E);
        if ((rc = launch_gemm_f32(t->inp[s], 4 * E, P + o[pb + DQ_ATT_W1], 64, P + o[pb + DQ_ATT_B1], t->hid1[s], 64, Ns, 64, 4 * E, 2, st))) return rc;
        if ((rc = launch_gemm_f32(t->hid1[s], 64, P + o[pb + DQ_ATT_W2], 16, P + o[pb + DQ_ATT_B2], t->hid2[s], 16, Ns, 16, 64, 2, st))) return rc;
        if ((rc = launch_gemm_f32(t->hid2[s], 16, P + o[pb + DQ_ATT_W3], 1, P + o[pb + DQ_ATT_B3], t->score[s], 1, Ns, 1, 16, 0, st))) return rc;
        if ((rc = cell_forward(t, N, t->aug[s], Kk, t->score[s], st))) return rc;
        RL4RS_HIP_TRY(hipMemcpy2DAsync(t->allf + s * NH2, (size_t)F * 4, t->aug[s].Hs + (size_t)(L - 1) * NH2, (size_t)L * NH2 * 4,
                                       (size_t)NH2 * 4, N, hipMemcpyDeviceToDevice, st));
    }
    if ((rc = launch_gemm_f32(dense, Dn, P + o[DP_DW1], U, P + o[DP_DB1], t->h1, U, N, U, Dn, 1, st))) return rc;
    RL4RS_HIP_TRY(hipMemcpyAsync(t->h1d, t->h1, (size_t)N * U * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_dropout, ew(N * U), b256, 0, st, t->h1d, t->mask1, N * U, U, dropout_rate, seed, step, 0u);
    if ((rc = launch_gemm_f32(t->h1d, U, P + o[DP_DW2], U, P + o[DP_DB2], t->h2, U, N, U, U, 1, st))) return rc;
    RL4RS_HIP_TRY(hipMemcpyAsync(t->d_h2, t->h2, (size_t)N * U * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_dropout, ew(N * U), b256, 0, st, t->d_h2, t->mask2, N * U, U, dropout_rate, seed, step, 1u);
    RL4RS_HIP_TRY(hipMemcpy2DAsync(t->allf + off_d, (size_t)F * 4, t->d_h2, (size_t)U * 4, (size_t)U * 4, N, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_catatt_train<false>, dim3(N), b256, sm_cat, st, cat, N, Cn, H, E, P + o[DP_CAT_EMB], t->allf + off_c, (int64_t)F,
                       (const float*)nullptr, (const float*)nullptr, (int64_t)0, (float*)nullptr);
    hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, cat, N, Cn, H, E, P + o[DP_CAT_EMB], t->allf, (int64_t)F, off_f);
    if ((rc = launch_gemm_f32(t->allf, F, P + o[DP_OBS_W], 256, P + o[DP_OBS_B], t->obs, 256, N, 256, F, 1, st))) return rc;
    if ((rc = launch_gemm_f32(t->obs, 256, P + o[DP_OUT_W], K, P + o[DP_OUT_B], t->logits, K, N, K, 256, 0, st))) return rc;
    hipLaunchKernelGGL(k_bce_softmax, ew(N), b256, 0, st, t->logits, labels, N, K, t->d_logits, t->loss_rows);
    if (loss_dev) hipLaunchKernelGGL(k_mean, dim3(1), b256, 0, st, t->loss_rows, N, loss_dev);
    RL4RS_LAUNCH_CHECK();
    // ---------------------------------------------------------------- backward
    RL4RS_HIP_TRY(hipMemsetAsync(G + o[DP_CAT_EMB], 0, (size_t)H * E * 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(G + o[DP_SEQ_EMB], 0, (size_t)H * E * 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(t->dq, 0, (size_t)N * E * 4, st));
    st_tn(t->cx, st, t->obs, 256, 256, t->d_logits, K, K, N, G + o[DP_OUT_W]);
    st_cs(t->cx, st, t->d_logits, K, K, N, G + o[DP_OUT_B]);
    if ((rc = st_back(t->cx, st, t->d_logits, K, K, P + o[DP_OUT_W], K, 256, t->d_obs, 256, N))) return rc;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, t->d_obs, (int64_t)256, t->obs, (int64_t)256, (const uint8_t*)nullptr, 0.f, N * 256, 256);
    st_tn(t->cx, st, t->allf, F, F, t->d_obs, 256, 256, N, G + o[DP_OBS_W]);
    st_cs(t->cx, st, t->d_obs, 256, 256, N, G + o[DP_OBS_B]);
    if ((rc = st_back(t->cx, st, t->d_obs, 256, 256, P + o[DP_OBS_W], 256, F, t->d_allf, F, N))) return rc;
    // category branch: self-attention pooling + Flatten, then the scatter into the table gradient
    hipLaunchKernelGGL(k_catatt_train<true>, dim3(N), b256, sm_cat, st, cat, N, Cn, H, E, P + o[DP_CAT_EMB], (float*)nullptr, (int64_t)0,
                       t->d_allf + off_c, t->d_allf + off_c + E, (int64_t)F, t->dC);
    hipLaunchKernelGGL(k_emb_flatten_bwd, g4, b256, 0, st, cat, N, Cn, H, E, t->dC, (int64_t)Cn * E, G + o[DP_CAT_EMB]);
    // dense tower
    float* d_tower = t->d_allf + off_d;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, d_tower, (int64_t)F, t->h2, (int64_t)U, t->mask2, dropout_rate, N * U, U);
    st_tn(t->cx, st, t->h1d, U, U, d_tower, F, U, N, G + o[DP_DW2]);
    st_cs(t->cx, st, d_tower, F, U, N, G + o[DP_DB2]);
    if ((rc = st_back(t->cx, st, d_tower, F, U, P + o[DP_DW2], U, U, t->d_h1, U, N))) return rc;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, t->d_h1, (int64_t)U, t->h1, (int64_t)U, t->mask1, dropout_rate, N * U, U);
    st_tn(t->cx, st, dense, Dn, Dn, t->d_h1, U, U, N, G + o[DP_DW1]);
    st_cs(t->cx, st, t->d_h1, U, U, N, G + o[DP_DB1]);
    // sequence inputs
    for (int s = 0; s < S; ++s) {
        const int pb = DP_SEQ0 + s * DP_PER_SEQ;
        const float* Kk = t->gru[s].Hs;
        // AUGRU: gradient of the final state; yields d a_t and the gradient of its inputs (= the keys)
        if ((rc = cell_backward(t, N, t->aug[s], Kk, t->d_allf + s * NH2, (int64_t)F, nullptr, t->score[s], t->d_score, t->dK, false, st)))
            return rc;
        // attention MLP (LocalActivationUnit, att_hidden_units = (64, 16), sigmoid; raw score)
        st_tn(t->cx, st, t->hid2[s], 16, 16, t->d_score, 1, 1, Ns, G + o[pb + DQ_ATT_W3]);
        st_cs(t->cx, st, t->d_score, 1, 1, Ns, G + o[pb + DQ_ATT_B3]);
        if ((rc = st_back(t->cx, st, t->d_score, 1, 1, P + o[pb + DQ_ATT_W3], 1, 16, t->d_hid2, 16, Ns))) return rc;
        hipLaunchKernelGGL(k_sig_bwd, ew(Ns * 16), b256, 0, st, t->d_hid2, t->hid2[s], Ns * 16);
        st_tn(t->cx, st, t->hid1[s], 64, 64, t->d_hid2, 16, 16, Ns, G + o[pb + DQ_ATT_W2]);
        st_cs(t->cx, st, t->d_hid2, 16, 16, Ns, G + o[pb + DQ_ATT_B2]);
        if ((rc = st_back(t->cx, st, t->d_hid2, 16, 16, P + o[pb + DQ_ATT_W2], 16, 64, t->d_hid1, 64, Ns))) return rc;
        hipLaunchKernelGGL(k_sig_bwd, ew(Ns * 64), b256, 0, st, t->d_hid1, t->hid1[s], Ns * 64);
        st_tn(t->cx, st, t->inp[s], 4 * E, 4 * E, t->d_hid1, 64, 64, Ns, G + o[pb + DQ_ATT_W1]);
        st_cs(t->cx, st, t->d_hid1, 64, 64, Ns, G + o[pb + DQ_ATT_B1]);
        if ((rc = st_back(t->cx, st, t->d_hid1, 64, 64, P + o[pb + DQ_ATT_W1], 64, 4 * E, t->d_inp, 4 * E, Ns))) return rc;
        hipLaunchKernelGGL(k_att_inp_bwd, ew(N * E), b256, 0, st, t->d_inp, t->q, Kk, t->dK, t->dq, N, L, E);
        // first GRU: every state has an upstream gradient (it is a key and an AUGRU input); its inputs are embedding rows
        if ((rc = cell_backward(t, N, t->gru[s], t->X[s], nullptr, 0, t->dK, nullptr, nullptr, t->d_inp /* scratch [N*L, E] */, false, st)))
            return rc;
        hipLaunchKernelGGL(k_emb_flatten_bwd, g4, b256, 0, st, seq[s], N, L, H, E, t->d_inp, (int64_t)L * E, G + o[DP_SEQ_EMB]);
    }
    hipLaunchKernelGGL(k_emb_mean_bwd, g4, b256, 0, st, t->ids10, N, 10, H, E, t->dq, (int64_t)E, G + o[DP_SEQ_EMB]);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_dientrain_step(rl4rs_dientrain* t, int32_t N, const float* dense, const int32_t* cat, const int32_t* const* seq,
                         const int32_t* labels, float lr, float beta1, float beta2, float eps, float dropout_rate, uint32_t seed,
                         uint32_t step, float* loss_dev, void* stream) {
    int rc = rl4rs_dientrain_grad(t, N, dense, cat, seq, labels, dropout_rate, seed, step, loss_dev, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    t->adam_t += 1;
    const double tt = (double)t->adam_t;
    const float lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, tt)) / (1.0 - pow((double)beta1, tt)));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((t->n_params + 255) / 256)), dim3(256), 0, st, t->params, t->grad, t->adam_m, t->adam_v,
                       (int)t->n_params, lr_t, beta1, beta2, eps, t->lr_dummy, 0.f);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // extern "C"
