// Supervised training of the DIEN simulator (rl4rs/nets/dien.py, script/supervised_train.py with model_type='dien') on the
// device: training-mode forward, keras binary_crossentropy, hand-written backward through the head, the category
// self-attention, the dense tower (with its Dropout), and per sequence input the DIN attention MLP, the AUGRU and the first
// GRU (explicit BPTT), Adam.  Same style as simtrain.hpp (included before this file): every x-side product and every
// parameter gradient is one GEMM / one sample-axis reduction over all (row, step) pairs; the h-side recurrences run as
// PERSISTENT kernels (recur_train.hpp: one launch per layer and direction for all sequence inputs - forwards the inference
// recurrence kernel with its per-step gates saved, backwards a BPTT kernel with the state gradient in registers) instead of
// nine dependent launches per step.
//
// Cells (TF 1.15 GRUCell / deepctr VecAttGRUCell, as restated for the scorer in dien.hip):
//   [r, u] = sigmoid([x, h] Wg + bg);  c = tanh([x, r*h] Wc + bc);  AUGRU: u <- (1 - a_t) u;  h' = u h + (1 - u) c
// Flat parameter layout: [cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | obs_w | obs_b | out_w | out_b |
//   per sequence input i: gru_gate_w, gru_gate_b, gru_cand_w, gru_cand_b, att_w1, att_b1, att_w2, att_b2, att_w3, att_b3,
//   augru_gate_w, augru_gate_b, augru_cand_w, augru_cand_b ]
#pragma once

namespace rl4rs {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

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
    float *A1, *R, *Ug, *C, *Hs, *RH;      // A1 [N*L, 3 Hd] = x-side pre-activations [r | u | c] incl. bias; the rest [N*L, Hd]
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
    float *d_logits, *d_obs, *d_allf, *d_h1, *d_h2, *d_hid2, *d_hid1, *d_inp, *dq, *dX, *hprev;
    float *d_score[4], *dK[4], *dAg[4], *dAc[4];            // per sequence input: the inputs' recurrences run in one launch
    float *pk_g[4], *pk_c[4], *pkT_g[4], *pkT_c[4];         // h-side weights in MFMA fragment order (forward) / transposed (backward)
    int32_t* iota;
    float *s_tmpw, *loss_rows, *lr_dummy;
    int64_t adam_t;
    std::vector<void*> owned;
    // Round 6: the work that follows a recurrent layer is a chain of ~35 small launches PER SEQUENCE INPUT (sample-axis reductions,
    // transposes, the attention MLP): 570 + 570 + 280 + 280 us of a 3.97 ms step, most of them far from filling the chip.  The odd
    // inputs' chains run on a second stream with their own scratch, beside the even ones' (rl4rs_dientrain_set_fork(0): one stream).
    hipStream_t side;
    hipEvent_t ev_fork, ev_join;
    TrainCtx cx2;
    float *hprev2, *dX2, *d_hid2b, *d_hid1b, *d_inp2, *dq2;
};


// scratch of one stream's per-input chains
struct InputScratch { TrainCtx* cx; float *hprev, *dX, *d_hid2, *d_hid1, *d_inp, *dq; };

namespace {

// Forward of one recurrent layer for ALL sequence inputs: x-side pre-activations by GEMM, h-side weights into fragment order,
// then ONE persistent launch (grid.y = S) that also saves the per-step gates.  which = 0: first GRU, 1: AUGRU.
int layer_forward(rl4rs_dientrain* t, int N, int which, const float* const* Xin, const float* const* att, hipStream_t st) {
    const int E = t->c.emb_size, L = t->c.maxlen, S = t->c.seq_num;
    RecurTrainFwd f;
    memset(&f, 0, sizeof(f));
    f.N = N; f.L = L; f.S = S; f.iota = t->iota;
    f.hard = 0; f.xblk[0] = 0; f.xblk[1] = 1; f.xblk[2] = 2;
    int rc;
    for (int s = 0; s < S; ++s) {
        const CellSave& cl = which == 0 ? t->gru[s] : t->aug[s];
        const int Hd = cl.Hd;
        f.Hd = Hd;
        const float* Wg = t->params + t->off[cl.pgw];
        const float* bg = t->params + t->off[cl.pgw + 1];
        const float* Wc = t->params + t->off[cl.pgw + 2];
        const float* bc = t->params + t->off[cl.pgw + 3];
        if ((rc = launch_gemm_f32(Xin[s], E, Wg, 2 * Hd, bg, cl.A1, 3 * Hd, N * L, 2 * Hd, E, 0, st))) return rc;
        if ((rc = launch_gemm_f32(Xin[s], E, Wc, Hd, bc, cl.A1 + 2 * Hd, 3 * Hd, N * L, Hd, E, 0, st))) return rc;
        if ((rc = launch_pack_frag(Wg, 2 * Hd, E, Hd, 2 * Hd, 0, t->pk_g[s], st))) return rc;
        if ((rc = launch_pack_frag(Wc, Hd, E, Hd, Hd, 0, t->pk_c[s], st))) return rc;
        f.a1[s] = cl.A1; f.wg[s] = t->pk_g[s]; f.wc[s] = t->pk_c[s]; f.att[s] = att ? att[s] : nullptr;
        f.R[s] = cl.R; f.U[s] = cl.Ug; f.C[s] = cl.C; f.H[s] = cl.Hs; f.RH[s] = cl.RH;
    }
    return launch_recur_train_fwd(f, st);
}

// BPTT of one recurrent layer for ALL sequence inputs in ONE persistent launch: up_last[s] [N, Hd] (row stride ld_up) =
// gradient of the final state, up_all[s] [N*L, Hd] = gradient of every state (either may be NULL); writes the pre-activation
// gradients dAg[s] / dAc[s] of every (row, step) and (AUGRU) d a_t into d_score[s].
int layer_backward(rl4rs_dientrain* t, int N, int which, const float* const* up_last, int64_t ld_up, const float* const* up_all,
                   const float* const* att, hipStream_t st) {
    const int E = t->c.emb_size, L = t->c.maxlen, S = t->c.seq_num;
    RecurTrainBwd b;
    memset(&b, 0, sizeof(b));
    b.N = N; b.L = L; b.S = S; b.ld_up = ld_up;
    int rc;
    for (int s = 0; s < S; ++s) {
        const CellSave& cl = which == 0 ? t->gru[s] : t->aug[s];
        const int Hd = cl.Hd;
        b.Hd = Hd;
        const float* Wg = t->params + t->off[cl.pgw];
        const float* Wc = t->params + t->off[cl.pgw + 2];
        // h-side weights transposed, in fragment order: Wc[E:, :]^T [Hd x Hd], Wg[E:, :]^T [2Hd x Hd]
        if ((rc = launch_pack_frag(Wc, Hd, E, Hd, Hd, 1, t->pkT_c[s], st))) return rc;
        if ((rc = launch_pack_frag(Wg, 2 * Hd, E, 2 * Hd, Hd, 1, t->pkT_g[s], st))) return rc;
        b.R[s] = cl.R; b.U[s] = cl.Ug; b.C[s] = cl.C; b.H[s] = cl.Hs; b.att[s] = att ? att[s] : nullptr;
        b.up_last[s] = up_last ? up_last[s] : nullptr; b.up_all[s] = up_all ? up_all[s] : nullptr;
        b.wcT[s] = t->pkT_c[s]; b.wgT[s] = t->pkT_g[s];
        b.dr[s] = t->dAg[s]; b.du[s] = t->dAg[s] + Hd; b.dc[s] = t->dAc[s]; b.d_score[s] = att ? t->d_score[s] : nullptr;
        b.ld_g = 2 * Hd; b.ld_c = Hd;
    }
    b.hard = 0;
    return launch_recur_train_bwd(b, st);
}

// Parameter gradients and the gradient of the layer input of one cell from its pre-activation gradients (sample-axis GEMM
// reductions over all N * L (row, step) pairs).  Accumulates dXin [N*L, E] INTO dXin_acc (accumulate) or overwrites it.
int cell_backward_post(rl4rs_dientrain* t, int N, const CellSave& cl, const float* Xin, const float* dAg, const float* dAc,
                       float* dXin_acc, bool accumulate, hipStream_t st, const InputScratch& sc) {
    const int E = t->c.emb_size, L = t->c.maxlen, Hd = cl.Hd;
    const float* Wg = t->params + t->off[cl.pgw];
    const float* Wc = t->params + t->off[cl.pgw + 2];
    float* gWg = t->grad + t->off[cl.pgw];
    float* gbg = t->grad + t->off[cl.pgw + 1];
    float* gWc = t->grad + t->off[cl.pgw + 2];
    float* gbc = t->grad + t->off[cl.pgw + 3];
    int rc;
    const dim3 b256(256);
    const int Ns = N * L;
    hipLaunchKernelGGL(k_shift_prev_w, dim3((Ns * Hd + 255) / 256), b256, 0, st, cl.Hs, sc.hprev, N, Hd, L);
    // gate_w = [x rows ; h rows] x 2Hd columns, cand_w likewise x Hd columns
    // (the bias gradients = column sums of dAg / dAc ride on the x-side reductions, whose first tile row holds those values anyway)
    st_tn_cs(*sc.cx, st, Xin, E, E, dAg, 2 * Hd, 2 * Hd, Ns, gWg, gbg);
    st_tn(*sc.cx, st, sc.hprev, Hd, Hd, dAg, 2 * Hd, 2 * Hd, Ns, gWg + (size_t)E * 2 * Hd);
    st_tn_cs(*sc.cx, st, Xin, E, E, dAc, Hd, Hd, Ns, gWc, gbc);
    st_tn(*sc.cx, st, cl.RH, Hd, Hd, dAc, Hd, Hd, Ns, gWc + (size_t)E * Hd);
    // gradient of the layer input: dAg Wg[:E]^T + dAc Wc[:E]^T
    float* dst = accumulate ? sc.dX : dXin_acc;
    if ((rc = st_back(*sc.cx, st, dAg, 2 * Hd, 2 * Hd, Wg, 2 * Hd, E, dst, E, Ns))) return rc;
    if (accumulate) hipLaunchKernelGGL(k_add_inplace, dim3((Ns * E + 255) / 256), b256, 0, st, dXin_acc, sc.dX, Ns * E);
    if ((rc = st_back(*sc.cx, st, dAc, Hd, Hd, Wc, Hd, E, sc.dX, E, Ns))) return rc;
    hipLaunchKernelGGL(k_add_inplace, dim3((Ns * E + 255) / 256), b256, 0, st, dXin_acc, sc.dX, Ns * E);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_dientrain_set_fork(int32_t on) {
    g_dientrain_fork = on ? 1 : 0;
    return RL4RS_OK;
}

int rl4rs_dientrain_destroy(rl4rs_dientrain* t) {
    if (!t) return RL4RS_OK;
    for (void* q : t->owned) (void)hipFree(q);
    if (t->side) { (void)hipStreamDestroy(t->side); (void)hipEventDestroy(t->ev_fork); (void)hipEventDestroy(t->ev_join); }
    delete t;
    return RL4RS_OK;
}

int rl4rs_dientrain_create(const rl4rs_dien_cfg* c, const rl4rs_dien_weights* w, int32_t max_batch, void* stream,
                           rl4rs_dientrain** out) {
    RL4RS_REQUIRE(c && w && out && max_batch > 0, "dientrain_create: bad argument");
    RL4RS_REQUIRE(c->emb_size == 128, "dientrain: emb_size must be 128 (the recurrence kernels are built for hidden widths 128 / 256), got %d",
                  c->emb_size);
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
            DT_FAIL(al(&cl.A1, Ns * 3 * cl.Hd)); DT_FAIL(al(&cl.R, Ns * cl.Hd));
            DT_FAIL(al(&cl.Ug, Ns * cl.Hd)); DT_FAIL(al(&cl.C, Ns * cl.Hd)); DT_FAIL(al(&cl.Hs, Ns * cl.Hd)); DT_FAIL(al(&cl.RH, Ns * cl.Hd));
        }
        DT_FAIL(al(&t->d_score[s], Ns)); DT_FAIL(al(&t->dK[s], Ns * E)); DT_FAIL(al(&t->dAg[s], Ns * 2 * NH2)); DT_FAIL(al(&t->dAc[s], Ns * NH2));
        DT_FAIL(al(&t->pk_g[s], NH2 * 2 * NH2)); DT_FAIL(al(&t->pk_c[s], NH2 * NH2));
        DT_FAIL(al(&t->pkT_g[s], 2 * NH2 * NH2)); DT_FAIL(al(&t->pkT_c[s], NH2 * NH2));
    }
    {
        float* p; DT_FAIL(al(&p, B)); t->iota = reinterpret_cast<int32_t*>(p);
        std::vector<int32_t> io(B);
        for (size_t i = 0; i < B; ++i) io[i] = (int32_t)i;
        DT_HIP(hipMemcpyAsync(t->iota, io.data(), B * 4, hipMemcpyHostToDevice, st));
        DT_HIP(hipStreamSynchronize(st));              // io is a local
    }
    DT_FAIL(al(&t->q, B * E)); DT_FAIL(al(&t->allf, B * t->F)); DT_FAIL(al(&t->h1, B * U)); DT_FAIL(al(&t->h1d, B * U));
    DT_FAIL(al(&t->h2, B * U)); DT_FAIL(al(&t->obs, B * 256)); DT_FAIL(al(&t->logits, B * K)); DT_FAIL(al(&t->dC, B * Cn * E));
    { float* p; DT_FAIL(al(&p, B * 10)); t->ids10 = reinterpret_cast<int32_t*>(p); }
    { float* p; DT_FAIL(al(&p, (B * U + 3) / 4 + 1)); t->mask1 = reinterpret_cast<uint8_t*>(p);
      DT_FAIL(al(&p, (B * U + 3) / 4 + 1)); t->mask2 = reinterpret_cast<uint8_t*>(p); }
    DT_FAIL(al(&t->d_logits, B * K)); DT_FAIL(al(&t->d_obs, B * 256)); DT_FAIL(al(&t->d_allf, B * t->F)); DT_FAIL(al(&t->d_h1, B * U));
    DT_FAIL(al(&t->d_h2, B * U)); DT_FAIL(al(&t->d_hid2, Ns * 16)); DT_FAIL(al(&t->d_hid1, Ns * 64));
    DT_FAIL(al(&t->d_inp, Ns * 4 * E)); DT_FAIL(al(&t->dq, B * E));
    DT_FAIL(al(&t->dX, Ns * E)); DT_FAIL(al(&t->hprev, Ns * NH2));
    DT_FAIL(al(&t->s_tmpw, 4)); DT_FAIL(al(&t->loss_rows, B));
    DT_FAIL(al(&t->lr_dummy, 4));
    // reduction scratch: the largest M x Nc of any weight gradient, times the number of 512-sample chunks of N * L
    int64_t wmax = (int64_t)t->F * 256;
    if (Dn * U > wmax) wmax = Dn * U;
    if (3 * E * 4 * E > wmax) wmax = 3 * E * 4 * E;
    t->cx.chunk = 512;
    const int nz_all = (int)((Ns + 511) / 512);
    DT_FAIL(al(&t->cx.wt, wmax));
    DT_FAIL(al(&t->cx.part, (size_t)nz_all * wmax));
    t->cx.part_cap = (size_t)nz_all * wmax;
    // the second stream's scratch (only per-input reductions run there: the widest is the AUGRU's [3E x 4E] gate gradient)
    t->side = nullptr; t->ev_fork = nullptr; t->ev_join = nullptr;
    if (S > 1) {
        const int64_t wmax2 = (int64_t)3 * E * 4 * E;
        t->cx2.chunk = 512;
        DT_FAIL(al(&t->cx2.wt, wmax2));
        DT_FAIL(al(&t->cx2.part, (size_t)nz_all * wmax2));
        t->cx2.part_cap = (size_t)nz_all * wmax2;
        DT_FAIL(al(&t->hprev2, Ns * NH2)); DT_FAIL(al(&t->dX2, Ns * E)); DT_FAIL(al(&t->d_hid2b, Ns * 16)); DT_FAIL(al(&t->d_hid1b, Ns * 64));
        DT_FAIL(al(&t->d_inp2, Ns * 4 * E)); DT_FAIL(al(&t->dq2, B * E));
        DT_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
        DT_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
        DT_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
    }
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
    const float* Xs[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* Ks[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* scs[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int s = 0; s < S; ++s) {
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, seq[s], N, L, H, E, P + o[DP_SEQ_EMB], t->X[s], (int64_t)L * E, 0);
        Xs[s] = t->X[s]; Ks[s] = t->gru[s].Hs; scs[s] = t->score[s];
    }
    if ((rc = layer_forward(t, N, 0, Xs, nullptr, st))) return rc;         // first GRU of every sequence input: one launch
    // the chains of the odd sequence inputs on the second stream (fork / join = one event each way)
    const bool two = t->side != nullptr && g_dientrain_fork && S > 1;
    auto fork = [&]() -> int {
        if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_fork, st)); RL4RS_HIP_TRY(hipStreamWaitEvent(t->side, t->ev_fork, 0)); }
        return RL4RS_OK;
    };
    auto join = [&]() -> int {
        if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_join, t->side)); RL4RS_HIP_TRY(hipStreamWaitEvent(st, t->ev_join, 0)); }
        return RL4RS_OK;
    };
    auto stream_of = [&](int s) { return (two && (s & 1)) ? t->side : st; };
    const InputScratch sc_main = {&t->cx, t->hprev, t->dX, t->d_hid2, t->d_hid1, t->d_inp, t->dq};
    const InputScratch sc_side = {&t->cx2, t->hprev2, t->dX2, t->d_hid2b, t->d_hid1b, t->d_inp2, t->dq2};
    auto scratch_of = [&](int s) -> const InputScratch& { return (two && (s & 1)) ? sc_side : sc_main; };
    if ((rc = fork())) return rc;
    for (int s = 0; s < S; ++s) {
        const int pb = DP_SEQ0 + s * DP_PER_SEQ;
        hipStream_t ss = stream_of(s);
        const float* Kk = t->gru[s].Hs;                                     // keys = first-GRU states [N*L, E]
        hipLaunchKernelGGL(k_att_inp, ew(Ns * E), b256, 0, ss, t->q, Kk, t->inp[s], N, L, E);
        if ((rc = launch_gemm_f32(t->inp[s], 4 * E, P + o[pb + DQ_ATT_W1], 64, P + o[pb + DQ_ATT_B1], t->hid1[s], 64, Ns, 64, 4 * E, 2, ss))) return rc;
        if ((rc = launch_gemm_f32(t->hid1[s], 64, P + o[pb + DQ_ATT_W2], 16, P + o[pb + DQ_ATT_B2], t->hid2[s], 16, Ns, 16, 64, 2, ss))) return rc;
        if ((rc = launch_gemm_f32(t->hid2[s], 16, P + o[pb + DQ_ATT_W3], 1, P + o[pb + DQ_ATT_B3], t->score[s], 1, Ns, 1, 16, 0, ss))) return rc;
    }
    if ((rc = join())) return rc;
    if ((rc = layer_forward(t, N, 1, Ks, scs, st))) return rc;              // AUGRU of every sequence input: one launch
    for (int s = 0; s < S; ++s)
        RL4RS_HIP_TRY(hipMemcpy2DAsync(t->allf + s * NH2, (size_t)F * 4, t->aug[s].Hs + (size_t)(L - 1) * NH2, (size_t)L * NH2 * 4,
                                       (size_t)NH2 * 4, N, hipMemcpyDeviceToDevice, st));
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
    if (two) RL4RS_HIP_TRY(hipMemsetAsync(t->dq2, 0, (size_t)N * E * 4, st));
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
    // sequence inputs.  AUGRU of every input in one launch: gradient of the final state in, d a_t and dAg / dAc out
    const float* ups[4] = {nullptr, nullptr, nullptr, nullptr};
    const float* dKs[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int s = 0; s < S; ++s) { ups[s] = t->d_allf + s * NH2; dKs[s] = t->dK[s]; }
    if ((rc = layer_backward(t, N, 1, ups, (int64_t)F, nullptr, scs, st))) return rc;
    if ((rc = fork())) return rc;
    for (int s = 0; s < S; ++s) {
        const int pb = DP_SEQ0 + s * DP_PER_SEQ;
        hipStream_t ss = stream_of(s);
        const InputScratch& sc = scratch_of(s);
        const float* Kk = t->gru[s].Hs;
        float* d_score = t->d_score[s];
        float* dK = t->dK[s];
        // AUGRU parameter gradients; the gradient of its inputs (= the keys) starts dK
        if ((rc = cell_backward_post(t, N, t->aug[s], Kk, t->dAg[s], t->dAc[s], dK, false, ss, sc))) return rc;
        // attention MLP (LocalActivationUnit, att_hidden_units = (64, 16), sigmoid; raw score)
        st_tn(*sc.cx, ss, t->hid2[s], 16, 16, d_score, 1, 1, Ns, G + o[pb + DQ_ATT_W3]);
        st_cs(*sc.cx, ss, d_score, 1, 1, Ns, G + o[pb + DQ_ATT_B3]);
        if ((rc = st_back(*sc.cx, ss, d_score, 1, 1, P + o[pb + DQ_ATT_W3], 1, 16, sc.d_hid2, 16, Ns))) return rc;
        hipLaunchKernelGGL(k_sig_bwd, ew(Ns * 16), b256, 0, ss, sc.d_hid2, t->hid2[s], Ns * 16);
        st_tn(*sc.cx, ss, t->hid1[s], 64, 64, sc.d_hid2, 16, 16, Ns, G + o[pb + DQ_ATT_W2]);
        st_cs(*sc.cx, ss, sc.d_hid2, 16, 16, Ns, G + o[pb + DQ_ATT_B2]);
        if ((rc = st_back(*sc.cx, ss, sc.d_hid2, 16, 16, P + o[pb + DQ_ATT_W2], 16, 64, sc.d_hid1, 64, Ns))) return rc;
        hipLaunchKernelGGL(k_sig_bwd, ew(Ns * 64), b256, 0, ss, sc.d_hid1, t->hid1[s], Ns * 64);
        st_tn(*sc.cx, ss, t->inp[s], 4 * E, 4 * E, sc.d_hid1, 64, 64, Ns, G + o[pb + DQ_ATT_W1]);
        st_cs(*sc.cx, ss, sc.d_hid1, 64, 64, Ns, G + o[pb + DQ_ATT_B1]);
        if ((rc = st_back(*sc.cx, ss, sc.d_hid1, 64, 64, P + o[pb + DQ_ATT_W1], 64, 4 * E, sc.d_inp, 4 * E, Ns))) return rc;
        hipLaunchKernelGGL(k_att_inp_bwd, ew(N * E), b256, 0, ss, sc.d_inp, t->q, Kk, dK, sc.dq, N, L, E);
    }
    if ((rc = join())) return rc;
    // first GRU of every input in one launch: every state has an upstream gradient (it is a key and an AUGRU input)
    if ((rc = layer_backward(t, N, 0, nullptr, 0, dKs, nullptr, st))) return rc;
    if ((rc = fork())) return rc;
    for (int s = 0; s < S; ++s) {
        hipStream_t ss = stream_of(s);
        const InputScratch& sc = scratch_of(s);
        // its inputs are embedding rows (the table gradient takes float atomics from both streams)
        if ((rc = cell_backward_post(t, N, t->gru[s], t->X[s], t->dAg[s], t->dAc[s], sc.d_inp /* scratch [N*L, E] */, false, ss, sc))) return rc;
        hipLaunchKernelGGL(k_emb_flatten_bwd, g4, b256, 0, ss, seq[s], N, L, H, E, sc.d_inp, (int64_t)L * E, G + o[DP_SEQ_EMB]);
    }
    if ((rc = join())) return rc;
    if (two) hipLaunchKernelGGL(k_add_inplace, ew(N * E), b256, 0, st, t->dq, t->dq2, N * E);      // the query gradient of the odd inputs
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
