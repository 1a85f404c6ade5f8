// Supervised training of the 'dnn', 'widedeep' and 'lstm' simulator families on the device: one optimiser step = forward (training mode:
// Dropout(0.2) after each dense-tower layer, utils.py:48-54) + keras binary_crossentropy on the softmax output against
// the one-hot label + backward + Adam - what `model.compile(loss='binary_crossentropy', optimizer='adam')` /
// `model.fit` do in script/supervised_train.py:37-42 for rl4rs/nets/dnn.py, widedeep.py and lstm.py.  Included at the end of policy.hip: it uses
// that translation unit's sample-axis gradient reductions (k_gemm_tn / k_colsum / k_reduce_chunks) and Adam kernel.
// Everything is fp32 on the fp32 MFMA GEMMs; batches are small (256 in the reference), so this path is launch-bound.
//
// Flat parameter / gradient / Adam-state layout (one buffer each; arrays a family does not have are skipped):
//   [ cat_emb H*E | seq_emb H*E (widedeep) | dense_w1 Dn*U | dense_b1 U | dense_w2 U*U | dense_b2 U |
//     fc_w (dnn: (E+U)*256, widedeep: S*E*256) | fc_b 256 | obs_w 256*256 (dnn) | obs_b 256 (dnn) | out_w OD*K | out_b K ]
//   OD = 256 (dnn) or 256 + U + Cn*E (widedeep: 'simulator_obs' is the concat itself)
#pragma once

namespace rl4rs {

// y = ELU(x) was applied by the GEMM; training-mode dropout on top: keep with probability 1 - rate, scale 1/(1 - rate).
__global__ void k_dropout(float* __restrict__ x, uint8_t* __restrict__ mask, int n, int cols, float rate, uint32_t seed,
                          uint32_t step, uint32_t layer) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = i / cols, c = i - row * cols;
    const bool keep = rate <= 0.f || uniform01(seed, step, (uint32_t)row, (uint32_t)c + layer * 65536u) >= rate;
    mask[i] = keep ? 1 : 0;
    x[i] = keep ? x[i] / (1.0f - rate) : 0.f;
}

// d_pre = d_out (* mask / (1 - rate)) * ELU'(y),  ELU'(x) = 1 for x > 0 else ELU(x) + 1;  y = the layer's (pre-dropout) output
__global__ void k_elu_bwd(float* __restrict__ d, int64_t ldd, const float* __restrict__ y, int64_t ldy,
                          const uint8_t* __restrict__ mask, float rate, int n, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = i / cols, c = i - row * cols;
    float g = d[(size_t)row * ldd + c];
    if (mask) g = mask[i] ? g / (1.0f - rate) : 0.f;
    const float yy = y[(size_t)row * ldy + c];
    d[(size_t)row * ldd + c] = g * (yy > 0.f ? 1.f : yy + 1.f);
}

// keras binary_crossentropy(one_hot(label), softmax(logits)): row loss = -(1/K) sum_k [y log p + (1-y) log(1-p)] with p
// clipped to [1e-7, 1 - 1e-7] (zero gradient where clipped); writes d loss_mean / d logits (scaled by 1/N).
__global__ void k_bce_softmax(const float* __restrict__ logits, const int32_t* __restrict__ labels, int N, int K,
                              float* __restrict__ dlogits, float* __restrict__ loss_rows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float l[8], p[8], dp[8];
    float mx = -3.4028235e38f;
    for (int k = 0; k < K; ++k) { l[k] = logits[(size_t)n * K + k]; mx = fmaxf(mx, l[k]); }
    float se = 0.f;
    for (int k = 0; k < K; ++k) { p[k] = expf(l[k] - mx); se += p[k]; }
    const int y = labels[n];
    float loss = 0.f, dot = 0.f;
    for (int k = 0; k < K; ++k) {
        p[k] /= se;
        const float pc = fminf(fmaxf(p[k], 1e-7f), 1.0f - 1e-7f);
        const bool inside = p[k] > 1e-7f && p[k] < 1.0f - 1e-7f;
        if (k == y) { loss -= logf(pc); dp[k] = inside ? -1.0f / pc : 0.f; }
        else { loss -= logf(1.0f - pc); dp[k] = inside ? 1.0f / (1.0f - pc) : 0.f; }
        dp[k] /= (float)K;
        dot += dp[k] * p[k];
    }
    loss_rows[n] = loss / (float)K;
    for (int k = 0; k < K; ++k) dlogits[(size_t)n * K + k] = p[k] * (dp[k] - dot) / (float)N;
}

__global__ void k_transpose(const float* __restrict__ w, int64_t ldw, int rows, int cols, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    wt[(size_t)c * rows + r] = w[(size_t)r * ldw + c];
}

// gradient of the mean-pooled embedding: g_table[ids[row, j]] += d_feat[row] / len   (float atomics)
__global__ __launch_bounds__(256) void k_emb_mean_bwd(const int32_t* __restrict__ ids, int n, int len, int H, int E,
                                                      const float* __restrict__ d_feat, int64_t ld, float* __restrict__ g_table) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    const float inv = 1.0f / (float)len;
    for (int j = 0; j < len; ++j) {
        const int id = min(max(ids[(size_t)row * len + j], 0), H - 1);
        for (int k = lane; k < E; k += 64) atomicAdd(&g_table[(size_t)id * E + k], d_feat[(size_t)row * ld + k] * inv);
    }
}

// gradient of Flatten(embedding rows): g_table[ids[row, j]][k] += d[row, j*E + k]
__global__ __launch_bounds__(256) void k_emb_flatten_bwd(const int32_t* __restrict__ ids, int n, int len, int H, int E,
                                                         const float* __restrict__ d, int64_t ld, float* __restrict__ g_table) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    for (int j = 0; j < len; ++j) {
        const int id = min(max(ids[(size_t)row * len + j], 0), H - 1);
        for (int k = lane; k < E; k += 64) atomicAdd(&g_table[(size_t)id * E + k], d[(size_t)row * ld + (size_t)j * E + k]);
    }
}

// ---- keras GRU (hard_sigmoid gates, reset_after = False: z, r = hs(x W + h U + b); hh = tanh(x Wh + (r*h) Uh + bh);
//      h = z h + (1-z) hh), training mode ---------------------------------------------------------------------------
// All per-sequence arrays are [N, len, W] row-major; a step works on the slice t (row stride len*W).
__device__ __forceinline__ float hard_sig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }
__device__ __forceinline__ float hard_sig_grad(float x) { return (x > -2.5f && x < 2.5f) ? 0.2f : 0.f; }

// z, r = hs(x-side + h_prev U_zr);  keeps the pre-activations (for hs') and r * h_prev (operand of the candidate GEMM)
__global__ void k_gru_gates(const float* __restrict__ a1, const float* __restrict__ g, const float* __restrict__ hprev,
                            int64_t ldh, float* __restrict__ azr, float* __restrict__ z, float* __restrict__ r,
                            float* __restrict__ rh, int N, int U, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * U) return;
    const int n = i / U, c = i - n * U;
    const size_t s3 = ((size_t)n * len + t) * 3 * U, s2 = ((size_t)n * len + t) * 2 * U, s1 = ((size_t)n * len + t) * U;
    const float az = a1[s3 + c] + g[(size_t)n * 2 * U + c];
    const float ar = a1[s3 + U + c] + g[(size_t)n * 2 * U + U + c];
    const float hp = hprev[(size_t)n * ldh + c];
    const float rr = hard_sig(ar);
    azr[s2 + c] = az; azr[s2 + U + c] = ar;
    z[s1 + c] = hard_sig(az); r[s1 + c] = rr;
    rh[s1 + c] = rr * hp;
}

// hh = tanh(x-side + (r*h_prev) U_h);  h = z h_prev + (1 - z) hh
__global__ void k_gru_update(const float* __restrict__ a1, const float* __restrict__ gh, const float* __restrict__ hprev,
                             int64_t ldh, const float* __restrict__ z, float* __restrict__ hh, float* __restrict__ h,
                             int N, int U, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * U) return;
    const int n = i / U, c = i - n * U;
    const size_t s3 = ((size_t)n * len + t) * 3 * U, s1 = ((size_t)n * len + t) * U;
    const float c_ = tanhf(a1[s3 + 2 * U + c] + gh[(size_t)n * U + c]);
    const float zz = z[s1 + c];
    hh[s1 + c] = c_;
    h[s1 + c] = zz * hprev[(size_t)n * ldh + c] + (1.0f - zz) * c_;
}

// BPTT step, part 1: dh = dh_a + dh_b (+ upstream at the last step);  da_h = dh (1 - z) (1 - hh^2)
__global__ void k_gru_bwd_pre(const float* __restrict__ dh_a, const float* __restrict__ dh_b, const float* __restrict__ up,
                              int64_t ld_up, float* __restrict__ dh, const float* __restrict__ z, const float* __restrict__ hh,
                              float* __restrict__ dA, int N, int U, int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * U) return;
    const int n = i / U, c = i - n * U;
    const size_t s3 = ((size_t)n * len + t) * 3 * U, s1 = ((size_t)n * len + t) * U;
    float d = (dh_a ? dh_a[i] : 0.f) + (dh_b ? dh_b[i] : 0.f) + (up ? up[(size_t)n * ld_up + c] : 0.f);
    dh[i] = d;
    const float c_ = hh[s1 + c];
    dA[s3 + 2 * U + c] = d * (1.0f - z[s1 + c]) * (1.0f - c_ * c_);
}

// part 2 (after d_rh = da_h U_h^T): da_z, da_r and the direct terms of dh_{t-1}
__global__ void k_gru_bwd_mid(const float* __restrict__ dh, const float* __restrict__ d_rh, const float* __restrict__ hprev,
                              int64_t ldh, const float* __restrict__ z, const float* __restrict__ r, const float* __restrict__ hh,
                              const float* __restrict__ azr, float* __restrict__ dA, float* __restrict__ dh_part, int N, int U,
                              int len, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * U) return;
    const int n = i / U, c = i - n * U;
    const size_t s3 = ((size_t)n * len + t) * 3 * U, s2 = ((size_t)n * len + t) * 2 * U, s1 = ((size_t)n * len + t) * U;
    const float d = dh[i], hp = hprev[(size_t)n * ldh + c], q = d_rh[i];
    dA[s3 + c] = d * (hp - hh[s1 + c]) * hard_sig_grad(azr[s2 + c]);
    dA[s3 + U + c] = q * hp * hard_sig_grad(azr[s2 + U + c]);
    dh_part[i] = d * z[s1 + c] + q * r[s1 + c];
}

// Hprev[n, t] = H[n, t-1] (0 at t = 0), as a contiguous [N*len, U] operand of the weight-gradient reductions
__global__ void k_shift_prev(const float* __restrict__ h, float* __restrict__ hprev, int N, int U, int len) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * len * U) return;
    const int t = (i / U) % len;
    hprev[i] = t == 0 ? 0.f : h[i - U];
}

__global__ void k_mean(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0] / (float)n;
}

}  // namespace rl4rs

enum { SP_CAT_EMB = 0, SP_SEQ_EMB, SP_DW1, SP_DB1, SP_DW2, SP_DB2, SP_FC_W, SP_FC_B, SP_OBS_W, SP_OBS_B, SP_OUT_W, SP_OUT_B,
       SP_GRU0,                    // lstm: 3 arrays (kernel, recurrent, bias) per GRU: the category GRU, then one per sequence
       SP_COUNT = SP_GRU0 + 3 * 5 };

// one keras GRU of the lstm family in training mode: everything the BPTT needs is kept from the forward
struct GruSave {
    float *X, *A1, *H, *Z, *R, *HH, *RH, *AZR;     // [N, len, E | 3U | U | U | U | U | U | 2U]
    float *pkg, *pkc, *dA;                         // persistent-kernel path: fragment-order recurrent weights, [N, len, 3U] gate gradients
    int len, emb, pk;                              // sequence length, SP_* index of its embedding table, of its kernel
    const int32_t* ids;
};

static int g_dientrain_fork = 1;       // rl4rs_dientrain_set_fork: per-input launch chains on two streams (dien and lstm trainers)

struct rl4rs_simtrain {
    rl4rs_simnet_cfg c;
    int64_t n_params;
    int64_t off[SP_COUNT], size[SP_COUNT];       // offsets / sizes in the flat buffers (size 0 = the family has no such array)
    size_t part_cap;                             // floats behind `part`
    int max_batch, chunk, nz, OD, FCK;           // OD = width of 'simulator_obs', FCK = input width of the fc layer
    float *params, *grad, *adam_m, *adam_v;
    float *feat, *h1, *h1d, *h2, *a1, *obs, *logits;            // activations (feat = input of fc, a1 = its output for dnn)
    float *d_logits, *d_obs, *d_a, *d_feat, *d_h1, *d_h2, *wt, *part, *loss_rows, *lr_dummy;
    uint8_t *mask1, *mask2;
    int64_t adam_t;
    // lstm: saved forwards of the 1 + seq_num GRUs and BPTT scratch
    GruSave gru[5];
    float *g_dA, *g_dX, *g_hprev, *g_G, *g_Gh, *g_dh, *g_dhp, *g_dhg, *g_drh, *g_zero, *g_uzrT, *g_uhT, *g_tmpw;
    int32_t* g_iota;                   // 0, 1, 2, ... (persistent recurrence kernels, recur_train.hpp); NULL: step-by-step form
    std::vector<void*> owned;
    // lstm, round 6: the category GRU (its own length: its own persistent launches) and the LAST sequence GRU's gradient chain run on
    // a second stream with their own scratch, beside the sequence GRUs' launches and the other chains (rl4rs_dientrain_set_fork(0):
    // one stream)
    hipStream_t side;
    hipEvent_t ev_fork, ev_mid, ev_join;
    float *g_dX2, *g_hprev2, *g_tmpw2, *wt2, *part2;
    size_t part_cap2;
};
struct GruScratch { float *dX, *hprev, *tmpw, *wt, *part; size_t part_cap; };

namespace {

// sample-axis reductions / transposed-weight GEMM over an explicit number of samples
struct TrainCtx { int chunk; float* part; float* wt; size_t part_cap = 0; };       // part_cap: floats behind `part` (0 = sized for `chunk` only)

// Long reductions with wide outputs go through the LDS-tiled 128 x 128 form (k_gemm_tn_t128, policy.hip); the chunk count is chosen
// to give every CU a workgroup (at most 64 chunks: the partials are summed by k_reduce_chunks4) within what `part` holds.  Returns the chunk count, 0 = not applicable (caller: k_gemm_tn).
inline int tn_t128_plan(const TrainCtx& x, const float* A, int lda, int M, const float* B, int ldb, int Nc, int Ns, bool bias, int* chunk_out) {
    if (Ns < 4096 || M < 64 || Nc < 64 || (lda | ldb | M | Nc) % 4 != 0 || ((uintptr_t)A | (uintptr_t)B) % 16 != 0) return 0;
    const int tiles = ((M + 127) / 128) * ((Nc + 127) / 128);
    int nz = (256 + tiles - 1) / tiles;
    if (nz > 64) nz = 64;
    const size_t per = (size_t)M * Nc + (bias ? Nc : 0);
    const size_t cap = x.part_cap ? x.part_cap : (size_t)((Ns + x.chunk - 1) / x.chunk) * per;
    if ((size_t)nz * per > cap) nz = (int)(cap / per);
    if (nz < 1) return 0;
    int chunk = (((Ns + nz - 1) / nz) + 15) / 16 * 16;
    if (chunk < 64) chunk = 64;
    *chunk_out = chunk;
    return (Ns + chunk - 1) / chunk;
}
constexpr int TN4_MAX_SAMPLES = 1024;      // up to here the whole sample axis is one workgroup's (four waves x Ns / 4 samples)
void st_tn(const TrainCtx& x, hipStream_t st, const float* A, int lda, int M, const float* B, int ldb, int Nc, int Ns, float* dst) {
    if (Ns <= TN4_MAX_SAMPLES) {
        hipLaunchKernelGGL(k_gemm_tn4, dim3(((M + 31) / 32) * ((Nc + 31) / 32)), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, dst, (float*)nullptr);
        return;
    }
    int big_chunk = 0;
    if (const int bz = tn_t128_plan(x, A, lda, M, B, ldb, Nc, Ns, false, &big_chunk)) {
        hipLaunchKernelGGL(k_gemm_tn_t128, dim3(((M + 127) / 128) * ((Nc + 127) / 128), bz), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, big_chunk,
                           bz == 1 ? dst : x.part, (float*)nullptr);
        if (bz > 1) hipLaunchKernelGGL(k_reduce_chunks4, dim3((M * Nc + 63) / 64), dim3(256), 0, st, x.part, M * Nc, bz, dst);
        return;
    }
    const int nz = (Ns + x.chunk - 1) / x.chunk;
    const int tiles = ((M + 31) / 32) * ((Nc + 31) / 32);
    hipLaunchKernelGGL(k_gemm_tn, dim3((tiles + 3) / 4, nz), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, x.chunk, nz == 1 ? dst : x.part, (float*)nullptr);
    if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((M * Nc + 255) / 256), dim3(256), 0, st, x.part, M * Nc, nz, dst);
}
void st_cs(const TrainCtx& x, hipStream_t st, const float* X, int ld, int Nc, int Ns, float* dst) {
    const int nz = (Ns + x.chunk - 1) / x.chunk;
    hipLaunchKernelGGL(k_colsum, dim3((Nc + 63) / 64, nz), dim3(64), 0, st, X, ld, Nc, Ns, x.chunk, nz == 1 ? dst : x.part);
    if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((Nc + 255) / 256), dim3(256), 0, st, x.part, Nc, nz, dst);
}
// weight AND bias gradient of one Linear layer in one launch: dW [M, Nc] = A^T B, db [Nc] = column sums of B.  x.part must hold
// nz * (M * Nc + Nc) floats when Ns > x.chunk.
void st_tn_cs(const TrainCtx& x, hipStream_t st, const float* A, int lda, int M, const float* B, int ldb, int Nc, int Ns, float* dW, float* db) {
    if (Ns <= TN4_MAX_SAMPLES) {
        hipLaunchKernelGGL(k_gemm_tn4, dim3(((M + 31) / 32) * ((Nc + 31) / 32)), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, dW, db);
        return;
    }
    int big_chunk = 0;
    if (const int bz = tn_t128_plan(x, A, lda, M, B, ldb, Nc, Ns, true, &big_chunk)) {
        float* bp = x.part + (size_t)bz * M * Nc;
        hipLaunchKernelGGL(k_gemm_tn_t128, dim3(((M + 127) / 128) * ((Nc + 127) / 128), bz), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, big_chunk,
                           bz == 1 ? dW : x.part, bz == 1 ? db : bp);
        if (bz > 1) {
            hipLaunchKernelGGL(k_reduce_chunks4, dim3((M * Nc + 63) / 64), dim3(256), 0, st, x.part, M * Nc, bz, dW);
            hipLaunchKernelGGL(k_reduce_chunks4, dim3((Nc + 63) / 64), dim3(256), 0, st, bp, Nc, bz, db);
        }
        return;
    }
    const int nz = (Ns + x.chunk - 1) / x.chunk;
    const int tiles = ((M + 31) / 32) * ((Nc + 31) / 32);
    float* bpart = x.part + (size_t)nz * M * Nc;
    hipLaunchKernelGGL(k_gemm_tn, dim3((tiles + 3) / 4, nz), dim3(256), 0, st, A, lda, M, B, ldb, Nc, Ns, x.chunk, nz == 1 ? dW : x.part,
                       nz == 1 ? db : bpart);
    if (nz > 1) {
        hipLaunchKernelGGL(k_reduce_chunks, dim3((M * Nc + 255) / 256), dim3(256), 0, st, x.part, M * Nc, nz, dW);
        hipLaunchKernelGGL(k_reduce_chunks, dim3((Nc + 255) / 256), dim3(256), 0, st, bpart, Nc, nz, db);
    }
}
int st_back(const TrainCtx& x, hipStream_t st, const float* dY, int ldy, int Nout, const float* W, int ldw, int Kin, float* dX, int ldx,
            int Ns) {      // dX [Ns, Kin] = dY [Ns, Nout] W^T,  W [Kin, Nout] with leading dimension ldw
    // minibatch-sized: no transposed weight copy, 32 x 32 tiles.  (Round 5: the bound is on the ROW count as well - the 16 384-row
    // input gradients of the simulator trainers' recurrent layers took this form at up to 94 us a call)
    if (Ns <= 2048 && (int64_t)((Ns + 127) / 128) * ((Kin + 63) / 64) < 512)
        return launch_gemm_nt(dY, ldy, W, ldw, dX, ldx, Ns, Kin, Nout, st);
    hipLaunchKernelGGL(k_transpose, dim3((Kin * Nout + 255) / 256), dim3(256), 0, st, W, (int64_t)ldw, Kin, Nout, x.wt);
    return launch_gemm_f32(dY, ldy, x.wt, Kin, nullptr, dX, ldx, Ns, Kin, Nout, 0, st);
}

static bool gru_persistent(const rl4rs_simtrain* t, int N, int len) {
    const int U = t->c.hidden_units;
    return t->g_iota && (U == 128 || U == 256) && len <= 64 && (int64_t)N * len * 3 * U * 4 < (int64_t)0x7fffffff;
}

// keras GRU forward for several GRUs of the SAME length in ONE persistent launch (grid.y = n): the inference recurrence kernel
// with hard_sigmoid gates, the keras column order [z | r | h] mapped onto its (reset, update, candidate) roles, and the per-step
// gates saved for the BPTT (recur_train.hpp)
int gru_forward_multi(rl4rs_simtrain* t, int N, GruSave* const* gs, int n, hipStream_t st) {
    const int E = t->c.emb_size, U = t->c.hidden_units, H = t->c.category_hash_size, len = gs[0]->len;
    RecurTrainFwd f;
    memset(&f, 0, sizeof(f));
    f.N = N; f.L = len; f.S = n; f.Hd = U; f.iota = t->g_iota; f.hard = 1;
    f.xblk[0] = 1; f.xblk[1] = 0; f.xblk[2] = 2;
    int rc;
    for (int k = 0; k < n; ++k) {
        GruSave& g = *gs[k];
        const float* K = t->params + t->off[g.pk];
        const float* Rw = t->params + t->off[g.pk + 1];
        const float* b = t->params + t->off[g.pk + 2];
        hipLaunchKernelGGL(k_emb_flatten, dim3((N + 3) / 4), dim3(256), 0, st, g.ids, N, len, H, E, t->params + t->off[g.emb], g.X,
                           (int64_t)len * E, 0);
        if ((rc = launch_gemm_f32(g.X, E, K, 3 * U, b, g.A1, 3 * U, N * len, 3 * U, E, 0, st))) return rc;
        if ((rc = launch_pack_frag(Rw + U, 3 * U, 0, U, U, 0, g.pkg, st))) return rc;                         // reset columns
        if ((rc = launch_pack_frag(Rw, 3 * U, 0, U, U, 0, g.pkg + (size_t)U * U, st))) return rc;             // update (z) columns
        if ((rc = launch_pack_frag(Rw + 2 * U, 3 * U, 0, U, U, 0, g.pkc, st))) return rc;
        f.a1[k] = g.A1; f.wg[k] = g.pkg; f.wc[k] = g.pkc; f.att[k] = nullptr;
        f.R[k] = g.R; f.U[k] = g.Z; f.C[k] = g.HH; f.H[k] = g.H; f.RH[k] = g.RH;
    }
    return launch_recur_train_fwd(f, st);
}

// keras GRU forward over `len` steps for N rows, keeping gates and states (utils.py:34,91: layers.GRU(units=U))
int gru_forward(rl4rs_simtrain* t, int N, GruSave& g, hipStream_t st) {
    const int E = t->c.emb_size, U = t->c.hidden_units, H = t->c.category_hash_size, len = g.len;
    const float* K = t->params + t->off[g.pk];
    const float* Rw = t->params + t->off[g.pk + 1];
    const float* b = t->params + t->off[g.pk + 2];
    int rc;
    hipLaunchKernelGGL(k_emb_flatten, dim3((N + 3) / 4), dim3(256), 0, st, g.ids, N, len, H, E, t->params + t->off[g.emb], g.X,
                       (int64_t)len * E, 0);
    if ((rc = launch_gemm_f32(g.X, E, K, 3 * U, b, g.A1, 3 * U, N * len, 3 * U, E, 0, st))) return rc;
    const dim3 ew((N * U + 255) / 256), b256(256);
    for (int ts = 0; ts < len; ++ts) {
        const float* hprev = ts == 0 ? t->g_zero : g.H + (size_t)(ts - 1) * U;
        const int64_t ldh = ts == 0 ? U : (int64_t)len * U;
        if ((rc = launch_gemm_f32(hprev, ldh, Rw, 3 * U, nullptr, t->g_G, 2 * U, N, 2 * U, U, 0, st))) return rc;
        hipLaunchKernelGGL(k_gru_gates, ew, b256, 0, st, g.A1, t->g_G, hprev, ldh, g.AZR, g.Z, g.R, g.RH, N, U, len, ts);
        if ((rc = launch_gemm_f32(g.RH + (size_t)ts * U, (int64_t)len * U, Rw + 2 * U, 3 * U, nullptr, t->g_Gh, U, N, U, U, 0, st))) return rc;
        hipLaunchKernelGGL(k_gru_update, ew, b256, 0, st, g.A1, t->g_Gh, hprev, ldh, g.Z, g.HH, g.H, N, U, len, ts);
    }
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// BPTT of one GRU from the gradient of its final state (`up`, row stride ld_up); accumulates into the embedding gradient
// BPTT of several GRUs of the SAME length as ONE persistent launch (k_recur_bwd): transposed recurrent weights in fragment order -
// rows [reset | update] of the second operand in the order of its [d a_r | d a_z] tile - and the gate gradients written
// straight into the keras [z | r | h] columns of each GRU's dA
int gru_backward_launch_multi(rl4rs_simtrain* t, int N, GruSave* const* gs, const float* const* ups, int64_t ld_up, int n, hipStream_t st) {
    const int U = t->c.hidden_units, len = gs[0]->len;
    RecurTrainBwd b;
    memset(&b, 0, sizeof(b));
    b.N = N; b.L = len; b.S = n; b.Hd = U; b.ld_up = ld_up; b.hard = 1;
    b.ld_g = 3 * U; b.ld_c = 3 * U;
    int rc;
    for (int k = 0; k < n; ++k) {
        GruSave& g = *gs[k];
        const float* Rw = t->params + t->off[g.pk + 1];
        if ((rc = launch_pack_frag(Rw + 2 * U, 3 * U, 0, U, U, 1, g.pkc, st))) return rc;                                   // U_h^T
        if ((rc = launch_pack_frag_slice(Rw + U, 3 * U, 0, U, U, 1, g.pkg, 2 * U, 0, st))) return rc;                       // U_r^T
        if ((rc = launch_pack_frag_slice(Rw, 3 * U, 0, U, U, 1, g.pkg, 2 * U, U, st))) return rc;                           // U_z^T
        b.R[k] = g.R; b.U[k] = g.Z; b.C[k] = g.HH; b.H[k] = g.H; b.att[k] = nullptr;
        b.up_last[k] = ups[k]; b.up_all[k] = nullptr; b.wcT[k] = g.pkc; b.wgT[k] = g.pkg;
        b.du[k] = g.dA; b.dr[k] = g.dA + U; b.dc[k] = g.dA + 2 * U;
    }
    return launch_recur_train_bwd(b, st);
}

// use_dA != NULL: the recurrence already ran (gru_backward_launch_multi) and left the gate gradients there - only the
// parameter gradients and the embedding scatter remain
int gru_backward(rl4rs_simtrain* t, int N, GruSave& g, const float* up, int64_t ld_up, float* use_dA, hipStream_t st, const GruScratch& sc) {
    const int E = t->c.emb_size, U = t->c.hidden_units, H = t->c.category_hash_size, len = g.len;
    const float* K = t->params + t->off[g.pk];
    const float* Rw = t->params + t->off[g.pk + 1];
    float* gK = t->grad + t->off[g.pk];
    float* gR = t->grad + t->off[g.pk + 1];
    float* gb = t->grad + t->off[g.pk + 2];
    int rc;
    const dim3 ew((N * U + 255) / 256), b256(256);
    float* dA = use_dA ? use_dA : t->g_dA;
    if (!use_dA) {
    hipLaunchKernelGGL(k_transpose, dim3((U * 2 * U + 255) / 256), b256, 0, st, Rw, (int64_t)3 * U, U, 2 * U, t->g_uzrT);        // [2U, U]
    hipLaunchKernelGGL(k_transpose, dim3((U * U + 255) / 256), b256, 0, st, Rw + 2 * U, (int64_t)3 * U, U, U, t->g_uhT);        // [U, U]
    const float* dh_a = nullptr;
    const float* dh_b = nullptr;
    for (int ts = len - 1; ts >= 0; --ts) {
        const float* hprev = ts == 0 ? t->g_zero : g.H + (size_t)(ts - 1) * U;
        const int64_t ldh = ts == 0 ? U : (int64_t)len * U;
        hipLaunchKernelGGL(k_gru_bwd_pre, ew, b256, 0, st, dh_a, dh_b, ts == len - 1 ? up : (const float*)nullptr, ld_up, t->g_dh, g.Z, g.HH,
                           t->g_dA, N, U, len, ts);
        if ((rc = launch_gemm_f32(t->g_dA + (size_t)ts * 3 * U + 2 * U, (int64_t)len * 3 * U, t->g_uhT, U, nullptr, t->g_drh, U, N, U, U, 0, st)))
            return rc;
        hipLaunchKernelGGL(k_gru_bwd_mid, ew, b256, 0, st, t->g_dh, t->g_drh, hprev, ldh, g.Z, g.R, g.HH, g.AZR, t->g_dA, t->g_dhp, N, U,
                           len, ts);
        if (ts > 0) {
            if ((rc = launch_gemm_f32(t->g_dA + (size_t)ts * 3 * U, (int64_t)len * 3 * U, t->g_uzrT, U, nullptr, t->g_dhg, U, N, U, 2 * U, 0, st)))
                return rc;
        }
        dh_a = t->g_dhp;
        dh_b = t->g_dhg;
    }
    }
    // parameter gradients over all (row, step) samples
    const int Ns = N * len;
    hipLaunchKernelGGL(k_shift_prev, dim3((Ns * U + 255) / 256), b256, 0, st, g.H, sc.hprev, N, U, len);
    const TrainCtx cx = {t->chunk, sc.part, sc.wt, sc.part_cap};
    st_tn_cs(cx, st, g.X, E, E, dA, 3 * U, 3 * U, Ns, gK, gb);
    st_tn(cx, st, sc.hprev, U, U, dA, 3 * U, 2 * U, Ns, sc.tmpw);                                   // [U, 2U]
    RL4RS_HIP_TRY(hipMemcpy2DAsync(gR, (size_t)3 * U * 4, sc.tmpw, (size_t)2 * U * 4, (size_t)2 * U * 4, U, hipMemcpyDeviceToDevice, st));
    st_tn(cx, st, g.RH, U, U, dA + 2 * U, 3 * U, U, Ns, sc.tmpw);                                      // [U, U]
    RL4RS_HIP_TRY(hipMemcpy2DAsync(gR + 2 * U, (size_t)3 * U * 4, sc.tmpw, (size_t)U * 4, (size_t)U * 4, U, hipMemcpyDeviceToDevice, st));
    if ((rc = st_back(cx, st, dA, 3 * U, 3 * U, K, 3 * U, E, sc.dX, E, Ns))) return rc;
    hipLaunchKernelGGL(k_emb_flatten_bwd, dim3((N + 3) / 4), b256, 0, st, g.ids, N, len, H, E, sc.dX, (int64_t)len * E,
                       t->grad + t->off[g.emb]);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_simtrain_destroy(rl4rs_simtrain* t) {
    if (!t) return RL4RS_OK;
    for (void* q : t->owned) (void)hipFree(q);
    if (t->side) { (void)hipStreamDestroy(t->side); (void)hipEventDestroy(t->ev_fork); (void)hipEventDestroy(t->ev_mid); (void)hipEventDestroy(t->ev_join); }
    delete t;
    return RL4RS_OK;
}

int rl4rs_simtrain_create(const rl4rs_simnet_cfg* c, const rl4rs_simnet_weights* w, int32_t max_batch, void* stream,
                          rl4rs_simtrain** out) {
    RL4RS_REQUIRE(c && w && out && max_batch > 0, "simtrain_create: bad argument");
    RL4RS_REQUIRE(c->algo >= RL4RS_SIMNET_DNN && c->algo <= RL4RS_SIMNET_LSTM,
                  "simtrain: algo must be 1 (dnn), 2 (widedeep) or 3 (lstm), got %d", c->algo);
    RL4RS_REQUIRE(c->emb_size > 0 && c->hidden_units > 0 && c->dense_feature_num > 0 && c->category_feature_num > 0 &&
                  c->category_hash_size > 0 && c->class_num >= 2 && c->class_num <= 8 && c->seq_num >= 1 && c->seq_num <= 4 &&
                  c->maxlen >= 1, "simtrain: bad sizes");
    const bool wd = c->algo == RL4RS_SIMNET_WIDEDEEP, ls = c->algo == RL4RS_SIMNET_LSTM;
    RL4RS_REQUIRE(w->cat_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->out_w && w->out_b,
                  "simtrain_create: weights missing");
    RL4RS_REQUIRE(ls || (w->fc_w && w->fc_b), "simtrain_create: fc weights missing");
    RL4RS_REQUIRE(wd || (w->obs_w && w->obs_b), "simtrain_create: obs weights missing");
    RL4RS_REQUIRE(!(wd || ls) || w->seq_emb, "simtrain_create: seq_emb missing");
    if (ls) {
        RL4RS_REQUIRE(w->cat_gru_kernel && w->cat_gru_recurrent && w->cat_gru_bias, "simtrain_create: category GRU weights missing");
        for (int s2 = 0; s2 < c->seq_num; ++s2)
            RL4RS_REQUIRE(w->seq_gru_kernel[s2] && w->seq_gru_recurrent[s2] && w->seq_gru_bias[s2],
                          "simtrain_create: GRU weights of sequence input %d missing", s2);
    }
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t E = c->emb_size, U = c->hidden_units, H = c->category_hash_size, Dn = c->dense_feature_num, K = c->class_num;
    const int64_t S = c->seq_num, Cn = c->category_feature_num;
    rl4rs_simtrain* t = new rl4rs_simtrain();
    t->c = *c;
    t->max_batch = max_batch;
    t->chunk = 512;
    t->nz = (max_batch + t->chunk - 1) / t->chunk;
    t->adam_t = 0;
    t->OD = wd ? (int)(256 + U + Cn * E) : 256;
    // FCK = width of the concat that feeds the first dense layer above the branches (fc for dnn / widedeep, obs for lstm)
    t->FCK = wd ? (int)(S * E) : (ls ? (int)(S * U + 2 * U + Cn * E) : (int)(E + U));
    int64_t sizes[SP_COUNT] = {H * E, (wd || ls) ? H * E : 0, Dn * U, U, U * U, U, ls ? 0 : (int64_t)t->FCK * 256, ls ? 0 : 256,
                               wd ? 0 : (ls ? (int64_t)t->FCK * 256 : 256 * 256), wd ? 0 : 256, (int64_t)t->OD * K, K};
    const float* src[SP_COUNT] = {w->cat_emb, w->seq_emb, w->dense_w1, w->dense_b1, w->dense_w2, w->dense_b2, w->fc_w, w->fc_b,
                                  w->obs_w, w->obs_b, w->out_w, w->out_b};
    for (int i = SP_GRU0; i < SP_COUNT; ++i) { sizes[i] = 0; src[i] = nullptr; }
    const int n_gru = ls ? 1 + (int)S : 0;
    for (int g = 0; g < n_gru; ++g) {
        sizes[SP_GRU0 + 3 * g] = E * 3 * U; sizes[SP_GRU0 + 3 * g + 1] = U * 3 * U; sizes[SP_GRU0 + 3 * g + 2] = 3 * U;
        src[SP_GRU0 + 3 * g] = g == 0 ? w->cat_gru_kernel : w->seq_gru_kernel[g - 1];
        src[SP_GRU0 + 3 * g + 1] = g == 0 ? w->cat_gru_recurrent : w->seq_gru_recurrent[g - 1];
        src[SP_GRU0 + 3 * g + 2] = g == 0 ? w->cat_gru_bias : w->seq_gru_bias[g - 1];
    }
    int64_t o = 0;
    for (int i = 0; i < SP_COUNT; ++i) { t->off[i] = o; t->size[i] = sizes[i]; o += sizes[i]; }
    t->n_params = o;
    int rc = RL4RS_OK;
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) t->owned.push_back(*dst);
        return r;
    };
#define ST_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_simtrain_destroy(t); return rc; } } while (0)
#define ST_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
        rl4rs_simtrain_destroy(t); return RL4RS_EHIP; } } while (0)
    ST_FAIL(al(&t->params, t->n_params));
    ST_FAIL(al(&t->grad, t->n_params));
    ST_FAIL(al(&t->adam_m, t->n_params));
    ST_FAIL(al(&t->adam_v, t->n_params));
    for (int i = 0; i < SP_COUNT; ++i)
        if (sizes[i]) ST_HIP(hipMemcpyAsync(t->params + t->off[i], src[i], (size_t)sizes[i] * 4, hipMemcpyHostToDevice, st));
    ST_HIP(hipMemsetAsync(t->adam_m, 0, (size_t)t->n_params * 4, st));
    ST_HIP(hipMemsetAsync(t->adam_v, 0, (size_t)t->n_params * 4, st));
    const size_t B = max_batch;
    ST_FAIL(al(&t->feat, B * t->FCK));
    ST_FAIL(al(&t->h1, B * U));
    ST_FAIL(al(&t->h1d, B * U));
    ST_FAIL(al(&t->h2, B * U));
    ST_FAIL(al(&t->a1, B * 256));
    ST_FAIL(al(&t->obs, B * t->OD));
    ST_FAIL(al(&t->logits, B * K));
    ST_FAIL(al(&t->d_logits, B * K));
    ST_FAIL(al(&t->d_obs, B * t->OD));
    ST_FAIL(al(&t->d_a, B * 256));
    ST_FAIL(al(&t->d_feat, B * t->FCK));
    ST_FAIL(al(&t->d_h1, B * U));
    ST_FAIL(al(&t->d_h2, B * U));
    int64_t wmax = Dn * U;
    if ((int64_t)t->FCK * 256 > wmax) wmax = (int64_t)t->FCK * 256;
    if (256 * 256 > wmax) wmax = 256 * 256;
    if ((int64_t)t->OD * K > wmax) wmax = (int64_t)t->OD * K;
    if (E * 3 * U > wmax) wmax = E * 3 * U;
    if (U * 3 * U > wmax) wmax = U * 3 * U;
    const int64_t maxlen_any = ls ? (c->maxlen > Cn ? c->maxlen : Cn) : 1;
    const int nz_all = (int)((B * maxlen_any + t->chunk - 1) / t->chunk);       // the GRU gradients reduce over N * len samples
    ST_FAIL(al(&t->wt, wmax));
    ST_FAIL(al(&t->part, (size_t)(nz_all > t->nz ? nz_all : t->nz) * wmax));
    t->part_cap = (size_t)(nz_all > t->nz ? nz_all : t->nz) * wmax;
    for (int g = 0; g < 5; ++g) memset(&t->gru[g], 0, sizeof(GruSave));
    if (ls) {
        for (int g = 0; g < n_gru; ++g) {
            GruSave& q = t->gru[g];
            q.len = g == 0 ? (int)Cn : c->maxlen;
            q.emb = g == 0 ? SP_CAT_EMB : SP_SEQ_EMB;
            q.pk = SP_GRU0 + 3 * g;
            const size_t n = B * q.len;
            ST_FAIL(al(&q.X, n * E)); ST_FAIL(al(&q.A1, n * 3 * U)); ST_FAIL(al(&q.H, n * U)); ST_FAIL(al(&q.Z, n * U));
            ST_FAIL(al(&q.R, n * U)); ST_FAIL(al(&q.HH, n * U)); ST_FAIL(al(&q.RH, n * U)); ST_FAIL(al(&q.AZR, n * 2 * U));
            ST_FAIL(al(&q.pkg, 2 * U * U)); ST_FAIL(al(&q.pkc, U * U)); ST_FAIL(al(&q.dA, n * 3 * U));
        }
        const size_t nm = B * maxlen_any;
        ST_FAIL(al(&t->g_dA, nm * 3 * U)); ST_FAIL(al(&t->g_dX, nm * E)); ST_FAIL(al(&t->g_hprev, nm * U));
        ST_FAIL(al(&t->g_G, B * 2 * U)); ST_FAIL(al(&t->g_Gh, B * U)); ST_FAIL(al(&t->g_dh, B * U)); ST_FAIL(al(&t->g_dhp, B * U));
        ST_FAIL(al(&t->g_dhg, B * U)); ST_FAIL(al(&t->g_drh, B * U)); ST_FAIL(al(&t->g_zero, B * U));
        ST_FAIL(al(&t->g_uzrT, 2 * U * U)); ST_FAIL(al(&t->g_uhT, U * U)); ST_FAIL(al(&t->g_tmpw, U * 2 * U));
        ST_HIP(hipMemsetAsync(t->g_zero, 0, B * U * 4, st));
        {   // second stream + its scratch (only GRU gradient chains run there: widest reduction [E x 3U] / [U x 3U])
            const int64_t wmax2 = (E > U ? E : U) * 3 * U;
            ST_FAIL(al(&t->g_dX2, nm * E)); ST_FAIL(al(&t->g_hprev2, nm * U)); ST_FAIL(al(&t->g_tmpw2, U * 2 * U));
            ST_FAIL(al(&t->wt2, wmax2)); ST_FAIL(al(&t->part2, (size_t)nz_all * wmax2));
            t->part_cap2 = (size_t)nz_all * wmax2;
            ST_HIP(hipStreamCreateWithFlags(&t->side, hipStreamNonBlocking));
            ST_HIP(hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming));
            ST_HIP(hipEventCreateWithFlags(&t->ev_mid, hipEventDisableTiming));
            ST_HIP(hipEventCreateWithFlags(&t->ev_join, hipEventDisableTiming));
        }
        {
            float* p; ST_FAIL(al(&p, B)); t->g_iota = reinterpret_cast<int32_t*>(p);
            std::vector<int32_t> io(B);
            for (size_t i = 0; i < B; ++i) io[i] = (int32_t)i;
            ST_HIP(hipMemcpyAsync(t->g_iota, io.data(), B * 4, hipMemcpyHostToDevice, st));
            ST_HIP(hipStreamSynchronize(st));          // io is a local
        }
    }
    ST_FAIL(al(&t->loss_rows, B));
    ST_FAIL(al(&t->lr_dummy, 4));
    {
        float* m;
        ST_FAIL(al(&m, (B * U + 3) / 4 + 1));
        t->mask1 = reinterpret_cast<uint8_t*>(m);
        ST_FAIL(al(&m, (B * U + 3) / 4 + 1));
        t->mask2 = reinterpret_cast<uint8_t*>(m);
    }
    ST_HIP(hipStreamSynchronize(st));
#undef ST_HIP
#undef ST_FAIL
    *out = t;
    return RL4RS_OK;
}

int rl4rs_simtrain_params(rl4rs_simtrain* t, float** params_dev, float** grad_dev, int64_t* count) {
    RL4RS_REQUIRE(t, "simtrain_params: null handle");
    if (params_dev) *params_dev = t->params;
    if (grad_dev) *grad_dev = t->grad;
    if (count) *count = t->n_params;
    return RL4RS_OK;
}

// the keep-masks (uint8 [N, hidden_units] each) the last rl4rs_simtrain_grad / _step drew for the two tower layers
int rl4rs_simtrain_masks(rl4rs_simtrain* t, uint8_t** mask1_dev, uint8_t** mask2_dev) {
    RL4RS_REQUIRE(t && mask1_dev && mask2_dev, "simtrain_masks: null argument");
    *mask1_dev = t->mask1;
    *mask2_dev = t->mask2;
    return RL4RS_OK;
}

// forward (training mode) + loss + backward into the handle's flat gradient buffer; loss_dev[0] = mean loss (may be NULL).
// seq: seq_num device pointers of int32 [N, maxlen] (widedeep; ignored for dnn, whose model never reads its sequences).
int rl4rs_simtrain_grad(rl4rs_simtrain* t, int32_t N, const float* dense, const int32_t* cat, const int32_t* const* seq,
                        const int32_t* labels, float dropout_rate, uint32_t seed, uint32_t step, float* loss_dev, void* stream) {
    RL4RS_REQUIRE(t && dense && cat && labels && N > 0 && N <= t->max_batch, "simtrain_grad: bad argument (N=%d, max_batch=%d)", N,
                  t ? t->max_batch : -1);
    RL4RS_REQUIRE(dropout_rate >= 0.f && dropout_rate < 1.f, "simtrain_grad: dropout_rate must be in [0, 1)");
    const bool wd = t->c.algo == RL4RS_SIMNET_WIDEDEEP, ls = t->c.algo == RL4RS_SIMNET_LSTM;
    if (wd || ls) {
        RL4RS_REQUIRE(seq, "simtrain_grad: widedeep / lstm need the sequence inputs");
        for (int s = 0; s < t->c.seq_num; ++s) RL4RS_REQUIRE(seq[s], "simtrain_grad: sequence input %d is NULL", s);
    }
    hipStream_t st = (hipStream_t)stream;
    const int E = t->c.emb_size, U = t->c.hidden_units, H = t->c.category_hash_size, Dn = t->c.dense_feature_num;
    const int Cn = t->c.category_feature_num, K = t->c.class_num, S = t->c.seq_num, L = t->c.maxlen;
    const int FCK = t->FCK, OD = t->OD;
    float* P = t->params;
    float* G = t->grad;
    const int64_t* o = t->off;
    int rc;
    auto ew = [](int n) { return dim3((n + 255) / 256); };
    const dim3 g4((N + 3) / 4), b256(256);
    // where the (dropped-out) dense-tower output lives, and its gradient
    float* tower_out = wd ? t->obs + 256 : (ls ? t->feat + S * U : t->feat + E);
    const int tower_ld = wd ? OD : FCK;
    float* d_tower = wd ? t->d_obs + 256 : (ls ? t->d_feat + S * U : t->d_feat + E);
    // ---- forward
    if ((rc = launch_gemm_f32(dense, Dn, P + o[SP_DW1], U, P + o[SP_DB1], t->h1, U, N, U, Dn, 1, st))) return rc;
    RL4RS_HIP_TRY(hipMemcpyAsync(t->h1d, t->h1, (size_t)N * U * 4, hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(k_dropout, ew(N * U), b256, 0, st, t->h1d, t->mask1, N * U, U, dropout_rate, seed, step, 0u);
    if ((rc = launch_gemm_f32(t->h1d, U, P + o[SP_DW2], U, P + o[SP_DB2], t->h2, U, N, U, U, 1, st))) return rc;
    RL4RS_HIP_TRY(hipMemcpyAsync(t->d_h2, t->h2, (size_t)N * U * 4, hipMemcpyDeviceToDevice, st));     // d_h2 as scratch: dropped copy
    hipLaunchKernelGGL(k_dropout, ew(N * U), b256, 0, st, t->d_h2, t->mask2, N * U, U, dropout_rate, seed, step, 1u);
    RL4RS_HIP_TRY(hipMemcpy2DAsync(tower_out, (size_t)tower_ld * 4, t->d_h2, (size_t)U * 4, (size_t)U * 4, N, hipMemcpyDeviceToDevice, st));
    if (wd) {
        for (int s = 0; s < S; ++s)
            hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, seq[s], N, L, H, E, P + o[SP_SEQ_EMB], t->feat, (int64_t)FCK, s * E);
        if ((rc = launch_gemm_f32(t->feat, FCK, P + o[SP_FC_W], 256, P + o[SP_FC_B], t->obs, OD, N, 256, FCK, 1, st))) return rc;
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, cat, N, Cn, H, E, P + o[SP_CAT_EMB], t->obs, (int64_t)OD, 256 + U);
    } else if (ls) {
        // [GRU finals of the sequences | tower | GRU final of the category embeddings | Flatten(category emb)] -> obs (lstm.py:31-36)
        for (int g = 0; g <= S; ++g) t->gru[g].ids = g == 0 ? cat : seq[g - 1];
        const bool pers = gru_persistent(t, N, t->gru[0].len) && gru_persistent(t, N, t->gru[1].len);
        if (pers) {
            // persistent recurrence kernels: the category GRU alone (its own length), the sequence GRUs together in ONE launch - the
            // two launches (32 + 64 workgroups at batch 256) and their input chains side by side on two streams
            const bool two = t->side != nullptr && g_dientrain_fork;
            hipStream_t s2nd = two ? t->side : st;
            if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_fork, st)); RL4RS_HIP_TRY(hipStreamWaitEvent(t->side, t->ev_fork, 0)); }
            GruSave* one[1] = {&t->gru[0]};
            if ((rc = gru_forward_multi(t, N, one, 1, s2nd))) return rc;
            GruSave* seqs[4];
            for (int s2 = 0; s2 < S; ++s2) seqs[s2] = &t->gru[1 + s2];
            if ((rc = gru_forward_multi(t, N, seqs, S, st))) return rc;
            if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_join, t->side)); RL4RS_HIP_TRY(hipStreamWaitEvent(st, t->ev_join, 0)); }
        }
        for (int g = 0; g <= S; ++g) {
            if (!pers && (rc = gru_forward(t, N, t->gru[g], st))) return rc;
            const int off = g == 0 ? S * U + U : (g - 1) * U;
            const GruSave& q = t->gru[g];
            RL4RS_HIP_TRY(hipMemcpy2DAsync(t->feat + off, (size_t)FCK * 4, q.H + (size_t)(q.len - 1) * U, (size_t)q.len * U * 4, (size_t)U * 4,
                                           N, hipMemcpyDeviceToDevice, st));
        }
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, cat, N, Cn, H, E, P + o[SP_CAT_EMB], t->feat, (int64_t)FCK, S * U + 2 * U);
        if ((rc = launch_gemm_f32(t->feat, FCK, P + o[SP_OBS_W], 256, P + o[SP_OBS_B], t->obs, 256, N, 256, FCK, 1, st))) return rc;
    } else {
        hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, cat, N, Cn, H, E, P + o[SP_CAT_EMB], t->feat, (int64_t)FCK, 0);
        if ((rc = launch_gemm_f32(t->feat, FCK, P + o[SP_FC_W], 256, P + o[SP_FC_B], t->a1, 256, N, 256, FCK, 1, st))) return rc;
        if ((rc = launch_gemm_f32(t->a1, 256, P + o[SP_OBS_W], 256, P + o[SP_OBS_B], t->obs, 256, N, 256, 256, 1, st))) return rc;
    }
    if ((rc = launch_gemm_f32(t->obs, OD, P + o[SP_OUT_W], K, P + o[SP_OUT_B], t->logits, K, N, K, OD, 0, st))) return rc;
    hipLaunchKernelGGL(k_bce_softmax, ew(N), b256, 0, st, t->logits, labels, N, K, t->d_logits, t->loss_rows);
    if (loss_dev) hipLaunchKernelGGL(k_mean, dim3(1), b256, 0, st, t->loss_rows, N, loss_dev);
    RL4RS_LAUNCH_CHECK();
    // ---- backward
    const int nz = (N + t->chunk - 1) / t->chunk;
    auto tn = [&](const float* A, int lda, int M, const float* B, int ldb, int Nc, float* dst) {      // dst = A^T B over samples
        int tiles = ((M + 31) / 32) * ((Nc + 31) / 32);
        hipLaunchKernelGGL(k_gemm_tn, dim3((tiles + 3) / 4, nz), b256, 0, st, A, lda, M, B, ldb, Nc, N, t->chunk, nz == 1 ? dst : t->part, (float*)nullptr);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((M * Nc + 255) / 256), b256, 0, st, t->part, M * Nc, nz, dst);
    };
    auto cs = [&](const float* X, int ld, int Nc, float* dst) {
        hipLaunchKernelGGL(k_colsum, dim3((Nc + 63) / 64, nz), dim3(64), 0, st, X, ld, Nc, N, t->chunk, nz == 1 ? dst : t->part);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((Nc + 255) / 256), b256, 0, st, t->part, Nc, nz, dst);
    };
    auto back = [&](const float* dY, int ldy, int Nout, const float* W, int Kin, float* dX, int ldx) -> int {   // dX = dY W^T
        hipLaunchKernelGGL(k_transpose, ew(Kin * Nout), b256, 0, st, W, (int64_t)Nout, Kin, Nout, t->wt);
        return launch_gemm_f32(dY, ldy, t->wt, Kin, nullptr, dX, ldx, N, Kin, Nout, 0, st);
    };
    RL4RS_HIP_TRY(hipMemsetAsync(G + o[SP_CAT_EMB], 0, (size_t)H * E * 4, st));
    tn(t->obs, OD, OD, t->d_logits, K, K, G + o[SP_OUT_W]);
    cs(t->d_logits, K, K, G + o[SP_OUT_B]);
    if ((rc = back(t->d_logits, K, K, P + o[SP_OUT_W], OD, t->d_obs, OD))) return rc;
    if (wd) {
        // 'simulator_obs' = [ELU(fc(pooled sequences)) | tower | Flatten(category emb)]: three gradient slices of d_obs
        hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, t->d_obs, (int64_t)OD, t->obs, (int64_t)OD, (const uint8_t*)nullptr, 0.f, N * 256, 256);
        tn(t->feat, FCK, FCK, t->d_obs, OD, 256, G + o[SP_FC_W]);
        cs(t->d_obs, OD, 256, G + o[SP_FC_B]);
        if ((rc = back(t->d_obs, OD, 256, P + o[SP_FC_W], FCK, t->d_feat, FCK))) return rc;
        RL4RS_HIP_TRY(hipMemsetAsync(G + o[SP_SEQ_EMB], 0, (size_t)H * E * 4, st));
        for (int s = 0; s < S; ++s)
            hipLaunchKernelGGL(k_emb_mean_bwd, g4, b256, 0, st, seq[s], N, L, H, E, t->d_feat + s * E, (int64_t)FCK, G + o[SP_SEQ_EMB]);
        hipLaunchKernelGGL(k_emb_flatten_bwd, g4, b256, 0, st, cat, N, Cn, H, E, t->d_obs + 256 + U, (int64_t)OD, G + o[SP_CAT_EMB]);
    } else if (ls) {
        hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, t->d_obs, (int64_t)256, t->obs, (int64_t)256, (const uint8_t*)nullptr, 0.f, N * 256, 256);
        tn(t->feat, FCK, FCK, t->d_obs, 256, 256, G + o[SP_OBS_W]);
        cs(t->d_obs, 256, 256, G + o[SP_OBS_B]);
        if ((rc = back(t->d_obs, 256, 256, P + o[SP_OBS_W], FCK, t->d_feat, FCK))) return rc;
        RL4RS_HIP_TRY(hipMemsetAsync(G + o[SP_SEQ_EMB], 0, (size_t)H * E * 4, st));
        hipLaunchKernelGGL(k_emb_flatten_bwd, g4, b256, 0, st, cat, N, Cn, H, E, t->d_feat + S * U + 2 * U, (int64_t)FCK, G + o[SP_CAT_EMB]);
        const bool pers = gru_persistent(t, N, t->gru[0].len) && gru_persistent(t, N, t->gru[1].len);
        const GruScratch sc_main = {t->g_dX, t->g_hprev, t->g_tmpw, t->wt, t->part, t->part_cap};
        const GruScratch sc_side = {t->g_dX2, t->g_hprev2, t->g_tmpw2, t->wt2, t->part2, t->part_cap2};
        const bool two = pers && t->side != nullptr && g_dientrain_fork;
        hipStream_t s2nd = two ? t->side : st;
        if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_fork, st)); RL4RS_HIP_TRY(hipStreamWaitEvent(t->side, t->ev_fork, 0)); }
        if (pers) {
            // second stream: the category GRU's BPTT launch and its gradient chain; first stream: the sequence GRUs' launch
            GruSave* one[1] = {&t->gru[0]};
            const float* up0[1] = {t->d_feat + S * U + U};
            if ((rc = gru_backward_launch_multi(t, N, one, up0, (int64_t)FCK, 1, s2nd))) return rc;
            GruSave* seqs[4];
            const float* ups[4];
            for (int s2 = 0; s2 < S; ++s2) { seqs[s2] = &t->gru[1 + s2]; ups[s2] = t->d_feat + s2 * U; }
            if ((rc = gru_backward_launch_multi(t, N, seqs, ups, (int64_t)FCK, S, st))) return rc;
            if (two) RL4RS_HIP_TRY(hipEventRecord(t->ev_mid, st));           // the sequence GRUs' gate gradients exist from here on
        }
        for (int g = 0; g <= S; ++g) {
            const int off = g == 0 ? S * U + U : (g - 1) * U;
            // chains: category GRU and the LAST sequence GRU on the second stream (the latter behind the sequence launch), the rest here
            const bool on_side = two && (g == 0 || (g == S && S > 1));
            if (on_side && g == S) RL4RS_HIP_TRY(hipStreamWaitEvent(t->side, t->ev_mid, 0));
            if ((rc = gru_backward(t, N, t->gru[g], t->d_feat + off, (int64_t)FCK, pers ? t->gru[g].dA : (float*)nullptr, on_side ? t->side : st,
                                   on_side ? sc_side : sc_main))) return rc;
        }
        if (two) { RL4RS_HIP_TRY(hipEventRecord(t->ev_join, t->side)); RL4RS_HIP_TRY(hipStreamWaitEvent(st, t->ev_join, 0)); }
    } else {
        hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, t->d_obs, (int64_t)256, t->obs, (int64_t)256, (const uint8_t*)nullptr, 0.f, N * 256, 256);
        tn(t->a1, 256, 256, t->d_obs, 256, 256, G + o[SP_OBS_W]);
        cs(t->d_obs, 256, 256, G + o[SP_OBS_B]);
        if ((rc = back(t->d_obs, 256, 256, P + o[SP_OBS_W], 256, t->d_a, 256))) return rc;
        hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, t->d_a, (int64_t)256, t->a1, (int64_t)256, (const uint8_t*)nullptr, 0.f, N * 256, 256);
        tn(t->feat, FCK, FCK, t->d_a, 256, 256, G + o[SP_FC_W]);
        cs(t->d_a, 256, 256, G + o[SP_FC_B]);
        if ((rc = back(t->d_a, 256, 256, P + o[SP_FC_W], FCK, t->d_feat, FCK))) return rc;
        hipLaunchKernelGGL(k_emb_mean_bwd, g4, b256, 0, st, cat, N, Cn, H, E, t->d_feat, (int64_t)FCK, G + o[SP_CAT_EMB]);
    }
    // dense tower: d_tower is the gradient of the dropped-out second layer output
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, d_tower, (int64_t)tower_ld, t->h2, (int64_t)U, t->mask2, dropout_rate, N * U, U);
    tn(t->h1d, U, U, d_tower, tower_ld, U, G + o[SP_DW2]);
    cs(d_tower, tower_ld, U, G + o[SP_DB2]);
    if ((rc = back(d_tower, tower_ld, U, P + o[SP_DW2], U, t->d_h1, U))) return rc;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, t->d_h1, (int64_t)U, t->h1, (int64_t)U, t->mask1, dropout_rate, N * U, U);
    tn(dense, Dn, Dn, t->d_h1, U, U, G + o[SP_DW1]);
    cs(t->d_h1, U, U, G + o[SP_DB1]);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// one optimiser step (keras Adam defaults: lr 1e-3, beta 0.9 / 0.999, epsilon 1e-7)
int rl4rs_simtrain_step(rl4rs_simtrain* t, int32_t N, const float* dense, const int32_t* cat, const int32_t* const* seq,
                        const int32_t* labels, float lr, float beta1, float beta2, float eps, float dropout_rate, uint32_t seed,
                        uint32_t step, float* loss_dev, void* stream) {
    int rc = rl4rs_simtrain_grad(t, N, dense, cat, seq, labels, dropout_rate, seed, step, loss_dev, stream);
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
