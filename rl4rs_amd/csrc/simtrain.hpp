// Supervised training of the 'dnn' and 'widedeep' simulator families on the device: one optimiser step = forward (training mode:
// Dropout(0.2) after each dense-tower layer, utils.py:48-54) + keras binary_crossentropy on the softmax output against
// the one-hot label + backward + Adam - what `model.compile(loss='binary_crossentropy', optimizer='adam')` /
// `model.fit` do in script/supervised_train.py:37-42 for rl4rs/nets/dnn.py and rl4rs/nets/widedeep.py.  Included at the end of policy.hip: it uses
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

__global__ void k_transpose(const float* __restrict__ w, int rows, int cols, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const int r = i / cols, c = i - r * cols;
    wt[(size_t)c * rows + r] = w[i];
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

enum { SP_CAT_EMB = 0, SP_SEQ_EMB, SP_DW1, SP_DB1, SP_DW2, SP_DB2, SP_FC_W, SP_FC_B, SP_OBS_W, SP_OBS_B, SP_OUT_W, SP_OUT_B, SP_COUNT };

struct rl4rs_simtrain {
    rl4rs_simnet_cfg c;
    int64_t n_params;
    int64_t off[SP_COUNT], size[SP_COUNT];       // offsets / sizes in the flat buffers (size 0 = the family has no such array)
    int max_batch, chunk, nz, OD, FCK;           // OD = width of 'simulator_obs', FCK = input width of the fc layer
    float *params, *grad, *adam_m, *adam_v;
    float *feat, *h1, *h1d, *h2, *a1, *obs, *logits;            // activations (feat = input of fc, a1 = its output for dnn)
    float *d_logits, *d_obs, *d_a, *d_feat, *d_h1, *d_h2, *wt, *part, *loss_rows, *lr_dummy;
    uint8_t *mask1, *mask2;
    int64_t adam_t;
    std::vector<void*> owned;
};

extern "C" {

int rl4rs_simtrain_destroy(rl4rs_simtrain* t) {
    if (!t) return RL4RS_OK;
    for (void* q : t->owned) (void)hipFree(q);
    delete t;
    return RL4RS_OK;
}

int rl4rs_simtrain_create(const rl4rs_simnet_cfg* c, const rl4rs_simnet_weights* w, int32_t max_batch, void* stream,
                          rl4rs_simtrain** out) {
    RL4RS_REQUIRE(c && w && out && max_batch > 0, "simtrain_create: bad argument");
    RL4RS_REQUIRE(c->algo == RL4RS_SIMNET_DNN || c->algo == RL4RS_SIMNET_WIDEDEEP,
                  "simtrain: the dnn (1) and widedeep (2) families can be trained on the device (got algo %d)", c->algo);
    RL4RS_REQUIRE(c->emb_size > 0 && c->hidden_units > 0 && c->dense_feature_num > 0 && c->category_feature_num > 0 &&
                  c->category_hash_size > 0 && c->class_num >= 2 && c->class_num <= 8 && c->seq_num >= 1 && c->seq_num <= 4 &&
                  c->maxlen >= 1, "simtrain: bad sizes");
    const bool wd = c->algo == RL4RS_SIMNET_WIDEDEEP;
    RL4RS_REQUIRE(w->cat_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->fc_w && w->fc_b && w->out_w &&
                  w->out_b && (wd ? w->seq_emb != nullptr : (w->obs_w && w->obs_b)), "simtrain_create: weights missing");
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
    t->FCK = wd ? (int)(S * E) : (int)(E + U);
    const int64_t sizes[SP_COUNT] = {H * E, wd ? H * E : 0, Dn * U, U, U * U, U, (int64_t)t->FCK * 256, 256, wd ? 0 : 256 * 256,
                                     wd ? 0 : 256, (int64_t)t->OD * K, K};
    const float* src[SP_COUNT] = {w->cat_emb, w->seq_emb, w->dense_w1, w->dense_b1, w->dense_w2, w->dense_b2, w->fc_w, w->fc_b,
                                  w->obs_w, w->obs_b, w->out_w, w->out_b};
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
    ST_FAIL(al(&t->wt, wmax));
    ST_FAIL(al(&t->part, (size_t)t->nz * wmax));
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
    const bool wd = t->c.algo == RL4RS_SIMNET_WIDEDEEP;
    if (wd) {
        RL4RS_REQUIRE(seq, "simtrain_grad: widedeep needs the sequence inputs");
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
    float* tower_out = wd ? t->obs + 256 : t->feat + E;
    const int tower_ld = wd ? OD : FCK;
    float* d_tower = wd ? t->d_obs + 256 : t->d_feat + E;
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
        hipLaunchKernelGGL(k_gemm_tn, dim3((tiles + 3) / 4, nz), b256, 0, st, A, lda, M, B, ldb, Nc, N, t->chunk, nz == 1 ? dst : t->part);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((M * Nc + 255) / 256), b256, 0, st, t->part, M * Nc, nz, dst);
    };
    auto cs = [&](const float* X, int ld, int Nc, float* dst) {
        hipLaunchKernelGGL(k_colsum, dim3((Nc + 63) / 64, nz), dim3(64), 0, st, X, ld, Nc, N, t->chunk, nz == 1 ? dst : t->part);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((Nc + 255) / 256), b256, 0, st, t->part, Nc, nz, dst);
    };
    auto back = [&](const float* dY, int ldy, int Nout, const float* W, int Kin, float* dX, int ldx) -> int {   // dX = dY W^T
        hipLaunchKernelGGL(k_transpose, ew(Kin * Nout), b256, 0, st, W, Kin, Nout, t->wt);
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
