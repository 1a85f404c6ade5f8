// Trainable raw-state policy (rl4rs/nets/rllib/rllib_rawstate_model.py:25-86 + mask wrapper rllib_mask_model.py:67-115) with
// RLlib's A2C / PPO losses: the forward of rl4rs_rawpolicy on raw (unpacked) weights, the loss core of the FC mask policy
// (policy_row_loss), and a hand-written backward through the two heads, the 256-wide context layer, the dense tower and the
// mean-pooled embeddings of both tables.  Included at the end of policy.hip after simtrain.hpp (reduction helpers).
// Flat parameter / gradient layout:
//   [ cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | ctx_w | ctx_b | head_w 256 x (A+1) = [out_w | value_w] |
//     head_b (A+1) ]
#pragma once

namespace rl4rs {

// loss + gradient wrt [logits | value] of one sample per wave, from the head outputs `ext`
__global__ __launch_bounds__(256) void k_rawpolicy_loss(PolDims d, int N, const float* __restrict__ ext, const uint32_t* __restrict__ mask,
                                                        LossArgs L, float* __restrict__ dOut, float4* __restrict__ terms) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s_out = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * d.AE;
    float* s_d = s_out + d.AE;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const uint32_t* mrow = mask ? mask + (size_t)n * d.W : nullptr;
    float mx = -3.4028235e38f;
    for (int a = lane; a < d.AE; a += 64) {
        float v = ext[(size_t)n * d.AE + a];
        if (a < d.A) {
            bool ok = mrow ? ((mrow[a >> 5] >> (a & 31)) & 1u) : true;
            if (!ok) v = v + (-3.4028235e38f);
            mx = fmaxf(mx, v);
        }
        s_out[a] = v;
    }
    mx = wave_max(mx);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float se = 0.f;
    for (int a = lane; a < d.A; a += 64) se += expf(s_out[a] - mx);
    const float lse = mx + logf(wave_sum(se));
    const float4 tm = policy_row_loss(d, L, s_out, lse, n, lane, s_d, dOut);
    if (lane == 0) terms[n] = tm;
}

}  // namespace rl4rs

enum { RT_CAT_EMB = 0, RT_SEQ_EMB, RT_DW1, RT_DB1, RT_DW2, RT_DB2, RT_CTX_W, RT_CTX_B, RT_HEAD_W, RT_HEAD_B, RT_COUNT };

struct rl4rs_rawtrain {
    rl4rs_rawpolicy_cfg c;
    PolDims d;
    int F;
    int64_t n_params, off[RT_COUNT];
    TrainCtx cx;
    float *params, *grad, *adam_m, *adam_v, *sumsq;
    float *feat, *h1, *ctx, *ext, *dOut, *d_ctx, *d_feat, *d_h1;
    float4* terms;
    int64_t adam_t;
    std::vector<void*> owned;
};

namespace {

int rawtrain_forward(rl4rs_rawtrain* p, int N, const int32_t* cat, const float* dense, const int32_t* const* seq, hipStream_t st) {
    const int E = p->c.emb_size, U = p->c.hidden_units, H = p->c.category_hash_size, S = p->c.seq_num, Dn = p->c.dense_feature_num;
    const int F = p->F, L = p->c.maxlen, Cn = p->c.category_feature_num, AE = p->d.AE;
    const float* P = p->params;
    const int64_t* o = p->off;
    const dim3 g4((N + 3) / 4), b256(256);
    int rc;
    for (int s = 0; s < S; ++s)
        hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, seq[s], N, L, H, E, P + o[RT_SEQ_EMB], p->feat, (int64_t)F, s * E);
    hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, cat, N, Cn, H, E, P + o[RT_CAT_EMB], p->feat, (int64_t)F, S * E + U);
    RL4RS_LAUNCH_CHECK();
    if ((rc = launch_gemm_f32(dense, Dn, P + o[RT_DW1], U, P + o[RT_DB1], p->h1, U, N, U, Dn, 1, st))) return rc;
    if ((rc = launch_gemm_f32(p->h1, U, P + o[RT_DW2], U, P + o[RT_DB2], p->feat + S * E, F, N, U, U, 1, st))) return rc;
    if ((rc = launch_gemm_f32(p->feat, F, P + o[RT_CTX_W], 256, P + o[RT_CTX_B], p->ctx, 256, N, 256, F, 1, st))) return rc;
    return launch_gemm_f32(p->ctx, 256, P + o[RT_HEAD_W], AE, P + o[RT_HEAD_B], p->ext, AE, N, AE, 256, 0, st);
}

}  // namespace

extern "C" {

int rl4rs_rawtrain_destroy(rl4rs_rawtrain* p) {
    if (!p) return RL4RS_OK;
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return RL4RS_OK;
}

int rl4rs_rawtrain_create(const rl4rs_rawpolicy_cfg* c, const rl4rs_rawpolicy_weights* w, void* stream, rl4rs_rawtrain** out) {
    RL4RS_REQUIRE(c && w && out, "rawtrain_create: null argument");
    RL4RS_REQUIRE(c->emb_size > 0 && c->hidden_units > 0 && c->maxlen >= 1 && c->seq_num >= 1 && c->seq_num <= 4 &&
                  c->category_feature_num >= 1 && c->category_hash_size > 0 && c->dense_feature_num > 0 && c->action_size > 1 &&
                  c->max_rows > 0, "rawtrain_create: bad sizes");
    RL4RS_REQUIRE(w->cat_emb && w->seq_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->ctx_w && w->ctx_b &&
                  w->out_w && w->out_b && w->value_w && w->value_b, "rawtrain_create: weights missing");
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t E = c->emb_size, U = c->hidden_units, H = c->category_hash_size, S = c->seq_num, Dn = c->dense_feature_num;
    const int64_t A = c->action_size, AE = A + 1;
    rl4rs_rawtrain* p = new rl4rs_rawtrain();
    p->c = *c;
    p->d.OD = 256; p->d.HID = 0; p->d.A = (int)A; p->d.AE = (int)AE; p->d.W = (int)((A + 31) / 32);
    p->F = (int)(S * E + U + E);
    p->adam_t = 0;
    const int64_t sizes[RT_COUNT] = {H * E, H * E, Dn * U, U, U * U, U, (int64_t)p->F * 256, 256, 256 * AE, AE};
    int64_t o = 0;
    for (int i = 0; i < RT_COUNT; ++i) { p->off[i] = o; o += sizes[i]; }
    p->n_params = o;
    int rc;
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) p->owned.push_back(*dst);
        return r;
    };
#define RT_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_rawtrain_destroy(p); return rc; } } while (0)
#define RT_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
        rl4rs_rawtrain_destroy(p); return RL4RS_EHIP; } } while (0)
    RT_FAIL(al(&p->params, p->n_params)); RT_FAIL(al(&p->grad, p->n_params)); RT_FAIL(al(&p->adam_m, p->n_params));
    RT_FAIL(al(&p->adam_v, p->n_params)); RT_FAIL(al(&p->sumsq, 4));
    const float* src[8] = {w->cat_emb, w->seq_emb, w->dense_w1, w->dense_b1, w->dense_w2, w->dense_b2, w->ctx_w, w->ctx_b};
    for (int i = 0; i < 8; ++i) RT_HIP(hipMemcpyAsync(p->params + p->off[i], src[i], (size_t)sizes[i] * 4, hipMemcpyHostToDevice, st));
    std::vector<float> hw((size_t)256 * AE), hb(AE);       // [out_w | value_w], [out_b | value_b]
    for (int k = 0; k < 256; ++k) {
        for (int a = 0; a < A; ++a) hw[(size_t)k * AE + a] = w->out_w[(size_t)k * A + a];
        hw[(size_t)k * AE + A] = w->value_w[k];
    }
    for (int a = 0; a < A; ++a) hb[a] = w->out_b[a];
    hb[A] = w->value_b[0];
    RT_HIP(hipMemcpyAsync(p->params + p->off[RT_HEAD_W], hw.data(), hw.size() * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemcpyAsync(p->params + p->off[RT_HEAD_B], hb.data(), hb.size() * 4, hipMemcpyHostToDevice, st));
    RT_HIP(hipMemsetAsync(p->adam_m, 0, (size_t)p->n_params * 4, st));
    RT_HIP(hipMemsetAsync(p->adam_v, 0, (size_t)p->n_params * 4, st));
    const size_t B = c->max_rows;
    RT_FAIL(al(&p->feat, B * p->F)); RT_FAIL(al(&p->h1, B * U)); RT_FAIL(al(&p->ctx, B * 256)); RT_FAIL(al(&p->ext, B * AE));
    RT_FAIL(al(&p->dOut, B * AE)); RT_FAIL(al(&p->d_ctx, B * 256)); RT_FAIL(al(&p->d_feat, B * p->F)); RT_FAIL(al(&p->d_h1, B * U));
    { float* t4; RT_FAIL(al(&t4, B * 4)); p->terms = reinterpret_cast<float4*>(t4); }
    int64_t wmax = (int64_t)p->F * 256;
    if (Dn * U > wmax) wmax = Dn * U;
    if (256 * AE > wmax) wmax = 256 * AE;
    p->cx.chunk = 512;
    RT_FAIL(al(&p->cx.wt, wmax));
    RT_FAIL(al(&p->cx.part, (size_t)((B + 511) / 512) * wmax));
    RT_HIP(hipStreamSynchronize(st));
#undef RT_HIP
#undef RT_FAIL
    *out = p;
    return RL4RS_OK;
}

int rl4rs_rawtrain_params(rl4rs_rawtrain* p, float** params_dev, float** grad_dev, int64_t* count) {
    RL4RS_REQUIRE(p, "rawtrain_params: null handle");
    if (params_dev) *params_dev = p->params;
    if (grad_dev) *grad_dev = p->grad;
    if (count) *count = p->n_params;
    return RL4RS_OK;
}

int rl4rs_rawtrain_act(rl4rs_rawtrain* p, int32_t N, const int32_t* cat, const float* dense, const int32_t* const* seq,
                       const uint32_t* mask_bits, uint32_t seed, uint32_t step, int32_t* actions, float* logp, float* value,
                       float* entropy, float* logits, void* stream) {
    RL4RS_REQUIRE(p && cat && dense && seq && actions && N > 0 && N <= p->c.max_rows, "rawtrain_act: bad argument (N=%d, max_rows=%d)", N,
                  p ? p->c.max_rows : -1);
    hipStream_t st = (hipStream_t)stream;
    int rc = rawtrain_forward(p, N, cat, dense, seq, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rawpolicy_head<true>, dim3((N + 3) / 4), dim3(256), (size_t)4 * p->d.AE * 4, st, p->d, N, p->ext, (int64_t)p->d.AE,
                       mask_bits, seed, step, actions, logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_rawtrain_evaluate(rl4rs_rawtrain* p, int32_t N, const int32_t* cat, const float* dense, const int32_t* const* seq,
                            const uint32_t* mask_bits, const int32_t* actions, float* logp, float* value, float* entropy, float* logits,
                            void* stream) {
    RL4RS_REQUIRE(p && cat && dense && seq && actions && N > 0 && N <= p->c.max_rows, "rawtrain_evaluate: bad argument (N=%d, max_rows=%d)",
                  N, p ? p->c.max_rows : -1);
    hipStream_t st = (hipStream_t)stream;
    int rc = rawtrain_forward(p, N, cat, dense, seq, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rawpolicy_head<false>, dim3((N + 3) / 4), dim3(256), (size_t)4 * p->d.AE * 4, st, p->d, N, p->ext, (int64_t)p->d.AE,
                       mask_bits, 0u, 0u, const_cast<int32_t*>(actions), logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// A2C (algo 0) / PPO (algo 1) loss and its gradient into the handle's flat gradient buffer (arguments as rl4rs_policy_loss_grad)
int rl4rs_rawtrain_loss_grad(rl4rs_rawtrain* p, int32_t algo, int32_t N, const int32_t* cat, const float* dense,
                             const int32_t* const* seq, const uint32_t* mask_bits, const int32_t* actions, const float* adv,
                             const float* ret, const float* old_logp, const float* old_value, const float* old_logits, float vf_coeff,
                             float ent_coeff, float clip, float vf_clip, float kl_coeff, float* stats_dev, void* stream) {
    RL4RS_REQUIRE(p && cat && dense && seq && actions && adv && ret && N > 0 && N <= p->c.max_rows,
                  "rawtrain_loss_grad: bad argument (N=%d, max_rows=%d)", N, p ? p->c.max_rows : -1);
    RL4RS_REQUIRE(algo == 0 || (algo == 1 && old_logp && old_value && old_logits), "rawtrain_loss_grad: PPO needs old_* inputs");
    hipStream_t st = (hipStream_t)stream;
    const int E = p->c.emb_size, U = p->c.hidden_units, H = p->c.category_hash_size, S = p->c.seq_num, Dn = p->c.dense_feature_num;
    const int F = p->F, L = p->c.maxlen, Cn = p->c.category_feature_num, AE = p->d.AE;
    const float* P = p->params;
    float* G = p->grad;
    const int64_t* o = p->off;
    int rc = rawtrain_forward(p, N, cat, dense, seq, st);
    if (rc) return rc;
    LossArgs La;
    La.algo = algo; La.vf_coeff = vf_coeff; La.ent_coeff = ent_coeff; La.clip = clip; La.vf_clip = vf_clip; La.kl_coeff = kl_coeff;
    La.scale = algo == 0 ? 1.0f : 1.0f / (float)N;
    La.actions = actions; La.adv = adv; La.ret = ret; La.old_logp = old_logp; La.old_value = old_value; La.old_logits = old_logits;
    hipLaunchKernelGGL(k_rawpolicy_loss, dim3((N + 3) / 4), dim3(256), (size_t)4 * 2 * AE * 4, st, p->d, N, p->ext, mask_bits, La, p->dOut,
                       p->terms);
    const dim3 g4((N + 3) / 4), b256(256);
    auto ew = [](int n) { return dim3((n + 255) / 256); };
    st_tn(p->cx, st, p->ctx, 256, 256, p->dOut, AE, AE, N, G + o[RT_HEAD_W]);
    st_cs(p->cx, st, p->dOut, AE, AE, N, G + o[RT_HEAD_B]);
    if ((rc = st_back(p->cx, st, p->dOut, AE, AE, P + o[RT_HEAD_W], AE, 256, p->d_ctx, 256, N))) return rc;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * 256), b256, 0, st, p->d_ctx, (int64_t)256, p->ctx, (int64_t)256, (const uint8_t*)nullptr, 0.f, N * 256, 256);
    st_tn(p->cx, st, p->feat, F, F, p->d_ctx, 256, 256, N, G + o[RT_CTX_W]);
    st_cs(p->cx, st, p->d_ctx, 256, 256, N, G + o[RT_CTX_B]);
    if ((rc = st_back(p->cx, st, p->d_ctx, 256, 256, P + o[RT_CTX_W], 256, F, p->d_feat, F, N))) return rc;
    RL4RS_HIP_TRY(hipMemsetAsync(G + o[RT_CAT_EMB], 0, (size_t)H * E * 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(G + o[RT_SEQ_EMB], 0, (size_t)H * E * 4, st));
    for (int s = 0; s < S; ++s)
        hipLaunchKernelGGL(k_emb_mean_bwd, g4, b256, 0, st, seq[s], N, L, H, E, p->d_feat + s * E, (int64_t)F, G + o[RT_SEQ_EMB]);
    hipLaunchKernelGGL(k_emb_mean_bwd, g4, b256, 0, st, cat, N, Cn, H, E, p->d_feat + S * E + U, (int64_t)F, G + o[RT_CAT_EMB]);
    // dense tower (inference-mode Dropout: RLlib calls the keras base model without the training flag)
    float* d_tower = p->d_feat + S * E;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, d_tower, (int64_t)F, p->feat + S * E, (int64_t)F, (const uint8_t*)nullptr, 0.f, N * U, U);
    st_tn(p->cx, st, p->h1, U, U, d_tower, F, U, N, G + o[RT_DW2]);
    st_cs(p->cx, st, d_tower, F, U, N, G + o[RT_DB2]);
    if ((rc = st_back(p->cx, st, d_tower, F, U, P + o[RT_DW2], U, U, p->d_h1, U, N))) return rc;
    hipLaunchKernelGGL(k_elu_bwd, ew(N * U), b256, 0, st, p->d_h1, (int64_t)U, p->h1, (int64_t)U, (const uint8_t*)nullptr, 0.f, N * U, U);
    st_tn(p->cx, st, dense, Dn, Dn, p->d_h1, U, U, N, G + o[RT_DW1]);
    st_cs(p->cx, st, p->d_h1, U, U, N, G + o[RT_DB1]);
    if (stats_dev) hipLaunchKernelGGL(k_reduce_terms, dim3(1), dim3(256), 0, st, p->terms, N, stats_dev);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_rawtrain_adam_step(rl4rs_rawtrain* p, float lr, float beta1, float beta2, float eps, float grad_clip, void* stream) {
    RL4RS_REQUIRE(p, "rawtrain_adam_step: null handle");
    hipStream_t st = (hipStream_t)stream;
    p->adam_t += 1;
    const double t = (double)p->adam_t;
    const float lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    if (grad_clip > 0.f) hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(256), 0, st, p->grad, (int)p->n_params, p->sumsq);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((p->n_params + 255) / 256)), dim3(256), 0, st, p->params, p->grad, p->adam_m, p->adam_v,
                       (int)p->n_params, lr_t, beta1, beta2, eps, p->sumsq, grad_clip);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // extern "C"
