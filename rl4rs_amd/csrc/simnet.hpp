// The reference's other simulator families (config['algo'] = dnn | widedeep | lstm) behind the 'simulator_obs' /
// 'simulator_reward' contract.  Included at the end of dien.hip: it re-uses that translation unit's recurrent kernel
// (k_recur in GRU mode with keras hard_sigmoid gates), the packed-weight GEMM and the softmax head.
//   topology: rl4rs/nets/dnn.py:31-37, widedeep.py:31-38, lstm.py:31-37, helpers rl4rs/nets/utils.py:7-97
//   keras GRU (utils.py:34,91) -> k_recur: r = keras r, u = keras z (h' = z h + (1-z) hh), candidate = keras h; the
//   x-side of every GRU is a table  emb @ kernel[:, r|z|h] + bias  ([H, 3U], built at create), so a step only adds
//   h @ recurrent on MFMA - the same hoist as the DIEN first GRU.
#pragma once

#include "gather_kernels.hpp"

struct rl4rs_simnet {
    rl4rs_simnet_cfg c;
    int obs_dim, F;
    float *cat_emb, *seq_emb, *dense_w1, *dense_b1, *dense_w2, *dense_b2, *fc_w, *fc_b, *obs_w, *obs_b, *out_w, *out_b;
    float *cat_tab, *cat_wg, *cat_wc;            // lstm: x-side table [H, 3U] (r | u | c, bias folded), packed h-side
    float *seq_tab[4], *seq_wg[4], *seq_wc[4];
    float* seqfeat[4];                           // [max_slots, W]: widedeep W = E (mean embedding), lstm W = U (final state)
    float *feat, *dh, *tmp, *obs_tmp;
    std::vector<void*> owned;
};

namespace {

int sn_upload(rl4rs_simnet* n, float** dst, const float* src, size_t count, hipStream_t st) {
    int rc = dev_alloc(dst, count);
    if (rc) return rc;
    n->owned.push_back(*dst);
    RL4RS_HIP_TRY(hipMemcpyAsync(*dst, src, count * 4, hipMemcpyHostToDevice, st));
    return RL4RS_OK;
}
int sn_alloc(rl4rs_simnet* n, float** dst, size_t count) {
    int rc = dev_alloc(dst, count);
    if (rc) return rc;
    n->owned.push_back(*dst);
    return RL4RS_OK;
}

// keras GRU variables -> k_recur operands.  table = emb @ kernel[:, r|z|h] + bias ; h-side gates [r | z], candidate h.
int sn_prepare_gru(rl4rs_simnet* n, const float* emb_dev, const float* kernel, const float* recurrent, const float* bias,
                   float** tab, float** wg, float** wc, std::vector<std::vector<float>>& keep, hipStream_t st) {
    const int E = n->c.emb_size, U = n->c.hidden_units, H = n->c.category_hash_size;
    std::vector<float> wx((size_t)E * 3 * U), bx(3 * U), ug((size_t)U * 2 * U), uc((size_t)U * U);
    for (int k = 0; k < E; ++k)
        for (int j = 0; j < U; ++j) {
            wx[(size_t)k * 3 * U + j] = kernel[(size_t)k * 3 * U + U + j];             // r
            wx[(size_t)k * 3 * U + U + j] = kernel[(size_t)k * 3 * U + j];             // z -> u
            wx[(size_t)k * 3 * U + 2 * U + j] = kernel[(size_t)k * 3 * U + 2 * U + j]; // h -> c
        }
    for (int j = 0; j < U; ++j) { bx[j] = bias[U + j]; bx[U + j] = bias[j]; bx[2 * U + j] = bias[2 * U + j]; }
    for (int k = 0; k < U; ++k)
        for (int j = 0; j < U; ++j) {
            ug[(size_t)k * 2 * U + j] = recurrent[(size_t)k * 3 * U + U + j];
            ug[(size_t)k * 2 * U + U + j] = recurrent[(size_t)k * 3 * U + j];
            uc[(size_t)k * U + j] = recurrent[(size_t)k * 3 * U + 2 * U + j];
        }
    float *d_wx, *d_bx;
    int rc;
    if ((rc = sn_upload(n, &d_wx, wx.data(), wx.size(), st))) return rc;
    if ((rc = sn_upload(n, &d_bx, bx.data(), bx.size(), st))) return rc;
    if ((rc = sn_alloc(n, tab, (size_t)H * 3 * U))) return rc;
    if ((rc = launch_gemm_f32(emb_dev, E, d_wx, 3 * U, d_bx, *tab, 3 * U, H, 3 * U, E, 0, st))) return rc;
    keep.push_back(std::move(wx)); keep.push_back(std::move(bx));
    keep.push_back(pack_frag(ug.data(), 2 * U, 0, U, 2 * U));
    if ((rc = sn_upload(n, wg, keep.back().data(), keep.back().size(), st))) return rc;
    keep.push_back(pack_frag(uc.data(), U, 0, U, U));
    if ((rc = sn_upload(n, wc, keep.back().data(), keep.back().size(), st))) return rc;
    return RL4RS_OK;
}

// final state of a keras GRU over `len` ids per row -> out[(slot_base + row) * out_ld + out_off + 0..U)
int sn_launch_gru(rl4rs_simnet* n, const float* tab, const float* wg, const float* wc, const int32_t* ids, int rows, int len,
                  float* out, int64_t out_ld, int out_off, int slot_base, hipStream_t st) {
    const int U = n->c.hidden_units;
    RecurArgs a;
    memset(&a, 0, sizeof(a));
    a.n_rows = rows; a.L = len; a.group = 1;
    a.xbase[0] = tab; a.xld = 3 * U; a.xoff = 0; a.xbytes = (int64_t)n->c.category_hash_size * 3 * U * 4;
    a.ids = ids; a.wg[0] = wg; a.wc[0] = wc;
    a.out = out; a.out_ld = out_ld; a.out_off = out_off; a.slot_base = slot_base;
    a.hard_gates = 1; a.final_only = 1;
    size_t smem = (size_t)(2 * 32 * (U + 4) + 32 * (len + 1) + 32) * 4;
    hipLaunchKernelGGL((k_recur<128, false, GRU_U>), dim3((rows + 31) / 32, 1), dim3(256), smem, st, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_simnet_create(const rl4rs_simnet_cfg* c, const rl4rs_simnet_weights* w, void* stream, rl4rs_simnet** out) {
    RL4RS_REQUIRE(c && w && out, "simnet_create: null argument");
    RL4RS_REQUIRE(c->algo >= RL4RS_SIMNET_DNN && c->algo <= RL4RS_SIMNET_LSTM, "simnet: algo must be 1 (dnn), 2 (widedeep) or 3 (lstm), got %d", c->algo);
    RL4RS_REQUIRE(c->emb_size > 0 && c->emb_size % 8 == 0 && c->hidden_units > 0 && c->hidden_units % 32 == 0,
                  "simnet: emb_size must be a multiple of 8, hidden_units a multiple of 32");
    RL4RS_REQUIRE(c->maxlen >= 1 && c->maxlen <= 64, "simnet: maxlen must be in 1..64 (got %d)", c->maxlen);
    RL4RS_REQUIRE(c->seq_num >= 1 && c->seq_num <= 4, "simnet: seq_num must be in 1..4");
    RL4RS_REQUIRE(c->class_num >= 1 && c->class_num <= 8, "simnet: class_num must be in 1..8");
    RL4RS_REQUIRE(c->category_feature_num >= 1 && c->category_feature_num <= 64, "simnet: category_feature_num must be in 1..64");
    RL4RS_REQUIRE(c->max_rows > 0 && c->max_slots > 0 && c->category_hash_size > 0 && c->dense_feature_num > 0, "simnet: bad sizes");
    if (c->algo == RL4RS_SIMNET_LSTM)
        RL4RS_REQUIRE(c->hidden_units == 128 && c->emb_size == 128,
                      "simnet lstm: the recurrent kernel is built for emb_size == hidden_units == 128 (got %d, %d)", c->emb_size, c->hidden_units);
    RL4RS_REQUIRE((int64_t)c->category_hash_size * 3 * c->hidden_units * 4 < (int64_t)0x7fffffff * 2,
                  "simnet: category_hash_size too large for 32-bit buffer offsets");
    RL4RS_REQUIRE(w->cat_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->out_w && w->out_b,
                  "simnet_create: cat_emb / dense tower / out weights missing");
    int ndev = rl4rs_device_count();
    if (ndev <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int E = c->emb_size, U = c->hidden_units, H = c->category_hash_size, S = c->seq_num, Cn = c->category_feature_num;
    const int Dn = c->dense_feature_num, K = c->class_num, L = c->maxlen;
    (void)L;
    rl4rs_simnet* n = new rl4rs_simnet();
    n->c = *c;
    n->cat_emb = n->seq_emb = n->fc_w = n->fc_b = n->obs_w = n->obs_b = nullptr;
    n->cat_tab = n->cat_wg = n->cat_wc = nullptr;
    for (int s = 0; s < 4; ++s) n->seq_tab[s] = n->seq_wg[s] = n->seq_wc[s] = n->seqfeat[s] = nullptr;
    n->obs_dim = c->algo == RL4RS_SIMNET_WIDEDEEP ? 256 + U + Cn * E : 256;
    n->F = c->algo == RL4RS_SIMNET_DNN ? E + U : (c->algo == RL4RS_SIMNET_WIDEDEEP ? S * E : S * U + 2 * U + Cn * E);
    int rc;
    std::vector<std::vector<float>> keep;
    keep.reserve(64);
#define SN_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_simnet_destroy(n); return rc; } } while (0)
#define SN_UP(dst, src, cnt) SN_FAIL(sn_upload(n, &n->dst, (src), (size_t)(cnt), st))
#define SN_PK(dst, src, kk, nn) do { keep.push_back(pack_gemm_weight((src), (nn), (kk), (nn))); \
        SN_FAIL(sn_upload(n, &n->dst, keep.back().data(), keep.back().size(), st)); } while (0)
    SN_UP(cat_emb, w->cat_emb, (size_t)H * E);
    SN_PK(dense_w1, w->dense_w1, Dn, U);
    SN_UP(dense_b1, w->dense_b1, U);
    SN_PK(dense_w2, w->dense_w2, U, U);
    SN_UP(dense_b2, w->dense_b2, U);
    SN_UP(out_w, w->out_w, (size_t)n->obs_dim * K);
    SN_UP(out_b, w->out_b, K);
    if (c->algo != RL4RS_SIMNET_DNN) {
        if (!w->seq_emb) { set_error("simnet_create: seq_emb missing"); rl4rs_simnet_destroy(n); return RL4RS_EINVAL; }
        SN_UP(seq_emb, w->seq_emb, (size_t)H * E);
    }
    if (c->algo != RL4RS_SIMNET_LSTM) {
        if (!w->fc_w || !w->fc_b) { set_error("simnet_create: fc_w / fc_b missing"); rl4rs_simnet_destroy(n); return RL4RS_EINVAL; }
        SN_PK(fc_w, w->fc_w, n->F, 256);
        SN_UP(fc_b, w->fc_b, 256);
    }
    if (c->algo != RL4RS_SIMNET_WIDEDEEP) {
        if (!w->obs_w || !w->obs_b) { set_error("simnet_create: obs_w / obs_b missing"); rl4rs_simnet_destroy(n); return RL4RS_EINVAL; }
        const int Kobs = c->algo == RL4RS_SIMNET_DNN ? 256 : n->F;
        SN_PK(obs_w, w->obs_w, Kobs, 256);
        SN_UP(obs_b, w->obs_b, 256);
    }
    if (c->algo == RL4RS_SIMNET_LSTM) {
        if (!w->cat_gru_kernel || !w->cat_gru_recurrent || !w->cat_gru_bias) {
            set_error("simnet_create: category GRU weights missing"); rl4rs_simnet_destroy(n); return RL4RS_EINVAL;
        }
        SN_FAIL(sn_prepare_gru(n, n->cat_emb, w->cat_gru_kernel, w->cat_gru_recurrent, w->cat_gru_bias, &n->cat_tab, &n->cat_wg,
                               &n->cat_wc, keep, st));
        for (int s = 0; s < S; ++s) {
            if (!w->seq_gru_kernel[s] || !w->seq_gru_recurrent[s] || !w->seq_gru_bias[s]) {
                set_error("simnet_create: GRU weights of sequence input %d missing", s); rl4rs_simnet_destroy(n); return RL4RS_EINVAL;
            }
            SN_FAIL(sn_prepare_gru(n, n->seq_emb, w->seq_gru_kernel[s], w->seq_gru_recurrent[s], w->seq_gru_bias[s], &n->seq_tab[s],
                                   &n->seq_wg[s], &n->seq_wc[s], keep, st));
        }
        if (raise_dyn_smem(reinterpret_cast<const void*>(&k_recur<128, false, GRU_U>), (size_t)(2 * 32 * (U + 4) + 32 * (64 + 1) + 32) * 4) != RL4RS_OK) {
            rl4rs_simnet_destroy(n);
            return RL4RS_EHIP;
        }
    }
    if (c->algo != RL4RS_SIMNET_DNN) {
        const int W = c->algo == RL4RS_SIMNET_WIDEDEEP ? E : U;
        for (int s = 0; s < S; ++s) SN_FAIL(sn_alloc(n, &n->seqfeat[s], (size_t)c->max_slots * W));
    }
    SN_FAIL(sn_alloc(n, &n->feat, (size_t)c->max_rows * n->F));
    SN_FAIL(sn_alloc(n, &n->dh, (size_t)c->max_rows * U));
    SN_FAIL(sn_alloc(n, &n->tmp, (size_t)c->max_rows * 256));
    SN_FAIL(sn_alloc(n, &n->obs_tmp, (size_t)c->max_rows * n->obs_dim));
#undef SN_PK
#undef SN_UP
#undef SN_FAIL
    hipError_t e = hipStreamSynchronize(st);      // host staging buffers die with this frame
    if (e != hipSuccess) {
        set_error("simnet_create: hipStreamSynchronize failed: %s", hipGetErrorString(e));
        rl4rs_simnet_destroy(n);
        return RL4RS_EHIP;
    }
    *out = n;
    return RL4RS_OK;
}

int rl4rs_simnet_destroy(rl4rs_simnet* n) {
    if (!n) return RL4RS_OK;
    for (void* p : n->owned) (void)hipFree(p);
    delete n;
    return RL4RS_OK;
}

int rl4rs_simnet_obs_dim(rl4rs_simnet* n, int32_t* dim) {
    RL4RS_REQUIRE(n && dim, "simnet_obs_dim: null argument");
    *dim = n->obs_dim;
    return RL4RS_OK;
}

int rl4rs_simnet_encode(rl4rs_simnet* n, int32_t s, const int32_t* ids, int32_t cnt, int32_t slot_base, void* stream) {
    RL4RS_REQUIRE(n && ids, "simnet_encode: null argument");
    RL4RS_REQUIRE(s >= 0 && s < n->c.seq_num, "simnet_encode: sequence input %d out of range", s);
    RL4RS_REQUIRE(cnt > 0 && slot_base >= 0 && slot_base + cnt <= n->c.max_slots,
                  "simnet_encode: slots [%d,%d) exceed max_slots=%d", slot_base, slot_base + cnt, n->c.max_slots);
    hipStream_t st = (hipStream_t)stream;
    const int E = n->c.emb_size, U = n->c.hidden_units, L = n->c.maxlen, H = n->c.category_hash_size;
    if (n->c.algo == RL4RS_SIMNET_WIDEDEEP) {
        hipLaunchKernelGGL(k_emb_mean, dim3((cnt + 3) / 4), dim3(256), 0, st, ids, cnt, L, H, E, n->seq_emb,
                           n->seqfeat[s] + (size_t)slot_base * E, (int64_t)E, 0);
        RL4RS_LAUNCH_CHECK();
    } else if (n->c.algo == RL4RS_SIMNET_LSTM) {
        return sn_launch_gru(n, n->seq_tab[s], n->seq_wg[s], n->seq_wc[s], ids, cnt, L, n->seqfeat[s], U, 0, slot_base, st);
    }
    return RL4RS_OK;      // dnn: the model never reads its sequence inputs (dnn.py:33-34)
}

int rl4rs_simnet_head_prob(rl4rs_simnet* n, int32_t R, const float* obs, float* prob, void* stream) {
    RL4RS_REQUIRE(n && obs && prob && R > 0, "simnet_head_prob: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_head_prob, dim3((R + 3) / 4), dim3(256), 0, st, obs, R, n->obs_dim, n->c.class_num, n->out_w, n->out_b, prob);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_simnet_forward(rl4rs_simnet* n, int32_t R, int32_t group, const float* dense, const int32_t* cat,
                         const int32_t* slots, float* obs, float* prob, void* stream) {
    RL4RS_REQUIRE(n && dense && cat && slots, "simnet_forward: null argument");
    RL4RS_REQUIRE(R > 0 && R <= n->c.max_rows, "simnet_forward: R=%d exceeds max_rows=%d", R, n->c.max_rows);
    RL4RS_REQUIRE(group >= 1 && R % group == 0, "simnet_forward: R=%d is not a multiple of group=%d", R, group);
    RL4RS_REQUIRE(obs || prob, "simnet_forward: both outputs are NULL");
    hipStream_t st = (hipStream_t)stream;
    const int E = n->c.emb_size, U = n->c.hidden_units, H = n->c.category_hash_size, S = n->c.seq_num;
    const int Cn = n->c.category_feature_num, Dn = n->c.dense_feature_num, F = n->F, OD = n->obs_dim;
    const int ngroups = R / group;
    float* o = obs ? obs : n->obs_tmp;
    const dim3 g4((R + 3) / 4), b256(256);
    int rc;
    // dense tower (utils.py:48-54; Dropout is inactive at inference)
    if ((rc = launch_gemm_packed(dense, Dn, n->dense_w1, n->dense_b1, n->dh, U, R, U, Dn, 1, st))) return rc;
    if (n->c.algo == RL4RS_SIMNET_DNN) {
        hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, cat, R, Cn, H, E, n->cat_emb, n->feat, (int64_t)F, 0);
        RL4RS_LAUNCH_CHECK();
        if ((rc = launch_gemm_packed(n->dh, U, n->dense_w2, n->dense_b2, n->feat + E, F, R, U, U, 1, st))) return rc;
        if ((rc = launch_gemm_packed(n->feat, F, n->fc_w, n->fc_b, n->tmp, 256, R, 256, F, 1, st))) return rc;
        if ((rc = launch_gemm_packed(n->tmp, 256, n->obs_w, n->obs_b, o, OD, R, 256, 256, 1, st))) return rc;
    } else if (n->c.algo == RL4RS_SIMNET_WIDEDEEP) {
        for (int s = 0; s < S; ++s) {
            hipLaunchKernelGGL(k_gather_slots, g4, b256, 0, st, n->seqfeat[s], E, slots + (size_t)s * ngroups, R, group, n->feat,
                               (int64_t)F, s * E);
            RL4RS_LAUNCH_CHECK();
        }
        if ((rc = launch_gemm_packed(n->feat, F, n->fc_w, n->fc_b, o, OD, R, 256, F, 1, st))) return rc;
        if ((rc = launch_gemm_packed(n->dh, U, n->dense_w2, n->dense_b2, o + 256, OD, R, U, U, 1, st))) return rc;
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, cat, R, Cn, H, E, n->cat_emb, o, (int64_t)OD, 256 + U);
        RL4RS_LAUNCH_CHECK();
    } else {
        for (int s = 0; s < S; ++s) {
            hipLaunchKernelGGL(k_gather_slots, g4, b256, 0, st, n->seqfeat[s], U, slots + (size_t)s * ngroups, R, group, n->feat,
                               (int64_t)F, s * U);
            RL4RS_LAUNCH_CHECK();
        }
        if ((rc = launch_gemm_packed(n->dh, U, n->dense_w2, n->dense_b2, n->feat + S * U, F, R, U, U, 1, st))) return rc;
        if ((rc = sn_launch_gru(n, n->cat_tab, n->cat_wg, n->cat_wc, cat, R, Cn, n->feat, F, S * U + U, 0, st))) return rc;
        hipLaunchKernelGGL(k_emb_flatten, g4, b256, 0, st, cat, R, Cn, H, E, n->cat_emb, n->feat, (int64_t)F, S * U + 2 * U);
        RL4RS_LAUNCH_CHECK();
        if ((rc = launch_gemm_packed(n->feat, F, n->obs_w, n->obs_b, o, OD, R, 256, F, 1, st))) return rc;
    }
    if (prob) {
        hipLaunchKernelGGL(k_head_prob, dim3((R + 3) / 4), dim3(256), 0, st, o, R, OD, n->c.class_num, n->out_w, n->out_b, prob);
        RL4RS_LAUNCH_CHECK();
    }
    return RL4RS_OK;
}

}  // extern "C"
