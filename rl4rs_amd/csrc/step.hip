// One batched env transition as ONE library call: RecSimBase._step (rl4rs/env/base.py:157-170) =
//   samples.act(action)            slate.py:193-214 / seqslate.py:92-126
//   next_obs = obs_fn(state)       slate.py:244-279   (simulator_obs layer of the scorer)
//   reward   = forward(model, ..)  slate.py:281-308 / seqslate.py:136-160 (only when a reward is due)
//   done     = step >= max_steps-1 base.py:165-168
// composed from the same entry points the host facade uses one by one (rl4rs_env_act_*, rl4rs_dien_forward / rl4rs_simnet_forward,
// rl4rs_env_build_complete_rows, *_head_prob, rl4rs_env_reward_split, rl4rs_env_obs_mask): identical kernels in identical order,
// so the results are bit-identical to the composed path (tests/test_gpu_facade.py::test_fused_step_is_bit_identical).  Nothing is
// allocated and nothing synchronises inside: all scratch belongs to the binding created by rl4rs_env_attach_scorer.
#include "common.hpp"

using namespace rl4rs;

struct rl4rs_stepper {
    rl4rs_env* env;
    rl4rs_dien* dien;          // exactly one of dien / simnet is set
    rl4rs_simnet* simnet;
    const int32_t* slots;      // [seq_num, B] cache slot of every env row per sequence input (caller-owned device memory)
    int32_t seq_num;
    rl4rs_env_cfg cfg;
    int n_complete;
    float* probs;              // [B * (n_complete - 1)] click probabilities of the complete-state rows
    float* p_last;             // [B] probability of the state row just scored (= the last complete-state row)
    const float* dense; const int32_t* cat; const int32_t* seq1; const float* c_dense; const int32_t* c_cat;
};

namespace {

int scorer_forward(rl4rs_stepper* s, int R, int group, const float* dense, const int32_t* cat, float* obs, float* prob, void* stream) {
    return s->dien ? rl4rs_dien_forward(s->dien, R, group, dense, cat, s->slots, obs, prob, stream)
                   : rl4rs_simnet_forward(s->simnet, R, group, dense, cat, s->slots, obs, prob, stream);
}
int scorer_head_prob(rl4rs_stepper* s, int R, const float* obs, float* prob, void* stream) {
    return s->dien ? rl4rs_dien_head_prob(s->dien, R, obs, prob, stream) : rl4rs_simnet_head_prob(s->simnet, R, obs, prob, stream);
}
int scorer_encode(rl4rs_stepper* s, int q, const int32_t* ids, int n, void* stream) {
    return s->dien ? rl4rs_dien_encode(s->dien, q, ids, n, 0, stream) : rl4rs_simnet_encode(s->simnet, q, ids, n, 0, stream);
}

// the constant outputs of a transition in one launch: done flags, and the zero reward of a step on which none is due
__global__ void k_step_tail(uint8_t* done, uint8_t v, double* zero_reward, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (done) done[i] = v;
        if (zero_reward) zero_reward[i] = 0.0;
    }
}

// everything after the act: observation, reward (when due), done, packed obs-side mask
int after_act(rl4rs_stepper* s, int cur_before, float* obs, double* reward, uint8_t* done, uint32_t* mask_bits, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    rl4rs_env* e = s->env;
    const int B = s->cfg.batch_size;
    int rc;
    if (s->cfg.is_seq && cur_before % s->cfg.page_items == 0) {
        // first act of a page: the second sequence input (items of the previous pages, seqslate.py:107-108) changed
        for (int q = 1; q < s->seq_num; ++q)
            if ((rc = scorer_encode(s, q, s->seq1, B, stream))) return rc;
    }
    if ((rc = scorer_forward(s, B, 1, s->dense, s->cat, obs, nullptr, stream))) return rc;
    double* zero_reward = nullptr;
    if (reward) {
        if (rl4rs_env_is_reward_step(e) == 1) {
            // the state row just scored IS the last complete-state row (slate.py:205-212 vs :119-130): score n - 1 rows per env
            const int m = s->n_complete - 1;
            if (m > 0) {
                if ((rc = rl4rs_env_build_complete_rows(e, m, stream))) return rc;
                if ((rc = scorer_forward(s, B * m, m, s->c_dense, s->c_cat, nullptr, s->probs, stream))) return rc;
            }
            if ((rc = scorer_head_prob(s, B, obs, s->p_last, stream))) return rc;
            if ((rc = rl4rs_env_reward_split(e, m > 0 ? s->probs : s->p_last, m > 0 ? s->p_last : nullptr, reward, stream))) return rc;
        } else {
            zero_reward = reward;
        }
    }
    if (done || zero_reward) {
        hipLaunchKernelGGL(k_step_tail, dim3((B + 255) / 256), dim3(256), 0, st, done, (uint8_t)(cur_before >= s->cfg.max_steps - 1 ? 1 : 0),
                           zero_reward, B);
        RL4RS_LAUNCH_CHECK();
    }
    if (mask_bits && (rc = rl4rs_env_obs_mask(e, mask_bits, 4, stream))) return rc;
    return RL4RS_OK;
}

int attach(rl4rs_env* env, rl4rs_dien* dien, rl4rs_simnet* simnet, const int32_t* slots_dev, int32_t seq_num, rl4rs_stepper** out) {
    RL4RS_REQUIRE(env && (dien || simnet) && slots_dev && out && seq_num >= 1 && seq_num <= 4, "env_attach_scorer: bad argument");
    rl4rs_stepper* s = new rl4rs_stepper();
    memset(s, 0, sizeof(*s));
    s->env = env; s->dien = dien; s->simnet = simnet; s->slots = slots_dev; s->seq_num = seq_num;
    int rc = rl4rs_env_get_cfg(env, &s->cfg);
    if (rc) { delete s; return rc; }
    s->n_complete = rl4rs_env_complete_rows(env);
    const size_t B = (size_t)s->cfg.batch_size;
    if ((rc = dev_alloc(&s->probs, B * (size_t)(s->n_complete > 1 ? s->n_complete - 1 : 1))) || (rc = dev_alloc(&s->p_last, B))) {
        if (s->probs) (void)hipFree(s->probs);
        delete s;
        return rc;
    }
    void* p;
    int64_t nb;
#define BUF(which, field, type) if ((rc = rl4rs_env_buffer(env, which, &p, &nb))) { rl4rs_stepper_destroy(s); return rc; } s->field = reinterpret_cast<type>(p)
    BUF(RL4RS_BUF_DENSE, dense, const float*);
    BUF(RL4RS_BUF_CATEGORY, cat, const int32_t*);
    BUF(RL4RS_BUF_SEQ1, seq1, const int32_t*);
    BUF(RL4RS_BUF_C_DENSE, c_dense, const float*);
    BUF(RL4RS_BUF_C_CATEGORY, c_cat, const int32_t*);
#undef BUF
    *out = s;
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_env_attach_scorer(rl4rs_env* env, rl4rs_dien* net, const int32_t* slots_dev, int32_t seq_num, rl4rs_stepper** out) {
    return attach(env, net, nullptr, slots_dev, seq_num, out);
}
int rl4rs_env_attach_simnet(rl4rs_env* env, rl4rs_simnet* net, const int32_t* slots_dev, int32_t seq_num, rl4rs_stepper** out) {
    return attach(env, nullptr, net, slots_dev, seq_num, out);
}

int rl4rs_stepper_destroy(rl4rs_stepper* s) {
    if (!s) return RL4RS_OK;
    if (s->probs) (void)hipFree(s->probs);
    if (s->p_last) (void)hipFree(s->p_last);
    delete s;
    return RL4RS_OK;
}

int rl4rs_env_step_discrete(rl4rs_stepper* s, const int32_t* actions_dev, float* obs_dev, double* reward_dev, uint8_t* done_dev,
                            uint32_t* mask_bits_dev, void* stream) {
    RL4RS_REQUIRE(s && actions_dev && obs_dev, "env_step_discrete: null argument");
    const int cur = rl4rs_env_cur_steps(s->env);
    int rc = rl4rs_env_act_discrete(s->env, actions_dev, stream);
    if (rc) return rc;
    return after_act(s, cur, obs_dev, reward_dev, done_dev, mask_bits_dev, stream);
}

int rl4rs_env_step_conti(rl4rs_stepper* s, const void* actions_dev, int is_f64, int32_t* chosen_dev, float* obs_dev,
                         double* reward_dev, uint8_t* done_dev, uint32_t* mask_bits_dev, void* stream) {
    RL4RS_REQUIRE(s && actions_dev && obs_dev, "env_step_conti: null argument");
    const int cur = rl4rs_env_cur_steps(s->env);
    int rc = rl4rs_env_act_conti(s->env, actions_dev, is_f64, chosen_dev, stream);
    if (rc) return rc;
    return after_act(s, cur, obs_dev, reward_dev, done_dev, mask_bits_dev, stream);
}

}  // extern "C"
