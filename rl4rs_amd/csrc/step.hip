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
    hipStream_t copy_stream;   // rl4rs_env_step_record_host: early device-to-host copies run here, beside the scorer's kernels
    hipEvent_t ev_ready, ev_copied;
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

// rl4rs_set_host_mirror(1): the head GEMM's epilogue writes the observation to the pinned host block itself.  OFF by default:
// measured on one box, 2 alternating pairs of 12 episode-batches (profiles/r06h_host_mirror.txt): 13.10 / 13.16 ms with the copy
// engine, 13.20 / 13.21 ms with the mirror - stores to host memory drain no faster than the copy engine moves the same 4 MB, and
// the GEMM's waves hold their CUs while they do.
int g_host_mirror = 0;

int ensure_copy_stream(rl4rs_stepper* s) {
    if (!s->copy_stream) {
        RL4RS_HIP_TRY(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
        RL4RS_HIP_TRY(hipEventCreateWithFlags(&s->ev_ready, hipEventDisableTiming));
        RL4RS_HIP_TRY(hipEventCreateWithFlags(&s->ev_copied, hipEventDisableTiming));
    }
    return RL4RS_OK;
}

// the constant outputs of a transition in one launch: done flags, and the zero reward of a step on which none is due
__global__ void k_step_tail(uint8_t* done, uint8_t v, double* zero_reward, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        if (done) done[i] = v;
        if (zero_reward) zero_reward[i] = 0.0;
    }
}

// everything after the act: observation, reward (when due), done, packed obs-side mask.  `on_obs` runs right after the
// observation forward has been enqueued (the record form sends the observation home from there on a reward step, beside the
// reward forward).
struct NoHook { int operator()() const { return RL4RS_OK; } };
template <typename Hook = NoHook>
int after_act(rl4rs_stepper* s, int cur_before, float* obs, double* reward, uint8_t* done, uint32_t* mask_bits, void* stream,
              Hook on_obs = Hook()) {
    hipStream_t st = (hipStream_t)stream;
    rl4rs_env* e = s->env;
    const int B = s->cfg.batch_size;
    int rc;
    if (s->cfg.is_seq && cur_before > 0 && cur_before % s->cfg.page_items == 0) {
        // first act of a page: the second sequence input (items of the previous pages, seqslate.py:107-108) changed - on every page
        // but the first (prev_actions[:0] is the [0] the episode started with: nothing to re-encode)
        for (int q = 1; q < s->seq_num; ++q)
            if ((rc = scorer_encode(s, q, s->seq1, B, stream))) return rc;
    }
    if ((rc = scorer_forward(s, B, 1, s->dense, s->cat, obs, nullptr, stream))) return rc;
    if ((rc = on_obs())) return rc;
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

// ---- record form: everything a host-returning caller needs from one transition, packed for ONE device-to-host copy

// observation row of the d3rlpy mask mode (slate.py:270-277 / seqslate.py:18-23): float64 [obs (256) | masked_actions | cur_steps]
__global__ void k_record_d3rl(const float* obs, int obs_dim, const int32_t* prev, int T, int c0, int ncols, int cur, double* out, int B) {
    const int w = obs_dim + ncols + 1;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * w) return;
    const int b = (int)(i / w), c = (int)(i - (int64_t)b * w);
    double v;
    if (c < obs_dim) v = (double)obs[(size_t)b * obs_dim + c];
    else if (c < obs_dim + ncols) v = (double)prev[(size_t)b * T + c0 + (c - obs_dim)];
    else v = (double)cur;
    out[i] = v;
}
// click probabilities of an env's complete-state rows in slate order (simulator_info_fetch, slate.py:299-301)
__global__ void k_record_click(const float* probs, const float* p_last, int m, float* out, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (m + 1)) return;
    const int b = i / (m + 1), j = i - b * (m + 1);
    out[i] = j < m ? probs[(size_t)b * m + j] : p_last[b];
}
// status words: [0] the env's sticky bad-action flag, [1] the scorer's fp16-range flag (read AND cleared, like rl4rs_dien_status)
__global__ void k_record_status(const int32_t* env_err, int32_t* range_flag, int32_t* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        out[0] = env_err ? *env_err : 0;
        out[1] = range_flag ? atomicExch(range_flag, 0) : 0;
    }
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
    if (s->ev_ready) (void)hipEventDestroy(s->ev_ready);
    if (s->ev_copied) (void)hipEventDestroy(s->ev_copied);
    if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
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

// ---- reference-shaped (host-returning) transition ---------------------------------------------------------------------
// The same transition as rl4rs_env_step_discrete / _conti with every output written into ONE device record (byte offsets from
// rl4rs_stepper_record_layout), host-visible part first: the caller brings the whole transition back with a single copy into
// pinned memory and one wait, instead of one blocking copy per returned object (obs, reward, mask, flags ...).
int rl4rs_stepper_record_layout(rl4rs_stepper* s, uint32_t want, int32_t conti, rl4rs_step_record* L) {
    RL4RS_REQUIRE(s && L, "stepper_record_layout: null argument");
    RL4RS_REQUIRE((want & ~(uint32_t)RL4RS_STEP_WANT_ALL) == 0, "stepper_record_layout: unknown want bits 0x%x", want);
    RL4RS_REQUIRE(!((want & RL4RS_STEP_WANT_CLICK_P) && s->n_complete < 2), "stepper_record_layout: click_p needs n_complete >= 2");
    const int64_t B = s->cfg.batch_size, A = s->cfg.action_size, W = (A + 31) / 32;
    int32_t od32 = 256;
    if (!s->dien) { int rc0 = rl4rs_simnet_obs_dim(s->simnet, &od32); if (rc0) return rc0; }
    const int64_t OD = od32;
    const int ncols = s->cfg.is_seq ? s->cfg.page_items : s->cfg.max_steps;
    int64_t off = 0;
    auto take = [&](int64_t bytes) { const int64_t o = off; off = (off + bytes + 63) & ~(int64_t)63; return o; };
    memset(L, 0xff, sizeof(*L));                                   // every offset -1 = absent
    L->status = take(8);
    L->reward = take(B * 8);
    L->done = take(B);
    L->chosen = take(B * 4);
    const bool d3rl = (want & RL4RS_STEP_WANT_D3RL_OBS) != 0;
    if (d3rl) L->obs_d3rl = take(B * (OD + ncols + 1) * 8); else L->obs = take(B * OD * 4);
    if (want & RL4RS_STEP_WANT_MASK_BITS) L->mask_bits = take(B * W * 4);
    if (want & RL4RS_STEP_WANT_CLICK_P) L->click_p = take(B * s->n_complete * 4);
    if (want & RL4RS_STEP_WANT_OFFLINE_ACTION) L->offline_action = take(conti ? B * s->cfg.action_emb_size * 8 : B * 4);
    // the int64 mask is by far the largest part (B * A * 8 bytes) and depends on the act alone: last of the host part, so that
    // rl4rs_env_step_record_host can send it home on its own while the scorer runs and bring the rest back as one prefix
    if (want & RL4RS_STEP_WANT_MASK_I64) L->mask_i64 = take(B * A * 8);
    L->host_bytes = off;
    if (d3rl) L->obs = take(B * OD * 4);                           // float32 activations: device-side scratch only in this mode
    L->total_bytes = off;
    L->obs_dim = (int32_t)OD;
    L->d3rl_cols = (int32_t)(OD + ncols + 1);
    return RL4RS_OK;
}

// `observe`: no transition - the record of the state the env is in (after a reset: what RecSimBase.sample returns, base.py:172-175);
// reward / done / chosen are then left alone and `actions_dev` is unused.
static int step_record(rl4rs_stepper* s, bool observe, const void* actions_dev, int32_t action_kind, uint32_t want, void* record_dev,
                       void* record_host, void* stream) {
    RL4RS_REQUIRE(s && (observe || actions_dev) && record_dev, "env_step_record: null argument");
    RL4RS_REQUIRE(action_kind >= 0 && action_kind <= 2, "env_step_record: action_kind must be 0 (int32 ids), 1 (float32) or 2 (float64 embeddings)");
    RL4RS_REQUIRE(!(observe && (want & RL4RS_STEP_WANT_CLICK_P)), "env_observe_record: click_p belongs to a reward step");
    rl4rs_step_record L;
    int rc = rl4rs_stepper_record_layout(s, want, action_kind != 0, &L);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    char* R = reinterpret_cast<char*>(record_dev);
    const int B = s->cfg.batch_size;
    float* obs = reinterpret_cast<float*>(R + L.obs);
    int32_t* chosen = reinterpret_cast<int32_t*>(R + L.chosen);
    const int cur = rl4rs_env_cur_steps(s->env) - (observe ? 1 : 0);      // `cur + 1` below = the step counter the record describes
    if (observe) {
        rc = RL4RS_OK;
    } else if (action_kind == 0) {
        RL4RS_HIP_TRY(hipMemcpyAsync(chosen, actions_dev, (size_t)B * 4, hipMemcpyDeviceToDevice, st));
        rc = rl4rs_env_act_discrete(s->env, reinterpret_cast<const int32_t*>(actions_dev), stream);
    } else {
        rc = rl4rs_env_act_conti(s->env, actions_dev, action_kind == 2 ? 1 : 0, chosen, stream);
    }
    if (rc) return rc;
    const int reward_step = observe ? 0 : rl4rs_env_is_reward_step(s->env);
    // the observation-side mask is a function of the env state the act just left (slate.py:90-97); nothing below changes that state
    if (L.mask_i64 >= 0 && (rc = rl4rs_env_obs_mask(s->env, R + L.mask_i64, 2, stream))) return rc;
    char* H = reinterpret_cast<char*>(record_host);
    int64_t prefix = L.host_bytes;
    bool early = false;
    if (H && L.mask_i64 >= 0) {
        if ((rc = ensure_copy_stream(s))) return rc;
        RL4RS_HIP_TRY(hipEventRecord(s->ev_ready, st));
        RL4RS_HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_ready, 0));
        RL4RS_HIP_TRY(hipMemcpyAsync(H + L.mask_i64, R + L.mask_i64, (size_t)(L.host_bytes - L.mask_i64), hipMemcpyDeviceToHost, s->copy_stream));
        RL4RS_HIP_TRY(hipEventRecord(s->ev_copied, s->copy_stream));
        prefix = L.mask_i64;
        early = true;
    }
    // the observation the caller gets: float32 activations, or the float64 row of the d3rlpy mode assembled from them
    const int64_t obs_off = L.obs_d3rl >= 0 ? L.obs_d3rl : L.obs;
    const int64_t obs_bytes = L.obs_d3rl >= 0 ? (int64_t)B * L.d3rl_cols * 8 : (int64_t)B * L.obs_dim * 4;
    bool obs_sent = false;
    // Steps without a reward forward (and resets): the observation is the LAST thing computed, so a device-to-host copy of its
    // 4 MB could only start when the GPU has nothing left to do (85 us at PCIe rate per step, GPU idle).  The head GEMM's epilogue
    // writes it to the pinned host block itself, tile by tile while the GEMM runs (k_gemm_h16's mirror destination): what is
    // left for the copy engine is the few KB around it.  (Reward steps overlap the copy with the reward forward instead; the
    // float64 d3rlpy rows are assembled by a kernel of their own and keep the copy.)
    bool mirror_armed = false;
    if (H && g_host_mirror && s->dien && reward_step != 1 && L.obs_d3rl < 0) {
        void* dp = nullptr;
        if (hipHostGetDevicePointer(&dp, H + L.obs, 0) == hipSuccess && dp) {
            rl4rs::dien_set_obs_mirror(s->dien, reinterpret_cast<float*>(dp));
            mirror_armed = true;
        } else {
            (void)hipGetLastError();          // not mapped for the device: the copy path below serves it
        }
    }
    auto finish_obs = [&]() -> int {
        if (mirror_armed && rl4rs::dien_obs_mirror_used(s->dien)) obs_sent = true;      // already on its way home
        if (L.obs_d3rl >= 0) {
            // masked_actions: all of prev_actions (slate.py:100-104) or the current page's columns (seqslate.py:18-23), POST-act step counter
            const int cur_after = cur + 1, T = s->cfg.max_steps, P = s->cfg.page_items;
            int c0 = 0, ncols = T;
            if (s->cfg.is_seq) {
                const int page_init = cur_after / P * P, page_end = (page_init + P - 1 < T - 1) ? page_init + P - 1 : T - 1;
                c0 = page_end + 1 - P; ncols = P;
            }
            void* pp; int64_t nb;
            int rc2 = rl4rs_env_buffer(s->env, RL4RS_BUF_PREV_ACTIONS, &pp, &nb);
            if (rc2) return rc2;
            const int64_t total = (int64_t)B * L.d3rl_cols;
            hipLaunchKernelGGL(k_record_d3rl, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, obs, L.obs_dim,
                               reinterpret_cast<const int32_t*>(pp), T, c0, ncols, cur_after, reinterpret_cast<double*>(R + L.obs_d3rl), B);
            RL4RS_LAUNCH_CHECK();
        }
        if (H && reward_step == 1) {
            // a reward step: 4 - 9 MB of observation would otherwise wait for the whole reward forward and then cross PCIe
            // with the GPU idle; it leaves now, on the copy stream, beside the reward forward
            int rc2 = ensure_copy_stream(s);
            if (rc2) return rc2;
            RL4RS_HIP_TRY(hipEventRecord(s->ev_ready, st));
            RL4RS_HIP_TRY(hipStreamWaitEvent(s->copy_stream, s->ev_ready, 0));
            RL4RS_HIP_TRY(hipMemcpyAsync(H + obs_off, R + obs_off, (size_t)obs_bytes, hipMemcpyDeviceToHost, s->copy_stream));
            RL4RS_HIP_TRY(hipEventRecord(s->ev_copied, s->copy_stream));
            obs_sent = true;
            early = true;
        }
        return RL4RS_OK;
    };
    if (observe) {
        if ((rc = scorer_forward(s, B, 1, s->dense, s->cat, obs, nullptr, stream))) return rc;
        if ((rc = finish_obs())) return rc;
        if (L.mask_bits >= 0 && (rc = rl4rs_env_obs_mask(s->env, R + L.mask_bits, 4, stream))) return rc;
    } else if ((rc = after_act(s, cur, obs, reinterpret_cast<double*>(R + L.reward), reinterpret_cast<uint8_t*>(R + L.done),
                               L.mask_bits >= 0 ? reinterpret_cast<uint32_t*>(R + L.mask_bits) : nullptr, stream, finish_obs))) {
        return rc;
    }
    if (L.click_p >= 0 && reward_step == 1) {
        const int m = s->n_complete - 1;
        hipLaunchKernelGGL(k_record_click, dim3((B * (m + 1) + 255) / 256), dim3(256), 0, st, s->probs, s->p_last, m,
                           reinterpret_cast<float*>(R + L.click_p), B);
        RL4RS_LAUNCH_CHECK();
    }
    if (L.offline_action >= 0 && cur + 1 < s->cfg.max_steps) {
        // the logged action of the NEXT step (slate.py:152-161), so a replay loop needs no extra device round trip for it
        if ((rc = rl4rs_env_offline_action(s->env, action_kind == 0 ? reinterpret_cast<int32_t*>(R + L.offline_action) : nullptr,
                                           action_kind != 0 ? reinterpret_cast<double*>(R + L.offline_action) : nullptr, stream))) return rc;
    }
    {
        void* ep; int64_t nb; int32_t* rf = nullptr;
        if ((rc = rl4rs_env_buffer(s->env, RL4RS_BUF_ERROR_FLAG, &ep, &nb))) return rc;
        if (s->dien && (rc = rl4rs_dien_status_word(s->dien, &rf))) return rc;
        hipLaunchKernelGGL(k_record_status, dim3(1), dim3(64), 0, st, reinterpret_cast<const int32_t*>(ep), rf, reinterpret_cast<int32_t*>(R + L.status));
        RL4RS_LAUNCH_CHECK();
    }
    if (H) {
        if (obs_sent) {     // the prefix around the observation that is already on its way
            RL4RS_HIP_TRY(hipMemcpyAsync(H, R, (size_t)obs_off, hipMemcpyDeviceToHost, st));
            const int64_t tail = obs_off + ((obs_bytes + 63) & ~(int64_t)63);
            if (prefix > tail) RL4RS_HIP_TRY(hipMemcpyAsync(H + tail, R + tail, (size_t)(prefix - tail), hipMemcpyDeviceToHost, st));
        } else {
            RL4RS_HIP_TRY(hipMemcpyAsync(H, R, (size_t)prefix, hipMemcpyDeviceToHost, st));
        }
        if (early) RL4RS_HIP_TRY(hipStreamWaitEvent(st, s->ev_copied, 0));     // one wait on `stream` covers every copy (the copy stream is in order)
    }
    return RL4RS_OK;
}

int rl4rs_set_host_mirror(int32_t on) {
    g_host_mirror = on ? 1 : 0;
    return RL4RS_OK;
}

int rl4rs_env_step_record(rl4rs_stepper* s, const void* actions_dev, int32_t action_kind, uint32_t want, void* record_dev, void* stream) {
    return step_record(s, false, actions_dev, action_kind, want, record_dev, nullptr, stream);
}

int rl4rs_env_step_record_host(rl4rs_stepper* s, const void* actions_dev, int32_t action_kind, uint32_t want, void* record_dev,
                               void* record_host, void* stream) {
    RL4RS_REQUIRE(record_host, "env_step_record_host: null host block");
    return step_record(s, false, actions_dev, action_kind, want, record_dev, record_host, stream);
}

int rl4rs_env_observe_record_host(rl4rs_stepper* s, int32_t conti, uint32_t want, void* record_dev, void* record_host, void* stream) {
    return step_record(s, true, nullptr, conti ? 2 : 0, want, record_dev, record_host, stream);
}

}  // extern "C"
