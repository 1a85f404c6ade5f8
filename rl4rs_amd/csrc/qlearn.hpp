// Offline-RL learner networks and losses (BASELINE configs[4] "BCQ offline RL"): the networks the
// reference trains with d3rlpy on the logged-policy dataset (script/batchrl_trainer.py:34-90), on the device.
//
//   custom encoder  rl4rs/nets/cql/encoder.py:9-67 (CustomVectorEncoder, with_q=True, hidden_units=[256]):
//       h   = relu(fc1(x))                                   x = [obs 256 | prev_actions 9 | cur_step]  (the WHOLE row feeds fc1)
//       enc = fc2([h | emb(x[-M:]) flattened])               emb: Embedding(action_size, 32) of the M = page_items+1 tail ids
//       enc[mask == 0] = 0                                   mask = location_mask[cur_step % 9 // 3], previous actions zeroed,
//                                                            special items zeroed once one was chosen (encoder.py:44-49,61-66)
//   plain encoder   d3rlpy VectorEncoder(hidden_units=[256,256], relu) - what DiscreteCQL gets, the custom factory being
//       commented out there (batchrl_trainer.py:82-86)
//   head            d3rlpy puts nn.Linear(encoder.get_feature_size(), action_size) on every encoder: the Q values of
//       DiscreteMeanQFunction, the logits of DiscreteImitator.
//
// d3rlpy 0.91 is a third-party dependency that is absent here (environment.yml:146); its losses are restated from the
// published algorithms (parity unpinned, checked against torch autograd of the same restatement in the tests):
//   imitation (DiscreteBC, the imitator of DiscreteBCQ):  nll_loss(log_softmax(logits), a) + beta * mean(logits^2)
//   TD (DQN / DoubleDQN):   mean huber(r + gamma * Q_targ(s')[a*] * (1 - terminal) - Q(s)[a]),  huber beta = 1
//       a* = argmax Q(s')                                     (DoubleDQN, DiscreteCQL)
//       a* = argmax (Q(s') - min Q(s')) * [log pi(s') - max log pi(s') > log(action_flexibility)]   (DiscreteBCQ)
//   conservative (DiscreteCQL):  alpha * mean(logsumexp(Q(s)) - Q(s)[a])
// Weight matrices are stored [in, out] (x @ W), flat parameter layout in rl4rs_qnet_create.  Sample-axis reductions use the
// fixed-order helpers of simtrain.hpp, so a step is bit-reproducible.
#pragma once

namespace rl4rs {

__global__ void k_relu_bwd(float* __restrict__ d, int64_t ldd, const float* __restrict__ y, int64_t ldy, int n, int cols) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int row = i / cols, c = i - row * cols;
    if (!(y[(size_t)row * ldy + c] > 0.f)) d[(size_t)row * ldd + c] = 0.f;
}

// cat[n, off + j*ES + e] = emb[id_j][e],  id_j = (long) x[n, D - M + j]      (encoder.py:57)
__global__ void k_q_tail_emb(const float* __restrict__ obs, int N, int D, int M, int A, int ES, const float* __restrict__ emb,
                             float* __restrict__ cat, int64_t ldc, int off, int* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * M * ES) return;
    const int n = i / (M * ES), r = i - n * (M * ES), j = r / ES, e = r - j * ES;
    int id = (int)obs[(size_t)n * D + D - M + j];
    if (id < 0 || id >= A) {            // torch.nn.Embedding raises IndexError
        atomicOr(err, 1);
        id = 0;
    }
    cat[(size_t)n * ldc + off + r] = emb[(size_t)id * ES + e];
}

// d_emb[id][e] = sum over (n, j) with id_j(n) == id of d_cat[n, off + j*ES + e]; one block per table row, fixed summation order
__global__ __launch_bounds__(256) void k_q_tail_emb_bwd(const float* __restrict__ obs, int N, int D, int M, int A, int ES,
                                                        const float* __restrict__ dcat, int64_t ldc, int off, float* __restrict__ demb) {
    __shared__ float part[256];
    const int id = blockIdx.x;
    const int e = threadIdx.x % ES, p = threadIdx.x / ES, np = 256 / ES;      // ES divides 256 (checked at create)
    float s = 0.f;
    for (int n = p; n < N; n += np)
        for (int j = 0; j < M; ++j) {
            int v = (int)obs[(size_t)n * D + D - M + j];
            if (v < 0 || v >= A) v = 0;
            if (v == id) s += dcat[(size_t)n * ldc + off + j * ES + e];
        }
    part[threadIdx.x] = s;
    __syncthreads();
    if (p == 0) {
        float t = 0.f;
        for (int q = 0; q < np; ++q) t += part[q * ES + e];
        demb[(size_t)id * ES + e] = t;
    }
}

// mask rule of the custom encoder (encoder.py:44-49,61-66): one wave per row; zeroes enc where the mask is 0 and keeps the bits
__global__ __launch_bounds__(256) void k_q_mask(const float* __restrict__ obs, int N, int D, int M, int A, int W,
                                                const uint32_t* __restrict__ loc_bits, int n_layers,
                                                const uint32_t* __restrict__ special_bits, float* __restrict__ enc,
                                                uint32_t* __restrict__ bits, int* __restrict__ err) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* orow = obs + (size_t)n * D + D - M;
    const int P = M - 1;
    int layer = ((int)orow[P] % 9) / 3;                     // x_mask_layer = cur_step % 9 // 3 (encoder.py:46)
    if (layer < 0 || layer >= n_layers) {
        if (lane == 0) atomicOr(err, 2);
        layer = 0;
    }
    bool sp = false;
    for (int j = lane; j < P; j += 64) {
        int id = (int)orow[j];
        if (id >= 0 && id < A) sp |= ((special_bits[id >> 5] >> (id & 31)) & 1u) != 0;
        else atomicOr(err, 1);
    }
    const bool any_special = __any(sp);
    uint32_t m = 0u;                                        // lane w holds mask word w (W <= 64, checked at create)
    if (lane < W) {
        m = loc_bits[layer * W + lane];
        for (int j = 0; j < P; ++j) {
            int id = (int)orow[j];
            if (id >= 0 && id < A && (id >> 5) == lane) m &= ~(1u << (id & 31));
        }
        if (any_special) m &= ~special_bits[lane];
        bits[(size_t)n * W + lane] = m;
    }
    for (int k = lane; k < A + (64 - A % 64) % 64; k += 64) {      // whole wave takes part in the shuffle
        const uint32_t word = __shfl(m, (k >> 5) < W ? (k >> 5) : 0);
        if (k < A && !((word >> (k & 31)) & 1u)) enc[(size_t)n * A + k] = 0.f;
    }
}

__global__ void k_q_mask_bwd(float* __restrict__ d_enc, const uint32_t* __restrict__ bits, int N, int A, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * A) return;
    const int n = i / A, k = i - n * A;
    if (!((bits[(size_t)n * W + (k >> 5)] >> (k & 31)) & 1u)) d_enc[i] = 0.f;
}

__device__ __forceinline__ float q_wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// first-max argmax over a row held as score(k), k = lane, lane+64, ...
template <typename F>
__device__ __forceinline__ int q_wave_argmax(int A, int lane, F score) {
    float best = 0.f;
    int best_k = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        const float s = score(k);
        if (best_k == 0x7fffffff || s > best) { best = s; best_k = k; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float os = __shfl_xor(best, off);
        const int ok = __shfl_xor(best_k, off);
        if (ok != 0x7fffffff && (best_k == 0x7fffffff || os > best || (os == best && ok < best_k))) { best = os; best_k = ok; }
    }
    return best_k;
}

// greedy action of a row: argmax q, or the BCQ rule when imitator logits are given
__device__ __forceinline__ int q_best_action(const float* __restrict__ q, const float* __restrict__ imit, int A, int lane, float log_flex) {
    if (!imit) return q_wave_argmax(A, lane, [&](int k) { return q[k]; });
    float mx = -3.4028235e38f, mn = 3.4028235e38f;
    for (int k = lane; k < A; k += 64) {
        mx = fmaxf(mx, imit[k]);
        mn = fminf(mn, q[k]);
    }
    mx = wave_max(mx);
    mn = q_wave_min(mn);
    // log_softmax shifts a row by a constant, so log pi - max log pi == logit - max logit
    return q_wave_argmax(A, lane, [&](int k) { return (q[k] - mn) * ((imit[k] - mx) > log_flex ? 1.f : 0.f); });
}

__global__ __launch_bounds__(256) void k_q_best_action(int N, int A, const float* __restrict__ q, const float* __restrict__ imit,
                                                       float log_flex, int32_t* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int a = q_best_action(q + (size_t)n * A, imit ? imit + (size_t)n * A : nullptr, A, lane, log_flex);
    if (lane == 0) out[n] = a;
}

// imitation loss of one row per wave; rows[n] = {nll, sum logits^2}; dlogits = d loss / d logits of the batch-mean loss
__global__ __launch_bounds__(256) void k_q_imitation(int N, int A, const float* __restrict__ logits, const int32_t* __restrict__ actions,
                                                     float beta, float* __restrict__ dlogits, float2* __restrict__ rows, int* __restrict__ err) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* l = logits + (size_t)n * A;
    int a = actions[n];
    if (a < 0 || a >= A) {
        if (lane == 0) atomicOr(err, 1);
        a = 0;
    }
    float mx = -3.4028235e38f, sq = 0.f;
    for (int k = lane; k < A; k += 64) {
        mx = fmaxf(mx, l[k]);
        sq += l[k] * l[k];
    }
    mx = wave_max(mx);
    sq = wave_sum(sq);
    float se = 0.f;
    for (int k = lane; k < A; k += 64) se += expf(l[k] - mx);
    const float lse = mx + logf(wave_sum(se));
    const float inv_n = 1.0f / (float)N, pen = beta * 2.0f / ((float)N * (float)A);
    for (int k = lane; k < A; k += 64)
        dlogits[(size_t)n * A + k] = (expf(l[k] - lse) - (k == a ? 1.f : 0.f)) * inv_n + pen * l[k];
    if (lane == 0) rows[n] = make_float2(lse - l[a], sq);
}

// TD (+ conservative) loss of one row per wave; rows[n] = {huber, logsumexp(q) - q[a]}
__global__ __launch_bounds__(256) void k_q_dqn(int N, int A, const float* __restrict__ q_t, const int32_t* __restrict__ actions,
                                               const float* __restrict__ rewards, const float* __restrict__ terminals,
                                               const float* __restrict__ q_next, const float* __restrict__ q_next_targ,
                                               const float* __restrict__ imit_next, float log_flex, float gamma, float alpha,
                                               float* __restrict__ dq, float2* __restrict__ rows, int32_t* __restrict__ best_out,
                                               int* __restrict__ err) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const float* q = q_t + (size_t)n * A;
    int a = actions[n];
    if (a < 0 || a >= A) {
        if (lane == 0) atomicOr(err, 1);
        a = 0;
    }
    const int best = q_best_action(q_next + (size_t)n * A, imit_next ? imit_next + (size_t)n * A : nullptr, A, lane, log_flex);
    if (best_out && lane == 0) best_out[n] = best;
    const float y = rewards[n] + gamma * q_next_targ[(size_t)n * A + best] * (1.0f - terminals[n]);
    const float diff = y - q[a];
    const float ad = fabsf(diff);
    const float huber = ad < 1.0f ? 0.5f * diff * diff : ad - 0.5f;
    const float g_td = -(ad < 1.0f ? diff : (diff > 0.f ? 1.f : -1.f));      // d huber / d q[a]
    float mx = -3.4028235e38f;
    for (int k = lane; k < A; k += 64) mx = fmaxf(mx, q[k]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int k = lane; k < A; k += 64) se += expf(q[k] - mx);
    const float lse = mx + logf(wave_sum(se));
    const float inv_n = 1.0f / (float)N;
    for (int k = lane; k < A; k += 64) {
        float g = alpha * expf(q[k] - lse);
        if (k == a) g += g_td - alpha;
        dq[(size_t)n * A + k] = g * inv_n;
    }
    if (lane == 0) rows[n] = make_float2(huber, lse - q[a]);
}

// out[0] = mean rows.x, out[1] = mean rows.y (single block, fixed order)
__global__ __launch_bounds__(256) void k_q_mean2(const float2* __restrict__ rows, int N, float* __restrict__ out) {
    __shared__ float2 sm[256];
    float2 s = make_float2(0.f, 0.f);
    for (int n = threadIdx.x; n < N; n += 256) { s.x += rows[n].x; s.y += rows[n].y; }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sm[threadIdx.x].x += sm[threadIdx.x + o].x; sm[threadIdx.x].y += sm[threadIdx.x + o].y; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = sm[0].x / (float)N; out[1] = sm[0].y / (float)N; }
}

}  // namespace rl4rs

enum { QP_W1 = 0, QP_B1, QP_EMB, QP_W2, QP_B2, QP_HW, QP_HB, QP_COUNT };

struct rl4rs_qnet {
    rl4rs_qnet_cfg c;
    bool custom;
    int W, F2, FH;              // mask words; fc2 input width; head input width
    int64_t n_params, off[QP_COUNT], size[QP_COUNT];
    TrainCtx cx;
    float *params, *grad, *adam_m, *adam_v;
    float *cat, *enc, *d_enc, *d_cat;   // custom: cat = [h1 | tail emb], enc = masked fc2 output; plain: cat = h1, enc = h2
    uint32_t *bits, *loc_bits, *special_bits;
    int* err;
    int64_t adam_t;
    std::vector<void*> owned;
};

extern "C" {

int rl4rs_qnet_destroy(rl4rs_qnet* p) {
    if (!p) return RL4RS_OK;
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return RL4RS_OK;
}

int rl4rs_qnet_create(const rl4rs_qnet_cfg* c, const float* params_host, const uint8_t* location_mask, const uint8_t* is_special,
                      void* stream, rl4rs_qnet** out) {
    RL4RS_REQUIRE(c && params_host && out, "qnet_create: null argument");
    RL4RS_REQUIRE(c->obs_dim > 0 && c->action_size > 1 && c->hidden1 > 0 && c->max_rows > 0 && c->mask_size >= 0 &&
                  c->mask_size < c->obs_dim, "qnet_create: bad sizes");
    const bool custom = c->mask_size > 0;
    if (custom) {
        RL4RS_REQUIRE(location_mask && is_special && c->n_layers > 0 && c->emb_size > 0 && 256 % c->emb_size == 0 && c->mask_size >= 2 &&
                      c->action_size <= 2048, "qnet_create: the custom encoder needs location_mask, is_special, n_layers, an emb_size dividing 256 "
                      "and action_size <= 2048");
    } else {
        RL4RS_REQUIRE(c->hidden2 > 0, "qnet_create: the plain encoder needs hidden2");
    }
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t D = c->obs_dim, A = c->action_size, H1 = c->hidden1, M = c->mask_size, ES = c->emb_size;
    rl4rs_qnet* p = new rl4rs_qnet();
    p->c = *c;
    p->custom = custom;
    p->W = (int)((A + 31) / 32);
    p->F2 = (int)(custom ? H1 + M * ES : H1);
    p->FH = (int)(custom ? A : c->hidden2);
    p->adam_t = 0;
    const int64_t n2 = custom ? A : c->hidden2;
    const int64_t sizes[QP_COUNT] = {D * H1, H1, custom ? A * ES : 0, (int64_t)p->F2 * n2, n2, (int64_t)p->FH * A, A};
    int64_t o = 0;
    for (int i = 0; i < QP_COUNT; ++i) { p->off[i] = o; p->size[i] = sizes[i]; o += sizes[i]; }
    p->n_params = o;
    int rc;
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) p->owned.push_back(*dst);
        return r;
    };
#define QN_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_qnet_destroy(p); return rc; } } while (0)
#define QN_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
        rl4rs_qnet_destroy(p); return RL4RS_EHIP; } } while (0)
    QN_FAIL(al(&p->params, p->n_params)); QN_FAIL(al(&p->grad, p->n_params));
    QN_FAIL(al(&p->adam_m, p->n_params)); QN_FAIL(al(&p->adam_v, p->n_params));
    QN_HIP(hipMemcpyAsync(p->params, params_host, (size_t)p->n_params * 4, hipMemcpyHostToDevice, st));
    QN_HIP(hipMemsetAsync(p->grad, 0, (size_t)p->n_params * 4, st));
    QN_HIP(hipMemsetAsync(p->adam_m, 0, (size_t)p->n_params * 4, st));
    QN_HIP(hipMemsetAsync(p->adam_v, 0, (size_t)p->n_params * 4, st));
    const size_t B = c->max_rows;
    QN_FAIL(al(&p->cat, B * p->F2)); QN_FAIL(al(&p->enc, B * p->FH)); QN_FAIL(al(&p->d_enc, B * p->FH)); QN_FAIL(al(&p->d_cat, B * p->F2));
    { float* t; QN_FAIL(al(&t, 1)); p->err = reinterpret_cast<int*>(t); }
    QN_HIP(hipMemsetAsync(p->err, 0, 4, st));
    std::vector<uint32_t> lb, sb;
    if (custom) {
        float* t;
        QN_FAIL(al(&t, B * p->W)); p->bits = reinterpret_cast<uint32_t*>(t);
        QN_FAIL(al(&t, (size_t)c->n_layers * p->W)); p->loc_bits = reinterpret_cast<uint32_t*>(t);
        QN_FAIL(al(&t, p->W)); p->special_bits = reinterpret_cast<uint32_t*>(t);
        lb.assign((size_t)c->n_layers * p->W, 0u);
        sb.assign(p->W, 0u);
        for (int l = 0; l < c->n_layers; ++l)
            for (int k = 0; k < A; ++k)
                if (location_mask[(size_t)l * A + k]) lb[(size_t)l * p->W + (k >> 5)] |= 1u << (k & 31);
        for (int k = 0; k < A; ++k)
            if (is_special[k]) sb[k >> 5] |= 1u << (k & 31);
        QN_HIP(hipMemcpyAsync(p->loc_bits, lb.data(), lb.size() * 4, hipMemcpyHostToDevice, st));
        QN_HIP(hipMemcpyAsync(p->special_bits, sb.data(), sb.size() * 4, hipMemcpyHostToDevice, st));
    }
    int64_t wmax = 0;
    for (int i = 0; i < QP_COUNT; ++i) if (sizes[i] > wmax) wmax = sizes[i];
    p->cx.chunk = 512;
    QN_FAIL(al(&p->cx.wt, wmax));
    QN_FAIL(al(&p->cx.part, (size_t)((B + 511) / 512) * (wmax + std::max<int64_t>(std::max<int64_t>(A, H1), n2))));      // + the bias partials (st_tn_cs)
    QN_HIP(hipStreamSynchronize(st));
#undef QN_HIP
#undef QN_FAIL
    *out = p;
    return RL4RS_OK;
}

int rl4rs_qnet_params(rl4rs_qnet* p, float** params_dev, float** grad_dev, int64_t* count) {
    RL4RS_REQUIRE(p, "qnet_params: null handle");
    if (params_dev) *params_dev = p->params;
    if (grad_dev) *grad_dev = p->grad;
    if (count) *count = p->n_params;
    return RL4RS_OK;
}

// Adam moments (device pointers, n_params floats each) and the step count: checkpointing (offline_rl save_model / load_model)
int rl4rs_qnet_adam_state(rl4rs_qnet* p, float** m_dev, float** v_dev, int64_t* step) {
    RL4RS_REQUIRE(p, "qnet_adam_state: null handle");
    if (m_dev) *m_dev = p->adam_m;
    if (v_dev) *v_dev = p->adam_v;
    if (step) *step = p->adam_t;
    return RL4RS_OK;
}
int rl4rs_qnet_set_adam_step(rl4rs_qnet* p, int64_t step) {
    RL4RS_REQUIRE(p && step >= 0, "qnet_set_adam_step: bad argument");
    p->adam_t = step;
    return RL4RS_OK;
}

int rl4rs_qnet_copy_params(rl4rs_qnet* dst, const rl4rs_qnet* src, void* stream) {
    RL4RS_REQUIRE(dst && src && dst->n_params == src->n_params, "qnet_copy_params: handles differ");
    RL4RS_HIP_TRY(hipMemcpyAsync(dst->params, src->params, (size_t)src->n_params * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RL4RS_OK;
}

int rl4rs_qnet_status(rl4rs_qnet* p, int32_t* flags, void* stream) {
    RL4RS_REQUIRE(p && flags, "qnet_status: null argument");
    int v = 0;
    RL4RS_HIP_TRY(hipMemcpyAsync(&v, p->err, 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    RL4RS_HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    *flags = v;
    return RL4RS_OK;
}

int rl4rs_qnet_forward(rl4rs_qnet* p, int32_t N, const float* obs, float* out, void* stream) {
    RL4RS_REQUIRE(p && obs && out && N > 0 && N <= p->c.max_rows, "qnet_forward: bad argument (N=%d, max_rows=%d)", N, p ? p->c.max_rows : -1);
    hipStream_t st = (hipStream_t)stream;
    const int D = p->c.obs_dim, A = p->c.action_size, H1 = p->c.hidden1, M = p->c.mask_size, ES = p->c.emb_size;
    const float* P = p->params;
    const int64_t* o = p->off;
    int rc;
    if ((rc = launch_gemm_f32(obs, D, P + o[QP_W1], H1, P + o[QP_B1], p->cat, p->F2, N, H1, D, 4, st))) return rc;     // relu
    if (p->custom) {
        hipLaunchKernelGGL(k_q_tail_emb, dim3((N * M * ES + 255) / 256), dim3(256), 0, st, obs, N, D, M, A, ES, P + o[QP_EMB], p->cat,
                           (int64_t)p->F2, H1, p->err);
        if ((rc = launch_gemm_f32(p->cat, p->F2, P + o[QP_W2], A, P + o[QP_B2], p->enc, A, N, A, p->F2, 0, st))) return rc;
        hipLaunchKernelGGL(k_q_mask, dim3((N + 3) / 4), dim3(256), 0, st, obs, N, D, M, A, p->W, p->loc_bits, p->c.n_layers, p->special_bits,
                           p->enc, p->bits, p->err);
    } else {
        if ((rc = launch_gemm_f32(p->cat, H1, P + o[QP_W2], p->FH, P + o[QP_B2], p->enc, p->FH, N, p->FH, H1, 4, st))) return rc;
    }
    RL4RS_LAUNCH_CHECK();
    return launch_gemm_f32(p->enc, p->FH, P + o[QP_HW], A, P + o[QP_HB], out, A, N, A, p->FH, 0, st);
}

// gradient of sum(out * dout) wrt every parameter into the handle's flat gradient buffer; must follow rl4rs_qnet_forward of the
// SAME rows (the activations live in the handle)
int rl4rs_qnet_backward(rl4rs_qnet* p, int32_t N, const float* obs, const float* dout, void* stream) {
    RL4RS_REQUIRE(p && obs && dout && N > 0 && N <= p->c.max_rows, "qnet_backward: bad argument (N=%d, max_rows=%d)", N, p ? p->c.max_rows : -1);
    hipStream_t st = (hipStream_t)stream;
    const int D = p->c.obs_dim, A = p->c.action_size, H1 = p->c.hidden1, M = p->c.mask_size, ES = p->c.emb_size, FH = p->FH, F2 = p->F2;
    const float* P = p->params;
    float* G = p->grad;
    const int64_t* o = p->off;
    auto ew = [](int n) { return dim3((n + 255) / 256); };
    const dim3 b256(256);
    int rc;
    st_tn_cs(p->cx, st, p->enc, FH, FH, dout, A, A, N, G + o[QP_HW], G + o[QP_HB]);      // weight + bias gradient: one launch
    if (p->custom) {
        if ((rc = st_back(p->cx, st, dout, A, A, P + o[QP_HW], A, FH, p->d_enc, FH, N))) return rc;
        hipLaunchKernelGGL(k_q_mask_bwd, ew(N * A), b256, 0, st, p->d_enc, p->bits, N, A, p->W);
    } else if ((rc = launch_gemm_nt(dout, A, P + o[QP_HW], A, p->d_enc, FH, N, FH, A, st, p->enc, FH))) {       // dY W^T with relu'(enc) folded in
        return rc;
    }
    st_tn_cs(p->cx, st, p->cat, F2, F2, p->d_enc, FH, FH, N, G + o[QP_W2], G + o[QP_B2]);
    if (p->custom) {
        if ((rc = st_back(p->cx, st, p->d_enc, FH, FH, P + o[QP_W2], FH, F2, p->d_cat, F2, N))) return rc;
        hipLaunchKernelGGL(k_q_tail_emb_bwd, dim3(A), b256, 0, st, obs, N, D, M, A, ES, p->d_cat, (int64_t)F2, H1, G + o[QP_EMB]);
        hipLaunchKernelGGL(k_relu_bwd, ew(N * H1), b256, 0, st, p->d_cat, (int64_t)F2, p->cat, (int64_t)F2, N * H1, H1);      // the hidden part only
    } else if ((rc = launch_gemm_nt(p->d_enc, FH, P + o[QP_W2], FH, p->d_cat, F2, N, F2, FH, st, p->cat, F2))) {    // F2 == H1: all of cat is relu(fc1)
        return rc;
    }
    st_tn_cs(p->cx, st, obs, D, D, p->d_cat, F2, H1, N, G + o[QP_W1], G + o[QP_B1]);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// torch.optim.Adam: p -= lr / (1 - b1^t) * m / (sqrt(v / (1 - b2^t)) + eps)  ==  the keras form of k_adam with eps * sqrt(1 - b2^t)
int rl4rs_qnet_adam_step(rl4rs_qnet* p, float lr, float beta1, float beta2, float eps, void* stream) {
    RL4RS_REQUIRE(p, "qnet_adam_step: null handle");
    hipStream_t st = (hipStream_t)stream;
    p->adam_t += 1;
    const double t = (double)p->adam_t;
    const double c2 = sqrt(1.0 - pow((double)beta2, t));
    const float lr_t = (float)(lr * c2 / (1.0 - pow((double)beta1, t)));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((p->n_params + 255) / 256)), dim3(256), 0, st, p->params, p->grad, p->adam_m, p->adam_v,
                       (int)p->n_params, lr_t, beta1, beta2, (float)(eps * c2), (const float*)nullptr, 0.f);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_q_best_action(int32_t N, int32_t A, const float* q, const float* imitator_logits, float action_flexibility, int32_t* actions,
                        void* stream) {
    RL4RS_REQUIRE(q && actions && N > 0 && A > 1, "q_best_action: bad argument");
    hipLaunchKernelGGL(k_q_best_action, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, A, q, imitator_logits,
                       logf(action_flexibility), actions);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_qloss_imitation(rl4rs_qnet* p, int32_t N, const float* logits, const int32_t* actions, float beta, float* dlogits,
                          float* rows_scratch, float* loss2, void* stream) {
    RL4RS_REQUIRE(p && logits && actions && dlogits && rows_scratch && loss2 && N > 0, "qloss_imitation: bad argument");
    hipStream_t st = (hipStream_t)stream;
    float2* rows = reinterpret_cast<float2*>(rows_scratch);
    hipLaunchKernelGGL(k_q_imitation, dim3((N + 3) / 4), dim3(256), 0, st, N, p->c.action_size, logits, actions, beta, dlogits, rows, p->err);
    hipLaunchKernelGGL(k_q_mean2, dim3(1), dim3(256), 0, st, rows, N, loss2);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_qloss_dqn(rl4rs_qnet* p, int32_t N, const float* q_t, const int32_t* actions, const float* rewards, const float* terminals,
                    const float* q_next, const float* q_next_target, const float* imitator_next_logits, float action_flexibility,
                    float gamma, float cql_alpha, float* dq, float* rows_scratch, float* loss2, int32_t* best_next_action, void* stream) {
    RL4RS_REQUIRE(p && q_t && actions && rewards && terminals && q_next && q_next_target && dq && rows_scratch && loss2 && N > 0,
                  "qloss_dqn: bad argument");
    hipStream_t st = (hipStream_t)stream;
    float2* rows = reinterpret_cast<float2*>(rows_scratch);
    hipLaunchKernelGGL(k_q_dqn, dim3((N + 3) / 4), dim3(256), 0, st, N, p->c.action_size, q_t, actions, rewards, terminals, q_next,
                       q_next_target, imitator_next_logits, imitator_next_logits ? logf(action_flexibility) : 0.f, gamma, cql_alpha, dq, rows,
                       best_next_action, p->err);
    hipLaunchKernelGGL(k_q_mean2, dim3(1), dim3(256), 0, st, rows, N, loss2);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // extern "C"
