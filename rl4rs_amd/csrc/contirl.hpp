// Continuous-action offline-RL learners (BASELINE configs[4]: "Continuous-action SlateRecEnv + K-NN item search, BCQ offline
// RL"): what the reference trains with d3rlpy.algos.BCQ ('BCQ-conti', script/batchrl_trainer.py:61-73) and d3rlpy.algos.CQL
// ('CQL-conti', :91-107) on the dataset whose actions are the 32-d item embeddings (data_generate_rl4rs_a_conti, :220-270); the
// learned policy's embedding is what the env's K-NN resolves (rl4rs/env/slate.py:180-198).  Both scripts leave the custom encoder
// factory commented out, so every network is d3rlpy's default:
//
//   amlp ("action MLP")   VectorEncoderWithAction(hidden_units=[256, 256], relu) on cat([x, action]) + one nn.Linear head
//       h1 = relu([x | a] W1 + b1);  h2 = relu(h1 W2 + b2);  out = act(h2 W3 + b3)         (act_dim = 0: plain VectorEncoder)
//   BCQ    ConditionalVAE   encoder amlp(x, action) -> [mu | logstd] (the two Linear heads side by side), decoder
//                           amlp(x, latent) -> tanh -> action;  loss = mse(decode(x, rsample), a) + beta * KL(N(mu, sigma) || N(0,1))
//          DeterministicResidualPolicy   a' = clamp(a + scale * tanh(amlp(x, a)), -1, 1),  scale = action_flexibility (0.05)
//          twin ContinuousMeanQFunction  amlp(x, a) -> 1;  loss = sum_c mean (Q_c(s, a) - y)^2
//          target  y = r + gamma * (1 - terminal) * max_n [(1 - lam) max_c + lam min_c] Q_targ_c(s', pi_targ(s', decode(s', z_n))),
//                  z_n = clamp(randn, -0.5, 0.5), n = 100 sampled actions per next observation, lam = 0.75
//          actor   -mean Q_1(s, pi(s, decode(s, z)));  soft target updates tau = 0.005
//   CQL    SquashedNormalPolicy amlp(x) -> [mu | logstd];  SAC actor / temperature losses;  twin critics with the conservative
//          term alpha * (w * (logsumexp_{3n} [Q(s, a_j) - log p_j] - Q(s, a)) - threshold), learned log alpha
// d3rlpy 0.91 is absent from this image (environment.yml:146): the algorithms are restated as published, PARITY UNPINNED, and
// checked against torch float64 autograd of the same restatement in tests/test_gpu_offline_conti.py.
//
// MI355X notes.  The heavy part of a BCQ update is the target: 4 forward networks over batch * 100 = 25 600 rows.  The observation
// (266 of the 298 first-layer inputs) is the SAME for the 100 sampled actions of a row, so the first layer is split: the
// observation side x W1[:D] + b1 is one GEMM over the 256 distinct rows, and the per-sample GEMM has K = act_dim only with that
// projection as a row-shared addend (k_gemm_f32's add_div) - 89 % of the first layer's FLOPs and all of the repeated-observation
// traffic (d3rlpy materialises the [25600, 266] expand) never happen.  Everything that is differentiated is exact fp32 on
// v_mfma_f32_32x32x2_f32; sample-axis reductions use the fixed-order helpers of simtrain.hpp, so an update is bit-reproducible
// given its noise.  The forwards that are NEVER differentiated - the batch x 100 target rows, the alpha step's rows, greedy
// evaluation - may run in the scorer's fp16x2 arithmetic as one fused launch per network (rl4rs_amlp_forward_h16 ->
// k_amlp_fwd_h16 in gemm.hip; the learners' `nograd_precision`, default 'fp16x2'): 3x on the evaluation rollout's predict.
#pragma once

namespace rl4rs {

// out[n] = dot(h[n, :H], w) + b : the one-output head of a critic (one wave per row)
__global__ __launch_bounds__(256) void k_amlp_head1(const float* __restrict__ h, int N, int H, const float* __restrict__ w,
                                                    const float* __restrict__ b, float* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float s = 0.f;
    for (int k = lane; k < H; k += 64) s += h[(size_t)n * H + k] * w[k];
    s = wave_sum(s);
    if (lane == 0) out[n] = s + b[0];
}

// dproj[r, c] = sum_{i < rep} d[(r * rep + i), c]   (fixed order)
__global__ void k_group_sum(const float* __restrict__ d, int R, int rep, int cols, float* __restrict__ dproj) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * cols) return;
    const int r = i / cols, c = i - r * cols;
    float s = 0.f;
    for (int k = 0; k < rep; ++k) s += d[((size_t)r * rep + k) * cols + c];
    dproj[i] = s;
}

// targ = (1 - tau) * targ + tau * src        (d3rlpy soft_sync)
__global__ void k_soft_update(float* __restrict__ targ, const float* __restrict__ src, int64_t n, float tau) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) targ[i] = targ[i] * (1.0f - tau) + tau * src[i];
}

// ConditionalVAE.encode(...).rsample():  z = mu + exp(clamp(logstd, lo, hi)) * eps,   enc = [mu | logstd] per row
__global__ void k_cvae_sample(int N, int L, const float* __restrict__ enc, const float* __restrict__ eps, float lo, float hi,
                              float* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int n = i / L, l = i - n * L;
    const float mu = enc[(size_t)n * 2 * L + l], ls = fminf(fmaxf(enc[(size_t)n * 2 * L + L + l], lo), hi);
    z[i] = mu + expf(ls) * eps[i];
}

// ConditionalVAE.compute_error: one wave per row.  y = decoder output (tanh already applied by the head);
// rows[n] = {sum_e (y - a)^2, sum_l KL(N(mu, sigma) || N(0, 1))};  d_dec = d mse / d (pre-tanh decoder output)
__global__ __launch_bounds__(256) void k_cvae_loss(int N, int E, int L, const float* __restrict__ y, const float* __restrict__ a,
                                                   const float* __restrict__ enc, float lo, float hi, float* __restrict__ d_dec,
                                                   float2* __restrict__ rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    float se = 0.f, kl = 0.f;
    const float sc = 2.0f / ((float)N * (float)E);
    for (int e = lane; e < E; e += 64) {
        const float yy = y[(size_t)n * E + e], d = yy - a[(size_t)n * E + e];
        se += d * d;
        d_dec[(size_t)n * E + e] = sc * d * (1.0f - yy * yy);
    }
    for (int l = lane; l < L; l += 64) {
        const float mu = enc[(size_t)n * 2 * L + l], ls = fminf(fmaxf(enc[(size_t)n * 2 * L + L + l], lo), hi);
        const float var = expf(2.0f * ls);
        kl += 0.5f * (var + mu * mu - 1.0f) - ls;        // torch kl_divergence(Normal, Normal) with q = N(0, 1)
    }
    se = wave_sum(se);
    kl = wave_sum(kl);
    if (lane == 0) rows[n] = make_float2(se, kl);
}

// gradient wrt [mu | logstd] given dz (the decoder's input gradient):  mse path through z = mu + sigma eps, plus beta * mean KL
__global__ void k_cvae_enc_grad(int N, int L, const float* __restrict__ enc, const float* __restrict__ eps,
                                const float* __restrict__ dz, float beta, float lo, float hi, float* __restrict__ d_enc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * L) return;
    const int n = i / L, l = i - n * L;
    const float mu = enc[(size_t)n * 2 * L + l], raw = enc[(size_t)n * 2 * L + L + l];
    const float ls = fminf(fmaxf(raw, lo), hi);
    const float sig = expf(ls), k = beta / ((float)N * (float)L);
    d_enc[(size_t)n * 2 * L + l] = dz[i] + k * mu;
    // torch.clamp passes the gradient where lo <= x <= hi
    d_enc[(size_t)n * 2 * L + L + l] = (raw >= lo && raw <= hi) ? dz[i] * eps[i] * sig + k * (sig * sig - 1.0f) : 0.f;
}

// DeterministicResidualPolicy.forward: out = clamp(a + scale * t, -1, 1),  t = tanh(fc(h)) (applied by the head)
__global__ void k_residual_action(int n, const float* __restrict__ a, const float* __restrict__ t, float scale, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fminf(fmaxf(a[i] + scale * t[i], -1.0f), 1.0f);
}

// d (pre-tanh head output) = d_out * [|a + scale t| inside the clamp] * scale * (1 - t^2)
__global__ void k_residual_grad(int n, const float* __restrict__ a, const float* __restrict__ t, float scale,
                                const float* __restrict__ d_out, float* __restrict__ d_pre) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = a[i] + scale * t[i];
    d_pre[i] = (v >= -1.0f && v <= 1.0f) ? d_out[i] * scale * (1.0f - t[i] * t[i]) : 0.f;
}

// compute_max_with_n_actions: per row b, v_j = (1 - lam) max(q1, q2) + lam min(q1, q2) over its n sampled actions; value = max_j
// (first maximum), y = r + gamma * value * (1 - terminal) when rewards are given.  q2 NULL: v_j = q1 (the greedy pick of predict).
__global__ __launch_bounds__(256) void k_bcq_target(int B, int n, const float* __restrict__ q1, const float* __restrict__ q2, float lam,
                                                    const float* __restrict__ rewards, const float* __restrict__ terminals, float gamma,
                                                    float* __restrict__ y, int32_t* __restrict__ best) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    float bv = 0.f;
    int bj = 0x7fffffff;
    for (int j = lane; j < n; j += 64) {
        const float a = q1[(size_t)b * n + j];
        float v = a;
        if (q2) {
            const float c = q2[(size_t)b * n + j];
            v = (1.0f - lam) * fmaxf(a, c) + lam * fminf(a, c);
        }
        if (bj == 0x7fffffff || v > bv) { bv = v; bj = j; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oj = __shfl_xor(bj, off);
        if (oj != 0x7fffffff && (bj == 0x7fffffff || ov > bv || (ov == bv && oj < bj))) { bv = ov; bj = oj; }
    }
    if (lane == 0) {
        if (y) y[b] = rewards ? rewards[b] + gamma * bv * (1.0f - terminals[b]) : bv;
        if (best) best[b] = bj;
    }
}

// out[b, :] = actions[b * n + best[b], :]
__global__ void k_pick_rows(int B, int n, int E, const float* __restrict__ actions, const int32_t* __restrict__ best, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * E) return;
    const int b = i / E, e = i - b * E;
    out[i] = actions[((size_t)b * n + best[b]) * E + e];
}

// EnsembleContinuousQFunction.compute_error (mean Q functions): loss = sum_c mean_n (q_c - y)^2;  dq_c = 2 (q_c - y) / N
// single block, fixed order; loss2 = {mean (q1 - y)^2, mean (q2 - y)^2}
__global__ __launch_bounds__(256) void k_critic_mse(int N, const float* __restrict__ q1, const float* __restrict__ q2, const float* __restrict__ y,
                                                    float* __restrict__ dq1, float* __restrict__ dq2, float* __restrict__ loss2) {
    __shared__ float2 sm[256];
    float2 s = make_float2(0.f, 0.f);
    const float k = 2.0f / (float)N;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float d1 = q1[n] - y[n], d2 = q2[n] - y[n];
        s.x += d1 * d1;
        s.y += d2 * d2;
        dq1[n] = k * d1;
        dq2[n] = k * d2;
    }
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { sm[threadIdx.x].x += sm[threadIdx.x + o].x; sm[threadIdx.x].y += sm[threadIdx.x + o].y; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { loss2[0] = sm[0].x / (float)N; loss2[1] = sm[0].y / (float)N; }
}


// ---- continuous CQL (d3rlpy.algos.CQL = SAC + conservative critic loss) -------------------------------------------------------
// SquashedNormalPolicy.sample(_n)_with_log_prob: head [R, 2A] = [mu | logstd] of R = N / rep observations, eps [N, A]:
//   u = mu + exp(clamp(logstd, lo, hi)) * eps,  a = tanh(u),
//   logp = sum_e [ Normal(mu, sigma).log_prob(u) - 2 (log 2 - u - softplus(-2 u)) ]            (d3rlpy _squash_action)
// eps NULL: the deterministic best_action a = tanh(mu) (no log-prob).  Sample i of observation r lands in destination row
// r * out_rep + out_off + i % rep of act_out [.., A] / logp_out: the caller lays several sample groups of one observation side
// by side (the [data | pi(s) | pi(s') | uniform] rows of the conservative loss).  One wave per sample row.
__global__ __launch_bounds__(256) void k_squashed_sample(int N, int rep, int A, const float* __restrict__ head, const float* __restrict__ eps,
                                                         float lo, float hi, int out_rep, int out_off, float* __restrict__ act_out,
                                                         float* __restrict__ logp_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + wave;
    if (i >= N) return;
    const int r = i / rep;
    const size_t dst = (size_t)r * out_rep + out_off + (i - r * rep);
    float lp = 0.f;
    for (int e = lane; e < A; e += 64) {
        const float mu = head[(size_t)r * 2 * A + e];
        float u = mu;
        if (eps) {
            const float ls = fminf(fmaxf(head[(size_t)r * 2 * A + A + e], lo), hi);
            const float ep = eps[(size_t)i * A + e];
            u = mu + expf(ls) * ep;
            const float m2u = -2.0f * u;
            const float softplus = fmaxf(m2u, 0.f) + log1pf(expf(-fabsf(m2u)));
            lp += -0.5f * ep * ep - ls - 0.9189385332046727f - 2.0f * (0.6931471805599453f - u - softplus);
        }
        act_out[dst * A + e] = tanhf(u);
    }
    if (eps && logp_out) {
        lp = wave_sum(lp);
        if (lane == 0) logp_out[dst] = lp;
    }
}

// SAC actor loss  mean_b [ T * logp_b - Qmin(s_b, a_b) ]  back to the policy head [mu | logstd]:  g_a [B, A] = gradient of
// -mean Qmin wrt the action (from the critics' backward, 1 / B included), T = exp(log_temp).
//   dL/du = T * 2 a / B + g_a (1 - a^2)     (d logp / du = 2 tanh(u): the Normal log-prob of an rsample does not move with mu)
//   dL/dmu = dL/du;   dL/dlogstd = [lo <= raw <= hi] (dL/du * sigma * eps - T / B)
__global__ void k_sac_actor_grad(int B, int A, const float* __restrict__ head, const float* __restrict__ eps, const float* __restrict__ act,
                                 const float* __restrict__ g_a, const float* __restrict__ log_temp, float lo, float hi,
                                 float* __restrict__ d_head) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * A) return;
    const int b = i / A, e = i - b * A;
    const float T = expf(log_temp[0]), inv_b = 1.0f / (float)B;
    const float a = act[i], raw = head[(size_t)b * 2 * A + A + e];
    const float ls = fminf(fmaxf(raw, lo), hi);
    const float du = T * 2.0f * a * inv_b + g_a[i] * (1.0f - a * a);
    d_head[(size_t)b * 2 * A + e] = du;
    d_head[(size_t)b * 2 * A + A + e] = (raw >= lo && raw <= hi) ? du * expf(ls) * eps[i] - T * inv_b : 0.f;
}

// twin-critic minimum and the gradient selector of -mean_b min(q1, q2): dq_c = -1/B on the smaller one (q1 on ties)
__global__ void k_twin_min(int B, const float* __restrict__ q1, const float* __restrict__ q2, float* __restrict__ qmin,
                           float* __restrict__ dq1, float* __restrict__ dq2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const bool first = q1[i] <= q2[i];
    qmin[i] = first ? q1[i] : q2[i];
    if (dq1) {
        dq1[i] = first ? -1.0f / (float)B : 0.f;
        dq2[i] = first ? 0.f : -1.0f / (float)B;
    }
}

// CQL critic loss over rows laid out [B][m]: column 0 = the dataset action (TD + the -Q(s, a) term), columns 1 .. m-1 = the
// sampled actions of the conservative term with their importance offsets offs (log-prob of a policy sample, log 0.5^A of a
// uniform one).  One wave per observation.  y NULL: values only (the alpha update).
//   sums[0..1] = sum_b (q_c[b,0] - y_b)^2,  sums[2..3] = sum_b logsumexp_j (q_c[b,j] - offs[b,j]),  sums[4..5] = sum_b q_c[b,0]
//   dq_c[b,0] = 2 (q_c[b,0] - y_b) / B - aw / (2 B),   dq_c[b,j] = aw / (2 B) * softmax_j(...),   aw = clipped alpha * weight
__global__ __launch_bounds__(256) void k_cql_rows(int B, int m, const float* __restrict__ q1, const float* __restrict__ q2,
                                                  const float* __restrict__ offs, const float* __restrict__ y, const float* __restrict__ aw_dev,
                                                  float* __restrict__ dq1, float* __restrict__ dq2, float* __restrict__ rows) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + wave;
    if (b >= B) return;
    const float aw = aw_dev ? aw_dev[0] : 0.f;
    const float k = aw / (2.0f * (float)B);
    float out[6];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float* q = (c == 0 ? q1 : q2) + (size_t)b * m;
        float* dq = (c == 0 ? dq1 : dq2);
        float mx = -3.4028235e38f;
        for (int j = 1 + lane; j < m; j += 64) mx = fmaxf(mx, q[j] - offs[(size_t)b * m + j]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int j = 1 + lane; j < m; j += 64) se += expf(q[j] - offs[(size_t)b * m + j] - mx);
        se = wave_sum(se);
        const float lse = mx + logf(se);
        const float d0 = q[0] - (y ? y[b] : 0.f);
        out[c] = d0 * d0;
        out[2 + c] = lse;
        out[4 + c] = q[0];
        if (y && dq) {
            for (int j = 1 + lane; j < m; j += 64) dq[(size_t)b * m + j] = k * expf(q[j] - offs[(size_t)b * m + j] - lse);
            if (lane == 0) dq[(size_t)b * m] = 2.0f * d0 / (float)B - k;
        }
    }
    if (lane == 0)
        for (int c = 0; c < 6; ++c) rows[(size_t)b * 6 + c] = out[c];
}

// sums[c] = sum_b rows[b][c], c < 6 (single block, fixed order)
__global__ __launch_bounds__(256) void k_sum6(const float* __restrict__ rows, int B, float* __restrict__ sums) {
    __shared__ float sm[6][256];
    float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = threadIdx.x; b < B; b += 256)
        for (int c = 0; c < 6; ++c) s[c] += rows[(size_t)b * 6 + c];
    for (int c = 0; c < 6; ++c) sm[c][threadIdx.x] = s[c];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o)
            for (int c = 0; c < 6; ++c) sm[c][threadIdx.x] += sm[c][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 6) sums[threadIdx.x] = sm[threadIdx.x][0];
}

}  // namespace rl4rs
#include "amlp_fused.hpp"

enum { AP_W1 = 0, AP_B1, AP_W2, AP_B2, AP_W3, AP_B3, AP_COUNT };

struct rl4rs_amlp {
    rl4rs_amlp_cfg c;
    int64_t n_params, off[AP_COUNT], size[AP_COUNT];
    TrainCtx cx;
    float *params, *grad, *adam_m, *adam_v;
    float *proj, *h1, *h2;              // forward: [max_rows, hidden1] observation-side projection (one row per DISTINCT observation), activations
    float *d_h1, *d_h2, *d_proj;        // backward scratch, [max_grad_rows, ...]
    int last_n, last_rep;               // rows of the last forward (the backward must match)
    float *w2t, *w3t, *w1at;            // transposed weights of the fused minibatch backward (amlp_fused.hpp; NULL: shape not eligible)
    bool t_valid;                       // the transposes equal the current parameters: set by the fused forward (which rebuilds them),
                                        // cleared by everything that may write the parameters (Adam, copies, soft updates, handing out the
                                        // raw pointer) and by a non-fused forward; the fused backward rebuilds them when it is unset
    float *w1p, *w2p, *w3p, *w1xp;      // fp16 hi / lo fragment planes of W1's action rows, W2, W3 (and W1's observation rows) for rl4rs_amlp_forward_h16 (NULL: shape not eligible)
    int64_t adam_t;
    std::vector<void*> owned;
};

// the shapes k_amlp_fwd_h16 (gemm.hip) takes: d3rlpy's default 256 x 256 encoder with an action input of 8 .. 64 values in
// multiples of 8 and at most 64 outputs - every network of BCQ and CQL except the plain-encoder policy of CQL
static bool amlp_h16_shape_ok(const rl4rs_amlp_cfg& c) {
    return c.act_dim >= 8 && c.act_dim <= 64 && (c.act_dim & 7) == 0 && c.hidden1 == 256 && c.hidden2 == 256 && c.out_dim <= 64 &&
           (c.head_act == ACT_NONE || c.head_act == ACT_TANH || c.head_act == ACT_RELU || c.head_act == ACT_SIGMOID);
}

// the shapes the fused minibatch kernels take (amlp_fused.hpp): d3rlpy's default 256 x 256 encoder, at most 64 outputs / action inputs
static bool amlp_fused_shape_ok(const rl4rs_amlp_cfg& c) {
    return c.hidden1 == 256 && c.hidden2 == 256 && c.out_dim <= 64 && c.act_dim <= 64 && c.obs_dim + c.act_dim <= 4096;
}
static int g_amlp_fused = 1;            // rl4rs_amlp_set_fused: 0 = the per-layer launches for every call, 2 = the 8-row fused form (tests, A/B runs)
static bool amlp_fused_call(const rl4rs_amlp* p, int N, int rep) { return g_amlp_fused && amlp_fused_shape_ok(p->c) && rep == 1 && N <= 2048; }

// n <= 4 networks with the same input widths over the SAME rows as one launch each way (amlp_fused.hpp)
static int amlp_forward_fused(int n, rl4rs_amlp* const* nets, int N, const float* obs, const float* act, float* const* outs, hipStream_t st) {
    int rc;
    // (every call: a cached lookup keyed by device and function - a process-wide flag skipped the second GPU of a process)
    if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_amlp_fwd4<1>), amlp_fwd4_smem(4096, 1)))) return rc;
    if ((rc = raise_dyn_smem(reinterpret_cast<const void*>(&k_amlp_fwd4<2>), amlp_fwd4_smem(4096, 2)))) return rc;
    // 4 rows per workgroup.  The 8-row form (MTW = 2: every weight value feeds two row tiles) was measured SLOWER at the learners'
    // 256-row minibatches - 32 workgroups instead of 64, BCQ update 0.371 -> 0.406 ms - and is kept for A/B only (rl4rs_amlp_set_fused(2))
    const int mtw = g_amlp_fused == 2 ? 2 : 1;
    AmlpFwd4x x;
    memset(&x, 0, sizeof(x));
    int t_elems = 0;
    for (int i = 0; i < n; ++i) {
        rl4rs_amlp* p = nets[i];
        const float* P = p->params;
        const int64_t* o = p->off;
        const int D = p->c.obs_dim, E = p->c.act_dim, K = p->c.out_dim;
        x.n[i] = AmlpFwd4{obs, act, P + o[AP_W1], P + o[AP_B1], P + o[AP_W2], P + o[AP_B2], P + o[AP_W3], P + o[AP_B3], p->h1, p->h2, outs[i],
                          p->w3t, p->w2t, p->w1at, N, D, E, K, p->c.head_act};
        if (p->w2t) { t_elems = std::max(t_elems, 256 * 256 + 256 * K + 256 * E); p->t_valid = true; }
        p->last_n = N;
        p->last_rep = 1;
    }
    const int row_wgs = (N + 4 * mtw - 1) / (4 * mtw), t_wgs = (t_elems + AMLP_T_PER_WG - 1) / AMLP_T_PER_WG;
    const size_t smem = amlp_fwd4_smem(nets[0]->c.obs_dim + nets[0]->c.act_dim, mtw);
    if (mtw == 2) hipLaunchKernelGGL(k_amlp_fwd4<2>, dim3(row_wgs + t_wgs, n), dim3(256), smem, st, x);
    else hipLaunchKernelGGL(k_amlp_fwd4<1>, dim3(row_wgs + t_wgs, n), dim3(256), smem, st, x);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}
static int amlp_backward_fused(int n, rl4rs_amlp* const* nets, int N, const float* obs, const float* act, const float* const* douts,
                               float* const* dacts, int want_param_grad, hipStream_t st) {
    AmlpBwd4x x;
    memset(&x, 0, sizeof(x));
    TnGroup g;
    memset(&g, 0, sizeof(g));
    g.Ns = N;
    auto add = [&](const float* A, int lda, int M, const float* B, int ldb, int Nc, float* dst, float* bias) {
        const int i = g.n++;
        g.A[i] = A; g.lda[i] = lda; g.M[i] = M; g.B[i] = B; g.ldb[i] = ldb; g.Nc[i] = Nc; g.out[i] = dst; g.bias[i] = bias;
        g.tile0[i + 1] = g.tile0[i] + ((M + 31) / 32) * ((Nc + 31) / 32);
    };
    for (int i = 0; i < n; ++i) {
        rl4rs_amlp* p = nets[i];
        const int64_t* o = p->off;
        float* G = p->grad;
        const int D = p->c.obs_dim, E = p->c.act_dim, K = p->c.out_dim, H1 = 256, H2 = 256;
        if (!p->t_valid) {
            // the forward of these rows did not take the fused path (rl4rs_amlp_set_fused toggled in between, a rep > 1 forward) or the
            // parameters may have been written since: nothing tracked that before ADVICE r5 and the chain read stale transposes
            const float* P = p->params;
            const int t_elems = 256 * 256 + 256 * K + 256 * E;
            hipLaunchKernelGGL(k_amlp_transposes, dim3((t_elems + 255) / 256), dim3(256), 0, st, P + o[AP_W3], K, p->w3t, P + o[AP_W2], p->w2t,
                               P + o[AP_W1] + (size_t)D * 256, E, p->w1at);
            p->t_valid = true;
        }
        x.n[i] = AmlpBwd4{douts[i], p->h1, p->h2, p->w3t, p->w2t, p->w1at, p->d_h2, p->d_h1, dacts ? dacts[i] : nullptr, N, K, E};
        if (want_param_grad) {
            add(p->h2, H2, H2, douts[i], K, K, G + o[AP_W3], G + o[AP_B3]);
            add(p->h1, H1, H1, p->d_h2, H2, H2, G + o[AP_W2], G + o[AP_B2]);
            add(obs, D, D, p->d_h1, H1, H1, G + o[AP_W1], G + o[AP_B1]);
            if (E > 0) add(act, E, E, p->d_h1, H1, H1, G + o[AP_W1] + (size_t)D * H1, nullptr);
        }
    }
    const int mtw = g_amlp_fused == 2 ? 2 : 1;          // (see amlp_forward_fused)
    if (mtw == 2) hipLaunchKernelGGL(k_amlp_bwd4<2>, dim3((N + 7) / 8, n), dim3(256), 0, st, x);
    else hipLaunchKernelGGL(k_amlp_bwd4<1>, dim3((N + 3) / 4, n), dim3(256), 0, st, x);
    if (want_param_grad) hipLaunchKernelGGL(k_gemm_tn4_group, dim3(g.tile0[g.n]), dim3(256), 0, st, g);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

extern "C" {

int rl4rs_amlp_set_fused(int32_t on) {
    g_amlp_fused = on == 2 ? 2 : (on ? 1 : 0);
    return RL4RS_OK;
}

int rl4rs_amlp_destroy(rl4rs_amlp* p) {
    if (!p) return RL4RS_OK;
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return RL4RS_OK;
}

int rl4rs_amlp_create(const rl4rs_amlp_cfg* c, const float* params_host, void* stream, rl4rs_amlp** out) {
    RL4RS_REQUIRE(c && params_host && out, "amlp_create: null argument");
    RL4RS_REQUIRE(c->obs_dim > 0 && c->act_dim >= 0 && c->hidden1 > 0 && c->hidden2 > 0 && c->out_dim > 0 && c->max_rows > 0 &&
                  c->max_grad_rows >= 0 && c->max_grad_rows <= c->max_rows && c->head_act >= 0 && c->head_act <= 4, "amlp_create: bad sizes");
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int64_t D = c->obs_dim, E = c->act_dim, H1 = c->hidden1, H2 = c->hidden2, K = c->out_dim;
    rl4rs_amlp* p = new rl4rs_amlp();
    p->c = *c;
    p->adam_t = 0;
    p->last_n = p->last_rep = 0;
    p->w1p = p->w2p = p->w3p = p->w1xp = nullptr;
    p->w2t = p->w3t = p->w1at = nullptr;
    p->t_valid = false;
    const int64_t sizes[AP_COUNT] = {(D + E) * H1, H1, H1 * H2, H2, H2 * K, K};
    int64_t o = 0;
    for (int i = 0; i < AP_COUNT; ++i) { p->off[i] = o; p->size[i] = sizes[i]; o += sizes[i]; }
    p->n_params = o;
    int rc;
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) p->owned.push_back(*dst);
        return r;
    };
#define AM_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_amlp_destroy(p); return rc; } } while (0)
#define AM_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error("%s failed: %s", #expr, hipGetErrorString(e_)); \
        rl4rs_amlp_destroy(p); return RL4RS_EHIP; } } while (0)
    AM_FAIL(al(&p->params, p->n_params)); AM_FAIL(al(&p->grad, p->n_params));
    AM_FAIL(al(&p->adam_m, p->n_params)); AM_FAIL(al(&p->adam_v, p->n_params));
    AM_HIP(hipMemcpyAsync(p->params, params_host, (size_t)p->n_params * 4, hipMemcpyHostToDevice, st));
    AM_HIP(hipMemsetAsync(p->grad, 0, (size_t)p->n_params * 4, st));
    AM_HIP(hipMemsetAsync(p->adam_m, 0, (size_t)p->n_params * 4, st));
    AM_HIP(hipMemsetAsync(p->adam_v, 0, (size_t)p->n_params * 4, st));
    const size_t B = c->max_rows, G = c->max_grad_rows;
    AM_FAIL(al(&p->proj, B * H1)); AM_FAIL(al(&p->h1, B * H1)); AM_FAIL(al(&p->h2, B * H2));
    if (G > 0) {
        AM_FAIL(al(&p->d_h1, G * H1)); AM_FAIL(al(&p->d_h2, G * H2)); AM_FAIL(al(&p->d_proj, G * H1));
        int64_t wmax = 0;
        for (int i = 0; i < AP_COUNT; ++i) if (sizes[i] > wmax) wmax = sizes[i];
        p->cx.chunk = 256;
        AM_FAIL(al(&p->cx.wt, wmax));
        if (amlp_fused_shape_ok(*c)) {
            AM_FAIL(al(&p->w2t, H1 * H2)); AM_FAIL(al(&p->w3t, H2 * K)); AM_FAIL(al(&p->w1at, std::max<int64_t>(H1 * E, 1)));
        }
        AM_FAIL(al(&p->cx.part, (size_t)((G + 255) / 256) * (wmax + std::max(H1, std::max(H2, K)))));
    }
    if (amlp_h16_shape_ok(*c)) {
        const size_t KB1 = (size_t)(E + 15) / 16, NT3 = (size_t)(K + 31) / 32;
        AM_FAIL(al(&p->w1p, 8 * KB1 * 512 + 256)); AM_FAIL(al(&p->w2p, 8 * 16 * 512 + 256)); AM_FAIL(al(&p->w3p, NT3 * 16 * 512 + NT3 * 32));
        AM_FAIL(al(&p->w1xp, 8 * (size_t)((D + 15) / 16) * 512 + 256));
    }
    AM_HIP(hipStreamSynchronize(st));
#undef AM_HIP
#undef AM_FAIL
    *out = p;
    return RL4RS_OK;
}

int rl4rs_amlp_params(rl4rs_amlp* p, float** params_dev, float** grad_dev, int64_t* count) {
    RL4RS_REQUIRE(p, "amlp_params: null handle");
    if (params_dev) { *params_dev = p->params; p->t_valid = false; }       // (a writable pointer leaves the library)
    if (grad_dev) *grad_dev = p->grad;
    if (count) *count = p->n_params;
    return RL4RS_OK;
}

int rl4rs_amlp_adam_state(rl4rs_amlp* p, float** m_dev, float** v_dev, int64_t* step) {
    RL4RS_REQUIRE(p, "amlp_adam_state: null handle");
    if (m_dev) *m_dev = p->adam_m;
    if (v_dev) *v_dev = p->adam_v;
    if (step) *step = p->adam_t;
    return RL4RS_OK;
}
int rl4rs_amlp_set_adam_step(rl4rs_amlp* p, int64_t step) {
    RL4RS_REQUIRE(p && step >= 0, "amlp_set_adam_step: bad argument");
    p->adam_t = step;
    return RL4RS_OK;
}

int rl4rs_amlp_copy_params(rl4rs_amlp* dst, const rl4rs_amlp* src, void* stream) {
    RL4RS_REQUIRE(dst && src && dst->n_params == src->n_params, "amlp_copy_params: handles differ");
    RL4RS_HIP_TRY(hipMemcpyAsync(dst->params, src->params, (size_t)src->n_params * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    dst->t_valid = false;
    return RL4RS_OK;
}

int rl4rs_amlp_soft_update(rl4rs_amlp* targ, const rl4rs_amlp* src, float tau, void* stream) {
    RL4RS_REQUIRE(targ && src && targ->n_params == src->n_params, "amlp_soft_update: handles differ");
    hipLaunchKernelGGL(k_soft_update, dim3((unsigned)((src->n_params + 255) / 256)), dim3(256), 0, (hipStream_t)stream, targ->params,
                       src->params, src->n_params, tau);
    RL4RS_LAUNCH_CHECK();
    targ->t_valid = false;
    return RL4RS_OK;
}

// obs [N / rep, obs_dim] (row r is the observation of rows r*rep .. r*rep + rep - 1), act [N, act_dim], out [N, out_dim]
int rl4rs_amlp_forward(rl4rs_amlp* p, int32_t N, int32_t rep, const float* obs, const float* act, float* out, void* stream) {
    RL4RS_REQUIRE(p && obs && out && N > 0 && rep > 0 && N <= p->c.max_rows && N % rep == 0 && (act || p->c.act_dim == 0) &&
                  (p->c.act_dim > 0 || rep == 1), "amlp_forward: bad argument (N=%d, rep=%d, max_rows=%d)", N, rep, p ? p->c.max_rows : -1);
    hipStream_t st = (hipStream_t)stream;
    const int D = p->c.obs_dim, E = p->c.act_dim, H1 = p->c.hidden1, H2 = p->c.hidden2, K = p->c.out_dim, R = N / rep;
    const float* P = p->params;
    const int64_t* o = p->off;
    int rc;
    if (amlp_fused_call(p, N, rep)) {
        // minibatch-sized: the three layers as ONE launch (amlp_fused.hpp); a trainable network's launch also rebuilds the transposed
        // weights its backward reads
        rl4rs_amlp* one[1] = {p};
        float* outs[1] = {out};
        return amlp_forward_fused(1, one, N, obs, act, outs, st);
    }
    if (E == 0) {
        if ((rc = launch_gemm_f32(obs, D, P + o[AP_W1], H1, P + o[AP_B1], p->h1, H1, N, H1, D, ACT_RELU, st))) return rc;
    } else if (rep == 1 && N <= 4096) {
        // cat([x, a]) W1 as two operand pairs of one launch
        if ((rc = launch_gemm_small(obs, D, P + o[AP_W1], H1, D, act, E, P + o[AP_W1] + (size_t)D * H1, H1, E, P + o[AP_B1], p->h1, H1, N, H1,
                                    ACT_RELU, st))) return rc;
    } else {
        if ((rc = launch_gemm_f32(obs, D, P + o[AP_W1], H1, P + o[AP_B1], p->proj, H1, R, H1, D, ACT_NONE, st))) return rc;
        if ((rc = launch_gemm_f32(act, E, P + o[AP_W1] + (size_t)D * H1, H1, nullptr, p->h1, H1, N, H1, E, ACT_RELU, st, p->proj, H1, rep)))
            return rc;
    }
    if ((rc = launch_gemm_f32(p->h1, H1, P + o[AP_W2], H2, P + o[AP_B2], p->h2, H2, N, H2, H1, ACT_RELU, st))) return rc;
    if (K == 1 && p->c.head_act == ACT_NONE) {
        hipLaunchKernelGGL(k_amlp_head1, dim3((N + 3) / 4), dim3(256), 0, st, p->h2, N, H2, P + o[AP_W3], P + o[AP_B3], out);
        RL4RS_LAUNCH_CHECK();
    } else if ((rc = launch_gemm_f32(p->h2, H2, P + o[AP_W3], K, P + o[AP_B3], out, K, N, K, H2, p->c.head_act, st))) {
        return rc;
    }
    p->last_n = N;
    p->last_rep = rep;
    p->t_valid = false;                 // (this launch sequence rebuilt nothing: a fused backward of these rows transposes first)
    return RL4RS_OK;
}

int rl4rs_amlp_h16_ok(const rl4rs_amlp* p) { return (p && p->w1p) ? 1 : 0; }

// The same forward for rows that will never see a backward (target values, greedy evaluation), fp16x2 arithmetic, ONE launch for the
// three layers (k_amlp_fwd_h16) behind the observation-side projection: the weights' fp16 hi / lo planes are rebuilt from the
// CURRENT fp32 parameters in front of every call (one small launch; the parameters may have been written through the raw
// pointer of rl4rs_amlp_params, so no dirty flag is trusted).  Leaves no activations: a following rl4rs_amlp_backward is refused.
int rl4rs_amlp_forward_h16(rl4rs_amlp* p, int32_t N, int32_t rep, const float* obs, const float* act, float* out, void* stream) {
    RL4RS_REQUIRE(p && obs && act && out && N > 0 && rep > 0 && N <= p->c.max_rows && N % rep == 0, "amlp_forward_h16: bad argument (N=%d, rep=%d, max_rows=%d)",
                  N, rep, p ? p->c.max_rows : -1);
    RL4RS_REQUIRE(p->w1p, "amlp_forward_h16: this network's shape has no fp16x2 form (rl4rs_amlp_h16_ok)");
    hipStream_t st = (hipStream_t)stream;
    const int D = p->c.obs_dim, E = p->c.act_dim, H1 = p->c.hidden1, H2 = p->c.hidden2, K = p->c.out_dim, R = N / rep;
    const float* P = p->params;
    const int64_t* o = p->off;
    int rc;
    // the observation-side projection: a few hundred distinct rows (a training minibatch) through the small fp32 form, thousands (an
    // evaluation batch) through the fp16x2 GEMM like the rest of this forward (the 128 x 64-tile fp32 form left half the CUs idle there)
    const bool proj16 = R >= 1024;
    const PackH16Desc pk[4] = {{P + o[AP_W1] + (size_t)D * H1, p->w1p, H1, E, H1}, {P + o[AP_W2], p->w2p, H2, H1, H2}, {P + o[AP_W3], p->w3p, K, H2, K},
                               {P + o[AP_W1], p->w1xp, H1, D, H1}};
    if ((rc = launch_pack_h16_dev(pk, proj16 ? 4 : 3, st))) return rc;
    if (proj16) rc = launch_gemm_h16(obs, D, p->w1xp, P + o[AP_B1], p->proj, H1, R, H1, D, ACT_NONE, st);
    else rc = launch_gemm_f32(obs, D, P + o[AP_W1], H1, P + o[AP_B1], p->proj, H1, R, H1, D, ACT_NONE, st);
    if (rc) return rc;
    AmlpFwdH16 a = {act, p->proj, reinterpret_cast<const char*>(p->w1p), reinterpret_cast<const char*>(p->w2p), reinterpret_cast<const char*>(p->w3p),
                    P + o[AP_B2], P + o[AP_B3], out, N, E, rep, K, p->c.head_act};
    if ((rc = launch_amlp_fwd_h16(a, st))) return rc;
    p->last_n = -1;
    p->last_rep = 0;
    return RL4RS_OK;
}

// dout [N, out_dim] = gradient wrt the head's PRE-activation output (the loss kernels fold the head activation's derivative in).
// want_param_grad: write the gradient of every parameter into the handle's flat gradient buffer; dact (optional) [N, act_dim]
// receives the gradient wrt the action input.  Must follow rl4rs_amlp_forward of the SAME rows.
int rl4rs_amlp_backward(rl4rs_amlp* p, int32_t N, int32_t rep, const float* obs, const float* act, const float* dout, float* dact,
                        int32_t want_param_grad, void* stream) {
    RL4RS_REQUIRE(p && obs && dout && N > 0 && rep > 0 && N % rep == 0 && (act || p->c.act_dim == 0), "amlp_backward: bad argument");
    RL4RS_REQUIRE(N <= p->c.max_grad_rows, "amlp_backward: N=%d exceeds max_grad_rows=%d", N, p->c.max_grad_rows);
    RL4RS_REQUIRE(N == p->last_n && rep == p->last_rep, "amlp_backward: does not follow the forward of the same rows (forward N=%d rep=%d)",
                  p->last_n, p->last_rep);
    RL4RS_REQUIRE(!dact || p->c.act_dim > 0, "amlp_backward: no action input to differentiate");
    hipStream_t st = (hipStream_t)stream;
    const int D = p->c.obs_dim, E = p->c.act_dim, H1 = p->c.hidden1, H2 = p->c.hidden2, K = p->c.out_dim, R = N / rep;
    const float* P = p->params;
    float* G = p->grad;
    const int64_t* o = p->off;
    auto ew = [](int n) { return dim3((n + 255) / 256); };
    const dim3 b256(256);
    int rc;
    if (amlp_fused_call(p, N, rep) && p->w2t && N <= TN4_MAX_SAMPLES) {
        // minibatch-sized: the whole input-gradient chain + every parameter gradient = two launches (amlp_fused.hpp; the transposed
        // weights were rebuilt by the forward's launch)
        rl4rs_amlp* one[1] = {p};
        const float* douts[1] = {dout};
        float* dacts[1] = {dact};
        return amlp_backward_fused(1, one, N, obs, act, douts, dacts, want_param_grad, st);
    }
    // per layer: ONE launch for the weight + bias gradient (k_gemm_tn with the column sums folded in) and ONE for the input
    // gradient (k_gemm_nt: no transposed weight copy, the ReLU derivative of the layer below in its epilogue)
    if (want_param_grad) st_tn_cs(p->cx, st, p->h2, H2, H2, dout, K, K, N, G + o[AP_W3], G + o[AP_B3]);
    if ((rc = launch_gemm_nt(dout, K, P + o[AP_W3], K, p->d_h2, H2, N, H2, K, st, p->h2, H2))) return rc;
    if (want_param_grad) st_tn_cs(p->cx, st, p->h1, H1, H1, p->d_h2, H2, H2, N, G + o[AP_W2], G + o[AP_B2]);
    if ((rc = launch_gemm_nt(p->d_h2, H2, P + o[AP_W2], H2, p->d_h1, H1, N, H1, H2, st, p->h1, H1))) return rc;
    if (want_param_grad) {
        const float* dp = p->d_h1;
        if (rep > 1) {
            hipLaunchKernelGGL(k_group_sum, ew(R * H1), b256, 0, st, p->d_h1, R, rep, H1, p->d_proj);
            dp = p->d_proj;
        }
        st_tn_cs(p->cx, st, obs, D, D, dp, H1, H1, R, G + o[AP_W1], G + o[AP_B1]);
        if (E > 0) st_tn(p->cx, st, act, E, E, p->d_h1, H1, H1, N, G + o[AP_W1] + (size_t)D * H1);
    }
    if (dact && (rc = launch_gemm_nt(p->d_h1, H1, P + o[AP_W1] + (size_t)D * H1, H1, dact, E, N, E, H1, st))) return rc;
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// torch.optim.Adam (see rl4rs_qnet_adam_step)
int rl4rs_amlp_adam_step(rl4rs_amlp* p, float lr, float beta1, float beta2, float eps, void* stream) {
    RL4RS_REQUIRE(p, "amlp_adam_step: null handle");
    hipStream_t st = (hipStream_t)stream;
    p->adam_t += 1;
    const double t = (double)p->adam_t;
    const double c2 = sqrt(1.0 - pow((double)beta2, t));
    const float lr_t = (float)(lr * c2 / (1.0 - pow((double)beta1, t)));
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((p->n_params + 255) / 256)), dim3(256), 0, st, p->params, p->grad, p->adam_m, p->adam_v,
                       (int)p->n_params, lr_t, beta1, beta2, (float)(eps * c2), (const float*)nullptr, 0.f);
    RL4RS_LAUNCH_CHECK();
    p->t_valid = false;
    return RL4RS_OK;
}

// torch.optim.Adam for n <= 8 networks in ONE launch, with the soft target updates that follow in the same phase: nets[i] steps
// with lr[i] when do_adam[i] != 0; targets[i] (may be NULL) receives (1 - tau) target + tau nets[i] - computed from the parameters
// AFTER this call's step.  Same element-wise arithmetic as rl4rs_amlp_adam_step / rl4rs_amlp_soft_update per network.
int rl4rs_amlp_adam_multi(int32_t n, rl4rs_amlp* const* nets, const float* lr, const int32_t* do_adam, rl4rs_amlp* const* targets, float beta1,
                          float beta2, float eps, float tau, void* stream) {
    RL4RS_REQUIRE(n >= 1 && n <= 8 && nets && lr && do_adam, "amlp_adam_multi: bad argument (n=%d)", n);
    AdamMulti a;
    memset(&a, 0, sizeof(a));
    a.n = n; a.b1 = beta1; a.b2 = beta2; a.tau = tau;
    for (int i = 0; i < n; ++i) {
        rl4rs_amlp* p = nets[i];
        RL4RS_REQUIRE(p, "amlp_adam_multi: null handle %d", i);
        rl4rs_amlp* tg = targets ? targets[i] : nullptr;
        RL4RS_REQUIRE(do_adam[i] || tg, "amlp_adam_multi: network %d has neither a step nor a target", i);
        RL4RS_REQUIRE(!tg || tg->n_params == p->n_params, "amlp_adam_multi: target %d has another shape", i);
        AdamMultiDesc& d = a.d[i];
        d.p = p->params; d.targ = tg ? tg->params : nullptr;
        if (tg) tg->t_valid = false;
        if (do_adam[i]) {
            p->t_valid = false;
            p->adam_t += 1;
            const double t = (double)p->adam_t;
            const double c2 = sqrt(1.0 - pow((double)beta2, t));
            d.g = p->grad; d.m = p->adam_m; d.v = p->adam_v;
            d.lr_t = (float)(lr[i] * c2 / (1.0 - pow((double)beta1, t)));
            d.eps_t = (float)(eps * c2);
        }
        a.start[i + 1] = a.start[i] + p->n_params;
    }
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)((a.start[n] + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// n <= 4 networks of the same input widths on the SAME rows (the twin critics): forward / backward of all of them as ONE launch each
// way when the call has the fused form (rl4rs_amlp_set_fused), else one rl4rs_amlp_forward / _backward per network (rep = 1).
int rl4rs_amlp_forward_multi(int32_t n, rl4rs_amlp* const* nets, int32_t N, const float* obs, const float* act, float* const* outs, void* stream) {
    RL4RS_REQUIRE(n >= 1 && n <= 4 && nets && outs && obs && N > 0, "amlp_forward_multi: bad argument (n=%d)", n);
    bool fused = true;
    for (int i = 0; i < n; ++i) {
        RL4RS_REQUIRE(nets[i] && outs[i], "amlp_forward_multi: null handle / output %d", i);
        RL4RS_REQUIRE(nets[i]->c.obs_dim == nets[0]->c.obs_dim && nets[i]->c.act_dim == nets[0]->c.act_dim, "amlp_forward_multi: network %d has other input widths", i);
        RL4RS_REQUIRE(N <= nets[i]->c.max_rows && (act || nets[i]->c.act_dim == 0), "amlp_forward_multi: bad argument for network %d", i);
        fused = fused && amlp_fused_call(nets[i], N, 1);
    }
    if (fused) return amlp_forward_fused(n, nets, N, obs, act, outs, (hipStream_t)stream);
    for (int i = 0; i < n; ++i) {
        const int rc = rl4rs_amlp_forward(nets[i], N, 1, obs, act, outs[i], stream);
        if (rc) return rc;
    }
    return RL4RS_OK;
}
int rl4rs_amlp_backward_multi(int32_t n, rl4rs_amlp* const* nets, int32_t N, const float* obs, const float* act, const float* const* douts,
                              float* const* dacts, int32_t want_param_grad, void* stream) {
    RL4RS_REQUIRE(n >= 1 && n <= 4 && nets && douts && obs && N > 0, "amlp_backward_multi: bad argument (n=%d)", n);
    bool fused = N <= TN4_MAX_SAMPLES;
    for (int i = 0; i < n; ++i) {
        RL4RS_REQUIRE(nets[i] && douts[i], "amlp_backward_multi: null handle / gradient %d", i);
        RL4RS_REQUIRE(nets[i]->c.obs_dim == nets[0]->c.obs_dim && nets[i]->c.act_dim == nets[0]->c.act_dim, "amlp_backward_multi: network %d has other input widths", i);
        RL4RS_REQUIRE(N <= nets[i]->c.max_grad_rows, "amlp_backward_multi: N=%d exceeds max_grad_rows=%d of network %d", N, nets[i]->c.max_grad_rows, i);
        RL4RS_REQUIRE(N == nets[i]->last_n && nets[i]->last_rep == 1, "amlp_backward_multi: network %d: does not follow the forward of the same rows", i);
        RL4RS_REQUIRE(!(dacts && dacts[i]) || nets[i]->c.act_dim > 0, "amlp_backward_multi: no action input to differentiate");
        fused = fused && amlp_fused_call(nets[i], N, 1) && nets[i]->w2t;
    }
    if (fused) return amlp_backward_fused(n, nets, N, obs, act, douts, dacts, want_param_grad, (hipStream_t)stream);
    for (int i = 0; i < n; ++i) {
        const int rc = rl4rs_amlp_backward(nets[i], N, 1, obs, act, douts[i], dacts ? dacts[i] : nullptr, want_param_grad, stream);
        if (rc) return rc;
    }
    return RL4RS_OK;
}

int rl4rs_cvae_sample(int32_t N, int32_t L, const float* enc_out, const float* eps, float min_logstd, float max_logstd, float* z,
                      void* stream) {
    RL4RS_REQUIRE(enc_out && eps && z && N > 0 && L > 0, "cvae_sample: bad argument");
    hipLaunchKernelGGL(k_cvae_sample, dim3((N * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, L, enc_out, eps, min_logstd, max_logstd, z);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_cvae_loss(int32_t N, int32_t E, int32_t L, const float* decoded, const float* actions, const float* enc_out, float min_logstd,
                    float max_logstd, float* d_dec_pre, float* rows_scratch, float* loss2, void* stream) {
    RL4RS_REQUIRE(decoded && actions && enc_out && d_dec_pre && rows_scratch && loss2 && N > 0 && E > 0 && L > 0, "cvae_loss: bad argument");
    hipStream_t st = (hipStream_t)stream;
    float2* rows = reinterpret_cast<float2*>(rows_scratch);
    hipLaunchKernelGGL(k_cvae_loss, dim3((N + 3) / 4), dim3(256), 0, st, N, E, L, decoded, actions, enc_out, min_logstd, max_logstd, d_dec_pre, rows);
    hipLaunchKernelGGL(k_q_mean2, dim3(1), dim3(256), 0, st, rows, N, loss2);     // {sum_e se / N, sum_l kl / N}: the caller divides by E, L
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_cvae_encoder_grad(int32_t N, int32_t L, const float* enc_out, const float* eps, const float* dz, float beta, float min_logstd,
                            float max_logstd, float* d_enc_out, void* stream) {
    RL4RS_REQUIRE(enc_out && eps && dz && d_enc_out && N > 0 && L > 0, "cvae_encoder_grad: bad argument");
    hipLaunchKernelGGL(k_cvae_enc_grad, dim3((N * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, L, enc_out, eps, dz, beta, min_logstd,
                       max_logstd, d_enc_out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_residual_action(int32_t N, int32_t E, const float* action, const float* tanh_out, float scale, float* out, void* stream) {
    RL4RS_REQUIRE(action && tanh_out && out && N > 0 && E > 0, "residual_action: bad argument");
    hipLaunchKernelGGL(k_residual_action, dim3((N * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, N * E, action, tanh_out, scale, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_residual_grad(int32_t N, int32_t E, const float* action, const float* tanh_out, float scale, const float* d_out, float* d_pre,
                        void* stream) {
    RL4RS_REQUIRE(action && tanh_out && d_out && d_pre && N > 0 && E > 0, "residual_grad: bad argument");
    hipLaunchKernelGGL(k_residual_grad, dim3((N * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, N * E, action, tanh_out, scale, d_out, d_pre);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_bcq_target(int32_t B, int32_t n, const float* q1, const float* q2, float lam, const float* rewards, const float* terminals,
                     float gamma, float* y, int32_t* best, void* stream) {
    RL4RS_REQUIRE(q1 && B > 0 && n > 0 && (y || best) && (!rewards || terminals), "bcq_target: bad argument");
    hipLaunchKernelGGL(k_bcq_target, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, B, n, q1, q2, lam, rewards, terminals, gamma, y, best);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_pick_rows(int32_t B, int32_t n, int32_t E, const float* rows, const int32_t* best, float* out, void* stream) {
    RL4RS_REQUIRE(rows && best && out && B > 0 && n > 0 && E > 0, "pick_rows: bad argument");
    hipLaunchKernelGGL(k_pick_rows, dim3((B * E + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, n, E, rows, best, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_critic_mse(int32_t N, const float* q1, const float* q2, const float* y, float* dq1, float* dq2, float* loss2, void* stream) {
    RL4RS_REQUIRE(q1 && q2 && y && dq1 && dq2 && loss2 && N > 0, "critic_mse: bad argument");
    hipLaunchKernelGGL(k_critic_mse, dim3(1), dim3(256), 0, (hipStream_t)stream, N, q1, q2, y, dq1, dq2, loss2);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_squashed_sample(int32_t N, int32_t rep, int32_t A, const float* head, const float* eps, float min_logstd, float max_logstd,
                          int32_t out_rep, int32_t out_off, float* act_out, float* logp_out, void* stream) {
    RL4RS_REQUIRE(head && act_out && N > 0 && rep > 0 && N % rep == 0 && A > 0 && out_rep >= rep && out_off >= 0 && out_off + rep <= out_rep,
                  "squashed_sample: bad argument");
    hipLaunchKernelGGL(k_squashed_sample, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, N, rep, A, head, eps, min_logstd, max_logstd,
                       out_rep, out_off, act_out, logp_out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_sac_actor_grad(int32_t B, int32_t A, const float* head, const float* eps, const float* act, const float* g_act, const float* log_temp,
                         float min_logstd, float max_logstd, float* d_head, void* stream) {
    RL4RS_REQUIRE(head && eps && act && g_act && log_temp && d_head && B > 0 && A > 0, "sac_actor_grad: bad argument");
    hipLaunchKernelGGL(k_sac_actor_grad, dim3((B * A + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, A, head, eps, act, g_act, log_temp,
                       min_logstd, max_logstd, d_head);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_twin_min(int32_t B, const float* q1, const float* q2, float* qmin, float* dq1, float* dq2, void* stream) {
    RL4RS_REQUIRE(q1 && q2 && qmin && B > 0 && (!dq1 == !dq2), "twin_min: bad argument");
    hipLaunchKernelGGL(k_twin_min, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, B, q1, q2, qmin, dq1, dq2);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_cql_critic_loss(int32_t B, int32_t m, const float* q1, const float* q2, const float* offs, const float* y, const float* alpha_w,
                          float* dq1, float* dq2, float* rows_scratch, float* sums6, void* stream) {
    RL4RS_REQUIRE(q1 && q2 && offs && rows_scratch && sums6 && B > 0 && m > 1 && (!y || (dq1 && dq2 && alpha_w)), "cql_critic_loss: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_cql_rows, dim3((B + 3) / 4), dim3(256), 0, st, B, m, q1, q2, offs, y, alpha_w, dq1, dq2, rows_scratch);
    hipLaunchKernelGGL(k_sum6, dim3(1), dim3(256), 0, st, rows_scratch, B, sums6);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}


// ---- one whole BCQ update as one host call (include/rl4rs_hip.h: rl4rs_bcq_step)
namespace {
struct BcqWs {
    float *enc_out, *z, *y, *d_dec, *rows, *loss2, *dz, *d_enc, *sampled, *t, *a_next, *q1n, *q2n, *yq, *q1v, *q2v, *dq1, *dq2, *closs2,
          *sampled2, *t2, *a_pi, *qv, *minus_inv_b, *da, *d_pre;
    int64_t total;
};
BcqWs bcq_ws(float* base, int64_t B, int64_t n, int64_t E, int64_t L) {
    BcqWs w;
    int64_t o = 0;
    auto take = [&](int64_t cnt) { float* p = base ? base + o : nullptr; o += (cnt + 3) / 4 * 4; return p; };
    const int64_t R = B * n;
    w.enc_out = take(B * 2 * L); w.z = take(B * L); w.y = take(B * E); w.d_dec = take(B * E); w.rows = take(B * 2); w.loss2 = take(2);
    w.dz = take(B * L); w.d_enc = take(B * 2 * L);
    w.sampled = take(R * E); w.t = take(R * E); w.a_next = take(R * E); w.q1n = take(R); w.q2n = take(R); w.yq = take(B);
    w.q1v = take(B); w.q2v = take(B); w.dq1 = take(B); w.dq2 = take(B); w.closs2 = take(2);
    w.sampled2 = take(B * E); w.t2 = take(B * E); w.a_pi = take(B * E); w.qv = take(B); w.minus_inv_b = take(B); w.da = take(B * E);
    w.d_pre = take(B * E);
    w.total = o;
    return w;
}
}  // namespace

int64_t rl4rs_bcq_workspace_floats(int32_t B, int32_t n, int32_t E, int32_t L) { return bcq_ws(nullptr, B, n, E, L).total; }

int rl4rs_bcq_update(const rl4rs_bcq_step* s, void* stream) {
    RL4RS_REQUIRE(s && s->imit_enc && s->imit_dec && s->policy && s->policy_targ && s->q1 && s->q2 && s->q1_targ && s->q2_targ, "bcq_update: null handle");
    RL4RS_REQUIRE(s->B > 0 && s->n > 0 && s->E > 0 && s->L > 0 && s->obs_dev && s->act_dev && s->rew_dev && s->nxt_dev && s->ter_dev && s->noise_dev &&
                  s->workspace_dev && s->metrics_dev && ((uintptr_t)s->workspace_dev & 15) == 0 && ((uintptr_t)s->noise_dev & 15) == 0,
                  "bcq_update: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, n = s->n, E = s->E, L = s->L, R = B * n;
    const BcqWs w = bcq_ws(s->workspace_dev, B, n, E, L);
    const float* eps = s->noise_dev;
    float* zt = s->noise_dev + (size_t)B * L;
    float* za = zt + (size_t)R * L;
    const float lo = -20.f, hi = 2.f;
    int rc;
#define BU(expr) do { if ((rc = (expr)) != RL4RS_OK) return rc; } while (0)
    {
        const long long nz = (long long)(R + B) * L;
        hipLaunchKernelGGL(k_bcq_prep, dim3((unsigned)((std::max<long long>(nz, B) + 255) / 256)), dim3(256), 0, st, zt, nz, w.minus_inv_b, B);
    }
    // --- imitator (BCQImpl.update_imitator: ConditionalVAE.compute_error)
    BU(rl4rs_amlp_forward(s->imit_enc, B, 1, s->obs_dev, s->act_dev, w.enc_out, stream));
    BU(rl4rs_cvae_sample(B, L, w.enc_out, eps, lo, hi, w.z, stream));
    BU(rl4rs_amlp_forward(s->imit_dec, B, 1, s->obs_dev, w.z, w.y, stream));
    BU(rl4rs_cvae_loss(B, E, L, w.y, s->act_dev, w.enc_out, lo, hi, w.d_dec, w.rows, w.loss2, stream));
    BU(rl4rs_amlp_backward(s->imit_dec, B, 1, s->obs_dev, w.z, w.d_dec, w.dz, 1, stream));
    BU(rl4rs_cvae_encoder_grad(B, L, w.enc_out, eps, w.dz, s->beta, lo, hi, w.d_enc, stream));
    BU(rl4rs_amlp_backward(s->imit_enc, B, 1, s->obs_dev, s->act_dev, w.d_enc, nullptr, 1, stream));
    {
        rl4rs_amlp* nets[2] = {s->imit_enc, s->imit_dec};
        const float lr[2] = {s->imitator_lr, s->imitator_lr};
        const int32_t on[2] = {1, 1};
        BU(rl4rs_amlp_adam_multi(2, nets, lr, on, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, stream));
    }
    if (s->do_rl) {
        // --- critic (DDPGBaseImpl.update_critic with BCQImpl.compute_target): the B * n target rows are never differentiated
        auto nograd_forward = [&](rl4rs_amlp* p, const float* act, float* out) {
            if (s->nograd_h16 && p->w1p && R >= s->h16_min_rows && ((uintptr_t)act & 15) == 0 && ((uintptr_t)out & 15) == 0)
                return rl4rs_amlp_forward_h16(p, R, n, s->nxt_dev, act, out, stream);
            return rl4rs_amlp_forward(p, R, n, s->nxt_dev, act, out, stream);
        };
        BU(nograd_forward(s->imit_dec, zt, w.sampled));
        BU(nograd_forward(s->policy_targ, w.sampled, w.t));
        BU(rl4rs_residual_action(R, E, w.sampled, w.t, s->action_flexibility, w.a_next, stream));
        BU(nograd_forward(s->q1_targ, w.a_next, w.q1n));
        BU(nograd_forward(s->q2_targ, w.a_next, w.q2n));
        BU(rl4rs_bcq_target(B, n, w.q1n, w.q2n, s->lam, s->rew_dev, s->ter_dev, s->gamma, w.yq, nullptr, stream));
        rl4rs_amlp* twin[2] = {s->q1, s->q2};
        float* qv2[2] = {w.q1v, w.q2v};
        BU(rl4rs_amlp_forward_multi(2, twin, B, s->obs_dev, s->act_dev, qv2, stream));
        BU(rl4rs_critic_mse(B, w.q1v, w.q2v, w.yq, w.dq1, w.dq2, w.closs2, stream));
        const float* dq[2] = {w.dq1, w.dq2};
        BU(rl4rs_amlp_backward_multi(2, twin, B, s->obs_dev, s->act_dev, dq, nullptr, 1, stream));
        const float lr[2] = {s->critic_lr, s->critic_lr};
        const int32_t on[2] = {1, 1};
        BU(rl4rs_amlp_adam_multi(2, twin, lr, on, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, stream));
        if (s->do_actor) {
            // --- actor (BCQImpl.compute_actor_loss: -Q_1(s, pi(s, decode(s, z))).mean()) + the soft target updates
            BU(rl4rs_amlp_forward(s->imit_dec, B, 1, s->obs_dev, za, w.sampled2, stream));
            BU(rl4rs_amlp_forward(s->policy, B, 1, s->obs_dev, w.sampled2, w.t2, stream));
            BU(rl4rs_residual_action(B, E, w.sampled2, w.t2, s->action_flexibility, w.a_pi, stream));
            BU(rl4rs_amlp_forward(s->q1, B, 1, s->obs_dev, w.a_pi, w.qv, stream));
            BU(rl4rs_amlp_backward(s->q1, B, 1, s->obs_dev, w.a_pi, w.minus_inv_b, w.da, 0, stream));
            BU(rl4rs_residual_grad(B, E, w.sampled2, w.t2, s->action_flexibility, w.da, w.d_pre, stream));
            BU(rl4rs_amlp_backward(s->policy, B, 1, s->obs_dev, w.sampled2, w.d_pre, nullptr, 1, stream));
            rl4rs_amlp* nets[3] = {s->policy, s->q1, s->q2};
            rl4rs_amlp* targ[3] = {s->policy_targ, s->q1_targ, s->q2_targ};
            const float lr3[3] = {s->actor_lr, 0.f, 0.f};
            const int32_t on3[3] = {1, 0, 0};
            BU(rl4rs_amlp_adam_multi(3, nets, lr3, on3, targ, 0.9f, 0.999f, 1e-8f, s->tau, stream));
        }
    }
#undef BU
    hipLaunchKernelGGL(k_bcq_metrics, dim3(1), dim3(256), 0, st, w.loss2, 1.0f / (float)E, s->beta / (float)L, s->do_rl ? w.closs2 : nullptr,
                       (s->do_rl && s->do_actor) ? w.qv : nullptr, B, s->metrics_dev);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}


// ---- one whole continuous-CQL update as one host call (include/rl4rs_hip.h: rl4rs_cql_step)
namespace {
struct CqlWs {
    float *head_nxt, *head_obs, *a_tmp, *logp, *acts, *offs, *q1m, *q2m, *rows, *sums, *aw, *a_next, *q1n, *q2n, *yq, *dq1, *dq2, *a_pi, *q1p, *q2p,
          *qmin, *dqa, *dqb, *g1, *g2, *d_head;
    int64_t total;
};
CqlWs cql_ws(float* base, int64_t B, int64_t n, int64_t A) {
    CqlWs w;
    int64_t o = 0;
    auto take = [&](int64_t cnt) { float* p = base ? base + o : nullptr; o += (cnt + 3) / 4 * 4; return p; };
    const int64_t m = 1 + 3 * n;
    w.head_nxt = take(B * 2 * A); w.head_obs = take(B * 2 * A); w.a_tmp = take(B * A); w.logp = take(B);
    w.acts = take(B * m * A); w.offs = take(B * m); w.q1m = take(B * m); w.q2m = take(B * m); w.rows = take(B * 6); w.sums = take(8); w.aw = take(4);
    w.a_next = take(B * A); w.q1n = take(B); w.q2n = take(B); w.yq = take(B); w.dq1 = take(B * m); w.dq2 = take(B * m);
    w.a_pi = take(B * A); w.q1p = take(B); w.q2p = take(B); w.qmin = take(B); w.dqa = take(B); w.dqb = take(B); w.g1 = take(B * A); w.g2 = take(B * A);
    w.d_head = take(B * 2 * A);
    w.total = o;
    return w;
}
}  // namespace

int64_t rl4rs_cql_workspace_floats(int32_t B, int32_t n, int32_t A) { return cql_ws(nullptr, B, n, A).total; }

int rl4rs_cql_update(const rl4rs_cql_step* s, void* stream) {
    RL4RS_REQUIRE(s && s->policy && s->q1 && s->q2 && s->q1_targ && s->q2_targ && s->log_temp_dev && s->log_alpha_dev, "cql_update: null handle");
    RL4RS_REQUIRE(s->B > 0 && s->n > 0 && s->A > 0 && s->obs_dev && s->act_dev && s->rew_dev && s->nxt_dev && s->ter_dev && s->normal_dev && s->uniform_dev &&
                  s->workspace_dev && s->metrics_dev && ((uintptr_t)s->workspace_dev & 15) == 0 && ((uintptr_t)s->normal_dev & 15) == 0,
                  "cql_update: bad argument");
    RL4RS_REQUIRE(s->policy->c.act_dim == 0 && s->policy->c.out_dim == 2 * s->A, "cql_update: the policy must be a plain encoder with a [mu | logstd] head");
    hipStream_t st = (hipStream_t)stream;
    const int B = s->B, n = s->n, A = s->A, m = 1 + 3 * n, R = B * m;
    const CqlWs w = cql_ws(s->workspace_dev, B, n, A);
    const float lo = -20.f, hi = 2.f;
    const float* eps_temp = s->normal_dev;
    const float* eps_alpha_t = eps_temp + (size_t)B * A;
    const float* eps_alpha_tp1 = eps_alpha_t + (size_t)B * n * A;
    const float* eps_critic_t = eps_alpha_tp1 + (size_t)B * n * A;
    const float* eps_critic_tp1 = eps_critic_t + (size_t)B * n * A;
    const float* eps_actor = eps_critic_tp1 + (size_t)B * n * A;
    const float* uni_alpha = s->uniform_dev;
    const float* uni_critic = uni_alpha + (size_t)B * n * A;
    const float log_uniform = (float)((double)A * log(0.5));
    int rc;
#define CU(expr) do { if ((rc = (expr)) != RL4RS_OK) return rc; } while (0)
    auto adam_c = [](int64_t step_after, double beta) { return (float)(1.0 / (1.0 - pow(beta, (double)step_after))); };
    // the [B, m, A] actions and [B, m] importance offsets of the conservative term (CQLImpl._compute_policy_is_values / _compute_random_is_values)
    auto conservative_rows = [&](const float* e_t, const float* e_tp1, const float* uni) {
        hipLaunchKernelGGL(k_cql_fill_rows, dim3((B * (1 + n) * A + 255) / 256), dim3(256), 0, st, w.acts, w.offs, s->act_dev, uni, B, m, n, A, log_uniform);
        int r2;
        if ((r2 = rl4rs_squashed_sample(B * n, n, A, w.head_obs, e_t, lo, hi, m, 1, w.acts, w.offs, stream))) return r2;
        return rl4rs_squashed_sample(B * n, n, A, w.head_nxt, e_tp1, lo, hi, m, 1 + n, w.acts, w.offs, stream);
    };
    // the policy does not change until the actor step: its heads on s' and s are computed once (s last: the handle keeps the
    // activations of s for the actor's backward)
    CU(rl4rs_amlp_forward(s->policy, B, 1, s->nxt_dev, nullptr, w.head_nxt, stream));
    CU(rl4rs_amlp_forward(s->policy, B, 1, s->obs_dev, nullptr, w.head_obs, stream));
    // --- temperature (SACImpl.update_temp)
    if (s->temp_lr > 0.f) {
        CU(rl4rs_squashed_sample(B, 1, A, w.head_obs, eps_temp, lo, hi, 1, 0, w.a_tmp, w.logp, stream));
        hipLaunchKernelGGL(k_sac_temp_step, dim3(1), dim3(256), 0, st, w.logp, B, A, s->log_temp_dev, s->log_temp_dev + 1, s->log_temp_dev + 2, s->temp_lr,
                           adam_c(s->temp_step + 1, 0.9), adam_c(s->temp_step + 1, 0.999), s->metrics_dev + 2);
    }
    // --- alpha (CQLImpl.update_alpha): the critics' values of the 31 rows are never differentiated here
    const bool alpha_on = s->alpha_lr > 0.f;
    if (alpha_on) {
        CU(conservative_rows(eps_alpha_t, eps_alpha_tp1, uni_alpha));
        auto nograd_forward = [&](rl4rs_amlp* p, float* out) {
            if (s->nograd_h16 && p->w1p && R >= s->h16_min_rows && ((uintptr_t)out & 15) == 0) return rl4rs_amlp_forward_h16(p, R, m, s->obs_dev, w.acts, out, stream);
            return rl4rs_amlp_forward(p, R, m, s->obs_dev, w.acts, out, stream);
        };
        CU(nograd_forward(s->q1, w.q1m));
        CU(nograd_forward(s->q2, w.q2m));
        CU(rl4rs_cql_critic_loss(B, m, w.q1m, w.q2m, w.offs, nullptr, nullptr, nullptr, nullptr, w.rows, w.sums, stream));
    }
    hipLaunchKernelGGL(k_cql_alpha_step, dim3(1), dim3(64), 0, st, w.sums, B, s->conservative_weight, s->alpha_threshold, s->log_alpha_dev,
                       s->log_alpha_dev + 1, s->log_alpha_dev + 2, s->alpha_lr, adam_c(s->alpha_step + 1, 0.9), adam_c(s->alpha_step + 1, 0.999),
                       alpha_on ? 1 : 0, s->metrics_dev + 3, w.aw);
    // --- critic (DDPGBaseImpl.update_critic with CQLImpl.compute_critic_loss / _compute_deterministic_target)
    CU(rl4rs_squashed_sample(B, 1, A, w.head_nxt, nullptr, lo, hi, 1, 0, w.a_next, nullptr, stream));
    {
        rl4rs_amlp* targ[2] = {s->q1_targ, s->q2_targ};
        float* qn[2] = {w.q1n, w.q2n};
        CU(rl4rs_amlp_forward_multi(2, targ, B, s->nxt_dev, w.a_next, qn, stream));
    }
    CU(rl4rs_bcq_target(B, 1, w.q1n, w.q2n, 1.0f, s->rew_dev, s->ter_dev, s->gamma, w.yq, nullptr, stream));
    CU(conservative_rows(eps_critic_t, eps_critic_tp1, uni_critic));
    CU(rl4rs_amlp_forward(s->q1, R, m, s->obs_dev, w.acts, w.q1m, stream));
    CU(rl4rs_amlp_forward(s->q2, R, m, s->obs_dev, w.acts, w.q2m, stream));
    CU(rl4rs_cql_critic_loss(B, m, w.q1m, w.q2m, w.offs, w.yq, w.aw, w.dq1, w.dq2, w.rows, w.sums, stream));
    CU(rl4rs_amlp_backward(s->q1, R, m, s->obs_dev, w.acts, w.dq1, nullptr, 1, stream));
    CU(rl4rs_amlp_backward(s->q2, R, m, s->obs_dev, w.acts, w.dq2, nullptr, 1, stream));
    rl4rs_amlp* twin[2] = {s->q1, s->q2};
    {
        const float lr[2] = {s->critic_lr, s->critic_lr};
        const int32_t on[2] = {1, 1};
        CU(rl4rs_amlp_adam_multi(2, twin, lr, on, nullptr, 0.9f, 0.999f, 1e-8f, 0.f, stream));
    }
    // --- actor (SACImpl.compute_actor_loss): (exp(log_temp) * logp - min_c Q_c(s, a)).mean()
    CU(rl4rs_squashed_sample(B, 1, A, w.head_obs, eps_actor, lo, hi, 1, 0, w.a_pi, w.logp, stream));
    {
        float* qp[2] = {w.q1p, w.q2p};
        CU(rl4rs_amlp_forward_multi(2, twin, B, s->obs_dev, w.a_pi, qp, stream));
        CU(rl4rs_twin_min(B, w.q1p, w.q2p, w.qmin, w.dqa, w.dqb, stream));
        const float* dq[2] = {w.dqa, w.dqb};
        float* ga[2] = {w.g1, w.g2};
        CU(rl4rs_amlp_backward_multi(2, twin, B, s->obs_dev, w.a_pi, dq, ga, 0, stream));
    }
    hipLaunchKernelGGL(k_add2, dim3((B * A + 255) / 256), dim3(256), 0, st, w.g1, w.g2, w.g1, B * A);
    CU(rl4rs_sac_actor_grad(B, A, w.head_obs, eps_actor, w.a_pi, w.g1, s->log_temp_dev, lo, hi, w.d_head, stream));
    // (the policy handle's activations are those of the forward on obs, the last one above)
    CU(rl4rs_amlp_backward(s->policy, B, 1, s->obs_dev, nullptr, w.d_head, nullptr, 1, stream));
    {
        rl4rs_amlp* nets[3] = {s->policy, s->q1, s->q2};
        rl4rs_amlp* targ[3] = {nullptr, s->q1_targ, s->q2_targ};
        const float lr3[3] = {s->actor_lr, 0.f, 0.f};
        const int32_t on3[3] = {1, 0, 0};
        CU(rl4rs_amlp_adam_multi(3, nets, lr3, on3, targ, 0.9f, 0.999f, 1e-8f, s->tau, stream));
    }
#undef CU
    hipLaunchKernelGGL(k_cql_metrics, dim3(1), dim3(256), 0, st, w.sums, B, s->conservative_weight, s->alpha_threshold, s->log_alpha_dev, s->log_temp_dev,
                       w.logp, w.qmin, s->metrics_dev, s->metrics_dev + 1);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // extern "C"
