// Action-masked policy net on gfx950: forward + sampling, loss + backward (A2C / PPO), Adam.
//
// Reference: rl4rs/nets/rllib/rllib_mask_model.py:7-64 (getMaskActionsModel): FullyConnectedNetwork with
// fcnet_hiddens=[64] (tanh, RLlib default), num_outputs = action_size, value head sharing the hidden layer
// (vf_share_layers=True, :34), and  logits + max(log(action_mask), float32.min)  (:61-62).
// Losses restate RLlib 1.5.1's a3c_tf_policy / ppo_tf_policy (third-party, not vendored: parity unpinned,
// checked against an fp64 autograd restatement in tests).  Hyper-parameters: script/modelfree_train.py:179-304.
//
// Parameters live in ONE flat fp32 buffer (= the all-reduce unit, rl4rs_amd/dist.py):
//   [ W1 (OD x HID) | b1 (HID) | W2e (HID x (A+1)) | b2e (A+1) ]      the value head is output column A of layer 2.
// Per-sample work is tiny (34.6 K MAC), so forward/backward run one wave per sample; the batch reductions of the
// parameter gradients are "A^T B" GEMMs over the sample axis on the matrix cores (exact fp32, split over samples
// into fixed chunks and summed in a fixed order => bit-reproducible gradients).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.hpp"
#include "gather_kernels.hpp"
#include "mfma4.hpp"

namespace rl4rs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct PolDims { int OD, HID, A, AE, W; };   // AE = A + 1 (logits | value), W = mask words

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// counter-based uniform STRICTLY inside (0,1): a pure function of (seed, step, row, action).  23 random bits + 0.5 is exactly
// representable in fp32, so the largest value is 1 - 2^-24; with 24 bits the top value rounded to 1.0f, and the Gumbel draw
// logit - log(-log(u)) became +inf - which let a MASKED action (logit -3.4e38) win about once per 2^24 draws
// (tests/test_gpu_policy.py::test_wider_hidden_layer_takes_the_same_paths caught it).
__device__ __forceinline__ float uniform01(uint32_t seed, uint32_t step, uint32_t row, uint32_t a) {
    uint32_t h = mix32(seed ^ mix32(step * 0x9E3779B9U + 0x85EBCA6BU) ^ mix32(row * 0xC2B2AE35U + a * 0x27D4EB2FU + 1U));
    h = mix32(h + a);
    return ((float)(h >> 9) + 0.5f) * (1.0f / 8388608.0f);
}

// Wave-wide reductions, result in every lane.  DPP row operations (quad_perm, row_half_mirror, row_mirror) reduce each row of
// 16 lanes in 4 steps of a few cycles, the four row results meet through v_readlane: ~10x less latency than the six dependent
// ds_bpermute steps of a __shfl_xor butterfly, which dominated the per-row loss code (8 reductions per sample).
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f32<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fmaxf(v, dpp_f32<0x4E>(v));      // quad_perm [2,3,0,1]
    v = fmaxf(v, dpp_f32<0x141>(v));     // row_half_mirror
    v = fmaxf(v, dpp_f32<0x140>(v));     // row_mirror
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f32<0xB1>(v);
    v += dpp_f32<0x4E>(v);
    v += dpp_f32<0x141>(v);
    v += dpp_f32<0x140>(v);
    const int b = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}

// Shared forward of one sample by one wave.  s_obs[OD], s_h[HID], s_out[AE] are this wave's LDS slots.
// On return: s_h = tanh hidden, s_out[0..A) = masked logits, s_out[A] = value; returns log-sum-exp of the logits.
__device__ __forceinline__ float policy_row_forward(const PolDims& d, const float* __restrict__ prm,
                                                    const float* __restrict__ obs_row,
                                                    const uint32_t* __restrict__ mask_row,
                                                    float* s_obs, float* s_h, float* s_out, int lane) {
    const float* W1 = prm;
    const float* b1 = W1 + (size_t)d.OD * d.HID;
    const float* W2 = b1 + d.HID;
    const float* b2 = W2 + (size_t)d.HID * d.AE;
    for (int k = lane; k < d.OD; k += 64) s_obs[k] = obs_row[k];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int j = lane; j < d.HID; j += 64) {
        float s = b1[j];
#pragma unroll 8
        for (int k = 0; k < d.OD; ++k) s = fmaf(s_obs[k], W1[(size_t)k * d.HID + j], s);
        s_h[j] = tanhf(s);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float mx = -3.4028235e38f;
    for (int a = lane; a < d.AE; a += 64) {
        float s = b2[a];
#pragma unroll 8
        for (int j = 0; j < d.HID; ++j) s = fmaf(s_h[j], W2[(size_t)j * d.AE + a], s);
        if (a < d.A) {
            // logits + max(log(mask), float32.min): 0 for allowed actions, -3.4028235e38 for masked ones
            bool ok = mask_row ? ((mask_row[a >> 5] >> (a & 31)) & 1u) : true;
            if (!ok) s = s + (-3.4028235e38f);
            mx = fmaxf(mx, s);
        }
        s_out[a] = s;
    }
    mx = wave_max(mx);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float se = 0.f;
    for (int a = lane; a < d.A; a += 64) se += expf(s_out[a] - mx);
    se = wave_sum(se);
    return mx + logf(se);
}

// Outputs of one sample from its masked logits / value in s_out (shared by the FC policy and the raw-state policy):
// entropy, optional logits copy, Gumbel-max draw (SAMPLE) or the given action, log-prob, value.
template <bool SAMPLE>
__device__ __forceinline__ void policy_row_outputs(const PolDims& d, const float* s_out, float lse, int n, int lane,
                                                   uint32_t seed, uint32_t step, int32_t* __restrict__ actions,
                                                   float* __restrict__ logp, float* __restrict__ value,
                                                   float* __restrict__ entropy, float* __restrict__ logits_out) {
    float ent = 0.f, best = -3.4028235e38f;
    int best_a = 0x7fffffff;
    for (int a = lane; a < d.A; a += 64) {
        const float l = s_out[a];
        const float lp = l - lse;
        const float p = expf(lp);
        if (p > 0.f) ent -= p * lp;
        if (logits_out) logits_out[(size_t)n * d.A + a] = l;
        if (SAMPLE) {
            const float u = uniform01(seed, step, (uint32_t)n, (uint32_t)a);
            const float g = l - logf(-logf(u));
            if (best_a == 0x7fffffff || g > best) { best = g; best_a = a; }
        }
    }
    ent = wave_sum(ent);
    int act;
    if (SAMPLE) {
        for (int o = 32; o > 0; o >>= 1) {
            float ob = __shfl_xor(best, o);
            int oa = __shfl_xor(best_a, o);
            if (oa != 0x7fffffff && (best_a == 0x7fffffff || ob > best || (ob == best && oa < best_a))) { best = ob; best_a = oa; }
        }
        act = best_a;
    } else {
        act = actions[n];
    }
    if (lane == 0) {
        if (SAMPLE) actions[n] = act;
        if (logp) logp[n] = s_out[act] - lse;
        if (value) value[n] = s_out[d.A];
        if (entropy) entropy[n] = ent;
    }
}

// act / evaluate: one wave per sample.  SAMPLE: Gumbel-max draw from the masked categorical.
template <bool SAMPLE>
__global__ __launch_bounds__(256) void k_policy_forward(PolDims d, const float* __restrict__ prm, int N,
                                                        const float* __restrict__ obs, const uint32_t* __restrict__ mask,
                                                        uint32_t seed, uint32_t step, int32_t* __restrict__ actions,
                                                        float* __restrict__ logp, float* __restrict__ value,
                                                        float* __restrict__ entropy, float* __restrict__ logits_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = d.OD + d.HID + d.AE;
    float* s_obs = reinterpret_cast<float*>(smem) + (size_t)wave * per;
    float* s_h = s_obs + d.OD;
    float* s_out = s_h + d.HID;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const uint32_t* mrow = mask ? mask + (size_t)n * d.W : nullptr;
    const float lse = policy_row_forward(d, prm, obs + (size_t)n * d.OD, mrow, s_obs, s_h, s_out, lane);
    policy_row_outputs<SAMPLE>(d, s_out, lse, n, lane, seed, step, actions, logp, value, entropy, logits_out);
}

// Raw-state policy head: [logits | value] rows come from a GEMM (context @ [out_w | value_w] + bias); this kernel adds the
// action mask (rllib_mask_model.py:61-62 form), then the same outputs as k_policy_forward.  One wave per sample.
template <bool SAMPLE>
__global__ __launch_bounds__(256) void k_rawpolicy_head(PolDims d, int N, const float* __restrict__ ext, int64_t ext_ld,
                                                        const uint32_t* __restrict__ mask, uint32_t seed, uint32_t step,
                                                        int32_t* __restrict__ actions, float* __restrict__ logp,
                                                        float* __restrict__ value, float* __restrict__ entropy,
                                                        float* __restrict__ logits_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* s_out = reinterpret_cast<float*>(smem) + (size_t)wave * d.AE;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const uint32_t* mrow = mask ? mask + (size_t)n * d.W : nullptr;
    float mx = -3.4028235e38f;
    for (int a = lane; a < d.AE; a += 64) {
        float v = ext[(size_t)n * ext_ld + a];
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
    policy_row_outputs<SAMPLE>(d, s_out, lse, n, lane, seed, step, actions, logp, value, entropy, logits_out);
}

struct LossArgs {
    int algo;            // 0 = A2C (a3c_tf_policy), 1 = PPO (ppo_tf_policy)
    float vf_coeff, ent_coeff, clip, vf_clip, kl_coeff, scale;   // scale = 1 (A2C: sums) or 1/N (PPO: means)
    const int32_t* actions; const float* adv; const float* ret;
    const float* old_logp; const float* old_value; const float* old_logits;   // PPO
};

// A2C / PPO loss of one sample from its masked logits / value in s_out and their log-sum-exp: writes d loss / d [logits |
// value] into s_d (LDS) and dOut (global, may be NULL), returns {pi_loss, vf_loss, entropy, kl}.  Shared by the FC mask policy
// and the raw-state policy.
// Store of a word ANOTHER workgroup reads later in the same launch (k_ppo_pass): write-through to the device's coherence point
// (global_store ... sc1), so the producer needs no L2 write-back fence before it arrives at the grid barrier - the guide's
// publish recipe R1 (write-through payload, drain, flag; consumer: one agent acquire, plain loads).
__device__ __forceinline__ void store_wt(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool WT = false>
__device__ __forceinline__ float4 policy_row_loss(const PolDims& d, const LossArgs& L, const float* s_out, float lse, int n, int lane,
                                                  float* s_d, float* __restrict__ dOut) {
    const int act = L.actions[n];
    const float adv = L.adv[n], ret = L.ret[n];
    const float old_lp = L.algo == 1 ? L.old_logp[n] : 0.f, old_v = L.algo == 1 ? L.old_value[n] : 0.f;
    const float v = s_out[d.A];
    const float lp_a = s_out[act] - lse;
    // PPO: this lane's old logits (columns lane, lane + 64, ...) are fetched ONCE, all loads in flight together; the four
    // passes below then run from registers (columns past 64 * OLR fall back to memory)
    constexpr int OLR = 8;
    float olr[OLR];
#pragma unroll
    for (int c = 0; c < OLR; ++c) olr[c] = 0.f;
    if (L.algo == 1) {                      // clamped addresses: unconditional loads issue together
#pragma unroll
        for (int c = 0; c < OLR; ++c) olr[c] = L.old_logits[(size_t)n * d.A + min(lane + 64 * c, d.A - 1)];
    }
    // entropy and (PPO) KL(old || new) need full sums first
    float ent = 0.f, kl = 0.f, old_lse = 0.f;
    if (L.algo == 1) {
        float om = -3.4028235e38f;
#pragma unroll
        for (int c = 0; c < OLR; ++c)
            if (lane + 64 * c < d.A) om = fmaxf(om, olr[c]);
        for (int a = lane + 64 * OLR; a < d.A; a += 64) om = fmaxf(om, L.old_logits[(size_t)n * d.A + a]);
        om = wave_max(om);
        float os = 0.f;
#pragma unroll
        for (int c = 0; c < OLR; ++c)
            if (lane + 64 * c < d.A) os += expf(olr[c] - om);
        for (int a = lane + 64 * OLR; a < d.A; a += 64) os += expf(L.old_logits[(size_t)n * d.A + a] - om);
        old_lse = om + logf(wave_sum(os));
    }
    // p = softmax probability and q = old probability of this lane's columns are kept for the gradient pass below (round 5: they
    // were recomputed there - 10 of the 30 libm expf per lane of a 284-action row; same expressions, same values)
    float pr[OLR], qr[OLR];
#pragma unroll
    for (int c = 0; c < OLR; ++c) {
        const int a = lane + 64 * c;
        pr[c] = 0.f; qr[c] = 0.f;
        if (a < d.A) {
            const float lp = s_out[a] - lse;
            const float p = expf(lp);
            pr[c] = p;
            if (p > 0.f) ent -= p * lp;
            if (L.algo == 1) {
                const float olp = olr[c] - old_lse;
                const float q = expf(olp);
                qr[c] = q;
                if (q > 0.f) kl += q * (olp - lp);
            }
        }
    }
    for (int a = lane + 64 * OLR; a < d.A; a += 64) {
        const float lp = s_out[a] - lse;
        const float p = expf(lp);
        if (p > 0.f) ent -= p * lp;
        if (L.algo == 1) {
            const float olp = L.old_logits[(size_t)n * d.A + a] - old_lse;
            const float q = expf(olp);
            if (q > 0.f) kl += q * (olp - lp);
        }
    }
    ent = wave_sum(ent);
    kl = wave_sum(kl);
    // d loss / d logp(action), d loss / d value
    float g_lp, g_v, pi_loss, vf_loss;
    if (L.algo == 0) {
        pi_loss = -lp_a * adv;
        vf_loss = 0.5f * (v - ret) * (v - ret);
        g_lp = -adv;
        g_v = L.vf_coeff * (v - ret);
    } else {
        const float ratio = expf(lp_a - old_lp);
        const float clipped = fminf(fmaxf(ratio, 1.f - L.clip), 1.f + L.clip);
        const float s1 = adv * ratio, s2 = adv * clipped;
        pi_loss = -fminf(s1, s2);
        g_lp = (s1 <= s2) ? -adv * ratio : 0.f;        // the clipped branch has zero gradient
        const float pv = old_v;
        const float l1 = (v - ret) * (v - ret);
        const float vc = pv + fminf(fmaxf(v - pv, -L.vf_clip), L.vf_clip);
        const float l2 = (vc - ret) * (vc - ret);
        vf_loss = fmaxf(l1, l2);
        float dv = (l1 >= l2) ? 2.f * (v - ret) : ((fabsf(v - pv) < L.vf_clip) ? 2.f * (vc - ret) : 0.f);
        g_v = L.vf_coeff * dv;
    }
    g_lp *= L.scale;
    g_v *= L.scale;
    const float ce = L.ent_coeff * L.scale, ck = (L.algo == 1) ? L.kl_coeff * L.scale : 0.f;
    for (int a = lane, c = 0; a < d.AE; a += 64, ++c) {
        float g;
        if (a < d.A) {
            const float lp = s_out[a] - lse;
            float p, q;
            switch (c) {                // register file is not indexable: select the lane's c-th kept pair
                case 0: p = pr[0]; q = qr[0]; break; case 1: p = pr[1]; q = qr[1]; break; case 2: p = pr[2]; q = qr[2]; break;
                case 3: p = pr[3]; q = qr[3]; break; case 4: p = pr[4]; q = qr[4]; break; case 5: p = pr[5]; q = qr[5]; break;
                case 6: p = pr[6]; q = qr[6]; break; case 7: p = pr[7]; q = qr[7]; break;
                default: p = expf(lp); q = L.algo == 1 ? expf(L.old_logits[(size_t)n * d.A + a] - old_lse) : 0.f;
            }
            // d logp_act/dl_a = [a==act] - p ; dH/dl_a = -p (log p + H) ; dKL/dl_a = p - q
            g = g_lp * ((a == act ? 1.f : 0.f) - p);
            if (p > 0.f) g += ce * p * (lp + ent);
            if (L.algo == 1) g += ck * (p - q);
        } else {
            g = g_v;
        }
        s_d[a] = g;
        if (dOut) {
            if (WT) store_wt(dOut + (size_t)n * d.AE + a, g); else dOut[(size_t)n * d.AE + a] = g;
        }
    }
    return make_float4(pi_loss, vf_loss, ent, kl);
}

// Training forward + per-sample backward down to the pre-activation of the hidden layer.
// Writes H [N,HID], dOut [N,AE] (d loss / d [logits | value]), dHpre [N,HID] and per-sample loss terms
// terms[n] = {pi_loss, vf_loss, entropy, kl}.
__global__ __launch_bounds__(256) void k_policy_train(PolDims d, const float* __restrict__ prm, int N,
                                                      const float* __restrict__ obs, const uint32_t* __restrict__ mask,
                                                      LossArgs L, float* __restrict__ H, float* __restrict__ dOut,
                                                      float* __restrict__ dHpre, float4* __restrict__ terms, int stage_w2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = d.OD + d.HID + 2 * d.AE;
    float* s_obs = reinterpret_cast<float*>(smem) + (size_t)wave * per;
    float* s_h = s_obs + d.OD;
    float* s_out = s_h + d.HID;
    float* s_d = s_out + d.AE;
    // backward needs row j of W2 per lane j: staged once per workgroup (rows of an odd length map the 64 lanes onto 64
    // different LDS banks; straight from memory that access is one cache line per lane per element)
    float* s_w2 = stage_w2 ? reinterpret_cast<float*>(smem) + (size_t)4 * per : nullptr;
    if (stage_w2) {
        const float* W2g = prm + (size_t)d.OD * d.HID + d.HID;
        for (int i = threadIdx.x; i < d.HID * d.AE; i += 256) s_w2[i] = W2g[i];
        __syncthreads();
    }
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const uint32_t* mrow = mask ? mask + (size_t)n * d.W : nullptr;
    const float lse = policy_row_forward(d, prm, obs + (size_t)n * d.OD, mrow, s_obs, s_h, s_out, lane);
    const float4 tm = policy_row_loss(d, L, s_out, lse, n, lane, s_d, dOut);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const float* W2 = prm + (size_t)d.OD * d.HID + d.HID;
    for (int j = lane; j < d.HID; j += 64) {
        float s = 0.f;
        const float* wr = (s_w2 ? s_w2 : W2) + (size_t)j * d.AE;
#pragma unroll 8
        for (int a = 0; a < d.AE; ++a) s = fmaf(s_d[a], wr[a], s);
        const float h = s_h[j];
        H[(size_t)n * d.HID + j] = h;
        dHpre[(size_t)n * d.HID + j] = s * (1.f - h * h);
    }
    if (lane == 0) terms[n] = tm;
}

// C_part[z][M][Nc] = sum over samples n in chunk z of A[n][m] * B[n][j]   ("A^T B" over the sample axis).
// One wave = one 32x32 tile; lane (i, half) feeds A[n = n0 + 2s + half][m0 + i] and B[..][j0 + i]: both are
// 128-byte coalesced reads of row-major sample-major matrices, no transpose needed.
// bias_part (optional): the column sums of B over the chunk (the bias gradient of the same Linear layer) from the waves of the
// first tile row, which hold every B value of their 32 columns anyway: bias_part[z][Nc].
__global__ __launch_bounds__(256) void k_gemm_tn(const float* __restrict__ A, int lda, int M,
                                                 const float* __restrict__ B, int ldb, int Nc, int Ns, int chunk,
                                                 float* __restrict__ part, float* __restrict__ bias_part) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31;
    const int tiles_n = (Nc + 31) / 32;
    const int tile = blockIdx.x * 4 + wave;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    if (tm * 32 >= M) return;
    const int m = tm * 32 + li, j = tn * 32 + li;
    const bool m_ok = m < M, j_ok = j < Nc;
    const int z = blockIdx.y;
    const int n_lo = z * chunk, n_hi = min(n_lo + chunk, Ns);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool want_cs = bias_part != nullptr && tm == 0;
    float cs = 0.f;
    // 8 sample pairs per trip: the 16 loads are issued together, then the 8 MFMAs (same accumulation order as a plain
    // loop; one load per MFMA made every step pay a full memory latency)
    // The NEXT trip's 16 values are requested before this trip's MFMAs (round 5: with load -> wait -> 8 MFMAs every trip paid a
    // memory round trip against 512 cycles of matrix work: 0.34 of the fp32 MFMA rate at the simulator trainers' 16 384-sample
    // reductions); same values, same order of accumulation: bit-identical.
    float an[8], bn[8];
    auto fetch = [&](int n) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nn = n + 2 * u + half;
            an[u] = (m_ok && nn < n_hi) ? A[(size_t)nn * lda + m] : 0.f;
            bn[u] = (j_ok && nn < n_hi) ? B[(size_t)nn * ldb + j] : 0.f;
        }
    };
    fetch(n_lo);
    for (int n = n_lo; n < n_hi; n += 16) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { a[u] = an[u]; b[u] = bn[u]; }
        fetch(n + 16);                              // (past the chunk: every lane's condition fails, nothing is read)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (n + 2 * u < n_hi) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        if (want_cs) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cs += b[u];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (want_cs) {
        cs += __shfl_xor(cs, 32);
        if (half == 0 && j_ok) bias_part[(size_t)z * Nc + j] = cs;
    }
    float* out = part + (size_t)z * M * Nc;
    for (int r = 0; r < 16; ++r) {
        int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < M && j_ok) out[(size_t)row * Nc + j] = acc[r];
    }
}

// The same reduction for LONG sample axes and wide outputs (round 5; the simulator trainers reduce over N * L = 16 384 (row, step)
// pairs): a 32 x 32 tile per wave re-reads A once per column tile and B once per row tile - 8 flop per byte, 536 MB of L2 / MALL
// traffic for the AUGRU's [256 x 512] gradient, 78 us where its MFMAs need 27.  Here a workgroup owns a 128 x 128 output tile:
// slabs of 16 samples of both operands go through LDS once (straight row copies: the reduction axis is the row axis of both
// operands, so the MFMA fragments are plain row reads, no transpose), each of the four waves multiplies a 64 x 64 quarter
// (4 MFMAs per pair of samples), the next slab's 4 float4 per thread are in flight during the current slab's 32 MFMAs.
// Same sample order inside a chunk as k_gemm_tn; chunk partials are summed by k_reduce_chunks in chunk order.
// Needs lda, ldb, M, Nc multiples of 4 and 16-byte aligned operands (float4 rows); chunk a multiple of 16.
__global__ __launch_bounds__(256) void k_gemm_tn_t128(const float* __restrict__ A, int lda, int M, const float* __restrict__ B, int ldb,
                                                      int Nc, int Ns, int chunk, float* __restrict__ part, float* __restrict__ bias_part) {
    __shared__ __attribute__((aligned(16))) float As[2][16][128];
    __shared__ __attribute__((aligned(16))) float Bs[2][16][128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int tiles_n = (Nc + 127) / 128;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n, z = blockIdx.y;
    const int m0 = tm * 128, j0 = tn * 128;
    const int n_lo = z * chunk, n_hi = min(n_lo + chunk, Ns);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int lr = tid >> 5, lc = (tid & 31) * 4;                  // this thread's float4: slab rows lr and lr + 8, columns lc .. lc + 3
    const bool a_ok = m0 + lc < M, b_ok = j0 + lc < Nc;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 ra[2], rb[2];
    auto fetch = [&](int n0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = n0 + lr + 8 * q;
            const bool in = row < n_hi;
            ra[q] = (in && a_ok) ? *reinterpret_cast<const f4*>(A + (size_t)row * lda + m0 + lc) : f4{0.f, 0.f, 0.f, 0.f};
            rb[q] = (in && b_ok) ? *reinterpret_cast<const f4*>(B + (size_t)row * ldb + j0 + lc) : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            *reinterpret_cast<f4*>(&As[buf][lr + 8 * q][lc]) = ra[q];
            *reinterpret_cast<f4*>(&Bs[buf][lr + 8 * q][lc]) = rb[q];
        }
    };
    const bool want_cs = bias_part != nullptr && tm == 0 && tid < 128;
    float cs = 0.f;
    fetch(n_lo);
    deposit(0);
    __syncthreads();
    int buf = 0;
    for (int n0 = n_lo; n0 < n_hi; n0 += 16, buf ^= 1) {
        const bool more = n0 + 16 < n_hi;
        if (more) fetch(n0 + 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float a0 = As[buf][2 * s + half][wm + li], a1 = As[buf][2 * s + half][wm + 32 + li];
            const float b0 = Bs[buf][2 * s + half][wn + li], b1 = Bs[buf][2 * s + half][wn + 32 + li];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (want_cs) {
#pragma unroll
            for (int k = 0; k < 16; ++k) cs += Bs[buf][k][tid];            // rows past the chunk hold zeros
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) deposit(buf ^ 1);
        __syncthreads();
    }
    if (want_cs && j0 + tid < Nc) bias_part[(size_t)z * Nc + j0 + tid] = cs;
    float* out = part + (size_t)z * M * Nc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j0 + wn + 32 * j + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < M && col < Nc) out[(size_t)row * Nc + col] = acc[i][j][r];
            }
        }
}

// The same reduction for a minibatch-sized sample axis (round 4): ONE 32x32 tile per workgroup, its four waves split the samples
// (a quarter each, interleaved by pairs) and meet in LDS - k_gemm_tn gives a tile to one wave, which then runs Ns / 2 dependent
// MFMAs (4 us at 256 samples) on 20 workgroups; here the same tile takes a quarter of that on 80.  Fixed summation order
// (wave 0 + 1 + 2 + 3), so results are reproducible; bias_out (optional) receives the column sums of B from tile row 0.
__device__ __forceinline__ void gemm_tn4_tile(const float* __restrict__ A, int lda, int M, const float* __restrict__ B, int ldb, int Nc,
                                              int Ns, float* __restrict__ out, float* __restrict__ bias_out, int tile) {
    __shared__ float red[4][16][64];
    __shared__ float cred[4][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int half = lane >> 5, li = lane & 31;
    const int tiles_n = (Nc + 31) / 32;
    const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
    const int m = tm * 32 + li, j = tn * 32 + li;
    const bool m_ok = m < M, j_ok = j < Nc;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool want_cs = bias_out != nullptr && tm == 0;
    float cs = 0.f;
    // wave w takes sample pairs w, w + 4, w + 8, ... (16 samples = 8 pairs per trip of the whole workgroup... per wave: 8 pairs)
    for (int n = wave * 16; n < Ns; n += 64) {
        float a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int nn = n + 2 * u + half;
            a[u] = (m_ok && nn < Ns) ? A[(size_t)nn * lda + m] : 0.f;
            b[u] = (j_ok && nn < Ns) ? B[(size_t)nn * ldb + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (n + 2 * u < Ns) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
        if (want_cs) {
#pragma unroll
            for (int u = 0; u < 8; ++u) cs += b[u];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    if (want_cs) {
        cs += __shfl_xor(cs, 32);
        if (half == 0) cred[wave][li] = cs;
    }
    __syncthreads();
    if (j_ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const int row = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < M) out[(size_t)row * Nc + j] = red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
        }
        if (want_cs && wave == 0 && half == 0) bias_out[j] = cred[0][li] + cred[1][li] + cred[2][li] + cred[3][li];
    }
}
__global__ __launch_bounds__(256) void k_gemm_tn4(const float* __restrict__ A, int lda, int M, const float* __restrict__ B, int ldb, int Nc,
                                                  int Ns, float* __restrict__ out, float* __restrict__ bias_out) {
    gemm_tn4_tile(A, lda, M, B, ldb, Nc, Ns, out, bias_out, blockIdx.x);
}
// Up to 16 such reductions over the SAME sample axis as one grid (round 5: every weight and bias gradient of a fused amlp
// backward - W3, W2, W1's observation rows, W1's action rows): tiles of problem i are workgroups [tile0[i], tile0[i + 1]).
constexpr int TN_GROUP_MAX = 16;
struct TnGroup {
    const float* A[TN_GROUP_MAX]; const float* B[TN_GROUP_MAX]; float* out[TN_GROUP_MAX]; float* bias[TN_GROUP_MAX];
    int lda[TN_GROUP_MAX], M[TN_GROUP_MAX], ldb[TN_GROUP_MAX], Nc[TN_GROUP_MAX], tile0[TN_GROUP_MAX + 1];
    int n, Ns;
};
__global__ __launch_bounds__(256) void k_gemm_tn4_group(TnGroup g) {
    int i = 0;
#pragma unroll
    for (int k = 1; k < TN_GROUP_MAX; ++k) i += (k < g.n && (int)blockIdx.x >= g.tile0[k]) ? 1 : 0;
    gemm_tn4_tile(g.A[i], g.lda[i], g.M[i], g.B[i], g.ldb[i], g.Nc[i], g.Ns, g.out[i], g.bias[i], (int)blockIdx.x - g.tile0[i]);
}

// column sums of X [Ns, ld] (first Nc columns) per sample chunk: part[z][Nc]
__global__ void k_colsum(const float* __restrict__ X, int ld, int Nc, int Ns, int chunk, float* __restrict__ part) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y;
    if (j >= Nc) return;
    const int n_lo = z * chunk, n_hi = min(n_lo + chunk, Ns);
    float s = 0.f;
    for (int n = n_lo; n < n_hi; n += 16) {       // 16 loads in flight, added in sample order
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) x[u] = (n + u < n_hi) ? X[(size_t)(n + u) * ld + j] : 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (n + u < n_hi) s += x[u];
    }
    part[(size_t)z * Nc + j] = s;
}

// dst[i] = sum_z part[z][i] in chunk order (fixed order => reproducible)
__global__ void k_reduce_chunks(const float* __restrict__ part, int count, int nz, float* __restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float s = 0.f;
    for (int z = 0; z < nz; ++z) s += part[(size_t)z * count + i];
    dst[i] = s;
}

// The same sum for MANY chunks (the 128 x 128 form splits the sample axis into up to 64): 64 outputs per workgroup, its four waves
// take the chunks z = w, w + 4, ... (eight independent loads in flight each) and meet in LDS; dst = ((w0 + w1) + w2) + w3 with
// every w a sum in ascending z: a fixed order, reproducible.
__global__ __launch_bounds__(256) void k_reduce_chunks4(const float* __restrict__ part, int count, int nz, float* __restrict__ dst) {
    __shared__ float sm[4][64];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int i = blockIdx.x * 64 + l;
    float s = 0.f;
    if (i < count) {
        int z = w;
        for (; z + 28 < nz; z += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(z + 4 * u) * count + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < nz; z += 4) s += part[(size_t)z * count + i];
    }
    sm[w][l] = s;
    __syncthreads();
    if (w == 0 && i < count) dst[i] = ((sm[0][l] + sm[1][l]) + sm[2][l]) + sm[3][l];
}

// stats[0..3] = sum over samples of {pi_loss, vf_loss, entropy, kl}; single block, fixed order
__global__ __launch_bounds__(256) void k_reduce_terms(const float4* __restrict__ terms, int N, float* __restrict__ stats) {
    __shared__ float4 sm[256];
    // eight loads in flight per thread, eight accumulators joined in a fixed order (one load at a time was a memory round trip per
    // term: 218 us for the 131 072 terms of a SeqSlate train batch)
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n0 = threadIdx.x; n0 < N; n0 += 256 * 8) {
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = terms[min(n0 + 256 * u, N - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (n0 + 256 * u < N) { acc[u].x += t[u].x; acc[u].y += t[u].y; acc[u].z += t[u].z; acc[u].w += t[u].w; }
    }
    float4 s;
    s.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
    s.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
    s.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
    s.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            float4 a = sm[threadIdx.x], b = sm[threadIdx.x + o];
            sm[threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { stats[0] = sm[0].x; stats[1] = sm[0].y; stats[2] = sm[0].z; stats[3] = sm[0].w; }
}

// acc[0..3] += x[0..3] (per-minibatch loss sums accumulated over a pass, fixed order)
__global__ void k_axpy4(const float* __restrict__ x, float* __restrict__ acc) {
    if (threadIdx.x < 4) acc[threadIdx.x] += x[threadIdx.x];
}

// sum of squares of a flat buffer -> out[0] (single block, fixed order)
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, int count, float* __restrict__ out) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += 256) s += g[i] * g[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

// Adam (tf.train.AdamOptimizer semantics: lr_t = lr * sqrt(1-b2^t)/(1-b1^t)); optional global-norm clipping
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                       int count, float lr_t, float b1, float b2, float eps, const float* __restrict__ sumsq, float clip) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float gi = g[i];
    if (clip > 0.f) {
        const float norm = sqrtf(sumsq[0]);
        if (norm > clip) gi *= clip / norm;      // tf.clip_by_global_norm
    }
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] -= lr_t * mi / (sqrtf(vi) + eps);
}


#include "ppo_pass.hpp"


// -------------------------------------------------------------------------------------------------
// act / evaluate / loss+backward of whole batches on MFMA tiles: the phase-A body of k_ppo_pass as an ordinary kernel, one
// workgroup of 8 waves per 8 samples (the per-row output / loss code is instruction bound, so one row per wave; the 32-row
// MFMA tiles stay mostly idle, which is free).  Against the one-wave-per-sample kernels above (every wave streams all
// 137 KB of weights through L2 and multiplies with scalar FMAs) the weights are read once per 8 samples and the products
// run on the matrix pipe.  MODE 0 = act (Gumbel-max sample), 1 = evaluate given actions, 2 = training forward + loss +
// backward to dHpre (needs w2t, the transposed W2e).
struct TileArgs {
    PolDims d;
    int N;
    const float* prm;
    const float* w2t;
    const float* obs;
    const uint32_t* mask;
    LossArgs L;
    uint32_t seed, step;
    int32_t* actions;
    float *logp, *value, *entropy, *logits_out;
    float *H, *dOut, *dHpre;
    float4* terms;
};

__global__ void k_w2_transpose(const float* __restrict__ W2, int HID, int AE, float* __restrict__ w2t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= HID * AE) return;
    const int j = i / AE, c = i - j * AE;
    w2t[(size_t)c * HID + j] = W2[i];
}

template <int MODE>
__global__ __launch_bounds__(512) void k_policy_tile(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PolDims d = a.d;
    const int OD = d.OD, HID = d.HID, AE = d.AE;
    constexpr int R = 8;
    const int SO = OD | 1, SH = HID | 1, SA = AE | 1;
    float* s_obs = reinterpret_cast<float*>(smem);                   // [R][SO]   (MFMA rows >= R read row R - 1)
    float* s_h = s_obs + R * SO;                                     // [R][SH]
    float* s_out = s_h + R * SH;                                     // [R][SA]
    float* s_d = s_out + R * SA;                                     // [R][SA]   (MODE 2)
    float* s_scr = s_d + (MODE == 2 ? R * SA : 0);                   // [8][1024] split-K partials
    uint32_t* s_mask = reinterpret_cast<uint32_t*>(s_scr + 8 * 1024);   // [R][W]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, li = lane & 31;
    const int lr = li < R ? li : R - 1;                              // the LDS row this lane feeds into the A operand
    const float* W1 = a.prm;
    const float* b1p = W1 + (size_t)OD * HID;
    const float* W2 = b1p + HID;
    const float* b2p = W2 + (size_t)HID * AE;
    const int NT1 = HID / 32, NT2 = (AE + 31) / 32, parts = 8 / NT1;
    const int r0 = blockIdx.x * R;
    const int nrow = min(R, a.N - r0);                               // live rows of this workgroup
    // stage observations (rows past N repeat the last live row) and mask words
    for (int i0 = tid; i0 < R * OD; i0 += 512 * 16) {
        float x[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = min(i0 + 512 * u, R * OD - 1);
            x[u] = a.obs[(size_t)(r0 + min(i / OD, nrow - 1)) * OD + i % OD];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = i0 + 512 * u;
            if (i < R * OD) s_obs[(i / OD) * SO + i % OD] = x[u];
        }
    }
    if (a.mask)
        for (int i = tid; i < R * d.W; i += 512) s_mask[i] = a.mask[(size_t)(r0 + min(i / d.W, nrow - 1)) * d.W + i % d.W];
    __syncthreads();
    {   // layer 1, split over K: wave -> (tile t, part q)
        const int t = wave % NT1, q = wave / NT1, kper = OD / parts;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = q * kper; k < (q + 1) * kper; k += 64) {
            float bv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) bv[u] = W1[(size_t)(k + 2 * u + half) * HID + t * 32 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_obs[lr * SO + k + 2 * u + half], bv[u], acc, 0, 0, 0);
        }
        float* part = s_scr + (size_t)(t * parts + q) * 1024;
        for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < R * HID; i += 512) {
        const int r = i / HID, j = i - r * HID, t = j >> 5, c = j & 31;
        float sum = b1p[j];
        for (int q = 0; q < parts; ++q) sum += s_scr[(size_t)(t * parts + q) * 1024 + r * 32 + c];
        const float h = tanhf(sum);
        s_h[r * SH + j] = h;
        if (MODE == 2 && r < nrow) a.H[(size_t)(r0 + r) * HID + j] = h;
    }
    __syncthreads();
    for (int t = wave; t < NT2; t += 8) {   // layer 2 (+ action mask)
        const int col = t * 32 + li;
        const bool c_ok = col < AE;
        const int colc = c_ok ? col : AE - 1;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k = 0; k < HID; k += 64) {
            float bv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) bv[u] = W2[(size_t)(k + 2 * u + half) * AE + colc];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(s_h[lr * SH + k + 2 * u + half], bv[u], acc, 0, 0, 0);
        }
        const float bias = b2p[colc];
        if (c_ok)
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < R) {
                    float v = acc[r] + bias;
                    const uint32_t mw = a.mask ? s_mask[row * d.W + (colc >> 5)] : 0xffffffffu;
                    if (col < d.A && !((mw >> (col & 31)) & 1u)) v = v + (-3.4028235e38f);
                    s_out[row * SA + col] = v;
                }
            }
    }
    __syncthreads();
    {   // one row per wave: log-sum-exp, then outputs (act / evaluate) or loss (train)
        const int row = wave;
        const float* so = s_out + row * SA;
        float mx = -3.4028235e38f;
        for (int c = lane; c < d.A; c += 64) mx = fmaxf(mx, so[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < d.A; c += 64) se += expf(so[c] - mx);
        const float lse = mx + logf(wave_sum(se));
        if (MODE != 2) {
            if (row < nrow)
                policy_row_outputs<MODE == 0>(d, so, lse, r0 + row, lane, a.seed, a.step, a.actions, a.logp, a.value, a.entropy, a.logits_out);
            return;
        }
        if (row < nrow) {
            const float4 tm = policy_row_loss(d, a.L, so, lse, r0 + row, lane, s_d + row * SA, a.dOut);
            if (lane == 0) a.terms[r0 + row] = tm;
        } else {
            for (int c = lane; c < AE; c += 64) s_d[row * SA + c] = 0.f;
        }
    }
    __syncthreads();
    {   // dH = dOut W2e^T, split over K; B operand from the transposed copy (coalesced)
        const int t = wave % NT1, q = wave / NT1;
        const int kper = ((AE + parts - 1) / parts + 1) & ~1;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* drow = s_d + lr * SA;
        const int k_hi = min((q + 1) * kper, AE);
        for (int k = q * kper; k < k_hi; k += 72) {
            float wv[36];
#pragma unroll
            for (int u = 0; u < 36; ++u) wv[u] = a.w2t[(size_t)min(k + 2 * u + half, AE - 1) * HID + t * 32 + li];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 36; ++u) {
                const int kk = k + 2 * u + half;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk < k_hi ? drow[min(kk, AE - 1)] : 0.f, wv[u], acc, 0, 0, 0);
            }
        }
        float* part = s_scr + (size_t)(t * parts + q) * 1024;
        for (int r = 0; r < 16; ++r) part[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
    }
    __syncthreads();
    for (int i = tid; i < nrow * HID; i += 512) {
        const int r = i / HID, j = i - r * HID, t = j >> 5, c = j & 31;
        float sum = 0.f;
        for (int q = 0; q < parts; ++q) sum += s_scr[(size_t)(t * parts + q) * 1024 + r * 32 + c];
        const float h = s_h[r * SH + j];
        a.dHpre[(size_t)(r0 + r) * HID + j] = sum * (1.f - h * h);
    }
}

#include "policy_tile_std.hpp"

}  // namespace rl4rs

using namespace rl4rs;

struct rl4rs_policy {
    PolDims d;
    int max_rows, n_params, nz, chunk;
    float *params, *adam_m, *adam_v;
    float *H, *dOut, *dHpre, *part, *sumsq, *w2t;
    float4* terms;
    unsigned* bar;
    unsigned* dead_host;       // pinned host mirror of bar[1] (a persistent pass gave up): read before every pass launch, no sync
    int64_t adam_t;
    bool train_attr, pass_launched, tile_attr[3];
    int pass_resident_wgs;     // workgroups of k_ppo_pass the device can hold at once (-1 = not queried yet)
    // rl4rs_policy_set_option (include/rl4rs_hip.h RL4RS_POLICY_OPT_*): kernel-path selection for A/B runs and tests
    bool opt_tile, opt_ppo_fused, opt_ppo_std;
    int opt_ppo_rows, opt_resident_cap;
    std::vector<void*> owned;
};

extern "C" {

int rl4rs_policy_param_count(int32_t obs_dim, int32_t hidden, int32_t action_size) {
    return obs_dim * hidden + hidden + hidden * (action_size + 1) + (action_size + 1);
}

int rl4rs_policy_create(int32_t obs_dim, int32_t hidden, int32_t action_size, int32_t max_rows,
                        const float* params_host, void* stream, rl4rs_policy** out) {
    RL4RS_REQUIRE(out && params_host && obs_dim > 0 && hidden > 0 && hidden <= 1024 && action_size > 1 && max_rows > 0,
                  "policy_create: bad argument");
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    rl4rs_policy* p = new rl4rs_policy();
    p->d.OD = obs_dim; p->d.HID = hidden; p->d.A = action_size; p->d.AE = action_size + 1; p->d.W = (action_size + 31) / 32;
    p->max_rows = max_rows;
    p->n_params = rl4rs_policy_param_count(obs_dim, hidden, action_size);
    p->chunk = 512;
    p->nz = (max_rows + p->chunk - 1) / p->chunk;
    p->adam_t = 0;
    p->train_attr = false;
    p->pass_launched = false;
    p->pass_resident_wgs = -1;
    p->opt_tile = true; p->opt_ppo_fused = true; p->opt_ppo_std = true; p->opt_ppo_rows = 0; p->opt_resident_cap = -1;
    p->dead_host = nullptr;
    p->tile_attr[0] = p->tile_attr[1] = p->tile_attr[2] = false;
    int rc;
    auto alloc = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) p->owned.push_back(*dst);
        return r;
    };
    if ((rc = alloc(&p->params, p->n_params))) return rc;
    if ((rc = alloc(&p->adam_m, p->n_params))) return rc;
    if ((rc = alloc(&p->adam_v, p->n_params))) return rc;
    if ((rc = alloc(&p->H, (size_t)max_rows * hidden))) return rc;
    if ((rc = alloc(&p->dOut, (size_t)max_rows * p->d.AE))) return rc;
    if ((rc = alloc(&p->dHpre, (size_t)max_rows * hidden))) return rc;
    size_t part_n = (size_t)p->nz * ((size_t)obs_dim * hidden > (size_t)hidden * p->d.AE ? (size_t)obs_dim * hidden : (size_t)hidden * p->d.AE);
    if ((rc = alloc(&p->part, part_n))) return rc;
    if ((rc = alloc(&p->sumsq, 4))) return rc;
    if ((rc = alloc(&p->w2t, (size_t)hidden * p->d.AE))) return rc;
    // bar[0] arrival counter, bar[1] sticky timeout flag, bar[64 .. 315] one arrival word per workgroup (grids of up to 252: ppo_pass.hpp)
    { float* b4; if ((rc = alloc(&b4, 320))) return rc; p->bar = reinterpret_cast<unsigned*>(b4); }
    float* t4;
    if ((rc = alloc(&t4, (size_t)max_rows * 4))) return rc;
    p->terms = reinterpret_cast<float4*>(t4);
    hipStream_t st = (hipStream_t)stream;
    RL4RS_HIP_TRY(hipMemcpyAsync(p->params, params_host, (size_t)p->n_params * 4, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemsetAsync(p->adam_m, 0, (size_t)p->n_params * 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(p->adam_v, 0, (size_t)p->n_params * 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(p->bar, 0, 320 * 4, st));
    RL4RS_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->dead_host), 64, hipHostMallocDefault));
    *p->dead_host = 0u;
    RL4RS_HIP_TRY(hipStreamSynchronize(st));
    *out = p;
    return RL4RS_OK;
}

int rl4rs_policy_destroy(rl4rs_policy* p) {
    if (!p) return RL4RS_OK;
    for (void* q : p->owned) (void)hipFree(q);
    if (p->dead_host) (void)hipHostFree(p->dead_host);
    delete p;
    return RL4RS_OK;
}

int rl4rs_policy_params(rl4rs_policy* p, float** params_dev, int32_t* count) {
    RL4RS_REQUIRE(p && params_dev, "policy_params: null argument");
    *params_dev = p->params;
    if (count) *count = p->n_params;
    return RL4RS_OK;
}

int rl4rs_policy_adam_state(rl4rs_policy* p, float** m_dev, float** v_dev, int64_t* step) {
    RL4RS_REQUIRE(p, "policy_adam_state: null handle");
    if (m_dev) *m_dev = p->adam_m;
    if (v_dev) *v_dev = p->adam_v;
    if (step) *step = p->adam_t;
    return RL4RS_OK;
}

int rl4rs_policy_set_adam_step(rl4rs_policy* p, int64_t step) {
    RL4RS_REQUIRE(p && step >= 0, "policy_set_adam_step: bad argument");
    p->adam_t = step;
    return RL4RS_OK;
}

static size_t fwd_smem(const PolDims& d, int extra) { return (size_t)4 * (d.OD + d.HID + d.AE + extra) * 4; }

}  // extern "C"

namespace {

// does k_policy_tile's tiling fit this policy?  (RL4RS_POLICY_OPT_TILE = 0 keeps the one-wave-per-sample kernels: A/B measurements)
bool tile_fits(const rl4rs_policy* p) {
    const PolDims& d = p->d;
    const int NT1 = d.HID / 32;
    return p->opt_tile && d.HID % 64 == 0 && (NT1 == 1 || NT1 == 2 || NT1 == 4 || NT1 == 8) && d.OD % 32 == 0 && (d.OD / (8 / NT1)) % 64 == 0;
}
size_t tile_smem(const PolDims& d, int mode) {
    return (size_t)(8 * ((d.OD | 1) + (d.HID | 1) + (mode == 2 ? 2 : 1) * (d.AE | 1) + d.W) + 8 * 1024) * 4;
}
// the default shape takes the 4x4x1 form (policy_tile_std.hpp); RL4RS_POLICY_OPT_PPO_STD = 0 keeps the 32x32x2 one for A/B runs
bool tile_is_std(const rl4rs_policy* p) {
    const PolDims& d = p->d;
    return p->opt_ppo_std && d.OD == 256 && d.HID == 64 && d.A == 284 && d.AE == 285 && d.W == 9;
}
template <int MODE>
int launch_policy_tile(rl4rs_policy* p, const TileArgs& a, hipStream_t st) {
    if (tile_is_std(p)) {
        int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_policy_tile_std<MODE>), TILE_STD_SMEM);      // (72 KB: above the 64 KB default)
        if (rca) return rca;
        hipLaunchKernelGGL(k_policy_tile_std<MODE>, dim3((a.N + 7) / 8), dim3(512), TILE_STD_SMEM, st, a);
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
    const size_t smem = tile_smem(p->d, MODE);
    if (!p->tile_attr[MODE]) {       // per-function limit, only ever raised (policies of different shapes share the kernel)
        int rca = raise_dyn_smem(reinterpret_cast<const void*>(&k_policy_tile<MODE>), smem);
        if (rca) return rca;
        p->tile_attr[MODE] = true;
    }
    hipLaunchKernelGGL(k_policy_tile<MODE>, dim3((a.N + 7) / 8), dim3(512), smem, st, a);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

}  // namespace

extern "C" {

int rl4rs_policy_act(rl4rs_policy* p, int32_t N, const float* obs, const uint32_t* mask_bits, uint32_t seed,
                     uint32_t step, int32_t* actions, float* logp, float* value, float* entropy, float* logits,
                     void* stream) {
    RL4RS_REQUIRE(p && obs && actions && N > 0, "policy_act: bad argument");
    if (tile_fits(p)) {
        TileArgs a;
        memset(&a, 0, sizeof(a));
        a.d = p->d; a.N = N; a.prm = p->params; a.obs = obs; a.mask = mask_bits; a.seed = seed; a.step = step;
        a.actions = actions; a.logp = logp; a.value = value; a.entropy = entropy; a.logits_out = logits;
        return launch_policy_tile<0>(p, a, (hipStream_t)stream);
    }
    hipLaunchKernelGGL(k_policy_forward<true>, dim3((N + 3) / 4), dim3(256), fwd_smem(p->d, 0), (hipStream_t)stream, p->d,
                       p->params, N, obs, mask_bits, seed, step, actions, logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_policy_evaluate(rl4rs_policy* p, int32_t N, const float* obs, const uint32_t* mask_bits,
                          const int32_t* actions, float* logp, float* value, float* entropy, float* logits,
                          void* stream) {
    RL4RS_REQUIRE(p && obs && actions && N > 0, "policy_evaluate: bad argument");
    if (tile_fits(p)) {
        TileArgs a;
        memset(&a, 0, sizeof(a));
        a.d = p->d; a.N = N; a.prm = p->params; a.obs = obs; a.mask = mask_bits;
        a.actions = const_cast<int32_t*>(actions); a.logp = logp; a.value = value; a.entropy = entropy; a.logits_out = logits;
        return launch_policy_tile<1>(p, a, (hipStream_t)stream);
    }
    hipLaunchKernelGGL(k_policy_forward<false>, dim3((N + 3) / 4), dim3(256), fwd_smem(p->d, 0), (hipStream_t)stream, p->d,
                       p->params, N, obs, mask_bits, 0u, 0u, const_cast<int32_t*>(actions), logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_policy_loss_grad(rl4rs_policy* p, int32_t algo, int32_t N, const float* obs, const uint32_t* mask_bits,
                           const int32_t* actions, const float* adv, const float* ret, const float* old_logp,
                           const float* old_value, const float* old_logits, float vf_coeff, float ent_coeff,
                           float clip, float vf_clip, float kl_coeff, float* grad_dev, float* stats_dev,
                           void* stream) {
    RL4RS_REQUIRE(p && obs && actions && adv && ret && grad_dev && N > 0 && N <= p->max_rows,
                  "policy_loss_grad: bad argument (N=%d, max_rows=%d)", N, p ? p->max_rows : -1);
    RL4RS_REQUIRE(algo == 0 || (algo == 1 && old_logp && old_value && old_logits), "policy_loss_grad: PPO needs old_* inputs");
    hipStream_t st = (hipStream_t)stream;
    const PolDims& d = p->d;
    LossArgs L;
    L.algo = algo; L.vf_coeff = vf_coeff; L.ent_coeff = ent_coeff; L.clip = clip; L.vf_clip = vf_clip; L.kl_coeff = kl_coeff;
    L.scale = algo == 0 ? 1.0f : 1.0f / (float)N;
    L.actions = actions; L.adv = adv; L.ret = ret; L.old_logp = old_logp; L.old_value = old_value; L.old_logits = old_logits;
    const size_t w2_bytes = (size_t)d.HID * d.AE * 4;
    const int stage_w2 = (fwd_smem(d, d.AE) + w2_bytes <= (size_t)150 * 1024) ? 1 : 0;
    const bool tiled = tile_fits(p);
    if (tiled) {
        hipLaunchKernelGGL(k_w2_transpose, dim3((d.HID * d.AE + 255) / 256), dim3(256), 0, st, p->params + (size_t)d.OD * d.HID + d.HID,
                           d.HID, d.AE, p->w2t);
        TileArgs a;
        memset(&a, 0, sizeof(a));
        a.d = d; a.N = N; a.prm = p->params; a.w2t = p->w2t; a.obs = obs; a.mask = mask_bits; a.L = L;
        a.H = p->H; a.dOut = p->dOut; a.dHpre = p->dHpre; a.terms = p->terms;
        int rc = launch_policy_tile<2>(p, a, st);
        if (rc) return rc;
    }
    if (!tiled && stage_w2 && !p->train_attr) {
        RL4RS_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_policy_train), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          150 * 1024));
        p->train_attr = true;
    }
    if (!tiled)
        hipLaunchKernelGGL(k_policy_train, dim3((N + 3) / 4), dim3(256), fwd_smem(d, d.AE) + (stage_w2 ? w2_bytes : 0), st, d, p->params,
                           N, obs, mask_bits, L, p->H, p->dOut, p->dHpre, p->terms, stage_w2);
    RL4RS_LAUNCH_CHECK();
    // sample chunks of the gradient reductions: 512 samples each up to 64 chunks, then longer chunks - a SeqSlate train batch (131 072
    // samples) was 256 partial matrices per gradient: 83 us of k_gemm_tn + 61 us of k_reduce_chunks twice per update
    int chunk = p->chunk;
    if ((N + chunk - 1) / chunk > 64) chunk = (((N + 63) / 64) + 63) / 64 * 64;
    const int nz = (N + chunk - 1) / chunk;
    float* gW1 = grad_dev;
    float* gb1 = gW1 + (size_t)d.OD * d.HID;
    float* gW2 = gb1 + d.HID;
    float* gb2 = gW2 + (size_t)d.HID * d.AE;
    auto tn = [&](const float* A, int lda, int M, const float* B, int ldb, int Nc, float* dst) {
        int tiles = ((M + 31) / 32) * ((Nc + 31) / 32);
        // one chunk (PPO minibatches): the partial IS the result, no reduction pass
        hipLaunchKernelGGL(k_gemm_tn, dim3((tiles + 3) / 4, nz), dim3(256), 0, st, A, lda, M, B, ldb, Nc, N, chunk, nz == 1 ? dst : p->part, (float*)nullptr);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((M * Nc + 255) / 256), dim3(256), 0, st, p->part, M * Nc, nz, dst);
    };
    auto cs = [&](const float* X, int ld, int Nc, float* dst) {
        hipLaunchKernelGGL(k_colsum, dim3((Nc + 63) / 64, nz), dim3(64), 0, st, X, ld, Nc, N, chunk, nz == 1 ? dst : p->part);
        if (nz > 1) hipLaunchKernelGGL(k_reduce_chunks, dim3((Nc + 255) / 256), dim3(256), 0, st, p->part, Nc, nz, dst);
    };
    tn(obs, d.OD, d.OD, p->dHpre, d.HID, d.HID, gW1);      // dW1  = obs^T dHpre
    cs(p->dHpre, d.HID, d.HID, gb1);                        // db1
    tn(p->H, d.HID, d.HID, p->dOut, d.AE, d.AE, gW2);       // dW2e = H^T dOut
    cs(p->dOut, d.AE, d.AE, gb2);                           // db2e
    RL4RS_LAUNCH_CHECK();
    if (stats_dev) {
        hipLaunchKernelGGL(k_reduce_terms, dim3(1), dim3(256), 0, st, p->terms, N, stats_dev);
        RL4RS_LAUNCH_CHECK();
    }
    return RL4RS_OK;
}

int rl4rs_policy_adam_step(rl4rs_policy* p, const float* grad_dev, float lr, float beta1, float beta2, float eps,
                           float grad_clip, void* stream) {
    RL4RS_REQUIRE(p && grad_dev, "policy_adam_step: null argument");
    hipStream_t st = (hipStream_t)stream;
    p->adam_t += 1;
    const double t = (double)p->adam_t;
    const float lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    if (grad_clip > 0.f) hipLaunchKernelGGL(k_sumsq, dim3(1), dim3(256), 0, st, grad_dev, p->n_params, p->sumsq);
    hipLaunchKernelGGL(k_adam, dim3((p->n_params + 255) / 256), dim3(256), 0, st, p->params, grad_dev, p->adam_m, p->adam_v,
                       p->n_params, lr_t, beta1, beta2, eps, p->sumsq, grad_clip);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// ---- raw-state policy encoder (rllib_rawstate_model.py:25-86, mask wrapper rllib_mask_model.py:67-115) ---------------
}  // extern "C"

struct rl4rs_rawpolicy {
    rl4rs_rawpolicy_cfg c;
    PolDims d;
    int F;
    float *cat_emb, *seq_emb, *dense_w1, *dense_b1, *dense_w2, *dense_b2, *ctx_w, *ctx_b, *head_w, *head_b;
    float *feat, *dh, *ctx, *ext;
    std::vector<void*> owned;
};

extern "C" {

int rl4rs_rawpolicy_destroy(rl4rs_rawpolicy* p) {
    if (!p) return RL4RS_OK;
    for (void* q : p->owned) (void)hipFree(q);
    delete p;
    return RL4RS_OK;
}

int rl4rs_rawpolicy_create(const rl4rs_rawpolicy_cfg* c, const rl4rs_rawpolicy_weights* w, void* stream,
                           rl4rs_rawpolicy** out) {
    RL4RS_REQUIRE(c && w && out, "rawpolicy_create: null argument");
    RL4RS_REQUIRE(c->emb_size > 0 && c->hidden_units > 0 && c->maxlen >= 1 && c->seq_num >= 1 && c->seq_num <= 4 &&
                  c->category_feature_num >= 1 && c->category_hash_size > 0 && c->dense_feature_num > 0 &&
                  c->action_size > 1 && c->max_rows > 0, "rawpolicy_create: bad sizes");
    RL4RS_REQUIRE(w->cat_emb && w->seq_emb && w->dense_w1 && w->dense_b1 && w->dense_w2 && w->dense_b2 && w->ctx_w && w->ctx_b &&
                  w->out_w && w->out_b && w->value_w && w->value_b, "rawpolicy_create: weights missing");
    if (rl4rs_device_count() <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    hipStream_t st = (hipStream_t)stream;
    const int E = c->emb_size, U = c->hidden_units, H = c->category_hash_size, S = c->seq_num, Dn = c->dense_feature_num;
    const int A = c->action_size, AE = A + 1, CTX = 256;
    rl4rs_rawpolicy* p = new rl4rs_rawpolicy();
    p->c = *c;
    p->d.OD = CTX; p->d.HID = 0; p->d.A = A; p->d.AE = AE; p->d.W = (A + 31) / 32;
    p->F = S * E + U + E;                 // [sequence_feature | dense_feature | category_feature] (rllib_rawstate_model.py:52)
    int rc;
    std::vector<std::vector<float>> keep;
    keep.reserve(16);
    auto up = [&](float** dst, const float* src, size_t n) {
        int r = dev_alloc(dst, n);
        if (r) return r;
        p->owned.push_back(*dst);
        hipError_t e = hipMemcpyAsync(*dst, src, n * 4, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("hipMemcpyAsync failed: %s", hipGetErrorString(e)); return (int)RL4RS_EHIP; }
        return (int)RL4RS_OK;
    };
    auto al = [&](float** dst, size_t n) {
        int r = dev_alloc(dst, n);
        if (r == RL4RS_OK) p->owned.push_back(*dst);
        return r;
    };
#define RP_FAIL(expr) do { if ((rc = (expr)) != RL4RS_OK) { rl4rs_rawpolicy_destroy(p); return rc; } } while (0)
#define RP_PK(dst, src, kk, nn) do { keep.push_back(pack_gemm_weight((src), (nn), (kk), (nn))); \
        RP_FAIL(up(&p->dst, keep.back().data(), keep.back().size())); } while (0)
    RP_FAIL(up(&p->cat_emb, w->cat_emb, (size_t)H * E));
    RP_FAIL(up(&p->seq_emb, w->seq_emb, (size_t)H * E));
    RP_PK(dense_w1, w->dense_w1, Dn, U);
    RP_FAIL(up(&p->dense_b1, w->dense_b1, U));
    RP_PK(dense_w2, w->dense_w2, U, U);
    RP_FAIL(up(&p->dense_b2, w->dense_b2, U));
    RP_PK(ctx_w, w->ctx_w, p->F, CTX);
    RP_FAIL(up(&p->ctx_b, w->ctx_b, CTX));
    {   // one head GEMM for [logits | value]
        std::vector<float> hw((size_t)CTX * AE), hb(AE);
        for (int k = 0; k < CTX; ++k) {
            for (int a = 0; a < A; ++a) hw[(size_t)k * AE + a] = w->out_w[(size_t)k * A + a];
            hw[(size_t)k * AE + A] = w->value_w[k];
        }
        for (int a = 0; a < A; ++a) hb[a] = w->out_b[a];
        hb[A] = w->value_b[0];
        keep.push_back(pack_gemm_weight(hw.data(), AE, CTX, AE));
        RP_FAIL(up(&p->head_w, keep.back().data(), keep.back().size()));
        keep.push_back(std::move(hb));
        RP_FAIL(up(&p->head_b, keep.back().data(), keep.back().size()));
    }
    RP_FAIL(al(&p->feat, (size_t)c->max_rows * p->F));
    RP_FAIL(al(&p->dh, (size_t)c->max_rows * U));
    RP_FAIL(al(&p->ctx, (size_t)c->max_rows * CTX));
    RP_FAIL(al(&p->ext, (size_t)c->max_rows * AE));
#undef RP_PK
#undef RP_FAIL
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) {
        set_error("rawpolicy_create: hipStreamSynchronize failed: %s", hipGetErrorString(e));
        rl4rs_rawpolicy_destroy(p);
        return RL4RS_EHIP;
    }
    *out = p;
    return RL4RS_OK;
}

static int rawpolicy_forward(rl4rs_rawpolicy* p, int N, const int32_t* cat, const float* dense, const int32_t* const* seq,
                             hipStream_t st) {
    const int E = p->c.emb_size, U = p->c.hidden_units, H = p->c.category_hash_size, S = p->c.seq_num, Dn = p->c.dense_feature_num;
    const int F = p->F, L = p->c.maxlen, Cn = p->c.category_feature_num;
    const dim3 g4((N + 3) / 4), b256(256);
    int rc;
    for (int s = 0; s < S; ++s)          // sequence_input_concat (utils.py:57-77): mean over all maxlen positions
        hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, seq[s], N, L, H, E, p->seq_emb, p->feat, (int64_t)F, s * E);
    hipLaunchKernelGGL(k_emb_mean, g4, b256, 0, st, cat, N, Cn, H, E, p->cat_emb, p->feat, (int64_t)F, S * E + U);
    RL4RS_LAUNCH_CHECK();
    if ((rc = launch_gemm_packed(dense, Dn, p->dense_w1, p->dense_b1, p->dh, U, N, U, Dn, 1, st))) return rc;
    if ((rc = launch_gemm_packed(p->dh, U, p->dense_w2, p->dense_b2, p->feat + S * E, F, N, U, U, 1, st))) return rc;
    if ((rc = launch_gemm_packed(p->feat, F, p->ctx_w, p->ctx_b, p->ctx, 256, N, 256, F, 1, st))) return rc;
    return launch_gemm_packed(p->ctx, 256, p->head_w, p->head_b, p->ext, p->d.AE, N, p->d.AE, 256, 0, st);
}

int rl4rs_rawpolicy_act(rl4rs_rawpolicy* p, int32_t N, const int32_t* cat, const float* dense, const int32_t* const* seq,
                        const uint32_t* mask_bits, uint32_t seed, uint32_t step, int32_t* actions, float* logp, float* value,
                        float* entropy, float* logits, void* stream) {
    RL4RS_REQUIRE(p && cat && dense && seq && actions && N > 0 && N <= p->c.max_rows,
                  "rawpolicy_act: bad argument (N=%d, max_rows=%d)", N, p ? p->c.max_rows : -1);
    for (int s = 0; s < p->c.seq_num; ++s) RL4RS_REQUIRE(seq[s], "rawpolicy_act: sequence input %d is NULL", s);
    hipStream_t st = (hipStream_t)stream;
    int rc = rawpolicy_forward(p, N, cat, dense, seq, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rawpolicy_head<true>, dim3((N + 3) / 4), dim3(256), (size_t)4 * p->d.AE * 4, st, p->d, N, p->ext,
                       (int64_t)p->d.AE, mask_bits, seed, step, actions, logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_rawpolicy_evaluate(rl4rs_rawpolicy* p, int32_t N, const int32_t* cat, const float* dense, const int32_t* const* seq,
                             const uint32_t* mask_bits, const int32_t* actions, float* logp, float* value, float* entropy,
                             float* logits, void* stream) {
    RL4RS_REQUIRE(p && cat && dense && seq && actions && N > 0 && N <= p->c.max_rows,
                  "rawpolicy_evaluate: bad argument (N=%d, max_rows=%d)", N, p ? p->c.max_rows : -1);
    for (int s = 0; s < p->c.seq_num; ++s) RL4RS_REQUIRE(seq[s], "rawpolicy_evaluate: sequence input %d is NULL", s);
    hipStream_t st = (hipStream_t)stream;
    int rc = rawpolicy_forward(p, N, cat, dense, seq, st);
    if (rc) return rc;
    hipLaunchKernelGGL(k_rawpolicy_head<false>, dim3((N + 3) / 4), dim3(256), (size_t)4 * p->d.AE * 4, st, p->d, N, p->ext,
                       (int64_t)p->d.AE, mask_bits, 0u, 0u, const_cast<int32_t*>(actions), logp, value, entropy, logits);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

// One PPO SGD pass (RLlib: num_sgd_iter passes over minibatches of sgd_minibatch_size, script/modelfree_train.py:231-254)
// over N already shuffled samples: loss + backward + Adam per minibatch of `minibatch` consecutive rows, the trailing
// N % minibatch rows are dropped.  Same arithmetic as calling rl4rs_policy_loss_grad + rl4rs_policy_adam_step per
// minibatch; exists so that a single-GPU trainer pays one host call per pass instead of two per minibatch.
}  // extern "C"

namespace {

struct PpoCall {
    int32_t N, minibatch;
    const float* obs; const uint32_t* mask_bits; const int32_t* actions; const float* adv; const float* ret;
    const float* old_logp; const float* old_value; const float* old_logits;
    float vf_coeff, ent_coeff, clip, vf_clip, kl_coeff, lr, beta1, beta2, eps;
};

size_t pass_smem_bytes(const PolDims& d) { return (size_t)32 * ((d.OD | 1) + (d.HID | 1) + 2 * (d.AE | 1) + d.A + 5 + d.W) * 4; }

// The compile-time instantiation of the pass (ppo_pass.hpp: STD) covers the shape every BASELINE.json configuration runs
bool pass_is_std(const rl4rs_policy* p, int minibatch);

int pass_rows_per_wg(const rl4rs_policy* p) {
    // rows per workgroup: the per-row loss code is the longest stretch of phase A, so it is spread over as many compute units
    // as the minibatch allows (8 rows = one row per wave; RL4RS_POLICY_OPT_PPO_ROWS = 16 / 32 for A/B runs of the all-runtime form;
    // 4 = the compile-time form's 4-row workgroups - one row per SIMD - where that form applies, 8 elsewhere)
    return (p->opt_ppo_rows == 32 || p->opt_ppo_rows == 16) ? p->opt_ppo_rows : 8;
}
// rows per workgroup of the compile-time instantiation (pass_is_std): 4 while MB / 4 workgroups still meet at the arrival-word
// barrier (<= 126: MB = 256 is 19.4 us per minibatch against 20.6 with 8 rows; MB = 512 would be 128 workgroups on the counter
// barrier: 23.7 against 22.3), 8 beyond; RL4RS_POLICY_OPT_PPO_ROWS = 4 / 8 pins it
int pass_std_rows(const rl4rs_policy* p, int minibatch) {
    if (p->opt_ppo_rows == 4) return 4;
    if (p->opt_ppo_rows == 8) return 8;
    return minibatch / 4 <= 126 ? 4 : 8;
}

// k_ppo_pass's dynamic-LDS opt-in is a property of the FUNCTION, not of a handle: raised once per process to the most any
// policy shape may ask for (160 KB minus the kernel's few bytes of static LDS), never lowered - two handles of different
// shapes cannot undercut each other.  Residency is still computed per handle with that handle's real LDS size.
bool pass_is_std(const rl4rs_policy* p, int minibatch) {
    const PolDims& d = p->d;
    return p->opt_ppo_std && d.OD == 256 && d.HID == 64 && d.A == 284 && d.AE == 285 && d.W == 9 && pass_rows_per_wg(p) == 8 && minibatch % 256 == 0;
}

// workgroups of one pass launch: the MB / rows that own samples; the compile-time instantiation adds workgroups that only take phase B
// tasks until every one of its 45 tasks has a workgroup of its own (ppo_pass.hpp)
int pass_grid(const rl4rs_policy* p, int minibatch) {
    if (pass_is_std(p, minibatch)) {
        const int n_a = minibatch / pass_std_rows(p, minibatch);
        return n_a > 45 ? n_a : 45;
    }
    return minibatch / pass_rows_per_wg(p);
}

constexpr size_t PASS_SMEM_MAX = (size_t)160 * 1024 - 64;
bool pass_opt_in() {
    // once per DEVICE (the attribute is per device and function), guarded: see raise_dyn_smem
    static std::mutex mu;
    static std::vector<std::pair<int, bool>> done;
    std::lock_guard<std::mutex> lock(mu);
    const int dev = current_device();
    for (auto& e : done)
        if (e.first == dev) return e.second;
    bool ok = true;
    for (const void* fn : {reinterpret_cast<const void*>(&k_ppo_pass<false, 8>), reinterpret_cast<const void*>(&k_ppo_pass<true, 8>),
                           reinterpret_cast<const void*>(&k_ppo_pass<true, 4>)})
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PASS_SMEM_MAX) != hipSuccess) {
            (void)hipGetLastError();          // no sticky error for the next launch check: the caller takes the per-minibatch kernels
            ok = false;
        }
    done.emplace_back(dev, ok);
    return ok;
}

// Does the persistent pass fit this policy / minibatch, and can its whole grid be resident at once?  The software grid barrier
// needs every workgroup on a compute unit at the same time: checked against the occupancy the runtime reports for this kernel
// and launch shape x the device's CU count (a partitioned or otherwise smaller device falls back to the per-minibatch kernels).
bool pass_fits(rl4rs_policy* p, int minibatch, float grad_clip) {
    const PolDims& d = p->d;
    const int NT1 = d.HID / 32;
    const size_t smem = pass_smem_bytes(d);
    const int rows = pass_rows_per_wg(p);
    const bool shape_ok = p->opt_ppo_fused && grad_clip <= 0.f && d.HID % 32 == 0 && (NT1 == 1 || NT1 == 2 || NT1 == 4 || NT1 == 8) &&
                          d.OD % 32 == 0 && (d.OD / (8 / NT1)) % 64 == 0 && d.HID % 64 == 0 && minibatch % 128 == 0 && minibatch / rows <= 128 &&
                          (size_t)32 * (d.OD | 1) >= 8192 && (size_t)(8 / NT1) * NT1 * 1024 <= (size_t)32 * (d.AE | 1) && smem <= PASS_SMEM_MAX;
    if (!shape_ok || !pass_opt_in()) return false;
    if (p->pass_resident_wgs < 0) {
        int dev = 0, cus = 0, per_cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
            (void)hipGetLastError();
            cus = 0;
        }
        // blocks per CU: the runtime's occupancy answer, capped by what the LDS alone admits (160 KB per CU); if the query itself
        // is refused (it is for some > 64 KB dynamic-LDS shapes) the LDS bound with one 512-thread block per CU minimum stands in
        const int by_lds = (int)((size_t)160 * 1024 / (smem ? smem : 1));
        // (both instantiations: 512 threads at <= 256 registers and the same LDS - one answer serves either)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&k_ppo_pass<false, 8>), 512, smem) != hipSuccess || per_cu <= 0) {
            (void)hipGetLastError();          // do not leave a sticky error for the next launch check
            per_cu = by_lds > 0 ? 1 : 0;
        }
        if (per_cu > by_lds) per_cu = by_lds;
        p->pass_resident_wgs = per_cu * cus;
        if (p->opt_resident_cap >= 0) p->pass_resident_wgs = p->opt_resident_cap;      // tests: pretend a smaller device
    }
    return pass_grid(p, minibatch) <= p->pass_resident_wgs;
}

// minibatches [mb_begin, mb_end) through k_ppo_pass; apply = 0 leaves the parameters alone (gradient of ONE minibatch only)
int launch_ppo_pass(rl4rs_policy* p, const PpoCall& c, int mb_begin, int mb_end, int apply, float* grad_dev, hipStream_t st) {
    const PolDims& d = p->d;
    PassArgs a;
    memset(&a, 0, sizeof(a));
    a.d = d; a.N = c.N; a.MB = c.minibatch;
    a.rows = pass_rows_per_wg(p);
    a.prm = p->params; a.am = p->adam_m; a.av = p->adam_v; a.w2t = p->w2t;
    a.obs = c.obs; a.mask = c.mask_bits;
    a.L.algo = 1; a.L.vf_coeff = c.vf_coeff; a.L.ent_coeff = c.ent_coeff; a.L.clip = c.clip; a.L.vf_clip = c.vf_clip; a.L.kl_coeff = c.kl_coeff;
    a.L.scale = 1.0f / (float)c.minibatch;
    a.L.actions = c.actions; a.L.adv = c.adv; a.L.ret = c.ret; a.L.old_logp = c.old_logp; a.L.old_value = c.old_value; a.L.old_logits = c.old_logits;
    a.H = p->H; a.dOut = p->dOut; a.dHpre = p->dHpre; a.terms = p->terms; a.grad = grad_dev; a.bar = p->bar; a.dead_host = p->dead_host;
    a.lr = c.lr; a.b1 = c.beta1; a.b2 = c.beta2; a.eps = c.eps; a.t0 = p->adam_t;
    a.mb_begin = mb_begin; a.mb_end = mb_end; a.apply = apply;
    a.trace = nullptr;
#ifdef RL4RS_PASS_TRACE
    static unsigned long long* trace_buf = nullptr;
    if (!trace_buf) (void)hipMalloc((void**)&trace_buf, 16 * 8);
    a.trace = trace_buf;
#endif
    // bar[0] = arrival counter (reset per launch), bar[1] = sticky "a grid barrier timed out" flag (rl4rs_policy_status).
    // A pass that finds bar[1] set leaves at its first grid barrier without updating anything, so launching another one would
    // hand back stale gradients / statistics as if it had run: the pinned mirror of the flag is looked at first (a plain host
    // load, no synchronisation) and the call fails until rl4rs_policy_status has reported and cleared the condition.
    if (p->dead_host && *reinterpret_cast<volatile unsigned*>(p->dead_host) != 0u) {
        set_error("policy: an earlier persistent PPO pass timed out at a grid barrier (its workgroups were not co-resident); no update was "
                  "made - call rl4rs_policy_status to acknowledge, then use RL4RS_POLICY_OPT_PPO_FUSED = 0 (per-minibatch kernels)");
        return RL4RS_ESTATE;
    }
    RL4RS_HIP_TRY(hipMemsetAsync(p->bar, 0, 4, st));
    RL4RS_HIP_TRY(hipMemsetAsync(p->bar + 64, 0, 256 * 4, st));         // (the per-workgroup arrival words count from 1 in every launch)
    if (pass_is_std(p, c.minibatch)) {
        if (pass_std_rows(p, c.minibatch) == 4) hipLaunchKernelGGL((k_ppo_pass<true, 4>), dim3(pass_grid(p, c.minibatch)), dim3(512), pass_smem_bytes(d), st, a);
        else hipLaunchKernelGGL((k_ppo_pass<true, 8>), dim3(pass_grid(p, c.minibatch)), dim3(512), pass_smem_bytes(d), st, a);
    } else {
        hipLaunchKernelGGL((k_ppo_pass<false, 8>), dim3(pass_grid(p, c.minibatch)), dim3(512), pass_smem_bytes(d), st, a);
    }
    RL4RS_LAUNCH_CHECK();
    p->pass_launched = true;
#ifdef RL4RS_PASS_TRACE
    {
        unsigned long long h[16];
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, a.trace, sizeof(h), hipMemcpyDeviceToHost);
        static const char* nm[] = {"stage", "layer1", "layer2", "loss", "dH", "barrier1", "phaseB", "barrier2"};
        for (int k = 0; k < 8; ++k) fprintf(stderr, "  %-14s %8.2f us\n", nm[k], (double)(h[k + 1] - h[k]) / 2400.0);   // core clocks at ~2.4 GHz
        fprintf(stderr, "  dH: mfma %.2f  partial+sync %.2f  combine %.2f\n", (double)(h[12] - h[4]) / 2400.0, (double)(h[13] - h[12]) / 2400.0,
                (double)(h[5] - h[13]) / 2400.0);
        fprintf(stderr, "  lr_t %.2f  tile-loop %.2f  adam %.2f | first loss row %.2f\n", (double)(h[10] - h[6]) / 2400.0,
                (double)(h[9] - h[10]) / 2400.0, (double)(h[7] - h[9]) / 2400.0, (double)(h[11] - h[3]) / 2400.0);
        fprintf(stderr, "  row loss: join %.2f  lse %.2f  loss %.2f\n", (double)(h[14] - h[3]) / 2400.0, (double)(h[15] - h[14]) / 2400.0, (double)(h[11] - h[15]) / 2400.0);
    }
#endif
    return RL4RS_OK;
}

}  // namespace

extern "C" {

// stats_dev (optional): 8 floats - [0..3] sums of {pi_loss, vf_loss, entropy, kl} over the LAST minibatch, [4..7] the same sums
// over every sample of the pass (RLlib's PPO reports the KL averaged over the minibatches of the pass and feeds it to the
// adaptive kl_coeff rule, script/modelfree_train.py:189,216).
int rl4rs_policy_ppo_epoch(rl4rs_policy* p, int32_t N, int32_t minibatch, const float* obs, const uint32_t* mask_bits,
                           const int32_t* actions, const float* adv, const float* ret, const float* old_logp,
                           const float* old_value, const float* old_logits, float vf_coeff, float ent_coeff, float clip,
                           float vf_clip, float kl_coeff, float lr, float beta1, float beta2, float eps, float grad_clip,
                           float* grad_dev, float* stats_dev, void* stream) {
    RL4RS_REQUIRE(p && obs && actions && adv && ret && old_logp && old_value && old_logits && grad_dev,
                  "policy_ppo_epoch: null argument");
    RL4RS_REQUIRE(minibatch > 0 && N >= minibatch && minibatch <= p->max_rows && N <= p->max_rows,
                  "policy_ppo_epoch: bad sizes (N=%d, minibatch=%d, max_rows=%d)", N, minibatch, p->max_rows);
    const PolDims& d = p->d;
    hipStream_t st = (hipStream_t)stream;
    const int nmb = N / minibatch;
    // fused persistent pass (k_ppo_pass) when the shapes fit its tiling, its grid can be co-resident, and no global-norm clip is asked for
    if (pass_fits(p, minibatch, grad_clip)) {
        PpoCall c = {N, minibatch, obs, mask_bits, actions, adv, ret, old_logp, old_value, old_logits,
                     vf_coeff, ent_coeff, clip, vf_clip, kl_coeff, lr, beta1, beta2, eps};
        int rc = launch_ppo_pass(p, c, 0, nmb, 1, grad_dev, st);
        if (rc) return rc;
        p->adam_t += nmb;
        if (stats_dev) {
            hipLaunchKernelGGL(k_reduce_terms, dim3(1), dim3(256), 0, st, p->terms + (size_t)(nmb - 1) * minibatch, minibatch, stats_dev);
            hipLaunchKernelGGL(k_reduce_terms, dim3(1), dim3(256), 0, st, p->terms, nmb * minibatch, stats_dev + 4);
            RL4RS_LAUNCH_CHECK();
        }
        return RL4RS_OK;
    }
    if (stats_dev) RL4RS_HIP_TRY(hipMemsetAsync(stats_dev + 4, 0, 16, st));
    for (int lo = 0; lo + minibatch <= N; lo += minibatch) {
        int rc = rl4rs_policy_loss_grad(p, 1, minibatch, obs + (size_t)lo * d.OD, mask_bits ? mask_bits + (size_t)lo * d.W : nullptr,
                                        actions + lo, adv + lo, ret + lo, old_logp + lo, old_value + lo,
                                        old_logits + (size_t)lo * d.A, vf_coeff, ent_coeff, clip, vf_clip, kl_coeff, grad_dev,
                                        stats_dev, stream);
        if (rc) return rc;
        if (stats_dev) {
            hipLaunchKernelGGL(k_axpy4, dim3(1), dim3(64), 0, st, stats_dev, stats_dev + 4);
            RL4RS_LAUNCH_CHECK();
        }
        if ((rc = rl4rs_policy_adam_step(p, grad_dev, lr, beta1, beta2, eps, grad_clip, stream))) return rc;
    }
    return RL4RS_OK;
}

// Gradient of ONE minibatch (rows [mb_index * minibatch, (mb_index + 1) * minibatch) of the shuffled pass) WITHOUT touching the
// parameters: the data-parallel form of the pass.  Each rank calls this, mean-all-reduces grad_dev (the one collective of the
// training loop) and then rl4rs_policy_adam_step - same arithmetic as the single-GPU pass on the averaged
// gradient.  Runs phase A + the gradient tiles of k_ppo_pass as one launch when the pass fits (else the per-minibatch kernels).
// stats_dev (optional): 4 floats, sums of {pi_loss, vf_loss, entropy, kl} over the minibatch.
int rl4rs_policy_ppo_minibatch_grad(rl4rs_policy* p, int32_t N, int32_t minibatch, int32_t mb_index, const float* obs,
                                    const uint32_t* mask_bits, const int32_t* actions, const float* adv, const float* ret,
                                    const float* old_logp, const float* old_value, const float* old_logits, float vf_coeff,
                                    float ent_coeff, float clip, float vf_clip, float kl_coeff, float* grad_dev,
                                    float* stats_dev, void* stream) {
    RL4RS_REQUIRE(p && obs && actions && adv && ret && old_logp && old_value && old_logits && grad_dev,
                  "policy_ppo_minibatch_grad: null argument");
    RL4RS_REQUIRE(minibatch > 0 && N >= minibatch && N <= p->max_rows && mb_index >= 0 && (mb_index + 1) * (int64_t)minibatch <= N,
                  "policy_ppo_minibatch_grad: bad sizes (N=%d, minibatch=%d, mb_index=%d, max_rows=%d)", N, minibatch, mb_index, p->max_rows);
    const PolDims& d = p->d;
    hipStream_t st = (hipStream_t)stream;
    if (pass_fits(p, minibatch, 0.f)) {
        PpoCall c = {N, minibatch, obs, mask_bits, actions, adv, ret, old_logp, old_value, old_logits,
                     vf_coeff, ent_coeff, clip, vf_clip, kl_coeff, 0.f, 0.9f, 0.999f, 1e-8f};
        int rc = launch_ppo_pass(p, c, mb_index, mb_index + 1, 0, grad_dev, st);
        if (rc) return rc;
        if (stats_dev) {
            hipLaunchKernelGGL(k_reduce_terms, dim3(1), dim3(256), 0, st, p->terms + (size_t)mb_index * minibatch, minibatch, stats_dev);
            RL4RS_LAUNCH_CHECK();
        }
        return RL4RS_OK;
    }
    const size_t lo = (size_t)mb_index * minibatch;
    return rl4rs_policy_loss_grad(p, 1, minibatch, obs + lo * d.OD, mask_bits ? mask_bits + lo * d.W : nullptr, actions + lo, adv + lo,
                                  ret + lo, old_logp + lo, old_value + lo, old_logits + lo * d.A, vf_coeff, ent_coeff, clip, vf_clip,
                                  kl_coeff, grad_dev, stats_dev, stream);
}

// Synchronises `stream` and reports (and clears) the handle's sticky status: RL4RS_POLICY_STATUS_PASS_TIMEOUT when a grid
// barrier of a persistent PPO pass timed out (its workgroups were not co-resident).  The pass stops updating at the barrier
// that failed, so the parameters are those of the last completed minibatch - but the pass is incomplete: callers treat it as
// an error (rl4rs_amd.train.Trainer checks after every iteration and in params() / close()).
int rl4rs_policy_status_words(rl4rs_policy* p, uint32_t** words_dev) {
    RL4RS_REQUIRE(p && words_dev, "policy_status_words: null argument");
    *words_dev = p->bar;
    return RL4RS_OK;
}

int rl4rs_policy_status(rl4rs_policy* p, int32_t* flags, void* stream) {
    RL4RS_REQUIRE(p && flags, "policy_status: null argument");
    hipStream_t st = (hipStream_t)stream;
    unsigned v[2] = {0u, 0u};
    *flags = 0;
    if (!p->pass_launched) return RL4RS_OK;
    RL4RS_HIP_TRY(hipMemcpyAsync(v, p->bar, 8, hipMemcpyDeviceToHost, st));
    RL4RS_HIP_TRY(hipStreamSynchronize(st));
    if (v[1]) {
        RL4RS_HIP_TRY(hipMemsetAsync(p->bar, 0, 8, st));
        if (p->dead_host) *reinterpret_cast<volatile unsigned*>(p->dead_host) = 0u;
        *flags = RL4RS_POLICY_STATUS_PASS_TIMEOUT;
    }
    return RL4RS_OK;
}

// Kernel-path selection of one handle (RL4RS_POLICY_OPT_*): what used to be process-wide environment switches.
int rl4rs_policy_set_option(rl4rs_policy* p, int32_t which, int32_t value) {
    RL4RS_REQUIRE(p, "policy_set_option: null handle");
    switch (which) {
        case RL4RS_POLICY_OPT_TILE: p->opt_tile = value != 0; break;
        case RL4RS_POLICY_OPT_PPO_FUSED: p->opt_ppo_fused = value != 0; break;
        case RL4RS_POLICY_OPT_PPO_ROWS:
            RL4RS_REQUIRE(value == 4 || value == 8 || value == 16 || value == 32, "policy_set_option: PPO_ROWS must be 4, 8, 16 or 32 (got %d)", value);
            p->opt_ppo_rows = value; p->pass_resident_wgs = -1; break;
        case RL4RS_POLICY_OPT_RESIDENT_WGS: p->opt_resident_cap = value; p->pass_resident_wgs = -1; break;
        case RL4RS_POLICY_OPT_PPO_STD: p->opt_ppo_std = value != 0; break;
        default: set_error("policy_set_option: unknown option %d", which); return RL4RS_EINVAL;
    }
    return RL4RS_OK;
}

}  // extern "C"

#include "simtrain.hpp"
#include "dientrain.hpp"
#include "rawtrain.hpp"
#include "qlearn.hpp"
#include "contirl.hpp"
