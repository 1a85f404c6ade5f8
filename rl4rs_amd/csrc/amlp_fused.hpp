// Fused minibatch kernels of the amlp network (contirl.hpp; included by it, compiled into policy.hip).
//
// A BCQ / CQL update differentiates 256-row minibatches through eight small networks; in per-layer launches that was 3 launches
// per forward and 6 - 7 per backward, every one of them at its 4 - 9 us latency floor (~95 launches, 0.62 ms per BCQ
// update with the GPU never idle).  Here a minibatch-sized call is
//     forward    ONE launch  k_amlp_fwd4   : [x | a] -> relu -> relu -> head, 4 rows per workgroup, every layer on
//                                            v_mfma_f32_4x4x1 (mfma4.hpp: a lane owns one hidden column of its wave's 64), the
//                                            activations of a row never leave LDS between the layers (h1, h2 are also written out
//                                            for the backward), weights streamed ROW-MAJOR as they are (no packing: they change
//                                            every update) through the register ring, which runs across the layers;
//     backward   TWO launches:   k_amlp_bwd4 (dout -> d_h2 -> d_h1 -> d action, ReLU masks in the epilogues, same machine mapping;
//                                            it reads W2^T, W3^T, W1_action^T - the weights with the OUTPUT index contiguous - which
//                                            extra workgroups of the FORWARD launch rebuild every time: a backward always follows a
//                                            forward of the same rows under the same parameters, so no dirty flag is needed)
//                                k_gemm_tn4_group (every weight and bias gradient: 4 sample-axis reductions per network, one
//                                            grid over all their 32 x 32 tiles)
//     networks that see the same rows (the twin critics) share each of these launches (grid.y = network)
// and the optimiser is one launch per phase for all its networks (k_adam_multi: torch Adam + the soft target updates).
// Exact fp32; the summation order of a dot product differs from the per-layer GEMMs (4 interleaved chains over k instead of
// MFMA-internal pairs): same tests, same bars.  Eligible: hidden 256 x 256, out_dim <= 64, act_dim <= 64, rep = 1, N <= 2048.
#pragma once
#include "mfma4.hpp"

namespace rl4rs {

struct AmlpFwd4 {
    const float* obs; const float* act;       // [N, D], [N, E] (E may be 0)
    const float* W1; const float* b1; const float* W2; const float* b2; const float* W3; const float* b3;
    float* h1; float* h2; float* out;         // [N, 256], [N, 256], [N, K3]
    float* w3t; float* w2t; float* w1at;      // trainable networks: the transposed weights the backward chain reads are rebuilt by
                                              // extra workgroups of THIS launch (a backward always follows a forward of the same rows
                                              // under the same parameters); NULL: none
    int N, D, E, K3, head_act;
};
// up to 4 networks over the same rows in one launch (grid.y = network): the twin critics
struct AmlpFwd4x { AmlpFwd4 n[4]; };

namespace a4 {
constexpr int H = 256, LDH = H + 4;
__device__ __forceinline__ float head_activation(float x, int act) {
    if (act == ACT_TANH) return tanhf(x);
    if (act == ACT_RELU) return fmaxf(x, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + expf(-x));
    if (act == ACT_ELU) return x > 0.f ? x : expm1f(x);
    return x;
}
// 4 consecutive rows k .. k + 3 of a row-major [*, ld] weight matrix at this lane's column: the B operands of 4 MFMAs
// (rows past the matrix come back as zero: the descriptor's num_records ends with the last row)
__device__ __forceinline__ float4 ldrows(__amdgpu_buffer_rsrc_t rs, int voff, int k, int ld4) {
    return make_float4(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, k * ld4, 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (k + 1) * ld4, 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (k + 2) * ld4, 0)),
                       __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, (k + 3) * ld4, 0)));
}
}  // namespace a4

// W3 [256, K3] -> w3t [K3, 256];  W2 [256, 256] -> w2t;  W1a = W1 rows D.. [E, 256] -> w1at [256, E]; element i of the three
__device__ __forceinline__ void amlp_transpose_element(int i, const float* __restrict__ W3, int K3, float* __restrict__ w3t,
                                                       const float* __restrict__ W2, float* __restrict__ w2t, const float* __restrict__ W1a,
                                                       int E, float* __restrict__ w1at) {
    constexpr int H = 256;
    if (i < H * H) { const int r = i >> 8, c = i & 255; w2t[i] = W2[c * H + r]; return; }
    i -= H * H;
    if (i < K3 * H) { const int r = i >> 8, c = i & 255; w3t[i] = W3[c * K3 + r]; return; }
    i -= K3 * H;
    if (i < H * E) { const int r = i / E, c = i - r * E; w1at[i] = W1a[c * H + r]; }
}
constexpr int AMLP_T_PER_WG = 2048;           // transposed elements per extra workgroup of the forward launch

// grid = (ceil(N / (4 MTW)) [+ transposing workgroups], networks) workgroups of 4 waves; dynamic LDS = amlp_fwd4_smem(D + E, MTW).
// MTW = 1 (4 rows per workgroup: 64 workgroups for a 256-row minibatch) is what runs; MTW = 2 (every weight value feeds two row
// tiles, half the workgroups) was measured slower there (BCQ update 0.371 -> 0.406 ms) and is built for A/B runs only.
template <int MTW>
__global__ __launch_bounds__(256) void k_amlp_fwd4(AmlpFwd4x x) {
    using namespace r8;
    using namespace a4;
    constexpr int RWS = 4 * MTW;
    const AmlpFwd4& a = x.n[blockIdx.y];
    const int row_wgs = (a.N + RWS - 1) / RWS;
    if ((int)blockIdx.x >= row_wgs) {
        if (a.w2t) {
            const int base = ((int)blockIdx.x - row_wgs) * AMLP_T_PER_WG;
            for (int i = threadIdx.x; i < AMLP_T_PER_WG; i += 256)
                amlp_transpose_element(base + i, a.W3, a.K3, a.w3t, a.W2, a.w2t, a.W1 + (size_t)a.D * 256, a.E, a.w1at);
        }
        return;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int KX = a.D + a.E, KXP = (KX + 63) / 64 * 64, LDX = KXP + 4;
    float* xs = reinterpret_cast<float*>(smem);          // [RWS][LDX]  [x | a | 0]
    float* h1s = xs + RWS * LDX;                         // [RWS][LDH]
    float* h2s = h1s + RWS * LDH;                        // [RWS][LDH]
    float* part = h2s + RWS * LDH;                       // [4 waves][RWS rows][64]  head partials
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RWS;
    const int col = wave * 64 + lane;
    for (int i = tid; i < RWS * KXP; i += 256) {
        const int r = i / KXP, k = i - r * KXP;
        const int gr = min(row0 + r, a.N - 1);
        float v = 0.f;
        if (k < a.D) v = a.obs[(size_t)gr * a.D + k];
        else if (k < KX) v = a.act[(size_t)gr * a.E + (k - a.D)];
        xs[r * LDX + k] = v;
    }
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W1), 0, KX * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W2), 0, H * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.W3), 0, H * a.K3 * 4, 0x00020000);
    const int vcol = col * 4;
    const int k3 = a.K3;
    const int v3 = min(lane, k3 - 1) * 4;                // head: lane = output column (lanes >= K3 compute a duplicate that is never stored)
    float4 ring[RS];
    auto ld1 = [&](int q, int) { return ldrows(rs1, vcol, 4 * q, H * 4); };
    auto ld2 = [&](int q, int) { return ldrows(rs2, vcol, 4 * q, H * 4); };
    auto ld3 = [&](int q, int) { return ldrows(rs3, v3, wave * 64 + 4 * q, k3 * 4); };       // this wave's quarter of k
    auto head2 = [&](int i) { return ld2(i, 0); };
    auto head3 = [&](int i) { return ld3(i, 0); };
    auto none = [&](int) { return make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
    for (int i = 0; i < RS; ++i) ring[i] = ld1(i, 0);
    __syncthreads();
    constexpr int P = 4 / MTW;                           // accumulator chains per tile (>= 4 independent MFMA chains per wave)
    const float* ap[MTW];
    f32x4_t acc[1][MTW][P];
    auto zero = [&]() {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[0][m][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto total = [&](int m) {
        f32x4_t s = acc[0][m][0];
#pragma unroll
        for (int p = 1; p < P; ++p) s += acc[0][m][p];
        return s;
    };
    // ---- layer 1
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = xs + (m * 4 + (lane & 3)) * LDX;
    phase_rt<1, MTW, P>(acc, ap, ring, ld1, head2, KXP / 64);
    {
        const float b = a.b1[col];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const f32x4_t s = total(m);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaxf(s[i] + b, 0.f);
                h1s[(m * 4 + i) * LDH + col] = v;
                if (row0 + m * 4 + i < a.N) a.h1[(size_t)(row0 + m * 4 + i) * H + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- layer 2
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = h1s + (m * 4 + (lane & 3)) * LDH;
    phase_rt<1, MTW, P>(acc, ap, ring, ld2, head3, H / 64);
    {
        const float b = a.b2[col];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const f32x4_t s = total(m);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float v = fmaxf(s[i] + b, 0.f);
                h2s[(m * 4 + i) * LDH + col] = v;
                if (row0 + m * 4 + i < a.N) a.h2[(size_t)(row0 + m * 4 + i) * H + col] = v;
            }
        }
    }
    __syncthreads();
    // ---- head: the four waves split k (64 each), lane = output column; partials meet in LDS and are summed in wave order
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = h2s + (m * 4 + (lane & 3)) * LDH + wave * 64;
    phase_rt<1, MTW, P>(acc, ap, ring, ld3, none, 1);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const f32x4_t s = total(m);
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(wave * RWS + m * 4 + i) * 64 + lane] = s[i];
    }
    __syncthreads();
    for (int e = tid; e < RWS * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        if (j < k3 && row0 + i < a.N) {
            const float v = ((part[(0 * RWS + i) * 64 + j] + part[(1 * RWS + i) * 64 + j]) + part[(2 * RWS + i) * 64 + j]) + part[(3 * RWS + i) * 64 + j];
            a.out[(size_t)(row0 + i) * k3 + j] = head_activation(v + a.b3[j], a.head_act);
        }
    }
}
inline size_t amlp_fwd4_smem(int KX, int MTW) { const int R = 4 * MTW; return (size_t)(R * ((KX + 63) / 64 * 64 + 4) + 2 * R * (256 + 4) + 4 * R * 64) * 4; }

// ---------------------------------------------------------------------------------------------------------------- backward chain
struct AmlpBwd4 {
    const float* dout; const float* h1; const float* h2;      // [N, K3] gradient wrt the head's pre-activation; saved activations
    const float* w3t; const float* w2t; const float* w1at;    // W3^T [K3, 256], W2^T [256, 256], W1[D:, :]^T [256, E]
    float* d_h2; float* d_h1; float* dact;                    // out: [N, 256], [N, 256], [N, E] or NULL
    int N, K3, E;
};
struct AmlpBwd4x { AmlpBwd4 n[4]; };

__global__ void k_amlp_transposes(const float* __restrict__ W3, int K3, float* __restrict__ w3t, const float* __restrict__ W2,
                                  float* __restrict__ w2t, const float* __restrict__ W1a, int E, float* __restrict__ w1at) {
    amlp_transpose_element(blockIdx.x * blockDim.x + threadIdx.x, W3, K3, w3t, W2, w2t, W1a, E, w1at);
}

template <int MTW>
__global__ __launch_bounds__(256) void k_amlp_bwd4(AmlpBwd4x x) {
    using namespace r8;
    using namespace a4;
    constexpr int RWS = 4 * MTW, P = 4 / MTW;
    const AmlpBwd4& a = x.n[blockIdx.y];
    __shared__ __attribute__((aligned(16))) float ds[RWS][64 + 4];      // dout rows, zero-padded to 64
    __shared__ __attribute__((aligned(16))) float d2s[RWS][LDH];
    __shared__ __attribute__((aligned(16))) float d1s[RWS][LDH];
    __shared__ float part[4 * RWS * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RWS;
    const int col = wave * 64 + lane;
    const int k3 = a.K3, E = a.E;
    for (int e = tid; e < RWS * 64; e += 256) {
        const int r = e >> 6, k = e & 63;
        const int gr = min(row0 + r, a.N - 1);
        ds[r][k] = k < k3 ? a.dout[(size_t)gr * k3 + k] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rs3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w3t), 0, k3 * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w2t), 0, H * H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w1at), 0, (E > 0 ? H * E : 1) * 4, 0x00020000);
    const int vcol = col * 4;
    const int ve = min(lane, max(E, 1) - 1) * 4;
    float4 ring[RS];
    auto ld3 = [&](int q, int) { return ldrows(rs3, vcol, 4 * q, H * 4); };
    auto ld2 = [&](int q, int) { return ldrows(rs2, vcol, 4 * q, H * 4); };
    auto ld1 = [&](int q, int) { return ldrows(rs1, ve, wave * 64 + 4 * q, E * 4); };
    auto head2 = [&](int i) { return ld2(i, 0); };
    auto head1 = [&](int i) { return ld1(i, 0); };
    auto none = [&](int) { return make_float4(0.f, 0.f, 0.f, 0.f); };
#pragma unroll
    for (int i = 0; i < RS; ++i) ring[i] = ld3(i, 0);
    // the ReLU masks of this lane's column (requested early, used in the epilogues)
    float m2[MTW][4], m1[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gr = min(row0 + m * 4 + i, a.N - 1);
            m2[m][i] = a.h2[(size_t)gr * H + col];
            m1[m][i] = a.h1[(size_t)gr * H + col];
        }
    __syncthreads();
    const float* ap[MTW];
    f32x4_t acc[1][MTW][P];
    auto zero = [&]() {
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int p = 0; p < P; ++p) acc[0][m][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
    auto total = [&](int m) {
        f32x4_t s = acc[0][m][0];
#pragma unroll
        for (int p = 1; p < P; ++p) s += acc[0][m][p];
        return s;
    };
    const bool want_dact = a.dact != nullptr && E > 0;
    // ---- d_h2 = (dout W3^T) * [h2 > 0]
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = &ds[m * 4 + (lane & 3)][0];
    phase_rt<1, MTW, P>(acc, ap, ring, ld3, head2, 1);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const f32x4_t s = total(m);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = m2[m][i] > 0.f ? s[i] : 0.f;
            d2s[m * 4 + i][col] = v;
            if (row0 + m * 4 + i < a.N) a.d_h2[(size_t)(row0 + m * 4 + i) * H + col] = v;
        }
    }
    __syncthreads();
    // ---- d_h1 = (d_h2 W2^T) * [h1 > 0]
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = &d2s[m * 4 + (lane & 3)][0];
    if (want_dact) phase_rt<1, MTW, P>(acc, ap, ring, ld2, head1, H / 64);
    else phase_rt<1, MTW, P>(acc, ap, ring, ld2, none, H / 64);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const f32x4_t s = total(m);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float v = m1[m][i] > 0.f ? s[i] : 0.f;
            d1s[m * 4 + i][col] = v;
            if (row0 + m * 4 + i < a.N) a.d_h1[(size_t)(row0 + m * 4 + i) * H + col] = v;
        }
    }
    if (!want_dact) return;
    __syncthreads();
    // ---- d action = d_h1 W1_action^T: the four waves split k, lane = action component
    zero();
#pragma unroll
    for (int m = 0; m < MTW; ++m) ap[m] = &d1s[m * 4 + (lane & 3)][wave * 64];
    phase_rt<1, MTW, P>(acc, ap, ring, ld1, none, 1);
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        const f32x4_t s = total(m);
#pragma unroll
        for (int i = 0; i < 4; ++i) part[(wave * RWS + m * 4 + i) * 64 + lane] = s[i];
    }
    __syncthreads();
    for (int e = tid; e < RWS * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        if (j < E && row0 + i < a.N)
            a.dact[(size_t)(row0 + i) * E + j] = ((part[(0 * RWS + i) * 64 + j] + part[(1 * RWS + i) * 64 + j]) + part[(2 * RWS + i) * 64 + j]) + part[(3 * RWS + i) * 64 + j];
    }
}

// --------------------------------------------------------------------------------------------------- optimiser: one launch per phase
// torch.optim.Adam for up to 8 flat parameter tensors + the soft target update targ = (1 - tau) targ + tau p of those that have a
// target network (g = NULL: soft update only).  Element-wise: same values as k_adam / k_soft_update per tensor.
struct AdamMultiDesc { float* p; const float* g; float* m; float* v; float* targ; float lr_t, eps_t; };
struct AdamMulti { AdamMultiDesc d[8]; long long start[9]; int n; float b1, b2, tau; };
__global__ void k_adam_multi(AdamMulti a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.start[a.n]) return;
    int t = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) t += (k < a.n && i >= a.start[k]) ? 1 : 0;
    const AdamMultiDesc& d = a.d[t];
    const long long j = i - a.start[t];
    float pj = d.p[j];
    if (d.g) {
        const float gj = d.g[j];
        const float mj = a.b1 * d.m[j] + (1.f - a.b1) * gj;
        const float vj = a.b2 * d.v[j] + (1.f - a.b2) * gj * gj;
        d.m[j] = mj;
        d.v[j] = vj;
        pj -= d.lr_t * mj / (sqrtf(vj) + d.eps_t);
        d.p[j] = pj;
    }
    if (d.targ) d.targ[j] = (1.f - a.tau) * d.targ[j] + a.tau * pj;
}

// ------------------------------------------------------------------------------------------------ whole-update helpers (rl4rs_bcq_update)
// z parts of the update's noise clamped to +-0.5 in place; the constant -1 / B seed of the actor loss
__global__ void k_bcq_prep(float* __restrict__ z, long long nz, float* __restrict__ minus_inv_b, int B) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nz) z[i] = fminf(fmaxf(z[i], -0.5f), 0.5f);
    if (i < B) minus_inv_b[i] = -1.0f / (float)B;
}
// metrics[0] = mse / E + beta kl / L;  [1] = sum of the critics' losses;  [2] = -mean q  (each only when its inputs are given)
__global__ __launch_bounds__(256) void k_bcq_metrics(const float* __restrict__ loss2, float inv_e, float beta_over_l, const float* __restrict__ closs2,
                                                     const float* __restrict__ qv, int B, float* __restrict__ metrics) {
    __shared__ float sm[256];
    if (threadIdx.x == 0) {
        metrics[0] = loss2[0] * inv_e + loss2[1] * beta_over_l;
        if (closs2) metrics[1] = closs2[0] + closs2[1];
    }
    if (!qv) return;
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += qv[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) metrics[2] = -sm[0] / (float)B;
}

// ---- rl4rs_cql_update: the learned scalars (SAC's log-temperature, CQL's log-alpha) with torch-style Adam on the device.
// st = {p, m, v}; c1 = 1 / (1 - beta1^t), c2 = 1 / (1 - beta2^t) of the step being taken (host-side powers, like the networks' Adam)
__device__ __forceinline__ void scalar_adam(float* __restrict__ p, float* __restrict__ m, float* __restrict__ v, float g, float lr, float c1, float c2) {
    const float mj = 0.9f * m[0] + (1.f - 0.9f) * g;
    const float vj = 0.999f * v[0] + (1.f - 0.999f) * g * g;
    m[0] = mj;
    v[0] = vj;
    p[0] -= lr * c1 * mj / (sqrtf(vj * c2) + 1e-8f);
}
// SACImpl.update_temp: loss = -(exp(log_temp) * mean(logp - A)); its gradient wrt log_temp is the same expression
__global__ __launch_bounds__(256) void k_sac_temp_step(const float* __restrict__ logp, int B, int A, float* p, float* m, float* v, float lr, float c1,
                                                       float c2, float* __restrict__ metric) {
    __shared__ float sm[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += logp[i] - (float)A;
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float g = -(expf(p[0]) * (sm[0] / (float)B));
        metric[0] = g;
        scalar_adam(p, m, v, g, lr, c1, c2);
    }
}
// CQLImpl.update_alpha from the six sums of rl4rs_cql_critic_loss: gap = conservative value - threshold, loss = -(clamp(e^la, 0, 1e6) gap),
// gradient -(e^la gap) while e^la <= 1e6 (the clamp blocks it above).  do_step = 0: no update.  Always: aw = clamp(e^la) * weight AFTER the step.
__global__ void k_cql_alpha_step(const float* __restrict__ sums, int B, float weight, float threshold, float* p, float* m, float* v, float lr,
                                 float c1, float c2, int do_step, float* __restrict__ metric, float* __restrict__ aw) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (do_step) {
        const float gap = weight * ((sums[2] + sums[3]) - (sums[4] + sums[5])) / (2.0f * (float)B) - threshold;
        const float ea = expf(p[0]);
        metric[0] = -(fminf(fmaxf(ea, 0.f), 1e6f) * gap);
        scalar_adam(p, m, v, ea <= 1e6f ? -(ea * gap) : 0.f, lr, c1, c2);
    }
    aw[0] = fminf(fmaxf(expf(p[0]), 0.f), 1e6f) * weight;
}
// rows of the conservative term: acts [B, m, A] column 0 <- the dataset action, columns 1 + 2n .. m - 1 <- the caller's uniform samples on [-1, 1)^A
// and their importance offsets offs [B, m]: 0 for the dataset action, A log 0.5 (the uniform density) for the uniform samples
__global__ void k_cql_fill_rows(float* __restrict__ acts, float* __restrict__ offs, const float* __restrict__ act, const float* __restrict__ uni,
                                int B, int m, int n, int A, float log_uniform) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (1 + n) * A) return;
    const int a = i % A, j = (i / A) % (1 + n), b = i / (A * (1 + n));
    const int col = j == 0 ? 0 : 2 * n + j;
    acts[((size_t)b * m + col) * A + a] = j == 0 ? act[(size_t)b * A + a] : uni[((size_t)b * n + (j - 1)) * A + a];
    if (a == 0) offs[(size_t)b * m + col] = j == 0 ? 0.f : log_uniform;
}
__global__ void k_add2(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] + y[i];
}
// critic_loss = (s0 + s1) / B + clamp(e^la) (conservative value - threshold);  actor_loss = mean(e^lt logp - qmin)
__global__ __launch_bounds__(256) void k_cql_metrics(const float* __restrict__ sums, int B, float weight, float threshold, const float* __restrict__ log_alpha,
                                                     const float* __restrict__ log_temp, const float* __restrict__ logp, const float* __restrict__ qmin,
                                                     float* __restrict__ critic_metric, float* __restrict__ actor_metric) {
    __shared__ float sm[256];
    if (threadIdx.x == 0) {
        const float cv = weight * ((sums[2] + sums[3]) - (sums[4] + sums[5])) / (2.0f * (float)B);
        critic_metric[0] = (sums[0] + sums[1]) / (float)B + fminf(fmaxf(expf(log_alpha[0]), 0.f), 1e6f) * (cv - threshold);
    }
    const float et = expf(log_temp[0]);
    float s = 0.f;
    for (int i = threadIdx.x; i < B; i += 256) s += et * logp[i] - qmin[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) actor_metric[0] = sm[0] / (float)B;
}

}  // namespace rl4rs
