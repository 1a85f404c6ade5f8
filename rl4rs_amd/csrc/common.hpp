// Shared host-side helpers for librl4rs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rl4rs_hip.h"

namespace rl4rs {

void set_error(const char* fmt, ...);

#define RL4RS_HIP_TRY(expr)                                                              \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            ::rl4rs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),    \
                               __FILE__, __LINE__);                                      \
            return RL4RS_EHIP;                                                           \
        }                                                                                \
    } while (0)

#define RL4RS_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            ::rl4rs::set_error(__VA_ARGS__);     \
            return RL4RS_EINVAL;                 \
        }                                        \
    } while (0)

#define RL4RS_LAUNCH_CHECK() RL4RS_HIP_TRY(hipGetLastError())

template <typename T>
inline int dev_alloc(T** p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(p), n * sizeof(T));
    if (e != hipSuccess) {
        set_error("hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
        return RL4RS_ENOMEM;
    }
    return RL4RS_OK;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// activation codes of the GEMM epilogues (the `act` argument of rl4rs_gemm_f32 in include/rl4rs_hip.h)
enum { ACT_NONE = 0, ACT_ELU = 1, ACT_SIGMOID = 2, ACT_TANH = 3, ACT_RELU = 4 };

// Launches of the generic fp32 MFMA GEMM (gemm.hip); ldw = leading dim of W [K,N].  Optional addend: row (m / add_div) of a
// row-major [ceil(M / add_div), ldadd] array is added before the activation.
int launch_gemm_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                    float* c, int64_t ldc, int M, int N, int K, int act, hipStream_t st, const float* addend = nullptr,
                    int64_t ldadd = 0, int add_div = 1);

// Small-M form (one 32 x 32 tile per workgroup, K split over its four waves; launch_gemm_f32 picks it by itself when the big
// tiling would leave most CUs idle) with an optional SECOND operand pair: C = act(A W + A2 W2 + bias + addend).
int launch_gemm_small(const float* a, int64_t lda, const float* w, int64_t ldw, int K, const float* a2, int64_t lda2, const float* w2,
                      int64_t ldw2, int K2, const float* bias, float* c, int64_t ldc, int M, int N, int act, hipStream_t st,
                      const float* addend = nullptr, int64_t ldadd = 0, int add_div = 1);
// dX [M, Kin] = dY [M, Nout] W[Kin, Nout]^T, zeroed where relu_of [M, Kin] (optional: the layer's forward output) is not > 0
int launch_gemm_nt(const float* dy, int64_t ldy, const float* w, int64_t ldw, float* dx, int64_t ldx, int M, int Kin, int Nout,
                   hipStream_t st, const float* relu_of = nullptr, int64_t ldr = 0);

// GEMM against a weight matrix pre-packed by pack_gemm_weight (gemm.hip); C = act(A W + bias + addend) with an optional
// row-major addend [M, ldadd]
int launch_gemm_packed(const float* a, int64_t lda, const float* wp, const float* bias, float* c, int64_t ldc,
                       int M, int N, int K, int act, hipStream_t st, const float* addend = nullptr, int64_t ldadd = 0);
std::vector<float> pack_gemm_weight(const float* w, int64_t ldw, int K, int N);
// fp16x2 form (scorer_mode FP16X2): weights pre-split into fp16 hi / lo fragment planes, activations split while staged
// mirror (optional): device-visible HOST memory that receives the same output elements from the epilogue, row stride ldm
int launch_gemm_h16(const float* a, int64_t lda, const float* wp16, const float* bias, float* c, int64_t ldc,
                    int M, int N, int K, int act, hipStream_t st, const float* addend = nullptr, int64_t ldadd = 0,
                    float* mirror = nullptr, int64_t ldm = 0);
std::vector<float> pack_gemm_weight_h16(const float* w, int64_t ldw, int K, int N);
// two chained layers in one launch (N1 <= 128, N1 % 16 == 0, N2 <= 128): c2 = act2(act1(a W1 + b1) W2 + b2)
int launch_gemm_h16_chain(const float* a, int64_t lda, const float* wp1, const float* bias1, int N1, int K1, int act1,
                          const float* wp2, const float* bias2, float* c2, int64_t ldc2, int N2, int act2, int M, hipStream_t st);

// device-side pack_gemm_weight_h16 (weights that change between launches): up to 4 matrices per launch, out = NT * KB * 512 + NT * 32 floats
struct PackH16Desc { const float* w; float* out; int ldw, K, N; };
struct PackH16Args { PackH16Desc d[4]; };
int launch_pack_h16_dev(const PackH16Desc* d, int n, hipStream_t st);
// whole no-grad forward of an amlp (contirl.hpp; hidden 256 x 256) in one launch, fp16x2 arithmetic; w*p = device-packed planes,
// proj [N / rep, 256] = x W1[:D] + b1, act [N, E] (E % 8 == 0, <= 64), out [N, K3] (K3 <= 64)
struct AmlpFwdH16 {
    const float* act; const float* proj; const char* w1p; const char* w2p; const char* w3p;
    const float* b2; const float* b3; float* out;
    int N, E, rep, K3, head_act;
};
int launch_amlp_fwd_h16(const AmlpFwdH16& a, hipStream_t st);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the FUNCTION, shared by every handle of the process: it is only ever
// raised (a second handle with a smaller shape must not lower the limit under the first one's launches - ADVICE r2 on k_ppo_pass,
// applied to every kernel whose LDS size depends on a handle's shape).  Defined in env.hip.
int raise_dyn_smem(const void* fn, size_t bytes);
int current_device();

// The NEXT rl4rs_dien_forward's observation also goes to `host_visible` (device-visible pinned host memory, [R, 256] floats) from
// the head GEMM's epilogue - step.hip's reference-shaped records (dien.hip); dien_obs_mirror_used: did that forward take it?
void dien_set_obs_mirror(rl4rs_dien* n, float* host_visible);
bool dien_obs_mirror_used(const rl4rs_dien* n);

// Training-mode recurrences as persistent kernels (recur_train.hpp, compiled into dien.hip; called from dientrain.hpp).
// Arrays are per sequence input (S <= 4 inputs run in ONE launch, grid.y = S); saved tensors are [N * L, Hd] row-major.
struct RecurTrainFwd {
    int N, L, S, Hd;
    const int32_t* iota;                  // device [N]: 0, 1, 2, ... (row n reads row block n of its input's pre-activations)
    const float* a1[4];                   // x-side pre-activations incl. bias, [N * L, 3 Hd] = [r | u | c]
    const float* wg[4]; const float* wc[4];      // h-side weights in MFMA fragment order (launch_pack_frag): [Hd x 2Hd], [Hd x Hd]
    const float* att[4];                  // attention rows [N, L] (AUGRU) or NULL (GRU)
    float *R[4], *U[4], *C[4], *H[4], *RH[4];    // out
    int hard;                             // keras recurrent_activation = hard_sigmoid for r / u (the lstm family's GRUs)
    int xblk[3];                          // column block of the r / u / c pre-activations in a1 (TF cells 0,1,2; keras [z|r|h]: 1,0,2)
};
struct RecurTrainBwd {
    int N, L, S, Hd;
    const float *R[4], *U[4], *C[4], *H[4];
    const float* att[4];
    const float* up_last[4]; int64_t ld_up;      // gradient of the final state (row stride ld_up) or NULL
    const float* up_all[4];               // gradient of every state or NULL
    const float* wcT[4]; const float* wgT[4];    // transposed h-side weights in fragment order: [Hd x Hd], [2Hd x Hd]
    // out: pre-activation gradients of the reset / update gates (row stride ld_g) and of the candidate (row stride ld_c), one
    // row per (sample, step).  TF cells: dr = dAg, du = dAg + Hd, ld_g = 2Hd; dc = dAc, ld_c = Hd.  keras [z|r|h] rows of
    // width 3Hd: du = dA, dr = dA + Hd, dc = dA + 2Hd, ld_g = ld_c = 3Hd.
    float *dr[4], *du[4], *dc[4];
    int64_t ld_g, ld_c;
    float* d_score[4];                    // out (AUGRU): d a_t [N, L] or NULL
    int hard;                             // hard_sigmoid gates: derivative 0.2 inside (0, 1), 0 at the clamps
};
int launch_recur_train_fwd(const RecurTrainFwd& f, hipStream_t st);
int launch_recur_train_bwd(const RecurTrainBwd& b, hipStream_t st);
// W rows k_off.. (leading dim ld) -> fragment order; transpose = 1 packs W^T (K = columns of W, N = rows taken)
int launch_pack_frag(const float* w, int64_t ld, int k_off, int K, int N, int transpose, float* out, hipStream_t st);
// the same for a K-slice of a taller operand: rows [k_dst, k_dst + K) of a [K_total x N] fragment buffer
int launch_pack_frag_slice(const float* w, int64_t ld, int k_off, int K, int N, int transpose, float* out, int K_total, int k_dst,
                           hipStream_t st);

// Power-of-two prescale of a weight tile that goes through the fp16 hi + lo split: s = 1 while the tile sits comfortably inside
// fp16 (2^-6 <= max |w| < 2^14), else s = 2^k with max |w| * s in [2^13, 2^14) - multiplying by s is exact in fp32, hi = fp16(w s)
// is then neither out of range (large tiles) nor a subnormal with a handful of significant bits (tiny tiles), and the kernels
// divide the accumulator by s (exactly) in their epilogues: the split form has no weight-range condition.  ORDINARY tiles are
// left alone on purpose.  Normalising every tile (so that every lo part becomes a normal fp16 number with all 10 mantissa bits
// in use) was measured: nothing to gain against the fp64 oracle (obs 2.2e-6 either way - the fp32 accumulation dominates),
// and the denser operand bits cost 2 % of the recurrence's speed on this power-limited kernel (same-box A/B);
// -DRL4RS_PRESCALE_UP builds that form.
inline float pow2_prescale(float maxabs) {
    if (!(maxabs > 0.f) || !(maxabs < 3.0e38f)) return 1.f;
    int e = 0;
    (void)frexpf(maxabs, &e);             // maxabs = m * 2^e, m in [0.5, 1)  ->  maxabs * 2^(14 - e) in [2^13, 2^14)
    int k = 14 - e;
#ifndef RL4RS_PRESCALE_UP
    if (e <= 14 && e > -6) k = 0;         // 2^-6 <= maxabs < 2^14: as is
#endif
    if (k > 100) k = 100;                 // keep s (and s * other weights) far from the fp32 range ends
    if (k < -100) k = -100;
    return ldexpf(1.f, k);
}

// fp32 -> fp16 bits, round to nearest even (subnormals kept)
inline uint16_t f32_to_f16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                 // overflow -> inf
    if (x < 0x38800000u) {                                                   // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int e = (int)(x >> 23);
        uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                                           // 14..24
        uint32_t r = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}
inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1f, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int s = 0; while (!(m & 0x400u)) { m <<= 1; ++s; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - s) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

}  // namespace rl4rs
