// v_mfma_f32_4x4x1_16B_f32 as a 4-row x 64-column GEMM step, and the weight-ring product phase built on it (shared by the 8-row
// training recurrences, recur8.hpp in dien.hip, and the fused minibatch networks of the continuous learners, contirl.hpp in
// policy.hip).  Layout (tools/mfma_4x4_probe.hip): A = X[lane % 4][k] the same for all 16 blocks, B = W[k][lane] = 64 distinct
// columns, register i of lane l = out[i][l].
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

namespace rl4rs {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

namespace r8 {
constexpr int RW = 8;           // rows per workgroup
constexpr int RS = 16;          // weight ring: 16-byte loads in flight per wave (64 registers), refilled in consumption order

// byte offset of lane l's 16-byte weight entry inside a (column-group of 64, kq) slot of a fragment-order buffer with KB k-blocks:
// tile nt = 2 * group + (l >> 5) -> + (l >> 5) * KB * 1024 ; column (l & 31) -> + (l & 31) * 16 ; (kb, half) = kq -> + kq * 512 (scalar)
__device__ __forceinline__ int lane_off(int lane, int KB) { return (lane >> 5) * KB * 1024 + (lane & 31) * 16; }

__device__ __forceinline__ float4 ldw(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    struct b4 { float x, y, z, w; } f = __builtin_bit_cast(b4, v);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ float el(const float4& v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }
__device__ __forceinline__ float hard_sig(float x) { return fminf(fmaxf(0.2f * x + 0.5f, 0.f), 1.f); }

// One product phase of a wave: acc[g][m][chain] += A[tile m rows][k] * W_g[k][own 64 columns] over NQ quads of k.
//   A: LDS rows (aptr[m] = row (lane & 3) of tile m), read one quad ahead into a ping-pong pair;
//   W: NG weight streams consumed quad by quad through the RS-slot register ring: the slot an MFMA group has just used is
//   refilled at once with the load RS positions further down the stream - `cur(q, g)` inside this phase, `next(i)` = the i-th
//   load of the FOLLOWING phase's stream during the last round (weights do not depend on the recurrence, so the ring runs
//   across phases and steps).  sched_barriers pin that order: left alone, the compiler sinks every load to its use and waits
//   with vmcnt(0) in front of each MFMA group (seen in the ISA of the first version of this file: 2x slower).
template <int NG, int MTW, int P, typename CurLd, typename NextLd>
__device__ __forceinline__ void phase_rt(f32x4_t (&acc)[NG][MTW][P], const float* (&aptr)[MTW], float4 (&ring)[RS], CurLd cur, NextLd next, const int ROUNDS) {
    constexpr int QR = RS / NG, PR = QR / 2;
    static_assert(QR % 4 == 0, "ring rounds");
    // A is read a PAIR of quads ahead (two quads of 8 - 16 MFMAs each cover the LDS round trip; one did not in the single-stream
    // phases: 88 cycles of MFMAs against ~100 of latency)
    float4 ab[2][2][MTW];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int m = 0; m < MTW; ++m) ab[0][k][m] = *reinterpret_cast<const float4*>(aptr[m] + k * 4);
    auto round = [&](int q0, auto last_tag) {
        constexpr bool last = decltype(last_tag)::value;
#pragma unroll
        for (int ip = 0; ip < PR; ++ip) {
            if (!(last && ip == PR - 1)) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
                        ab[(ip + 1) & 1][k][m] = *reinterpret_cast<const float4*>(aptr[m] + (q0 + 2 * ip + 2 + k) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = 2 * ip + k;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int m = 0; m < MTW; ++m)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g][m][j % P] = __builtin_amdgcn_mfma_f32_4x4x1f32(el(ab[ip & 1][k][m], j), el(ring[i * NG + g], j), acc[g][m][j % P], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) ring[i * NG + g] = last ? next(i * NG + g) : cur(q0 + i + QR, g);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll 1
    for (int r = 0; r < ROUNDS - 1; ++r) round(r * QR, std::false_type());
    round((ROUNDS - 1) * QR, std::true_type());
}
// compile-time length: NQ quads of k (a multiple of the quads per ring round)
template <int NQ, int NG, int MTW, int P, typename CurLd, typename NextLd>
__device__ __forceinline__ void phase(f32x4_t (&acc)[NG][MTW][P], const float* (&aptr)[MTW], float4 (&ring)[RS], CurLd cur, NextLd next) {
    static_assert(NQ % (RS / NG) == 0 && NQ >= RS / NG, "ring rounds");
    phase_rt<NG, MTW, P>(acc, aptr, ring, cur, next, NQ / (RS / NG));
}
}  // namespace r8

}  // namespace rl4rs
