// Shared host-side helpers for librl4rs_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
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

// Launches of the generic fp32 MFMA GEMM (gemm.hip); ldw = leading dim of W [K,N].
int launch_gemm_f32(const float* a, int64_t lda, const float* w, int64_t ldw, const float* bias,
                    float* c, int64_t ldc, int M, int N, int K, int act, hipStream_t st);

// GEMM against a weight matrix pre-packed by pack_gemm_weight (gemm.hip)
int launch_gemm_packed(const float* a, int64_t lda, const float* wp, const float* bias, float* c, int64_t ldc,
                       int M, int N, int K, int act, hipStream_t st);
std::vector<float> pack_gemm_weight(const float* w, int64_t ldw, int K, int N);

}  // namespace rl4rs
