// Small embedding-gather kernels shared by the simulator families (simnet.hpp, in dien.hip's translation unit) and the
// raw-state policy encoder (policy.hip).  `static`: each translation unit gets its own copy.
#pragma once
#include <cstdint>

namespace rl4rs {

// mean over `len` embedding rows (keras GlobalAveragePooling1D: every position counts, id 0 included). One wave per row.
static __global__ __launch_bounds__(256) void k_emb_mean(const int32_t* __restrict__ ids, int n, int len, int H, int E,
                                                  const float* __restrict__ table, float* __restrict__ out, int64_t ld,
                                                  int off) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    const float inv = 1.0f / (float)len;
    for (int k = lane; k < E; k += 64) {
        float s = 0.f;
        for (int j = 0; j < len; ++j) {
            int id = min(max(ids[(size_t)row * len + j], 0), H - 1);
            s += table[(size_t)id * E + k];
        }
        out[(size_t)row * ld + off + k] = s * inv;
    }
}

// Flatten(embedding rows): out[row, off + j*E + k] = table[ids[row, j], k]
static __global__ __launch_bounds__(256) void k_emb_flatten(const int32_t* __restrict__ ids, int n, int len, int H, int E,
                                                     const float* __restrict__ table, float* __restrict__ out,
                                                     int64_t ld, int off) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    for (int j = 0; j < len; ++j) {
        int id = min(max(ids[(size_t)row * len + j], 0), H - 1);
        for (int k = lane; k < E; k += 64) out[(size_t)row * ld + off + (size_t)j * E + k] = table[(size_t)id * E + k];
    }
}

// out[row, off + k] = src[slots[row / group], k]
static __global__ __launch_bounds__(256) void k_gather_slots(const float* __restrict__ src, int W, const int32_t* __restrict__ slots,
                                                      int n, int group, float* __restrict__ out, int64_t ld, int off) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + wave;
    if (row >= n) return;
    const int slot = slots[row / group];
    for (int k = lane; k < W; k += 64) out[(size_t)row * ld + off + k] = src[(size_t)slot * W + k];
}

}  // namespace rl4rs
