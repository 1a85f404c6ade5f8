// Probe for the "skinny tile" lever of DESIGN.md section 9 (what comes next): v_mfma_f32_4x4x1_16B_f32 as a 4-row x 64-column
// GEMM step (A broadcast over the 16 blocks, B = one weight row of 64 columns).  Checks the operand / result layout against a
// scalar reference and times the instruction (cycles per MFMA, one wave per SIMD and two).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_ab/mfma_4x4_probe tools/mfma_4x4_probe.hip && tools/_ab/mfma_4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// out[4 rows][64 cols] = X[4][K] W[K][64]: lane l supplies A = X[l % 4][k] (block l / 4; the same for every block) and
// B = W[k][(l / 4) * 4 + l % 4] = W[k][l]; D: lane l, register i = out[i][l]?  (the layout this probe verifies)
__global__ void k_check(const float* X, const float* W, float* out, int K) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < K; ++k) acc = __builtin_amdgcn_mfma_f32_4x4x1f32(X[(l & 3) * K + k], W[k * 64 + l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[i * 64 + l] = acc[i];
}

__global__ __launch_bounds__(512) void k_time(int iters, int waves, float* out, long long* cyc) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = lane * 0.001f, b = 1.0f - lane * 0.002f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < waves)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 0, 0, 0);
        }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}

int main() {
    const int K = 37;
    float hX[4 * K], hW[K * 64], hO[4 * 64];
    for (int i = 0; i < 4 * K; ++i) hX[i] = sinf(0.37f * i);
    for (int i = 0; i < K * 64; ++i) hW[i] = cosf(0.11f * i);
    float *X, *W, *O; long long* cyc;
    (void)hipMalloc(&X, sizeof(hX)); (void)hipMalloc(&W, sizeof(hW)); (void)hipMalloc(&O, 256 * 512 * 4); (void)hipMalloc(&cyc, 64);
    (void)hipMemcpy(X, hX, sizeof(hX), hipMemcpyHostToDevice); (void)hipMemcpy(W, hW, sizeof(hW), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check, dim3(1), dim3(64), 0, 0, X, W, O, K);
    (void)hipMemcpy(hO, O, sizeof(hO), hipMemcpyDeviceToHost);
    double worst = 0.0;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 64; ++j) {
            double r = 0.0;
            for (int k = 0; k < K; ++k) r += (double)hX[i * K + k] * hW[k * 64 + j];
            worst = fmax(worst, fabs(r - hO[i * 64 + j]));
        }
    printf("layout check: out[i][l] in register i of lane l, A = X[l %% 4][k], B = W[k][l]: max |error| vs float64 = %.3g (%s)\n", worst,
           worst < 1e-4 ? "layout CONFIRMED" : "layout WRONG");
    const int iters = 4000;
    for (int waves = 4; waves <= 8; waves += 4) {
        hipLaunchKernelGGL(k_time, dim3(256), dim3(512), 0, 0, iters, waves, O, cyc);
        (void)hipDeviceSynchronize();
        long long h[8];
        (void)hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
        printf("%d waves per SIMD: %.2f cycles per v_mfma_f32_4x4x1_16B_f32 per wave (512 flop each: %.1f flop / cycle / SIMD; "
               "v_mfma_f32_32x32x2_f32 = 4096 flop / 64 cycles = 64)\n", waves / 4, (double)h[0] / (iters * 8.0),
               512.0 * waves / 4 / ((double)h[waves - 1] / (iters * 8.0)));
    }
    return 0;
}
