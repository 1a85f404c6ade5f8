// Microbenchmark: do the matrix pipe and the VALU of one SIMD overlap?  One workgroup per CU, 8 waves (2 per SIMD), every wave
// runs `iters` iterations of 4 MFMA (v_mfma_f32_32x32x16_f16, 4 accumulators) and / or 28 VALU instructions of one kind.
//   mode 0: all waves MFMA only        mode 1: all waves VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
//   mode 3: every wave alternates (1 MFMA, 7 VALU) in one instruction stream
// kinds: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 v_add_u32, 4 v_cvt_pk_f16_f32, 5 ds_read_b128 (LDS instead of VALU)
// build + run:  hipcc --offload-arch=gfx950 -O3 -o tools/_ab/mfma_valu_overlap tools/mfma_valu_overlap.hip && tools/_ab/mfma_valu_overlap
// Result on MI355X (profiles/r04p_mfma_valu_overlap.txt): see DESIGN.md section 9.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu7(float (&v)[8], f32x2 (&p)[4], unsigned (&u)[8], const char* lds) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (KIND == 0) asm volatile("v_fma_f32 %0, %0, 1.0, %0" : "+v"(v[i]));
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p[i & 3]));
        else if (KIND == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        else if (KIND == 3) asm volatile("v_add_u32 %0, %0, %0" : "+v"(u[i]));
        else if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(v[i]));
        else { float4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + i * 1024)); v[i] += t.x; }
    }
}

template <int KIND>
__global__ __launch_bounds__(512) void k(int mode, int iters, float* out, long long* cyc) {
    __shared__ char lds[8192];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.001f + i); b[i] = (_Float16)(1.0f - lane * 0.002f); }
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float v[8]; f32x2 p[4]; unsigned u[8];
    for (int i = 0; i < 8; ++i) { v[i] = 0.001f * (lane + i); u[i] = lane + i; }
    for (int i = 0; i < 4; ++i) p[i] = f32x2{0.5f, 0.25f};
    for (int i = threadIdx.x; i < 2048; i += 512) reinterpret_cast<float*>(lds)[i] = 0.f;
    const bool do_mfma = mode == 0 || (mode == 2 && wave < 4);
    const bool do_valu = mode == 1 || (mode == 2 && wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    if (mode == 3) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
                valu7<KIND>(v, p, u, lds);
            }
        }
    } else {
        if (do_mfma)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[t], 0, 0, 0);
            }
        if (do_valu)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 4; ++r) valu7<KIND>(v, p, u, lds);
            }
    }
    if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(0)");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int t = 0; t < 4; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    for (int i = 0; i < 8; ++i) s += v[i] + (float)u[i];
    for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0 && blockIdx.x == 0) cyc[mode * 8 + wave] = t1 - t0;
}

template <int KIND>
void run(const char* name, float* out, long long* cyc) {
    const int iters = 2000;
    double w0[4], w4[4];
    for (int mode = 0; mode < 4; ++mode) {
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(512), 0, 0, mode, iters, out, cyc);
        hipDeviceSynchronize();
        long long h[8];
        (void)hipMemcpy(h, cyc + mode * 8, 64, hipMemcpyDeviceToHost);
        w0[mode] = (double)h[0] / iters; w4[mode] = (double)h[4] / iters;
    }
    printf("%-18s alone: MFMA %.0f (x2 waves %.0f)  VALU %.0f (x2 waves %.0f) | MFMA wave + VALU wave on a SIMD: %.0f / %.0f  (sum if serial %.0f) | "
           "interleaved in one wave: %.0f (x2 waves %.0f; MFMA + VALU alone = %.0f)\n", name, w0[0], w4[0], w0[1], w4[1], w0[2], w4[2],
           w0[0] + w0[1], w0[3], w4[3], w0[0] + w0[1]);
}

int main() {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipMalloc(&cyc, 4 * 8 * 8);
    printf("cycles per iteration = 4 MFMA (4 x 32 cycles of matrix pipe) and / or 28 instructions of the named kind, per wave\n");
    run<0>("v_fma_f32", out, cyc);
    run<1>("v_pk_fma_f32", out, cyc);
    run<2>("v_exp_f32", out, cyc);
    run<3>("v_add_u32", out, cyc);
    run<4>("v_cvt_pk_f16_f32", out, cyc);
    run<5>("ds_read_b128", out, cyc);
    return 0;
}
