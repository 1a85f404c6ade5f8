// act / evaluate / loss + backward of whole batches at the DEFAULT policy shape (256 -> 64 -> 284 + 1, 9 mask words) on
// v_mfma_f32_4x4x1 tiles: the phase-A body of k_ppo_pass<true> (ppo_pass.hpp) as an ordinary kernel.  Included by policy.hip behind
// k_policy_tile (needs TileArgs, policy_row_outputs, policy_row_loss, pass_rsrc / bld).
//
// Reference: rl4rs/nets/rllib/rllib_mask_model.py:7-64 (the FC mask model; losses: policy.hip's header).
//
// k_policy_tile (the 32x32x2 form, kept for every other shape) runs 8 samples per workgroup in 32-row MFMA tiles: 24 of 32 rows
// idle, two waves per SIMD's matrix pipe - its three products are MFMA issue (round 6).  Here a
// workgroup's 8 rows are two 4-row tiles with no idle rows, every stage splits K over the 8 waves (B operand = one 256-byte weight
// row per k, lane = column), partial sums meet in LDS in a fixed order, never more than 63 requests in flight per wave, each stage's
// weights requested behind the MFMAs of the stage before.  72 KB of LDS: two workgroups per CU.
//   MODE 0 = act (Gumbel-max sample), 1 = evaluate given actions, 2 = training forward + loss + backward to dHpre (needs w2t).
#pragma once

// (std_slice / std_put: ppo_pass.hpp)
constexpr size_t TILE_STD_SMEM = (size_t)(2080 + 2304 + 544 + 8192 + 2304 + 72) * 4;      // s_x | s_g | s_hh | s_p | s_lg | s_mk

template <int MODE>
__global__ __launch_bounds__(512) void k_policy_tile_std(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PolDims d = a.d;
    d.OD = 256; d.HID = 64; d.A = 284; d.AE = 285; d.W = 9;
    float* s_x = reinterpret_cast<float*>(smem);      // [8][260] observation rows
    float* s_g = s_x + 2080;                           // [8][288] d loss / d [logits | value], pad columns 285..287 zero (MODE 2)
    float* s_hh = s_g + 2304;                          // [8][68]  hidden
    float* s_p = s_hh + 544;                           // 8192 floats of partial sums (layouts as in k_ppo_pass<true>)
    float* s_lg = s_p + 8192;                          // [8][288] masked logits | value
    uint32_t* s_mk = reinterpret_cast<uint32_t*>(s_lg + 2304);      // [8][9]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int n_prm = 256 * 64 + 64 + 64 * 285 + 285, o_b1 = 256 * 64, o_w2 = o_b1 + 64, o_b2 = o_w2 + 64 * 285;
    const int r0 = blockIdx.x * 8;
    const int nrow = min(8, a.N - r0);                 // live rows of this workgroup (rows past them read zeros and store nothing)
    const __amdgpu_buffer_rsrc_t rs_prm = pass_rsrc(a.prm, (size_t)n_prm * 4);
    const int vl = lane * 4;
    float bv1[32], bv2[32], bv2t[8], b2v[5];
#pragma unroll
    for (int u = 0; u < 32; ++u) bv1[u] = bld(rs_prm, vl, (wave * 32 + u) * 256);                         // W1 rows 32 w .. 32 w + 31
    const float b1v = bld(rs_prm, vl, o_b1 * 4);
#pragma unroll
    for (int i = 0; i < 5; ++i) b2v[i] = bld(rs_prm, vl, (o_b2 + 64 * i) * 4);                            // (past the buffer: 0)
#pragma unroll
    for (int u = 0; u < 8; ++u) bv2t[u] = bld(rs_prm, vl, (o_w2 + (wave * 8 + u) * 285 + 256) * 4);       // W2e rows 8 w .. 8 w + 7, columns 256 ..
    {
        const __amdgpu_buffer_rsrc_t rs_o = pass_rsrc(a.obs + (size_t)r0 * 256, (size_t)nrow * 256 * 4);
        float x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = bld(rs_o, tid * 4, 2048 * u);
        uint32_t mw = 0;
        if (a.mask) mw = __builtin_amdgcn_raw_buffer_load_b32(pass_rsrc(a.mask + (size_t)r0 * 9, (size_t)nrow * 9 * 4), tid * 4, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u) s_x[(u * 2 + (tid >> 8)) * 260 + (tid & 255)] = x[u];
        if (a.mask && tid < 72) s_mk[tid] = mw;
    }
    __syncthreads();
    f32x4_t o0, o1;
    std_slice<8>(s_x + (lane & 3) * 260 + wave * 32, 260, bv1, o0, o1);
#pragma unroll
    for (int u = 0; u < 32; ++u) bv2[u] = bld(rs_prm, vl, (o_w2 + ((wave >> 2) * 32 + u) * 285 + (wave & 3) * 64) * 4);
    __builtin_amdgcn_sched_barrier(0);
    std_put(s_p + wave * 512, 64, lane, o0, o1);
    __syncthreads();
    {   // thread = (row wave, column lane)
        float sum = b1v;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) sum += s_p[pp * 512 + wave * 64 + lane];
        const float h = tanhf(sum);
        s_hh[wave * 68 + lane] = h;
        if (MODE == 2 && wave < nrow) a.H[(size_t)(r0 + wave) * 64 + lane] = h;
    }
    __syncthreads();
    std_slice<8>(s_hh + (lane & 3) * 68 + (wave >> 2) * 32, 68, bv2, o0, o1);
    f32x4_t t0, t1;
    std_slice<2>(s_hh + (lane & 3) * 68 + wave * 8, 68, bv2t, t0, t1);
    float wv[36];
    if (MODE == 2) {
        const __amdgpu_buffer_rsrc_t rs_w2t = pass_rsrc(a.w2t, (size_t)64 * 285 * 4);
#pragma unroll
        for (int u = 0; u < 36; ++u) wv[u] = bld(rs_w2t, vl, (wave * 36 + u) * 256);                      // (rows >= 285: past the buffer, 0)
        __builtin_amdgcn_sched_barrier(0);
    }
    std_put(s_p + (wave >> 2) * 2048 + (wave & 3) * 64, 256, lane, o0, o1);
    std_put(s_p + 4096 + wave * 512, 64, lane, t0, t1);
    __syncthreads();
    {   // one row per wave: join the row's partial sums (+ bias, + action mask), log-sum-exp, then outputs or loss
        float* so = s_lg + wave * 288;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int c = lane + 64 * i;
            if (c < 285) {
                float v = b2v[i];
                if (i < 4) {
                    v = (v + s_p[wave * 256 + c]) + s_p[2048 + wave * 256 + c];
                } else {
#pragma unroll
                    for (int pp = 0; pp < 8; ++pp) v += s_p[4096 + pp * 512 + wave * 64 + lane];
                }
                const uint32_t mw = a.mask ? s_mk[wave * 9 + (c >> 5)] : 0xffffffffu;
                if (c < 284 && !((mw >> (c & 31)) & 1u)) v = v + (-3.4028235e38f);
                so[c] = v;
            }
        }
        if (MODE == 2 && lane < 3) s_g[wave * 288 + 285 + lane] = 0.f;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float mx = -3.4028235e38f;
        for (int c = lane; c < 284; c += 64) mx = fmaxf(mx, so[c]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int c = lane; c < 284; c += 64) se += expf(so[c] - mx);
        const float lse = mx + logf(wave_sum(se));
        if (MODE != 2) {
            if (wave < nrow)
                policy_row_outputs<MODE == 0>(d, so, lse, r0 + wave, lane, a.seed, a.step, a.actions, a.logp, a.value, a.entropy, a.logits_out);
            return;
        }
        if (wave < nrow) {
            const float4 tm = policy_row_loss(d, a.L, so, lse, r0 + wave, lane, s_g + wave * 288, a.dOut);
            if (lane == 0) a.terms[r0 + wave] = tm;
        } else {
            for (int c = lane; c < 285; c += 64) s_g[wave * 288 + c] = 0.f;
        }
    }
    __syncthreads();
    // dH = dOut W2e^T: wave w multiplies k in [36 w, 36 w + 36)
    std_slice<9>(s_g + (lane & 3) * 288 + wave * 36, 288, wv, o0, o1);
    std_put(s_p + wave * 512, 64, lane, o0, o1);
    __syncthreads();
    if (wave < nrow) {
        float sum = 0.f;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) sum += s_p[pp * 512 + wave * 64 + lane];
        const float h = s_hh[wave * 68 + lane];
        a.dHpre[(size_t)(r0 + wave) * 64 + lane] = sum * (1.f - h * h);
    }
}
