// Types shared by the translation units of the DIEN scorer: dien.hip (every scorer kernel but one, built without SLP vectorisation:
// packed fp32 VALU serialises with the matrix pipe, build.py) and augru_x.hip (k_augru_x, built WITH it: the reward-sized form
// sits at the 256-register limit and spills 8 registers without - VERDICT r4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RL4RS_X_NRES
#define RL4RS_X_NRES 12          // k_augru_x: weight items of a wave's step resident in registers (of 48)
#endif
#ifndef RL4RS_X_RING
#define RL4RS_X_RING 4           // ... register ring of the streamed rest (48 - NRES items, a multiple of RING)
#endif
#ifndef RL4RS_X2_NRES
#define RL4RS_X2_NRES 4          // the 64-row form of k_augru_x (registers hold the second row tile's accumulators instead)
#endif
#ifndef RL4RS_X2_RING
#define RL4RS_X2_RING 2
#endif

namespace rl4rs {

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
struct h8bits { half8_t v; };

// fp16 hi / lo split of two fp32 values: hi = RNE(x) as a packed pair, lo = RNE(x - hi).  The difference comes from ONE
// v_fma_mix_f32 (f16 source read straight out of the packed pair, exact) instead of a conversion back plus a subtraction.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_h16_pair(float x0, float x1, half2_t& hi, half2_t& lo) {
    unsigned h;
    float d0, d1;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d0) : "v"(h), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d1) : "v"(h), "v"(x1));
    unsigned l;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(l) : "v"(d0), "v"(d1));
    hi = __builtin_bit_cast(half2_t, h);
    lo = __builtin_bit_cast(half2_t, l);
}

struct RecurArgs {
    int n_rows, L, group;
    const float* xbase[4]; int64_t xld; int xoff; int64_t xbytes;   // xbytes = size of one x table (< 4 GB)
    const int32_t* ids;      // GRU: [n_rows, L]
    const int32_t* slots; int64_t slots_stride;    // AUGRU: [n_seq][n_rows/group]
    const float* wg[4]; const float* wc[4];
    const float* att; int64_t att_stride;          // AUGRU: [n_seq][att_stride] rows of L
    float* out; int64_t out_ld; int out_off; int out_seq_off;   // GRU: h1 cache rows ; AUGRU: allf
    int slot_base;
    unsigned long long* trace;   // -DRL4RS_H16_TRACE timing experiments only
    int* range_flag;             // k_augru_h16: set to 1 when a state leaves the fp16 range (|h| >= 6e4 or NaN)
    int hard_gates;              // GRU mode: keras hard_sigmoid gates (simnet.hpp) instead of sigmoid
    int steps;                   // debug: run only the first `steps` recurrence steps (0 = all L)
    const int32_t* order;        // k_augru_x: processing order of the row groups (NULL = identity)
    int final_only;              // GRU mode: write only the last state, to out[(slot_base + row) * out_ld + out_off]
    // k_gru_h16: [L][NH] - row t = the state after t + 1 steps on item id 0 from h = 0 (the same for every row: pad_sequences pads in
    // FRONT, rl4rs/utils/datautil.py:44).  A workgroup whose rows all start with at least t0 zero ids copies rows 0 .. t0 - 1 of this
    // table into their outputs and starts the recurrence at step t0 from row t0 - 1.  NULL = every step is computed.
    const float* pad;
    int32_t* lead_out;           // k_gru_h16 (with pad): [slot] the number of leading zero ids of every encoded row
    // k_augru_x<.., PAD = true>: lead[sq][slot] as written by k_gru_h16, pad_slot = the cache slot that holds the projections of the
    // all-zero sequence.  Step t of a row whose sequence starts with more than t zero ids reads the pad slot's row t instead of its
    // own (the same bytes: the first GRU's state after t + 1 zero ids does not depend on the row) - 4096 envs then share ONE copy of
    // their padding's projections in L2 instead of streaming 4096 identical ones from HBM.
    const int32_t* lead[4];
    int pad_slot;
    // fp16x2 AUGRU kernels: every 32-column tile of the reset / update / candidate weight matrices (and the same columns of the
    // cached x-side projections, biases folded) is stored multiplied by its own power of two s (rl4rs_dien_create: max |w| * s in
    // [2^13, 2^14) over the tile), so the fp16 hi + lo split keeps its 22 bits whatever the scale of a checkpoint's weights, no
    // weight is "too large for fp16", and an outlier costs precision in its own tile only.  A wave owns exactly one tile per
    // gate, so its three constants are wave-uniform scalars.  The pre-activation is acc / s; the division rides on the constant
    // the activation multiplies by anyway: sigmoid(acc / s) = 1 / (1 + exp2(acc * k)), k = -log2(e) / s; tanh: k = 2 log2(e) / s -
    // exact (powers of two), same instruction count.
    float k_r[4][8], k_u[4][8], k_c[4][8];         // [sequence input][column tile = wave]
    // k_recur<..., SAVE = true> (training forward, recur_train.hpp): per sequence input the attention rows [n_rows, L] (NULL = 0:
    // a plain GRU) and, per (row, step), everything BPTT needs - reset gate, update gate BEFORE the attention factor, candidate,
    // new state, r * h_prev - each [n_rows * L, NH] row-major.
    const float* sv_att[4];
    float *sv_r[4], *sv_u[4], *sv_c[4], *sv_h[4], *sv_rh[4];
    int sv_blk[3];               // SAVE: column block of the r / u / c pre-activations inside a row of xbase (TF cells 0,1,2; keras GRU 1,0,2)
};

// activations of a pre-activation that is stored scaled by a power of two (k = -log2(e) / s resp. 2 log2(e) / s, RecurArgs)
__device__ __forceinline__ float gate_sigmoid_k(float x, float k) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k * x)); }
__device__ __forceinline__ float gate_tanh_k(float x, float k) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k * x)); }

// one 16-byte buffer load as 8 fp16 values (a weight fragment plane of the fp16x2 kernels)
__device__ __forceinline__ half8_t buf_load_h8(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return __builtin_bit_cast(half8_t, v);
}

}  // namespace rl4rs
