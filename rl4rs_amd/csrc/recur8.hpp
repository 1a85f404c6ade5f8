// Small-tile forms of the persistent TRAINING recurrences (included by recur_train.hpp, compiled into dien.hip).
//
// Why: script/supervised_train.py:12-46 trains with batch 256.  In 32-row tiles (k_recur<NH, SAVE>, k_recur_bwd<NH>) that is
// 8 workgroups per sequence input - 16 of 256 CUs for the DIEN cells - each MFMA-bound per CU: one AUGRU step is a
// [32 x 256] x [256 x 768] fp32 product = 12.6 MFLOP at the 0.61 TF/s of ONE CU = 20 us of the 27 us a step takes, times 64
// dependent steps.  The step time does not depend on how many CUs run, so the only lever is a smaller tile:
// v_mfma_f32_4x4x1_16B_f32 multiplies a 4-ROW tile at close to the full fp32 matrix rate (tools/mfma_4x4_probe.hip:
// A = X[lane % 4][k] the same for all 16 blocks, B = W[k][lane] = 64 distinct columns, register i of lane l = out[i][l]).
//
//   workgroup = 8 rows (two 4-row tiles) of ONE sequence input, 4 waves (one per SIMD);
//   NH = 256: wave w owns hidden columns [64w, 64w + 64) of all three gates for BOTH tiles (every weight fragment is fetched once per
//             workgroup and step: 786 KB through the 64 B/clk L1 return path = 12.3 k cycles, beside 1536 MFMAs per wave);
//   NH = 128: wave (w & 1) owns 64 hidden columns, wave >> 1 picks the tile (weights fetched twice: 393 KB per step);
//   a lane owns ONE hidden column: 4 rows of a tile in the 4 registers of an accumulator, so every gate / blend is a
//   per-register scalar op, the saved tensors leave as 256-byte coalesced row segments, and h / r*h go to LDS as plain
//   [8][NH + 4] rows (A operand = ds_read_b128 of 4 consecutive k of row lane % 4: four broadcast addresses, conflict-free);
//   weights are read from the SAME fragment-order buffers the 32-row kernels use (launch_pack_frag): the 16-byte entry of
//   (32-column tile, k-block, half, column) holds 4 consecutive k of one column - a wave's b128 load is two contiguous 512-byte
//   pieces;
//   independent accumulator chains (>= 4 per wave and phase: x4x1 MFMAs are 2 passes, a dependent one would stall), the x-side
//   pre-activations enter as the first chain's initial value (MFMA C-in), weights stream through a 4-deep register ring that runs
//   across phases and steps (they do not depend on the recurrence).
// 256 samples x 2 inputs = 64 workgroups instead of 16.  Same equations, same saved quantities, same argument structs as the
// 32-row forms; the launchers in recur_train.hpp pick the form from the grid it would fill.
#pragma once
#include "mfma4.hpp"

namespace rl4rs {


// ------------------------------------------------------------------------------------------------------------------ forward
// RecurArgs as k_recur<NH, true, U, 0, true> takes them (recur_train_fwd_t): xbase = [N * L, 3 NH] pre-activations, row n reads
// row block n; wg / wc in fragment order; sv_* outputs; sv_att NULL = plain GRU; hard_gates; sv_blk.
// A FIFTH wave moves everything that is not a weight: it requests the x-side pre-activations a whole step ahead (they wait in its
// registers, then go to an LDS stage the matrix waves take their accumulators' initial values from) and copies the saved r, u,
// r*h, c, h tiles of a step from LDS to memory.  The matrix waves issue no global load but their weight ring and no global store:
// vmcnt retires in order, so every x load or store in their queue was a wait in front of the ring (9.9 -> 8.x us per AUGRU step).
// RWS = rows per workgroup: 8 (two 4-row tiles) or, NH = 256 only, 4 (ONE tile per wave: half the MFMAs per step against the same
// weight stream - a 256-sample minibatch x 2 inputs is 128 workgroups; round 6).
template <int NH, int RWS = 8>
__global__ __launch_bounds__(320) void k_recur8_fwd(RecurArgs a) {
    using namespace r8;
    constexpr int RW = RWS;                                       // (shadows r8::RW)
    static_assert(RWS == 8 || (RWS == 4 && NH == 256), "rows per workgroup");
    constexpr int NCW = NH / 64, MTW = (NH == 256) ? RWS / 4 : 1, KB = NH / 8, NQ = NH / 4, LDH = NH + 4, NWT = NH / 32, ST = RW * NH;
    constexpr int P1 = (2 * MTW >= 4) ? 1 : 2, P2 = 4 / MTW;     // accumulator chains per (gate, tile) in phase 1 / 2
    static_assert(NH == 128 || NH == 256, "hidden width");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* hb = reinterpret_cast<float*>(smem);                  // [8][LDH]  h_{t-1} (A operand of phase 1; saved as H)
    float* rhb = hb + RW * LDH;                                  // [8][LDH]  r * h_{t-1} (A operand of phase 2; saved as RH)
    float* tr = rhb + RW * LDH;                                  // [8][NH]   r, u, c of the step (saved by the fifth wave)
    float* tu = tr + ST;
    float* tc = tu + ST;
    float* xs = tc + ST;                                         // [3][8][NH] x-side pre-activations of the step: r | u | c blocks
    float* s_att = xs + 3 * ST;                                  // [8][L + 1]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RW, sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;

    for (int i = tid; i < RW * LDH; i += 320) hb[i] = 0.f;
    for (int i = tid; i < RW * L; i += 320) {
        const int r = i / L, t = i - r * L;
        const int gr = min(row0 + r, a.n_rows - 1);
        s_att[r * LDT + t] = a.sv_att[sq] ? a.sv_att[sq][(size_t)gr * L + t] : 0.f;
    }
    if (wave == 4) {
        // ---- fifth wave.  float4 k of lane l covers element 4 * (l + 64 k) of an [8][NH] tile
        constexpr int NV = ST / 4 / 64;
        const float* xb = a.xbase[sq];
        const int blk[3] = {a.sv_blk[0] * NH, a.sv_blk[1] * NH, a.sv_blk[2] * NH};
        // (f32x4_t, not float4: a struct copy between address spaces is a memcpy the optimiser does not turn into register values - the
        // arrays stayed in scratch memory)
        f32x4_t vru[2][NV], vc[NV];
        float* const sv_r = a.sv_r[sq]; float* const sv_u = a.sv_u[sq]; float* const sv_c = a.sv_c[sq];
        float* const sv_h = a.sv_h[sq]; float* const sv_rh = a.sv_rh[sq];
        // (macros and inline index arithmetic: index arrays / by-reference closures left the register arrays in scratch memory)
#define R8_ER(k) (((lane + 64 * (k)) * 4) / NH)
#define R8_EC(k) (((lane + 64 * (k)) * 4) % NH)
#define R8_GRL(k) ((size_t)min(row0 + R8_ER(k), a.n_rows - 1) * L)
#define R8_REQUEST_RU(T)                                                                                                               \
        _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                                                  \
            _Pragma("unroll") for (int k = 0; k < NV; ++k)                                                                             \
                vru[g][k] = *reinterpret_cast<const f32x4_t*>(xb + (R8_GRL(k) + (T)) * (3 * NH) + blk[g] + R8_EC(k));
#define R8_REQUEST_C(T)                                                                                                                \
        _Pragma("unroll") for (int k = 0; k < NV; ++k)                                                                                 \
            vc[k] = *reinterpret_cast<const f32x4_t*>(xb + (R8_GRL(k) + (T)) * (3 * NH) + blk[2] + R8_EC(k));
#define R8_DEPOSIT_RU()                                                                                                                \
        _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                                                  \
            _Pragma("unroll") for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4_t*>(xs + g * ST + (lane + 64 * k) * 4) = vru[g][k];
#define R8_DEPOSIT_C()                                                                                                                 \
        _Pragma("unroll") for (int k = 0; k < NV; ++k) *reinterpret_cast<f32x4_t*>(xs + 2 * ST + (lane + 64 * k) * 4) = vc[k];
        // an LDS tile (row stride LD) -> rows of step T of a saved tensor
#define R8_SAVE(DST, TILE, LD, T)                                                                                                      \
        _Pragma("unroll") for (int k = 0; k < NV; ++k)                                                                                 \
            if (row0 + R8_ER(k) < a.n_rows) *reinterpret_cast<f32x4_t*>((DST) + (R8_GRL(k) + (T)) * NH + R8_EC(k)) = *reinterpret_cast<const f32x4_t*>((TILE) + R8_ER(k) * (LD) + R8_EC(k));
        R8_REQUEST_RU(0)
        R8_DEPOSIT_RU()
        R8_REQUEST_C(0)
        R8_REQUEST_RU(min(1, L - 1))
        __syncthreads();
        // (requests past the last step re-read the last one and their deposits are never consumed: no conditional touches the
        // register arrays)
        for (int t = 0; t < L; ++t) {
            // -- while the matrix waves run phase 1 of step t: c, h of step t - 1 leave; the candidate's x of step t arrives
            if (t > 0) { R8_SAVE(sv_c, tc, NH, t - 1) R8_SAVE(sv_h, hb, LDH, t - 1) }
            R8_DEPOSIT_C()
            R8_REQUEST_C(min(t + 1, L - 1))
            __syncthreads();
            // -- phase 2 of step t: r, u, r*h of step t leave; the gates' x of step t + 1 arrives
            R8_SAVE(sv_r, tr, NH, t) R8_SAVE(sv_u, tu, NH, t) R8_SAVE(sv_rh, rhb, LDH, t)
            R8_DEPOSIT_RU()
            R8_REQUEST_RU(min(t + 2, L - 1))
            __syncthreads();
        }
        R8_SAVE(sv_c, tc, NH, L - 1) R8_SAVE(sv_h, hb, LDH, L - 1)
#undef R8_REQUEST_RU
#undef R8_REQUEST_C
#undef R8_DEPOSIT_RU
#undef R8_DEPOSIT_C
#undef R8_SAVE
#undef R8_ER
#undef R8_EC
#undef R8_GRL
        return;
    }
    const int cw = wave % NCW, mt0 = (wave / NCW) * MTW;
    const int col = cw * 64 + lane;
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[sq]), 0, NH * NH * 4, 0x00020000);
    const int vlw = lane_off(lane, KB);
    const int so_r = (2 * cw) * KB * 1024, so_u = (NWT + 2 * cw) * KB * 1024, so_c = (2 * cw) * KB * 1024;     // + kq * 512
    const float* arow[MTW];
    const float* rrow[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        arow[m] = hb + ((mt0 + m) * 4 + (lane & 3)) * LDH;
        rrow[m] = rhb + ((mt0 + m) * 4 + (lane & 3)) * LDH;
    }
    f32x4_t h_own[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) h_own[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // weight streams: phase 1 consumes (q, r), (q, u), (q + 1, r), ... ; phase 2 (q) of the candidate matrix
    float4 ring[RS];
    auto ld1 = [&](int q, int g) { return ldw(rs_wg, vlw, (g == 0 ? so_r : so_u) + q * 512); };
    auto ld2 = [&](int q, int) { return ldw(rs_wc, vlw, so_c + q * 512); };
    auto head1 = [&](int i) { return ld1(i >> 1, i & 1); };
    auto head2 = [&](int i) { return ld2(i, 0); };
#pragma unroll
    for (int i = 0; i < RS; ++i) ring[i] = head1(i);
    __syncthreads();

    for (int t = 0; t < L; ++t) {
        // ---- phase 1: [r | u] pre-activations = x + h W   (x from the stage: the first chain's initial value, MFMA C-in)
        f32x4_t a1[2][MTW][P1];                       // [r | u][tile][chain]
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const float* xr = xs + (mt0 + m) * 4 * NH + col;
#pragma unroll
            for (int p = 0; p < P1; ++p) {
                a1[0][m][p] = p == 0 ? f32x4_t{xr[0], xr[NH], xr[2 * NH], xr[3 * NH]} : f32x4_t{0.f, 0.f, 0.f, 0.f};
                a1[1][m][p] = p == 0 ? f32x4_t{xr[ST], xr[ST + NH], xr[ST + 2 * NH], xr[ST + 3 * NH]} : f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        phase<NQ, 2, MTW, P1>(a1, arow, ring, ld1, head2);
        f32x4_t ug[MTW];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4_t pr = a1[0][m][0], pu = a1[1][m][0];
#pragma unroll
            for (int p = 1; p < P1; ++p) { pr += a1[0][m][p]; pu += a1[1][m][p]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (mt0 + m) * 4 + i;
                const float rg = a.hard_gates ? hard_sig(pr[i]) : gate_sigmoid(pr[i]);
                const float u = a.hard_gates ? hard_sig(pu[i]) : gate_sigmoid(pu[i]);
                ug[m][i] = u;
                rhb[row * LDH + col] = rg * h_own[m][i];
                tr[row * NH + col] = rg;
                tu[row * NH + col] = u;
            }
        }
        __syncthreads();
        // ---- phase 2: candidate pre-activation = x + (r h) Wc, then the state update
        f32x4_t a2[1][MTW][P2];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            const float* xc = xs + 2 * ST + (mt0 + m) * 4 * NH + col;
#pragma unroll
            for (int p = 0; p < P2; ++p)
                a2[0][m][p] = p == 0 ? f32x4_t{xc[0], xc[NH], xc[2 * NH], xc[3 * NH]} : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
        phase<NQ, 1, MTW, P2>(a2, rrow, ring, ld2, head1);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4_t pc = a2[0][m][0];
#pragma unroll
            for (int p = 1; p < P2; ++p) pc += a2[0][m][p];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (mt0 + m) * 4 + i;
                const float c = gate_tanh(pc[i]);
                const float u = (1.0f - s_att[row * LDT + t]) * ug[m][i];
                const float hn = u * h_own[m][i] + (1.0f - u) * c;
                h_own[m][i] = hn;
                hb[row * LDH + col] = hn;
                tc[row * NH + col] = c;
            }
        }
        __syncthreads();
    }
}

inline size_t recur8_fwd_smem(int NH, int L, int rows = 8) { return (size_t)(2 * rows * (NH + 4) + 6 * rows * NH + rows * (L + 1)) * 4; }

// ----------------------------------------------------------------------------------------------------------------- backward
// RecurBwdArgs as k_recur_bwd<NH> takes them; the same per-step equations (header of recur_train.hpp).
// A FIFTH wave does nothing but fetch: while the four matrix waves work on step t it stages step t - 1's rows of the saved
// tensors (R, U, C, H_{t-2}, the per-step upstream gradient) in LDS.  The matrix waves used to load those 32 - 40 values per lane
// at the top of every step and use them at once - a full memory round trip in front of each step's first product (16.7 us per
// AUGRU step against 9.9 us forwards with the same MFMA count), and vmcnt retires in order, so requesting them earlier from the
// same wave would only move the stall into the weight ring.  The loader's waits are its own.
template <int NH, int RWS = 8>
__global__ __launch_bounds__(320) void k_recur8_bwd(RecurBwdArgs a) {
    using namespace r8;
    constexpr int RW = RWS;                                       // (shadows r8::RW; see k_recur8_fwd)
    static_assert(RWS == 8 || (RWS == 4 && NH == 256), "rows per workgroup");
    constexpr int NCW = NH / 64, MTW = (NH == 256) ? RWS / 4 : 1, KB1 = NH / 8, KB2 = 2 * NH / 8, NQ1 = NH / 4, NQ2 = 2 * NH / 4;
    constexpr int LDA = NH + 4, LDG = 2 * NH + 4, PC = 4 / MTW;        // accumulator chains per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tA = reinterpret_cast<float*>(smem);          // [8][LDA]   dAc (A operand of the first product)
    float* tG = tA + RW * LDA;                           // [8][LDG]   [dAg_r | dAg_u] (A operand of the second product)
    float* s_att = tG + RW * LDG;                        // [8][L + 1]
    float* s_red = s_att + RW * (a.L + 1);               // [NCW][RW]  per-wave partial row sums of -dup u
    float* stage = s_red + NCW * RW;                     // [5][8][NH] R, U, C, H_prev, upstream of the step about to be processed
    constexpr int ST = RW * NH, LDR = NH + 8;
    float* s_rv = stage + 5 * ST;                        // [8][LDR]   -dup u of every (row, column): the loader wave sums the rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * RW, sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;
    const bool aug = a.att[sq] != nullptr;
    for (int i = tid; i < RW * L; i += 320) {
        const int r = i / L, t = i - r * L;
        const int gr = min(row0 + r, a.n_rows - 1);
        s_att[r * LDT + t] = aug ? a.att[sq][(size_t)gr * L + t] : 0.f;
    }
    if (wave == 4) {
        // ---- loader wave: float4 k of lane l covers element 4 * (l + 64 k) of an [8][NH] tile
        constexpr int NV = ST / 4 / 64;                  // float4 per lane and array (4 at NH = 128, 8 at NH = 256)
        const float* src[5] = {a.R[sq], a.U[sq], a.C[sq], a.H[sq], a.up_all[sq]};
        // the rows of step tt (H of step tt - 1) are requested a whole step before they are needed and wait in registers: their memory
        // round trip overlaps a full step of the matrix waves, and this wave's barriers do not wait for loads in flight
        float4 v[5][NV];
        auto request = [&](int tt) __attribute__((always_inline)) {
#pragma unroll
            for (int arr = 0; arr < 5; ++arr) {
                const bool zero = src[arr] == nullptr || (arr == 3 && tt == 0);
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int e = (lane + 64 * k) * 4, r = e / NH, c = e - r * NH;
                    const int gr = min(row0 + r, a.n_rows - 1);
                    const int ts = arr == 3 ? tt - 1 : tt;
                    v[arr][k] = zero ? make_float4(0.f, 0.f, 0.f, 0.f)
                                     : *reinterpret_cast<const float4*>(src[arr] + ((size_t)gr * L + ts) * NH + c);
                }
            }
        };
        auto deposit = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int arr = 0; arr < 5; ++arr)
#pragma unroll
                for (int k = 0; k < NV; ++k) *reinterpret_cast<float4*>(stage + arr * ST + (lane + 64 * k) * 4) = v[arr][k];
        };
        request(L - 1);
        deposit();
        if (L > 1) request(L - 2);
        __syncthreads();
        for (int t = L - 1; t >= 0; --t) {
            __syncthreads();                             // the matrix waves have taken step t out of the stage
            if (aug && a.d_score[sq]) {
                // d a_t of a row = sum over ALL hidden columns of -dup u: 64 / RW lanes per row, NH / LPR values each, then log2(LPR)
                // shuffle steps - a fixed order, off the matrix waves' critical path
                constexpr int LPR = 64 / RW;
                const int r = lane / LPR, c0 = lane % LPR;
                float sum = 0.f;
#pragma unroll 8
                for (int j = 0; j < NH / LPR; ++j) sum += s_rv[r * LDR + c0 + LPR * j];
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
                if (c0 == 0 && row0 + r < a.n_rows) a.d_score[sq][(size_t)(row0 + r) * L + t] = sum;
            }
            if (t > 0) deposit();                        // step t - 1, requested during step t + 1
            if (t > 1) request(t - 2);
            __syncthreads();
        }
        return;
    }
    const int cw = wave % NCW, mt0 = (wave / NCW) * MTW;
    const int col = cw * 64 + lane;
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wcT[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wgT[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const int vl1 = lane_off(lane, KB1), vl2 = lane_off(lane, KB2);
    const int so_c = (2 * cw) * KB1 * 1024, so_g = (2 * cw) * KB2 * 1024;       // this wave's 64 output columns; + kq * 512
    size_t so[MTW][4];
    int grow_[MTW][4];
    bool live[MTW][4];
#pragma unroll
    for (int m = 0; m < MTW; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (mt0 + m) * 4 + i;
            const int gr = min(row0 + row, a.n_rows - 1);
            grow_[m][i] = gr;
            so[m][i] = (size_t)gr * L * NH + col;
            live[m][i] = row0 + row < a.n_rows;
        }
    const float* arow[MTW];
    const float* grw[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) {
        arow[m] = tA + ((mt0 + m) * 4 + (lane & 3)) * LDA;
        grw[m] = tG + ((mt0 + m) * 4 + (lane & 3)) * LDG;
    }
    f32x4_t dh[MTW];
#pragma unroll
    for (int m = 0; m < MTW; ++m) dh[m] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float4 ring[RS];
    auto ldc = [&](int q, int) { return ldw(rs_c, vl1, so_c + q * 512); };
    auto ldg = [&](int q, int) { return ldw(rs_g, vl2, so_g + q * 512); };
    auto headc = [&](int i) { return ldc(i, 0); };
    auto headg = [&](int i) { return ldg(i, 0); };
#pragma unroll
    for (int i = 0; i < RS; ++i) ring[i] = headc(i);
    __syncthreads();

    for (int t = L - 1; t >= 0; --t) {
        float dd[MTW][4], up[MTW][4], rg[MTW][4], hp[MTW][4], dagu[MTW][4];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (mt0 + m) * 4 + i;
                const int sx = row * NH + col;
                const float u = stage[ST + sx], c = stage[2 * ST + sx];
                rg[m][i] = stage[sx];
                hp[m][i] = stage[3 * ST + sx];
                const float at = s_att[row * LDT + t];
                float d = dh[m][i] + stage[4 * ST + sx];
                if (t == L - 1 && a.up_last[sq]) d += a.up_last[sq][(size_t)grow_[m][i] * a.ld_up + col];
                dd[m][i] = d;
                up[m][i] = (1.0f - at) * u;
                const float dac = d * (1.0f - up[m][i]) * (1.0f - c * c);
                const float dup = d * (hp[m][i] - c);
                dagu[m][i] = dup * (1.0f - at) * (a.hard ? ((u > 0.f && u < 1.f) ? 0.2f : 0.f) : u * (1.0f - u));
                tA[row * LDA + col] = dac;
                if (live[m][i]) a.dc[sq][((size_t)grow_[m][i] * L + t) * a.ld_c + col] = dac;
                if (aug) s_rv[row * LDR + col] = -dup * u;                 // summed over the columns by the loader wave
            }
        __syncthreads();
        // ---- d(r h) = dAc Wc_h^T
        f32x4_t acc[1][MTW][PC];
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int p = 0; p < PC; ++p) acc[0][m][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        phase<NQ1, 1, MTW, PC>(acc, arow, ring, ldc, headg);
        float dhp[MTW][4];
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4_t s = acc[0][m][0];
#pragma unroll
            for (int p = 1; p < PC; ++p) s += acc[0][m][p];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (mt0 + m) * 4 + i;
                const float drh = s[i];
                const float r = rg[m][i];
                const float dagr = drh * hp[m][i] * (a.hard ? ((r > 0.f && r < 1.f) ? 0.2f : 0.f) : r * (1.0f - r));
                dhp[m][i] = dd[m][i] * up[m][i] + drh * r;
                tG[row * LDG + col] = dagr;
                tG[row * LDG + NH + col] = dagu[m][i];
                if (live[m][i]) {
                    const size_t gi = ((size_t)grow_[m][i] * L + t) * a.ld_g + col;
                    a.dr[sq][gi] = dagr;
                    a.du[sq][gi] = dagu[m][i];
                }
            }
        }
        __syncthreads();
        // ---- dh_{t-1} = dhp + [dAg_r | dAg_u] Wg_h^T   (computed at t = 0 as well, unused there: the ring stays in step)
#pragma unroll
        for (int m = 0; m < MTW; ++m)
#pragma unroll
            for (int p = 0; p < PC; ++p) acc[0][m][p] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        phase<NQ2, 1, MTW, PC>(acc, grw, ring, ldg, headc);
#pragma unroll
        for (int m = 0; m < MTW; ++m) {
            f32x4_t s = acc[0][m][0];
#pragma unroll
            for (int p = 1; p < PC; ++p) s += acc[0][m][p];
#pragma unroll
            for (int i = 0; i < 4; ++i) dh[m][i] = dhp[m][i] + (t > 0 ? s[i] : 0.f);
        }
        // (tA is read before this step's second barrier and written after the NEXT step's loads; tG and s_red are read before
        // the next step's first barrier and written after it / after this step's second barrier: no further barrier needed)
    }
}

inline size_t recur8_bwd_smem(int NH, int L, int rows = 8) {
    return (size_t)(rows * (NH + 4) + rows * (2 * NH + 4) + rows * (L + 1) + (NH / 64) * rows + 5 * rows * NH + rows * (NH + 8)) * 4;
}

}  // namespace rl4rs
