// k_augru_x: the AUGRU recurrence of the DIEN scorer in fp16x2 form, second generation (included by dien.hip).
//
// Same arithmetic as k_augru_h16 (rl4rs/nets/utils.py:120-124 via deepctr VecAttGRUCell; operands split into fp16 hi + lo,
// every product as W_hi*h_hi + W_lo*h_hi + W_hi*h_lo on v_mfma_f32_32x32x16_f16, fp32 accumulation), different machine
// mapping.  What bounded k_augru_h16 (profiles/r01g_pmc.md: matrix pipe 47-52 % busy): 8 waves per CU pulling 786 KB of weight
// fragments per recurrence step through the 64 B/clk L1 return path, the two waves of a SIMD in lock step so that no
// epilogue hid behind the other's MFMAs, a ~6K-cycle exposed candidate epilogue and 384 dword-per-lane projection loads per
// step.  This kernel:
//
//   * ONE wave per SIMD (4 waves, 256 threads, up to 512 registers each): wave w owns hidden columns [64w, 64w+64) of all three
//     gates as two 32-column tiles (ct 0 / 1).  The freed registers hold weight fragments for the whole kernel (NRES items of
//     a step in VGPR/AGPR, NLDS more in the LDS the operand planes leave free): less than 60 % of the 96 weight items of a step
//     still stream from L2, through a RING-deep register ring.
//   * TRANSPOSED tiles: the MFMA computes (W^T h^T), A = weight fragment, B = state fragment, so a lane owns ONE batch row and
//     16 hidden columns in runs of 4.  Per lane: the attention score is one scalar per step, the cached input projections
//     arrive as 16-byte loads (4 columns of one row), the new state leaves as packed 8-byte LDS writes (4 fp16 values) and the
//     final state as 16-byte stores.  4x fewer memory / LDS instructions than the row-per-register layout.
//   * every epilogue in the MFMA shadow.  A step is a fixed sequence of 96 weight items (gate, column tile, k-block):
//         [ 0,16)  R  k-blocks 2,3 mod 4  ("late" columns: those of the ct = 1 tiles)
//         [16,48)  U  all k-blocks                        || reset gate r = sigmoid(acc_r), r*h -> fp16 planes
//         -- barrier (r*h planes complete)
//         [48,65)  C  tile 0  (+ k-block 0 of tile 1)     || update gate u = (1 - a_t) sigmoid(acc_u)
//         [65,80)  C  tile 1                              || candidate + blend of tile 0 -> new state planes, "early" columns
//         -- barrier (early columns of h' complete)
//         [80,96)  R of the NEXT step, k-blocks 0,1 mod 4 || candidate + blend of tile 1 -> "late" columns
//         -- barrier
//     i.e. the next step's reset-gate product starts on the half of the new state that is already written while the other half
//     is still being blended; nothing but the three barriers is exposed.
//   * ONE issue window per step for the cached input projections (they come from HBM and vector-memory loads return in
//     order: a slow load in front of the weight ring stalls every later weight wait).  Right after item 49 the wave requests
//     all projections of step t+1 - r-gate rows straight into the (retired) r accumulators, u- and c-gate rows into staging
//     registers that become the C operand of the first MFMA of their chain - and the NRES + NLDS items that follow are exactly
//     the resident ones: ~3.8K cycles without a vector-memory wait.
//   * out-of-range rows (|h| >= 6e4: the fp16 planes cannot carry them; or NaN) are POISONED: the whole output row becomes NaN,
//     so the observation / click probability / reward of that env is NaN on the device without any host synchronisation, and
//     the sticky status bit is raised as before (rl4rs_dien_status).
#pragma once

namespace rl4rs {
namespace xk {
constexpr int NI = 96;
constexpr int kb_late(int j) { return (j >> 1) * 4 + 2 + (j & 1); }       // j = 0..7 -> 2,3,6,7,10,11,14,15
constexpr int kb_early(int j) { return (j >> 1) * 4 + (j & 1); }          //             0,1,4,5, 8, 9,12,13
constexpr int gate(int i) { return i < 16 ? 0 : (i < 48 ? 1 : (i < 80 ? 2 : 0)); }
constexpr int ct(int i) { return i < 48 ? (i & 1) : (i == 48 ? 0 : (i == 49 ? 1 : (i < 65 ? 0 : (i < 80 ? 1 : (i & 1))))); }
constexpr int kb(int i) {
    return i < 16 ? kb_late(i >> 1) : (i < 48 ? ((i - 16) >> 1) : (i < 50 ? 0 : (i < 65 ? i - 49 : (i < 80 ? i - 64 : kb_early((i - 80) >> 1)))));
}
constexpr bool starts_group(int i) { return (i < 48 || i >= 80) ? ((i & 1) == 0) : true; }     // state-fragment groups (64 per step)
constexpr int group_parity(int i) {
    int g = 0;
    for (int k = 0; k <= i; ++k) g += starts_group(k) ? 1 : 0;
    return (g - 1) & 1;
}
constexpr bool from_rh(int i) { return i >= 48 && i < 80; }                 // B operand: r*h planes (candidate) or h planes
constexpr bool after_barrier(int i) { return i == 0 || i == 48 || i == 80; }
constexpr int FIRST_RES = 50;                                               // first item after the projection issue window
constexpr int rel(int i) { return (i - FIRST_RES + NI) % NI; }
}  // namespace xk

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

#ifdef RL4RS_X_TRACE       // s_memtime marks of workgroup (0,0), steps 8..11: [wave][step][mark]
#define RL4RS_XT(k) do { if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && t >= 8 && t < 12) \
        a.trace[(wave * 4 + (t - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RL4RS_XT(k) do { } while (0)
#endif

template <int MT, int NRES, int NLDS, int RING>
__global__ __launch_bounds__(256) void k_augru_x(RecurArgs a) {
    using namespace xk;
    constexpr int NH = 256, KB = 16, LDP = NH + 8, MR = MT * 32;
    constexpr int NS = NI - NRES - NLDS, LA = RING - 1;
    static_assert(NS > 0 && NS % RING == 0 && RING >= 2 && LA < NS, "weight ring");
    static_assert(NRES + NLDS <= NI - 50, "the resident items follow the issue window inside one step");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16* hp_hi = reinterpret_cast<_Float16*>(smem);     // [MR][LDP] each: h planes, r*h planes
    _Float16* hp_lo = hp_hi + MR * LDP;
    _Float16* rp_hi = hp_lo + MR * LDP;
    _Float16* rp_lo = rp_hi + MR * LDP;
    float* s_att = reinterpret_cast<float*>(rp_lo + MR * LDP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * MR;
    const int sq = blockIdx.y;
    const int L = a.L, LDT = L + 1;
    const int xld4 = (int)a.xld * 4;
    // packed fp16 planes: [ntile][KB][plane hi/lo][64 lanes][8 halfs] -> 1 KB per (ntile, kb, plane) (pack_frag_h16)
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xbase[sq]), 0, (int)a.xbytes, 0x00020000);
    const int vl16 = lane * 16;

    for (int i = tid; i < 2 * MR * LDP / 2; i += 256) reinterpret_cast<uint32_t*>(hp_hi)[i] = 0u;     // hi and lo planes of h = 0
    for (int i = tid; i < MR * L; i += 256) {
        const int r = i / L, t = i - r * L;
        const int gr = min(row0 + r, a.n_rows - 1);
        s_att[r * LDT + t] = a.att[(size_t)sq * a.att_stride + (size_t)gr * L + t];
    }
    uint32_t* s_bad = reinterpret_cast<uint32_t*>(s_att + MR * LDT);       // [MR] row poison flags
    if (tid < MR) s_bad[tid] = 0u;
    char* lds_w = reinterpret_cast<char*>(s_bad + MR) + (size_t)wave * NLDS * 2048;   // this wave's LDS-resident items
    // byte offset of this lane's row(s) in the projection cache at t = 0, plus the lane's 16-byte column sub-offset
    int xrow[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int gr = min(row0 + m * 32 + li, a.n_rows - 1);
        xrow[m] = (int)((uint32_t)a.slots[(size_t)sq * a.slots_stride + gr / a.group] * (uint32_t)L * (uint32_t)xld4) + half * 16;
    }
    const int hoff = li * LDP + half * 8;                 // B-fragment offset (halfs) of this lane inside a row tile
    // weight item i of a step -> buffer + scalar byte offset (+ plane * 1024)
    int sb_r = (2 * wave) * KB * 2048, sb_u = (8 + 2 * wave) * KB * 2048, sb_c = (2 * wave) * KB * 2048;
    auto wload = [&](int i, half8_t& hi, half8_t& lo) {
        const int g = gate(i), off = (ct(i) * KB + kb(i)) * 2048;
        if (g == 0) { hi = buf_load_h8(rs_wg, vl16, sb_r + off); lo = buf_load_h8(rs_wg, vl16, sb_r + off + 1024); }
        else if (g == 1) { hi = buf_load_h8(rs_wg, vl16, sb_u + off); lo = buf_load_h8(rs_wg, vl16, sb_u + off + 1024); }
        else { hi = buf_load_h8(rs_wc, vl16, sb_c + off); lo = buf_load_h8(rs_wc, vl16, sb_c + off + 1024); }
    };
    // item index of streamed-sequence position js (the streamed items are cyclically contiguous in execution order)
    auto streamed_item = [](int js) { return (FIRST_RES + NRES + NLDS + js) % NI; };
    // cached input projection of gate block `blk` (0 r, 1 u, 2 c), step t, for tile m / column tile c -> 16 floats, 4 loads of 16 B
    auto load_x = [&](f32x16& dst, int m, int c, int t, int blk) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = buf_load4(rs_x, xrow[m], t * xld4 + (a.xoff + blk * NH + (2 * wave + c) * 32 + 8 * q) * 4);
            dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    };

    f32x16 acc_r[MT][2], acc_u[MT][2], acc_c[MT][2], h_own[MT][2], xs_u[MT][2], xs_c[MT][2];
    half8_t res_h[NRES > 0 ? NRES : 1], res_l[NRES > 0 ? NRES : 1];      // items FIRST_RES .. FIRST_RES + NRES - 1
    half8_t ring_h[RING], ring_l[RING], lw_h[2], lw_l[2];
    half8_t bh[2][MT], bl[2][MT];
    float amax[MT], oma[MT];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NRES; ++i) wload((FIRST_RES + i) % NI, res_h[i], res_l[i]);
#pragma unroll
    for (int i = 0; i < NLDS; ++i) {
        half8_t hi, lo;
        wload((FIRST_RES + NRES + i) % NI, hi, lo);
        *reinterpret_cast<half8_t*>(lds_w + i * 2048 + vl16) = hi;
        *reinterpret_cast<half8_t*>(lds_w + i * 2048 + 1024 + vl16) = lo;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        amax[m] = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) h_own[m][c][r] = 0.f;
            load_x(acc_r[m][c], m, c, 0, 0);          // h = 0: the R products of step 0 vanish, acc_r = x_r(0)
            load_x(xs_u[m][c], m, c, 0, 1);
            load_x(xs_c[m][c], m, c, 0, 2);
        }
    }
#pragma unroll
    for (int js = 0; js < LA; ++js) wload(streamed_item(js), ring_h[js % RING], ring_l[js % RING]);

    auto hfrag = [&](int buf, int i) {                 // state fragments (B operand) of item i's k-block, all row tiles
        const _Float16* ph = from_rh(i) ? rp_hi : hp_hi;
        const _Float16* pl = from_rh(i) ? rp_lo : hp_lo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bh[buf][m] = *reinterpret_cast<const half8_t*>(ph + m * 32 * LDP + hoff + kb(i) * 16);
            bl[buf][m] = *reinterpret_cast<const half8_t*>(pl + m * 32 * LDP + hoff + kb(i) * 16);
        }
    };
    // four consecutive hidden columns of a lane's row -> the fp16 hi / lo planes (8-byte LDS writes)
    auto plane_store = [&](_Float16* p_hi, _Float16* p_lo, int m, int c, int q, const float* v) {
        half4_t vh, vl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const _Float16 h = (_Float16)v[j];
            vh[j] = h;
            vl[j] = (_Float16)(v[j] - (float)h);
        }
        const int o = (m * 32 + li) * LDP + wave * 64 + c * 32 + 8 * q + 4 * half;
        *reinterpret_cast<half4_t*>(p_hi + o) = vh;
        *reinterpret_cast<half4_t*>(p_lo + o) = vl;
    };

#pragma unroll 1
    for (int t = 0; t < L; ++t) {
        asm volatile("" : "+s"(sb_r), "+s"(sb_u), "+s"(sb_c));     // keep the 96 per-item scalar offsets out of SGPR-hoisting
        RL4RS_XT(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) oma[m] = 1.0f - s_att[(m * 32 + li) * LDT + t];
        hfrag(group_parity(0), 0);
        float rq[MT][4], cq[MT][4];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i == 16) RL4RS_XT(1);
            if (i == 48 || i == 80) {
                if (i == 48) RL4RS_XT(2); else RL4RS_XT(4);
                __syncthreads();                       // r*h planes complete / early half of the new state complete
                if (i == 48) RL4RS_XT(3); else RL4RS_XT(5);
                hfrag(group_parity(i), i);
            }
            const int g = gate(i), c = ct(i), cur = group_parity(i);
            // ---- fetch ahead: state fragments of the next group, LDS-resident weights one item ahead, streamed weights LA ahead
            if (i + 1 < NI && starts_group(i + 1) && !after_barrier(i + 1)) hfrag(group_parity(i + 1), i + 1);
            if (NLDS > 0) {
                const int rn = rel((i + 1) % NI);
                if (rn >= NRES && rn < NRES + NLDS) {
                    lw_h[(rn - NRES) & 1] = *reinterpret_cast<const half8_t*>(lds_w + (rn - NRES) * 2048 + vl16);
                    lw_l[(rn - NRES) & 1] = *reinterpret_cast<const half8_t*>(lds_w + (rn - NRES) * 2048 + 1024 + vl16);
                }
            }
            const int ri = rel(i);
            if (ri >= NRES + NLDS) {
                const int js = ri - NRES - NLDS;
                wload(streamed_item((js + LA) % NS), ring_h[(js + LA) % RING], ring_l[(js + LA) % RING]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const half8_t wh = ri < NRES ? res_h[ri < NRES ? ri : 0] : (ri < NRES + NLDS ? lw_h[(ri - NRES) & 1] : ring_h[(ri - NRES - NLDS) % RING]);
            const half8_t wl = ri < NRES ? res_l[ri < NRES ? ri : 0] : (ri < NRES + NLDS ? lw_l[(ri - NRES) & 1] : ring_l[(ri - NRES - NLDS) % RING]);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                f32x16& acc = g == 0 ? acc_r[m][c] : (g == 1 ? acc_u[m][c] : acc_c[m][c]);
                // the cached input projection enters as the C operand of the first MFMA of a chain
                const f32x16 cin = (i == 16 || i == 17) ? xs_u[m][c] : ((i == 48 || i == 49) ? xs_c[m][c] : acc);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[cur][m], cin, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[cur][m], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[cur][m], acc, 0, 0, 0);
            }
            // ---- the epilogue work that rides in this item's MFMA shadow
            int nvalu = 0;
            if (i >= 16 && i < 48) {                   // reset gate: one element per item, packed per 4 columns
                const int e = i - 16, ec = e >> 4, r = e & 15;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    rq[m][r & 3] = gate_sigmoid(acc_r[m][ec][r]) * h_own[m][ec][r];
                    if ((r & 3) == 3) plane_store(rp_hi, rp_lo, m, ec, r >> 2, rq[m]);
                }
                nvalu = 3;
            } else if (i >= 48 && i < 64) {            // update gate: two elements per item
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int e = 2 * (i - 48) + k, ec = e >> 4, r = e & 15;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        float pre = acc_u[m][ec][r];
                        asm volatile("" : "+v"(pre));      // keeps this element's chain inside this item
                        acc_u[m][ec][r] = oma[m] * gate_sigmoid(pre);
                    }
                }
                nvalu = 4;
            } else if (i >= 65) {                      // candidate + blend: tile 0 during [65,80), tile 1 during [80,96)
                const int ec = i < 80 ? 0 : 1;
                const int e0 = i < 80 ? i - 65 : i - 80, ne = (i == 79) ? 2 : 1;
#pragma unroll
                for (int k = 0; k < ne; ++k) {
                    const int r = e0 + k;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float cnd = gate_tanh(acc_c[m][ec][r]);
                        const float hn = __builtin_fmaf(acc_u[m][ec][r], h_own[m][ec][r] - cnd, cnd);     // u h + (1-u) c
                        amax[m] = fmaxf(amax[m], fabsf(hn));
                        h_own[m][ec][r] = hn;
                        cq[m][r & 3] = hn;
                        if ((r & 3) == 3) plane_store(hp_hi, hp_lo, m, ec, r >> 2, cq[m]);
                    }
                }
                nvalu = 4;
            }
            if (i == 49 && t + 1 < L) {
                // ---- the ONE projection issue window of the step (the resident items follow: no vector-memory wait for ~3.8K cycles)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int cc = 0; cc < 2; ++cc) {
                        load_x(acc_r[m][cc], m, cc, t + 1, 0);
                        load_x(xs_u[m][cc], m, cc, t + 1, 1);
                        load_x(xs_c[m][cc], m, cc, t + 1, 2);
                    }
            }
            // a wave issues in order: without this the VALU chunk only overlaps the last MFMA of the item
            if (nvalu == 3) {
#pragma unroll
                for (int q = 0; q < 3 * MT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);         // VALU in its shadow
                }
            } else if (nvalu == 4) {
#pragma unroll
                for (int q = 0; q < 3 * MT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        RL4RS_XT(6);
        __syncthreads();                               // late half of the new state complete
        RL4RS_XT(7);
    }
    // ---- poison rows that left the fp16 range (or went NaN) and write the final state (16-byte stores)
    bool any_bad = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        bool bad = !(amax[m] < 6.0e4f);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) bad |= !(fabsf(h_own[m][c][r]) < 6.0e4f);
        if (bad) atomicOr(&s_bad[m * 32 + li], 1u);
        any_bad |= bad;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = row0 + m * 32 + li;
        const bool poison = s_bad[m * 32 + li] != 0u;
        if (row < a.n_rows) {
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4(h_own[m][c][4 * q], h_own[m][c][4 * q + 1], h_own[m][c][4 * q + 2], h_own[m][c][4 * q + 3]);
                    if (poison) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                    *reinterpret_cast<float4*>(a.out + (int64_t)row * a.out_ld + a.out_off + sq * a.out_seq_off + wave * 64 + c * 32 + 8 * q + 4 * half) = v;
                }
        }
    }
    if (any_bad && a.range_flag) atomicOr(a.range_flag, 1);
}

static size_t augru_x_smem(int mt, int L, int nlds) {
    return (size_t)4 * mt * 32 * (256 + 8) * 2 + (size_t)(mt * 32 * (L + 1) + mt * 32) * 4 + (size_t)4 * nlds * 2048;
}

}  // namespace rl4rs
