// k_augru_x: the AUGRU recurrence of the DIEN scorer in fp16x2 form, second generation (included by dien.hip).
//
// Same arithmetic as k_augru_h16 (rl4rs/nets/utils.py:120-124 via deepctr VecAttGRUCell; operands split into fp16 hi + lo,
// every product as W_hi*h_hi + W_lo*h_hi + W_hi*h_lo on v_mfma_f32_32x32x16_f16, fp32 accumulation), different machine
// mapping.  What bounded k_augru_h16 (profiles/r01g_pmc.md: matrix pipe 47-52 % busy): the two waves of a SIMD moved through
// the step in lock step, so neither's epilogue hid behind the other's MFMAs (a ~6K-cycle exposed candidate epilogue per
// ~20K-cycle step), 786 KB of weight fragments per step through the 64 B/clk L1 return path, and 384 dword-per-lane
// projection loads per step whose HBM latency sat in the in-order vector-memory queue in front of the weight ring.
//
//   * TRANSPOSED tiles: the MFMA computes (W^T h^T) - A = weight fragment, B = state fragment - so a lane owns ONE batch row and
//     16 hidden columns in runs of 4: the attention score is one scalar per lane and step, new state leaves as packed 8-byte
//     LDS writes (4 fp16 values), the final state as 16-byte stores.
//   * ROLES: 8 waves, wave w owns hidden columns [32w, 32w+32) of all three gates.  Waves 0-3 ("early": k-blocks 0..7 of the
//     state) and 4-7 ("late") land pairwise on the same SIMD (wave k and k+4) and run the candidate phase out of step:
//         all    R (late k-blocks) , U            || reset gate r = sigmoid(acc_r), r*h -> fp16 planes
//         -- barrier 1 (r*h planes complete)
//         early  C || update gate, then candidate + blend -> early columns of h'    (VALU while its partner runs C)
//         late   update gate, then C                                                 (MFMA while its partner blends)
//         -- barrier a (early half of h' complete)
//         all    R of the NEXT step on the early k-blocks;  late: || candidate + blend -> late columns of h'
//         -- barrier b
//     so whenever one wave of a SIMD is in an epilogue its partner keeps the matrix pipe busy, and the next step's reset-gate
//     product starts on the half of the new state that is already written.
//   * cached input projections through LDS-DMA: one issue window per step (right in front of the register-resident weight
//     items, i.e. in front of a stretch without vector-memory waits) requests the three gates' rows of step t+1 with
//     buffer_load ... lds: 4 instructions per gate, each one 8 rows x 128 contiguous bytes (8 cache lines instead of the 32 a
//     row-per-lane register load touches), no staging registers; a chunk rotation in the SOURCE address makes the lane-linear
//     LDS image conflict-free for the row-per-lane reads that feed the accumulators (MFMA C-in).
//   * LDS (all 160 KB): state planes in slab order [k-block][k-half][row][8 halfs] - fragment reads and 8-byte writes are
//     conflict-free without padding (64 KB) - and 3 x 4 KB of projection staging per wave (96 KB).
//   * out-of-range rows (|h| >= 6e4: the fp16 planes cannot carry them; or NaN) are POISONED: the whole output row becomes NaN,
//     so the observation / click probability / reward of that env is NaN on the device without any host synchronisation, and
//     the sticky status bit is raised as before (rl4rs_dien_status).
#pragma once

namespace rl4rs {
namespace xk {
constexpr int NI = 48;                                       // weight items per wave and step
constexpr int gate(int i) { return i < 8 ? 0 : (i < 24 ? 1 : (i < 40 ? 2 : 0)); }
constexpr int kb(int i) { return i < 8 ? 8 + i : (i < 24 ? i - 8 : (i < 40 ? i - 24 : i - 40)); }
constexpr bool from_rh(int i) { return i >= 24 && i < 40; }  // B operand: r*h planes (candidate) or h planes
constexpr bool after_barrier(int i) { return i == 0 || i == 24 || i == 40; }
#ifndef RL4RS_X_DMA_AUX
#define RL4RS_X_DMA_AUX 2        // cache-policy bits of the projection DMA (1 sc0, 2 nt, 16 sc1).  nt: a projection row is used once (by the
                                // envs of one history, which run side by side), so it should not push the weight fragments - re-read 64
                                // times per launch - out of L2.  Same-box A/B, 5 alternating pairs: 11.41 -> 11.32 ms per episode-batch
                                // (+0.8 %), SeqSlate T=32 38.55 -> 38.12 ms (+1.1 %); sc0 flat, nt|sc0 like nt.  0 = round 2 .. 5's default
#endif
#ifndef RL4RS_X_AB
#define RL4RS_X_AB 0            // timing ablations (results are WRONG when non-zero): 1 no epilogue math, 2 no weight streaming,
#endif                          // 4 no projection DMA / reads, 8 no state-fragment reads, 16 no barriers
#ifndef RL4RS_X_SGB
#define RL4RS_X_SGB 0           // VALU instructions pinned behind each MFMA of an item that carries epilogue work (0 = compiler's order)
#endif
#ifndef RL4RS_X_PRIO
#define RL4RS_X_PRIO 0
#endif
#ifndef RL4RS_X_SPREAD
#define RL4RS_X_SPREAD 0
#endif
// where in a step the staged projections of the update gate / the candidate are read into their accumulators (MFMA C-in): right
// in front of the gate's first item (8 / 24) or some items earlier, so that the LDS round trip is not in front of that item's MFMAs
// (both accumulators are free from item 0; the staging is per wave and was filled one step ago; must stay below the DMA window)
#ifndef RL4RS_X_LATE_UPD_SHADOW
#define RL4RS_X_LATE_UPD_SHADOW 0   // 1: the late waves also compute their update gate in the shadow of their first C items (like the early
#endif                              // waves) instead of as a VALU-only stretch in front of them (round 4 A/B)
#ifndef RL4RS_X_XU_AT
#define RL4RS_X_XU_AT 8
#endif
#ifndef RL4RS_X_XC_AT
#define RL4RS_X_XC_AT 24
#endif
#ifndef RL4RS_X_SPLITPAIR
#define RL4RS_X_SPLITPAIR 1     // plane_store through split_h16_pair (v_cvt_pk_f16_f32 + v_fma_mix_f32: 2 VALU per element instead of
#endif                          // ~3; same roundings, bit-identical planes; same-box A/B: 0.8595 -> 0.8503 ms per launch) - 0: the C++ casts
#ifndef RL4RS_X_AMAX
#define RL4RS_X_AMAX 0          // 1: track max |h| over the steps for the range check (0: the final state alone decides, see the epilogue)
#endif
// which items of a step keep their weight fragments in registers: a contiguous stretch behind the projection issue window
// (RL4RS_X_SPREAD = 0) or every (48 / NRES)-th item (1: uniform load on the L1 return path)
// (RL4RS_X_RESMASK: an explicit 48-bit item mask for experiments, 32-row form only, with RL4RS_X_WINDOW = the item the
// projection requests are issued in front of)
template <int NRES> constexpr bool is_res(int i) {
#ifdef RL4RS_X_RESMASK
    if (NRES == RL4RS_X_NRES) return ((RL4RS_X_RESMASK >> i) & 1ull) != 0;
#endif
    return RL4RS_X_SPREAD ? ((i % (NI / NRES)) == (NI / NRES) - 1 && i / (NI / NRES) < NRES) : (i >= 40 - NRES && i < 40);
}
// compile-time tables of a step's schedule: index of a resident item in the register file, position of a streamed item in
// the (cyclic) streamed sequence that starts at the first streamed item >= 40, and its inverse
struct Sched { int res_idx[NI]; int js_of[NI]; int item_of[NI]; int js_first; };
template <int NRES> constexpr Sched make_sched() {
    Sched s = {};
    int n = 0;
    for (int i = 0; i < NI; ++i) { s.res_idx[i] = n; n += is_res<NRES>(i) ? 1 : 0; }
    int k = 40;
    while (is_res<NRES>(k % NI)) ++k;
    int js = 0;
    for (int c = 0; c < NI; ++c) {
        const int i = (k + c) % NI;
        s.js_of[i] = js;
        if (!is_res<NRES>(i)) { s.item_of[js] = i; ++js; }
    }
    int f = 0;
    while (is_res<NRES>(f)) ++f;
    s.js_first = s.js_of[f];                               // step 0 enters the sequence at its first streamed item >= 0
    return s;
}
}  // namespace xk

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#ifdef RL4RS_X_TRACE       // s_memtime marks of workgroup (0,0), steps 8..11: [wave][step][mark]
#define RL4RS_XT(k) do { if (a.trace && blockIdx.x == 0 && blockIdx.y == 0 && lane == 0 && t >= 8 && t < 12) \
        a.trace[(wave * 4 + (t - 8)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RL4RS_XT(k) do { } while (0)
#endif

// MT = 1: 32-row workgroups, per-row projection staging (any row -> slot map).  MT = 2: 64-row workgroups for launches whose
// rows come in groups of 8 consecutive rows per cache slot (the reward forward: 8 complete-state rows per env): every weight
// fragment feeds two row tiles (half the weight bytes per row through the L1 return path), and the 8 distinct projection rows
// of the workgroup are staged once (1 KB per gate and wave, one DMA instruction).
// PAD (instantiated for MT = 1): steps below a row's count of leading zero ids read the projections of the pad slot (RecurArgs::lead / pad_slot).
template <int MT, int NRES, int RING, bool PAD = false>
__global__ __launch_bounds__(512) void k_augru_x(RecurArgs a) {
    using namespace xk;
    constexpr int NH = 256, KB = 16, PLANE = 32 * NH * 2;          // bytes per plane (16 KB)
#if defined(RL4RS_X_RESMASK) && defined(RL4RS_X_WINDOW)
    constexpr int NS = NI - NRES, LA = RING - 1, WINDOW = (MT == 1) ? RL4RS_X_WINDOW : 40 - NRES;
#else
    constexpr int NS = NI - NRES, LA = RING - 1, WINDOW = 40 - NRES;     // WINDOW: the item the projection requests are issued in front of
#endif
    constexpr Sched SC = make_sched<NRES>();
    static_assert(NS > 0 && NS % RING == 0 && RING >= 2 && NRES >= 1 && NRES <= 14 && (!RL4RS_X_SPREAD || NI % NRES == 0), "weight ring / resident items");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // planes in slab order [kb 16][k-half 2][row 32][8 halfs]; tile m's four planes (h hi/lo, r*h hi/lo) at m * 4 * PLANE
    char* hp_hi = smem;
    char* hp_lo = smem + PLANE;
    char* rp_hi = smem + 2 * PLANE;
    char* rp_lo = smem + 3 * PLANE;
    constexpr int TILE = 4 * PLANE, STG = MT == 1 ? 3 * 4096 : 3 * 1024;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5, li = lane & 31;
    const int row0 = blockIdx.x * 32 * MT;
    const int sq = blockIdx.y;
    const int L = a.L;
    const int xld4 = (int)a.xld * 4;
    char* stage = smem + MT * TILE + wave * STG;                   // this wave's projection staging: gate g at + g * STG / 3
    // packed fp16 planes: [ntile][KB][plane hi/lo][64 lanes][8 halfs] -> 1 KB per (ntile, kb, plane) (pack_frag_h16)
    const __amdgpu_buffer_rsrc_t rs_wg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wg[sq]), 0, 2 * NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_wc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wc[sq]), 0, NH * NH * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.xbase[sq]), 0, (int)a.xbytes, 0x00020000);
    const int vl16 = lane * 16;

#pragma unroll
    for (int m = 0; m < MT; ++m)
        for (int i = tid; i < 2 * PLANE / 16; i += 512) reinterpret_cast<uint4*>(hp_hi + m * TILE)[i] = make_uint4(0u, 0u, 0u, 0u);     // h = 0
    // ---- projection staging geometry.  DMA instruction j (0..3) of a gate covers rows 8j .. 8j+7: lane l fetches, for row
    // r = 8j + l/8, the 16-byte chunk c = (l%8 - r/2) mod 8 of the wave's 128 bytes of that row; it lands at slot + r*128 +
    // (l%8)*16.  The reader (row li, column run q of half `half`: chunk 2q + half) finds it at position (2q + half + li/2) mod 8.
    // processing order: tile position p works on batch row phys(p) - with a row order (rl4rs_dien_set_row_order: env groups sorted
    // by their history's cache slot) the rows of a tile, and of tiles that run at the same time, share projection rows in L2
    auto phys = [&](int p) {
        p = min(p, a.n_rows - 1);
        return a.order ? a.order[p / a.group] * a.group + p % a.group : p;
    };
    int dma_off[4];
    int pad_delta[4] = {0, 0, 0, 0}, lead_j[4] = {0, 0, 0, 0};     // PAD: (pad slot - own slot) in bytes, leading zero ids of the lane's row
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        // MT = 1: row r = 8j + l/8, rotated chunk.  MT = 2 (only j = 0 is used): lane l fetches chunk l%8 of distinct row d = l/8,
        // i.e. of batch row row0 + 8d (rows 8d .. 8d+7 share its cache slot)
        const int r = MT == 1 ? 8 * j + (lane >> 3) : 8 * (lane >> 3);
        const int gr = phys(row0 + r);
        const int c = MT == 1 ? (((lane & 7) - (r >> 1)) & 7) : (lane & 7);
        const uint32_t slot = (uint32_t)a.slots[(size_t)sq * a.slots_stride + gr / a.group];
        dma_off[j] = (int)(slot * (uint32_t)L * (uint32_t)xld4) + c * 16;
        if constexpr (PAD) {
            lead_j[j] = a.lead[sq][slot];
            pad_delta[j] = (int)(((uint32_t)a.pad_slot - slot) * (uint32_t)L * (uint32_t)xld4);
        }
    }
    const float* att_row[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        att_row[m] = a.att + (size_t)sq * a.att_stride + (size_t)phys(row0 + m * 32 + li) * L;
    const int xs_base = wave * 128 + a.xoff * 4;                  // byte offset of the wave's 32 columns inside a gate block
    auto x_dma = [&](int t) {                                      // the three gates' rows of step t -> staging
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int j = 0; j < (MT == 1 ? 4 : 1); ++j) {
                int voff = dma_off[j];
                if constexpr (PAD) voff += t < lead_j[j] ? pad_delta[j] : 0;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(stage + g * (STG / 3) + j * 1024), 16, voff,
                                                         t * xld4 + xs_base + g * NH * 4, 0, RL4RS_X_DMA_AUX);
            }
    };
    auto x_read = [&](f32x16& dst, int g, int m) {                 // staged projection rows -> accumulator (MFMA C-in)
        const int rot = half + (li >> 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const char* src = MT == 1 ? stage + g * 4096 + li * 128 + (((2 * q + rot) & 7) << 4)
                                      : stage + g * 1024 + (4 * m + (li >> 3)) * 128 + ((2 * q + half) << 4);
            const float4 v = *reinterpret_cast<const float4*>(src);
            dst[4 * q + 0] = v.x; dst[4 * q + 1] = v.y; dst[4 * q + 2] = v.z; dst[4 * q + 3] = v.w;
        }
    };
    // weight item i of a step -> buffer + scalar byte offset (the lo plane sits 1 KB behind the hi plane: immediate offset)
    int sb_r = wave * KB * 2048, sb_u = (8 + wave) * KB * 2048, sb_c = wave * KB * 2048;
    auto wload = [&](int i, half8_t& hi, half8_t& lo) {
        const int g = gate(i), off = (RL4RS_X_AB & 32) ? 0 : kb(i) * 2048;       // 32: every streamed load hits the same (L1-resident) fragment
        if (g == 0) { hi = buf_load_h8(rs_wg, vl16, sb_r + off); lo = buf_load_h8(rs_wg, vl16 + 1024, sb_r + off); }
        else if (g == 1) { hi = buf_load_h8(rs_wg, vl16, sb_u + off); lo = buf_load_h8(rs_wg, vl16 + 1024, sb_u + off); }
        else { hi = buf_load_h8(rs_wc, vl16, sb_c + off); lo = buf_load_h8(rs_wc, vl16 + 1024, sb_c + off); }
    };
    const int foff = half * 512 + li * 16;                         // this lane's fragment inside a (plane, k-block) slab

    f32x16 acc_r[MT], acc_u[MT], acc_c[MT], h_own[MT];
    half8_t res_h[NRES], res_l[NRES], ring_h[RING], ring_l[RING];
    half8_t bh[2][MT], bl[2][MT];
    float amax[MT], att_cur[MT], att_next[MT], oma[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        amax[m] = 0.f;
        att_next[m] = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) h_own[m][r] = 0.f;
    }
    x_dma(0);
#pragma unroll
    for (int i = 0; i < NI; ++i)
        if (is_res<NRES>(i)) wload(i, res_h[SC.res_idx[i]], res_l[SC.res_idx[i]]);
#pragma unroll
    for (int k = 0; k < LA; ++k) {
        const int js = (SC.js_first + k) % NS;
        wload(SC.item_of[js], ring_h[js % RING], ring_l[js % RING]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        att_cur[m] = att_row[m][0];
        x_read(acc_r[m], 0, m);                                    // h = 0: the R products of step 0 vanish, acc_r = x_r(0)
    }
    __syncthreads();

    auto hfrag = [&](int buf, int i) {                             // state fragments (B operand) of item i's k-block
        const char* ph = from_rh(i) ? rp_hi : hp_hi;
        const char* pl = from_rh(i) ? rp_lo : hp_lo;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            bh[buf][m] = *reinterpret_cast<const half8_t*>(ph + m * TILE + kb(i) * 1024 + foff);
            bl[buf][m] = *reinterpret_cast<const half8_t*>(pl + m * TILE + kb(i) * 1024 + foff);
        }
    };
    // four consecutive hidden columns (run q) of this lane's row -> the fp16 hi / lo planes (8-byte LDS writes)
    auto plane_store = [&](char* p_hi, char* p_lo, int m, int q, const float* v) {
        half4_t vh, vl;
#if RL4RS_X_SPLITPAIR
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
            half2_t h2, l2;
            split_h16_pair(v[j], v[j + 1], h2, l2);        // same roundings: hi = RNE(x), lo = RNE(x - hi) from ONE fp32 value each
            vh[j] = h2[0]; vh[j + 1] = h2[1];
            vl[j] = l2[0]; vl[j + 1] = l2[1];
        }
#else
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // the value must be ONE rounded fp32 number for both uses below: left transparent, the compiler contracts the
            // producing multiply into the fp16 conversion for one of them (v_fma_mixlo_f16: single rounding from the exact
            // product) but not for the other, and hi + lo then misses v by an fp16 ulp in the double-rounding cases
            float x = v[j];
            asm volatile("" : "+v"(x));
            const _Float16 h = (_Float16)x;
            vh[j] = h;
            vl[j] = (_Float16)(x - (float)h);
        }
#endif
        // column 32w + 8q + 4half + j -> k-block 2w + q/2, k-half q%2, element 4half + j
        const int o = m * TILE + (2 * wave + (q >> 1)) * 1024 + (q & 1) * 512 + li * 16 + half * 8;
        *reinterpret_cast<half4_t*>(p_hi + o) = vh;
        *reinterpret_cast<half4_t*>(p_lo + o) = vl;
    };
    const float k_r = a.k_r[sq][wave], k_u = a.k_u[sq][wave], k_c = a.k_c[sq][wave];   // activations of prescaled pre-activations (RecurArgs)
    float quad[MT][4];
    auto reset_gate = [&](int r) {                                 // element r of the reset gate: r*h -> planes
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            quad[m][r & 3] = ((RL4RS_X_AB & 1) ? acc_r[m][r] : gate_sigmoid_k(acc_r[m][r], k_r)) * h_own[m][r];
            if ((r & 3) == 3) plane_store(rp_hi, rp_lo, m, r >> 2, quad[m]);
        }
    };
    auto update_gate = [&](int r) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            float pre = acc_u[m][r];
            asm volatile("" : "+v"(pre));                          // keeps this element's chain where it is written
            acc_u[m][r] = oma[m] * ((RL4RS_X_AB & 1) ? pre : gate_sigmoid_k(pre, k_u));
        }
    };
    auto blend = [&](int r) {                                      // candidate + state update of element r -> h planes
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const float cnd = (RL4RS_X_AB & 1) ? acc_c[m][r] : gate_tanh_k(acc_c[m][r], k_c);
            const float hn = __builtin_fmaf(acc_u[m][r], h_own[m][r] - cnd, cnd);       // u h + (1-u) c
#if RL4RS_X_AMAX
            amax[m] = fmaxf(amax[m], fabsf(hn));
#endif
            h_own[m][r] = hn;
            quad[m][r & 3] = hn;
            if ((r & 3) == 3) plane_store(hp_hi, hp_lo, m, r >> 2, quad[m]);
        }
    };

    const bool early = wave < 4;
#if RL4RS_X_PRIO == 1 || RL4RS_X_PRIO == 2
    // the late waves carry their candidate epilogue next to their own MFMAs (R-early phase) and are the younger half of the
    // workgroup (the arbitration losers): one static priority raise, no per-phase flips (MI355X_MICROARCH.md, two waves per SIMD #4)
    if (!early) __builtin_amdgcn_s_setprio(RL4RS_X_PRIO);
#endif
    const int TL = a.steps > 0 ? a.steps : L;
#pragma unroll 1
    for (int t = 0; t < TL; ++t) {
        asm volatile("" : "+s"(sb_r), "+s"(sb_u), "+s"(sb_c));     // keep the per-item scalar offsets out of SGPR-hoisting
        RL4RS_XT(0);
#pragma unroll
        for (int m = 0; m < MT; ++m) oma[m] = 1.0f - att_cur[m];
        hfrag(0, 0);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int g = gate(i), cur = i & 1;
            if (i == 8) RL4RS_XT(1);
            if (i == RL4RS_X_XU_AT) {
                if (!(RL4RS_X_AB & (4 | 128))) {                  // x_u(t)   (128: no staging reads, DMA keeps going): staged one step ago
#pragma unroll
                    for (int m = 0; m < MT; ++m) x_read(acc_u[m], 1, m);
                }
            }
            if (i == RL4RS_X_XC_AT && RL4RS_X_XC_AT != 24) {
                if (!(RL4RS_X_AB & (4 | 128))) {                  // x_c(t), ahead of the barrier
#pragma unroll
                    for (int m = 0; m < MT; ++m) x_read(acc_c[m], 2, m);
                }
            }
            if (i == 24) {
                RL4RS_XT(2);
                if (!(RL4RS_X_AB & 16)) __syncthreads();           // r*h planes complete
                RL4RS_XT(3);
                if (RL4RS_X_XC_AT == 24 && !(RL4RS_X_AB & (4 | 128))) {                  // x_c(t)
#pragma unroll
                    for (int m = 0; m < MT; ++m) x_read(acc_c[m], 2, m);
                }
                if (!early && !RL4RS_X_LATE_UPD_SHADOW) {
                    // late role: the whole update gate first (VALU only) - its partner on the SIMD is already in its C items
#pragma unroll
                    for (int r = 0; r < 16; ++r) update_gate(r);
#if RL4RS_X_PRIO == 3
                    __builtin_amdgcn_s_setprio(1);        // ... and then must not lose every MFMA arbitration to the (older) early wave
#endif
                }
                hfrag(cur, i);
            }
            if (i == 40) {
                if (early) {
                    // early role: candidate + blend after its C items, while its partner runs C on the matrix pipe
#pragma unroll
                    for (int r = 0; r < 16; ++r) blend(r);
                }
                RL4RS_XT(4);
#if RL4RS_X_PRIO == 3
                if (!early) __builtin_amdgcn_s_setprio(0);
#endif
                if (!(RL4RS_X_AB & 16)) __syncthreads();           // early half of the new state complete
                RL4RS_XT(5);
                if (!(RL4RS_X_AB & (4 | 128))) {                  // x_r(t+1) (requested 12+ items ago)
#pragma unroll
                    for (int m = 0; m < MT; ++m) x_read(acc_r[m], 0, m);
                }
                hfrag(cur, i);
            }
            if (i == WINDOW && t + 1 < L && !(RL4RS_X_AB & (4 | 64))) {       // 64: no projection DMA (reads keep going)
                // ---- the ONE projection issue window of the step: the register-resident items follow (no vector-memory wait)
                x_dma(t + 1);
#pragma unroll
                for (int m = 0; m < MT; ++m) att_next[m] = att_row[m][t + 1];
            }
            // ---- fetch ahead: state fragments of the next item, streamed weights LA items ahead
            if (i + 1 < NI && !after_barrier(i + 1) && !(RL4RS_X_AB & 8)) hfrag(cur ^ 1, i + 1);
            const bool resident = is_res<NRES>(i);
            const int js = resident ? 0 : SC.js_of[i];             // position in the streamed sequence (if streamed)
            const int ridx = resident ? SC.res_idx[i] : 0;
            if (!resident && !(RL4RS_X_AB & 2)) wload(SC.item_of[(js + LA) % NS], ring_h[(js + LA) % RING], ring_l[(js + LA) % RING]);
            __builtin_amdgcn_sched_barrier(0);
            const half8_t wh = resident ? res_h[ridx] : ring_h[js % RING];
            const half8_t wl = resident ? res_l[ridx] : ring_l[js % RING];
            // product terms outermost: with two row tiles the dependent MFMAs of one accumulator are a tile apart
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    f32x16& acc = g == 0 ? acc_r[m] : (g == 1 ? acc_u[m] : acc_c[m]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 1 ? wl : wh, term == 2 ? bl[cur][m] : bh[cur][m], acc, 0, 0, 0);
                }
            // ---- epilogue work in this item's MFMA shadow
            if (i >= 8 && i < 24) {
                reset_gate(i - 8);
            } else if ((early || RL4RS_X_LATE_UPD_SHADOW) && i >= 24 && i < 32) {
                update_gate(2 * (i - 24));
                update_gate(2 * (i - 24) + 1);
            } else if (!early && i >= 40) {
                blend(2 * (i - 40));
                blend(2 * (i - 40) + 1);
            }
#if RL4RS_X_SGB
            if ((i >= 8 && i < 24) || (early && i >= 24 && i < 32) || (!early && i >= 40)) {
                // a wave issues in order: spread the VALU chunk over the item's three MFMAs instead of behind the last one
#pragma unroll
                for (int q = 0; q < 3 * MT; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, RL4RS_X_SGB, 0);     // VALU in its shadow
                }
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) att_cur[m] = att_next[m];
        RL4RS_XT(6);
        if (!(RL4RS_X_AB & 16)) __syncthreads();                   // late half of the new state complete
        RL4RS_XT(7);
    }
    // ---- poison rows that left the fp16 range (or went NaN) and write the final state (16-byte stores).  The final state alone
    // decides: a state element beyond the largest finite fp16 number becomes +-inf in the hi plane and -+inf in the lo plane at
    // the step it appears; from then on its row's accumulators hold inf - inf = NaN or a saturated gate times inf, i.e. the
    // state stays inf / NaN to the end (nothing maps them back to a finite number: sigmoid / tanh of +-inf give 0 / 1 / +-1 and
    // the blend multiplies the non-finite h by them or by 0), so |h_final| < 6e4 fails for exactly those rows - no per-step
    // running maximum needed (16 v_max3 per wave and step).  Rows that pass through [6e4, 65504] and come back were carried
    // exactly and are not errors.
    uint32_t* s_bad = reinterpret_cast<uint32_t*>(rp_hi);          // the planes are dead now
    if (tid < 32 * MT) s_bad[tid] = 0u;
    __syncthreads();
    bool any_bad = false;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        bool bad = !(amax[m] < 6.0e4f);
#pragma unroll
        for (int r = 0; r < 16; ++r) bad |= !(fabsf(h_own[m][r]) < 6.0e4f);
        if (bad) atomicOr(&s_bad[m * 32 + li], 1u);
        any_bad |= bad;
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = phys(row0 + m * 32 + li);
        const bool poison = s_bad[m * 32 + li] != 0u;
        if (row0 + m * 32 + li < a.n_rows) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = make_float4(h_own[m][4 * q], h_own[m][4 * q + 1], h_own[m][4 * q + 2], h_own[m][4 * q + 3]);
                if (poison) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                *reinterpret_cast<float4*>(a.out + (int64_t)row * a.out_ld + a.out_off + sq * a.out_seq_off + wave * 32 + 8 * q + 4 * half) = v;
            }
        }
    }
    if (any_bad && a.range_flag) atomicOr(a.range_flag, 1);
}

// MT = 1: 64 KB planes + 96 KB staging = all 160 KB;  MT = 2: 128 KB planes + 24 KB staging
static size_t augru_x_smem(int mt) { return (size_t)mt * 4 * 32 * 256 * 2 + (size_t)8 * (mt == 1 ? 3 * 4096 : 3 * 1024); }

}  // namespace rl4rs
