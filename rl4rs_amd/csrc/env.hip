// Env state machine on gfx950: SlateState / SeqSlateState (rl4rs/env/slate.py, rl4rs/env/seqslate.py)
// as HIP kernels behind the C ABI of include/rl4rs_hip.h.
//
// HBM layout (all row-major, one row per env b):
//   prev_actions int32 [B,T]            action_mask / special_mask as bit rows uint32 [B,W], W=ceil(A/32)
//   dense f32 [B,Dn]  category i32 [B,Cn]  seq1 i32 [B,L]         (the state rows feature_extraction emits)
//   c_dense f32 [B*n,Dn]  c_category i32 [B*n,Cn]                 (complete-state rows for the reward net)
// The catalogue (item_vec f32 [A,D] = 45 KB, price/action_emb f64, special/location bit rows) is tiny:
// item_vec is staged into LDS once per workgroup and rows are assembled with 16-byte LDS reads and
// 16-byte coalesced global stores; everything else is L2-resident.
//
// Bound: HBM writes.  Algorithmic bytes per built row = Dn*4 + Cn*4 written + UD*4 + UC*4 + 36 read
// = 2016 B at the reference config.
#include <cstdlib>
#include <mutex>

#include "common.hpp"

namespace rl4rs {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
}

struct EnvDev {
    int B, T, A, W, E, P, D, UD, UC, L, Dn, Cn, logT, is_seq;
    const float* item_vec;        // [A,D]
    const double* price;          // [A]
    const double* action_emb;     // [A,E]
    const uint32_t* special_bits; // [W]
    const uint32_t* loc_bits;     // [4,W]
    const uint8_t* is_special;    // [A]
    const int32_t* exposed;       // [B,logT]
    const int32_t* feedback;      // [B,logT]
    const int32_t* hist;          // [B,L]
    const float* ud;              // [B,UD]
    const int32_t* ucat;          // [B,UC]
    int32_t* prev;                // [B,T]
    uint32_t* amask;              // [B,W]
    uint32_t* smask;              // [B,W]
    float* dense;                 // [B,Dn]
    int32_t* cat;                 // [B,Cn]
    int32_t* seq1;                // [B,L]
    float* c_dense;               // [B*n,Dn]
    int32_t* c_cat;               // [B*n,Cn]
    int32_t* err;                 // [1]
};

__device__ __forceinline__ uint32_t full_word(int A, int w) {
    int lo = w * 32;
    if (A - lo >= 32) return 0xffffffffu;
    if (A - lo <= 0) return 0u;
    return (1u << (A - lo)) - 1u;
}

// One wave assembles one feature row (slate.py:205-212 / seqslate.py:111-122 + datautil.py:52-65):
//   dense = user_dense | item_vec[page slice] | item_vec[a]      post-padded / truncated to Dn
//   cat   = user_cat | sequence_id | page slice | a              post-padded / truncated to Cn
// s_item: LDS copy of item_vec; s_prev: this wave's LDS copy of prev_actions[b,:].
__device__ __forceinline__ void build_row(const EnvDev& e, const float* s_item, const int32_t* s_prev,
                                          int b, int page_init, int npage, int seq_id, int a,
                                          float* __restrict__ drow, int32_t* __restrict__ crow, int lane) {
    const int D = e.D, UD = e.UD;
    if (((D | UD | e.Dn) & 3) == 0) {
        const int n4 = e.Dn >> 2;
        const float4* ud4 = reinterpret_cast<const float4*>(e.ud + (size_t)b * UD);
        float4* out4 = reinterpret_cast<float4*>(drow);
        for (int q = lane; q < n4; q += 64) {
            int e0 = q << 2;
            float4 v;
            if (e0 < UD) {
                v = ud4[q];
            } else {
                int r = e0 - UD;
                int j = r / D;
                int d = r - j * D;
                int id = -1;
                if (j < npage) id = s_prev[page_init + j];
                else if (j == npage) id = a;
                if (id >= 0) v = *reinterpret_cast<const float4*>(s_item + id * D + d);
                else v = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            out4[q] = v;
        }
    } else {
        for (int x = lane; x < e.Dn; x += 64) {
            float v = 0.f;
            if (x < UD) {
                v = e.ud[(size_t)b * UD + x];
            } else {
                int r = x - UD;
                int j = r / D;
                int d = r - j * D;
                int id = -1;
                if (j < npage) id = s_prev[page_init + j];
                else if (j == npage) id = a;
                if (id >= 0) v = s_item[id * D + d];
            }
            drow[x] = v;
        }
    }
    for (int c = lane; c < e.Cn; c += 64) {
        int v = 0;
        if (c < e.UC) v = e.ucat[(size_t)b * e.UC + c];
        else if (c == e.UC) v = seq_id;
        else {
            int j = c - e.UC - 1;
            if (j < npage) v = s_prev[page_init + j];
            else if (j == npage) v = a;
        }
        crow[c] = v;
    }
}

__device__ __forceinline__ void stage_items(const EnvDev& e, float* s_item) {
    const int n = e.A * e.D;
    if ((n & 3) == 0) {
        const float4* src = reinterpret_cast<const float4*>(e.item_vec);
        float4* dst = reinterpret_cast<float4*>(s_item);
        for (int i = threadIdx.x; i < (n >> 2); i += blockDim.x) dst[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < n; i += blockDim.x) s_item[i] = e.item_vec[i];
    }
}

#ifndef RL4RS_ROWS_PER_ENV
#define RL4RS_ROWS_PER_ENV 1     // complete rows: 1 = one wave per env writes all its rows, 0 = one wave per row (A/B builds)
#endif

// mode 0: un-acted state rows (reset);  mode 1: act(actions) then state rows;  mode 2: complete rows.
// One wave assembles one row; waves are independent (each keeps its env's prev_actions row in a wave-private
// LDS slot, ordered by wave-level fences only), so after the one-time catalogue staging there is no
// workgroup barrier on the row loop.  USE_LDS = false reads the 45 KB catalogue through L1/L2 instead of
// staging it (kept for A/B measurements).
template <int MODE, bool USE_LDS>
__global__ __launch_bounds__(1024) void k_env_rows(EnvDev e, const int32_t* __restrict__ actions,
                                                   int cur, int n_complete, int j_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_item_lds = reinterpret_cast<float*>(smem);
    const size_t item_bytes = USE_LDS ? (((size_t)e.A * e.D * 4 + 15) & ~size_t(15)) : 0;
    int32_t* s_prev_all = reinterpret_cast<int32_t*>(smem + item_bytes);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    int32_t* s_prev = s_prev_all + wave * e.T;
    if (USE_LDS && MODE != 0) {
        stage_items(e, s_item_lds);
        __syncthreads();
    }
    const float* s_item = USE_LDS ? s_item_lds : e.item_vec;
    const int R = (MODE == 2) ? e.B * n_complete : e.B;
    const int total_waves = gridDim.x * nw;
    if (MODE == 2 && RL4RS_ROWS_PER_ENV) {
        // complete rows: one wave writes ALL n_complete rows of its env - prev_actions and the user columns are fetched once,
        // then the wave only issues stores (a row per wave paid one dependent global round trip per 1.8 KB row)
        for (int b = blockIdx.x * nw + wave; b < e.B; b += total_waves) {
            for (int j = lane; j < e.T; j += 64) s_prev[j] = e.prev[(size_t)b * e.T + j];
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int jj = 0; jj < n_complete; ++jj) {
                const int j = j_base + jj;
                const size_t r = (size_t)b * n_complete + jj;
                int page_init = 0, npage = e.T, seq_id = 1;
                if (e.is_seq) {
                    page_init = j / e.P * e.P;
                    npage = min(e.P, e.T - page_init);
                    seq_id = j / e.P + 1;
                }
                build_row(e, s_item, s_prev, b, page_init, npage, seq_id, s_prev[j], e.c_dense + r * e.Dn, e.c_cat + r * e.Cn, lane);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // s_prev is rewritten for the next env
        }
        return;
    }
    for (int r = blockIdx.x * nw + wave; r < R; r += total_waves) {
        const int b = (MODE == 2) ? r / n_complete : r;
        int a = 0;
        for (int j = lane; j < e.T; j += 64) s_prev[j] = e.prev[(size_t)b * e.T + j];
        if (MODE == 1) {
            a = actions[b];
            if (a < 0 || a >= e.A) {      // numpy would raise IndexError (slate.py:199)
                if (lane == 0) atomicExch(e.err, 1);
                a = 0;
            }
            s_prev[cur] = a;   // every lane writes the same value
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (MODE == 0) {
            // RecState.__init__: dense = user_dense, category = user_cat, both padded (base.py:30-31)
            build_row(e, s_item, s_prev, b, 0, -1, 0, 0, e.dense + (size_t)b * e.Dn, e.cat + (size_t)b * e.Cn, lane);
            for (int l = lane; l < e.L; l += 64) e.seq1[(size_t)b * e.L + l] = 0;
        } else if (MODE == 1) {
            // ---- act (slate.py:198-202 / seqslate.py:98-102)
            if (lane == 0) {
                e.prev[(size_t)b * e.T + cur] = a;
                e.amask[(size_t)b * e.W + (a >> 5)] &= ~(1u << (a & 31));
            }
            bool hit = false;
            for (int j = lane; j < e.T; j += 64) hit |= (e.is_special[s_prev[j]] != 0);
            const bool any_hit = __any(hit);
            const bool page_end = e.is_seq && ((cur + 1) % e.P == 0);
            for (int w = lane; w < e.W; w += 64) {
                if (page_end) {          // seqslate.py:124-126
                    uint32_t f = full_word(e.A, w);
                    e.amask[(size_t)b * e.W + w] = f;
                    e.smask[(size_t)b * e.W + w] = f;
                } else if (any_hit) {
                    e.smask[(size_t)b * e.W + w] &= ~e.special_bits[w];
                }
            }
            // ---- rebuild the state row (slate.py:203-213 / seqslate.py:103-122)
            int page_init = 0, npage = e.T, seq_id = 1;
            if (e.is_seq) {
                page_init = cur / e.P * e.P;
                npage = min(e.P, e.T - page_init);
                seq_id = cur / e.P + 1;
                // second sequence = items of the previous pages, pre-padded (seqslate.py:107-108)
                for (int l = lane; l < e.L; l += 64) {
                    int v = 0;
                    if (page_init > 0) {
                        int idx = page_init - e.L + l;
                        if (idx >= 0) v = s_prev[idx];
                    }
                    e.seq1[(size_t)b * e.L + l] = v;
                }
            }
            build_row(e, s_item, s_prev, b, page_init, npage, seq_id, a, e.dense + (size_t)b * e.Dn,
                      e.cat + (size_t)b * e.Cn, lane);
        } else {
            // ---- complete-state row j of env b (slate.py:117-131 / seqslate.py:27-50)
            const int j = j_base + (r - b * n_complete);
            int page_init = 0, npage = e.T, seq_id = 1;
            if (e.is_seq) {
                page_init = j / e.P * e.P;
                npage = min(e.P, e.T - page_init);
                seq_id = j / e.P + 1;
            }
            a = s_prev[j];
            build_row(e, s_item, s_prev, b, page_init, npage, seq_id, a, e.c_dense + (size_t)r * e.Dn,
                      e.c_cat + (size_t)r * e.Cn, lane);
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // s_prev is rewritten by the next row
    }
}

// ---------------------------------------------------------------------------------------------
// masked K-NN argmax in float64 (slate.py:186-191): one wave per env.
template <typename TA>
__global__ __launch_bounds__(256) void k_knn(const TA* __restrict__ actions, int n, const double* __restrict__ emb,
                                             int A, int E, const uint32_t* __restrict__ amask,
                                             const uint32_t* __restrict__ smask, const uint32_t* __restrict__ loc,
                                             int W, const uint8_t* __restrict__ dense_mask,
                                             int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* s_act = reinterpret_cast<double*>(smem);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    double* my = s_act + wave * E;
    const int b = blockIdx.x * nw + wave;
    if (b < n)
        for (int j = lane; j < E; j += 64) my[j] = (double)actions[(size_t)b * E + j];
    __syncthreads();
    if (b >= n) return;
    double best = -1.0e308;
    int best_k = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        double s = 0.0;
        const double* row = emb + (size_t)k * E;
        for (int j = 0; j < E; ++j) s = __dadd_rn(s, __dmul_rn(my[j], row[j]));
        if (amask) {
            uint32_t m = amask[(size_t)b * W + (k >> 5)] & smask[(size_t)b * W + (k >> 5)] & loc[k >> 5];
            if (!((m >> (k & 31)) & 1u)) s = -2147483648.0;     // action_score[mask < 0.5] = -2**31
        }
        if (dense_mask && !dense_mask[(size_t)b * A + k]) s = -2147483648.0;
        if (best_k == 0x7fffffff || s > best) { best = s; best_k = k; }
    }
    // first-max wins: lower index on ties (np.argmax)
    for (int off = 32; off > 0; off >>= 1) {
        double os = __shfl_xor(best, off);
        int ok = __shfl_xor(best_k, off);
        if (ok != 0x7fffffff && (best_k == 0x7fffffff || os > best || (os == best && ok < best_k))) {
            best = os;
            best_k = ok;
        }
    }
    if (lane == 0) out[b] = best_k;
}

// The same K-NN with the action-embedding table staged in LDS (A x E doubles = 72 KB for the 284 x 32 catalogue): in k_knn
// every lane walks its own table row straight from memory, 8 bytes at a 256-byte stride - 64 cache lines per load instruction
// and the whole table once per env.  Here a workgroup of NW waves copies the table once (coalesced), rows padded to E + 1
// doubles so the row-per-lane reads are conflict-free.  Same products, same summation order: bit-identical choices.
template <typename TA>
__global__ __launch_bounds__(1024) void k_knn_lds(const TA* __restrict__ actions, int n, const double* __restrict__ emb,
                                                  int A, int E, const uint32_t* __restrict__ amask,
                                                  const uint32_t* __restrict__ smask, const uint32_t* __restrict__ loc,
                                                  int W, const uint8_t* __restrict__ dense_mask,
                                                  int32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LDE = E + 1;
    double* s_emb = reinterpret_cast<double*>(smem);            // [A][E + 1]
    double* s_act = s_emb + (size_t)A * LDE;                    // [NW][E]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < A * E; i += blockDim.x) s_emb[(i / E) * LDE + (i % E)] = emb[i];
    double* my = s_act + wave * E;
    const int b = blockIdx.x * nw + wave;
    if (b < n)
        for (int j = lane; j < E; j += 64) my[j] = (double)actions[(size_t)b * E + j];
    __syncthreads();
    if (b >= n) return;
    double best = -1.0e308;
    int best_k = 0x7fffffff;
    for (int k = lane; k < A; k += 64) {
        double s = 0.0;
        const double* row = s_emb + (size_t)k * LDE;
        for (int j = 0; j < E; ++j) s = __dadd_rn(s, __dmul_rn(my[j], row[j]));
        if (amask) {
            uint32_t m = amask[(size_t)b * W + (k >> 5)] & smask[(size_t)b * W + (k >> 5)] & loc[k >> 5];
            if (!((m >> (k & 31)) & 1u)) s = -2147483648.0;     // action_score[mask < 0.5] = -2**31
        }
        if (dense_mask && !dense_mask[(size_t)b * A + k]) s = -2147483648.0;
        if (best_k == 0x7fffffff || s > best) { best = s; best_k = k; }
    }
    for (int off = 32; off > 0; off >>= 1) {                    // first-max wins: lower index on ties (np.argmax)
        double os = __shfl_xor(best, off);
        int ok = __shfl_xor(best_k, off);
        if (ok != 0x7fffffff && (best_k == 0x7fffffff || os > best || (os == best && ok < best_k))) {
            best = os;
            best_k = ok;
        }
    }
    if (lane == 0) out[b] = best_k;
}

// policy_model.predict_with_mask (rl4rs/policy/policy_model.py:17-41) / CustomVectorEncoder mask rule
// (rl4rs/nets/cql/encoder.py:42-67): the action mask is re-derived from the observation tail
// [prev_actions (page_items) | cur_step]: mask = location_mask[cur_step % page_items // 3] with every previous
// action zeroed; masked scores = -2**15; if a special item was already chosen, all special items = -2**15;
// argmax (first max).  One wave per row; scores float32 [N, A].
__global__ __launch_bounds__(256) void k_predict_with_mask(EnvDev e, int N, const float* __restrict__ scores,
                                                            const int32_t* __restrict__ prev, int prev_cols,
                                                            const int32_t* __restrict__ cur_step, int32_t* __restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int layer = (cur_step[n] % e.P) / 3;
    const int32_t* pr = prev + (size_t)n * prev_cols;
    bool sp = false;
    for (int j = lane; j < prev_cols; j += 64) {
        int id = pr[j];
        if (id >= 0 && id < e.A) sp |= (e.is_special[id] != 0);
    }
    const bool any_special = __any(sp);
    float best = 0.f;
    int best_k = 0x7fffffff;
    for (int k = lane; k < e.A; k += 64) {
        bool ok = (e.loc_bits[layer * e.W + (k >> 5)] >> (k & 31)) & 1u;
        for (int j = 0; j < prev_cols; ++j) ok = ok && (pr[j] != k);
        float s = scores[(size_t)n * e.A + k];
        if (!ok) s = -32768.0f;
        if (any_special && e.is_special[k]) s = -32768.0f;
        if (best_k == 0x7fffffff || s > best) { best = s; best_k = k; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        float os = __shfl_xor(best, off);
        int ok = __shfl_xor(best_k, off);
        if (ok != 0x7fffffff && (best_k == 0x7fffffff || os > best || (os == best && ok < best_k))) { best = os; best_k = ok; }
    }
    if (lane == 0) out[n] = best_k;
}

// numpy float64 add.reduce order (0 + pairwise_sum, loops_utils.h.src) for n <= 128; term(i) yields the i-th addend (the terms
// are generated in the order the sum consumes them: no per-thread array - a `double terms[128]` was 1 040 bytes of scratch)
template <typename Term>
__device__ __forceinline__ double np_pairwise(Term term, int n) {
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res = __dadd_rn(res, term(i));
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = term(j);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = __dadd_rn(r[j], term(i + j));
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                           __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    for (; i < n; ++i) res = __dadd_rn(res, term(i));
    return res;
}

__device__ __forceinline__ bool loc_ok(const EnvDev& e, int layer, int id) {
    return (e.loc_bits[layer * e.W + (id >> 5)] >> (id & 31)) & 1u;
}

// get_violation (slate.py:133-147 / seqslate.py:52-69); cur = cur_steps.  Thread per env.
__device__ int violation_of(const EnvDev& e, int b, int cur) {
    const int32_t* pa = e.prev + (size_t)b * e.T;
    int ok = 1;
    for (int s = 0; s < cur; ++s) {
        int layer = e.is_seq ? (s % e.P) / 3 : s / 3;
        ok &= loc_ok(e, layer, pa[s]) ? 1 : 0;
    }
    int n1 = max(cur - 1, 1), n2 = max(cur - 2, 1);
    for (int s = 0; s < n1; ++s) ok &= (pa[s] != pa[s + 1]);
    for (int s = 0; s < n2; ++s) ok &= (pa[s] != pa[s + 2]);
    // more than one DISTINCT special item (np.intersect1d) in the row / in each inspected page
    if (!e.is_seq) {
        int first = -1;
        for (int s = 0; s < e.T; ++s) {
            int id = pa[s];
            if (e.is_special[id]) {
                if (first < 0) first = id;
                else if (id != first) ok = 0;
            }
        }
    } else {
        int pages = cur % e.P + 1;
        for (int j = 0; j < pages; ++j) {
            int first = -1;
            for (int s = e.P * j; s < min(e.P * (j + 1), e.T); ++s) {
                int id = pa[s];
                if (e.is_special[id]) {
                    if (first < 0) first = id;
                    else if (id != first) ok = 0;
                }
            }
        }
    }
    return ok;
}

__global__ void k_violation(EnvDev e, int cur, int32_t* out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < e.B) out[b] = violation_of(e, b, cur);
}

// reward = np.sum(price * probs, axis=1); reward[violation < 0.5] = 0 (slate.py:294-307, seqslate.py:147-157)
// p_last (optional): probability of the LAST row of each env supplied separately; probs then holds n-1 per env
__global__ void k_reward(EnvDev e, int cur, int n, int j_base, int zero_on_violation,
                         const float* __restrict__ probs, const float* __restrict__ p_last, double* __restrict__ out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= e.B) return;
    const int m = p_last ? n - 1 : n;
    double r = np_pairwise([&](int j) {
        const int id = e.prev[(size_t)b * e.T + j_base + j];
        const float pj = (j < m) ? probs[(size_t)b * m + j] : p_last[b];
        return __dmul_rn(e.price[id], (double)pj);
    }, n);
    if (zero_on_violation && violation_of(e, b, cur) == 0) r = 0.0;
    out[b] = r;
}

__global__ void k_obs_mask(EnvDev e, int layer, void* out, int dtype) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)e.B * e.A;
    if (i >= total) return;
    int b = (int)(i / e.A), k = (int)(i - (size_t)b * e.A);
    uint32_t m = e.amask[(size_t)b * e.W + (k >> 5)] & e.smask[(size_t)b * e.W + (k >> 5)] &
                 e.loc_bits[layer * e.W + (k >> 5)];
    int v = (m >> (k & 31)) & 1u;
    switch (dtype) {
        case 0: reinterpret_cast<uint8_t*>(out)[i] = (uint8_t)v; break;
        case 1: reinterpret_cast<int32_t*>(out)[i] = v; break;
        case 2: reinterpret_cast<int64_t*>(out)[i] = v; break;
        default: reinterpret_cast<float*>(out)[i] = (float)v; break;
    }
}

// packed form of the same mask: one uint32 word per 32 actions (what the policy kernels take as mask_bits)
__global__ void k_obs_mask_bits(EnvDev e, int layer, uint32_t* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= e.B * e.W) return;
    const int w = i % e.W;
    out[i] = e.amask[i] & e.smask[i] & e.loc_bits[layer * e.W + w];
}

__global__ void k_offline_action(EnvDev e, int cur, int32_t* ids, double* emb) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= e.B) return;
    int id = (cur < e.T && cur < e.logT) ? e.exposed[(size_t)b * e.logT + cur] : 0;   // slate.py:152-161
    if (ids) ids[b] = id;
    if (emb) {
        if (id < 0 || id >= e.A) { atomicExch(e.err, 1); id = 0; }
        for (int j = 0; j < e.E; ++j) emb[(size_t)b * e.E + j] = e.action_emb[(size_t)id * e.E + j];
    }
}

// Slate: python sum() left to right over zip(price, label) (slate.py:169-173);
// SeqSlate: np.sum over the page just finished (seqslate.py:76-85)
__global__ void k_offline_reward(EnvDev e, int cur, double* out) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= e.B) return;
    const int32_t* ex = e.exposed + (size_t)b * e.logT;
    const int32_t* fb = e.feedback + (size_t)b * e.logT;
    double r = 0.0;
    if (!e.is_seq) {
        if (cur >= e.T)
            for (int j = 0; j < e.logT; ++j) r = __dadd_rn(r, __dmul_rn(e.price[ex[j]], (double)fb[j]));
    } else if (cur % 9 == 0) {
        const int lo = max(cur - e.P, 0), n = max(min(cur, e.logT) - lo, 0);
        r = np_pairwise([&](int j) { return __dmul_rn(e.price[ex[lo + j]], (double)fb[lo + j]); }, n);
    }
    out[b] = r;
}

// RecDataBase.sample + SlateState.__init__ (base.py:92-100, slate.py:8-27) as ONE launch: the sampled log lines' columns
// from the resident log tables into the env's batch buffers, the episode state reset (prev_actions = 0, both masks all ones),
// and the batch's DISTINCT histories (what the scorer encodes once each) into a caller buffer.  One wave per row.
__global__ __launch_bounds__(256) void k_load_lines(EnvDev e, const int32_t* __restrict__ s_exposed, const int32_t* __restrict__ s_feedback,
                                                    const int32_t* __restrict__ s_hist, const float* __restrict__ s_ud,
                                                    const int32_t* __restrict__ s_ucat, int n_lines, const int32_t* __restrict__ line_idx,
                                                    const int32_t* __restrict__ uniq_idx, int n_uniq, int32_t* __restrict__ hist_unique) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wave;
    if (r < e.B) {
        int line = line_idx[r];
        line = line < 0 ? 0 : (line >= n_lines ? n_lines - 1 : line);
        // (EnvDev carries the batch columns read-only for every other kernel: this is the one that fills them)
        int32_t* d_exposed = const_cast<int32_t*>(e.exposed);
        int32_t* d_feedback = const_cast<int32_t*>(e.feedback);
        int32_t* d_hist = const_cast<int32_t*>(e.hist);
        float* d_ud = const_cast<float*>(e.ud);
        int32_t* d_ucat = const_cast<int32_t*>(e.ucat);
        for (int j = lane; j < e.logT; j += 64) {
            d_exposed[(size_t)r * e.logT + j] = s_exposed[(size_t)line * e.logT + j];
            d_feedback[(size_t)r * e.logT + j] = s_feedback[(size_t)line * e.logT + j];
        }
        for (int j = lane; j < e.L; j += 64) d_hist[(size_t)r * e.L + j] = s_hist[(size_t)line * e.L + j];
        for (int j = lane; j < e.UD; j += 64) d_ud[(size_t)r * e.UD + j] = s_ud[(size_t)line * e.UD + j];
        for (int j = lane; j < e.UC; j += 64) d_ucat[(size_t)r * e.UC + j] = s_ucat[(size_t)line * e.UC + j];
        for (int j = lane; j < e.T; j += 64) e.prev[(size_t)r * e.T + j] = 0;
        for (int j = lane; j < e.W; j += 64) { e.amask[(size_t)r * e.W + j] = 0xffffffffu; e.smask[(size_t)r * e.W + j] = 0xffffffffu; }
    } else if (r - e.B < n_uniq) {
        const int u = r - e.B;
        int line = uniq_idx[u];
        line = line < 0 ? 0 : (line >= n_lines ? n_lines - 1 : line);
        for (int j = lane; j < e.L; j += 64) hist_unique[(size_t)u * e.L + j] = s_hist[(size_t)line * e.L + j];
    }
}

}  // namespace rl4rs

// =================================================================================================
// C ABI
// =================================================================================================
using namespace rl4rs;

struct rl4rs_env {
    rl4rs_env_cfg cfg;
    EnvDev d;
    int device;
    int cur_steps;
    int n_complete;
    bool catalog_set, batch_set;
    bool state_fresh;          // rl4rs_env_load_lines already wrote prev_actions = 0 and the all-ones masks: reset skips its memsets
    int rows_variant;          // rl4rs_env_set_option(RL4RS_ENV_OPT_ROWS_VARIANT)
    // owned device memory
    float* item_vec; double* price; double* action_emb; uint32_t* special_bits; uint32_t* loc_bits;
    uint8_t* is_special;
    int32_t* exposed; int32_t* feedback; int32_t* hist; float* ud; int32_t* ucat;
    int32_t* prev; uint32_t* amask; uint32_t* smask; float* dense; int32_t* cat; int32_t* seq1;
    float* c_dense; int32_t* c_cat; int32_t* err; int32_t* knn_tmp;
};

static size_t rows_smem(const rl4rs_env* e, int waves, bool lds) {
    return (lds ? (((size_t)e->d.A * e->d.D * 4 + 15) & ~size_t(15)) : 0) + (size_t)waves * e->d.T * 4;
}
static int check_rows_smem(const rl4rs_env* e) {
    if (rows_smem(e, 16, true) > 65536) {
        set_error("catalogue of %d x %d floats does not fit the 64 KB LDS staging buffer", e->d.A, e->d.D);
        return RL4RS_EINVAL;
    }
    return RL4RS_OK;
}
// launch geometry: persistent workgroups of `waves` waves, each wave strides over the rows
struct RowsLaunch { int grid, threads; size_t smem; bool lds; };
static RowsLaunch rows_launch(const rl4rs_env* e, int R, int mode) {
    RowsLaunch L;
    L.lds = (e->rows_variant == 0) && mode != 0;    // RL4RS_ENV_OPT_ROWS_VARIANT: 0 = LDS-staged catalogue (default), 1 = catalogue via L1/L2
    int waves = 16;
    if (R < 256 * 16) waves = 4;                 // small batches: more, smaller workgroups
    int grid = (R + waves - 1) / waves;
    if (mode == 2 && RL4RS_ROWS_PER_ENV) {       // one wave per env
        const int B = e->cfg.batch_size;
        waves = B < 256 * 16 ? 4 : 16;
        grid = (B + waves - 1) / waves;
    }
    const int cap = L.lds ? 256 * (waves == 16 ? 2 : 3) : 256 * 8;   // LDS staging: amortise the 45 KB copy
    if (grid > cap) grid = cap;
    if (grid < 1) grid = 1;
    L.grid = grid;
    L.threads = waves * 64;
    L.smem = rows_smem(e, waves, L.lds);
    return L;
}
template <int MODE>
static int launch_rows(rl4rs_env* e, int R, const int32_t* actions, int cur, int n_complete, int j_base, hipStream_t st) {
    if (MODE == 1) e->state_fresh = false;      // an act writes prev_actions / the masks: the next reset must clear them again
    RowsLaunch L = rows_launch(e, R, MODE);
    if (L.lds)
        hipLaunchKernelGGL((k_env_rows<MODE, true>), dim3(L.grid), dim3(L.threads), L.smem, st, e->d, actions, cur, n_complete, j_base);
    else
        hipLaunchKernelGGL((k_env_rows<MODE, false>), dim3(L.grid), dim3(L.threads), L.smem, st, e->d, actions, cur, n_complete, j_base);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

namespace rl4rs {
// The attribute is a property of (device, function): the cache is keyed by both and guarded, so handles created on a second
// device or from another thread of the process still opt in (a process normally drives one device).
int current_device() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev;
}
int raise_dyn_smem(const void* fn, size_t bytes) {
    struct Entry { int dev; const void* fn; size_t bytes; };
    static std::vector<Entry> cur;              // a handful of kernels: a linear scan is fine
    static std::mutex mu;
    std::lock_guard<std::mutex> lock(mu);
    const int dev = current_device();
    for (auto& e : cur)
        if (e.dev == dev && e.fn == fn) {
            if (bytes <= e.bytes) return RL4RS_OK;
            RL4RS_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            e.bytes = bytes;
            return RL4RS_OK;
        }
    RL4RS_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur.push_back(Entry{dev, fn, bytes});
    return RL4RS_OK;
}
}  // namespace rl4rs

extern "C" {

const char* rl4rs_last_error(void) { return g_err.c_str(); }
int rl4rs_abi_version(void) { return RL4RS_ABI_VERSION; }
int rl4rs_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return RL4RS_EHIP;
    }
    return n;
}

int rl4rs_copy_d2d(void* dst, const void* src, int64_t n, void* stream) {
    RL4RS_REQUIRE(dst && src && n >= 0, "rl4rs_copy_d2d: bad argument");
    RL4RS_HIP_TRY(hipMemcpyAsync(dst, src, (size_t)n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RL4RS_OK;
}

int rl4rs_env_create(const rl4rs_env_cfg* c, rl4rs_env** out) {
    RL4RS_REQUIRE(c && out, "rl4rs_env_create: null argument");
    RL4RS_REQUIRE(c->batch_size > 0 && c->max_steps > 0 && c->action_size > 1, "bad batch/max_steps/action_size");
    RL4RS_REQUIRE(c->page_items > 0 && c->page_items <= 128 && c->max_steps <= 128, "page_items/max_steps must be <= 128");
    RL4RS_REQUIRE(c->item_dim > 0 && c->user_dense_dim >= 0 && c->user_cat_dim >= 0, "bad feature dims");
    RL4RS_REQUIRE(c->maxlen > 0 && c->dense_feature_num > 0 && c->category_feature_num > 0, "bad feature widths");
    RL4RS_REQUIRE(c->log_steps > 0 && c->log_steps <= 128, "log_steps must be in 1..128");
    RL4RS_REQUIRE(c->action_emb_size > 0, "bad action_emb_size");
    int ndev = rl4rs_device_count();
    if (ndev <= 0) {
        set_error("no HIP device visible: librl4rs_hip has no CPU fallback");
        return RL4RS_EHIP;
    }
    rl4rs_env* e = new rl4rs_env();
    memset(e, 0, sizeof(*e));
    e->cfg = *c;
    RL4RS_HIP_TRY(hipGetDevice(&e->device));
    const int B = c->batch_size, T = c->max_steps, A = c->action_size, W = (A + 31) / 32;
    e->n_complete = c->is_seq ? c->page_items : T;
    int rc;
    e->d.A = A; e->d.D = c->item_dim; e->d.T = T;
    if ((rc = check_rows_smem(e)) != RL4RS_OK) { delete e; return rc; }
#define ALLOC(field, n) if ((rc = dev_alloc(&e->field, (size_t)(n))) != RL4RS_OK) return rc
    ALLOC(item_vec, (size_t)A * c->item_dim);
    ALLOC(price, A);
    ALLOC(action_emb, (size_t)A * c->action_emb_size);
    ALLOC(special_bits, W);
    ALLOC(loc_bits, 4 * W);
    ALLOC(is_special, A);
    ALLOC(exposed, (size_t)B * c->log_steps);
    ALLOC(feedback, (size_t)B * c->log_steps);
    ALLOC(hist, (size_t)B * c->maxlen);
    ALLOC(ud, (size_t)B * c->user_dense_dim);
    ALLOC(ucat, (size_t)B * c->user_cat_dim);
    ALLOC(prev, (size_t)B * T);
    ALLOC(amask, (size_t)B * W);
    ALLOC(smask, (size_t)B * W);
    ALLOC(dense, (size_t)B * c->dense_feature_num);
    ALLOC(cat, (size_t)B * c->category_feature_num);
    ALLOC(seq1, (size_t)B * c->maxlen);
    ALLOC(c_dense, (size_t)B * e->n_complete * c->dense_feature_num);
    ALLOC(c_cat, (size_t)B * e->n_complete * c->category_feature_num);
    ALLOC(err, 1);
    ALLOC(knn_tmp, B);
#undef ALLOC
    RL4RS_HIP_TRY(hipMemset(e->err, 0, sizeof(int32_t)));
    EnvDev& d = e->d;
    d.B = B; d.T = T; d.A = A; d.W = W; d.E = c->action_emb_size; d.P = c->page_items; d.D = c->item_dim;
    d.UD = c->user_dense_dim; d.UC = c->user_cat_dim; d.L = c->maxlen; d.Dn = c->dense_feature_num;
    d.Cn = c->category_feature_num; d.logT = c->log_steps; d.is_seq = c->is_seq;
    d.item_vec = e->item_vec; d.price = e->price; d.action_emb = e->action_emb;
    d.special_bits = e->special_bits; d.loc_bits = e->loc_bits; d.is_special = e->is_special;
    d.exposed = e->exposed; d.feedback = e->feedback; d.hist = e->hist; d.ud = e->ud; d.ucat = e->ucat;
    d.prev = e->prev; d.amask = e->amask; d.smask = e->smask; d.dense = e->dense; d.cat = e->cat;
    d.seq1 = e->seq1; d.c_dense = e->c_dense; d.c_cat = e->c_cat; d.err = e->err;
    *out = e;
    return RL4RS_OK;
}

// Kernel-path selection of one env handle (A/B measurements): RL4RS_ENV_OPT_ROWS_VARIANT 0 = the row kernels stage the catalogue
// in LDS (default), 1 = they read it through L1 / L2.  Same rows either way.
int rl4rs_env_set_option(rl4rs_env* e, int32_t which, int32_t value) {
    RL4RS_REQUIRE(e, "env_set_option: null handle");
    switch (which) {
        case RL4RS_ENV_OPT_ROWS_VARIANT:
            RL4RS_REQUIRE(value == 0 || value == 1, "env_set_option: ROWS_VARIANT must be 0 or 1 (got %d)", value);
            e->rows_variant = value; break;
        default: set_error("env_set_option: unknown option %d", which); return RL4RS_EINVAL;
    }
    return RL4RS_OK;
}

int rl4rs_env_destroy(rl4rs_env* e) {
    if (!e) return RL4RS_OK;
    void* ptrs[] = {e->item_vec, e->price, e->action_emb, e->special_bits, e->loc_bits, e->is_special,
                    e->exposed, e->feedback, e->hist, e->ud, e->ucat, e->prev, e->amask, e->smask,
                    e->dense, e->cat, e->seq1, e->c_dense, e->c_cat, e->err, e->knn_tmp};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete e;
    return RL4RS_OK;
}

int rl4rs_env_set_catalog(rl4rs_env* e, const float* item_vec, const double* price, const double* action_emb,
                          const uint8_t* is_special, const uint8_t* location_mask, void* stream) {
    RL4RS_REQUIRE(e && item_vec && price && action_emb && is_special && location_mask, "set_catalog: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int A = e->d.A, W = e->d.W;
    std::string bits((size_t)5 * W * 4, '\0');
    uint32_t* sb = reinterpret_cast<uint32_t*>(&bits[0]);
    uint32_t* lb = sb + W;
    for (int k = 0; k < A; ++k) {
        if (is_special[k]) sb[k >> 5] |= 1u << (k & 31);
        for (int l = 0; l < 4; ++l)
            if (location_mask[(size_t)l * A + k]) lb[l * W + (k >> 5)] |= 1u << (k & 31);
    }
    RL4RS_HIP_TRY(hipMemcpyAsync(e->item_vec, item_vec, (size_t)A * e->d.D * 4, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->price, price, (size_t)A * 8, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->action_emb, action_emb, (size_t)A * e->d.E * 8, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->is_special, is_special, (size_t)A, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->special_bits, sb, (size_t)W * 4, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->loc_bits, lb, (size_t)4 * W * 4, hipMemcpyHostToDevice, st));
    RL4RS_HIP_TRY(hipStreamSynchronize(st));   // host staging buffers go out of scope
    e->catalog_set = true;
    return RL4RS_OK;
}

int rl4rs_env_load_batch(rl4rs_env* e, const int32_t* exposed, const int32_t* feedback, const int32_t* hist,
                         const float* ud, const int32_t* ucat, void* stream) {
    RL4RS_REQUIRE(e && exposed && feedback && hist && ud && ucat, "load_batch: null argument");
    hipStream_t st = (hipStream_t)stream;
    const EnvDev& d = e->d;
    RL4RS_HIP_TRY(hipMemcpyAsync(e->exposed, exposed, (size_t)d.B * d.logT * 4, hipMemcpyDeviceToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->feedback, feedback, (size_t)d.B * d.logT * 4, hipMemcpyDeviceToDevice, st));
    RL4RS_HIP_TRY(hipMemcpyAsync(e->hist, hist, (size_t)d.B * d.L * 4, hipMemcpyDeviceToDevice, st));
    if (d.UD) RL4RS_HIP_TRY(hipMemcpyAsync(e->ud, ud, (size_t)d.B * d.UD * 4, hipMemcpyDeviceToDevice, st));
    if (d.UC) RL4RS_HIP_TRY(hipMemcpyAsync(e->ucat, ucat, (size_t)d.B * d.UC * 4, hipMemcpyDeviceToDevice, st));
    e->batch_set = true;
    e->state_fresh = false;
    return RL4RS_OK;
}

int rl4rs_env_load_lines(rl4rs_env* e, const int32_t* s_exposed, const int32_t* s_feedback, const int32_t* s_hist, const float* s_ud,
                         const int32_t* s_ucat, int32_t n_lines, int32_t store_log_steps, const int32_t* line_idx_dev,
                         const int32_t* uniq_idx_dev, int32_t n_uniq, int32_t* hist_unique_dev, void* stream) {
    RL4RS_REQUIRE(e && s_exposed && s_feedback && s_hist && s_ud && s_ucat && line_idx_dev && n_lines > 0, "load_lines: null argument");
    RL4RS_REQUIRE(store_log_steps == e->d.logT, "load_lines: the log tables have %d logged steps per line, the env was created for %d",
                  store_log_steps, e->d.logT);
    RL4RS_REQUIRE(n_uniq >= 0 && (n_uniq == 0 || (uniq_idx_dev && hist_unique_dev)), "load_lines: distinct-history outputs missing");
    const int rows = e->d.B + n_uniq;
    hipLaunchKernelGGL(k_load_lines, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, e->d, s_exposed, s_feedback, s_hist, s_ud,
                       s_ucat, n_lines, line_idx_dev, uniq_idx_dev, n_uniq, hist_unique_dev);
    RL4RS_LAUNCH_CHECK();
    e->batch_set = true;
    e->state_fresh = true;
    return RL4RS_OK;
}

int rl4rs_env_reset(rl4rs_env* e, void* stream) {
    RL4RS_REQUIRE(e, "reset: null env");
    if (!e->catalog_set || !e->batch_set) {
        set_error("rl4rs_env_reset: catalogue and batch must be loaded first");
        return RL4RS_ESTATE;
    }
    hipStream_t st = (hipStream_t)stream;
    const EnvDev& d = e->d;
    if (!e->state_fresh) {
        RL4RS_HIP_TRY(hipMemsetAsync(e->prev, 0, (size_t)d.B * d.T * 4, st));
        // all-ones masks; bits >= A of the last word are irrelevant but kept clear by page resets
        RL4RS_HIP_TRY(hipMemsetAsync(e->amask, 0xff, (size_t)d.B * d.W * 4, st));
        RL4RS_HIP_TRY(hipMemsetAsync(e->smask, 0xff, (size_t)d.B * d.W * 4, st));
    }
    e->state_fresh = false;
    e->cur_steps = 0;
    return launch_rows<0>(e, d.B, nullptr, 0, 0, 0, st);
}

int rl4rs_env_act_discrete(rl4rs_env* e, const int32_t* actions, void* stream) {
    RL4RS_REQUIRE(e && actions, "act_discrete: null argument");
    if (e->cur_steps >= e->d.T) {    // prev_actions[:, cur_steps] would raise IndexError (slate.py:198)
        set_error("act at cur_steps=%d >= max_steps=%d", e->cur_steps, e->d.T);
        return RL4RS_ESTATE;
    }
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_rows<1>(e, e->d.B, actions, e->cur_steps, 0, 0, st);
    if (rc) return rc;
    e->cur_steps += 1;
    return RL4RS_OK;
}

static int knn_launch(const void* actions, int is_f64, int n, const double* emb, int A, int E,
                      const uint32_t* amask, const uint32_t* smask, const uint32_t* loc, int W,
                      const uint8_t* dense_mask, int32_t* out, hipStream_t st) {
    // batches: 16 envs per workgroup around ONE LDS copy of the table; small calls / huge catalogues: one row per lane from memory
    const size_t lds_bytes = ((size_t)A * (E + 1) + (size_t)16 * E) * sizeof(double);
    static bool lds_ok[2] = {false, false}, lds_tried[2] = {false, false};
    if (n >= 64 && lds_bytes <= 150 * 1024) {
        const int v = is_f64 ? 1 : 0;
        if (!lds_tried[v]) {
            lds_tried[v] = true;
            const void* fn = is_f64 ? reinterpret_cast<const void*>(&k_knn_lds<double>) : reinterpret_cast<const void*>(&k_knn_lds<float>);
            lds_ok[v] = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) == hipSuccess;
            if (!lds_ok[v]) (void)hipGetLastError();
        }
        if (lds_ok[v]) {
            dim3 g16((n + 15) / 16), b1024(1024);
            if (is_f64)
                hipLaunchKernelGGL(k_knn_lds<double>, g16, b1024, lds_bytes, st, (const double*)actions, n, emb, A, E, amask, smask, loc, W, dense_mask, out);
            else
                hipLaunchKernelGGL(k_knn_lds<float>, g16, b1024, lds_bytes, st, (const float*)actions, n, emb, A, E, amask, smask, loc, W, dense_mask, out);
            RL4RS_LAUNCH_CHECK();
            return RL4RS_OK;
        }
    }
    dim3 grid((n + 3) / 4), block(256);
    size_t smem = (size_t)4 * E * sizeof(double);
    if (is_f64)
        hipLaunchKernelGGL(k_knn<double>, grid, block, smem, st, (const double*)actions, n, emb, A, E, amask, smask, loc, W, dense_mask, out);
    else
        hipLaunchKernelGGL(k_knn<float>, grid, block, smem, st, (const float*)actions, n, emb, A, E, amask, smask, loc, W, dense_mask, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_act_conti(rl4rs_env* e, const void* actions, int is_f64, int32_t* chosen, void* stream) {
    RL4RS_REQUIRE(e && actions, "act_conti: null argument");
    if (e->cur_steps >= e->d.T) {
        set_error("act at cur_steps=%d >= max_steps=%d", e->cur_steps, e->d.T);
        return RL4RS_ESTATE;
    }
    hipStream_t st = (hipStream_t)stream;
    const EnvDev& d = e->d;
    int layer = d.is_seq ? (e->cur_steps % d.P) / 3 : e->cur_steps / 3;   // PRE-increment (slate.py:195)
    RL4RS_REQUIRE(layer < 4, "location layer %d out of range (cur_steps=%d)", layer, e->cur_steps);
    int rc = knn_launch(actions, is_f64, d.B, d.action_emb, d.A, d.E, d.amask, d.smask, d.loc_bits + layer * d.W,
                        d.W, nullptr, e->knn_tmp, st);
    if (rc) return rc;
    if (chosen) RL4RS_HIP_TRY(hipMemcpyAsync(chosen, e->knn_tmp, (size_t)d.B * 4, hipMemcpyDeviceToDevice, st));
    return rl4rs_env_act_discrete(e, e->knn_tmp, stream);
}

int rl4rs_knn(const void* actions, int is_f64, int32_t n, const double* emb, int32_t A, int32_t E,
              const uint8_t* mask, int32_t* out, void* stream) {
    RL4RS_REQUIRE(actions && emb && out && n > 0 && A > 0 && E > 0, "rl4rs_knn: bad argument");
    return knn_launch(actions, is_f64, n, emb, A, E, nullptr, nullptr, nullptr, 0, mask, out, (hipStream_t)stream);
}

int rl4rs_env_get_cfg(const rl4rs_env* e, rl4rs_env_cfg* out) {
    RL4RS_REQUIRE(e && out, "env_get_cfg: null argument");
    *out = e->cfg;
    return RL4RS_OK;
}
int rl4rs_env_complete_rows(const rl4rs_env* e) { return e ? e->n_complete : RL4RS_EINVAL; }
int rl4rs_env_cur_steps(const rl4rs_env* e) { return e ? e->cur_steps : RL4RS_EINVAL; }
int rl4rs_env_is_reward_step(const rl4rs_env* e) {
    if (!e) return RL4RS_EINVAL;
    if (e->d.is_seq) return (e->cur_steps % e->d.P == 0) ? 1 : 0;     // seqslate.py:138
    return (e->cur_steps >= e->d.T) ? 1 : 0;                           // slate.py:283
}

static int complete_base(const rl4rs_env* e) {
    // Slate: j = 0..T-1 (slate.py:119); SeqSlate: the last page_items entries of range(cur_steps) (seqslate.py:142)
    return e->d.is_seq ? e->cur_steps - e->d.P : 0;
}

int rl4rs_env_build_complete_rows(rl4rs_env* e, int32_t rows_per_env, void* stream);
int rl4rs_env_build_complete(rl4rs_env* e, void* stream) {
    RL4RS_REQUIRE(e, "build_complete: null env");
    return rl4rs_env_build_complete_rows(e, e->n_complete, stream);
}

int rl4rs_env_build_complete_rows(rl4rs_env* e, int32_t rows_per_env, void* stream) {
    RL4RS_REQUIRE(e, "build_complete: null env");
    RL4RS_REQUIRE(rows_per_env >= 1 && rows_per_env <= e->n_complete, "build_complete: rows_per_env=%d not in 1..%d",
                  rows_per_env, e->n_complete);
    int jb = complete_base(e);
    if (jb < 0 || jb + e->n_complete > e->d.T) {
        set_error("build_complete: cur_steps=%d has no complete page", e->cur_steps);
        return RL4RS_ESTATE;
    }
    hipStream_t st = (hipStream_t)stream;
    int R = e->d.B * rows_per_env;
    return launch_rows<2>(e, R, nullptr, e->cur_steps, rows_per_env, jb, st);
}

int rl4rs_env_reward_split(rl4rs_env* e, const float* probs, const float* p_last, double* reward, void* stream);
int rl4rs_env_reward(rl4rs_env* e, const float* probs, double* reward, void* stream) {
    return rl4rs_env_reward_split(e, probs, nullptr, reward, stream);
}

int rl4rs_env_reward_split(rl4rs_env* e, const float* probs, const float* p_last, double* reward, void* stream) {
    RL4RS_REQUIRE(e && probs && reward, "reward: null argument");
    int jb = complete_base(e);
    if (jb < 0 || jb + e->n_complete > e->d.T) {
        set_error("reward: cur_steps=%d has no complete page", e->cur_steps);
        return RL4RS_ESTATE;
    }
    hipLaunchKernelGGL(k_reward, dim3((e->d.B + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d,
                       e->cur_steps, e->n_complete, jb, e->cfg.violation_zeroes_reward, probs, p_last, reward);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_violation(rl4rs_env* e, int32_t* out, void* stream) {
    RL4RS_REQUIRE(e && out, "violation: null argument");
    hipLaunchKernelGGL(k_violation, dim3((e->d.B + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d,
                       e->cur_steps, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_obs_mask(rl4rs_env* e, void* out, int dtype, void* stream) {
    RL4RS_REQUIRE(e && out && dtype >= 0 && dtype <= 4, "obs_mask: bad argument");
    const EnvDev& d = e->d;
    int layer = d.is_seq ? (e->cur_steps % d.P) / 3 : e->cur_steps / 3;   // POST-increment (slate.py:93)
    RL4RS_REQUIRE(layer < 4, "location layer %d out of range (cur_steps=%d)", layer, e->cur_steps);
    if (dtype == 4) {
        hipLaunchKernelGGL(k_obs_mask_bits, dim3((unsigned)((d.B * d.W + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->d, layer,
                           reinterpret_cast<uint32_t*>(out));
        RL4RS_LAUNCH_CHECK();
        return RL4RS_OK;
    }
    size_t total = (size_t)d.B * d.A;
    hipLaunchKernelGGL(k_obs_mask, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, e->d,
                       layer, out, dtype);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_offline_action(rl4rs_env* e, int32_t* ids, double* emb, void* stream) {
    RL4RS_REQUIRE(e && (ids || emb), "offline_action: null argument");
    hipLaunchKernelGGL(k_offline_action, dim3((e->d.B + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d,
                       e->cur_steps, ids, emb);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_offline_reward(rl4rs_env* e, double* out, void* stream) {
    RL4RS_REQUIRE(e && out, "offline_reward: null argument");
    hipLaunchKernelGGL(k_offline_reward, dim3((e->d.B + 127) / 128), dim3(128), 0, (hipStream_t)stream, e->d,
                       e->cur_steps, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_predict_with_mask(rl4rs_env* e, int32_t N, const float* scores, const int32_t* prev, int32_t prev_cols,
                                const int32_t* cur_step, int32_t* out, void* stream) {
    RL4RS_REQUIRE(e && scores && prev && cur_step && out && N > 0 && prev_cols > 0, "predict_with_mask: bad argument");
    if (!e->catalog_set) {
        set_error("predict_with_mask: catalogue not loaded");
        return RL4RS_ESTATE;
    }
    hipLaunchKernelGGL(k_predict_with_mask, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, e->d, N, scores, prev,
                       prev_cols, cur_step, out);
    RL4RS_LAUNCH_CHECK();
    return RL4RS_OK;
}

int rl4rs_env_buffer(rl4rs_env* e, int which, void** p, int64_t* n) {
    RL4RS_REQUIRE(e && p, "env_buffer: null argument");
    const EnvDev& d = e->d;
    int64_t bytes = 0;
    void* ptr = nullptr;
    switch (which) {
        case RL4RS_BUF_PREV_ACTIONS: ptr = e->prev; bytes = (int64_t)d.B * d.T * 4; break;
        case RL4RS_BUF_ACTION_MASK: ptr = e->amask; bytes = (int64_t)d.B * d.W * 4; break;
        case RL4RS_BUF_SPECIAL_MASK: ptr = e->smask; bytes = (int64_t)d.B * d.W * 4; break;
        case RL4RS_BUF_DENSE: ptr = e->dense; bytes = (int64_t)d.B * d.Dn * 4; break;
        case RL4RS_BUF_CATEGORY: ptr = e->cat; bytes = (int64_t)d.B * d.Cn * 4; break;
        case RL4RS_BUF_SEQ0: ptr = e->hist; bytes = (int64_t)d.B * d.L * 4; break;
        case RL4RS_BUF_SEQ1: ptr = e->seq1; bytes = (int64_t)d.B * d.L * 4; break;
        case RL4RS_BUF_C_DENSE: ptr = e->c_dense; bytes = (int64_t)d.B * e->n_complete * d.Dn * 4; break;
        case RL4RS_BUF_C_CATEGORY: ptr = e->c_cat; bytes = (int64_t)d.B * e->n_complete * d.Cn * 4; break;
        case RL4RS_BUF_ERROR_FLAG: ptr = e->err; bytes = 4; break;
        default: set_error("env_buffer: unknown buffer id %d", which); return RL4RS_EINVAL;
    }
    *p = ptr;
    if (n) *n = bytes;
    return RL4RS_OK;
}

}  // extern "C"
