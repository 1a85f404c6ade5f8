/*
 * rl4rs_hip.h — C ABI of librl4rs_hip.so: the MI355X (gfx950) implementation of the RL4RS batched
 * env.step() hot path (boundary row b).
 *
 * The reference is pure Python; it has no FFI.  Each entry point below replaces the Python call a
 * reference maintainer would bind through ctypes; the reference interface it
 * replaces is cited as rl4rs/<file>:<line>.
 *
 * Conventions
 *   - every function returns 0 on success, a negative RL4RS_E* code on failure; the message of the
 *     last failure on the calling thread is available from rl4rs_last_error().
 *   - "dev" pointers are device (HBM) addresses, "host" pointers are host addresses; the caller owns
 *     every buffer it passes in; handles own their internal state.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Calls only enqueue work on
 *     that stream: no allocation, no host synchronisation inside step / forward calls.
 *   - a handle is bound to the HIP device that was current when it was created and is not thread-safe.
 *   - plain C types only: no torch / STL types cross this boundary.
 */
#ifndef RL4RS_HIP_H
#define RL4RS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RL4RS_ABI_VERSION 1

enum {
    RL4RS_OK = 0,
    RL4RS_EINVAL = -1,   /* bad argument / configuration */
    RL4RS_EHIP = -2,     /* a HIP runtime call failed */
    RL4RS_ESTATE = -3,   /* call not valid in the handle's current state (e.g. step past the horizon) */
    RL4RS_ENOMEM = -4
};

const char* rl4rs_last_error(void);
int rl4rs_abi_version(void);
/* number of HIP devices visible; <0 on error.  The library never falls back to a CPU path. */
int rl4rs_device_count(void);

/* Async device-to-device copy on `stream` (lets a host binding snapshot env-owned buffers into
 * caller-owned ones without linking the HIP runtime itself). */
int rl4rs_copy_d2d(void* dst_dev, const void* src_dev, int64_t n_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Env state machine: SlateState / SeqSlateState (rl4rs/env/slate.py:8-214, rl4rs/env/seqslate.py:8-126)
 * ---------------------------------------------------------------------------------------------- */

typedef struct rl4rs_env rl4rs_env;

typedef struct rl4rs_env_cfg {
    int32_t batch_size;            /* config['batch_size']                       slate.py:11 */
    int32_t max_steps;             /* config['max_steps']                        slate.py:14 */
    int32_t action_size;           /* config['action_size']                      slate.py:12 */
    int32_t action_emb_size;       /* config['action_emb_size'] (32)             slate.py:13 */
    int32_t page_items;            /* config['page_items'] (9)                   seqslate.py:11 */
    int32_t item_dim;              /* len(item_vec) (40)                         slate.py:35 */
    int32_t user_dense_dim;        /* len(user_protrait[10:]) (32)               slate.py:78 */
    int32_t user_cat_dim;          /* len(user_protrait[:10]) (10)               slate.py:79 */
    int32_t maxlen;                /* config['maxlen'] (64)                      datautil.py:12 */
    int32_t dense_feature_num;     /* config['dense_feature_num'] (432)          datautil.py:15 */
    int32_t category_feature_num;  /* config['category_feature_num'] (21)        datautil.py:16 */
    int32_t log_steps;             /* columns of exposed_items / user_feedback in the loaded batch */
    int32_t is_seq;                /* 0 = SlateState rules, 1 = SeqSlateState rules */
    int32_t violation_zeroes_reward; /* slate.py:303-307 (always) / seqslate.py:154-157 (mask modes only) */
} rl4rs_env_cfg;

int rl4rs_env_create(const rl4rs_env_cfg* cfg, rl4rs_env** out);
int rl4rs_env_destroy(rl4rs_env* env);

/* Catalogue tables parsed on the host from item_info.csv
 * (SlateState.get_iteminfo_from_file slate.py:28-53, get_mask_from_file slate.py:55-65).
 *   item_vec_host     [action_size, item_dim] float32 (row 0 = zeros)
 *   price_host        [action_size] float64
 *   action_emb_host   [action_size, action_emb_size] float64
 *   is_special_host   [action_size] uint8
 *   location_mask_host[4, action_size] uint8
 */
int rl4rs_env_set_catalog(rl4rs_env* env, const float* item_vec_host, const double* price_host,
                          const double* action_emb_host, const uint8_t* is_special_host,
                          const uint8_t* location_mask_host, void* stream);

/* One sampled batch of log records in columnar form (RecDataBase.sample base.py:92-100 +
 * SlateState.records_to_state slate.py:67-83 + pad_sequences of the history, datautil.py:43-46).
 * Device pointers; copied into the env (async on `stream`).
 *   exposed_dev   [B, log_steps] int32      exposed_items
 *   feedback_dev  [B, log_steps] int32      user_feedback
 *   history_dev   [B, maxlen]    int32      user_seqfeature, pre-padded / pre-truncated
 *   user_dense_dev[B, user_dense_dim] float32
 *   user_cat_dev  [B, user_cat_dim]   int32
 */
int rl4rs_env_load_batch(rl4rs_env* env, const int32_t* exposed_dev, const int32_t* feedback_dev,
                         const int32_t* history_dev, const float* user_dense_dev,
                         const int32_t* user_cat_dev, void* stream);
/* The same as ONE launch when the log is resident on the device (rl4rs_amd LogStore: the sample file parsed once into
 * columnar tables of n_lines rows): gathers the sampled lines line_idx_dev int32 [B] (RecDataBase.sample, base.py:92-100) into
 * the batch buffers, writes the start-of-episode state (prev_actions = 0, both masks all ones; slate.py:8-27) so that the
 * following rl4rs_env_reset only builds the state rows, and - optionally - copies the histories of the batch's n_uniq DISTINCT
 * lines uniq_idx_dev int32 [n_uniq] to hist_unique_dev int32 [n_uniq, maxlen] (what the scorer encodes once per history).
 * store_*: exposed / feedback int32 [n_lines, log_steps], history int32 [n_lines, maxlen], user_dense float32
 * [n_lines, user_dense_dim], user_cat int32 [n_lines, user_cat_dim]. */
int rl4rs_env_load_lines(rl4rs_env* env, const int32_t* store_exposed, const int32_t* store_feedback, const int32_t* store_history,
                         const float* store_user_dense, const int32_t* store_user_cat, int32_t n_lines, int32_t store_log_steps,
                         const int32_t* line_idx_dev, const int32_t* uniq_idx_dev, int32_t n_uniq, int32_t* hist_unique_dev,
                         void* stream);

/* HOST-side parser of a text buffer of '\n'-separated '@'-records (blank lines skipped) into the columnar HOST
 * arrays above: FeatureUtil.record_split (datautil.py:20-32) + records_to_state (slate.py:67-83) + pad_sequences
 * of the history (datautil.py:43-46), for a whole sample file at once.  exposed/feedback are zero padded /
 * truncated to log_steps columns; exposed_len_host (optional) receives each record's true exposed_items count. */
int rl4rs_parse_records(const char* text, int64_t len, int32_t max_records, int32_t maxlen, int32_t log_steps,
                        int32_t user_dense_dim, int32_t user_cat_dim, int32_t* exposed_host, int32_t* feedback_host,
                        int32_t* history_host, float* user_dense_host, int32_t* user_cat_host,
                        int32_t* exposed_len_host, int32_t* n_parsed);

/* HOST-side CRC-32C (Castagnoli) of a byte range, continuing the running checksum `crc` (0 starts one): the
 * checksum of the TFRecord framing (tf.io behind FeatureUtil.to_tfrecord / read_tfrecord, datautil.py:71-230)
 * and of the tensor-bundle checkpoints tf.train.Saver writes (base.py:129,151; supervised_train.py:44-46). */
uint32_t rl4rs_crc32c(const void* data, int64_t len, uint32_t crc);

/* RecState.__init__ (base.py:27-31) + SlateState.__init__ (slate.py:15-19): zero prev_actions, masks
 * to all-ones, cur_steps = 0, feature rows = the un-acted state. */
int rl4rs_env_reset(rl4rs_env* env, void* stream);

/* SlateState.act / SeqSlateState.act with integer actions (slate.py:198-214, seqslate.py:98-126). */
int rl4rs_env_act_discrete(rl4rs_env* env, const int32_t* actions_dev, void* stream);

/* Continuous actions: masked K-NN argmax in float64 (slate.py:187-197, seqslate.py:93-97) followed by
 * act.  `actions_dev` is [B, action_emb_size] float32 (is_f64 = 0) or float64 (is_f64 = 1);
 * `chosen_dev` (optional) receives the item ids that were played. */
int rl4rs_env_act_conti(rl4rs_env* env, const void* actions_dev, int is_f64, int32_t* chosen_dev,
                        void* stream);

/* SlateState.get_nearest_neighbor / get_nearest_neighbor_with_mask (slate.py:180-191, static):
 * [n,E] -> [n].  mask_dev: optional uint8 [n, action_size] (0 = score forced to -2**31). */
int rl4rs_knn(const void* actions_dev, int is_f64, int32_t n, const double* action_emb_dev,
              int32_t action_size, int32_t emb_size, const uint8_t* mask_dev, int32_t* out_dev,
              void* stream);

/* get_complete_states + feature_extraction for the reward rows (slate.py:117-131,289-294;
 * seqslate.py:27-50,141-146): fills the env's complete-state buffers ([B*n, ...], env-major). */
int rl4rs_env_build_complete(rl4rs_env* env, void* stream);
/* Same, emitting only the first rows_per_env (<= complete_rows) rows of each env ([B*rows_per_env, ...]).
 * The LAST complete row of an env equals its current state row (same prev_actions, action = the item just
 * played; slate.py:205-212 vs :119-130), so a caller that already scored the state row can skip it. */
int rl4rs_env_build_complete_rows(rl4rs_env* env, int32_t rows_per_env, void* stream);
/* rows per env produced by build_complete: max_steps (Slate) or page_items (SeqSlate). */
int rl4rs_env_complete_rows(const rl4rs_env* env);
/* 1 when SlateRecEnv.forward / SeqSlateRecEnv.forward would call the reward net now
 * (slate.py:283, seqslate.py:138). */
int rl4rs_env_is_reward_step(const rl4rs_env* env);
int rl4rs_env_cur_steps(const rl4rs_env* env);

/* reward = sum_j price*prob in float64 (numpy pairwise order), zeroed on violation
 * (slate.py:298-307, seqslate.py:150-157).  probs_dev [B, n] float32, reward_dev [B] float64. */
int rl4rs_env_reward(rl4rs_env* env, const float* probs_dev, double* reward_dev, void* stream);

/* Same with the last row's probability supplied separately: probs_dev [B, n-1], p_last_dev [B] (NULL = probs_dev
 * holds all n). */
int rl4rs_env_reward_split(rl4rs_env* env, const float* probs_dev, const float* p_last_dev, double* reward_dev,
                           void* stream);

/* get_violation (slate.py:133-147, seqslate.py:52-69): out_dev [B] int32 in {0,1}. */
int rl4rs_env_violation(rl4rs_env* env, int32_t* out_dev, void* stream);

/* state['action_mask'] = action_mask & location_mask[layer(post-increment cur_steps)] & special_mask
 * (slate.py:92-97, seqslate.py:15-17).  out_dev [B, action_size]; dtype: 0=uint8 1=int32 2=int64 3=float32;
 * dtype 4: packed, out_dev [B, (action_size + 31) / 32] uint32 words, bit k & 31 of word k >> 5 = action k (the mask_bits
 * layout of rl4rs_policy_*) */
int rl4rs_env_obs_mask(rl4rs_env* env, void* out_dev, int dtype, void* stream);

/* offline_action (slate.py:149-162): ids_dev [B] int32, and/or emb_dev [B, E] float64 (conti mode). */
int rl4rs_env_offline_action(rl4rs_env* env, int32_t* ids_dev, double* emb_dev, void* stream);
/* offline_reward (slate.py:164-174, seqslate.py:71-86): out_dev [B] float64. */
int rl4rs_env_offline_reward(rl4rs_env* env, double* out_dev, void* stream);

/* policy_model.predict_with_mask (rl4rs/policy/policy_model.py:17-41; same mask rule as
 * CustomVectorEncoder.forward, rl4rs/nets/cql/encoder.py:42-67): masked argmax where the mask is re-derived
 * from the observation tail.  scores_dev [N, action_size] f32, prev_dev [N, prev_cols] int32 (the obs tail's
 * previous actions), cur_step_dev [N] int32, out_dev [N] int32.  Uses the env's catalogue tables only. */
int rl4rs_env_predict_with_mask(rl4rs_env* env, int32_t N, const float* scores_dev, const int32_t* prev_dev,
                                int32_t prev_cols, const int32_t* cur_step_dev, int32_t* out_dev, void* stream);

/* Kernel-path selection of one env handle (A/B measurements; same results): ROWS_VARIANT 0 [default] = the row kernels stage
 * the catalogue in LDS, 1 = they read it through L1 / L2. */
enum { RL4RS_ENV_OPT_ROWS_VARIANT = 0 };
int rl4rs_env_set_option(rl4rs_env* env, int32_t which, int32_t value);

/* The configuration the env was created with. */
int rl4rs_env_get_cfg(const rl4rs_env* env, rl4rs_env_cfg* out);

/* Device views of env-owned buffers (valid until destroy).  `which`: */
enum {
    RL4RS_BUF_PREV_ACTIONS = 0,   /* int32  [B, max_steps] */
    RL4RS_BUF_ACTION_MASK = 1,    /* uint32 [B, ceil(A/32)] bit k of word k/32 = action_mask[b,k] */
    RL4RS_BUF_SPECIAL_MASK = 2,   /* uint32 [B, ceil(A/32)] */
    RL4RS_BUF_DENSE = 3,          /* float32 [B, dense_feature_num]      feature_extraction()[1] */
    RL4RS_BUF_CATEGORY = 4,       /* int32  [B, category_feature_num]    feature_extraction()[2] */
    RL4RS_BUF_SEQ0 = 5,           /* int32  [B, maxlen]                  sequence 0 (history)     */
    RL4RS_BUF_SEQ1 = 6,           /* int32  [B, maxlen]                  sequence 1               */
    RL4RS_BUF_C_DENSE = 7,        /* float32 [B*n, dense_feature_num]    complete-state rows      */
    RL4RS_BUF_C_CATEGORY = 8,     /* int32  [B*n, category_feature_num] */
    RL4RS_BUF_ERROR_FLAG = 9      /* int32  [1]  sticky: 1 = an action id outside [0, action_size) */
};
int rl4rs_env_buffer(rl4rs_env* env, int which, void** dev_ptr, int64_t* n_bytes);

/* ------------------------------------------------------------------------------------------------
 * DIEN simulator net: rl4rs/nets/dien.py:8-45, rl4rs/nets/utils.py:16-25,48-54,100-129,
 * called from SlateRecEnv.obs_fn / forward (slate.py:265-267, 295-297) as obs_layer / reward_layer.
 * ---------------------------------------------------------------------------------------------- */

typedef struct rl4rs_dien rl4rs_dien;

typedef struct rl4rs_dien_cfg {
    int32_t maxlen;                /* 64  */
    int32_t emb_size;              /* 128 */
    int32_t hidden_units;          /* 128 */
    int32_t dense_feature_num;     /* 432 */
    int32_t category_feature_num;  /* 21  */
    int32_t category_hash_size;    /* 100000 */
    int32_t seq_num;               /* 2   */
    int32_t class_num;             /* 2   */
    int32_t max_rows;              /* largest R passed to rl4rs_dien_forward */
    int32_t max_slots;             /* sequence-cache slots per sequence input */
    int32_t scorer_mode;           /* RL4RS_SCORER_* : arithmetic of the AUGRU recurrence (the dominant kernel) */
    uint32_t kernel_opts;          /* RL4RS_DIEN_OPT_* bits, 0 = the defaults */
} rl4rs_dien_cfg;

/* Kernel-path selection of ONE scorer handle (A/B measurements, and parity tests that pin a path): configuration, never
 * process environment, so that handles with different paths can live side by side.  Every combination computes the same
 * function inside the parity bars of DESIGN.md 2; forms marked (=) are bit-identical to the default.
 *   AUGRU_H16       first-generation fp16x2 recurrence k_augru_h16 instead of k_augru_x
 *   AUGRU_ROWS32    k_augru_x: 32-row workgroups for every launch (=)
 *   AUGRU_ROWS64    k_augru_x: 64-row workgroups whenever the launch shape admits them (R % 64 == 0, rows in whole groups
 *                   of 8 per cache slot), not only when that still fills the chip twice over (=); see rl4rs_dien_set_augru_rows
 *   DIN_V1          first-generation attention-score kernel k_din_scores instead of k_din_x
 *   NO_DIN16 / NO_GRU16 / NO_GEMM16 / NO_CAT16   keep the exact-fp32 MFMA form of the attention MLP / first GRU / plain GEMMs /
 *                   category self-attention in fp16x2 mode
 *   NO_DENSE_CHAIN  dense tower as two GEMM launches instead of one chained launch (=)
 *   NO_HEAD_TABLES  head GEMM over all 3456 inputs instead of the per-category-slot tables
 *   NO_HEAD_FUSED   k_head_finish as its own launch instead of the table sums inside k_cat_attn
 *   CAT_V1          category branch through the first-generation k_cat_attn (full [Cn, E] LDS image per row: 12 rows in flight
 *                   per CU) instead of k_cat_attn2 (half-K image, 16+ rows per CU; E = 128 and Cn <= 24 only)
 *   NO_CAT_GROUP    launches whose rows come in groups of 8 or 9 per cache slot (the reward forward over the complete states):
 *                   category branch per row (k_cat_attn2) instead of one workgroup per group that gathers the category rows the
 *                   group shares once (k_cat_attn2g; a group that shares nothing takes the per-row body inside it) (=) */
enum {
    RL4RS_DIEN_OPT_AUGRU_H16 = 1 << 0,
    RL4RS_DIEN_OPT_AUGRU_ROWS32 = 1 << 1,
    RL4RS_DIEN_OPT_AUGRU_ROWS64 = 1 << 2,
    RL4RS_DIEN_OPT_DIN_V1 = 1 << 3,
    RL4RS_DIEN_OPT_NO_DIN16 = 1 << 4,
    RL4RS_DIEN_OPT_NO_GRU16 = 1 << 5,
    RL4RS_DIEN_OPT_NO_GEMM16 = 1 << 6,
    RL4RS_DIEN_OPT_NO_CAT16 = 1 << 7,
    RL4RS_DIEN_OPT_NO_DENSE_CHAIN = 1 << 8,
    RL4RS_DIEN_OPT_NO_HEAD_TABLES = 1 << 9,
    RL4RS_DIEN_OPT_NO_HEAD_FUSED = 1 << 10,
    RL4RS_DIEN_OPT_CAT_V1 = 1 << 11,
    RL4RS_DIEN_OPT_NO_CAT_GROUP = 1 << 12,
    RL4RS_DIEN_OPT_DENSE_FORK = 1 << 13,       /* the dense tower on a second stream of the handle, joined in front of the head GEMM */
    RL4RS_DIEN_OPT_NO_GRU_PAD = 1 << 14,       /* first GRU: compute the steps on leading zero ids (front padding) per row instead of
                                                  taking them from the handle's table of pad states (bit-identical either way) */
    RL4RS_DIEN_OPT_ALL = (1 << 15) - 1
};

/* Every mode accumulates in fp32 and meets the fp32 parity bar against the fp64 oracle (same measured error):
 *   FP32   v_mfma_f32_32x32x2_f32, operands exact.
 *   FP16X2 operands split into fp16 hi + lo, 3 v_mfma_f32_32x32x16_f16 per product (hi*hi + hi*lo + lo*hi):
 *          ~2^-22 relative error per product, 2.3x faster.  No weight-range condition: a weight tile that would not fit
 *          fp16 (max |w| >= 2^14) is stored times a power of two and the accumulator divided by it, exactly (32-column tiles
 *          of the recurrent matrices, columns of the plain GEMMs).  Only NON-FINITE weights need FP32.  ACTIVATIONS are
 *          range-checked at run time: a row whose state / GEMM input leaves the fp16 range comes back NaN + status bit.
 *   AUTO   FP16X2 for every finite checkpoint, else FP32. */
enum { RL4RS_SCORER_AUTO = 0, RL4RS_SCORER_FP32 = 1, RL4RS_SCORER_FP16X2 = 2 };

/* Host pointers to float32 arrays, shapes in rl4rs_amd/nets/dien.py (dien_spec). seq arrays have
 * seq_num entries (max 4). */
typedef struct rl4rs_dien_weights {
    const float* cat_emb;
    const float* dense_w1; const float* dense_b1;
    const float* dense_w2; const float* dense_b2;
    const float* seq_emb;
    const float* gru_gate_w[4];   const float* gru_gate_b[4];
    const float* gru_cand_w[4];   const float* gru_cand_b[4];
    const float* att_w1[4]; const float* att_b1[4];
    const float* att_w2[4]; const float* att_b2[4];
    const float* att_w3[4]; const float* att_b3[4];
    const float* augru_gate_w[4]; const float* augru_gate_b[4];
    const float* augru_cand_w[4]; const float* augru_cand_b[4];
    const float* obs_w; const float* obs_b;
    const float* out_w; const float* out_b;
} rl4rs_dien_weights;

int rl4rs_dien_create(const rl4rs_dien_cfg* cfg, const rl4rs_dien_weights* w, void* stream,
                      rl4rs_dien** out);
int rl4rs_dien_destroy(rl4rs_dien* net);
/* The mode the handle resolved to (RL4RS_SCORER_FP32 or RL4RS_SCORER_FP16X2). */
int rl4rs_dien_scorer_mode(rl4rs_dien* net, int32_t* mode);
/* Row-tile form of k_augru_x for the following forwards of this handle: 0 automatic, 32, 64 (the AUGRU_ROWS* options above,
 * switchable on a live handle so that one cache can be scored by both forms). */
int rl4rs_dien_set_augru_rows(rl4rs_dien* net, int32_t rows);
/* Status bits since the last call (synchronises `stream`, then clears them).  RL4RS_DIEN_STATUS_FP16_RANGE: in
 * FP16X2 mode a recurrent state left the fp16 range (|h| >= 6e4, or NaN) - possible only when the attention scores
 * push the update gate outside [0, 1] until the state diverges; results of the affected forwards are invalid, use
 * RL4RS_SCORER_FP32 for such a model. */
enum { RL4RS_DIEN_STATUS_FP16_RANGE = 1 };
int rl4rs_dien_status(rl4rs_dien* net, int32_t* flags, void* stream);
/* The status word itself: device pointer to ONE int32 owned by the handle, != 0 <=> RL4RS_DIEN_STATUS_FP16_RANGE pending.  A
 * caller that copies a record to the host every step anyway (rl4rs_env_step_record) reads it there. */
int rl4rs_dien_status_word(rl4rs_dien* net, int32_t** word_dev);

/* Encode `n` id sequences of sequence input `s` into cache slots [slot_base, slot_base+n):
 * embedding lookup + first GRU over all maxlen steps (utils.py:119-120) and the input-side
 * projections of the attention MLP / AUGRU that depend only on it.  ids_dev [n, maxlen] int32.
 * Front padding (rl4rs/utils/datautil.py:44: pad_sequences pads in FRONT): the states after k leading zero ids are the same for
 * every sequence, so (FP16X2 mode, unless RL4RS_DIEN_OPT_NO_GRU_PAD) the handle keeps them in a table and in one extra cache slot
 * behind max_slots; encode starts every 32-row block at its shortest zero prefix and records each slot's prefix length, and the
 * forward's AUGRU / DIN kernels read the steps of a row's prefix from that one slot.  Bit-identical to computing them per row. */
int rl4rs_dien_encode(rl4rs_dien* net, int32_t s, const int32_t* ids_dev, int32_t n,
                      int32_t slot_base, void* stream);

/* Forward R rows (obs_layer / reward_layer, slate.py:265-267,295-297).
 *   dense_dev [R, dense_feature_num] f32, cat_dev [R, category_feature_num] i32
 *   slot_dev  [seq_num, R/group] int32: cache slot of sequence input s for each group of `group`
 *             consecutive rows (rows of one env share their sequences)
 *   obs_dev   [R, 256] f32  'simulator_obs' activations (may be NULL)
 *   prob_dev  [R] f32       softmax('simulator_reward')[:, 1] (may be NULL)
 */
int rl4rs_dien_forward(rl4rs_dien* net, int32_t R, int32_t group, const float* dense_dev,
                       const int32_t* cat_dev, const int32_t* slot_dev, float* obs_dev,
                       float* prob_dev, void* stream);

/* softmax(obs @ out_w + out_b)[:, 1] of already computed 'simulator_obs' activations (dien.py:36):
 * obs_dev [R, 256] -> prob_dev [R]. */
int rl4rs_dien_head_prob(rl4rs_dien* net, int32_t R, const float* obs_dev, float* prob_dev, void* stream);

/* Intermediate activations for parity tests; `which`: */
enum {
    RL4RS_DIEN_ALL_FEATURE = 0,   /* float32 [max_rows, ld] concat input of simulator_obs; ld = n_bytes / (4 max_rows):
                                     2E*seq_num + U + (Cn+1)E, or only the first 2E*seq_num + U + E columns when the
                                     Flatten(category_emb) part lives in the head tables */
    RL4RS_DIEN_SCORES = 1,        /* float32 [seq_num, max_rows, maxlen] attention scores */
    RL4RS_DIEN_QUERY = 2,         /* float32 [max_rows, E] */
    RL4RS_DIEN_H1 = 3             /* float32 [seq_num, max_slots, maxlen, E] first-GRU states */
};
int rl4rs_dien_buffer(rl4rs_dien* net, int which, void** dev_ptr, int64_t* n_bytes);

/* Processing order of the row groups (envs) of the following forwards: order_dev[i] = group handled at position i, a
 * permutation of 0..n_groups-1 in caller-owned device memory (NULL = identity).  A locality hint only (results per row are
 * unchanged): envs sorted by the cache slot of their history make duplicate histories share L2 lines
 * (RecDataBase.sample draws with replacement, rl4rs/env/base.py:92-100).  Applies when n_groups == R / group. */
int rl4rs_dien_set_row_order(rl4rs_dien* net, const int32_t* order_dev, int32_t n_groups);

/* Per-kernel HIP-event timing (bench.py roofline).  With profiling enabled every kernel class launched
 * by rl4rs_dien_encode / rl4rs_dien_forward is bracketed by an event pair on the caller's stream.
 * rl4rs_dien_profile_read synchronises on the recorded pairs and returns the cumulative milliseconds
 * and launch count of kernel class `which` since the last reset.
 * enable: 0 off; 1 every kernel class (~0.4 ms of event records per episode-batch of the bench workload: a breakdown
 * pass); 2 only the AUGRU recurrence, the dominant kernel (two records per forward: what a timed region can carry). */
int rl4rs_dien_set_profiling(rl4rs_dien* net, int enable);
int rl4rs_dien_kernel_count(void);
const char* rl4rs_dien_kernel_name(int which);
/* The kernel(s) THIS handle launches for class `which`, named as a rocprofv3 kernel trace shows them (they depend on the
 * scorer mode and the handle's kernel_opts; rl4rs_dien_kernel_name is the static class name).  NUL-terminated into buf[cap]. */
int rl4rs_dien_kernel_label(rl4rs_dien* net, int which, char* buf, int32_t cap);
int rl4rs_dien_profile_read(rl4rs_dien* net, int which, double* ms_total, int64_t* launches);
int rl4rs_dien_profile_reset(rl4rs_dien* net);

/* ------------------------------------------------------------------------------------------------
 * The reference's other simulator families behind the same 'simulator_obs' / 'simulator_reward' contract
 * (SlateRecEnv.get_model, rl4rs/env/slate.py:228-241: config['algo'] = 'dnn' | 'widedeep' | 'lstm'):
 *   DNN       rl4rs/nets/dnn.py:31-37       obs_dim 256
 *   WIDEDEEP  rl4rs/nets/widedeep.py:31-38  obs_dim 256 + hidden_units + category_feature_num * emb_size
 *   LSTM      rl4rs/nets/lstm.py:31-37      obs_dim 256 (keras GRU layers; needs emb_size == hidden_units == 128)
 * Same calling pattern as rl4rs_dien: `encode` caches the per-sequence feature of a cache slot (WIDEDEEP: mean of
 * the sequence embeddings, LSTM: final GRU state, DNN: nothing - that model never reads its sequences), `forward`
 * scores R rows whose groups of `group` consecutive rows share their slots.
 * Weight arrays (host float32, shapes in rl4rs_amd/nets/simnets.py); unused ones may be NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rl4rs_simnet rl4rs_simnet;
enum { RL4RS_SIMNET_DNN = 1, RL4RS_SIMNET_WIDEDEEP = 2, RL4RS_SIMNET_LSTM = 3 };

typedef struct rl4rs_simnet_cfg {
    int32_t algo;                  /* RL4RS_SIMNET_* */
    int32_t maxlen;                /* 64  */
    int32_t emb_size;              /* 128 */
    int32_t hidden_units;          /* 128 */
    int32_t dense_feature_num;     /* 432 */
    int32_t category_feature_num;  /* 21  */
    int32_t category_hash_size;    /* 100000 */
    int32_t seq_num;               /* 2   */
    int32_t class_num;             /* 2   */
    int32_t max_rows;
    int32_t max_slots;
} rl4rs_simnet_cfg;

typedef struct rl4rs_simnet_weights {
    const float* cat_emb;
    const float* seq_emb;
    const float* dense_w1; const float* dense_b1;
    const float* dense_w2; const float* dense_b2;
    const float* fc_w; const float* fc_b;
    const float* obs_w; const float* obs_b;
    const float* out_w; const float* out_b;
    const float* cat_gru_kernel; const float* cat_gru_recurrent; const float* cat_gru_bias;
    const float* seq_gru_kernel[4]; const float* seq_gru_recurrent[4]; const float* seq_gru_bias[4];
} rl4rs_simnet_weights;

int rl4rs_simnet_create(const rl4rs_simnet_cfg* cfg, const rl4rs_simnet_weights* w, void* stream,
                        rl4rs_simnet** out);
int rl4rs_simnet_destroy(rl4rs_simnet* net);
int rl4rs_simnet_obs_dim(rl4rs_simnet* net, int32_t* dim);
/* ids_dev [n, maxlen] int32 -> cache slots [slot_base, slot_base + n) of sequence input s */
int rl4rs_simnet_encode(rl4rs_simnet* net, int32_t s, const int32_t* ids_dev, int32_t n, int32_t slot_base,
                        void* stream);
/* dense_dev [R, dense_feature_num] f32, cat_dev [R, category_feature_num] i32, slot_dev [seq_num, R/group] i32,
 * obs_dev [R, obs_dim] f32 (may be NULL), prob_dev [R] f32 = softmax('simulator_reward')[:, 1] (may be NULL) */
int rl4rs_simnet_forward(rl4rs_simnet* net, int32_t R, int32_t group, const float* dense_dev,
                         const int32_t* cat_dev, const int32_t* slot_dev, float* obs_dev, float* prob_dev,
                         void* stream);
/* softmax(obs @ out_w + out_b)[:, 1] of already computed 'simulator_obs' rows */
int rl4rs_simnet_head_prob(rl4rs_simnet* net, int32_t R, const float* obs_dev, float* prob_dev, void* stream);

/* ------------------------------------------------------------------------------------------------
 * One batched transition as ONE call: RecSimBase._step (rl4rs/env/base.py:157-170) = act -> obs_fn -> forward (reward
 * when due) -> done, for an env bound to its scorer.  Same kernels in the same order as calling rl4rs_env_act_* /
 * rl4rs_dien_forward / rl4rs_env_build_complete_rows / rl4rs_dien_head_prob / rl4rs_env_reward_split one by one
 * (bit-identical results); nothing is allocated, nothing synchronises inside.
 *   slots_dev  int32 [seq_num, batch_size]: cache slot of every env row per sequence input (caller-owned; the caller
 *              encodes the history sequences with rl4rs_dien_encode after every rl4rs_env_load_batch and keeps row 0
 *              current; SeqSlate's second input is re-encoded here on the first act of a page, seqslate.py:107-108)
 *   obs_dev    float32 [B, D]    'simulator_obs' of the new state (slate.py:265-267); D = 256 for the DIEN scorer,
 *              rl4rs_simnet_obs_dim() for an attached simnet (widedeep: 256 + hidden_units + Cn * emb_size)
 *   reward_dev float64 [B] (optional)  0 unless a reward is due (slate.py:283, seqslate.py:138)
 *   done_dev   uint8 [B] (optional)    1 once cur_steps (before the act) >= max_steps - 1 (base.py:165-168)
 *   mask_bits_dev uint32 [B, ceil(A/32)] (optional)  obs-side action mask of the NEXT slot (slate.py:92-97), packed as
 *              rl4rs_env_obs_mask dtype 4
 * ---------------------------------------------------------------------------------------------- */
typedef struct rl4rs_stepper rl4rs_stepper;
int rl4rs_env_attach_scorer(rl4rs_env* env, rl4rs_dien* net, const int32_t* slots_dev, int32_t seq_num,
                            rl4rs_stepper** out);
int rl4rs_env_attach_simnet(rl4rs_env* env, rl4rs_simnet* net, const int32_t* slots_dev, int32_t seq_num,
                            rl4rs_stepper** out);
int rl4rs_stepper_destroy(rl4rs_stepper* s);
int rl4rs_env_step_discrete(rl4rs_stepper* s, const int32_t* actions_dev, float* obs_dev, double* reward_dev,
                            uint8_t* done_dev, uint32_t* mask_bits_dev, void* stream);
/* continuous actions: [B, action_emb_size] float32 / float64 resolved by the masked float64 K-NN first (slate.py:187-197);
 * chosen_dev (optional) receives the item ids played */
int rl4rs_env_step_conti(rl4rs_stepper* s, const void* actions_dev, int is_f64, int32_t* chosen_dev, float* obs_dev,
                         double* reward_dev, uint8_t* done_dev, uint32_t* mask_bits_dev, void* stream);

/* Reference-shaped form of the same transition (RecEnvBase.step hands lists / ndarrays to the caller, base.py:256-263,
 * slate.py:244-279): every output a host-returning caller needs is written into ONE device record, host-visible part first,
 * so the facade brings a whole transition back with a single device-to-host copy into pinned memory and ONE wait.
 * `want` selects the optional parts; offsets are bytes from the start of the record, -1 = absent; all parts 64-byte aligned.
 *   status          int32 [2]: [0] sticky bad-action flag of the env (RL4RS_BUF_ERROR_FLAG), [1] the scorer's fp16-range status
 *                   (RL4RS_DIEN_STATUS_FP16_RANGE; read and cleared)
 *   reward          float64 [B]                      done   uint8 [B]          chosen   int32 [B] item ids played
 *   obs             float32 [B, obs_dim]             'simulator_obs'  (behind host_bytes when D3RL_OBS is wanted)
 *   obs_d3rl        float64 [B, obs_dim + cols + 1]  support_d3rl_mask observation: obs | masked_actions | cur_steps
 *                   (slate.py:270-277; cols = max_steps, or page_items for SeqSlate: seqslate.py:18-23)
 *   mask_i64        int64 [B, action_size]           support_rllib_mask observation-side mask (slate.py:90-97)
 *   mask_bits       uint32 [B, ceil(A/32)]           the same mask packed
 *   click_p         float32 [B, n_complete]          simulator_info_fetch (slate.py:299-301); written on reward steps only
 *   offline_action  int32 [B] (discrete) / float64 [B, action_emb_size] (continuous): the logged action of the NEXT step
 *                   (slate.py:152-161); written while a next step exists
 * host_bytes: length of the prefix a host caller copies; total_bytes: size of the record buffer to allocate. */
enum {
    RL4RS_STEP_WANT_MASK_I64 = 1, RL4RS_STEP_WANT_MASK_BITS = 2, RL4RS_STEP_WANT_D3RL_OBS = 4, RL4RS_STEP_WANT_CLICK_P = 8,
    RL4RS_STEP_WANT_OFFLINE_ACTION = 16, RL4RS_STEP_WANT_ALL = 31
};
typedef struct rl4rs_step_record {
    int64_t status, reward, done, chosen, obs, obs_d3rl, mask_i64, mask_bits, click_p, offline_action;
    int64_t host_bytes, total_bytes;
    int32_t obs_dim, d3rl_cols;
} rl4rs_step_record;
int rl4rs_stepper_record_layout(rl4rs_stepper* s, uint32_t want, int32_t conti, rl4rs_step_record* out);
/* action_kind: 0 = int32 item ids [B]; 1 / 2 = float32 / float64 action embeddings [B, action_emb_size] (masked K-NN first) */
int rl4rs_env_step_record(rl4rs_stepper* s, const void* actions_dev, int32_t action_kind, uint32_t want, void* record_dev,
                          void* stream);
/* The same transition, and the record's host part brought home by the library: `record_host` = host_bytes of (pinned) host
 * memory, filled when `stream` has drained - ONE wait on `stream` by the caller.  The int64 mask (the bulk of a
 * support_rllib_mask record, B * action_size * 8 bytes, and a function of the act alone) leaves on a copy stream of the
 * stepper as soon as the act is done, beside the scorer's kernels; on a reward step the observation follows it as soon as the
 * observation forward is done, beside the reward forward; the rest is copied on `stream` after the last kernel. */
int rl4rs_env_step_record_host(rl4rs_stepper* s, const void* actions_dev, int32_t action_kind, uint32_t want, void* record_dev,
                               void* record_host, void* stream);
/* The record of the state the env is IN, without a transition - what RecSimBase.sample returns right after a reset
 * (base.py:172-175: obs_fn(samples.state)): obs / obs_d3rl / mask_i64 / mask_bits / offline_action (the logged action of the
 * CURRENT step) / status in the same layout (`conti` selects the offline_action form); reward, done and chosen are left alone.
 * The caller has encoded the batch's sequences (rl4rs_dien_encode).  record_host may be NULL (no copies). */
int rl4rs_env_observe_record_host(rl4rs_stepper* s, int32_t conti, uint32_t want, void* record_dev, void* record_host,
                                  void* stream);
/* On steps without a reward forward (and on resets) the float32 observation reaches `record_host` from the epilogue of the head
 * GEMM itself (device-visible pinned memory: hipHostGetDevicePointer must succeed on record_host, else the copy engine serves it
 * as before) - the copy would otherwise start only when the GPU has nothing left to run.  Process-wide switch, default 0: on the
 * boxes measured the mirror is no faster than the copy engine (13.2 against 13.1 ms per episode-batch); kept for A/B
 * measurements.  Replaces nothing in the reference (its observations never leave host memory: rl4rs/env/slate.py:244-279). */
int rl4rs_set_host_mirror(int32_t on);

/* ------------------------------------------------------------------------------------------------
 * Action-masked policy net: rl4rs/nets/rllib/rllib_mask_model.py:7-64 (FC obs->hidden(tanh)->action_size
 * logits, value head on the shared hidden layer, logits + max(log(action_mask), float32.min)).
 * Parameters, gradients and Adam state are ONE flat float32 buffer each:
 *   [ W1 (obs_dim x hidden) | b1 (hidden) | W2e (hidden x (action_size+1)) | b2e (action_size+1) ]
 * (column action_size of layer 2 is the value head), so a data-parallel trainer all-reduces one buffer.
 * mask_bits_dev: uint32 [N, ceil(action_size/32)] bit rows in the layout of RL4RS_BUF_ACTION_MASK, or NULL.
 * ---------------------------------------------------------------------------------------------- */
typedef struct rl4rs_policy rl4rs_policy;

int rl4rs_policy_param_count(int32_t obs_dim, int32_t hidden, int32_t action_size);
int rl4rs_policy_create(int32_t obs_dim, int32_t hidden, int32_t action_size, int32_t max_rows,
                        const float* params_host, void* stream, rl4rs_policy** out);
int rl4rs_policy_destroy(rl4rs_policy* pol);
/* device pointer of the flat parameter buffer (owned by the handle) */
int rl4rs_policy_params(rl4rs_policy* pol, float** params_dev, int32_t* count);

/* Sample actions ~ Categorical(softmax(masked logits)) by Gumbel-max with a counter-based RNG keyed on
 * (seed, step, row, action); outputs (each may be NULL except actions): log-prob of the draw, value, entropy,
 * masked logits [N, action_size]. */
int rl4rs_policy_act(rl4rs_policy* pol, int32_t N, const float* obs_dev, const uint32_t* mask_bits_dev,
                     uint32_t seed, uint32_t step, int32_t* actions_dev, float* logp_dev, float* value_dev,
                     float* entropy_dev, float* logits_dev, void* stream);
/* Same forward for GIVEN actions (no sampling). */
int rl4rs_policy_evaluate(rl4rs_policy* pol, int32_t N, const float* obs_dev, const uint32_t* mask_bits_dev,
                          const int32_t* actions_dev, float* logp_dev, float* value_dev, float* entropy_dev,
                          float* logits_dev, void* stream);

/* Loss + gradient over N samples into grad_dev (flat, same layout as the parameters).
 *   algo 0 = A2C (RLlib a3c_tf_policy: -sum(logp*adv) + vf_coeff*0.5*sum((V-R)^2) - ent_coeff*sum(H))
 *   algo 1 = PPO (RLlib ppo_tf_policy: mean(-min(adv*r, adv*clip(r,1-c,1+c)) + kl_coeff*KL(old||new)
 *                 + vf_coeff*max((V-R)^2, (Vclip-R)^2) - ent_coeff*H)); needs old_logp / old_value / old_logits
 * stats_dev (optional) float[4] = sums over samples of {policy loss, value loss, entropy, kl}.
 * Gradients are bit-reproducible (fixed sample chunks, fixed summation order). */
int rl4rs_policy_loss_grad(rl4rs_policy* pol, int32_t algo, int32_t N, const float* obs_dev,
                           const uint32_t* mask_bits_dev, const int32_t* actions_dev, const float* adv_dev,
                           const float* ret_dev, const float* old_logp_dev, const float* old_value_dev,
                           const float* old_logits_dev, float vf_coeff, float ent_coeff, float clip,
                           float vf_clip, float kl_coeff, float* grad_dev, float* stats_dev, void* stream);
/* Adam update of the handle's parameters from grad_dev (tf.train.AdamOptimizer form); grad_clip > 0 applies
 * tf.clip_by_global_norm first. */
int rl4rs_policy_adam_step(rl4rs_policy* pol, const float* grad_dev, float lr, float beta1, float beta2,
                           float eps, float grad_clip, void* stream);
/* One PPO SGD pass over N already shuffled samples in minibatches of `minibatch` consecutive rows (loss + backward
 * + Adam per minibatch; the trailing N % minibatch rows are dropped).  Identical arithmetic to
 * rl4rs_policy_loss_grad(algo 1) + rl4rs_policy_adam_step per minibatch, one host call (one persistent kernel when the
 * shapes fit and its grid can be co-resident on this device, checked against the runtime's occupancy answer).
 * stats_dev (optional) float[8]: [0..3] = sums of {policy loss, value loss, entropy, kl} over the LAST minibatch,
 * [4..7] = the same sums over every sample of the pass (RLlib reports the pass-mean KL and feeds it to its adaptive
 * kl_coeff rule: script/modelfree_train.py:189 kl_coeff, :216 kl_target). */
int rl4rs_policy_ppo_epoch(rl4rs_policy* pol, int32_t N, int32_t minibatch, const float* obs_dev,
                           const uint32_t* mask_bits_dev, const int32_t* actions_dev, const float* adv_dev,
                           const float* ret_dev, const float* old_logp_dev, const float* old_value_dev,
                           const float* old_logits_dev, float vf_coeff, float ent_coeff, float clip, float vf_clip,
                           float kl_coeff, float lr, float beta1, float beta2, float eps, float grad_clip,
                           float* grad_dev, float* stats_dev, void* stream);
/* Data-parallel form of the pass (gradient all-reduce only): the PPO gradient of minibatch `mb_index`
 * (rows [mb_index*minibatch, (mb_index+1)*minibatch) of the N shuffled samples) into grad_dev WITHOUT updating the
 * parameters.  Each rank calls it on its own shard, mean-all-reduces grad_dev and applies rl4rs_policy_adam_step.
 * stats_dev (optional) float[4] = sums over the minibatch. */
int rl4rs_policy_ppo_minibatch_grad(rl4rs_policy* pol, int32_t N, int32_t minibatch, int32_t mb_index,
                                    const float* obs_dev, const uint32_t* mask_bits_dev, const int32_t* actions_dev,
                                    const float* adv_dev, const float* ret_dev, const float* old_logp_dev,
                                    const float* old_value_dev, const float* old_logits_dev, float vf_coeff,
                                    float ent_coeff, float clip, float vf_clip, float kl_coeff, float* grad_dev,
                                    float* stats_dev, void* stream);
/* Status bits since the last call (synchronises `stream`, then clears them).  RL4RS_POLICY_STATUS_PASS_TIMEOUT: a grid
 * barrier of a persistent PPO pass timed out (workgroups not co-resident: another process holds compute units); the pass
 * stopped at the last completed minibatch and is incomplete. */
enum { RL4RS_POLICY_STATUS_PASS_TIMEOUT = 1 };
int rl4rs_policy_status(rl4rs_policy* pol, int32_t* flags, void* stream);
/* The same status word without a synchronisation: device pointer to 2 x uint32 owned by the handle, word [1] != 0 <=> a
 * persistent pass timed out.  A training loop copies it asynchronously together with its loss statistics and looks at it one
 * iteration later (rl4rs_amd/train.py), so that validating every pass costs no pipeline drain. */
int rl4rs_policy_status_words(rl4rs_policy* pol, uint32_t** words_dev);
/* Once a pass has timed out, rl4rs_policy_ppo_epoch / rl4rs_policy_ppo_minibatch_grad return RL4RS_ESTATE (nothing is
 * launched, the Adam step counter does not advance) until rl4rs_policy_status has reported and cleared the condition: the
 * kernel raises a pinned host word next to the device flag, which the launch path reads without synchronising.
 *
 * Kernel-path selection of ONE handle, for A/B measurements and tests (defaults in brackets):
 *   TILE          [1] 0 = one-wave-per-sample forward / loss kernels instead of k_policy_tile
 *   PPO_FUSED     [1] 0 = per-minibatch kernel chain instead of the persistent k_ppo_pass
 *   PPO_ROWS      [automatic] samples per workgroup of k_ppo_pass: 8, 16 or 32 pin the all-runtime instantiation's (automatic: 8);
 *                     where the compile-time instantiation applies: 4 or 8 (automatic: 4 while MB / 4 <= 126 workgroups, else 8)
 *   RESIDENT_WGS [-1] >= 0: pretend the device holds only this many workgroups of k_ppo_pass at once (co-residency tests)
 *   PPO_STD       [1] 0 = the all-runtime instantiation of k_ppo_pass even at the default shape (256 -> 64 -> 284 + 1, 8 rows,
 *                     minibatch % 256 == 0), which otherwise runs the compile-time one (bit-identical results) */
enum { RL4RS_POLICY_OPT_TILE = 0, RL4RS_POLICY_OPT_PPO_FUSED = 1, RL4RS_POLICY_OPT_PPO_ROWS = 2, RL4RS_POLICY_OPT_RESIDENT_WGS = 3,
       RL4RS_POLICY_OPT_PPO_STD = 4 };
int rl4rs_policy_set_option(rl4rs_policy* pol, int32_t which, int32_t value);
/* Adam state of the handle (first / second moments, device pointers owned by the handle; same layout as the
 * parameters) and its step counter: a data-parallel trainer broadcasts rank 0's at start, a checkpoint saves them. */
int rl4rs_policy_adam_state(rl4rs_policy* pol, float** m_dev, float** v_dev, int64_t* step);
int rl4rs_policy_set_adam_step(rl4rs_policy* pol, int64_t step);

/* Raw-state policy encoder: rl4rs/nets/rllib/rllib_rawstate_model.py:25-86 (and its action-mask wrapper,
 * rllib_mask_model.py:67-115) for envs with config['rawstate_as_obs'] (rl4rs/env/slate.py:250-262):
 *   context = ELU([mean seq emb (per sequence, one shared table) | dense tower | mean category emb] @ ctx_w + ctx_b)  (256)
 *   logits  = context @ out_w + out_b (+ max(log(mask), float32.min)),   value = context @ value_w + value_b
 * Forward only (act / evaluate with the outputs of rl4rs_policy_act / rl4rs_policy_evaluate); shapes of the host
 * float32 weights in rl4rs_amd/nets/rawpolicy.py.  seq_dev: seq_num device pointers, each int32 [N, maxlen]. */
typedef struct rl4rs_rawpolicy rl4rs_rawpolicy;
typedef struct rl4rs_rawpolicy_cfg {
    int32_t maxlen, emb_size, hidden_units, dense_feature_num, category_feature_num, category_hash_size, seq_num,
        action_size, max_rows;
} rl4rs_rawpolicy_cfg;
typedef struct rl4rs_rawpolicy_weights {
    const float* cat_emb; const float* seq_emb;
    const float* dense_w1; const float* dense_b1; const float* dense_w2; const float* dense_b2;
    const float* ctx_w; const float* ctx_b;
    const float* out_w; const float* out_b;
    const float* value_w; const float* value_b;
} rl4rs_rawpolicy_weights;
int rl4rs_rawpolicy_create(const rl4rs_rawpolicy_cfg* cfg, const rl4rs_rawpolicy_weights* w, void* stream,
                           rl4rs_rawpolicy** out);
int rl4rs_rawpolicy_destroy(rl4rs_rawpolicy* pol);
int rl4rs_rawpolicy_act(rl4rs_rawpolicy* pol, int32_t N, const int32_t* cat_dev, const float* dense_dev,
                        const int32_t* const* seq_dev, const uint32_t* mask_bits_dev, uint32_t seed, uint32_t step,
                        int32_t* actions_dev, float* logp_dev, float* value_dev, float* entropy_dev, float* logits_dev,
                        void* stream);
int rl4rs_rawpolicy_evaluate(rl4rs_rawpolicy* pol, int32_t N, const int32_t* cat_dev, const float* dense_dev,
                             const int32_t* const* seq_dev, const uint32_t* mask_bits_dev, const int32_t* actions_dev,
                             float* logp_dev, float* value_dev, float* entropy_dev, float* logits_dev, void* stream);

/* Trainable raw-state policy: the model of rl4rs_rawpolicy with RLlib's A2C / PPO losses (rl4rs_policy_loss_grad semantics),
 * backward through both heads, the context layer, the dense tower and the mean-pooled embeddings, Adam.  Flat parameter /
 * gradient layout: [ cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | ctx_w | ctx_b |
 *                    head_w 256 x (A+1) = [out_w | value_w] | head_b (A+1) ].  act / evaluate as rl4rs_rawpolicy_*. */
typedef struct rl4rs_rawtrain rl4rs_rawtrain;
int rl4rs_rawtrain_create(const rl4rs_rawpolicy_cfg* cfg, const rl4rs_rawpolicy_weights* w, void* stream, rl4rs_rawtrain** out);
int rl4rs_rawtrain_destroy(rl4rs_rawtrain* pol);
int rl4rs_rawtrain_params(rl4rs_rawtrain* pol, float** params_dev, float** grad_dev, int64_t* count);
int rl4rs_rawtrain_act(rl4rs_rawtrain* pol, int32_t N, const int32_t* cat_dev, const float* dense_dev,
                       const int32_t* const* seq_dev, const uint32_t* mask_bits_dev, uint32_t seed, uint32_t step,
                       int32_t* actions_dev, float* logp_dev, float* value_dev, float* entropy_dev, float* logits_dev,
                       void* stream);
int rl4rs_rawtrain_evaluate(rl4rs_rawtrain* pol, int32_t N, const int32_t* cat_dev, const float* dense_dev,
                            const int32_t* const* seq_dev, const uint32_t* mask_bits_dev, const int32_t* actions_dev,
                            float* logp_dev, float* value_dev, float* entropy_dev, float* logits_dev, void* stream);
int rl4rs_rawtrain_loss_grad(rl4rs_rawtrain* pol, int32_t algo, int32_t N, const int32_t* cat_dev, const float* dense_dev,
                             const int32_t* const* seq_dev, const uint32_t* mask_bits_dev, const int32_t* actions_dev,
                             const float* adv_dev, const float* ret_dev, const float* old_logp_dev,
                             const float* old_value_dev, const float* old_logits_dev, float vf_coeff, float ent_coeff,
                             float clip, float vf_clip, float kl_coeff, float* stats_dev, void* stream);
int rl4rs_rawtrain_adam_step(rl4rs_rawtrain* pol, float lr, float beta1, float beta2, float eps, float grad_clip,
                             void* stream);

/* Supervised training of the 'dnn' / 'widedeep' / 'lstm' simulators (rl4rs/nets/dnn.py, widedeep.py, lstm.py) on the device: what
 * script/supervised_train.py:37-42 does with model.compile(loss='binary_crossentropy', optimizer='adam') + model.fit -
 * forward in training mode (Dropout after each dense-tower layer, utils.py:48-54), keras binary_crossentropy of the
 * softmax output against the one-hot label, backward (BPTT through the keras GRUs of the lstm family), Adam.
 * cfg->algo = RL4RS_SIMNET_DNN, RL4RS_SIMNET_WIDEDEEP or RL4RS_SIMNET_LSTM.
 * Parameters, gradients and Adam state are flat float32 buffers (arrays a family does not have are skipped):
 *   [ cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | fc_w | fc_b | obs_w | obs_b | out_w | out_b |
 *     lstm: cat_gru kernel, recurrent, bias | seq0_gru kernel, recurrent, bias | seq1_gru ... ]
 * dense_dev [N, dense_feature_num] f32, cat_dev [N, category_feature_num] i32, seq_dev: seq_num pointers of int32
 * [N, maxlen] (widedeep, lstm; may be NULL for dnn), labels_dev [N] i32 in [0, class_num); the dropout masks are a pure
 * function of (seed, step, row, column). */
typedef struct rl4rs_simtrain rl4rs_simtrain;
int rl4rs_simtrain_create(const rl4rs_simnet_cfg* cfg, const rl4rs_simnet_weights* w, int32_t max_batch, void* stream,
                          rl4rs_simtrain** out);
int rl4rs_simtrain_destroy(rl4rs_simtrain* tr);
/* device pointers of the flat parameter / gradient buffers (owned by the handle) and their length */
int rl4rs_simtrain_params(rl4rs_simtrain* tr, float** params_dev, float** grad_dev, int64_t* count);
/* the dropout keep-masks (uint8 [N, hidden_units] each) of the last grad / step call (for gradient checks) */
int rl4rs_simtrain_masks(rl4rs_simtrain* tr, uint8_t** mask1_dev, uint8_t** mask2_dev);
/* forward + loss + backward into the gradient buffer; loss_dev[0] = mean loss (may be NULL) */
int rl4rs_simtrain_grad(rl4rs_simtrain* tr, int32_t N, const float* dense_dev, const int32_t* cat_dev,
                        const int32_t* const* seq_dev, const int32_t* labels_dev, float dropout_rate, uint32_t seed,
                        uint32_t step, float* loss_dev, void* stream);
/* rl4rs_simtrain_grad + one Adam update (keras defaults: lr 1e-3, beta 0.9 / 0.999, epsilon 1e-7) */
int rl4rs_simtrain_step(rl4rs_simtrain* tr, int32_t N, const float* dense_dev, const int32_t* cat_dev,
                        const int32_t* const* seq_dev, const int32_t* labels_dev, float lr, float beta1, float beta2,
                        float eps, float dropout_rate, uint32_t seed, uint32_t step, float* loss_dev, void* stream);

/* Supervised training of the DIEN simulator (rl4rs/nets/dien.py; script/supervised_train.py with model_type='dien') on the
 * device: training-mode forward (Dropout after each dense-tower layer), keras binary_crossentropy, backward through the
 * head, the category self-attention, the dense tower and, per sequence input, the DIN attention MLP, the AUGRU and the
 * first GRU (explicit BPTT), Adam.  Weights / sizes as for rl4rs_dien_create (max_rows, max_slots, scorer_mode unused).
 * Flat parameter / gradient layout:
 *   [ cat_emb | seq_emb | dense_w1 | dense_b1 | dense_w2 | dense_b2 | obs_w | obs_b | out_w | out_b |
 *     per sequence input: gru_gate_w, gru_gate_b, gru_cand_w, gru_cand_b, att_w1, att_b1, att_w2, att_b2, att_w3, att_b3,
 *                         augru_gate_w, augru_gate_b, augru_cand_w, augru_cand_b ]
 * Arguments of grad / step as for rl4rs_simtrain_grad / rl4rs_simtrain_step (seq_dev is required). */
/* Row-tile form of the persistent training recurrences (the GRU / AUGRU layers of rl4rs_dientrain_* and of the lstm family of
 * rl4rs_simtrain_*; reference: model.fit with batch_size 256, script/supervised_train.py:12-46): 0 = automatic (8-row workgroups
 * on v_mfma_f32_4x4x1 while 32-row ones would occupy fewer than half of the CUs - a 256-sample minibatch is 64 workgroups
 * instead of 16; hidden width 256 goes on to 4-row workgroups - 128 of them - while the 8-row ones would), 4 (hidden width 256;
 * other widths take 8), 8 or 32 = pinned.  Process-wide; same arithmetic up to the summation order of a dot product. */
int rl4rs_recur_train_set_rows(int32_t rows);

typedef struct rl4rs_dientrain rl4rs_dientrain;
int rl4rs_dientrain_create(const rl4rs_dien_cfg* cfg, const rl4rs_dien_weights* w, int32_t max_batch, void* stream,
                           rl4rs_dientrain** out);
int rl4rs_dientrain_destroy(rl4rs_dientrain* tr);
/* The launch chains that follow a recurrent layer (parameter-gradient reductions, the attention MLP's forward / backward) of the
 * ODD sequence inputs run on a second stream of the handle, beside the even inputs' (each ~35 small launches; own scratch; joined in
 * front of the next recurrent launch).  Process-wide switch, default 1; 0 = everything on the caller's stream (A/B measurements).
 * Same kernels on the same operands either way (the embedding-table gradient takes float atomics from both). */
int rl4rs_dientrain_set_fork(int32_t on);
int rl4rs_dientrain_params(rl4rs_dientrain* tr, float** params_dev, float** grad_dev, int64_t* count);
int rl4rs_dientrain_masks(rl4rs_dientrain* tr, uint8_t** mask1_dev, uint8_t** mask2_dev);
int rl4rs_dientrain_grad(rl4rs_dientrain* tr, int32_t N, const float* dense_dev, const int32_t* cat_dev,
                         const int32_t* const* seq_dev, const int32_t* labels_dev, float dropout_rate, uint32_t seed,
                         uint32_t step, float* loss_dev, void* stream);
int rl4rs_dientrain_step(rl4rs_dientrain* tr, int32_t N, const float* dense_dev, const int32_t* cat_dev,
                         const int32_t* const* seq_dev, const int32_t* labels_dev, float lr, float beta1, float beta2,
                         float eps, float dropout_rate, uint32_t seed, uint32_t step, float* loss_dev, void* stream);

/* Offline-RL learner networks and losses: what script/batchrl_trainer.py:34-90 trains with d3rlpy (DiscreteBC, DiscreteBCQ,
 * DiscreteCQL) on the logged-policy dataset, BASELINE configs[4].  A qnet is one encoder + d3rlpy's Linear head:
 *   mask_size = page_items + 1 > 0: CustomVectorEncoder of rl4rs/nets/cql/encoder.py:9-67 (with_q=True):
 *       relu(fc1(x)) | Embedding(action_size, emb_size)(x[-mask_size:]) -> fc2 -> [action_size], entries whose mask is 0 set to 0
 *       (mask = location_mask[cur_step % 9 // 3], previous actions and - once one was chosen - special items removed), then
 *       head Linear(action_size, action_size);
 *   mask_size = 0: d3rlpy VectorEncoder relu(fc1) -> relu(fc2) [hidden2], head Linear(hidden2, action_size).
 * Flat float32 parameter / gradient layout, matrices stored [in, out]:
 *   [ fc1_w obs_dim x hidden1 | fc1_b | emb action_size x emb_size (custom only) | fc2_w | fc2_b | head_w | head_b ]
 * params_host, location_mask [n_layers, action_size] and is_special [action_size] are HOST pointers (the masks may be NULL for
 * the plain encoder).  forward keeps the activations in the handle; backward must follow the forward of the SAME rows and
 * writes the gradient of sum(out * dout_dev).  adam_step is torch.optim.Adam.  status: bit 0 = an id in an observation tail
 * or an action was out of range (torch would raise IndexError), bit 1 = mask layer out of range. */
typedef struct rl4rs_qnet rl4rs_qnet;
typedef struct rl4rs_qnet_cfg {
    int32_t obs_dim;
    int32_t action_size;
    int32_t mask_size;
    int32_t emb_size;
    int32_t hidden1;
    int32_t hidden2;
    int32_t n_layers;
    int32_t max_rows;
} rl4rs_qnet_cfg;
int rl4rs_qnet_create(const rl4rs_qnet_cfg* cfg, const float* params_host, const uint8_t* location_mask,
                      const uint8_t* is_special, void* stream, rl4rs_qnet** out);
int rl4rs_qnet_destroy(rl4rs_qnet* net);
int rl4rs_qnet_params(rl4rs_qnet* net, float** params_dev, float** grad_dev, int64_t* count);
int rl4rs_qnet_copy_params(rl4rs_qnet* dst, const rl4rs_qnet* src, void* stream);
/* Adam moments (device pointers, `count` floats each) and step count of the handle: checkpointing (d3rlpy save_model / load_model,
 * script/batchrl_train.py:132,141). */
int rl4rs_qnet_adam_state(rl4rs_qnet* net, float** m_dev, float** v_dev, int64_t* step);
int rl4rs_qnet_set_adam_step(rl4rs_qnet* net, int64_t step);
int rl4rs_qnet_status(rl4rs_qnet* net, int32_t* flags, void* stream);
int rl4rs_qnet_forward(rl4rs_qnet* net, int32_t N, const float* obs_dev, float* out_dev, void* stream);
int rl4rs_qnet_backward(rl4rs_qnet* net, int32_t N, const float* obs_dev, const float* dout_dev, void* stream);
int rl4rs_qnet_adam_step(rl4rs_qnet* net, float lr, float beta1, float beta2, float eps, void* stream);
/* Greedy action per row [N]: argmax q (first maximum), or - with imitator logits - the DiscreteBCQ rule
 * argmax (q - min q) * [log pi - max log pi > log(action_flexibility)]. */
int rl4rs_q_best_action(int32_t N, int32_t A, const float* q_dev, const float* imitator_logits_dev,
                        float action_flexibility, int32_t* actions_dev, void* stream);
/* DiscreteImitator.compute_error: loss2_dev = {mean nll_loss(log_softmax(logits), a), mean_n sum_k logits^2}; dlogits_dev =
 * gradient of nll + beta * mean(logits^2).  rows_scratch_dev: [N, 2] float32. */
int rl4rs_qloss_imitation(rl4rs_qnet* net, int32_t N, const float* logits_dev, const int32_t* actions_dev, float beta,
                          float* dlogits_dev, float* rows_scratch_dev, float* loss2_dev, void* stream);
/* DoubleDQN temporal-difference loss (+ DiscreteCQL's conservative term when cql_alpha > 0): the next action is chosen
 * from q_next_dev by rl4rs_q_best_action's rule (imitator_next_logits_dev NULL = argmax), evaluated on q_next_target_dev;
 * loss2_dev = {mean huber(r + gamma * Q_targ(s')[a*] * (1 - terminal) - Q(s)[a]), mean(logsumexp Q(s) - Q(s)[a])};
 * dq_dev = gradient of td + cql_alpha * conservative wrt q_t_dev.  best_next_action_dev (optional) receives a*. */
int rl4rs_qloss_dqn(rl4rs_qnet* net, int32_t N, const float* q_t_dev, const int32_t* actions_dev, const float* rewards_dev,
                    const float* terminals_dev, const float* q_next_dev, const float* q_next_target_dev,
                    const float* imitator_next_logits_dev, float action_flexibility, float gamma, float cql_alpha,
                    float* dq_dev, float* rows_scratch_dev, float* loss2_dev, int32_t* best_next_action_dev, void* stream);

/* Continuous-action offline-RL learners: what script/batchrl_trainer.py:61-73 ('BCQ-conti', d3rlpy.algos.BCQ) and :91-107
 * ('CQL-conti', d3rlpy.algos.CQL) train on the continuous dataset of data_generate_rl4rs_a_conti (:220-270, actions = 32-d item
 * embeddings), BASELINE configs[4].  Both leave the custom encoder factory commented out, so every network is d3rlpy's default
 * VectorEncoderWithAction([256, 256], relu) on cat([x, action]) + one Linear head = an "amlp":
 *     h1 = relu([x | a] W1 + b1);  h2 = relu(h1 W2 + b2);  out = head_act(h2 W3 + b3)      (act_dim = 0: plain VectorEncoder)
 * head_act: the activation codes of rl4rs_gemm_f32 (0 none, 3 tanh ...).
 * Flat float32 parameter / gradient layout, matrices stored [in, out]:
 *     [ W1 (obs_dim + act_dim) x hidden1 (observation rows first, like torch.cat([x, action])) | b1 | W2 | b2 | W3 | b3 ]
 * forward: obs_dev [N / rep, obs_dim] - row r is the observation of rows r*rep .. r*rep + rep - 1 of act_dev [N, act_dim]
 * (rep sampled actions per observation: the observation side of the first layer is computed once per observation);
 * out_dev [N, out_dim].  The activations stay in the handle: backward must follow the forward of the SAME rows, takes the
 * gradient wrt the head's PRE-activation output (the loss entry points below fold the head activation in), writes every
 * parameter gradient into the handle when want_param_grad, and the gradient wrt act_dev into dact_dev when that is not NULL.
 * N <= max_rows for forward, N <= max_grad_rows for backward.  adam_step is torch.optim.Adam; soft_update is d3rlpy's
 * soft_sync  targ = (1 - tau) * targ + tau * src. */
typedef struct rl4rs_amlp rl4rs_amlp;
typedef struct rl4rs_amlp_cfg {
    int32_t obs_dim;
    int32_t act_dim;
    int32_t hidden1;
    int32_t hidden2;
    int32_t out_dim;
    int32_t head_act;
    int32_t max_rows;
    int32_t max_grad_rows;
} rl4rs_amlp_cfg;
int rl4rs_amlp_create(const rl4rs_amlp_cfg* cfg, const float* params_host, void* stream, rl4rs_amlp** out);
int rl4rs_amlp_destroy(rl4rs_amlp* net);
int rl4rs_amlp_params(rl4rs_amlp* net, float** params_dev, float** grad_dev, int64_t* count);
int rl4rs_amlp_copy_params(rl4rs_amlp* dst, const rl4rs_amlp* src, void* stream);
int rl4rs_amlp_adam_state(rl4rs_amlp* net, float** m_dev, float** v_dev, int64_t* step);
int rl4rs_amlp_set_adam_step(rl4rs_amlp* net, int64_t step);
int rl4rs_amlp_soft_update(rl4rs_amlp* targ, const rl4rs_amlp* src, float tau, void* stream);
int rl4rs_amlp_forward(rl4rs_amlp* net, int32_t N, int32_t rep, const float* obs_dev, const float* act_dev, float* out_dev,
                       void* stream);
/* The same forward for rows that never see a backward (target values, greedy evaluation: batch x n_action_samples rows), in
 * fp16x2 arithmetic (operands as fp16 hi + lo, three f16 MFMAs per product, fp32 accumulation - the scorer's form) as ONE launch
 * for the three layers behind the observation-side projection.  rl4rs_amlp_h16_ok: 1 when the network's shape has this form
 * (hidden 256 x 256, act_dim 8..64 in multiples of 8, out_dim <= 64).  The fp16 planes are rebuilt from the current fp32
 * parameters in front of every call; rows whose activations leave the fp16 range come back NaN; a following
 * rl4rs_amlp_backward is refused (no activations are kept).  Replaces the torch.no_grad() forwards of
 * d3rlpy BCQImpl.compute_target / _predict_best_action (script/batchrl_trainer.py:61-73 trains and evaluates through them). */
int rl4rs_amlp_h16_ok(const rl4rs_amlp* net);
int rl4rs_amlp_forward_h16(rl4rs_amlp* net, int32_t N, int32_t rep, const float* obs_dev, const float* act_dev, float* out_dev,
                           void* stream);
int rl4rs_amlp_backward(rl4rs_amlp* net, int32_t N, int32_t rep, const float* obs_dev, const float* act_dev,
                        const float* dout_dev, float* dact_dev, int32_t want_param_grad, void* stream);
int rl4rs_amlp_adam_step(rl4rs_amlp* net, float lr, float beta1, float beta2, float eps, void* stream);
/* The optimiser of ONE phase of an update as one launch: torch.optim.Adam for the n <= 8 networks with do_adam[i] != 0 (learning
 * rate lr[i]) and, for every i with targets[i] != NULL, d3rlpy's soft_sync targets[i] = (1 - tau) targets[i] + tau nets[i] computed
 * from the parameters AFTER this call's step (do_adam[i] = 0: soft update only).  targets may be NULL.  Element for element the
 * arithmetic of rl4rs_amlp_adam_step / rl4rs_amlp_soft_update. */
int rl4rs_amlp_adam_multi(int32_t n, rl4rs_amlp* const* nets, const float* lr, const int32_t* do_adam, rl4rs_amlp* const* targets,
                          float beta1, float beta2, float eps, float tau, void* stream);
/* Minibatch-sized rl4rs_amlp_forward / rl4rs_amlp_backward calls (hidden 256 x 256, out_dim / act_dim <= 64, rep = 1, N <= 2048 /
 * 1024) run as fused launches - one for the three layers forwards; transposes + input-gradient chain + all parameter gradients
 * backwards - instead of one launch per layer and product.  on = 0 restores the per-layer launches, 2 selects the 8-rows-per-
 * workgroup form of the fused kernels (measured slower at 256 rows; default 1 = 4 rows).  Process-wide; tests, A/B. */
int rl4rs_amlp_set_fused(int32_t on);
/* n <= 4 networks with the same input widths on the SAME rows (d3rlpy's twin critics: both Q functions see (s, a)) - forward /
 * backward of all of them as one launch each way when the call has the fused form, otherwise one rl4rs_amlp_forward / _backward
 * per network.  rep = 1.  outs_dev[i] [N, out_dim_i]; douts_dev[i] as for rl4rs_amlp_backward; dacts_dev (or its entries) may be NULL. */
int rl4rs_amlp_forward_multi(int32_t n, rl4rs_amlp* const* nets, int32_t N, const float* obs_dev, const float* act_dev,
                             float* const* outs_dev, void* stream);
int rl4rs_amlp_backward_multi(int32_t n, rl4rs_amlp* const* nets, int32_t N, const float* obs_dev, const float* act_dev,
                              const float* const* douts_dev, float* const* dacts_dev, int32_t want_param_grad, void* stream);
/* d3rlpy ConditionalVAE (the BCQ imitator).  enc_out_dev [N, 2L] = [mu | logstd] (the encoder amlp's two Linear heads side by
 * side); sample: z = mu + exp(clamp(logstd, min, max)) * eps (Normal.rsample with the caller's noise).
 * loss (compute_error): decoded_dev [N, E] = tanh output of the decoder amlp on (x, z); loss2_dev = {mean_n sum_e (y - a)^2,
 * mean_n sum_l KL(N(mu, sigma) || N(0, 1))} - the loss is loss2[0] / E + beta * loss2[1] / L; d_dec_pre_dev = gradient of the
 * mse term wrt the decoder's pre-tanh output.  rows_scratch_dev: [N, 2] float32.
 * encoder_grad: d_enc_out_dev [N, 2L] from dz_dev (the decoder's action-input gradient) and the beta-weighted KL term. */
int rl4rs_cvae_sample(int32_t N, int32_t L, const float* enc_out_dev, const float* eps_dev, float min_logstd, float max_logstd,
                      float* z_dev, void* stream);
int rl4rs_cvae_loss(int32_t N, int32_t E, int32_t L, const float* decoded_dev, const float* actions_dev, const float* enc_out_dev,
                    float min_logstd, float max_logstd, float* d_dec_pre_dev, float* rows_scratch_dev, float* loss2_dev,
                    void* stream);
int rl4rs_cvae_encoder_grad(int32_t N, int32_t L, const float* enc_out_dev, const float* eps_dev, const float* dz_dev, float beta,
                            float min_logstd, float max_logstd, float* d_enc_out_dev, void* stream);
/* d3rlpy DeterministicResidualPolicy: out = clamp(action + scale * tanh_out, -1, 1) with tanh_out_dev [N, E] the tanh head
 * of the policy amlp on (x, action); residual_grad: gradient wrt the policy head's pre-tanh output given d_out_dev. */
int rl4rs_residual_action(int32_t N, int32_t E, const float* action_dev, const float* tanh_out_dev, float scale, float* out_dev,
                          void* stream);
int rl4rs_residual_grad(int32_t N, int32_t E, const float* action_dev, const float* tanh_out_dev, float scale,
                        const float* d_out_dev, float* d_pre_dev, void* stream);
/* BCQ target (d3rlpy compute_max_with_n_actions): per row b the value max_j [(1 - lam) max(q1, q2) + lam min(q1, q2)] over its n
 * sampled actions (q*_dev [B * n]); y_dev [B] = rewards + gamma * value * (1 - terminals), or the bare value when rewards_dev
 * is NULL; best_dev [B] (optional) = the first maximising j.  q2_dev NULL: the value is q1 (the greedy pick of predict). */
int rl4rs_bcq_target(int32_t B, int32_t n, const float* q1_dev, const float* q2_dev, float lam, const float* rewards_dev,
                     const float* terminals_dev, float gamma, float* y_dev, int32_t* best_dev, void* stream);
/* out_dev [B, E] = rows_dev [b * n + best_dev[b], :] */
int rl4rs_pick_rows(int32_t B, int32_t n, int32_t E, const float* rows_dev, const int32_t* best_dev, float* out_dev, void* stream);
/* Twin ContinuousMeanQFunction error: loss2_dev = {mean (q1 - y)^2, mean (q2 - y)^2} (the critic loss is their sum),
 * dq*_dev = 2 (q* - y) / N. */
int rl4rs_critic_mse(int32_t N, const float* q1_dev, const float* q2_dev, const float* y_dev, float* dq1_dev, float* dq2_dev,
                     float* loss2_dev, void* stream);

/* Continuous CQL (d3rlpy.algos.CQL = SAC + the conservative critic term; 'CQL-conti', script/batchrl_trainer.py:91-107).
 * squashed_sample: d3rlpy SquashedNormalPolicy.sample(_n)_with_log_prob.  head_dev [N / rep, 2A] = [mu | logstd] (the policy
 * amlp's two Linear heads side by side, act_dim = 0), eps_dev [N, A] the caller's Gaussian noise (NULL: the deterministic
 * best_action a = tanh(mu), no log-prob).  u = mu + exp(clamp(logstd)) * eps, a = tanh(u), logp = sum_e [Normal log-prob of u
 * - 2 (log 2 - u - softplus(-2u))].  Sample i of observation r is written to destination row r * out_rep + out_off + i % rep of
 * act_out_dev [.., A] and logp_out_dev (so several sample groups of an observation can sit side by side).
 * sac_actor_grad: gradient of mean_b [exp(log_temp) * logp_b - Qmin(s_b, a_b)] wrt the head, given g_act_dev [B, A] = gradient
 * of -mean Qmin wrt the action (the critics' rl4rs_amlp_backward dact, 1 / B included); log_temp_dev: device scalar.
 * twin_min: qmin = min(q1, q2) and (optional) the selector dq_c = -1/B on the smaller critic (q1 on ties).
 * cql_critic_loss: rows laid out [B][m], column 0 = the dataset action, columns 1..m-1 = sampled actions with importance
 * offsets offs_dev (log-prob of a policy sample, A * log 0.5 of a uniform one):
 *   sums6_dev = {sum_b (q1[b,0] - y_b)^2, same for q2, sum_b logsumexp_j (q1[b,j] - offs[b,j]), same for q2, sum_b q1[b,0], sum_b q2[b,0]}
 *   dq_c[b,0] = 2 (q_c[b,0] - y_b) / B - aw / (2B),  dq_c[b,j>0] = aw / (2B) * softmax_j,  aw = *alpha_w_dev (clipped alpha *
 *   conservative weight, device scalar).  y_dev NULL: sums only (the alpha update).  rows_scratch_dev: [B, 6] float32. */
int rl4rs_squashed_sample(int32_t N, int32_t rep, int32_t A, const float* head_dev, const float* eps_dev, float min_logstd,
                          float max_logstd, int32_t out_rep, int32_t out_off, float* act_out_dev, float* logp_out_dev, void* stream);
int rl4rs_sac_actor_grad(int32_t B, int32_t A, const float* head_dev, const float* eps_dev, const float* act_dev,
                         const float* g_act_dev, const float* log_temp_dev, float min_logstd, float max_logstd, float* d_head_dev,
                         void* stream);
int rl4rs_twin_min(int32_t B, const float* q1_dev, const float* q2_dev, float* qmin_dev, float* dq1_dev, float* dq2_dev, void* stream);
int rl4rs_cql_critic_loss(int32_t B, int32_t m, const float* q1_dev, const float* q2_dev, const float* offs_dev, const float* y_dev,
                          const float* alpha_w_dev, float* dq1_dev, float* dq2_dev, float* rows_scratch_dev, float* sums6_dev,
                          void* stream);

/* One whole BCQ update (d3rlpy BCQ._update: imitator step, critic step, actor step, soft target updates - the sequence of
 * rl4rs_amlp_* / rl4rs_cvae_* / rl4rs_residual_* / rl4rs_bcq_target / rl4rs_critic_mse calls rl4rs_amd/offline_rl.py::BCQ.update
 * makes, with the same arguments, as ONE host call: after the fused kernels an update was bound by the ~55 Python -> C calls it
 * takes, not by the GPU).  Single-process only: the data-parallel learner keeps the per-phase calls around its all-reduces.
 *   noise_dev      (B + B * n + B) * L floats of N(0, 1): eps [B, L] | z_target [B * n, L] | z_actor [B, L]; the two z parts are
 *                  clamped to +-0.5 IN PLACE (BCQImpl: the decoder's latent is clipped when actions are sampled)
 *   workspace_dev  rl4rs_bcq_workspace_floats(B, n, E, L) floats, 16-byte aligned
 *   metrics_dev    float[3] = {imitator loss, critic loss, actor loss} (entries of skipped phases are left alone)
 *   do_rl / do_actor   total_step >= rl_start_step / total_step % update_actor_interval == 0;  nograd_h16: the B * n target rows
 *                  through rl4rs_amlp_forward_h16 where the network and the row count allow (the learner's nograd_precision) */
typedef struct rl4rs_bcq_step {
    rl4rs_amlp *imit_enc, *imit_dec, *policy, *policy_targ, *q1, *q2, *q1_targ, *q2_targ;
    int32_t B, n, E, L;
    float beta, action_flexibility, lam, gamma, tau, imitator_lr, critic_lr, actor_lr;
    int32_t do_rl, do_actor, nograd_h16, h16_min_rows;
    const float *obs_dev, *act_dev, *rew_dev, *nxt_dev, *ter_dev;
    float* noise_dev;
    float* workspace_dev;
    float* metrics_dev;
} rl4rs_bcq_step;
int64_t rl4rs_bcq_workspace_floats(int32_t B, int32_t n, int32_t E, int32_t L);
int rl4rs_bcq_update(const rl4rs_bcq_step* step, void* stream);

/* One whole continuous-CQL update (d3rlpy CQL._update as 'CQL-conti' runs it: temperature step, alpha step, critic step with the
 * conservative term over m = 1 + 3 n rows per transition, actor step, soft critic-target updates) as ONE host call - the sequence
 * rl4rs_amd/offline_rl.py::CQL.update issues, with the two learned scalars' Adam on the device.  Single-process only.
 *   log_temp / log_alpha   device float[3] each = {value, Adam m, Adam v}; *_step = the number of Adam steps taken so far (host)
 *   normal_dev   N(0, 1): eps_temp [B, A] | alpha eps_t [B n, A] | alpha eps_tp1 [B n, A] | critic eps_t | critic eps_tp1 | eps_actor [B, A]
 *   uniform_dev  U[-1, 1): alpha [B, n, A] | critic [B, n, A]
 *   rew_dev      rewards as the critic sees them (the caller applies the reward scaler)
 *   workspace_dev  rl4rs_cql_workspace_floats(B, n, A) floats, 16-byte aligned;  metrics_dev float[4] = {critic, actor, temp, alpha loss} */
typedef struct rl4rs_cql_step {
    rl4rs_amlp *policy, *q1, *q2, *q1_targ, *q2_targ;
    int32_t B, n, A;
    float gamma, tau, actor_lr, critic_lr, temp_lr, alpha_lr, alpha_threshold, conservative_weight;
    int32_t nograd_h16, h16_min_rows;
    int64_t temp_step, alpha_step;
    float* log_temp_dev;
    float* log_alpha_dev;
    const float *obs_dev, *act_dev, *rew_dev, *nxt_dev, *ter_dev;
    const float* normal_dev;
    const float* uniform_dev;
    float* workspace_dev;
    float* metrics_dev;
} rl4rs_cql_step;
int64_t rl4rs_cql_workspace_floats(int32_t B, int32_t n, int32_t A);
int rl4rs_cql_update(const rl4rs_cql_step* step, void* stream);

/* Plain fp32 GEMM used by the scorer, exposed for tests: C[M,N] = act(A[M,K] @ W[K,N] + bias).
 * act: 0 none, 1 ELU, 2 sigmoid, 3 tanh, 4 ReLU. */
int rl4rs_gemm_f32(const float* a_dev, int64_t lda, const float* w_dev, int64_t ldw,
                   const float* bias_dev, float* c_dev, int64_t ldc, int32_t M, int32_t N, int32_t K,
                   int act, void* stream);

/* Same GEMM through the pre-packed-weight kernel the scorer uses (w_host is a HOST pointer; packs, uploads,
 * runs and synchronises: test entry point only). */
int rl4rs_gemm_f32_packed(const float* a_dev, int64_t lda, const float* w_host, int64_t ldw,
                          const float* bias_dev, float* c_dev, int64_t ldc, int32_t M, int32_t N, int32_t K,
                          int act, void* stream);

/* The fp16x2 form of that GEMM (scorer_mode RL4RS_SCORER_FP16X2: operands as fp16 hi + lo pairs, three f16 MFMAs per
 * product, fp32 accumulation; an activation outside the fp16 range turns its output row into NaN).  Test entry point. */
int rl4rs_gemm_h16_packed(const float* a_dev, int64_t lda, const float* w_host, int64_t ldw,
                          const float* bias_dev, float* c_dev, int64_t ldc, int32_t M, int32_t N, int32_t K,
                          int act, void* stream);
/* Test hook: the device-side fragment packer (k_pack_h16_dev, used in front of rl4rs_amlp_forward_h16) against the host one
 * (pack_gemm_weight_h16, used when a scorer is loaded): number of 32-bit words that differ, 0 = bit-identical. */
int rl4rs_pack_h16_selftest(const float* w_host, int64_t ldw, int32_t K, int32_t N, int64_t* mismatches, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RL4RS_HIP_H */
