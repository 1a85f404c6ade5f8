// Host-side parser of RL4RS log records into the columnar arrays rl4rs_env_load_batch takes.
//
// Replaces the per-reset Python text parsing of the reference (FeatureUtil.record_split,
// rl4rs/utils/datautil.py:20-32; SlateState.records_to_state, rl4rs/env/slate.py:67-83; pad_sequences of the
// history, datautil.py:43-46) for the whole sample file at once, so a log is parsed ONE time and then lives in HBM.
// Record: timestamp@session_id@sequence_id@exposed_items@user_feedback@user_seqfeature@user_protrait@item_feature@behavior_policy_id
// Numbers are read with strtod / strtol (correctly rounded, same values as Python's float()/int()):
//   user_cat   = int(portrait[j])  for j < user_cat_dim      (datautil.py:49 truncates the float ids)
//   user_dense = float32(portrait[user_cat_dim + j])          (pad_sequences(dtype='float32'), datautil.py:52-58)
//   history    = last `maxlen` ids, left-padded with 0        (Keras pad_sequences defaults)
#include <cerrno>
#include <cstdlib>
#include <cstring>

#include "common.hpp"

namespace {

struct Field { const char* b; const char* e; };

// split [b,e) on `sep` into at most max fields; returns count (or -1 if more than max)
int split(const char* b, const char* e, char sep, Field* out, int max) {
    int n = 0;
    const char* s = b;
    for (const char* p = b;; ++p) {
        if (p == e || *p == sep) {
            if (n == max) return -1;
            out[n].b = s;
            out[n].e = p;
            ++n;
            s = p + 1;
            if (p == e) break;
        }
    }
    return n;
}

bool parse_int_list(const Field& f, int32_t* dst, int cap, int* count) {
    int n = 0;
    const char* p = f.b;
    if (p == f.e) return false;                       // int('') raises in the reference
    while (p < f.e) {
        char* end = nullptr;
        errno = 0;
        long v = strtol(p, &end, 10);
        if (end == p || errno) return false;
        if (n < cap) dst[n] = (int32_t)v;
        ++n;
        p = end;
        if (p < f.e) {
            if (*p != ',') return false;
            ++p;
            if (p == f.e) return false;
        }
    }
    *count = n;
    return true;
}

}  // namespace

extern "C" int rl4rs_parse_records(const char* text, int64_t len, int32_t max_records, int32_t maxlen, int32_t log_steps,
                                   int32_t user_dense_dim, int32_t user_cat_dim, int32_t* exposed, int32_t* feedback,
                                   int32_t* history, float* user_dense, int32_t* user_cat, int32_t* exposed_len,
                                   int32_t* n_parsed) {
    using rl4rs::set_error;
    RL4RS_REQUIRE(text && exposed && feedback && history && user_dense && user_cat && n_parsed && len >= 0,
                  "parse_records: null argument");
    RL4RS_REQUIRE(maxlen > 0 && log_steps > 0 && user_dense_dim >= 0 && user_cat_dim >= 0 && max_records >= 0,
                  "parse_records: bad sizes");
    const char* p = text;
    const char* end = text + len;
    int32_t rec = 0;
    std::vector<int32_t> tmp;
    tmp.resize(1 << 16);
    while (p < end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* lt = le;
        while (lt > p && (lt[-1] == '\r' || lt[-1] == ' ' || lt[-1] == '\t')) --lt;      // rstrip()
        if (lt > p) {
            if (rec >= max_records) {
                set_error("parse_records: more than max_records=%d records", max_records);
                return RL4RS_EINVAL;
            }
            Field f[9];
            if (split(p, lt, '@', f, 9) != 9) {
                set_error("parse_records: record %d does not have 9 '@' fields (datautil.py:22-23)", rec);
                return RL4RS_EINVAL;
            }
            int n = 0;
            // exposed_items / user_feedback (zero padded / truncated to log_steps columns)
            int32_t* ex = exposed + (size_t)rec * log_steps;
            int32_t* fb = feedback + (size_t)rec * log_steps;
            memset(ex, 0, sizeof(int32_t) * log_steps);
            memset(fb, 0, sizeof(int32_t) * log_steps);
            if (!parse_int_list(f[3], ex, log_steps, &n)) { set_error("parse_records: bad exposed_items in record %d", rec); return RL4RS_EINVAL; }
            if (exposed_len) exposed_len[rec] = n;
            if (!parse_int_list(f[4], fb, log_steps, &n)) { set_error("parse_records: bad user_feedback in record %d", rec); return RL4RS_EINVAL; }
            // history: keep the last maxlen ids, right aligned
            if (!parse_int_list(f[5], tmp.data(), (int)tmp.size(), &n)) { set_error("parse_records: bad user_seqfeature in record %d", rec); return RL4RS_EINVAL; }
            if (n > (int)tmp.size()) { set_error("parse_records: history of record %d longer than %zu", rec, tmp.size()); return RL4RS_EINVAL; }
            int32_t* hi = history + (size_t)rec * maxlen;
            memset(hi, 0, sizeof(int32_t) * maxlen);
            int keep = n < maxlen ? n : maxlen;
            for (int i = 0; i < keep; ++i) hi[maxlen - keep + i] = tmp[n - keep + i];
            // portrait: user_cat_dim ids (stored as numbers) then user_dense_dim floats
            const char* q = f[6].b;
            int idx = 0;
            const int want = user_cat_dim + user_dense_dim;
            while (q < f[6].e) {
                char* e2 = nullptr;
                errno = 0;
                double v = strtod(q, &e2);
                if (e2 == q) { set_error("parse_records: bad user_protrait in record %d", rec); return RL4RS_EINVAL; }
                if (idx < user_cat_dim) user_cat[(size_t)rec * user_cat_dim + idx] = (int32_t)(long long)v;
                else if (idx < want) user_dense[(size_t)rec * user_dense_dim + (idx - user_cat_dim)] = (float)v;
                ++idx;
                q = e2;
                if (q < f[6].e) {
                    if (*q != ',') { set_error("parse_records: bad user_protrait in record %d", rec); return RL4RS_EINVAL; }
                    ++q;
                }
            }
            if (idx < want) {
                set_error("parse_records: record %d has %d portrait values, expected >= %d", rec, idx, want);
                return RL4RS_EINVAL;
            }
            ++rec;
        }
        if (!nl) break;
        p = nl + 1;
    }
    *n_parsed = rec;
    return RL4RS_OK;
}


// CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), slicing-by-8: the checksum of the TFRecord framing
// (rl4rs/utils/datautil.py:71-230 reads / writes TFRecords through tf.io) and of TF tensor-bundle checkpoints
// (tf.train.Saver, rl4rs/env/base.py:129,151).  `crc` continues a running checksum (0 starts one).
namespace {
struct Crc32cTables {
    uint32_t t[8][256];
    Crc32cTables() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 0xFF] ^ (t[k - 1][i] >> 8);
    }
};
}  // namespace

extern "C" uint32_t rl4rs_crc32c(const void* data, int64_t len, uint32_t crc) {
    static const Crc32cTables T;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = ~crc;
    while (len >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T.t[7][lo & 0xFF] ^ T.t[6][(lo >> 8) & 0xFF] ^ T.t[5][(lo >> 16) & 0xFF] ^ T.t[4][lo >> 24] ^
            T.t[3][hi & 0xFF] ^ T.t[2][(hi >> 8) & 0xFF] ^ T.t[1][(hi >> 16) & 0xFF] ^ T.t[0][hi >> 24];
        p += 8;
        len -= 8;
    }
    while (len-- > 0) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return ~c;
}
