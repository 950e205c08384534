// npr_text.cpp -- cigar and SAM record text: the forms a job formats its output in (utils.py:597-605)
// (one of the translation units of the C ABI, include/nprealign.h; what they share: npr_api_internal.h)
#include "npr_api_internal.h"

namespace {
// cigar text of n op lists; op q of list i is get(i, q) -> (code, length)
template <typename Count, typename Get>
int64_t format_cigars(int64_t n, Count count, Get get, int64_t *str_off, char *out, int64_t cap) {
    static const char code[3] = {'M', 'I', 'D'};
    auto digits = [](int64_t v) { int k = 1; while (v >= 10) v /= 10, ++k; return k; };
    const int threads = usable_cpus();
    std::vector<int64_t> len(n);
    std::atomic<int> bad{0};
    parallel_for((n + 255) / 256, threads, [&](int64_t c) {
        for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
            int64_t k = 0;
            for (int64_t q = 0, m = count(i); q < m; ++q) {
                const std::pair<int32_t, int64_t> o = get(i, q);
                if (o.first < 0 || o.first > 2 || o.second < 0) bad = 1;
                k += digits(o.second) + 1;
            }
            len[i] = k ? k : 1;  // an empty cigar is "*"
        }
    });
    if (bad) return NPR_ERR_INVALID;
    str_off[0] = 0;
    for (int64_t i = 0; i < n; ++i) str_off[i + 1] = str_off[i] + len[i];
    if (!out) return str_off[n];
    if (cap < str_off[n]) return NPR_ERR_CAPACITY;
    parallel_for((n + 255) / 256, threads, [&](int64_t c) {
        for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
            char *w = out + str_off[i];
            const int64_t m = count(i);
            if (m == 0) *w = '*';
            for (int64_t q = 0; q < m; ++q) {
                const std::pair<int32_t, int64_t> o = get(i, q);
                int64_t v = o.second;
                const int k = digits(v);
                for (int j = k - 1; j >= 0; --j) w[j] = static_cast<char>('0' + v % 10), v /= 10;
                w[k] = code[o.first];
                w += k + 1;
            }
        }
    });
    return str_off[n];
}
}  // namespace

extern "C" {

int64_t npr_format_cigars(int64_t n, const int64_t *ops_off, const int32_t *ops, int64_t *str_off, char *out, int64_t cap) {
    if (n < 0 || (n && (!ops_off || !str_off)) || (n && ops_off[n] > 0 && !ops)) return NPR_ERR_INVALID;
    try {
        return format_cigars(n, [&](int64_t i) { return ops_off[i + 1] - ops_off[i]; },
                             [&](int64_t i, int64_t q) { return std::pair<int32_t, int64_t>(ops[2 * (ops_off[i] + q)], ops[2 * (ops_off[i] + q) + 1]); }, str_off, out, cap);
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_format_sam_records(int64_t n, const char *qnames, const int64_t *qname_off, const int32_t *flag, const char *rnames,
                               const int64_t *rname_off, const int32_t *ref_index, const int64_t *pos, const int32_t *mapq,
                               const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, const char *seq, const int64_t *seq_off,
                               int64_t *rec_off, char *out, int64_t cap) {
    if (n < 0 || (n && (!qnames || !qname_off || !rnames || !rname_off || !ref_index || !pos || !word_off || !n_ops || !seq || !seq_off || !rec_off)))
        return NPR_ERR_INVALID;
    try {
        static const char code[3] = {'M', 'I', 'D'};
        auto digits = [](int64_t v) { int k = 1; while (v >= 10) v /= 10, ++k; return k; };
        auto put = [](char *&w, int64_t v, int k) {
            for (int j = k - 1; j >= 0; --j) w[j] = static_cast<char>('0' + v % 10), v /= 10;
            w += k;
        };
        const int threads = usable_cpus();
        std::atomic<int> bad{0};
        std::vector<int64_t> len(n);
        // fixed part of a record: ten tabs, "*", "0", "0", "*", newline
        parallel_for((n + 255) / 256, threads, [&](int64_t c) {
            for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
                int64_t k = 0;
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t w = words[word_off[i] + q];
                    if ((w & 3u) > 2u) bad = 1;
                    k += digits(static_cast<int64_t>(w >> 2)) + 1;
                }
                if (n_ops[i] < 0 || pos[i] < 0 || (flag && flag[i] < 0) || (mapq && mapq[i] < 0) || ref_index[i] < 0) bad = 1;
                const int64_t r = ref_index[i] < 0 ? 0 : ref_index[i];
                len[i] = (qname_off[i + 1] - qname_off[i]) + digits(flag ? flag[i] : 0) + (rname_off[r + 1] - rname_off[r]) + digits(pos[i]) +
                         digits(mapq ? mapq[i] : 255) + (k ? k : 1) + std::max<int64_t>(seq_off[i + 1] - seq_off[i], 1) + 10 + 5;
            }
        });
        if (bad) return NPR_ERR_INVALID;
        rec_off[0] = 0;
        for (int64_t i = 0; i < n; ++i) rec_off[i + 1] = rec_off[i] + len[i];
        if (!out) return rec_off[n];
        if (cap < rec_off[n]) return NPR_ERR_CAPACITY;
        parallel_for((n + 63) / 64, threads, [&](int64_t c) {
            for (int64_t i = c * 64, hi = std::min(n, (c + 1) * 64); i < hi; ++i) {
                char *w = out + rec_off[i];
                const int64_t ql = qname_off[i + 1] - qname_off[i], r = ref_index[i], rl = rname_off[r + 1] - rname_off[r],
                              sl = seq_off[i + 1] - seq_off[i];
                std::memcpy(w, qnames + qname_off[i], static_cast<size_t>(ql)), w += ql;
                *w++ = '\t';
                put(w, flag ? flag[i] : 0, digits(flag ? flag[i] : 0));
                *w++ = '\t';
                std::memcpy(w, rnames + rname_off[r], static_cast<size_t>(rl)), w += rl;
                *w++ = '\t';
                put(w, pos[i], digits(pos[i]));
                *w++ = '\t';
                put(w, mapq ? mapq[i] : 255, digits(mapq ? mapq[i] : 255));
                *w++ = '\t';
                if (n_ops[i] == 0) *w++ = '*';
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t cw = words[word_off[i] + q];
                    const int64_t v = static_cast<int64_t>(cw >> 2);
                    put(w, v, digits(v));
                    *w++ = code[cw & 3u];
                }
                std::memcpy(w, "\t*\t0\t0\t", 7), w += 7;
                if (sl == 0) *w++ = '*';  // an empty SEQ is "*" in SAM
                std::memcpy(w, seq + seq_off[i], static_cast<size_t>(sl)), w += sl;
                std::memcpy(w, "\t*\n", 3), w += 3;
            }
        });
        return rec_off[n];
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_format_cigars_packed(int64_t n, const int64_t *word_off, const int64_t *n_ops, const uint32_t *words, int64_t *str_off, char *out,
                                 int64_t cap) {
    if (n < 0 || (n && (!word_off || !n_ops || !str_off || !words))) return NPR_ERR_INVALID;
    try {
        return format_cigars(n, [&](int64_t i) { return n_ops[i]; },
                             [&](int64_t i, int64_t q) {
                                 const uint32_t w = words[word_off[i] + q];
                                 return std::pair<int32_t, int64_t>(static_cast<int32_t>(w & 3u), static_cast<int64_t>(w >> 2));
                             },
                             str_off, out, cap);
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

void npr_encode_bases(const uint8_t *ascii, int64_t n, uint8_t *codes) {
    for (int64_t i = 0; i < n; ++i) codes[i] = encode_base(ascii[i]);
}

}  // extern "C"
