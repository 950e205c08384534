// npr_io.cpp -- bulk text ingest and splice for the file side of the realign path (include/nprealign.h, "bulk text
// ingest"): the record loop of realignSamFile2TargetFn / realignSamFile3TargetFn (nanopore/analyses/utils.py:557-609) and the
// FASTA / FASTQ dictionaries (utils.py:233-245) over the whole text of a file at once.  Host code, threaded, no HIP.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "nprealign.h"
#include "npr_threads.h"

using npr::parallel_for;
using npr::usable_cpus;

namespace {

inline const char *find(const char *p, const char *end, char c) {
    const void *q = p < end ? std::memchr(p, c, static_cast<size_t>(end - p)) : nullptr;
    return q ? static_cast<const char *>(q) : end;
}

inline bool blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

// decimal integer spanning the whole field (an optional leading '-'); false when anything else is in it
inline bool parse_int(const char *p, const char *end, int64_t &v) {
    if (p >= end) return false;
    bool neg = false;
    if (*p == '-') neg = true, ++p;
    if (p >= end || end - p > 18) return false;
    int64_t x = 0;
    for (; p < end; ++p) {
        if (*p < '0' || *p > '9') return false;
        x = x * 10 + (*p - '0');
    }
    v = neg ? -x : x;
    return true;
}

// SAM cigar operation letters in pysam's numbering: M I D N S H P = X (-1: anything else)
struct OpTable {
    int8_t t[256];
    constexpr OpTable() : t{} {
        for (int i = 0; i < 256; ++i) t[i] = -1;
        t['M'] = 0, t['I'] = 1, t['D'] = 2, t['N'] = 3, t['S'] = 4, t['H'] = 5, t['P'] = 6, t['='] = 7, t['X'] = 8;
    }
    constexpr int operator[](unsigned char c) const { return t[c]; }
};
constexpr OpTable kOpCode{};
inline int op_code(char c) {
    switch (c) {
        case 'M': return 0;
        case 'I': return 1;
        case 'D': return 2;
        case 'N': return 3;
        case 'S': return 4;
        case 'H': return 5;
        case 'P': return 6;
        case '=': return 7;
        case 'X': return 8;
        default: return -1;
    }
}

inline int digits(int64_t v) {
    int k = 1;
    while (v >= 10) v /= 10, ++k;
    return k;
}
// ... of a cigar length (30 bits), without a division: a job formats 10^8 of them
inline int digits30(uint32_t v) {
    return v < 10u ? 1 : v < 100u ? 2 : v < 1000u ? 3 : v < 10000u ? 4 : v < 100000u ? 5 : v < 1000000u ? 6 : v < 10000000u ? 7 : v < 100000000u ? 8 : v < 1000000000u ? 9 : 10;
}

}  // namespace

extern "C" {

int64_t npr_sam_index(const char *text, int64_t len, int64_t *header_end, int64_t *span, int64_t cap) {
    if (len < 0 || (len && !text)) return NPR_ERR_INVALID;
    try {
        const char *const end = text + len;
        // the header: the leading run of '@' lines
        const char *p = text;
        while (p < end && *p == '@') {
            const char *nl = find(p, end, '\n');
            p = nl < end ? nl + 1 : end;
        }
        const int64_t body = p - text;
        if (header_end) *header_end = body;
        // lines of the body, found piecewise: a piece owns the lines that START in it
        const int threads = usable_cpus();
        const int64_t pieces = std::max<int64_t>(1, std::min<int64_t>(threads * 4, (len - body) >> 16));
        std::vector<std::vector<int64_t>> found(pieces);
        parallel_for(pieces, threads, [&](int64_t k) {
            const int64_t lo = body + (len - body) * k / pieces, hi = body + (len - body) * (k + 1) / pieces;
            const char *q = text + lo;
            if (lo > body && q[-1] != '\n') {  // in the middle of a line that belongs to the piece before
                q = find(q, end, '\n');
                q = q < end ? q + 1 : end;
            }
            std::vector<int64_t> &out = found[k];
            while (q < text + hi) {
                const char *nl = find(q, end, '\n');
                const char *e = nl;
                if (e > q && e[-1] == '\r') --e;
                if (e > q && *q != '@') out.push_back(q - text), out.push_back(e - text);  // (a stray header line is skipped, as sam.py's reader does)
                q = nl < end ? nl + 1 : end;
            }
        });
        int64_t n = 0;
        for (const auto &f : found) n += static_cast<int64_t>(f.size() / 2);
        if (!span) return n;
        if (cap < n) return NPR_ERR_CAPACITY;
        int64_t at = 0;
        for (const auto &f : found) {
            if (!f.empty()) std::memcpy(span + at, f.data(), f.size() * sizeof(int64_t));
            at += static_cast<int64_t>(f.size());
        }
        return n;
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int32_t npr_sam_parse(const char *text, const int64_t *span, int64_t n, const char *rnames, const int64_t *rname_off, int64_t n_refs,
                      int64_t *fields) {
    if (n < 0 || n_refs < 0 || (n && (!text || !span || !fields)) || (n_refs && (!rnames || !rname_off))) return NPR_ERR_INVALID;
    try {
        std::unordered_map<std::string_view, int64_t> tid;
        tid.reserve(static_cast<size_t>(n_refs) * 2);
        for (int64_t k = 0; k < n_refs; ++k)
            tid.emplace(std::string_view(rnames + rname_off[k], static_cast<size_t>(rname_off[k + 1] - rname_off[k])), k);  // (first wins, as sam.py's dict would not: names are unique in a header)
        parallel_for((n + 127) / 128, usable_cpus(), [&](int64_t c) {
            for (int64_t i = c * 128, hi = std::min(n, (c + 1) * 128); i < hi; ++i) {
                int64_t *f = fields + i * NPR_SAM_COLS;
                std::fill(f, f + NPR_SAM_COLS, int64_t(0));
                f[10] = -1;
                f[15] = NPR_ERR_INVALID;
                const char *const ls = text + span[2 * i], *const le = text + span[2 * i + 1];
                // the eleven mandatory columns: ten tabs (where QUAL ends is nobody's business here: half of the line's bytes not scanned)
                const char *col[12];
                col[0] = ls;
                int got = 1;
                for (const char *q = ls; got < 11;) {
                    const char *t = find(q, le, '\t');
                    if (t >= le) break;
                    col[got++] = t + 1, q = t + 1;
                }
                if (got < 11) continue;
                const char *col_end[11];
                for (int k = 0; k < 11; ++k) col_end[k] = (k + 1 < got) ? col[k + 1] - 1 : le;
                f[0] = col_end[0] - text;
                f[1] = col[2] - text, f[2] = col_end[2] - text;
                f[3] = col[5] - text, f[4] = col_end[5] - text;
                f[5] = col[9] - text, f[6] = col_end[9] - text;
                int64_t flag, pos, mapq, dummy;
                if (!parse_int(col[1], col_end[1], flag) || !parse_int(col[3], col_end[3], pos) || !parse_int(col[4], col_end[4], mapq) ||
                    !parse_int(col[7], col_end[7], dummy) || !parse_int(col[8], col_end[8], dummy))
                    continue;
                f[7] = flag, f[8] = pos - 1, f[9] = mapq;
                const std::string_view rn(col[2], static_cast<size_t>(col_end[2] - col[2]));
                const auto it = tid.find(rn);
                f[10] = it == tid.end() ? -1 : it->second;
                if (it == tid.end()) {  // no reference: "*" is a record samIterator drops, any other name is an error of the file
                    f[15] = (rn.size() == 1 && rn[0] == '*') ? NPR_SAM_NO_REFERENCE : NPR_SAM_UNKNOWN_REFERENCE;
                    continue;
                }
                const bool no_seq = col_end[9] - col[9] == 1 && *col[9] == '*';
                const int64_t seq_len = no_seq ? 0 : col_end[9] - col[9];
                // the cigar: clips at both ends, M / I / D count, reference bases consumed
                const char *q = col[5];
                const char *const ce = col_end[5];
                bool ok = !(ce - q == 1 && *q == '*') && q < ce;
                int64_t lead = 0, trail = 0, mid = 0, refspan = 0;
                bool in_lead = true;
                // (10^8 operations per job: the digits without a bound test -- the column ends in a tab, which is no digit --, the
                // letter through a table, M / I / D first)
                while (ok && q < ce) {
                    uint64_t v = 0;
                    const char *d = q;
                    unsigned c;
                    while ((c = static_cast<unsigned>(static_cast<unsigned char>(*d)) - '0') <= 9u) v = v * 10 + c, ++d;
                    if (d == q || d - q > 10 || d >= ce) {
                        ok = false;
                        break;
                    }
                    const int op = kOpCode[static_cast<unsigned char>(*d)];
                    q = d + 1;
                    if (static_cast<unsigned>(op) <= 2u && v < (uint64_t(1) << 29)) {
                        in_lead = false, trail = 0, ++mid;
                        if (op != 1) refspan += static_cast<int64_t>(v);
                    } else if (op < 0 || v >= (uint64_t(1) << 29)) {
                        ok = false;
                    } else if (op == 4) {
                        if (in_lead) lead += static_cast<int64_t>(v);
                        trail += static_cast<int64_t>(v);
                    } else if (op == 5) {
                        // hard clips: no bases in SEQ, no operation (pysam's qstart / qend skip them)
                    } else {
                        ok = false;  // N P = X: the reference asserts op in (0, 1, 2, 4, 5) (utils.py:171)
                    }
                }
                if (!ok) continue;
                const int64_t qs = std::min(lead, seq_len), qe = std::max(qs, seq_len - trail);
                f[11] = f[5] + (no_seq ? 0 : qs), f[12] = f[5] + (no_seq ? 0 : qe);
                f[13] = mid, f[14] = refspan;
                f[15] = NPR_OK;
            }
        });
        return NPR_OK;
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int32_t npr_sam_guides(const char *text, const int64_t *fields, int64_t n, const int64_t *guide_off, int32_t *guide_ops) {
    if (n < 0 || (n && (!text || !fields || !guide_off)) || (n && guide_off[n] > guide_off[0] && !guide_ops)) return NPR_ERR_INVALID;
    try {
        parallel_for((n + 127) / 128, usable_cpus(), [&](int64_t c) {
            for (int64_t i = c * 128, hi = std::min(n, (c + 1) * 128); i < hi; ++i) {
                const int64_t *f = fields + i * NPR_SAM_COLS;
                if (f[15] != NPR_OK) continue;
                int32_t *w = guide_ops + 2 * guide_off[i];
                const int32_t *const we = guide_ops + 2 * guide_off[i + 1];
                const char *q = text + f[3];
                const char *const ce = text + f[4];
                while (q < ce && w < we) {  // (a column npr_sam_parse accepted: numbers below 2^29, each followed by a letter, a tab behind it)
                    uint32_t v = 0;
                    unsigned c;
                    while ((c = static_cast<unsigned>(static_cast<unsigned char>(*q)) - '0') <= 9u) v = v * 10 + c, ++q;
                    if (q >= ce) break;
                    const int op = kOpCode[static_cast<unsigned char>(*q++)];
                    if (static_cast<unsigned>(op) <= 2u) w[0] = op, w[1] = static_cast<int32_t>(v), w += 2;
                }
            }
        });
        return NPR_OK;
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_sam_splice(const char *text, const int64_t *span, const int64_t *fields, int64_t n, const int64_t *word_off,
                       const int64_t *n_ops, const uint32_t *words, int64_t *rec_off, char *out, int64_t cap) {
    if (n < 0 || (n && (!text || !span || !fields || !word_off || !n_ops || !rec_off))) return NPR_ERR_INVALID;
    try {
        static const char code[3] = {'M', 'I', 'D'};
        const int threads = usable_cpus();
        std::atomic<int> bad{0};
        std::vector<int64_t> lens(n);
        parallel_for((n + 255) / 256, threads, [&](int64_t c) {
            for (int64_t i = c * 256, hi = std::min(n, (c + 1) * 256); i < hi; ++i) {
                const int64_t *f = fields + i * NPR_SAM_COLS;
                int64_t k = 0;
                if (n_ops[i] < 0 || (n_ops[i] && !words)) {
                    bad = 1;
                    continue;
                }
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t w = words[word_off[i] + q];
                    if ((w & 3u) > 2u) bad = 1;
                    k += digits30(w >> 2) + 1;
                }
                lens[i] = (f[3] - span[2 * i]) + (k ? k : 1) + (span[2 * i + 1] - f[4]) + 1;
            }
        });
        if (bad) return NPR_ERR_INVALID;
        rec_off[0] = 0;
        for (int64_t i = 0; i < n; ++i) rec_off[i + 1] = rec_off[i] + lens[i];
        if (!out) return rec_off[n];
        if (cap < rec_off[n]) return NPR_ERR_CAPACITY;
        parallel_for((n + 63) / 64, threads, [&](int64_t c) {
            for (int64_t i = c * 64, hi = std::min(n, (c + 1) * 64); i < hi; ++i) {
                const int64_t *f = fields + i * NPR_SAM_COLS;
                char *w = out + rec_off[i];
                const size_t head = static_cast<size_t>(f[3] - span[2 * i]), tail = static_cast<size_t>(span[2 * i + 1] - f[4]);
                std::memcpy(w, text + span[2 * i], head), w += head;
                if (n_ops[i] == 0) *w++ = '*';
                for (int64_t q = 0; q < n_ops[i]; ++q) {
                    const uint32_t cw = words[word_off[i] + q];
                    uint32_t v = cw >> 2;
                    if (v < 10u) {  // (most operations of a realigned nanopore read)
                        w[0] = static_cast<char>('0' + v), w[1] = code[cw & 3u], w += 2;
                        continue;
                    }
                    const int k = digits30(v);
                    for (int j = k - 1; j >= 0; --j) w[j] = static_cast<char>('0' + v % 10u), v /= 10u;
                    w[k] = code[cw & 3u];
                    w += k + 1;
                }
                std::memcpy(w, text + f[4], tail), w += tail;
                *w = '\n';
            }
        });
        return rec_off[n];
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_fasta_index(const char *text, int64_t len, int64_t *rec, int64_t *seq_len, int64_t cap) {
    if (len < 0 || (len && !text)) return NPR_ERR_INVALID;
    const char *const end = text + len;
    int64_t n = 0;
    const char *p = text;
    bool open = false;
    int64_t bases = 0;
    auto close_record = [&](const char *at) {
        if (open && rec && n <= cap) rec[4 * (n - 1) + 3] = at - text, seq_len[n - 1] = bases;
    };
    while (p < end) {
        const char *nl = find(p, end, '\n');
        if (*p == '>') {
            close_record(p);
            ++n, open = true, bases = 0;
            if (rec) {
                if (n > cap) return NPR_ERR_CAPACITY;
                const char *a = p + 1;
                while (a < nl && blank(*a)) ++a;
                const char *b = a;
                while (b < nl && !blank(*b)) ++b;
                rec[4 * (n - 1)] = a - text, rec[4 * (n - 1) + 1] = b - text;
                rec[4 * (n - 1) + 2] = (nl < end ? nl + 1 : end) - text;
            }
        } else if (open) {
            const char *a = p, *b = nl;
            while (a < b && blank(*a)) ++a;
            while (b > a && blank(b[-1])) --b;
            bases += b - a;
        }
        p = nl < end ? nl + 1 : end;
    }
    close_record(end);
    return n;
}

int32_t npr_fasta_pack(const char *text, const int64_t *rec, int64_t n, const int64_t *seq_off, uint8_t *out) {
    if (n < 0 || (n && (!text || !rec || !seq_off || !out))) return NPR_ERR_INVALID;
    try {
        std::atomic<int> bad{0};
        parallel_for(n, usable_cpus(), [&](int64_t k) {
            const char *p = text + rec[4 * k + 2], *const end = text + rec[4 * k + 3];
            uint8_t *w = out + seq_off[k], *const we = out + seq_off[k + 1];
            while (p < end) {
                const char *nl = find(p, end, '\n');
                const char *a = p, *b = nl;
                while (a < b && blank(*a)) ++a;
                while (b > a && blank(b[-1])) --b;
                if (w + (b - a) > we) {
                    bad = 1;
                    return;
                }
                std::memcpy(w, a, static_cast<size_t>(b - a)), w += b - a;
                p = nl < end ? nl + 1 : end;
            }
            if (w != we) bad = 1;
        });
        return bad ? NPR_ERR_INVALID : NPR_OK;
    } catch (const std::exception &) {
        return NPR_ERR_NOMEM;
    }
}

int64_t npr_fastq_index(const char *text, int64_t len, int64_t *rec, int64_t cap) {
    if (len < 0 || (len && !text)) return NPR_ERR_INVALID;
    const char *const end = text + len;
    int64_t n = 0;
    const char *p = text;
    auto line = [&](const char *&a, const char *&b) {  // next line without its line end; false at the end of the text
        if (p >= end) return false;
        const char *nl = find(p, end, '\n');
        a = p, b = nl;
        while (b > a && (b[-1] == '\r')) --b;
        p = nl < end ? nl + 1 : end;
        return true;
    };
    const char *a, *b;
    while (line(a, b)) {
        if (a == b) continue;  // blank lines between records
        if (*a != '@') return NPR_ERR_INVALID;
        const char *sa, *sb, *pa, *pb, *qa, *qb;
        if (!line(sa, sb) || !line(pa, pb) || pa == pb || *pa != '+') return NPR_ERR_INVALID;
        (void)line(qa, qb);
        ++n;
        if (rec) {
            if (n > cap) return NPR_ERR_CAPACITY;
            const char *x = a + 1;
            while (x < b && blank(*x)) ++x;
            const char *y = x;
            while (y < b && !blank(*y)) ++y;
            rec[4 * (n - 1)] = x - text, rec[4 * (n - 1) + 1] = y - text, rec[4 * (n - 1) + 2] = sa - text, rec[4 * (n - 1) + 3] = sb - text;
        }
    }
    return n;
}

}  // extern "C"
