// npr_chain.cpp -- chaining of a read's local hits into the one co-linear chain the realigner is seeded with
// (SURVEY.md 8f next #1).  Host code, no GPU.
//
// The reference's chainFn (nanopore/analyses/utils.py:388-426) sorts the hits of a (read, reference) pair by their first
// reference position and, for every hit, looks at ALL earlier ones ("sloppy quadratic algorithm", :401): hit j may precede
// hit i when i starts after j ends in both sequences, both lie on the same strand and the two gaps sum to at most maxGap;
// a chain's score is the sum of its hits' scores (aligned pairs).  Because both gaps are at least one base, a predecessor
// of i ends within maxGap reference bases before i starts -- so instead of all earlier hits only the hits whose END lies in
// that window are examined: sort by start, sort by end, binary-search the window.  Same chain, ties included: among equally
// good predecessors the one that sorts first wins (the reference replaces only on a strictly better sum), among equally
// good chain ends the one that sorts last (sorted(...)[-1], :416).
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "nprealign.h"

extern "C" int64_t npr_chain_hits(int64_t n, const int64_t *ref_start, const int64_t *read_start, const int64_t *ref_end,
                                  const int64_t *read_end, const uint8_t *reverse, const int64_t *score, int64_t max_gap, int64_t *chain) {
    if (n < 0 || (n && (!ref_start || !read_start || !ref_end || !read_end || !reverse || !score || !chain))) return NPR_ERR_INVALID;
    if (n == 0) return 0;
    std::vector<int64_t> order(n), pos(n), by_end(n), best(score, score + n), back(n, -1);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return ref_start[a] < ref_start[b]; });
    for (int64_t k = 0; k < n; ++k) pos[order[k]] = k;
    std::iota(by_end.begin(), by_end.end(), 0);
    std::sort(by_end.begin(), by_end.end(), [&](int64_t a, int64_t b) { return ref_end[a] != ref_end[b] ? ref_end[a] < ref_end[b] : pos[a] < pos[b]; });
    for (int64_t k = 0; k < n; ++k) {
        const int64_t i = order[k];
        // hits whose last reference position lies in [ref_start[i] - max_gap + 1, ref_start[i] - 1]
        const int64_t lo_end = ref_start[i] - max_gap + 1;
        auto it = std::lower_bound(by_end.begin(), by_end.end(), lo_end, [&](int64_t j, int64_t v) { return ref_end[j] < v; });
        int64_t pick = -1;
        for (; it != by_end.end() && ref_end[*it] < ref_start[i]; ++it) {
            const int64_t j = *it;
            if (pos[j] >= k) continue;  // the reference only looks at hits that sort before i
            if (read_start[i] <= read_end[j] || reverse[i] != reverse[j]) continue;
            if ((ref_start[i] - ref_end[j]) + (read_start[i] - read_end[j]) > max_gap) continue;
            if (pick < 0 || best[j] > best[pick] || (best[j] == best[pick] && pos[j] < pos[pick])) pick = j;
        }
        if (pick >= 0 && score[i] + best[pick] > best[i]) best[i] = score[i] + best[pick], back[i] = pick;
    }
    int64_t end = order[0];
    for (int64_t k = 1; k < n; ++k)
        if (best[order[k]] >= best[end]) end = order[k];
    int64_t len = 0;
    for (int64_t i = end; i >= 0; i = back[i]) chain[len++] = i;
    std::reverse(chain, chain + len);
    return len;
}

// The global alignment a chain of local hits stands for (mergeChainedAlignedReads, nanopore/analyses/utils.py:295-386, with
// the spans it asserts at :381-382): block k is a local alignment whose first aligned pair sits at reference position
// ref_pos[k] and at position read_pos[k] of the record's SEQ orientation (its leading hard + soft clips -- the same number
// on either strand, since a reverse-strand SEQ is the reverse complement of the read and the merged record's SEQ is too),
// with the M / I / D operations ops[2 * ops_off[k] ..).  Between consecutive blocks, before the first and after the last,
// the unaligned reference bases become one D and the unaligned read bases one I (in that order); neighbours of the same
// kind are merged into one operation.  The blocks must be in chain order and must not overlap (NPR_ERR_INVALID: the
// reference's asserts :344, :350, :366-375), and the result spans exactly ref_len x read_len.
extern "C" int64_t npr_chain_merge(int64_t n_blocks, const int64_t *ref_pos, const int64_t *read_pos, const int64_t *ops_off,
                                   const int32_t *ops, int64_t ref_len, int64_t read_len, int32_t *out_ops, int64_t cap_pairs) {
    if (n_blocks < 0 || ref_len < 0 || read_len < 0 || (n_blocks && (!ref_pos || !read_pos || !ops_off)) || cap_pairs < 0 || (cap_pairs && !out_ops))
        return NPR_ERR_INVALID;
    int64_t n = 0, x = 0, y = 0;  // operations written; reference / read bases consumed
    bool overflow = false;
    auto put = [&](int32_t op, int64_t len) {
        if (len <= 0) return;
        if (n > 0 && out_ops[2 * (n - 1)] == op && !overflow) {
            out_ops[2 * (n - 1) + 1] += static_cast<int32_t>(len);
            return;
        }
        if (n >= cap_pairs) {
            overflow = true;
            return;
        }
        out_ops[2 * n] = op, out_ops[2 * n + 1] = static_cast<int32_t>(len), ++n;
    };
    for (int64_t k = 0; k < n_blocks; ++k) {
        if (ref_pos[k] < x || read_pos[k] < y) return NPR_ERR_INVALID;  // out of order or overlapping its predecessor
        put(NPR_OP_D, ref_pos[k] - x), x = ref_pos[k];
        put(NPR_OP_I, read_pos[k] - y), y = read_pos[k];
        for (int64_t q = ops_off[k]; q < ops_off[k + 1]; ++q) {
            const int32_t op = ops[2 * q], len = ops[2 * q + 1];
            if (op < 0 || op > 2 || len < 0) return NPR_ERR_INVALID;
            put(op, len);
            if (op != NPR_OP_I) x += len;
            if (op != NPR_OP_D) y += len;
        }
    }
    if (x > ref_len || y > read_len) return NPR_ERR_INVALID;
    put(NPR_OP_D, ref_len - x);
    put(NPR_OP_I, read_len - y);
    return overflow ? NPR_ERR_CAPACITY : n;
}
