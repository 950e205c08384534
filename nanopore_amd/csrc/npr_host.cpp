// npr_host.cpp -- host-side stages of the realigner: band construction / matrix splitting from the guide
// alignment, the maximum-expected-accuracy chain and cigar emission, model table preparation.
//
// These are the non-DP stages of cactus_realign, the
// program the reference forks per read at nanopore/analyses/utils.py:587.  The DP itself
// (rows a5.3-a5.5) runs on the GPU: npr_kernels.hip.
#include <algorithm>
#include <cmath>
#include <cstring>

#include "npr_internal.h"

namespace npr {

namespace {

struct LatticePoint {
    int64_t x, y;
    bool operator==(const LatticePoint &o) const { return x == o.x && y == o.y; }
};

bool guide_is_global(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops) {
    int64_t ref = 0, read = 0;
    for (int64_t i = 0; i < nops; ++i) {
        const int32_t op = ops[2 * i], len = ops[2 * i + 1];
        if (len < 0) return false;
        switch (op) {
            case NPR_OP_M: ref += len; read += len; break;
            case NPR_OP_I: read += len; break;
            case NPR_OP_D: ref += len; break;
            default: return false;  // utils.py:171-172: only M/I/D survive into the exonerate cigar
        }
    }
    return ref == lX && read == lY;  // utils.py:381-382
}

void finish_segment(Segment &s) {
    s.cells = 0;
    s.max_width = 0;
    for (int32_t w : s.n) {
        s.cells += w;
        s.max_width = std::max(s.max_width, w);
    }
    s.staircase = true;
    for (size_t d = 1; d < s.lo.size(); ++d) {
        const int32_t step = s.lo[d] - s.lo[d - 1];
        if (step != 1 && step != -1) {
            s.staircase = false;
            break;
        }
    }
}

// Band of a segment from its chain of lattice points pts[0]=(0,0) ... pts.back()=(lX,lY) (segment-local,
// non-decreasing in both coordinates).  Between two consecutive points the band on anti-diagonal d is the
// cut of the rectangle they span, widened by `expansion` in x-y units; adjacent anchors on one diagonal
// therefore give a stripe of half-width `expansion` (+1 on odd steps).
void fill_band(Segment &s, const std::vector<LatticePoint> &pts, int64_t expansion) {
    const int64_t lX = s.xe - s.xs, lY = s.ye - s.ys, D = lX + lY;
    s.lo.assign(D + 1, 0);
    s.n.assign(D + 1, 0);
    const size_t last = pts.size() - 1;
    for (size_t k = 0; k < std::max<size_t>(last, 1); ++k) {
        const LatticePoint a = pts[k], b = pts[std::min(k + 1, last)];
        const int64_t d0 = a.x + a.y;
        const int64_t d1 = (k + 1 >= last) ? D : (b.x + b.y - 1);  // the final interval owns its end diagonal
        for (int64_t d = d0; d <= d1; ++d) {
            int64_t lo = std::max(2 * a.x - d, d - 2 * b.y) - expansion;
            int64_t hi = std::min(2 * b.x - d, d - 2 * a.y) + expansion;
            lo = std::max({lo, -d, d - 2 * lY});
            hi = std::min({hi, d, 2 * lX - d});
            if ((lo ^ d) & 1) ++lo;
            if ((hi ^ d) & 1) --hi;
            s.lo[d] = static_cast<int32_t>(lo);
            s.n[d] = static_cast<int32_t>((hi - lo) / 2 + 1);
        }
    }
    finish_segment(s);
}

void close_segment(Plan &plan, LatticePoint origin, LatticePoint corner, int ragged_start, int ragged_end,
                   std::vector<LatticePoint> &pts, int64_t expansion) {
    Segment s;
    s.xs = origin.x, s.ys = origin.y, s.xe = corner.x, s.ye = corner.y;
    s.ragged_start = ragged_start, s.ragged_end = ragged_end;
    for (auto &p : pts) p.x -= origin.x, p.y -= origin.y;
    fill_band(s, pts, expansion);
    plan.segs.push_back(std::move(s));
}

int32_t plan_from_anchors(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, Plan &plan) {
    const int64_t trim = p.constraint_trim, N = p.split_threshold;
    // a5.1: anchors = M columns of the guide, `trim` columns dropped at both ends of each gapless block.
    // The pair of 0-based bases (x,y) is the lattice point (x+1,y+1).
    std::vector<LatticePoint> chain;
    chain.push_back({0, 0});
    int64_t x = 0, y = 0;
    for (int64_t i = 0; i < nops; ++i) {
        const int64_t len = ops[2 * i + 1];
        if (ops[2 * i] == NPR_OP_M) {
            for (int64_t t = trim; t + trim < len; ++t) chain.push_back({x + t + 1, y + t + 1});
            x += len, y += len;
        } else if (ops[2 * i] == NPR_OP_I) {
            y += len;
        } else {
            x += len;
        }
    }
    const LatticePoint corner{lX, lY};
    if (!(chain.back() == corner)) chain.push_back(corner);

    // a5.2: cut the matrix wherever the unanchored rectangle between two consecutive points is larger than
    // N*N cells; each side keeps at most N (or half the gap) of it and the cut ends are "ragged".
    std::vector<LatticePoint> pts{chain[0]};
    LatticePoint origin = chain[0];
    int ragged = 0;
    for (size_t i = 0; i + 1 < chain.size(); ++i) {
        const LatticePoint a = chain[i], b = chain[i + 1];
        const int64_t gx = b.x - a.x, gy = b.y - a.y;
        if (gx * gy > N * N) {
            const int64_t hx = std::min(gx / 2, N), hy = std::min(gy / 2, N);
            const LatticePoint stop{a.x + hx, a.y + hy}, resume{b.x - hx, b.y - hy};
            if (!(pts.back() == stop)) pts.push_back(stop);
            close_segment(plan, origin, stop, ragged, 1, pts, p.diagonal_expansion);
            origin = resume;
            ragged = 1;
            pts.assign(1, resume);
            if (!(b == resume)) pts.push_back(b);
        } else {
            pts.push_back(b);
        }
    }
    close_segment(plan, origin, corner, ragged, 0, pts, p.diagonal_expansion);
    return NPR_OK;
}

// Fixed-width band: on every anti-diagonal the cells whose x-y is within W/2 of where the guide path
// crosses it (a match step jumps over one diagonal; that diagonal takes the step's own x-y).
int32_t plan_fixed_width(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, Plan &plan) {
    Segment s;
    s.xe = lX, s.ye = lY;
    const int64_t D = lX + lY, half = p.fixed_width / 2;
    std::vector<int64_t> centre(D + 1, 0);
    int64_t x = 0, y = 0;
    for (int64_t i = 0; i < nops; ++i) {
        const int32_t op = ops[2 * i];
        for (int64_t t = ops[2 * i + 1]; t > 0; --t) {
            if (op == NPR_OP_M) {
                centre[x + y + 1] = x - y;
                ++x, ++y;
            } else if (op == NPR_OP_D) {
                ++x;
            } else {
                ++y;
            }
            centre[x + y] = x - y;
        }
    }
    s.lo.resize(D + 1);
    s.n.resize(D + 1);
    for (int64_t d = 0; d <= D; ++d) {
        int64_t lo = std::max({centre[d] - half, -d, d - 2 * lY});
        int64_t hi = std::min({centre[d] + half, d, 2 * lX - d});
        if ((lo ^ d) & 1) ++lo;
        if ((hi ^ d) & 1) --hi;
        s.lo[d] = static_cast<int32_t>(lo);
        s.n[d] = static_cast<int32_t>((hi - lo) / 2 + 1);
    }
    finish_segment(s);
    plan.segs.push_back(std::move(s));
    return NPR_OK;
}

}  // namespace

int32_t build_plan(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, Plan &out) {
    out.segs.clear();
    if (lX < 0 || lY < 0 || (nops > 0 && !ops)) return NPR_ERR_INVALID;
    if (!guide_is_global(lX, lY, ops, nops)) return NPR_ERR_INVALID;
    if (lX + lY >= (int64_t(1) << 30)) return NPR_ERR_INVALID;
    // a fixed band narrower than two cells has empty odd anti-diagonals: the lattice falls apart, and the read would only
    // surface as NPR_ERR_ZERO_PROB after a full GPU launch
    if (p.band_mode == NPR_BAND_FIXED && p.fixed_width < 2) return NPR_ERR_INVALID;
    if (p.band_mode == NPR_BAND_FIXED) return plan_fixed_width(p, lX, lY, ops, nops, out);
    if (p.band_mode == NPR_BAND_ANCHOR) return plan_from_anchors(p, lX, lY, ops, nops, out);
    return NPR_ERR_INVALID;
}

// ------------------------------------------------------------------------------------------------------
// MEA chain (a5.6).  Posteriors are quantised to 1e-7; a pair's weight is its posterior minus gapGamma times
// the gap mass of its row and column (1 - sum of match posteriors there); pairs whose weight does not
// exceed matchGamma are dropped; the heaviest chain strictly increasing in x and y is found with a
// prefix-maximum tree over y; ties go to the candidate that sorts last by (x, y).
// ------------------------------------------------------------------------------------------------------
namespace {

struct Best {
    int64_t score = 0;
    int64_t who = -1;
    bool beats(const Best &o) const { return score > o.score || (score == o.score && who > o.who); }
};

class PrefixMax {
  public:
    PrefixMax(std::vector<Best> &store, int64_t size) : t_(store) { t_.assign(size + 2, Best{}); }
    Best query(int64_t upto_exclusive) const {  // best over keys < upto_exclusive
        Best b;
        for (int64_t i = upto_exclusive; i > 0; i &= i - 1)
            if (t_[i].beats(b)) b = t_[i];
        return b;
    }
    void insert(int64_t key, Best v) {
        for (int64_t i = key + 1; i < static_cast<int64_t>(t_.size()); i += i & -i)
            if (v.beats(t_[i])) t_[i] = v;
    }

  private:
    std::vector<Best> &t_;  // caller-owned storage, reused from read to read
};

void push_op(std::vector<int32_t> &ops, size_t first, int32_t op, int64_t len) {
    if (len <= 0) return;
    if (ops.size() > first && ops[ops.size() - 2] == op) {
        ops.back() += static_cast<int32_t>(len);
    } else {
        ops.push_back(op);
        ops.push_back(static_cast<int32_t>(len));
    }
}

}  // namespace

int32_t mea_cigar(int64_t lX, int64_t lY, const Pair *pairs, int64_t n, double gap_gamma, double match_gamma,
                  std::vector<int32_t> &ops, double &score) {
    // Gap mass of a row (reference base) / column (read base) = 1 - sum of the match posteriors on it.  Rows are
    // summed by scanning the (x,y)-sorted list, columns in an array over the read positions the pairs touch:
    // nothing here is sized by the reference length (a chained record spans a whole contig, utils.py:381).
    // scratch is kept per host thread and only grows: with hundreds of worker threads, eight fresh 100 KB+ vectors
    // per read turn into mmap / munmap traffic that serialises them in the kernel
    struct Cand {
        int32_t x, y;
        int64_t q, w;
    };
    thread_local std::vector<int64_t> q, rowsum, colsum, total, back, path;
    thread_local std::vector<Cand> c;
    thread_local std::vector<Best> tree_store;
    q.assign(n, 0), rowsum.assign(n, 0);
    int32_t ymin = 0, ymax = -1;
    for (int64_t i = 0; i < n; ++i) {
        if (pairs[i].x < 0 || pairs[i].x >= lX || pairs[i].y < 0 || pairs[i].y >= lY) return NPR_ERR_INVALID;
        q[i] = static_cast<int64_t>(std::floor(static_cast<double>(pairs[i].p) * static_cast<double>(PROB_ONE)));
        if (i == 0 || pairs[i].y < ymin) ymin = pairs[i].y;
        if (i == 0 || pairs[i].y > ymax) ymax = pairs[i].y;
    }
    for (int64_t g = 0; g < n;) {
        int64_t h = g, sum = 0;
        while (h < n && pairs[h].x == pairs[g].x) sum += q[h++];
        for (int64_t i = g; i < h; ++i) rowsum[i] = sum;
        g = h;
    }
    colsum.assign(n ? ymax - ymin + 1 : 0, 0);
    for (int64_t i = 0; i < n; ++i) colsum[pairs[i].y - ymin] += q[i];
    const int64_t floor_w = static_cast<int64_t>(std::floor(match_gamma * static_cast<double>(PROB_ONE)));
    c.clear();
    c.reserve(n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t gap = std::max<int64_t>(PROB_ONE - rowsum[i], 0) + std::max<int64_t>(PROB_ONE - colsum[pairs[i].y - ymin], 0);
        const int64_t w = q[i] - static_cast<int64_t>(std::floor(gap_gamma * static_cast<double>(gap)));
        if (w > floor_w) c.push_back({pairs[i].x, pairs[i].y, q[i], w});
    }
    // input is sorted by (x,y); filtering keeps the order
    const int64_t m = static_cast<int64_t>(c.size());
    total.assign(m, 0), back.assign(m, 0);
    PrefixMax tree(tree_store, n ? ymax - ymin + 1 : 0);  // keyed by y - ymin
    Best overall;
    for (int64_t g = 0; g < m;) {
        int64_t h = g;
        while (h < m && c[h].x == c[g].x) ++h;
        for (int64_t i = g; i < h; ++i) {
            const Best b = tree.query(c[i].y - ymin);
            total[i] = c[i].w + b.score;
            back[i] = b.who;
        }
        for (int64_t i = g; i < h; ++i) {
            const Best v{total[i], i};
            tree.insert(c[i].y - ymin, v);
            if (v.beats(overall)) overall = v;
        }
        g = h;
    }
    path.clear();
    for (int64_t i = overall.who; i >= 0; i = back[i]) path.push_back(i);
    std::reverse(path.begin(), path.end());

    const size_t first = ops.size();
    int64_t px = -1, py = -1, mass = 0;
    for (int64_t i : path) {
        push_op(ops, first, NPR_OP_D, c[i].x - px - 1);  // unaligned reference bases first,
        push_op(ops, first, NPR_OP_I, c[i].y - py - 1);  // then unaligned read bases,
        push_op(ops, first, NPR_OP_M, 1);                // then the aligned pair
        px = c[i].x, py = c[i].y;
        mass += c[i].q;
    }
    push_op(ops, first, NPR_OP_D, lX - 1 - px);
    push_op(ops, first, NPR_OP_I, lY - 1 - py);
    score = path.empty() ? 0.0 : static_cast<double>(mass) / (static_cast<double>(path.size()) * PROB_ONE);
    return NPR_OK;
}

double rescore(const int32_t *guide_ops, int64_t n_guide_ops, const Pair *pairs, int64_t n) {
    // walk the guide's M columns and the (x,y)-sorted pair list together
    int64_t x = 0, y = 0, columns = 0, k = 0;
    double sum = 0.0;
    for (int64_t i = 0; i < n_guide_ops; ++i) {
        const int32_t op = guide_ops[2 * i];
        const int64_t len = guide_ops[2 * i + 1];
        if (op == NPR_OP_M) {
            for (int64_t t = 0; t < len; ++t, ++x, ++y) {
                while (k < n && (pairs[k].x < x || (pairs[k].x == x && pairs[k].y < y))) ++k;
                if (k < n && pairs[k].x == x && pairs[k].y == y) sum += static_cast<double>(pairs[k].p);
            }
            columns += len;
        } else if (op == NPR_OP_I) {
            y += len;
        } else {
            x += len;
        }
    }
    return columns > 0 ? sum / static_cast<double>(columns) : 0.0;
}

// ------------------------------------------------------------------------------------------------------
// model tables (a7)
// ------------------------------------------------------------------------------------------------------
int32_t make_dev_model(const double *T, const double *E, DevModel &m) {
    // the five-state cell update evaluates exactly these transitions; anything else must be zero
    static const bool used[5][5] = {{1, 1, 1, 1, 1}, {1, 1, 1, 0, 0}, {1, 1, 1, 0, 0}, {1, 0, 0, 1, 0}, {1, 0, 0, 0, 1}};
    for (int a = 0; a < 5; ++a)
        for (int b = 0; b < 5; ++b) {
            const double t = T[a * 5 + b];
            if (!(t >= 0.0) || !std::isfinite(t)) return NPR_ERR_MODEL;
            if (!used[a][b] && t != 0.0) return NPR_ERR_MODEL;
            m.T[a * 5 + b] = static_cast<float>(t);
        }
    for (int i = 0; i < 80; ++i)
        if (!(E[i] >= 0.0) || !std::isfinite(E[i])) return NPR_ERR_MODEL;
    for (int x = 0; x < 5; ++x)
        for (int y = 0; y < 5; ++y) m.em[x * 5 + y] = (x < 4 && y < 4) ? static_cast<float>(E[x * 4 + y]) : 0.0625f;
    for (int s = 0; s < 5; ++s)
        for (int b = 0; b < 5; ++b) {
            double over_read = 0.25, over_ref = 0.25;
            if (b < 4) {
                over_read = over_ref = 0.0;
                for (int o = 0; o < 4; ++o) {
                    over_read += E[s * 16 + b * 4 + o];
                    over_ref += E[s * 16 + o * 4 + b];
                }
            }
            m.ex[s * 5 + b] = static_cast<float>(over_read);
            m.ey[s * 5 + b] = static_cast<float>(over_ref);
        }
    for (int s = 0; s < 5; ++s) {
        m.start[s] = s == 0 ? 1.0f : 0.0f;             // global: start in match
        m.start[5 + s] = (s == 3 || s == 4) ? 1.0f : 0.0f;  // ragged: start inside a long gap
        m.end[s] = m.T[s * 5 + 0];                     // global: close with the transition to match
    }
    m.end[5 + 0] = m.T[0 * 5 + 3];  // ragged: leave by opening / extending a long gap
    m.end[5 + 1] = m.T[0 * 5 + 3];
    m.end[5 + 2] = m.T[0 * 5 + 4];
    m.end[5 + 3] = m.T[3 * 5 + 3];
    m.end[5 + 4] = m.T[4 * 5 + 4];
    return NPR_OK;
}

// Model used when the reference passes no --loadHmm (abstractMapper.py:36-37).  Its numbers live in the
// absent cactus sources: UNPINNED; recalled cPecan defaults, see DESIGN.md.
void stock_model(double *T, double *E) {
    const double cont = 0.9703833696510062, sopen = 0.0129868352330243, sext = 0.7126062401851738,
                 sswitch = 0.0073673675173412815, lext = 0.99656342579062;
    const double lopen = (1.0 - cont - 2.0 * sopen) / 2.0;
    std::memset(T, 0, sizeof(double) * 25);
    T[0] = cont, T[1] = sopen, T[2] = sopen, T[3] = lopen, T[4] = lopen;
    T[5] = 1.0 - sext - sswitch, T[6] = sext, T[7] = sswitch;
    T[10] = 1.0 - sext - sswitch, T[11] = sswitch, T[12] = sext;
    T[15] = 1.0 - lext, T[18] = lext;
    T[20] = 1.0 - lext, T[24] = lext;
    const double same = 0.12064298095701059, transition = 0.018577373224845586, transversion = 0.010396478746046977;
    double sum = 0.0;
    for (int x = 0; x < 4; ++x)
        for (int y = 0; y < 4; ++y) {
            const bool ts = (x ^ y) == 2;  // A<->G, C<->T
            E[x * 4 + y] = x == y ? same : (ts ? transition : transversion);
            sum += E[x * 4 + y];
        }
    for (int i = 0; i < 16; ++i) E[i] /= sum;
    for (int i = 16; i < 80; ++i) E[i] = 1.0 / 16.0;
}

}  // namespace npr
