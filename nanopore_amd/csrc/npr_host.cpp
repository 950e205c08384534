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

PlanPoint point(int64_t x, int64_t y, bool diag = false) {
    return PlanPoint{static_cast<int32_t>(x), static_cast<uint32_t>(y) | (diag ? 0x80000000u : 0u)};
}

// closes the segment whose points (absolute coordinates) were collected in `pts`: origin .. corner
void close_segment(PointPlan &plan, int64_t ox, int64_t oy, int64_t cx, int64_t cy, int ragged_start, int ragged_end,
                   std::vector<PlanPoint> &pts) {
    SegPlan s;
    s.xs = ox, s.ys = oy, s.xe = cx, s.ye = cy;
    s.ragged_start = ragged_start, s.ragged_end = ragged_end;
    s.point_first = static_cast<int64_t>(plan.points.size());
    if (pts.size() < 2) pts.push_back(pts.back());  // a segment has at least one piece
    for (const PlanPoint &p : pts) plan.points.push_back(point(p.x - ox, p.yy() - oy, p.diag()));
    s.pieces = static_cast<int32_t>(pts.size() - 1);
    plan.segs.push_back(s);
}

// a5.1: anchors = M columns of the guide, `trim` columns dropped at both ends of each gapless block; the pair of 0-based
// bases (x, y) is the lattice point (x + 1, y + 1).  The anchors of one block are a run on one diagonal.
// a5.2: the matrix is cut wherever the unanchored rectangle between two consecutive points is larger than N * N cells; each
// side keeps at most N (or half the gap) of it and the cut ends are "ragged".
int32_t points_from_anchors(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, PointPlan &plan) {
    const int64_t trim = p.constraint_trim, N = p.split_threshold;
    std::vector<PlanPoint> pts{point(0, 0)};
    int64_t ox = 0, oy = 0;      // origin of the current segment
    int64_t ax = 0, ay = 0;      // the last chain point
    int ragged = 0;
    // link the chain point (ax, ay) to (bx, by): an unanchored rectangle
    auto gap = [&](int64_t bx, int64_t by) {
        const int64_t gx = bx - ax, gy = by - ay;
        if (gx * gy > N * N) {
            const int64_t hx = std::min(gx / 2, N), hy = std::min(gy / 2, N);
            const int64_t sx = ax + hx, sy = ay + hy, rx = bx - hx, ry = by - hy;
            if (!(pts.back().x == sx && pts.back().yy() == sy)) pts.push_back(point(sx, sy));
            close_segment(plan, ox, oy, sx, sy, ragged, 1, pts);
            ox = rx, oy = ry, ragged = 1;
            pts.assign(1, point(rx, ry));
            if (!(bx == rx && by == ry)) pts.push_back(point(bx, by));
        } else {
            pts.push_back(point(bx, by));
        }
        ax = bx, ay = by;
    };
    int64_t x = 0, y = 0;
    for (int64_t i = 0; i < nops; ++i) {
        const int64_t len = ops[2 * i + 1];
        if (ops[2 * i] == NPR_OP_M) {
            const int64_t m = len - 2 * trim;  // anchors of this block: (x + trim + 1 + t, y + trim + 1 + t), t = 0 .. m-1
            if (m > 0) {
                const int64_t fx = x + trim + 1, fy = y + trim + 1;
                gap(fx, fy);
                if (m > 1) {
                    if (N >= 1) {  // unit steps along the diagonal: 1 x 1 rectangles, never cut
                        pts.back() = point(fx, fy, true);
                        pts.push_back(point(fx + m - 1, fy + m - 1));
                        ax = fx + m - 1, ay = fy + m - 1;
                    } else {       // splitMatrixBiggerThanThis = 0 cuts between any two points
                        for (int64_t t = 1; t < m; ++t) gap(fx + t, fy + t);
                    }
                }
            }
            x += len, y += len;
        } else if (ops[2 * i] == NPR_OP_I) {
            y += len;
        } else {
            x += len;
        }
    }
    if (!(ax == lX && ay == lY)) gap(lX, lY);
    close_segment(plan, ox, oy, lX, lY, ragged, 0, pts);
    return NPR_OK;
}

// Fixed-width band: one piece per operation of the guide (band_row centres the band on the guide path).
int32_t points_fixed_width(int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, PointPlan &plan) {
    SegPlan s;
    s.xe = lX, s.ye = lY;
    s.point_first = static_cast<int64_t>(plan.points.size());
    int64_t x = 0, y = 0;
    plan.points.push_back(point(0, 0));
    for (int64_t i = 0; i < nops; ++i) {
        const int64_t len = ops[2 * i + 1];
        if (ops[2 * i] != NPR_OP_I) x += len;
        if (ops[2 * i] != NPR_OP_D) y += len;
        plan.points.push_back(point(x, y));
    }
    if (nops == 0) plan.points.push_back(point(0, 0));
    s.pieces = static_cast<int32_t>(plan.points.size() - s.point_first - 1);
    plan.segs.push_back(s);
    return NPR_OK;
}

}  // namespace

int32_t plan_points(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, PointPlan &out) {
    if (lX < 0 || lY < 0 || (nops > 0 && !ops)) return NPR_ERR_INVALID;
    if (!guide_is_global(lX, lY, ops, nops)) return NPR_ERR_INVALID;
    if (lX + lY >= (int64_t(1) << 30)) return NPR_ERR_INVALID;
    // a fixed band narrower than two cells has empty odd anti-diagonals: the lattice falls apart, and the read would only
    // surface as NPR_ERR_ZERO_PROB after a full GPU launch
    if (p.band_mode == NPR_BAND_FIXED && p.fixed_width < 2) return NPR_ERR_INVALID;
    if (p.band_mode == NPR_BAND_FIXED) return points_fixed_width(lX, lY, ops, nops, out);
    if (p.band_mode == NPR_BAND_ANCHOR) return points_from_anchors(p, lX, lY, ops, nops, out);
    return NPR_ERR_INVALID;
}

int32_t build_plan(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, Plan &out) {
    out.segs.clear();
    PointPlan pp;
    const int32_t rc = plan_points(p, lX, lY, ops, nops, pp);
    if (rc != NPR_OK) return rc;
    const int32_t fixed = p.band_mode == NPR_BAND_FIXED;
    const int32_t width = fixed ? p.fixed_width / 2 : p.diagonal_expansion;
    for (const SegPlan &sp : pp.segs) {
        Segment s;
        s.xs = sp.xs, s.ys = sp.ys, s.xe = sp.xe, s.ye = sp.ye;
        s.ragged_start = sp.ragged_start, s.ragged_end = sp.ragged_end;
        const int32_t slX = static_cast<int32_t>(sp.xe - sp.xs), slY = static_cast<int32_t>(sp.ye - sp.ys), D = slX + slY;
        const PlanPoint *P = pp.points.data() + sp.point_first;
        s.lo.resize(D + 1), s.n.resize(D + 1);
        int32_t k = 0;
        for (int32_t d = 0; d <= D; ++d) {
            while (k + 1 < sp.pieces && P[k + 1].d0() <= d) ++k;
            const BandRow r = band_row_of_piece(fixed, width, slX, slY, P[k], P[k + 1], d);
            s.lo[d] = r.lo, s.n[d] = r.n;
        }
        finish_segment(s);
        out.segs.push_back(std::move(s));
    }
    return NPR_OK;
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
        // the list is a caller's (npr_mea_cigar is public): values that are no probabilities (NaN, negative, above 1) and pairs out of (x, y) order or
        // twice in the list are refused -- the chain below assumes the order, and a weight made of NaN is undefined behaviour before it is a wrong cigar
        if (!(pairs[i].p >= 0.f && pairs[i].p <= 1.0f + 0x1p-10f)) return NPR_ERR_INVALID;  // (fp32 rounding puts a certain match a few 2^-23 above 1)
        if (i > 0 && (pairs[i].x < pairs[i - 1].x || (pairs[i].x == pairs[i - 1].x && pairs[i].y <= pairs[i - 1].y))) return NPR_ERR_INVALID;
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
