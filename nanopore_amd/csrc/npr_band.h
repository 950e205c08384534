// npr_band.h -- the band of a segment as a short list of lattice points, and the one function that turns it into a band
// row.  Shared by the host planner (npr_host.cpp: build_plan, the npr_plan_* entry points) and the device planner
// (npr_plan.hip), so that both expand exactly the same rows: the host does the O(#cigar ops) part of cactus_realign's
// band construction (anchors, trimming, matrix splits), the device the O(#anti-diagonals)
// part.
//
// A segment's plan is a chain of points P_0 = (0, 0) ... P_m = (lX, lY) (segment-local lattice coordinates, non-decreasing
// in both); piece k runs from P_k to P_{k+1} and owns the anti-diagonals [x_k + y_k, x_{k+1} + y_{k+1} - 1], the last piece
// also the segment's last anti-diagonal.
//   NPR_BAND_FIXED   one piece per cigar operation of the guide; the band is centred on the guide path: x - y of the path
//                    on anti-diagonal d (a match step jumps over one anti-diagonal, which takes the step's own x - y),
//                    +- fixed_width / 2.
//   NPR_BAND_ANCHOR  a RECT piece spans the unanchored rectangle between two consecutive anchor points: the band is the cut
//                    of the rectangle, widened by diagonal_expansion in x - y; a DIAG piece (flag on its first point) is a
//                    run of anchor points on one diagonal, i.e. a chain of 1 x 1 rectangles.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define NPR_HD __host__ __device__
#else
#define NPR_HD
#endif

namespace npr {

struct PlanPoint {
    int32_t x;
    uint32_t y;  // bit 31: the piece that starts here is a DIAG run
    NPR_HD int32_t yy() const { return static_cast<int32_t>(y & 0x7fffffffu); }
    NPR_HD bool diag() const { return (y >> 31) != 0; }
    NPR_HD int32_t d0() const { return x + yy(); }
};

struct BandRow {
    int32_t lo, n;
};

// index of the piece that owns anti-diagonal d: the last k in [0, m) with d0(P_k) <= d
NPR_HD inline int32_t band_piece(const PlanPoint *P, int32_t m, int32_t d) {
    int32_t lo = 0, hi = m - 1;
    while (lo < hi) {
        const int32_t mid = (lo + hi + 1) >> 1;
        if (P[mid].d0() <= d) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// band row of anti-diagonal d; `width` = diagonal_expansion (anchor mode) or fixed_width / 2 (fixed mode)
NPR_HD inline BandRow band_row_of_piece(int32_t fixed_mode, int32_t width, int32_t lX, int32_t lY, const PlanPoint &a,
                                        const PlanPoint &b, int32_t d) {
    const int32_t ax = a.x, ay = a.yy(), bx = b.x, by = b.yy();
    int32_t lo, hi;
    if (fixed_mode) {
        const int32_t slope = (bx > ax && by > ay) ? 0 : (bx > ax ? 1 : -1);
        const int32_t centre = (ax - ay) + slope * (d - (ax + ay));
        lo = centre - width, hi = centre + width;
    } else if (a.diag()) {
        const int32_t odd = (d - (ax + ay)) & 1;
        lo = (ax - ay) - odd - width, hi = (ax - ay) + odd + width;
    } else {
        const int32_t l0 = 2 * ax - d, l1 = d - 2 * by, h0 = 2 * bx - d, h1 = d - 2 * ay;
        lo = (l0 > l1 ? l0 : l1) - width, hi = (h0 < h1 ? h0 : h1) + width;
    }
    // the lattice: 0 <= x <= lX, 0 <= y <= lY
    if (lo < -d) lo = -d;
    if (lo < d - 2 * lY) lo = d - 2 * lY;
    if (hi > d) hi = d;
    if (hi > 2 * lX - d) hi = 2 * lX - d;
    if ((lo ^ d) & 1) ++lo;  // x - y has the parity of x + y
    if ((hi ^ d) & 1) --hi;
    return BandRow{lo, (hi - lo) / 2 + 1};
}

NPR_HD inline BandRow band_row(int32_t fixed_mode, int32_t width, int32_t lX, int32_t lY, const PlanPoint *P, int32_t m, int32_t d) {
    const int32_t k = band_piece(P, m, d);
    return band_row_of_piece(fixed_mode, width, lX, lY, P[k], P[k + 1], d);
}

}  // namespace npr
