// npr_internal.h -- shared declarations of libnprealign (host side).  Not installed.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "nprealign.h"
#include "npr_band.h"

namespace npr {

constexpr int32_t E_DEAD = -(1 << 28);     // exponent of a dead (zero-probability) cell
constexpr int64_t PROB_ONE = 10000000LL;  // posterior quantum of the MEA stage

// One independent banded DP problem: the lattice rectangle (xs,ys)-(xe,ye) of a read, cut out by the
// split rule of cactus_realign (--splitMatrixBiggerThanThis), with its per-anti-diagonal band.
struct Segment {
    int64_t xs = 0, ys = 0, xe = 0, ye = 0;
    int32_t ragged_start = 0, ragged_end = 0;
    std::vector<int32_t> lo;  // [D+1] first in-band xmy on each anti-diagonal (segment-local)
    std::vector<int32_t> n;   // [D+1] in-band cells on each anti-diagonal
    int64_t cells = 0;
    int32_t max_width = 0;
    bool staircase = true;  // consecutive diagonals' first x-y differ by exactly +-1 (register kernel eligible)
    int64_t D() const { return (xe - xs) + (ye - ys); }
};

struct Plan {
    std::vector<Segment> segs;
};

// The O(#cigar ops) half of stages a5.1-a5.2: the segments of a read (matrix splits) and, per segment, the chain of lattice
// points whose pieces define its band (npr_band.h).  The band rows themselves are expanded per anti-diagonal -- by
// build_plan on the host, by npr_plan.hip on the device.
struct SegPlan {
    int64_t xs = 0, ys = 0, xe = 0, ye = 0;
    int32_t ragged_start = 0, ragged_end = 0;
    int64_t point_first = 0;  // first point of the segment in PointPlan::points (pieces + 1 of them)
    int32_t pieces = 0;
    int64_t owner = 0;        // the read it belongs to (set by the caller that plans many reads into one PointPlan)
};
struct PointPlan {
    std::vector<SegPlan> segs;
    std::vector<PlanPoint> points;
};
int32_t plan_points(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, PointPlan &out);

// stages a5.1-a5.2: anchors from the guide, band, split.  Returns NPR_OK / NPR_ERR_INVALID.
int32_t build_plan(const npr_params &p, int64_t lX, int64_t lY, const int32_t *ops, int64_t nops, Plan &out);

struct Pair {
    int32_t x, y;
    float p;
};

// stage a5.6: gapGamma-reweighted maximum-expected-accuracy chain -> global cigar.
// pairs must be sorted by (x, y).  Appends (op,len) pairs to `ops`; returns NPR_OK.
int32_t mea_cigar(int64_t lX, int64_t lY, const Pair *pairs, int64_t n, double gap_gamma, double match_gamma,
                  std::vector<int32_t> &ops, double &score);

// stage a5.7: mean posterior of the guide's M columns (pairs sorted by (x,y)).
double rescore(const int32_t *guide_ops, int64_t n_guide_ops, const Pair *pairs, int64_t n);

inline uint8_t encode_base(uint8_t c) {
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

// Device-side model tables (fp32), one per model slot.
struct DevModel {
    float T[25];      // T[from*5+to]
    float em[25];     // match emission [x*5+y], index 4 = N (flat 1/16)
    float ex[25];     // gap-X emission [state*5+x]: marginal of the state's block over the read base
    float ey[25];     // gap-Y emission [state*5+y]: marginal over the reference base
    float start[10];  // [ragged*5+state]
    float end[10];    // [ragged*5+state]
};

int32_t make_dev_model(const double *T25, const double *E80, DevModel &m);
void stock_model(double *T25, double *E80);

}  // namespace npr
