// npr_sched.h -- per-anti-diagonal schedules derived from a segment's band rows, written once for the host and the device:
// the frame schedule of the register kernels on a frame that follows the anti-diagonal (k_dp_stair / k_dp_wide: control
// words) and the stripe table of k_dp_tile.  The host runs them for the npr_plan_* entry points and the tests, the device
// (npr_plan.hip) when a batch is staged.
#pragma once
#include <cstdint>

#include "npr_band.h"
#include "npr_device.h"

namespace npr {

// Frame schedule (see npr_kernel_stair.hip).  The wavefront(s) hold a frame of C = 64*R*NW slots of the current
// anti-diagonal, slot j = lattice point (x0 + j, y0 - j); the frame takes an X-step (x0 += 1) into every odd anti-diagonal
// and a Y-step (y0 += 1) into every even one, so its first x-y, flo, just alternates.  The band (first x-y `lo`, n cells)
// must stay inside the frame; when it drifts to an edge the frame is REBASED by one slot (flo +- 2) between two
// anti-diagonals.  A rebase towards higher x-y may only precede an X-step and one towards lower x-y a Y-step (the kernel
// re-injects the base that left the wavefront at the step before), so the decision looks one anti-diagonal ahead.  Control
// words per anti-diagonal: row offset in the forward scratch (cells), and jlo | n << 13 | (rebase + 1) << 26.  Returns false
// when the band cannot be followed; `ctl` and `cells` may be null.
//
// The one-wavefront class with two slots per lane (k_dp_stair<2>, the north-star class) gets its words READY TO USE
// instead (stair_packed): the scalar unit is shared by a CU's four SIMDs and issues about one instruction per cycle
// (tools/issue_mix), and deriving two lane masks and a row address from (jlo, n, offset) took ~40 of the ~80 scalar
// instructions of a step.  Word 0: byte offset of where lane 0 of the row would land in the task's scratch, biased by
// row_bias (added to the lanes' own offsets in the row's buffer instruction).  Word 1: lo0 | lo1 << 7 | w0 << 14 | w1 << 21 |
// (rebase + 1) << 28 -- slot r of the lanes [lo_r, lo_r + w_r) is inside the band (w_r <= 63: the class takes bands of at
// most 126 cells).  jlo = lo0 + lo1, n = w0 + w1.
struct StairState {
    int32_t flo;   // x-y of slot 0 of the current frame
    uint32_t off;  // scratch cells of the rows so far (a schedule that needs 2^32 or more is refused)
    int32_t a1, b1, a2, b2;  // slots [a, b) of the band on the previous anti-diagonal / the one before, in the frame's present slots
};
NPR_HD inline bool stair_packed(int R, int NW) { return R == 2 && NW == 1; }
// widest band a class takes
NPR_HD inline int32_t stair_max_width(int R, int NW) { return 64 * R * NW - 1 - (stair_packed(R, NW) ? 1 : 0); }
NPR_HD inline bool stair_begin(StairState &st, int32_t lo0, int32_t n0, int32_t max_width, int R, int NW) {
    const int32_t C = 64 * R * NW;
    if (n0 != 1 || max_width > stair_max_width(R, NW)) return false;
    st.flo = lo0 - 2 * ((C - 1) / 2);
    st.off = 0;
    st.a1 = st.b1 = st.a2 = st.b2 = 0;
    return true;
}
// one anti-diagonal: its band (lo, n), the next one's (lo_nx, n_nx; ignored when d == D) -> its two control words.
// R is 1, 2 or 4: rshift = log2 R (the device walks this loop on one lane; no divisions).
NPR_HD inline bool stair_step(StairState &st, int32_t d, int32_t D, int32_t lo, int32_t n, int32_t lo_nx, int32_t n_nx, int rshift, int32_t C,
                              uint32_t &w0, uint32_t &w1) {
    // Written without data-dependent branches (bitwise logic and selects; only d's parity and the class are tested, which are
    // the same for all the segments a wavefront walks): on the device every lane walks its own segment, and the nested
    // ifs of the first version compiled to twenty exec-mask branches per step -- 1500 cycles per anti-diagonal.
    const int32_t span = 2 * (C - 1), R = 1 << rshift;
    const int32_t hi = lo + 2 * (n - 1), hi_nx = lo_nx + 2 * (n_nx - 1);
    int ok = n >= 1;
    int32_t reb = 0;
    if (d > 0) {
        const int next = d < D;
        if (d & 1) {  // X-step; the next one is a Y-step
            const int32_t f = st.flo + 1;
            reb = (hi > f + span) | (next & (hi_nx > f - 1 + span));
            st.flo = f + 2 * reb;
        } else {
            const int32_t f = st.flo - 1;
            reb = -((lo < f) | (next & (lo_nx < f + 1)));
            st.flo = f + 2 * reb;
        }
        ok &= (lo >= st.flo) & (hi <= st.flo + span);
    }
    const int32_t jlo = (lo - st.flo) >> 1;  // lo >= flo, same parity
    const int32_t l0 = jlo >> rshift, l1 = (jlo + n + R - 1) >> rshift;
    const uint32_t row = static_cast<uint32_t>(l1 - l0) << rshift;
    // Bit 30 of the second word: the band does not occupy the slots it occupied two anti-diagonals ago (a rebase moves a
    // held row by a slot: +1 takes every slot's upper neighbour).  k_dp_rs writes a row over the one two anti-diagonals
    // away under the band's lane mask and clears what lies outside it only when this says that something may.
    const int32_t a2 = st.a2 - reb, b2 = st.b2 - reb;
    const uint32_t moved = (d >= 2) & ((a2 != jlo) | (b2 != jlo + n));
    st.a2 = st.a1 - reb, st.b2 = st.b1 - reb;
    st.a1 = jlo, st.b1 = jlo + n;
    if (rshift == 1 && C == 128) {  // stair_packed
        const uint32_t lo0 = static_cast<uint32_t>(jlo + 1) >> 1, lo1 = static_cast<uint32_t>(jlo) >> 1;
        const uint32_t hi0 = static_cast<uint32_t>(jlo + n + 1) >> 1, hi1 = static_cast<uint32_t>(jlo + n) >> 1;
        ok &= (hi0 - lo0 <= 63u) & (hi1 - lo1 <= 63u);
        ok &= st.off + row < (1u << 29) - 512u;  // the row offsets are 32-bit byte offsets (stair_fits)
        w0 = ((st.off - 2u * lo1) << 3) + row_bias<2>();
        w1 = lo0 | (lo1 << 7) | ((hi0 - lo0) << 14) | ((hi1 - lo1) << 21) | (static_cast<uint32_t>(reb + 1) << 28) | (moved << 30);
    } else {
        ok &= st.off + row >= st.off;  // 2^32 cells
        w0 = st.off;
        w1 = static_cast<uint32_t>(jlo) | (static_cast<uint32_t>(n) << 13) | (static_cast<uint32_t>(reb + 1) << 26) | (moved << 30);
    }
    st.off += row;
    return ok != 0;
}
NPR_HD inline int stair_rshift(int R) { return R == 1 ? 0 : (R == 2 ? 1 : 2); }
NPR_HD inline bool stair_schedule(const int32_t *lo_, const int32_t *n_, int64_t D, int32_t max_width, int R, int NW, uint32_t *ctl,
                                  int64_t *cells) {
    StairState st;
    if (D < 0 || !stair_begin(st, lo_[0], n_[0], max_width, R, NW)) return false;
    const int rshift = stair_rshift(R);
    const int32_t C = 64 * R * NW, Di = static_cast<int32_t>(D);
    for (int32_t d = 0; d <= Di; ++d) {
        uint32_t w0, w1;
        if (!stair_step(st, d, Di, lo_[d], n_[d], d < Di ? lo_[d + 1] : 0, d < Di ? n_[d + 1] : 0, rshift, C, w0, w1)) return false;
        if (ctl) ctl[2 * d] = w0, ctl[2 * d + 1] = w1;
    }
    if (cells) *cells = static_cast<int64_t>(st.off);
    return true;
}

// Stripe table of k_dp_tile (npr_kernel_tile.hip): the lattice columns 0..lX cut into stripes of 64*R columns; per stripe
// the first / last anti-diagonal on which the band has cells in it and the index of its first row in the task's scratch
// (one row per anti-diagonal of a stripe).  stripe_ranges finds df / dl of the S = lX / (64 R) + 1 stripes (arrays with a
// stride, so that the device can write straight into the table) and returns the rows; stripe_fill completes the table
// (out[0] is the header {stripes, rows}).
NPR_HD inline int64_t stripe_ranges(const int32_t *lo_, const int32_t *n_, int64_t D, int64_t lX, int R, int32_t *df, int32_t *dl, int stride) {
    const int64_t K = 64 * R;
    const int64_t S = lX / K + 1;
    for (int64_t k = 0; k < S; ++k) df[k * stride] = 1, dl[k * stride] = 0;
    for (int64_t d = 0; d <= D; ++d) {
        if (n_[d] < 1) continue;
        const int64_t xlo = (d + lo_[d]) >> 1, xhi = xlo + n_[d] - 1;
        int64_t k0 = xlo / K, k1 = xhi / K;
        if (k0 < 0) k0 = 0;
        if (k1 > S - 1) k1 = S - 1;
        for (int64_t k = k0; k <= k1; ++k) {
            if (dl[k * stride] < df[k * stride]) df[k * stride] = static_cast<int32_t>(d);
            dl[k * stride] = static_cast<int32_t>(d);
        }
    }
    int64_t rows = 0;
    for (int64_t k = 0; k < S; ++k)
        if (dl[k * stride] >= df[k * stride]) rows += dl[k * stride] - df[k * stride] + 1;
    return rows;
}
// What k_dp_tile<2> needs to know about one row (anti-diagonal d of a stripe whose first lattice column is X), ready to
// use: which lanes hold a band cell in their slot 0 / slot 1 (slot j = column X + j, two slots per lane).  As for
// k_dp_stair<2> (stair_packed above) the scalar unit is the scarce resource: clipping the band to the stripe and building
// the two masks from (lo, n) took ~35 scalar instructions per step.  Word: lo0 | sh0 << 6 | lo1 << 12 | sh1 << 18 with
// mask_r = (~0 << lo_r) & (~0 >> sh_r), every field 0..63 (s_lshl_b64 / s_lshr_b64 take six bits of the count); an
// empty run is lo = 1, sh = 63.
NPR_HD inline uint32_t tile_row_word(int32_t d, int32_t lo, int32_t n, int32_t X) {
    const int32_t xlo = (d + lo) >> 1;  // lo has the parity of d
    const int32_t j0 = xlo - X > 0 ? xlo - X : 0, j1 = xlo + n - 1 - X < 127 ? xlo + n - 1 - X : 127;
    uint32_t w = 0;
    for (int r = 0; r < 2; ++r) {
        int32_t a = 1, b = 1;  // lanes [a, b)
        if (j1 >= j0) a = (j0 - r + 1) >> 1, b = (j1 + 1 - r + 1) >> 1;
        uint32_t f = 1u | (63u << 6);
        if (b > a) f = static_cast<uint32_t>(a) | (static_cast<uint32_t>(64 - b) << 6);
        w |= f << (12 * r);
    }
    return w;
}

// out[1 + k].df / .dl already hold the ranges
NPR_HD inline void stripe_fill(Stripe *out, int64_t lX, int R) {
    const int64_t K = 64 * R;
    const int64_t S = lX / K + 1;
    int64_t rows = 0;
    for (int64_t k = 0; k < S; ++k) {
        Stripe &st = out[1 + k];
        st.X = static_cast<int32_t>(k * K);
        st.K = static_cast<int32_t>(K);
        st.row0 = static_cast<uint32_t>(rows);
        st.pad[0] = st.pad[1] = st.pad[2] = 0;
        if (st.dl >= st.df) rows += st.dl - st.df + 1;
    }
    out[0].X = static_cast<int32_t>(S);
    out[0].K = static_cast<int32_t>(rows);
    out[0].df = out[0].dl = 0;
    out[0].row0 = 0;
    out[0].pad[0] = out[0].pad[1] = out[0].pad[2] = 0;
}

}  // namespace npr
