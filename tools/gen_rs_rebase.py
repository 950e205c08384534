#!/usr/bin/env python3
"""tools/gen_rs_rebase.py -- prints the in-place frame-rebase helpers of nanopore_amd/csrc/npr_rs.h (the block between the
"generated" markers): one asm statement per register group with the "no rebase" test inside it, so that the compiler sees no
control flow around the 20-odd registers of the held rows.  Two kinds of statement:
  rows:    the cells of the held anti-diagonals move by one slot (dir +1: every slot takes its upper neighbour, the vacated top
           slot takes 0; -1: its lower neighbour);
  streams: the base streams move with them; the base a stream takes in at its open end is either the one that left it at the
           last step (a scalar) or the next one of its feed (read with v_readlane from the feed's current or next block) --
           which of the two depends on the sweep and the direction: forward up X <- feed, Y <- cap; forward down X <- cap,
           Y <- feed; backward up X <- cap, Y <- feed; backward down X <- feed, Y <- cap.
Bring-up tool: the output is pasted into the header."""


def dpp(reg, sh, zero):
    return "v_mov_b32_dpp %s, %s wave_%s:1 row_mask:0xf bank_mask:0xf%s\\n\\t" % (reg, reg, sh, " bound_ctrl:0" if zero else "")


def rot(g, R, up):
    s = ""
    if up:
        for r in range(R - 1):
            s += "v_swap_b32 %s, %s\\n\\t" % (g[r], g[r + 1])
    else:
        for r in range(R - 1, 0, -1):
            s += "v_swap_b32 %s, %s\\n\\t" % (g[r], g[r - 1])
    return s


def frame(d, up, down):
    """The statement's control flow.  The rebase itself lies OUT OF LINE (.subsection 1: behind the kernel's own code, in its
    section), so that the anti-diagonals without a rebase -- nineteen in twenty -- fall through one compare and one branch that is
    not taken; laid out in line they jumped over the two bodies, a taken branch per step and direction on the serial chain of
    every step (round 4: a launch that is one long read's chain)."""
    return ('    asm volatile("s_cmp_lg_u32 %s, 0\\n\\t"\n                 "s_cbranch_scc1 9f\\n\\t"\n                 "2:\\n\\t"\n'
            '                 ".subsection 1\\n\\t"\n                 "9:\\n\\t"\n                 "s_cmp_lt_i32 %s, 0\\n\\t"\n                 "s_cbranch_scc1 1f\\n\\t"\n'
            '                 "%s"\n                 "s_branch 2b\\n\\t"\n                 "1:\\n\\t"\n                 "%s"\n                 "s_branch 2b\\n\\t"\n'
            '                 ".subsection 0"\n' % (d, d, up, down))


def rows(R, nrows):
    k, groups, ops = 0, [], []
    for row in range(nrows):
        for st in ("m", "sx", "sy", "lx", "ly"):
            g = []
            for r in range(R):
                g.append("%%%d" % k)
                k += 1
                ops.append('"+v"(%s.c[%d].%s)' % ("P" if row == 0 else "Q", r, st))
            groups.append(g)
    d = "%%%d" % k
    def body(up):
        s = "".join(rot(g, R, up) for g in groups) + "s_nop 1\\n\\t"
        for g in groups:
            s += dpp(g[R - 1] if up else g[0], "shl" if up else "shr", True)
        return s
    sig = "RDiag<%d> &P, RDiag<%d> &Q" % (R, R) if nrows == 2 else "RDiag<%d> &P" % R
    out = "__device__ __forceinline__ void rs_rebase_rows(%s, int dir) {\n" % sig
    out += frame(d, body(True), body(False))
    out += '                 : %s\n                 : "s"(dir)\n                 : "scc");\n}\n' % ", ".join(ops)
    return out


def streams(R, fwd):
    k = 0
    X = ["%%%d" % (k + i) for i in range(R)]; k += R
    Y = ["%%%d" % (k + i) for i in range(R)]; k += R
    tmp = "%%%d" % k; k += 1
    d, fxc, fxn, fyc, fyn, ox, oy, xcap, ycap = ("%%%d" % (k + i) for i in range(9))
    def feed_read(cur, nxt, off):  # tmp <- the feed's base at `off` (0 .. 127): current block or next
        return ("s_cmp_lt_i32 %s, 64\\n\\ts_cbranch_scc0 3f\\n\\ts_nop 3\\n\\tv_readlane_b32 %s, %s, %s\\n\\ts_branch 4f\\n\\t3:\\n\\t"
                "s_sub_i32 %s, %s, 64\\n\\ts_nop 3\\n\\tv_readlane_b32 %s, %s, %s\\n\\t4:\\n\\t" % (off, tmp, cur, off, tmp, off, tmp, nxt, tmp))
    def body(up):
        # which stream takes the feed: forward up X, forward down Y, backward up Y, backward down X
        x_from_feed = (fwd and up) or (not fwd and not up)
        s = rot(X, R, up) + rot(Y, R, up) + "s_nop 1\\n\\t"
        xr, yr = (X[R - 1], Y[R - 1]) if up else (X[0], Y[0])
        lane = 63 if up else 0
        s += dpp(xr, "shl" if up else "shr", False) + dpp(yr, "shl" if up else "shr", False)
        if x_from_feed:
            s += feed_read(fxc, fxn, ox) + "s_nop 3\\n\\tv_writelane_b32 %s, %s, %d\\n\\tv_writelane_b32 %s, %s, %d\\n\\t" % (xr, tmp, lane, yr, ycap, lane)
        else:
            s += feed_read(fyc, fyn, oy) + "s_nop 3\\n\\tv_writelane_b32 %s, %s, %d\\n\\tv_writelane_b32 %s, %s, %d\\n\\t" % (yr, tmp, lane, xr, xcap, lane)
        return s
    # the numeric labels 3 / 4 appear in both branches: make them unique per branch
    up_b = body(True).replace("3f", "31f").replace("3:", "31:").replace("4f", "41f").replace("4:", "41:")
    dn_b = body(False).replace("3f", "32f").replace("3:", "32:").replace("4f", "42f").replace("4:", "42:")
    ops = ['"+v"(X.b[%d])' % i for i in range(R)] + ['"+v"(Y.b[%d])' % i for i in range(R)] + ['"=&s"(tmp)']
    ins = ['"s"(dir)', '"v"(fx.cur)', '"v"(fx.nxt)', '"v"(fy.cur)', '"v"(fy.nxt)', '"s"(offX)', '"s"(offY)', '"s"(xcap)', '"s"(ycap)']
    out = "__device__ __forceinline__ void rs_rebase_streams_%s(Bases<%d> &X, Bases<%d> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {\n" % ("fwd" if fwd else "bwd", R, R)
    out += "    int tmp;\n"
    out += frame(d, up_b, dn_b)
    out += '                 : %s\n                 : %s\n                 : "scc");\n}\n' % (", ".join(ops), ", ".join(ins))
    return out


def merged(R, fwd):
    """rows + streams of one rebase in ONE statement (R <= 2: 10R + 2R tied registers): one skip test per anti-diagonal instead of two."""
    k, groups, ops = 0, [], []
    for row in ("P", "Q"):
        for st in ("m", "sx", "sy", "lx", "ly"):
            g = []
            for r in range(R):
                g.append("%%%d" % k)
                k += 1
                ops.append('"+v"(%s.c[%d].%s)' % (row, r, st))
            groups.append(g)
    X = ["%%%d" % (k + i) for i in range(R)]; k += R
    Y = ["%%%d" % (k + i) for i in range(R)]; k += R
    ops += ['"+v"(X.b[%d])' % i for i in range(R)] + ['"+v"(Y.b[%d])' % i for i in range(R)] + ['"=&s"(tmp)']
    tmp = "%%%d" % k; k += 1
    d, fxc, fxn, fyc, fyn, ox, oy, xcap, ycap = ("%%%d" % (k + i) for i in range(9))
    def feed_read(cur, nxt, off):
        return ("s_cmp_lt_i32 %s, 64\\n\\ts_cbranch_scc0 3f\\n\\ts_nop 3\\n\\tv_readlane_b32 %s, %s, %s\\n\\ts_branch 4f\\n\\t3:\\n\\t"
                "s_sub_i32 %s, %s, 64\\n\\ts_nop 3\\n\\tv_readlane_b32 %s, %s, %s\\n\\t4:\\n\\t" % (off, tmp, cur, off, tmp, off, tmp, nxt, tmp))
    def body(up):
        x_from_feed = (fwd and up) or (not fwd and not up)
        s = "".join(rot(g, R, up) for g in groups) + rot(X, R, up) + rot(Y, R, up) + "s_nop 1\\n\\t"
        for g in groups:
            s += dpp(g[R - 1] if up else g[0], "shl" if up else "shr", True)
        xr, yr = (X[R - 1], Y[R - 1]) if up else (X[0], Y[0])
        lane = 63 if up else 0
        s += dpp(xr, "shl" if up else "shr", False) + dpp(yr, "shl" if up else "shr", False)
        if x_from_feed:
            s += feed_read(fxc, fxn, ox) + "s_nop 3\\n\\tv_writelane_b32 %s, %s, %d\\n\\tv_writelane_b32 %s, %s, %d\\n\\t" % (xr, tmp, lane, yr, ycap, lane)
        else:
            s += feed_read(fyc, fyn, oy) + "s_nop 3\\n\\tv_writelane_b32 %s, %s, %d\\n\\tv_writelane_b32 %s, %s, %d\\n\\t" % (yr, tmp, lane, xr, xcap, lane)
        return s
    up_b = body(True).replace("3f", "31f").replace("3:", "31:").replace("4f", "41f").replace("4:", "41:")
    dn_b = body(False).replace("3f", "32f").replace("3:", "32:").replace("4f", "42f").replace("4:", "42:")
    ins = ['"s"(dir)', '"v"(fx.cur)', '"v"(fx.nxt)', '"v"(fy.cur)', '"v"(fy.nxt)', '"s"(offX)', '"s"(offY)', '"s"(xcap)', '"s"(ycap)']
    out = "__device__ __forceinline__ void rs_rebase_all_%s(RDiag<%d> &P, RDiag<%d> &Q, Bases<%d> &X, Bases<%d> &Y, const Feed &fx, const Feed &fy, int dir, int offX, int offY, int xcap, int ycap) {\n" % ("fwd" if fwd else "bwd", R, R, R, R)
    out += "    int tmp;\n"
    out += frame(d, up_b, dn_b)
    out += '                 : %s\n                 : %s\n                 : "scc");\n}\n' % (", ".join(ops), ", ".join(ins))
    return out


if __name__ == "__main__":
    print("// ---- generated by tools/gen_rs_rebase.py ----")
    print(rows(1, 2) + rows(2, 2) + rows(4, 1), end="")
    print("__device__ __forceinline__ void rs_rebase_rows(RDiag<4> &P, RDiag<4> &Q, int dir) { rs_rebase_rows(P, dir), rs_rebase_rows(Q, dir); }")
    for R in (1, 2, 4):
        print(streams(R, True) + streams(R, False), end="")
    for R in (1, 2):
        print(merged(R, True) + merged(R, False), end="")
    print("// ---- end of generated code ----")
