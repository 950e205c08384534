"""Host-side stages of libnprealign (band / split planner, MEA chain + cigar, rescore, base encoding) called
through the C ABI WITHOUT a GPU, against the independently written oracle."""
import numpy as np
import pytest

from helpers import cigar_spans, oracle_hmm, orc, random_pair
from nanopore_amd import _lib
from nanopore_amd import realign as R


def _same_plan(a, b):
    return len(a) == len(b) and all(
        all(s[k] == t[k] for k in ("xs", "ys", "xe", "ye", "ragged_start", "ragged_end", "D", "cells"))
        and (s["lo"] == t["lo"]).all() and (s["n"] == t["n"]).all() for s, t in zip(a, b))


@pytest.mark.parametrize("seed", range(4))
def test_planner_equals_oracle(seed):
    rng = np.random.default_rng(400 + seed)
    for it in range(60):
        X, Y, ops = random_pair(rng, int(rng.integers(1, 500)), indel=rng.random() * 0.3, max_indel=int(rng.integers(1, 50)))
        kw = dict(band_mode=it % 2, diagonal_expansion=int(rng.integers(0, 8)) * 2, constraint_trim=int(rng.integers(0, 16)),
                  split_threshold=int(rng.integers(0, 40)), fixed_width=int(rng.integers(2, 300)))
        assert _same_plan(orc.plan(len(X), len(Y), ops, orc.make_params(**kw)),
                          R.plan(R.make_params(**kw), len(X), len(Y), ops)), kw


def test_planner_reference_call_site_parameters():
    """The three parameter sets the reference hard-codes in its call strings (SURVEY.md 8c item vi)."""
    rng = np.random.default_rng(7)
    X, Y, ops = random_pair(rng, 3000, indel=0.1, max_indel=150)
    for split in (3000, 100, 300):  # utils.py:587, alignmentUncertainty.py:41 / marginAlignSnpCaller.py:136, utils.py:511
        kw = dict(band_mode=0, diagonal_expansion=10, constraint_trim=14, split_threshold=split)
        a = orc.plan(len(X), len(Y), ops, orc.make_params(**kw))
        assert _same_plan(a, R.plan(R.make_params(**kw), len(X), len(Y), ops))
    assert len(orc.plan(len(X), len(Y), ops, orc.make_params(split_threshold=100))) >= 1


def test_planner_rejects_non_global_guides():
    P = R.make_params()
    for lX, lY, ops in ((10, 10, [(0, 9)]), (10, 10, [(0, 10), (1, 1)]), (5, 5, [(4, 5)]), (5, 5, [(0, -5)])):
        with pytest.raises(_lib.NprError) as e:
            R.plan(P, lX, lY, ops)
        assert e.value.code == _lib.ERR_INVALID
    # degenerate but legal
    assert R.plan(P, 0, 0, [])[0]["cells"] == 1
    assert R.plan(P, 5, 0, [(2, 5)])[0]["cells"] == 6


@pytest.mark.parametrize("seed", range(5))
def test_mea_and_rescore_equal_oracle(seed):
    rng = np.random.default_rng(500 + seed)
    h = oracle_hmm()
    X, Y, ops = random_pair(rng, int(rng.integers(30, 500)), indel=0.2, max_indel=8)
    seg = orc.plan(len(X), len(Y), ops, orc.make_params(band_mode=orc.BAND_FIXED, fixed_width=80))[0]
    r = orc.fb_f32(h, X, Y, seg["lo"], seg["n"])          # float32 posteriors: what the device hands over
    perm = rng.permutation(len(r["px"]))                   # the C ABI accepts pairs in any order
    for gg, mg in ((0.5, 0.0), (0.0, 0.0), (0.8, 0.2)):
        want, ws = orc.mea_cigar(len(X), len(Y), r["px"], r["py"], r["pp"].astype(np.float64), gg, mg, brute_force=True)
        got, gs = R.mea_cigar(len(X), len(Y), r["px"][perm], r["py"][perm], r["pp"][perm], gg, mg)
        assert got == want and gs == ws
        assert cigar_spans(got) == (len(X), len(Y))
    assert R.rescore(ops, r["px"][perm], r["py"][perm], r["pp"][perm]) == pytest.approx(
        orc.rescore(ops, r["px"], r["py"], r["pp"].astype(np.float64)), abs=1e-15)
    assert R.mea_cigar(4, 3, [], [], [])[0] == [(2, 4), (1, 3)]
    with pytest.raises(_lib.NprError):
        R.mea_cigar(4, 3, [4], [0], [0.5])  # x out of range


def test_encode_bases():
    assert R.encode(b"ACGTacgtNnXU-").tolist() == [0, 1, 2, 3, 0, 1, 2, 3, 4, 4, 4, 4, 4]


def _check_frame_schedule(seg, sch, slots, per_lane):
    """Replays the frame: X-step into odd anti-diagonals, Y-step into even ones, rebases where scheduled."""
    lo, n, D = seg["lo"].astype(np.int64), seg["n"].astype(np.int64), seg["D"]
    jlo, reb, off = sch["jlo"].astype(np.int64), sch["rebase"].astype(np.int64), sch["row_off"].astype(np.int64)
    assert reb[0] == 0 and (np.abs(reb) <= 1).all()
    d = np.arange(D + 1)
    assert (reb[(d % 2 == 0)] <= 0).all() and (reb[(d % 2 == 1)] >= 0).all()      # +1 only before X-steps, -1 before Y
    flo = lo[0] - 2 * jlo[0]                                                         # x-y of slot 0
    expect_off = 0
    for k in range(D + 1):
        if k > 0:
            flo += (1 if k % 2 else -1) + 2 * reb[k]
        assert lo[k] - flo == 2 * jlo[k], (k, lo[k], flo, jlo[k])                   # the control word places the band
        assert 0 <= jlo[k] and jlo[k] + n[k] <= slots                               # ... inside the frame
        assert off[k] == expect_off
        expect_off += per_lane * (-(-(jlo[k] + n[k]) // per_lane) - jlo[k] // per_lane)   # whole lanes per row
    assert sch["cells"] == expect_off and expect_off >= seg["cells"]
    return int(np.abs(reb).sum())


@pytest.mark.parametrize("seed", range(3))
def test_register_kernel_frame_schedule(seed):
    """build_stair_schedule (npr_api_internal.h) through npr_plan_frame_schedule: the band stays inside the frame, rebases
    obey the parity rule the kernels rely on, rows are laid out as the kernels address them."""
    rng = np.random.default_rng(900 + seed)
    followed = rebases = 0
    for it in range(40):
        L = int(rng.integers(40, 1500))
        X, Y, ops = random_pair(rng, L, indel=rng.random() * 0.3, max_indel=int(rng.integers(1, 60)))
        if it % 2:
            kw = dict(band_mode=1, fixed_width=int(rng.integers(4, 250)))
        else:
            kw = dict(band_mode=0, diagonal_expansion=int(rng.integers(1, 8)) * 2, constraint_trim=int(rng.integers(0, 16)),
                      split_threshold=int(rng.integers(20, 3000)))
        P = R.make_params(**kw)
        for s, seg in enumerate(R.plan(P, len(X), len(Y), ops)):
            width = int(seg["n"].max())
            for slots, per_lane in ((64, 1), (128, 2), (256, 4), (512, 2), (1024, 2), (3072, 4)):
                sch = R.frame_schedule(P, len(X), len(Y), ops, slots, per_lane, segment=s)
                if width >= slots:
                    assert sch is None
                    continue
                if sch is None:      # allowed (an edge that jumps further than the slack): the batch then uses a bigger frame
                    continue
                followed += 1
                rebases += _check_frame_schedule(seg, sch, slots, per_lane)
    assert followed > 200 and rebases > 50


def _clamp_walk(lo, n, slots, flo0, chunk):
    """The device planner's form of the walk (npr_plan.hip: k_sched_compose / k_sched_starts): a step is a clamp of flo, the
    x-y of slot 0 -- an X-step takes it to max(flo + 1, c), a Y-step to min(flo - 1, c) -- and `chunk` steps compose to
    min(max(flo + A, L), U).  Returns flo after every step (step by step) and flo at the head of every chunk (composed)."""
    D, span, BIG = len(lo) - 1, 2 * (slots - 1), 1 << 29
    hi = lo + 2 * (n - 1)
    flo, step_by_step, heads = flo0, [flo0], []
    for d0 in range(0, D + 1, chunk):
        A, L, U = 0, -BIG, BIG
        for d in range(max(d0, 1), min(d0 + chunk, D + 1)):
            if d % 2:
                c = max(hi[d] - span, hi[d + 1] - span + 1 if d < D else -BIG)
                A, L, U = A + 1, max(L + 1, c), max(U + 1, c)
                flo = max(flo + 1, c)
            else:
                c = min(lo[d], lo[d + 1] - 1 if d < D else BIG)
                A, L, U = A - 1, min(L - 1, c), min(U - 1, c)
                flo = min(flo - 1, c)
            step_by_step.append(flo)
        heads.append((A, L, U))
    composed, f = [], flo0
    for A, L, U in heads:
        composed.append(f)
        f = min(max(f + A, L), U)
    return np.array(step_by_step), np.array(composed)


@pytest.mark.parametrize("seed", range(3))
def test_frame_schedule_is_a_clamp_that_composes_in_chunks(seed):
    """The device planner does not walk a schedule step after step: it composes chunks of steps as clamps of the frame's
    position and walks the chunks side by side (npr_plan.hip).  Wherever the host planner's sequential walk
    (build_stair_schedule, through npr_plan_frame_schedule) can follow a band, both forms give the same frame position on every
    anti-diagonal, for every chunk length."""
    rng = np.random.default_rng(1700 + seed)
    followed = rebased = 0
    for it in range(30):
        Lr = int(rng.integers(40, 2500))
        X, Y, ops = random_pair(rng, Lr, indel=rng.random() * 0.3, max_indel=int(rng.integers(1, 80)))
        if it % 2:
            kw = dict(band_mode=1, fixed_width=int(rng.integers(4, 250)))
        else:
            kw = dict(band_mode=0, diagonal_expansion=int(rng.integers(1, 8)) * 2, constraint_trim=int(rng.integers(0, 16)),
                      split_threshold=int(rng.integers(20, 3000)))
        P = R.make_params(**kw)
        for s, seg in enumerate(R.plan(P, len(X), len(Y), ops)):
            lo, n = seg["lo"].astype(np.int64), seg["n"].astype(np.int64)
            for slots, per_lane in ((64, 1), (128, 2), (256, 4), (1024, 2)):
                sch = R.frame_schedule(P, len(X), len(Y), ops, slots, per_lane, segment=s)
                if sch is None:
                    continue
                want = lo - 2 * sch["jlo"].astype(np.int64)
                for chunk in (1, 7, 256):
                    steps, heads = _clamp_walk(lo, n, slots, int(want[0]), chunk)
                    assert np.array_equal(steps, want), (it, s, slots, chunk)
                    head_rows = np.maximum(np.arange(0, len(want), chunk) - 1, 0)     # a chunk's head: before its first step
                    assert np.array_equal(heads, want[head_rows]), (it, s, slots, chunk)
                followed += 1
                rebased += int(np.abs(sch["rebase"]).sum() > 0)
    assert followed > 60 and rebased > 10


def test_frame_schedule_long_gap_rebases_every_other_step():
    """A 600-base deletion inside a W = 100 band: the frame drifts one slot per two anti-diagonals."""
    ops = [(0, 300), (2, 600), (0, 300)]
    P = R.make_params(band_mode=1, fixed_width=100)
    seg = R.plan(P, 1200, 600, ops)[0]
    sch = R.frame_schedule(P, 1200, 600, ops, 64, 1)
    assert sch is not None and _check_frame_schedule(seg, sch, 64, 1) >= 250
    d = np.arange(seg["D"] + 1)
    inside = (d > 700) & (d < 1100)                                  # anti-diagonals crossing the deletion
    assert (sch["rebase"][inside & (d % 2 == 1)] == 1).mean() > 0.9


def test_fixed_band_narrower_than_two_cells_is_rejected():
    """A fixed band of width 0 or 1 leaves every odd anti-diagonal empty (a disconnected lattice): the plan refuses it
    instead of letting the read fail on the device with NPR_ERR_ZERO_PROB."""
    from nanopore_amd import _lib, realign as R
    for w in (0, 1):
        with pytest.raises(_lib.NprError) as e:
            R.plan(R.make_params(band_mode=R.BAND_FIXED, fixed_width=w), 10, 10, [(0, 10)])
        assert e.value.code == _lib.ERR_INVALID
    assert len(R.plan(R.make_params(band_mode=R.BAND_FIXED, fixed_width=2), 10, 10, [(0, 10)])) == 1


def test_sam_records_are_formatted_natively_like_the_python_writer():
    """npr_format_sam_records against a record-by-record rendering of the same fields (the eleven mandatory SAM columns
    as realignSamFile3TargetFn's writer prints them, nanopore/analyses/utils.py:591-609): empty cigar, shared reference
    names, default and explicit FLAG / MAPQ, error codes."""
    from nanopore_amd import realign as R
    rng = np.random.default_rng(5)
    n = 300
    ref_names = [b"chr_%d" % k for k in range(7)]
    ref_index = rng.integers(0, 7, n).astype(np.int32)
    pos = rng.integers(1, 10 ** 7, n)
    pos[:3] = (1, 9, 10)
    lens = rng.integers(0, 400, n)
    lens[5] = 0
    seq_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    seq = rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), int(seq_off[-1]))
    nops = rng.integers(0, 30, n)
    nops[7] = 0
    word_off = np.concatenate([[0], np.cumsum(nops)])[:-1].astype(np.int64) + 3  # the lists may start anywhere
    words = ((rng.integers(1, 5000, int(nops.sum()) + 3) << 2) | rng.integers(0, 3, int(nops.sum()) + 3)).astype(np.uint32)
    qnames = [b"read/%d_%s" % (i, b"x" * int(rng.integers(0, 9))) for i in range(n)]
    flag = rng.choice([0, 16, 256, 2048], n).astype(np.int32)
    mapq = rng.integers(0, 256, n).astype(np.int32)

    def render(i, f, q):
        cig = b"".join(b"%d%s" % (int(wd) >> 2, b"MID"[int(wd) & 3:(int(wd) & 3) + 1]) for wd in words[word_off[i]:word_off[i] + nops[i]]) or b"*"
        return b"\t".join((qnames[i], b"%d" % f, ref_names[ref_index[i]], b"%d" % pos[i], b"%d" % q, cig, b"*", b"0", b"0",
                           seq[seq_off[i]:seq_off[i + 1]].tobytes() or b"*", b"*")) + b"\n"   # an empty SEQ is "*" in SAM

    buf, off = R.format_sam_records(qnames, ref_names, ref_index, pos, word_off, nops, words, seq, seq_off)
    assert buf.tobytes() == b"".join(render(i, 0, 255) for i in range(n))
    assert all(buf[off[i]:off[i + 1]].tobytes() == render(i, 0, 255) for i in (0, 5, 7, n - 1))
    buf2, _ = R.format_sam_records(qnames, ref_names, ref_index, pos, word_off, nops, words, seq, seq_off, flag=flag, mapq=mapq)
    assert buf2.tobytes() == b"".join(render(i, int(flag[i]), int(mapq[i])) for i in range(n))
    empty, eoff = R.format_sam_records([], ref_names, [], [], [], [], words, seq, [0])
    assert len(empty) == 0 and list(eoff) == [0]
    bad = words.copy()
    bad[int(word_off[20])] |= 3
    if nops[20]:
        with pytest.raises(R.NprError):
            R.format_sam_records(qnames, ref_names, ref_index, pos, word_off, nops, bad, seq, seq_off)
    with pytest.raises(R.NprError):
        R.format_sam_records(qnames, ref_names, ref_index, -pos, word_off, nops, words, seq, seq_off)
