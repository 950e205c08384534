"""Seeded synthetic workloads for the realigner.

A read is a reference slice passed through the error channel of a five-state pair-HMM: the chain is walked
with the model's transition matrix and bases are emitted from its emission tables, so the workload has the
indel / substitution statistics the shipped model (nanopore/mappers/blasr_hmm_0.txt) was trained on.  The
guide alignment handed to the realigner is the true generating path with every indel block slid by up to
+-`jitter` columns along the neighbouring match runs, so the realigner has real work to do.

Everything is vectorised over reads (numpy); no Python loop per base.
"""
import numpy as np

OP_M, OP_I, OP_D = 0, 1, 2
_ASCII = np.frombuffer(b"ACGT", dtype=np.uint8)


def random_reference(rng, length, gc=0.5):
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return rng.choice(4, size=length, p=p).astype(np.uint8)


def _sample_rows(rng, cdf_rows, idx):
    """Draw one category per element of idx from the row cdf_rows[idx]."""
    u = rng.random(len(idx))
    return (u[:, None] >= cdf_rows[idx]).sum(axis=1).astype(np.int64)


def error_channel(rng, ref_codes, ref_off, T, E):
    """Walk the 5-state chain over each reference slice.

    ref_codes: uint8 codes (0..3) of all slices concatenated; ref_off: int64 offsets [n+1].
    Returns (read_codes, read_off, ops_flat, ops_off): ops are per-column codes (0 M, 1 I, 2 D) of the TRUE
    global alignment, concatenated per read.
    """
    T = np.asarray(T, dtype=np.float64).reshape(5, 5)
    E = np.asarray(E, dtype=np.float64).reshape(5, 4, 4)
    Tc = np.cumsum(T / T.sum(axis=1, keepdims=True), axis=1)[:, :4]
    match_given_x = E[0] / E[0].sum(axis=1, keepdims=True)
    Mc = np.cumsum(match_given_x, axis=1)[:, :3]
    ins = E[2].sum(axis=0)
    Ic = np.cumsum(ins / ins.sum())[:3][None, :]
    n = len(ref_off) - 1
    lens = (ref_off[1:] - ref_off[:-1]).astype(np.int64)
    cap = int(lens.max() * 1.6) + 64
    reads = np.zeros((n, cap), dtype=np.uint8)
    ops = np.zeros((n, 2 * cap), dtype=np.uint8)
    rlen = np.zeros(n, dtype=np.int64)
    nops = np.zeros(n, dtype=np.int64)
    xpos = np.zeros(n, dtype=np.int64)
    state = np.zeros(n, dtype=np.int64)
    active = np.nonzero(lens > 0)[0]
    while len(active):
        s = _sample_rows(rng, Tc, state[active])
        state[active] = s
        is_m, is_x, is_y = s == 0, (s == 1) | (s == 3), (s == 2) | (s == 4)
        # reads that would overflow their buffers are forced to consume reference
        full = (rlen[active] >= cap - 1) | (nops[active] >= 2 * cap - 1)
        is_y &= ~full
        is_x |= full & ~is_m
        a_m, a_x, a_y = active[is_m], active[is_x], active[is_y]
        if len(a_m):
            xb = ref_codes[ref_off[a_m] + xpos[a_m]]
            yb = _sample_rows(rng, Mc, xb)
            reads[a_m, rlen[a_m]] = yb
            rlen[a_m] += 1
            ops[a_m, nops[a_m]] = OP_M
            xpos[a_m] += 1
        if len(a_x):
            ops[a_x, nops[a_x]] = OP_D
            xpos[a_x] += 1
        if len(a_y):
            yb = _sample_rows(rng, Ic, np.zeros(len(a_y), dtype=np.int64))
            reads[a_y, rlen[a_y]] = yb
            rlen[a_y] += 1
            ops[a_y, nops[a_y]] = OP_I
        nops[active] += 1
        active = active[xpos[active] < lens[active]]
    read_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(rlen, out=read_off[1:])
    ops_off = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(nops, out=ops_off[1:])
    mask_r = np.arange(cap)[None, :] < rlen[:, None]
    mask_o = np.arange(2 * cap)[None, :] < nops[:, None]
    return reads[mask_r], read_off, ops[mask_o], ops_off


def columns_to_runs(ops_flat, ops_off):
    """Per-column op codes -> run-length (op,len) pairs, CSR by read (runs never cross a read boundary)."""
    total = len(ops_flat)
    n = len(ops_off) - 1
    if total == 0:
        return np.zeros((0, 2), dtype=np.int32), np.zeros(n + 1, dtype=np.int64)
    brk = np.ones(total, dtype=bool)
    brk[1:] = ops_flat[1:] != ops_flat[:-1]
    starts_of_reads = ops_off[:-1][ops_off[:-1] < total]
    brk[starts_of_reads] = True
    starts = np.nonzero(brk)[0]
    lengths = np.diff(np.append(starts, total))
    runs = np.stack([ops_flat[starts].astype(np.int32), lengths.astype(np.int32)], axis=1)
    run_off = np.searchsorted(starts, ops_off, side="left").astype(np.int64)
    return runs, run_off


def jitter_guide(rng, runs, run_off, jitter=20):
    """Slide every indel block by a random number of columns in [-jitter, jitter] along the flanking match
    runs (never emptying a match run, never across a read boundary).  Spans are preserved, so the result is
    still a valid global alignment of the same two sequences."""
    runs = runs.copy()
    nruns = len(runs)
    if nruns == 0 or jitter <= 0:
        return runs
    is_m = runs[:, 0] == OP_M
    first = np.zeros(nruns, dtype=bool)
    first[run_off[:-1][run_off[:-1] < nruns]] = True
    # an indel block starts at a non-M run that follows an M run of the same read
    idx = np.arange(nruns)
    prev_is_m = np.zeros(nruns, dtype=bool)
    prev_is_m[1:] = is_m[:-1]
    block_start = (~is_m) & prev_is_m & (~first)
    # end of the block: next M run index
    next_m = np.full(nruns, -1, dtype=np.int64)
    last = -1
    m_idx = np.nonzero(is_m)[0]
    pos = np.searchsorted(m_idx, idx, side="left")
    ok = pos < len(m_idx)
    next_m[ok] = m_idx[pos[ok]]
    # read id of each run, to forbid crossing
    read_of = np.searchsorted(run_off, idx, side="right") - 1
    bs = np.nonzero(block_start)[0]
    left = bs - 1
    right = next_m[bs]
    valid = (right >= 0)
    valid[valid] &= read_of[right[valid]] == read_of[bs[valid]]
    bs, left, right = bs[valid], left[valid], right[valid]
    # each match run may give at most half of its spare columns to either side
    spare_l = (runs[left, 1] - 1) // 2
    spare_r = (runs[right, 1] - 1) // 2
    k = rng.integers(-jitter, jitter + 1, size=len(bs))
    k = np.clip(k, -np.minimum(spare_l, jitter), np.minimum(spare_r, jitter))
    # k > 0 moves the block right: the left run grows, the right run shrinks
    np.add.at(runs[:, 1], left, k.astype(np.int32))
    np.add.at(runs[:, 1], right, (-k).astype(np.int32))
    assert (runs[:, 1] > 0).all()
    return runs


def read_to_ref_ratio(T):
    """Expected read bases per reference base of the channel: from the chain's stationary distribution."""
    T = np.asarray(T, dtype=np.float64).reshape(5, 5)
    P = T / T.sum(axis=1, keepdims=True)
    w, v = np.linalg.eig(P.T)
    pi = np.real(v[:, np.argmin(np.abs(w - 1.0))])
    pi = pi / pi.sum()
    return float((pi[0] + pi[2] + pi[4]) / (pi[0] + pi[1] + pi[3]))


def pilot_ratio(seed, read_len, T, E, n=256):
    """Read/reference length ratio of the channel at THIS read length, from a seeded pilot simulation (the
    stationary ratio over-corrects short reads, which rarely meet a long deletion)."""
    rng = np.random.default_rng([seed, 7])
    stat = read_to_ref_ratio(T)
    ilen = int(round(read_len / stat ** 0.5))
    off = np.arange(n + 1, dtype=np.int64) * ilen
    codes = rng.integers(0, 4, size=n * ilen).astype(np.uint8)
    _, roff, _, _ = error_channel(rng, codes, off, T, E)
    return float(roff[-1]) / float(n * ilen)


def make_workload(seed, n_reads, read_len, T, E, flank=0, ref_slice_len=None, genome=None, length_sigma=0.0,
                  len_min=None, len_max=None, jitter=20, uniform_len=None, windowed=False):
    """Builds one synthetic batch.

    read_len      target read length (the reference interval the read is drawn from has this length)
    flank         reference bases kept on each side of the true interval (C3: 2*W)
    ref_slice_len if given, the slice is exactly this long with the true interval placed uniformly inside
                  (north-star shape: 10 kb reads x 50 kb slice)
    genome        optional uint8 code array to cut intervals from (C3: 4.6 Mb synthetic E. coli stand-in);
                  otherwise every slice is fresh uniform-random sequence
    length_sigma  >0: interval lengths ~ lognormal(mean=read_len, sigma) clipped to [len_min, len_max]
    uniform_len   (lo, hi): interval lengths uniform in [lo, hi] (C5)
    windowed      False: the guide is made global over the slice by leading / trailing deletions;
                  True: the guide keeps its own span and comes with `guide_start` = (its first reference position
                  in the slice, 0), like the coordinates of the exonerate cigar the reference hands to cactus_realign
    Returns dict with ASCII buffers + CSR offsets ready for Context.stage_csr, plus the true alignment.
    """
    rng = np.random.default_rng(seed)
    # reference interval lengths are the read-length targets divided by the channel's read/ref ratio, so
    # that the READS have the named length on average (the shipped model deletes more than it inserts)
    ratio = pilot_ratio(seed, read_len, T, E)
    read_len = int(round(read_len / ratio))
    if uniform_len is not None:
        uniform_len = (int(uniform_len[0] / ratio), int(uniform_len[1] / ratio))
        ilen = rng.integers(uniform_len[0], uniform_len[1] + 1, size=n_reads)
    elif length_sigma > 0:
        mu = np.log(read_len) - 0.5 * length_sigma ** 2
        ilen = np.exp(rng.normal(mu, length_sigma, size=n_reads)).astype(np.int64)
        ilen = np.clip(ilen, int((len_min or 1) / ratio), int((len_max or (1 << 30)) / ratio))
    else:
        ilen = np.full(n_reads, read_len, dtype=np.int64)
    if ref_slice_len is not None:
        lead = rng.integers(0, np.maximum(ref_slice_len - ilen, 0) + 1)
        slen = np.full(n_reads, ref_slice_len, dtype=np.int64)
        slen = np.maximum(slen, ilen)
    else:
        lead = np.full(n_reads, flank, dtype=np.int64)
        slen = ilen + 2 * flank
    ref_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(slen, out=ref_off[1:])
    start = None
    if genome is None:
        ref_codes = rng.integers(0, 4, size=int(ref_off[-1])).astype(np.uint8)
    else:
        start = rng.integers(0, len(genome) - slen.max(), size=n_reads)
        ref_codes = np.empty(int(ref_off[-1]), dtype=np.uint8)
        for i in range(n_reads):
            ref_codes[ref_off[i]:ref_off[i + 1]] = genome[start[i]:start[i] + slen[i]]
    # the read is generated from the true interval only
    iv_off = np.zeros(n_reads + 1, dtype=np.int64)
    np.cumsum(ilen, out=iv_off[1:])
    sel = np.repeat(ref_off[:-1] + lead - iv_off[:-1], ilen) + np.arange(int(iv_off[-1]))
    iv_codes = ref_codes[sel]
    read_codes, read_off, cols, cols_off = error_channel(rng, iv_codes, iv_off, T, E)
    runs, run_off = columns_to_runs(cols, cols_off)
    true_runs = runs.copy()
    runs = jitter_guide(rng, runs, run_off, jitter)
    # wrap with the leading / trailing deletions that make the guide global over the slice
    trail = slen - lead - ilen
    out_runs = []
    out_off = np.zeros(n_reads + 1, dtype=np.int64)
    true_out = []
    for i in range(n_reads):
        r = runs[run_off[i]:run_off[i + 1]]
        tr = true_runs[run_off[i]:run_off[i + 1]]
        parts, tparts = [], []
        if lead[i] > 0 and not windowed:
            parts.append(np.array([[OP_D, lead[i]]], dtype=np.int32))
        parts.append(r)
        if trail[i] > 0 and not windowed:
            parts.append(np.array([[OP_D, trail[i]]], dtype=np.int32))
        g = np.concatenate(parts) if parts else np.zeros((0, 2), dtype=np.int32)
        # merge adjacent equal ops created by the wrapping
        if len(g) > 1:
            keep = np.ones(len(g), dtype=bool)
            keep[1:] = g[1:, 0] != g[:-1, 0]
            grp = np.cumsum(keep) - 1
            merged = np.zeros((grp[-1] + 1, 2), dtype=np.int32)
            merged[:, 0] = g[keep, 0]
            np.add.at(merged[:, 1], grp, g[:, 1])
            g = merged
        out_runs.append(g)
        out_off[i + 1] = out_off[i] + len(g)
    guide_ops = np.concatenate(out_runs) if out_runs else np.zeros((0, 2), dtype=np.int32)
    guide_start = None
    if windowed:
        guide_start = np.stack([np.asarray(lead, dtype=np.int64), np.zeros(n_reads, dtype=np.int64)], axis=1)
    return dict(ref=_ASCII[ref_codes], ref_off=ref_off, read=_ASCII[read_codes], read_off=read_off,
                guide_ops=guide_ops, guide_off=out_off, lead=lead, interval_len=ilen,
                true_runs=true_runs, true_off=run_off, guide_start=guide_start, genome_start=start)


# ---- the named configurations of BASELINE.json / BASELINE.md ----

def config_c2(T, E, n_reads=1000):
    """Synthetic 1 k reads x 1 kb, band=100 (BASELINE.json configs[1]); seed 1001."""
    return make_workload(1001, n_reads, 1000, T, E, flank=0), 100


def config_north_star(T, E, n_reads=4096, seed=1003, windowed=True):
    """10 kb reads x 50 kb reference slice, band 200 (BASELINE.json north_star target shape).  The guide carries the
    coordinates of the read's interval inside the slice (`guide_start`), as the exonerate cigar the reference pipes into
    cactus_realign does (nanopore/analyses/utils.py:173-186,587); windowed=False instead spells the flanks out as
    leading / trailing deletions of a guide that is global over the 50 kb."""
    return make_workload(seed, n_reads, 10000, T, E, ref_slice_len=50000, windowed=windowed), 200


def config_c3(T, E, n_reads=50000, genome_len=4641652, gc=0.508):
    """E. coli-sized synthetic reference (length / GC of K-12 MG1655) x ~8 kb lognormal reads, band=200,
    slice = true interval +- 2W (BASELINE.json configs[2]); seed 1002."""
    rng = np.random.default_rng(1002)
    genome = random_reference(rng, genome_len, gc)
    return make_workload(1002, n_reads, 8000, T, E, flank=400, genome=genome, length_sigma=0.3, len_min=2000,
                         len_max=20000), 200


def config_c3_shared(T, E, n_reads=50000, genome_len=4641652, gc=0.508):
    """configs[2] the way the reference holds it: ONE 4.6 Mb contig (the reference FASTA, argv[1] of cactus_realign) shared
    by all reads through `ref_index`, every guide carrying the coordinates of its window on the contig -- the exonerate
    cigar line of nanopore/analyses/utils.py:173-186.  Same reads, same windows, same bands as config_c3."""
    rng = np.random.default_rng(1002)
    genome = random_reference(rng, genome_len, gc)
    w = make_workload(1002, n_reads, 8000, T, E, flank=400, genome=genome, length_sigma=0.3, len_min=2000,
                      len_max=20000, windowed=True)
    return shared_contig(w, genome), 200


def shared_contig(w, genome):
    """A workload cut from `genome` with windowed guides, restated against the genome itself: one reference sequence,
    ref_index = 0 for every read, guide_start = the window's first position on the contig."""
    n = len(w["read_off"]) - 1
    out = dict(w)
    out["ref"] = _ASCII[genome]
    out["ref_off"] = np.array([0, len(genome)], dtype=np.int64)
    out["ref_index"] = np.zeros(n, dtype=np.int32)
    gs = np.zeros((n, 2), dtype=np.int64)
    gs[:, 0] = w["genome_start"] + w["lead"]
    out["guide_start"] = gs
    return out


def csr_take(buf, off, idx):
    """Rows `idx` of a CSR (buf, off): (new buf, new off).  buf may have trailing dimensions."""
    off = np.asarray(off)
    k = off[np.asarray(idx) + 1] - off[idx]
    new_off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(k, out=new_off[1:])
    src = np.repeat(off[idx] - new_off[:-1], k) + np.arange(int(new_off[-1]))
    return buf[src], new_off


def take_reads(w, idx):
    """The sub-workload of the reads `idx` (a rank's shard): reads, guides and per-read fields; a shared reference
    (ref_index present) stays whole, per-read slices are cut out."""
    idx = np.asarray(idx, dtype=np.int64)
    out = {}
    out["read"], out["read_off"] = csr_take(w["read"], w["read_off"], idx)
    out["guide_ops"], out["guide_off"] = csr_take(w["guide_ops"], w["guide_off"], idx)
    if w.get("ref_index") is not None:
        out["ref"], out["ref_off"], out["ref_index"] = w["ref"], w["ref_off"], np.ascontiguousarray(w["ref_index"][idx])
    else:
        out["ref"], out["ref_off"] = csr_take(w["ref"], w["ref_off"], idx)
        out["ref_index"] = None
    gs = w.get("guide_start")
    out["guide_start"] = None if gs is None else np.ascontiguousarray(gs[idx])
    for k in ("lead", "interval_len", "genome_start"):
        if w.get(k) is not None:
            out[k] = w[k][idx]
    return out


def config_c5(T, E, n_reads_per_type=10000):
    """Mixed 10-50 kb reads, per-read-type HMMs, band=200 (BASELINE.json configs[4]); seed 1005.
    Returns (workload, W, model_slot): slot 0 = 2D (hmm_0), 1 = template (hmm_20), 2 = complement (hmm_40)."""
    w = make_workload(1005, 3 * n_reads_per_type, 30000, T, E, flank=400, uniform_len=(10000, 50000))
    slot = np.repeat(np.arange(3, dtype=np.int32), n_reads_per_type)
    return w, 200, slot


# ---- a workload as the files the pipeline's entry points take ----

def write_fasta(path, names, ref, ref_off, width=100):
    """FASTA with `width` bases per line (what fastaWrite emits), vectorised: a 4.6 Mb contig is one reshape."""
    with open(path, "wb") as fh:
        for k, name in enumerate(names):
            seq = np.ascontiguousarray(ref[ref_off[k]:ref_off[k + 1]], dtype=np.uint8)
            fh.write(b">" + name.encode() + b"\n")
            full = len(seq) // width * width
            if full:
                body = np.empty((full // width, width + 1), dtype=np.uint8)
                body[:, :width] = seq[:full].reshape(-1, width)
                body[:, width] = 10
                fh.write(body.tobytes())
            if len(seq) > full:
                fh.write(seq[full:].tobytes() + b"\n")


def write_workload_files(w, sam_path, fasta_path, fastq_path=None, ref_names=None, read_names=None):
    """A workload dict as SAM + FASTA (+ FASTQ): one record per read with FLAG 0, POS = where the guide starts on its
    reference (1-based), CIGAR = the guide, SEQ = the read -- a mapper's (or chainSamFile's) output for these reads.  The
    records are formatted natively (npr_format_sam_records).  Returns (ref_names, read_names)."""
    from . import realign
    n = len(w["read_off"]) - 1
    n_refs = len(w["ref_off"]) - 1
    ref_names = ref_names or ["ref_%d" % k for k in range(n_refs)]
    read_names = read_names or ["read_%d" % i for i in range(n)]
    write_fasta(fasta_path, ref_names, w["ref"], w["ref_off"])
    g = np.asarray(w["guide_ops"], dtype=np.int64).reshape(-1, 2)
    words = ((g[:, 1] << 2) | g[:, 0]).astype(np.uint32)
    goff = np.asarray(w["guide_off"], dtype=np.int64)
    ri = w.get("ref_index")
    ref_index = np.asarray(ri, dtype=np.int32) if ri is not None else np.arange(n, dtype=np.int32)
    gs = w.get("guide_start")
    pos = (np.asarray(gs, dtype=np.int64)[:, 0] if gs is not None else np.zeros(n, dtype=np.int64)) + 1
    buf, _ = realign.format_sam_records([s.encode() for s in read_names], [s.encode() for s in ref_names], ref_index, pos, goff[:-1],
                                        goff[1:] - goff[:-1], words, np.ascontiguousarray(w["read"], dtype=np.uint8),
                                        np.asarray(w["read_off"], dtype=np.int64))
    with open(sam_path, "wb") as fh:
        fh.write(b"@HD\tVN:1.0\tSO:unsorted\n")
        for k in range(n_refs):
            fh.write(("@SQ\tSN:%s\tLN:%d\n" % (ref_names[k], int(w["ref_off"][k + 1] - w["ref_off"][k]))).encode())
        fh.write(memoryview(buf))
    if fastq_path is not None:
        ro = np.asarray(w["read_off"], dtype=np.int64)
        with open(fastq_path, "wb") as fh:
            for i in range(n):
                s = np.ascontiguousarray(w["read"][ro[i]:ro[i + 1]], dtype=np.uint8).tobytes()
                fh.write(b"@" + read_names[i].encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
    return ref_names, read_names
