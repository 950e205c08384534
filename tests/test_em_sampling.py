"""em.sampleAlignments: the trainer's sampling and chunking options (nanopore/analyses/utils.py:516-517:
maxAlignmentLengthToSample = 50 000 000, maxAlignmentLengthPerJob = 700 000) -- host logic, no GPU."""
import numpy as np

from nanopore_amd import em


def test_a_small_sam_comes_back_whole_as_one_batch():
    o = em.Options()
    parts = em.sampleAlignments([8000] * 50, o)
    assert len(parts) == 1 and list(parts[0]) == list(range(50))
    assert em.sampleAlignments([], o) == []


def test_sample_limit_and_batches():
    rng = np.random.default_rng(1)
    lengths = rng.integers(2000, 20000, 20000)        # a config-3-sized training set: 2.2e8 alignment columns
    o = em.Options()
    o.seed = 11
    parts = em.sampleAlignments(lengths, o)
    taken = np.concatenate(parts)
    assert len(set(taken.tolist())) == len(taken) and lengths[taken].sum() <= o.maxAlignmentLengthToSample
    assert lengths[taken].sum() > o.maxAlignmentLengthToSample - lengths.max()          # filled up to the limit
    assert all(lengths[p].sum() <= o.maxAlignmentLengthPerJob * o.jobsPerBatch for p in parts) and all((np.diff(p) > 0).all() for p in parts)
    assert len(parts) == 2                                                                # 5e7 columns in batches of 64 x 7e5
    again = em.sampleAlignments(lengths, o)
    assert all(np.array_equal(a, b) for a, b in zip(parts, again))                        # seeded
    o.seed = 12
    assert not np.array_equal(np.concatenate(em.sampleAlignments(lengths, o)), taken)
    # an alignment longer than a batch is a batch of its own; one longer than the sample limit is still taken
    o.maxAlignmentLengthPerJob, o.jobsPerBatch, o.maxAlignmentLengthToSample = 1000, 1, 10 ** 9
    parts = em.sampleAlignments([5000, 300, 300, 300, 5000], o)
    assert sorted(len(p) for p in parts) == [1, 1, 3]
    o.maxAlignmentLengthToSample = 10
    assert sum(len(p) for p in em.sampleAlignments([5000, 300], o)) == 1
