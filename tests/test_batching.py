"""CPU tests of the host-side length bucketing (speechless_amd/batching.py; a new feature for BASELINE config 5)."""
import numpy as np

from speechless_amd.batching import bucket_batches, padding_waste, steps_for_ranks


class Utt:
    def __init__(self, n):
        self.n = n

    def z_normalized_transposed_spectrogram(self):
        return np.zeros((self.n, 1))


def test_bucketing_keeps_every_example_once_and_cuts_padding():
    rng = np.random.RandomState(3)
    utts = [Utt(int(n)) for n in rng.randint(2000, 8001, size=203)]
    random_batches = [utts[i:i + 8] for i in range(0, len(utts), 8)]
    bucketed = bucket_batches(utts, 8, seed=1)
    flat = [u for b in bucketed for u in b]
    assert sorted(id(u) for u in flat) == sorted(id(u) for u in utts)
    assert all(1 <= len(b) <= 8 for b in bucketed)
    assert padding_waste(bucketed) < 0.06 < padding_waste(random_batches)


def test_drop_last_and_empty_input():
    assert bucket_batches([], 4) == []
    utts = [Utt(n) for n in (10, 11, 12, 13, 50)]
    batches = bucket_batches(utts, 4, drop_last=True, shuffle=False)
    assert all(len(b) == 4 for b in batches) and len(batches) == 1


def test_rank_balancing_is_even():
    rng = np.random.RandomState(4)
    utts = [Utt(int(n)) for n in rng.randint(2000, 8001, size=1024)]
    steps = steps_for_ranks(utts, 8, 8, seed=2)
    assert steps and all(len(step) == 8 and all(len(b) == 8 for b in step) for step in steps)
    seen = [id(u) for step in steps for b in step for u in b]
    assert len(seen) == len(set(seen))

    def cost(b):
        return max(u.n for u in b) * len(b)
    # a data-parallel step is as slow as its most expensive rank: within every step the costs must be close
    ratios = sorted(max(cost(b) for b in step) / min(cost(b) for b in step) for step in steps)
    # steps cut from one bucket are even to ~1 %; the steps assembled from bucket left-overs span neighbouring buckets
    assert ratios[len(ratios) // 2] < 1.03 and ratios[-2] < 1.12 and ratios[-1] < 1.8
    assert padding_waste([b for step in steps for b in step]) < 0.10


def test_config5_steps_come_from_one_global_corpus_and_pad_alike_on_every_rank():
    """bench.py --config 5 --gpus N forms its steps with steps_for_ranks from ONE global corpus (seed 3):
    the ranks of a step hold disjoint utterances of one length bucket and zero-pad to the SAME length (the step's longest
    utterance), so every rank launches the same grids; an utterance's data depends on its global id only."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    for world in (2, 8):
        steps, lengths, pads = bench.long_form_steps(world)
        assert len(lengths) == 64 * world and len(steps) == 8
        seen = [i for step in steps for b in step for i in b]
        assert sorted(seen) == list(range(64 * world))
        for step, t_pad in zip(steps, pads):
            assert len(step) == world and all(len(b) == 8 for b in step)
            assert t_pad == max(lengths[i] for b in step for i in b)
            # dealt round-robin in length order: the ranks' own maxima are neighbours in the sorted step
            tops = sorted(max(lengths[i] for i in b) for b in step)
            assert tops[-1] == t_pad
    # two ranks of a two-rank run: same padded shapes step by step, different utterances, same generator per utterance
    r0, waste0 = bench.long_form_batches(0, 2)
    r1, waste1 = bench.long_form_batches(1, 2)
    assert waste0 == waste1 and 0 < waste0 < 0.12
    for (x0, l0, ll0, p0, t0), (x1, l1, ll1, p1, t1) in zip(r0, r1):
        assert x0.shape == x1.shape and x0.shape[0] == 8
        assert -(-x0.shape[1] // 2) == -(-x1.shape[1] // 2)          # same output frames -> same padded tile count
        assert not np.array_equal(t0, t1) or not np.array_equal(x0, x1)
        assert (p0 == t0 // 2).all() and (ll0 <= np.minimum(200, t0 // 4)).all() and (ll0 >= 20).all()
        assert (l0[np.arange(8), ll0 - 1] >= 0).all() and l0.max() < 28
