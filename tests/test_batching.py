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
