"""CPU tests of the host-side length bucketing (speechless_amd/batching.py; a new feature for BASELINE config 5)."""
import numpy as np

from speechless_amd.batching import balance_across_ranks, bucket_batches, padding_waste


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
    utts = [Utt(int(n)) for n in rng.randint(2000, 8001, size=512)]
    batches = bucket_batches(utts, 8, seed=2)
    per_rank = balance_across_ranks(batches, 8)
    assert len(per_rank) == 8 and len({len(r) for r in per_rank}) == 1
    def cost(b):
        return max(u.n for u in b) * len(b)
    frames = [sum(cost(b) for b in r) for r in per_rank]
    assert max(frames) / min(frames) < 1.02
    # a data-parallel step is as slow as its most expensive rank: within every step the costs must be close
    ratios = []
    for step in range(len(per_rank[0])):
        costs = [cost(per_rank[r][step]) for r in range(8)]
        ratios.append(max(costs) / min(costs))
    ratios.sort()
    assert ratios[len(ratios) // 2] < 1.1 and ratios[-1] < 1.6  # only the step holding the partial batches is uneven
