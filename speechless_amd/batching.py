"""Host-side batch formation for the long-form regime (BASELINE config 5: 257-bin x up-to-8000-frame utterances).

The reference samples batches at random and zero-pads every batch to its longest member
(speechless/corpus.py:224-226, speechless/net.py:578-587), so the padded frames are convolved like real ones; with
lengths U{2000..8000} that wastes ~40 % of the FLOPs.  Length bucketing is a NEW host feature of this implementation
("long-context"): utterances are sorted into buckets of similar length, batches are cut from one
bucket at a time, and in data-parallel runs the batches of a step are balanced across ranks by total frame count.
Semantics inside a batch are unchanged (still zero-padded to the batch maximum, exactly like the reference), so parity
per batch is unaffected; only the composition of batches differs from the reference's random sampling.
"""
import random


def bucket_batches(examples, batch_size, length_of=None, bucket_width=None, shuffle=True, seed=0, drop_last=False):
    """Groups `examples` into batches of similar length.

    length_of: example -> frame count (default: rows of example.z_normalized_transposed_spectrogram()).
    bucket_width: frames per bucket (default: spread / 16).  Returns a list of lists (each <= batch_size long)."""
    if length_of is None:
        def length_of(e):
            return e.z_normalized_transposed_spectrogram().shape[0]
    items = [(length_of(e), i, e) for i, e in enumerate(examples)]
    if not items:
        return []
    lo = min(n for n, _, _ in items)
    hi = max(n for n, _, _ in items)
    if bucket_width is None:
        bucket_width = max(1, (hi - lo + 16) // 16)
    rng = random.Random(seed)
    buckets = {}
    for n, i, e in items:
        buckets.setdefault((n - lo) // bucket_width, []).append((n, i, e))
    batches = []
    leftovers = []
    for key in sorted(buckets):
        members = buckets[key]
        if shuffle:
            rng.shuffle(members)
        while len(members) >= batch_size:
            batches.append([e for _, _, e in members[:batch_size]])
            members = members[batch_size:]
        leftovers.extend(members)  # sorted by bucket: neighbours in length
    while leftovers and not (drop_last and len(leftovers) < batch_size):
        batches.append([e for _, _, e in leftovers[:batch_size]])
        leftovers = leftovers[batch_size:]
    if shuffle:
        rng.shuffle(batches)
    return batches


def padding_waste(batches, length_of=None):
    """Fraction of convolved frames that are padding (0 = none)."""
    if length_of is None:
        def length_of(e):
            return e.z_normalized_transposed_spectrogram().shape[0]
    real = padded = 0
    for batch in batches:
        lengths = [length_of(e) for e in batch]
        real += sum(lengths)
        padded += max(lengths) * len(lengths)
    return 1.0 - real / padded if padded else 0.0


def steps_for_ranks(examples, batch_size, world_size, length_of=None, bucket_width=None, shuffle=True, seed=0):
    """Data-parallel batch formation: a step is a GLOBAL batch of world_size*batch_size utterances cut from one length
    bucket and dealt to the ranks round-robin in length order, so every rank of a step pads to (nearly) the same
    length and does (nearly) the same work -- a synchronous step is as slow as its most expensive rank.
    Returns a list of steps; each step is a list of world_size batches (rank r takes step[r])."""
    if length_of is None:
        def length_of(e):
            return e.z_normalized_transposed_spectrogram().shape[0]
    global_batches = bucket_batches(examples, batch_size * world_size, length_of=length_of, bucket_width=bucket_width,
                                    shuffle=shuffle, seed=seed, drop_last=True)
    steps = []
    for gb in global_batches:
        ordered = sorted(gb, key=length_of)
        steps.append([ordered[r::world_size] for r in range(world_size)])
    return steps


def step_pad_length(step, length_of=None):
    """The frame count every rank of a data-parallel step (one element of steps_for_ranks) pads its batch to: the step's
    longest utterance.  Ranks dealt round-robin from one bucket differ by a few frames in their own maxima; padding all
    of them to the same length gives every rank the same padded geometry -- the same buffer set, tile counts and launch
    grids -- so no rank of a synchronous step runs a longer kernel sequence than the others."""
    if length_of is None:
        def length_of(e):
            return e.z_normalized_transposed_spectrogram().shape[0]
    return max(length_of(e) for batch in step for e in batch)
