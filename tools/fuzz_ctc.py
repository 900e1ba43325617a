#!/usr/bin/env python
"""sl_ctc_loss_grad against the float64 oracle on RANDOM regimes: class counts 5..29, 20..700 frames, ragged input lengths,
labels from empty to the longest the frames allow (and a few that do not fit), and emission regimes mixed per utterance --
near-uniform, sharp random, blank collapse, a learnt alignment of varying strength (optionally of a partly different
transcript), and half-and-half mixtures of two of these along the time axis.  Per case: loss and gradient against the oracle,
and whether the repair pass had to run (default variant vs the probability-domain lattice alone).

    python tools/fuzz_ctc.py [--cases 40] [--seed 0]"""
import argparse
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def regime_logits(rng, label, t, k, kind):
    lg = rng.randn(t, k).astype(np.float32)
    if kind == "uniform":
        lg *= 0.01
    elif kind == "sharp":
        lg *= rng.uniform(4, 12)
    elif kind == "collapse":
        lg[:, k - 1] += rng.uniform(15, 40)
    elif kind in ("learnt", "wrong"):
        lab = list(label)
        if kind == "wrong":
            lab = [int((c + 1 + rng.randint(0, k - 2)) % (k - 1)) if rng.rand() < 0.4 else c for c in lab]
        seq = []
        for j, c in enumerate(lab):
            if j and c == lab[j - 1]:
                seq.append(k - 1)
            seq.append(int(c))
        strength = rng.uniform(6, 40)
        if len(seq) and len(seq) <= t:
            cuts = np.sort(rng.choice(np.arange(1, t), size=len(seq) - 1, replace=False)) if len(seq) > 1 else np.array([], int)
            bounds = np.concatenate([[0], cuts, [t]]).astype(int)
            for j, sym in enumerate(seq):
                lg[bounds[j], sym] += strength
                lg[bounds[j] + 1:bounds[j + 1], k - 1] += strength
        else:
            lg[:, k - 1] += strength
    return lg


def draw_case(rng, kinds=("uniform", "sharp", "collapse", "learnt", "wrong"), tight=False):
    """the next case of the stream: (k, t, input_len, lab_len, labels_list, logits, regime names).  tight: labels that fill
    60 .. 100 % of the frames (up to 255 graphemes) -- alignments with next to no slack, the regime of the lane that missed its
    lift (round 6); always with a second regime in the other half of the utterance"""
    kinds = list(kinds)
    k = int(rng.choice([5, 12, 29]))
    t = int(rng.choice([rng.randint(20, 60), rng.randint(60, 300), rng.randint(300, 700)]))
    b = int(rng.randint(1, 5))
    input_len = [int(rng.randint(max(2, t // 2), t + 1)) for _ in range(b)]
    if tight:
        lab_len = [int(min(255, rng.randint(int(0.6 * il), il + 1))) for il in input_len]
    else:
        lab_len = [int(rng.randint(0, min(200, il) + 1)) if rng.rand() < 0.9 else int(min(250, il + rng.randint(1, 10)))
                   for il in input_len]
    labels_list = [list(rng.randint(0, k - 1, size=n)) for n in lab_len]
    logits = np.zeros((b, t, k), dtype=np.float32)
    desc = []
    for i in range(b):
        a, c = rng.choice(kinds), rng.choice(kinds)
        la = regime_logits(rng, labels_list[i], input_len[i], k, a)
        if rng.rand() < (1.0 if tight else 0.4):
            lc = regime_logits(rng, labels_list[i], input_len[i], k, c)
            h = input_len[i] // 2
            la[h:] = lc[h:]
            a = a + "|" + c
        logits[i, :input_len[i]] = la
        desc.append(a)
    return k, t, input_len, lab_len, labels_list, logits, desc


def replay_case(seed, index):
    """case `index` of `python tools/fuzz_ctc.py --seed <seed>` (the stream is replayed on the host up to it)"""
    rng = np.random.RandomState(seed)
    case = None
    for _ in range(index + 1):
        case = draw_case(rng)
    return case


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--pair", action="store_true", help="the wave-PAIR lattice (sl_ctc_select 8 / 9) instead of the default one")
    ap.add_argument("--tight", action="store_true", help="labels filling 60 .. 100 %% of the frames, two regimes per utterance")
    ap.add_argument("--helped", action="store_true", help="the lattice wave with a helper wave (sl_ctc_select 10 / 11)")
    args = ap.parse_args()
    from oracle import w2l_oracle as o
    from speechless_amd import _lib
    from test_gpu_parity import run_ctc_kernel
    lib = _lib.lib()
    rng = np.random.RandomState(args.seed)
    kinds = ["uniform", "sharp", "collapse", "learnt", "wrong"]
    repaired = 0
    worst_loss = worst_grad = 0.0
    for case in range(args.cases):
        k, t, input_len, lab_len, labels_list, logits, desc = draw_case(rng, kinds, tight=args.tight)
        b = logits.shape[0]
        labels = o.pack_label_batch([l if l else [-1] for l in labels_list])
        ref_p = o.softmax(logits.astype(np.float64))
        ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
        ref_dl = o.softmax_backward(ref_p, ref_dp)
        lib.call("sl_ctc_select", 10 if args.helped else (8 if args.pair else 0))
        _, loss, dl = run_ctc_kernel(lib, logits, labels, lab_len, input_len)
        lib.call("sl_ctc_select", 11 if args.helped else (9 if args.pair else 2))
        _, loss2, dl2 = run_ctc_kernel(lib, logits, labels, lab_len, input_len)
        lib.call("sl_ctc_select", 0)
        needed_repair = not (np.array_equal(loss, loss2, equal_nan=True) and np.array_equal(dl, dl2, equal_nan=True))
        repaired += int(needed_repair)
        fin = np.isfinite(ref_loss)
        assert np.array_equal(np.isinf(loss), ~fin), (case, desc, loss, ref_loss)
        el = float(np.max(np.abs(loss[fin] - ref_loss[fin]) / np.maximum(np.abs(ref_loss[fin]), 20.0))) if fin.any() else 0.0
        eg = float(np.abs(dl[fin] - ref_dl[fin]).max()) if fin.any() else 0.0
        worst_loss, worst_grad = max(worst_loss, el), max(worst_grad, eg)
        print("case %3d  k %2d  t %3d  labels %s  %s  loss err %.1e  gradient err %.1e%s" % (
            case, k, t, lab_len, desc, el, eg, "  (repair pass ran)" if needed_repair else ""), flush=True)
        assert el < 2e-5 and eg < 1e-4, (case, desc, el, eg)
        for i in range(b):
            assert not dl[i, input_len[i]:].any()
    print("all %d cases passed; worst loss error %.1e (relative to max(|loss|, 20)), worst gradient error %.1e absolute; "
          "the repair pass ran in %d case(s)" % (args.cases, worst_loss, worst_grad, repaired))


if __name__ == "__main__":
    main()
