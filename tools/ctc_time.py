#!/usr/bin/env python
"""Times sl_ctc_loss_grad alone (lattice + gradient [+ repair pass]) at a BASELINE shape, per lattice variant.
    python tools/ctc_time.py [--batch 32] [--frames 500] [--lmax 200]        (SL_LIB_PATH selects a probe build)"""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=500)
    ap.add_argument("--lmax", type=int, default=200)
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--logit-scale", type=float, default=1.0, help="std of the random logits (0.01: the near-uniform softmax "
                    "of a freshly initialised net)")
    ap.add_argument("--blank-bias", type=float, default=0.0, help="added to the blank logit (25: the 'blank collapse' every "
                    "CTC training run passes through early on)")
    ap.add_argument("--aligned", type=float, default=0.0,
                    help="> 0: the regime of a net that has LEARNT its labels -- this is added to the logit of the symbol a "
                         "random valid alignment of each label puts on a frame (20: p ~ 1 on the path, 1e-9 elsewhere)")
    ap.add_argument("--mismatch", type=float, default=0.0,
                    help="with --aligned: this fraction of each label is replaced by other graphemes AFTER the logits were "
                         "built -- the net is sure of a transcript that is partly wrong")
    ap.add_argument("--ragged", action="store_true", help="input lengths U{frames/4 .. frames} instead of all = frames")
    args = ap.parse_args()
    import torch
    from speechless_amd import _lib
    lib = _lib.lib()
    b, t, k = args.batch, args.frames, 29
    rng = np.random.RandomState(0)
    dev = "cuda:0"
    lg = (rng.randn(b, t, k) * args.logit_scale).astype(np.float32)
    lg[:, :, k - 1] += args.blank_bias
    logits = torch.tensor(lg, device=dev)
    probs = torch.zeros((b, t, k), dtype=torch.float32, device=dev)
    logq = torch.zeros_like(probs)
    lab_len = rng.randint(20, args.lmax + 1, size=b).astype(np.int32)
    labels = np.zeros((b, args.lmax), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, k - 1, size=n)
    if args.aligned > 0:
        lg = (rng.randn(b, t, k) * args.logit_scale).astype(np.float32)
        for i, n in enumerate(lab_len):
            # a random monotone alignment: every label once (blank between repeats), the rest of the frames blank / repeats
            seq = []
            for j in range(n):
                if j and labels[i, j] == labels[i, j - 1]:
                    seq.append(k - 1)
                seq.append(int(labels[i, j]))
            cuts = np.sort(rng.choice(np.arange(1, t), size=len(seq) - 1, replace=False)) if len(seq) > 1 else np.array([], int)
            bounds = np.concatenate([[0], cuts, [t]])
            for j, sym in enumerate(seq):
                lo, hi = int(bounds[j]), int(bounds[j + 1])
                lg[i, lo, sym] += args.aligned
                lg[i, lo + 1:hi, k - 1] += args.aligned  # hold with blanks
        logits = torch.tensor(lg, device=dev)
        if args.mismatch > 0:
            for i, n in enumerate(lab_len):
                wrong = rng.rand(n) < args.mismatch
                labels[i, :n] = np.where(wrong, (labels[i, :n] + 1 + rng.randint(0, k - 2, size=n)) % (k - 1), labels[i, :n])
    lab = torch.tensor(labels, device=dev)
    ll = torch.tensor(lab_len, device=dev)
    il = torch.full((b,), t, dtype=torch.int32, device=dev)
    if args.ragged:
        il = torch.tensor(rng.randint(t // 4, t + 1, size=b).astype(np.int32), device=dev)
    loss = torch.zeros((b,), dtype=torch.float32, device=dev)
    dl = torch.zeros((b, t, 128), dtype=torch.bfloat16, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lib.call("sl_softmax_logq", logits.data_ptr(), probs.data_ptr(), logq.data_ptr(), b, t, k, k, t * k, 1e-8, st)
    need = lib.raw("sl_ctc_workspace_bytes")(b, t, args.lmax)
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    ref = None
    for variant, name in ((1, "log-domain lattice"), (2, "wave lattice (double), no repair pass"),
                          (4, "wave lattice (double) + repair pass"), (6, "wave lattice (float), no repair pass"),
                          (5, "wave lattice (float) + repair pass"), (9, "wave PAIR lattice (double), no repair pass"),
                          (8, "wave PAIR lattice (double) + repair pass"), (11, "wave lattice + HELPER wave, no repair pass"),
                          (10, "wave lattice + HELPER wave + repair pass"), (0, "automatic (bf16 output)")):
        lib.call("sl_ctc_select", variant)

        def run():
            lib.call("sl_ctc_loss_grad", probs.data_ptr(), logq.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(),
                     loss.data_ptr(), dl.data_ptr(), b, t, k, args.lmax, 0, 128, t * 128, _lib.SL_BF16, 1e-8, 1.0 / b,
                     ws.data_ptr(), need, st)
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        g = dl.float().cpu().numpy()
        if ref is None:
            ref = (loss.cpu().numpy().copy(), g)
        print("{:44s} {:8.1f} us per call   (mean loss {:.4f}; vs log-domain: loss {:.2e}, gradient rel-L2 {:.2e})".format(
            name, e0.elapsed_time(e1) / args.reps * 1e3, float(loss.mean()),
            float(np.abs(loss.cpu().numpy() - ref[0]).max() / np.abs(ref[0]).max()),
            float(np.linalg.norm(g - ref[1]) / np.linalg.norm(ref[1]))))
    lib.call("sl_ctc_select", 0)


if __name__ == "__main__":
    main()
