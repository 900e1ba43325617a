#!/usr/bin/env python
"""Runs ONE conv launch of the config-3 step repeatedly (for rocprofv3 --pmc passes on a single kernel).

    python tools/run_one.py --kind fwd|dgrad|wgrad|wgrad_grouped --layer big_conv_1 [--cfg 0] [--reps 20]
"""
import argparse
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="fwd")
    ap.add_argument("--layer", default="big_conv_1")
    ap.add_argument("--cfg", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--bins", type=int, default=128, help="input bins (257: config 5)")
    args = ap.parse_args()
    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import Engine, wav2letter_layer_specs

    specs = wav2letter_layer_specs(args.bins, 29)
    eng = Engine(specs, 29, dtype="bf16")
    rng = np.random.RandomState(0)
    weights = []
    for s in specs:
        limit = np.sqrt(6.0 / (s.kernel_size * (s.cin + s.cout)))
        weights.append((rng.uniform(-limit, limit, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32),
                        rng.uniform(-0.05, 0.05, size=(s.cout,)).astype(np.float32)))
    eng.set_weights(weights)
    x = rng.randn(args.batch, args.frames, args.bins).astype(np.float32)
    lab_len = rng.randint(20, 201, size=args.batch)
    labels = -np.ones((args.batch, 200), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    eng.load_input(x)
    eng.set_labels(labels, lab_len, np.full(args.batch, args.frames // 2))
    eng.forward()
    eng.ctc()
    eng.backward()
    torch.cuda.synchronize()
    buf = eng.cur
    p = [q for q in eng.plans if q.spec.name == args.layer][0]
    i = p.index
    n = len(eng.plans)
    ws = torch.empty((512 << 20,), dtype=torch.uint8, device=eng.device)
    wsz = torch.zeros((64 << 20,), dtype=torch.uint8, device=eng.device)  # zeroed: work-group tickets of bwd1x1
    st = torch.cuda.current_stream().cuda_stream
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(args.reps):
        if args.kind == "fwd":
            last = i == n - 1
            xin = buf.x0 if i == 0 else buf.y[i - 1]
            out = buf.logits if last else buf.y[i]
            _, bias = eng.layer_param_views(eng.params, p)
            eng.lib.call("sl_conv1d_nt", xin.data_ptr(), eng.w_fwd[i].data_ptr(), bias.data_ptr(), None, out.data_ptr(),
                         ctypes.byref(buf.fwd_geom[i]), _lib.EPI_BIAS if last else _lib.EPI_BIAS_RELU, eng.dtype_code,
                         1 if last else 0, args.cfg, ws.data_ptr(), ws.numel(), st)
        elif args.kind == "dgrad":
            eng.lib.call("sl_conv1d_nt", buf.g[i].data_ptr(), eng.w_dgrad[i].data_ptr(), None, buf.y[i - 1].data_ptr(),
                         buf.g[i - 1].data_ptr(), ctypes.byref(buf.dgrad_geom[i]), _lib.EPI_RELU_MASK, eng.dtype_code, 0,
                         args.cfg, ws.data_ptr(), ws.numel(), st)
        elif args.kind == "wgrad_multi":  # the balanced launch for striding_conv + inner_conv_1..7 (--layer ignored)
            eng._launch_wgrad_multi(buf, list(range(0, 8)) if args.cfg == 0 else list(range(1, 8)), st)
        elif args.kind == "bwd1x1":  # both gradients of the 1x1 output layer in one launch (--layer output_conv)
            dw, _ = eng.layer_param_views(eng.grads, p)
            eng.lib.call("sl_conv1d_backward_1x1", buf.y[i - 1].data_ptr(), buf.g[i].data_ptr(), eng.w_dgrad[i].data_ptr(),
                         buf.g[i - 1].data_ptr(), dw.data_ptr(), ctypes.byref(buf.wgrad_geom[i]), _lib.EPI_RELU_MASK, 29,
                         eng.dtype_code, args.cfg, wsz.data_ptr(), wsz.numel(), st)
        elif args.kind == "wgrad_grouped":  # --layer = first layer of a run of identical layers (inner_conv_1)
            lo, hi = [r for r in eng.runs if r[0] == i][0]
            dw_lo, _ = eng.layer_param_views(eng.grads, p)
            stride_elems = buf.batch * buf.rows * p.cin_pad
            eng.lib.call("sl_conv1d_wgrad_grouped", buf.y[lo - 1].data_ptr(), buf.g[lo].data_ptr(), dw_lo.data_ptr(),
                         ctypes.byref(buf.wgrad_geom[lo]), hi - lo + 1, stride_elems, stride_elems,
                         p.w_numel + p.cout_pad, args.cfg, buf.wgrad_ws.data_ptr(), buf.wgrad_ws.numel(), st)
        else:
            xin = buf.x0 if i == 0 else buf.y[i - 1]
            dw, _ = eng.layer_param_views(eng.grads, p)
            eng.lib.call("sl_conv1d_wgrad", xin.data_ptr(), buf.g[i].data_ptr(), dw.data_ptr(),
                         ctypes.byref(buf.wgrad_geom[i]), eng.dtype_code, args.cfg, ws.data_ptr(), ws.numel(), st)
    stop.record()
    torch.cuda.synchronize()
    flops = 2.0 * args.batch * buf.t_out * p.spec.kernel_size * p.spec.cin * p.spec.cout
    ms = start.elapsed_time(stop) / args.reps
    print("{}:{} cfg={} {:.4f} ms {:.0f} TFLOP/s".format(args.kind, args.layer, args.cfg, ms, flops / ms / 1e9))


if __name__ == "__main__":
    main()
