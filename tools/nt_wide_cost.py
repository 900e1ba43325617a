#!/usr/bin/env python
"""Fixed cost vs per-step cost of the WIDE-layer NT launches (256 x 256 tiles, one work-group per CU): times sl_conv1d_nt on
synthetic operands of big_conv_2's shape (1 x 1, N = 2048 output channels) for several contraction lengths and tile counts
(batch 16 / 32 / 48 / 40 utterances of 512 rows = 256 / 512 / 768 / 640 tiles), back to back.  A linear fit over the
contraction length separates what a tile pays per 64-channel step from what it pays once (prologue, epilogue, launch).

    python tools/nt_wide_cost.py [--taps 1] [--cfg 0] [--epilogue 2] -> gpurun_out/nt_wide_cost.json
"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taps", type=int, default=1)
    ap.add_argument("--cfg", default="0")
    ap.add_argument("--cout", type=int, default=2048)
    ap.add_argument("--epilogue", type=int, default=2, help="2 = bias + ReLU (forward), 3 = ReLU mask (input gradient)")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--batches", default="16,32,48,40")
    ap.add_argument("--cins", default="256,512,1024,2048,4096")
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "nt_wide_cost.json"))
    args = ap.parse_args()
    import torch
    from speechless_amd import _lib
    L = _lib.lib()
    dev = "cuda:0"
    cfg = int(args.cfg, 0)
    rows_pad, halo, t_out = 512, 16, 500
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty((1 << 30,), dtype=torch.uint8, device=dev)
    out = {"rows": []}
    for batch in [int(b) for b in args.batches.split(",")]:
        per_batch = []
        for cin in [int(c) for c in args.cins.split(",")]:
            rows = halo + rows_pad + halo
            x = (torch.randn((batch, rows, cin), device=dev) * 0.5).to(torch.bfloat16)
            w = (torch.randn((args.cout, args.taps, cin), device=dev) * 0.02).to(torch.bfloat16)
            y = torch.zeros((batch, rows, args.cout), dtype=torch.bfloat16, device=dev)
            mask = (torch.randn((batch, rows, args.cout), device=dev)).to(torch.bfloat16)
            bias = torch.zeros((args.cout,), dtype=torch.float32, device=dev)
            g = _lib.ConvGeom()
            g.batch, g.t_out, g.taps, g.cin, g.cout = batch, t_out, args.taps, cin, args.cout
            g.x_row0, g.x_row_stride, g.x_batch_stride = halo - (args.taps - 1) // 2, cin, rows * cin
            g.y_row0, g.y_row_stride, g.y_batch_stride = halo, args.cout, rows * args.cout

            def launch():
                L.call("sl_conv1d_nt", x.data_ptr(), w.data_ptr(), bias.data_ptr(), mask.data_ptr(), y.data_ptr(),
                       ctypes.byref(g), args.epilogue, _lib.SL_BF16, 0, cfg, ws.data_ptr(), ws.numel(), st)
            for _ in range(3):
                launch()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                launch()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / args.reps * 1e3
            nsteps = args.taps * cin // 64
            tiles = batch * 2 * (args.cout // 256)
            flops = 2.0 * batch * t_out * args.taps * cin * args.cout
            row = {"batch": batch, "tiles": tiles, "cin": cin, "nsteps": nsteps, "us": us, "TFLOPs": flops / us / 1e6}
            per_batch.append(row)
            out["rows"].append(row)
            print(row, flush=True)
            del x, w, y, mask
        n = np.array([r["nsteps"] for r in per_batch], dtype=np.float64)
        t = np.array([r["us"] for r in per_batch])
        slope, icpt = np.polyfit(n, t, 1)
        rounds = -(-per_batch[0]["tiles"] // 256)
        fit = {"batch": batch, "tiles": per_batch[0]["tiles"], "rounds": rounds, "us_per_step_and_launch": float(slope),
               "fixed_us_per_launch": float(icpt), "us_per_step_and_round": float(slope) / rounds,
               "fixed_us_per_round": float(icpt) / rounds}
        out.setdefault("fits", []).append(fit)
        print(fit, flush=True)
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
