#!/usr/bin/env python
"""Fixed cost vs per-step cost of the short-layer NT kernel: times the forward launch of a 250 -> 250 channel
convolution at the BASELINE config-3 row count (B=32, 500 output frames) for kernel sizes 1 ... 31, i.e. 4 ... 124
K-steps of the same tile grid, (a) repeated back to back and (b) each launch right after the launch that produces its
input (cold L2, as inside the training step).  A linear fit gives the per-step time and the fixed per-launch cost.

    python tools/nt_scaling.py [--cfg wm,wn,stages,ksplit,it[,ilv]] -> gpurun_out/nt_scaling.json
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
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--cfgs", default="0")  # comma separated raw cfg words (0 = auto)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "nt_scaling.json"))
    ap.add_argument("--attribute", action="store_true", help="also time the launch behind kernels that rewrite only the "
                    "activations / only the weights / 1 GiB of unrelated memory")
    args = ap.parse_args()

    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import Engine, wav2letter_layer_specs

    rng = np.random.RandomState(0)
    out = {}
    for cfg in [int(c, 0) for c in args.cfgs.split(",")]:
        rows = []
        for k in (1, 5, 7, 15, 31):
            specs = wav2letter_layer_specs(128, 29, inner_count=1, inner_kernel=k, out_filter_count=256, big_kernel=1)
            eng = Engine(specs, 29, dtype="bf16")
            weights = []
            for s in specs:
                limit = np.sqrt(6.0 / (s.kernel_size * (s.cin + s.cout)))
                weights.append((rng.uniform(-limit, limit, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32),
                                rng.uniform(-0.05, 0.05, size=(s.cout,)).astype(np.float32)))
            eng.set_weights(weights)
            eng.load_input(rng.randn(args.batch, args.frames, 128).astype(np.float32))
            eng.forward()
            torch.cuda.synchronize()
            buf = eng.cur
            ws = torch.empty((64 << 20,), dtype=torch.uint8, device=eng.device)
            st = torch.cuda.current_stream().cuda_stream

            def fwd(i, c):
                p = eng.plans[i]
                xin = buf.x0 if i == 0 else buf.y[i - 1]
                _, bias = eng.layer_param_views(eng.params, p)
                eng.lib.call("sl_conv1d_nt", xin.data_ptr(), eng.w_fwd[i].data_ptr(), bias.data_ptr(), None,
                             buf.y[i].data_ptr(), ctypes.byref(buf.fwd_geom[i]), _lib.EPI_BIAS_RELU, eng.dtype_code, 0, c,
                             ws.data_ptr(), ws.numel(), st)

            try:
                for _ in range(3):
                    fwd(1, cfg)
            except Exception as e:  # noqa: BLE001  (a tile configuration this kernel size rules out)
                print(cfg, "taps", k, "skipped:", str(e)[:80], flush=True)
                continue
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fwd(1, cfg)
            b.record()
            torch.cuda.synchronize()
            warm = a.elapsed_time(b) / args.reps
            def behind(producer):
                evs = []
                for _ in range(args.reps):
                    producer()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    fwd(1, cfg)
                    e.record()
                    evs.append((s, e))
                torch.cuda.synchronize()
                return float(np.median([s.elapsed_time(e) for s, e in evs])) * 1e3

            cold = behind(lambda: fwd(0, 0))  # the producer of the input (the step's situation)
            row = {"taps": k, "nsteps": buf.fwd_geom[1].taps * (buf.fwd_geom[1].cin // 64), "back_to_back_us": warm * 1e3,
                   "after_producer_us": cold}
            if args.attribute:  # which operand's cold first touch costs what
                a_saved, w_saved = buf.y[0].clone(), eng.w_fwd[1].clone()
                big = torch.empty((1 << 30,), dtype=torch.uint8, device=eng.device)
                row["after_rewriting_activations_us"] = behind(lambda: buf.y[0].copy_(a_saved))
                row["after_rewriting_weights_us"] = behind(lambda: eng.w_fwd[1].copy_(w_saved))
                row["after_flushing_the_caches_us"] = behind(lambda: big.fill_(1))
                del big
            rows.append(row)
            print(cfg, rows[-1], flush=True)
            del eng, buf, ws
        n = np.array([r["nsteps"] for r in rows], dtype=np.float64)
        fit = {}
        for key in ("back_to_back_us", "after_producer_us"):
            y = np.array([r[key] for r in rows])
            slope, icpt = np.polyfit(n, y, 1)
            fit[key] = {"us_per_step": float(slope), "fixed_us": float(icpt)}
        print(cfg, fit, flush=True)
        out[str(cfg)] = {"rows": rows, "fit": fit}
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
