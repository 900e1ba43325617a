#!/usr/bin/env python
"""Times every NT tile configuration (sl_conv1d_nt cfg word) for every layer's forward and dgrad launch at the
BASELINE config-3 shape (B=32, 1000 frames) on the GPU box and checks each against the default configuration's output.
Writes gpurun_out/tune_nt.json; the winners are hard-coded in conv_nt_bf16.hip:auto_cfg (copy kept under profiles/).

    python tools/tune_kernels.py [--batch 32] [--frames 1000] [--reps 5]
"""
import argparse
import ctypes
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def cfg_word(wm, wn, stages, ksplit, it=4, m32=0, gml=0, slab=0, ilv=0):
    return (wm | (wn << 4) | (stages << 8) | (ksplit << 12) | (it << 20) | (m32 << 24) | (gml << 25) | (slab << 29) |
            (ilv << 30))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--out", default=str(ROOT / "gpurun_out" / "tune_nt.json"))
    ap.add_argument("--layers", default="", help="comma separated layer names (default: all)")
    ap.add_argument("--after-producer", action="store_true",
                    help="time each launch right after a kernel that rewrites its input (the situation inside the step: "
                         "operands come from the Infinity Cache, not from a warm L2) instead of back to back")
    args = ap.parse_args()

    import torch
    from speechless_amd import _lib
    from speechless_amd.engine import Engine, wav2letter_layer_specs

    specs = wav2letter_layer_specs(128, 29)
    eng = Engine(specs, 29, dtype="bf16")
    rng = np.random.RandomState(0)
    weights = []
    for s in specs:
        limit = np.sqrt(6.0 / (s.kernel_size * (s.cin + s.cout)))
        weights.append((rng.uniform(-limit, limit, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32),
                        rng.uniform(-0.05, 0.05, size=(s.cout,)).astype(np.float32)))
    eng.set_weights(weights)
    x = rng.randn(args.batch, args.frames, 128).astype(np.float32)
    lab_len = rng.randint(20, 201, size=args.batch)
    labels = -np.ones((args.batch, 200), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, 28, size=n)
    eng.load_input(x)
    eng.set_labels(labels, lab_len, np.full(args.batch, args.frames // 2))
    eng.forward()
    eng.ctc()
    eng.backward()  # fills every y[] and g[] with realistic data
    torch.cuda.synchronize()
    buf = eng.cur
    ws = torch.empty((512 << 20,), dtype=torch.uint8, device=eng.device)
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(2, 2, 2, 4), (2, 2, 3, 4), (2, 2, 4, 4), (4, 2, 2, 4), (4, 2, 3, 4), (2, 4, 2, 4), (2, 4, 3, 4), (4, 4, 2, 4),
              (4, 2, 2, 2), (4, 2, 3, 2), (4, 2, 4, 2), (8, 2, 2, 2), (8, 2, 3, 2),
              (2, 2, 2, 2), (2, 2, 3, 2), (2, 2, 4, 2), (2, 4, 2, 2), (2, 4, 3, 2),  # 64-row tiles: 2+ work-groups per CU
              # stages | 8 = register-pipelined loop
              (2, 2, 10, 4), (2, 2, 11, 4), (2, 2, 12, 4), (4, 2, 10, 2), (4, 2, 11, 2), (4, 2, 12, 2),
              (4, 2, 10, 4), (4, 2, 11, 4), (2, 4, 10, 4), (2, 4, 11, 4),
              (2, 4, 10, 8), (2, 4, 2, 8), (2, 2, 10, 8), (2, 2, 11, 8),  # it = 8: 128x64 wave patches, 8 waves per 256-row tile
              # it = 32: the v_mfma_f32_32x32x16_bf16 variant (64x64 wave patch)
              (2, 2, 2, 32), (2, 2, 3, 32), (2, 2, 4, 32), (4, 2, 2, 32), (4, 2, 3, 32), (2, 4, 2, 32), (2, 4, 3, 32),
              (4, 4, 2, 32)]
    # slab variant (chunk-major, activation slab kept in LDS): it + 100
    shapes += [(4, 2, 10, 302), (4, 2, 11, 302), (4, 2, 12, 302), (2, 2, 10, 304), (2, 2, 11, 304), (2, 2, 12, 304),
               (2, 4, 10, 308),  # it + 300: interleaved tap-major
               (4, 4, 2, 104), (2, 4, 10, 108), (2, 4, 10, 208), (2, 2, 10, 204), (2, 2, 11, 204), (2, 2, 12, 204), (2, 2, 11, 405), (2, 2, 12, 405), (4, 2, 11, 102), (4, 2, 3, 102), (2, 2, 11, 104), (2, 2, 12, 104)]
    results = {}
    n = len(eng.plans)
    only = set(filter(None, args.layers.split(",")))

    def timed(fn, inputs):
        """ms per launch of fn(): back to back, or (--after-producer) each launch behind a copy kernel that rewrites
        its input tensors, with events around the launch alone"""
        if not args.after_producer:
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start.record()
            for _ in range(args.reps):
                fn()
            stop.record()
            torch.cuda.synchronize()
            return start.elapsed_time(stop) / args.reps
        saved = [t.clone() for t in inputs]
        evs = []
        for _ in range(args.reps):
            for t, sv in zip(inputs, saved):
                t.copy_(sv)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            evs.append((a, b))
        torch.cuda.synchronize()
        return float(np.median([a.elapsed_time(b) for a, b in evs]))

    def run(kind, p, cfg):
        i = p.index
        if kind == "fwd":
            last = i == n - 1
            xin = buf.x0 if i == 0 else buf.y[i - 1]
            out = buf.logits if last else buf.y[i]
            _, bias = eng.layer_param_views(eng.params, p)
            eng.lib.call("sl_conv1d_nt", xin.data_ptr(), eng.w_fwd[i].data_ptr(), bias.data_ptr(), None, out.data_ptr(),
                         ctypes.byref(buf.fwd_geom[i]), _lib.EPI_BIAS if last else _lib.EPI_BIAS_RELU, eng.dtype_code,
                         1 if last else 0, cfg, ws.data_ptr(), ws.numel(), st)
            return out
        eng.lib.call("sl_conv1d_nt", buf.g[i].data_ptr(), eng.w_dgrad[i].data_ptr(), None, buf.y[i - 1].data_ptr(),
                     buf.g[i - 1].data_ptr(), ctypes.byref(buf.dgrad_geom[i]), _lib.EPI_RELU_MASK, eng.dtype_code, 0, cfg,
                     ws.data_ptr(), ws.numel(), st)
        return buf.g[i - 1]

    # dgrad overwrites g[i-1], which is the input of the next dgrad: tune from the first layer up so inputs stay intact
    jobs = [("fwd", p) for p in eng.plans] + [("dgrad", p) for p in eng.plans[1:]]
    for kind, p in jobs:
        if only and p.spec.name not in only:
            continue
        geom = buf.fwd_geom[p.index] if kind == "fwd" else buf.dgrad_geom[p.index]
        nsteps = geom.taps * (geom.cin // 64)
        key = "{}:{}".format(kind, p.spec.name)
        ref = run(kind, p, cfg_word(2, 2, 2, 1)).clone()
        torch.cuda.synchronize()
        rows = []
        for (wm, wn, stg, it) in shapes:
            if geom.cout % (64 * wn):
                continue
            slab = 100 <= it < 300
            if slab and (geom.taps < (stg & 7) or geom.taps > 33):
                continue
            for ks in (1, 2, 4, 8):
                if ks > 1 and nsteps < 12 * ks:
                    continue
                if slab and ks > geom.cin // 64:
                    continue
                cfg = (cfg_word(wm, wn, stg, ks, 4, 1) if it == 32 else
                       cfg_word(wm, wn, stg, ks, 5, ilv=1) if it == 405 else
                       cfg_word(wm, wn, stg, ks, it - 300, ilv=1) if it >= 300 else
                       cfg_word(wm, wn, stg, ks, it - 200, slab=1, ilv=1) if it >= 200 else
                       cfg_word(wm, wn, stg, ks, it - 100, slab=1) if slab else cfg_word(wm, wn, stg, ks, it))
                try:
                    out = run(kind, p, cfg)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    rows.append({"cfg": [wm, wn, stg, ks, it], "error": str(e)[:200]})
                    continue
                err = float((out.float() - ref.float()).abs().max() / (ref.float().abs().max() + 1e-30))
                src = (buf.x0 if p.index == 0 else buf.y[p.index - 1]) if kind == "fwd" else buf.g[p.index]
                ms = timed(lambda: run(kind, p, cfg), [src])
                rows.append({"cfg": [wm, wn, stg, ks, it], "ms": ms, "max_rel_diff": err})
        run(kind, p, cfg_word(2, 2, 2, 1))  # restore the reference output for downstream jobs
        ok = [r for r in rows if "ms" in r]
        ok.sort(key=lambda r: r["ms"])
        flops = 2.0 * args.batch * buf.t_out * p.spec.kernel_size * p.spec.cin * p.spec.cout
        results[key] = {"algorithmic_gflop": flops / 1e9, "nsteps": nsteps, "cout_pad": geom.cout,
                        "best": ok[:4], "all": rows}
        best = ok[0]
        print("{:22s} best [wm,wn,stages,ksplit,it] {} {:.4f} ms {:.0f} TFLOP/s | next {} | default 2,2,2,1,4: {:.4f} ms".format(
            key, best["cfg"], best["ms"], flops / best["ms"] / 1e9,
            [(r["cfg"], round(r["ms"], 4)) for r in ok[1:3]],
            [r["ms"] for r in ok if r["cfg"] == [2, 2, 2, 1, 4]][0]))
        bad = [r for r in ok if r["max_rel_diff"] > 2e-2]
        if bad:
            print("   !! configurations deviating from the default output:", bad[:3])
    # ---- wgrad: cfg word wm | wn<<4 | stages<<8 | splits<<12
    for p in eng.plans:
        if only and p.spec.name not in only:
            continue
        i = p.index
        geom = buf.wgrad_geom[i]
        xin = buf.x0 if i == 0 else buf.y[i - 1]
        dw, _ = eng.layer_param_views(eng.grads, p)
        key = "wgrad:" + p.spec.name

        def run_w(cfg):
            eng.lib.call("sl_conv1d_wgrad", xin.data_ptr(), buf.g[i].data_ptr(), dw.data_ptr(), ctypes.byref(geom),
                         eng.dtype_code, cfg, ws.data_ptr(), ws.numel(), st)
        run_w(cfg_word(2, 2, 2, 0, 0))
        ref = dw.clone()
        rows = []
        for (wm, wn, stg) in [(2, 2, 2), (2, 2, 3), (2, 2, 4), (4, 2, 2), (4, 2, 3), (2, 4, 2), (2, 4, 3), (4, 4, 2),
                              (4, 4, 10)]:
            if geom.cin % (64 * wm) or geom.cout % (64 * wn):
                continue
            for sp in (0, 1, 2, 4, 8, 16, 32):
                if sp > args.batch:
                    continue
                cfg = cfg_word(wm, wn, stg, sp, 0)
                try:
                    run_w(cfg)
                    torch.cuda.synchronize()
                except Exception as e:  # noqa: BLE001
                    rows.append({"cfg": [wm, wn, stg, sp], "error": str(e)[:200]})
                    continue
                err = float((dw - ref).abs().max() / (ref.abs().max() + 1e-30))
                ms = timed(lambda: run_w(cfg), [buf.g[i]])
                rows.append({"cfg": [wm, wn, stg, sp], "ms": ms, "max_rel_diff": err})
        ok = sorted([r for r in rows if "ms" in r], key=lambda r: r["ms"])
        flops = 2.0 * args.batch * buf.t_out * p.spec.kernel_size * p.spec.cin * p.spec.cout
        results[key] = {"algorithmic_gflop": flops / 1e9, "best": ok[:4], "all": rows}
        print("{:22s} best [wm,wn,stages,splits] {} {:.4f} ms {:.0f} TFLOP/s | next {} | errs {}".format(
            key, ok[0]["cfg"], ok[0]["ms"], flops / ok[0]["ms"] / 1e9, [(r["cfg"], round(r["ms"], 4)) for r in ok[1:4]],
            [r for r in rows if "error" in r or r.get("max_rel_diff", 0) > 1e-3][:2]))
    Path(args.out).parent.mkdir(exist_ok=True)
    Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
