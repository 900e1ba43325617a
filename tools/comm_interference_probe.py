#!/usr/bin/env python
"""What does a gradient all-reduce running UNDER backward cost the MFMA kernels?  (single GPU; VERDICT r1 item 5 ii)

The data-parallel step overlaps the all-reduce of bucket 0 (84 MB: output_conv, big_conv_2, big_conv_1) with the rest of
backward and reduces the 20 MB remainder at the end.  No multi-GPU node is available to the builder, so this probe puts a
stand-in for the LOCAL side of that exchange on the communication stream at exactly those two points of the real step:
a kernel that reads two buffers and writes one (dst += src) over the bucket's bytes, run by a fixed, small number of
work-groups so that it moves data at roughly the rate the xGMI links would deliver (7 links x ~50-75 GB/s usable).
Reported: step time without it, with it (per work-group count), the stand-in's own duration and achieved GB/s.

    python tools/comm_interference_probe.py [--steps 30]      -> gpurun_out/comm_interference.json
"""
import argparse
import ctypes
import json
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--exclusive", action="store_true",
                    help="the stand-in's work-groups hold 96 KB of LDS each: they OWN their CUs (no MFMA work-group fits beside "
                         "them), the pessimistic model of RCCL's communication kernels")
    ap.add_argument("--available-cus", type=int, default=0,
                    help="tell the library's grid choosers how many CUs the step's kernels can count on (sl_set_available_cus) "
                         "while the stand-in runs; 0 = leave them at the device's CU count")
    ap.add_argument("--blocks", default="8,16,32,64,128")
    args = ap.parse_args()
    import torch
    import bench
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    from speechless_amd.parallel import GradBucketReducer
    so = Path("/tmp/libcomm_probe.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", str(so),
                    str(ROOT / "tools" / "comm_probe.hip")], check=True)
    probe = ctypes.CDLL(str(so))
    probe.probe_reduce_like.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    probe.probe_reduce_like_excl.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p]

    specs = wav2letter_layer_specs(bench.MEL, bench.K_CLASSES)
    eng = Engine(specs, bench.K_CLASSES, dtype="bf16")
    eng.set_weights(Wav2Letter._glorot_uniform(specs, 2))
    x, labels, lab_len, pred_len = bench.synthetic_batch(0, bench.BATCH_PER_GPU)
    eng.load_input(torch.from_numpy(x).cuda())
    eng.set_labels(labels, lab_len, pred_len)
    ranges = eng.bucket_ranges()
    scratch_src = torch.zeros_like(eng.grads)
    scratch_dst = torch.zeros_like(eng.grads)

    class StandIn(GradBucketReducer):
        """The reducer's stream / event choreography with the collective replaced by the local stand-in kernel."""
        blocks = 0
        durations = []

        def reduce_bucket(self, index):
            lo, hi = self.ranges[index]
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                t0 = torch.cuda.Event(enable_timing=True)
                t1 = torch.cuda.Event(enable_timing=True)
                t0.record()
                if self.blocks and args.exclusive:
                    probe.probe_reduce_like_excl(scratch_dst[lo:hi].data_ptr(), scratch_src[lo:hi].data_ptr(), hi - lo,
                                                 self.blocks, 96 * 1024, self.comm_stream.cuda_stream)
                elif self.blocks:
                    probe.probe_reduce_like(scratch_dst[lo:hi].data_ptr(), scratch_src[lo:hi].data_ptr(), hi - lo,
                                            self.blocks, self.comm_stream.cuda_stream)
                t1.record()
                self.durations.append((index, t0, t1))
                done = torch.cuda.Event()
                done.record(self.comm_stream)
            self._pending.append(done)

    red = StandIn(eng.grads, ranges, force=True)

    def run(blocks):
        red.blocks = blocks
        for _ in range(5):
            eng.train_step_resident(red)
        torch.cuda.synchronize()
        red.durations = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step_resident(red)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        per_bucket = {}
        for index, a, b in red.durations:
            per_bucket.setdefault(index, []).append(a.elapsed_time(b))
        return ms, {i: float(np.mean(v)) for i, v in per_bucket.items()}

    eng_plain_ms = []
    for _ in range(5):
        eng.train_step_resident()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step_resident()
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / args.steps * 1e3
    out = {"step_ms_single_gpu_path": plain, "bucket_bytes": [(hi - lo) * 4 for lo, hi in ranges], "runs": []}
    base, _ = run(0)
    out["step_ms_reducer_choreography_no_traffic"] = base
    out["exclusive_stand_in"] = bool(args.exclusive)
    for blocks in [int(v) for v in args.blocks.split(",")]:
        if args.available_cus:  # the engine tells the choosers of backward's kernels (Engine.comm_cus)
            red.comm_cus = blocks if args.available_cus < 0 else 256 - args.available_cus
        ms, dur = run(blocks)
        entry = {"work_groups": blocks, "step_ms": ms, "slowdown_ms": ms - base, "stand_in_ms_per_bucket": dur,
                 "available_cus_hint": (256 - blocks if args.available_cus < 0 else args.available_cus) or None}
        entry["stand_in_GBps_per_bucket"] = {i: 3 * out["bucket_bytes"][i] / (d * 1e-3) / 1e9 for i, d in dur.items()}
        out["runs"].append(entry)
        print(entry, flush=True)
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    name = "comm_interference{}{}.json".format("_exclusive" if args.exclusive else "",
                                               "_hint" if args.available_cus else "")
    (ROOT / "gpurun_out" / name).write_text(json.dumps(out, indent=1))
    print(json.dumps({k: v for k, v in out.items() if k != "runs"}))


if __name__ == "__main__":
    main()
