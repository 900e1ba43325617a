#!/usr/bin/env python
"""A few resident config-3 (or --config 5) training steps for `rocprofv3 --kernel-trace`: the per-kernel start / end
timestamps show what overlaps what (e.g. the CTC of one half-batch under the top layers of the other, Engine.split_top).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python tools/trace_step.py [--split 0|1]
    python tools/trace_step.py --summarize gpurun_out/trace/.../t_kernel_trace.csv   # one step, kernels in start order
"""
import argparse
import csv
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def summarize(path, step=-2):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # steps are delimited by the optimizer launch
    ends = [i for i, r in enumerate(rows) if "adam_pack_multi" in r["Kernel_Name"]]
    lo, hi = ends[step - 1] + 1, ends[step] + 1
    t0 = int(rows[lo]["Start_Timestamp"])
    prev_end = 0
    print("{:>9} {:>9} {:>8} {:>7}  {}".format("start_us", "end_us", "dur_us", "gap_us", "kernel (stream / queue)"))
    for r in rows[lo:hi]:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
        print("{:9.1f} {:9.1f} {:8.1f} {:7.1f}  {} [q{}]".format(s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name,
                                                             r.get("Queue_Id", "?")))
        prev_end = max(prev_end, e)
    print("step: {:.1f} us".format((int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--split", type=int, default=1)
    ap.add_argument("--config", type=int, default=3)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--summarize", default=None)
    args = ap.parse_args()
    if args.summarize:
        return summarize(args.summarize)
    import torch
    import bench

    class A:
        profile_steps = 1
    b = bench.Bench(args.config, A(), 1, 0, "cuda:0")
    b.eng.split_top = bool(args.split)
    n = len(b.resident) if args.config == 5 else 1
    for _ in range(max(args.steps, 2 * n)):
        b.step()
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
