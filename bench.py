#!/usr/bin/env python
"""bench.py -- utterances/sec of the Wav2Letter training hot path (fwd + CTC + bwd + Adam) on N MI355X.

Workload = BASELINE.json config 3 per GPU: synthetic 128-mel x 1000-frame batch of 32 utterances, labels
L_i ~ U{20..200} over the 28 graphemes, random-init (glorot) Wav2Letter, bf16 activations/weights with fp32 accumulate
and fp32 CTC.  A "step" = one full optimisation step over one batch: 11 conv layers forward, softmax, CTC loss and
gradient, 10 dgrads + 11 wgrads + 11 bias grads, [gradient all-reduce over RCCL when N > 1], Adam, weight repack.
Inputs, labels and lengths are resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...

--config 2 / --config 5 run BASELINE.json's other GPU configurations with the same contract (they are not the headline
line): 2 = forward + greedy decode only on the same batch; 5 = long-form 257-bin power spectrograms, 8 utterances per
GPU, lengths U{2000..8000} frames in length-bucketed batches, full training step, reported with the achieved GB/s of
its HBM-bound kernels next to the MFMA fraction.

Rank 0 prints ONE JSON line (see the contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MEL, FRAMES, BATCH_PER_GPU, K_CLASSES = 128, 1000, 32, 29
BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3     # same guide: v_mfma_f32_32x32x2_f32, exact fp32 = the fp32 vector peak


def layer_flops_per_utt(specs, t_out):
    """Algorithmic (unpadded) forward FLOPs per utterance and layer: 2 * T' * k * Cin * Cout."""
    return [2.0 * t_out * s.kernel_size * s.cin * s.cout for s in specs]


def synthetic_batch(rank, batch):
    """Config 3/4 generator: rank r takes utterances [32r, 32r+32) of the global stream."""
    x = np.random.RandomState(1000 + rank).randn(batch, FRAMES, MEL).astype(np.float32)
    rng = np.random.RandomState(2000 + rank)
    lab_len = rng.randint(20, 201, size=batch)
    labels = -np.ones((batch, int(lab_len.max())), dtype=np.int32)
    for i, n in enumerate(lab_len):
        labels[i, :n] = rng.randint(0, K_CLASSES - 1, size=n)
    return x, labels, lab_len.astype(np.int32), np.full((batch,), FRAMES // 2, dtype=np.int32)


LONG_BINS, LONG_BATCH, LONG_CORPUS = 257, 8, 64  # config 5: 257-bin power spectrograms, 8 utterances per GPU and step


def long_form_batches(rank):
    """Config 5 generator: per rank 64 utterances, T_i ~ U{2000..8000} frames (seed 3 + rank),
    L_i <= min(200, T_i / 4), cut into length-bucketed batches of 8 (speechless_amd.batching.bucket_batches).
    Returns [(x (8, Tmax, 257) float32 zero padded, labels, label_lengths, prediction_lengths, true_lengths)]."""
    from speechless_amd.batching import bucket_batches, padding_waste
    rng = np.random.RandomState(3 + rank)
    lengths = rng.randint(2000, 8001, size=LONG_CORPUS)
    corpus = list(range(LONG_CORPUS))
    batches = bucket_batches(corpus, LONG_BATCH, length_of=lambda i: int(lengths[i]), shuffle=False)
    out = []
    for members in batches:
        t_max = max(int(lengths[i]) for i in members)
        x = np.zeros((len(members), t_max, LONG_BINS), dtype=np.float32)
        lab_len = np.zeros(len(members), dtype=np.int32)
        for j, i in enumerate(members):
            x[j, :lengths[i]] = np.random.RandomState(5000 + 100 * rank + i).randn(int(lengths[i]), LONG_BINS)
            lab_len[j] = rng.randint(20, min(200, int(lengths[i]) // 4) + 1)
        labels = -np.ones((len(members), int(lab_len.max())), dtype=np.int32)
        for j, n in enumerate(lab_len):
            labels[j, :n] = rng.randint(0, K_CLASSES - 1, size=n)
        true_len = np.array([int(lengths[i]) for i in members], dtype=np.int32)
        out.append((x, labels, lab_len, (true_len // 2).astype(np.int32), true_len))
    waste = padding_waste([[int(lengths[i]) for i in members] for members in batches], length_of=lambda n: n)
    return out, waste


def cpu_model_string():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(specs_oracle, weights, sample_utts=BATCH_PER_GPU, steps=3):
    """The torch-CPU fp32 port of the same step (oracle/w2l_torch_cpu.py) timed on this node's host cores, on the SAME
    batch as the GPU step (configuration 3, 32 utterances): 1 warm-up + `steps` timed steps."""
    import torch
    from oracle import w2l_torch_cpu as tc
    x, labels, lab_len, pred_len = synthetic_batch(0, sample_utts)
    times = tc.timed_training_steps(specs_oracle, weights, x, labels, pred_len, lab_len, steps=steps, warmup=1)
    mean, best = float(np.mean(times)), float(min(times))
    return {"value": sample_utts / mean, "unit": "utterances/sec", "cores": int(torch.get_num_threads()),
            "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model_string(),
            "value_best_step": sample_utts / best, "step_seconds": [round(t, 3) for t in times],
            "sample": "{} utterances x {} frames (the GPU step's batch), 1 warm-up + {} timed fwd+CTC+bwd+Adam steps, "
                      "torch-CPU fp32, mean step {:.2f} s, best {:.2f} s".format(sample_utts, FRAMES, steps, mean, best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5),
                    help="BASELINE.json configuration: 3 = headline training step (default), 2 = forward + greedy "
                         "decode only, 5 = long-form 257-bin x 2000..8000-frame bucketed batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra event-instrumented steps for the roofline leg")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from speechless_amd.engine import Engine, wav2letter_layer_specs
    from speechless_amd.net import Wav2Letter
    from speechless_amd.parallel import GradBucketReducer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node {} for --gpus {}".format(
                args.gpus, args.gpus))
    # Test hook (single-GPU boxes only): SL_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and moves the bytes with gloo, so
    # that the multi-rank control flow of this script can be exercised where only one GPU exists.  Never set in a
    # measurement: the ranks then share one device.
    share_gpu = os.environ.get("SL_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:{}".format(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))

    bins = LONG_BINS if args.config == 5 else MEL
    specs = wav2letter_layer_specs(bins, K_CLASSES)
    weights = Wav2Letter._glorot_uniform(specs, 2)  # Keras default init, same on every rank
    # configuration 2 asks for bit-exact decoded indices against the CPU path: its headline figure is the fp32 path
    # (fp32 storage, exact-fp32 MFMA); the bf16 path is timed next to it and its disagreements are counted
    main_dtype = "f32" if args.config == 2 else "bf16"
    peak_tflops = F32_MFMA_PEAK_TFLOPS if main_dtype == "f32" else BF16_DENSE_PEAK_TFLOPS
    eng = Engine(specs, K_CLASSES, dtype=main_dtype, device=device)
    eng.set_weights(weights)
    reducer = None
    if world > 1 and args.config != 2:
        ranges, _ = eng.bucket_ranges()
        reducer = GradBucketReducer(eng.grads, ranges)

    # ---- the step of this configuration (inputs, labels and lengths resident in HBM before the timed region)
    waste = 0.0
    if args.config == 5:
        eng.max_cached_shapes = 16
        host_batches, waste = long_form_batches(rank)
        resident = [(torch.from_numpy(x).to(device), lab, ll, pl, tl) for (x, lab, ll, pl, tl) in host_batches]
        batch_per_gpu = LONG_BATCH
        cursor = [0]

        def step():
            x_dev, lab, ll, pl, _ = resident[cursor[0] % len(resident)]
            cursor[0] += 1
            eng.load_input(x_dev)  # fp32 -> bf16 halo'd layout on the GPU (sl_pack_input), part of the step
            eng.set_labels(lab, ll, pl)
            return eng.train_step_resident(reducer)
        frames_per_step = float(np.mean([tl.sum() for (_, _, _, _, tl) in host_batches]))
        flops_per_step = float(np.mean([sum(3 * sum(layer_flops_per_utt(specs, -(-int(t) // 2))) -
                                            layer_flops_per_utt(specs, -(-int(t) // 2))[0] for t in tl)
                                        for (_, _, _, _, tl) in host_batches]))
    else:
        x, labels, lab_len, pred_len = synthetic_batch(rank, BATCH_PER_GPU)
        eng.load_input(torch.from_numpy(x).to(device))
        eng.set_labels(labels, lab_len, pred_len)
        batch_per_gpu = BATCH_PER_GPU
        fl = layer_flops_per_utt(specs, FRAMES // 2)
        if args.config == 3:
            def step():
                return eng.train_step_resident(reducer)
            flops_per_step = (3 * sum(fl) - fl[0]) * BATCH_PER_GPU
        else:
            def step():
                eng.forward()
                return eng.greedy_decode()
            flops_per_step = sum(fl) * BATCH_PER_GPU
        frames_per_step = float(BATCH_PER_GPU * FRAMES)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(out.mean().item()) if args.config != 2 else None
    bf16_leg = None
    if args.config == 2:
        decoded32, argmax32 = out
        eng16 = Engine(specs, K_CLASSES, dtype="bf16", device=device)
        eng16.set_weights(weights)
        eng16.load_input(torch.from_numpy(x).to(device))
        eng16.set_input_lengths(pred_len)
        eng.set_input_lengths(pred_len)
        for _ in range(args.warmup):
            eng16.forward()
            out16 = eng16.greedy_decode()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            eng16.forward()
            out16 = eng16.greedy_decode()
        sync()
        el16 = time.perf_counter() - t1
        decoded32, argmax32 = eng.greedy_decode()
        decoded16, argmax16 = out16
        bf16_leg = {"dtype": "bf16", "value": batch_per_gpu * world * args.steps / el16, "unit": "utterances/sec",
                    "ms_per_step": el16 / args.steps * 1e3,
                    "mismatching_frames_vs_f32": int((argmax16 != argmax32).sum()), "frames": int(argmax32.size),
                    "mismatching_sequences_vs_f32": int(sum(a != b for a, b in zip(decoded16, decoded32))),
                    "sequences": len(decoded32),
                    "note": "bf16 storage is NOT bit-exact against the fp32 CPU path at random init (near-flat softmax): "
                            "the headline value above is the fp32 path, whose decoded indices are "
                            "(tests/test_gpu_round2.py::test_config2_greedy_decode_bit_exact_at_batch_32)"}

    # ---- data-parallel diagnostics (N > 1): everything needed to read the first multi-GPU run from its one JSON line
    dp = None
    if reducer is not None:
        def timed(n):
            sync()
            t1 = time.perf_counter()
            for _ in range(n):
                step()
            sync()
            dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            return float(dt.item()) / n * 1e3
        # one more step, then: are the reduced gradients and the updated weights bit-identical on every rank?
        step()
        torch.cuda.synchronize()
        sums = torch.stack([eng.grads.double().sum(), eng.grads.view(torch.int32).long().sum().double(),
                            eng.params.double().sum(), eng.params.view(torch.int32).long().sum().double()])
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(gathered, sums)
        same = all(bool(torch.equal(g, gathered[0])) for g in gathered)
        # the exchange alone: both buckets back to back on the communication stream, nothing else running
        ranges, _ = eng.bucket_ranges()
        bucket_bytes = [int((hi - lo) * 4) for lo, hi in ranges]
        scratch = torch.zeros_like(eng.grads)
        for _ in range(3):
            for lo, hi in ranges:
                dist.all_reduce(scratch[lo:hi])
        sync()
        t1 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            for lo, hi in ranges:
                dist.all_reduce(scratch[lo:hi])
        sync()
        ar_ms = (time.perf_counter() - t1) / reps * 1e3
        del scratch
        alg = sum(bucket_bytes) / (ar_ms * 1e-3) / 1e9
        with_comm_ms = timed(args.steps)
        reducer.skip_collective = True  # same choreography, no bytes on the wire (timing only: weights now differ per rank)
        without_comm_ms = timed(args.steps)
        reducer.skip_collective = False
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # gloo test hook / CPU build
            rccl = None
        dp = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
              "bucket_bytes": bucket_bytes,
              "reduced_gradients_and_weights_identical_on_all_ranks": same,
              "gradient_checksum": float(gathered[0][0].item()), "weight_checksum": float(gathered[0][2].item()),
              "allreduce_alone_ms": ar_ms, "allreduce_algbw_GBps": alg,
              "allreduce_busbw_GBps": alg * 2 * (world - 1) / world,
              "step_ms_with_allreduce": with_comm_ms, "step_ms_without_allreduce": without_comm_ms,
              "exposed_communication_ms": with_comm_ms - without_comm_ms,
              "note": "exposed communication = step time with the bucketed all-reduce minus the same step with the "
                      "collective left out (streams and events unchanged); allreduce_alone = both buckets back to "
                      "back with nothing else on the GPU; busbw = algbw * 2(n-1)/n"}

    # ---- roofline leg: a few more steps with HIP events around every launch (same stream as the kernels)
    def timeline_pass():
        eng.timeline = []
        n_steps = args.profile_steps if args.config != 5 else len(resident)
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        per_tag = {}
        for tag, start, stop in eng.timeline:
            per_tag.setdefault(tag, []).append(start.elapsed_time(stop))
        eng.timeline = None
        return {tag: float(np.sum(v)) / n_steps for tag, v in per_tag.items()}  # ms per step and tag

    live_ms = timeline_pass()  # same conditions as the timed region (bias gradients overlapped on the side stream)

    # ---- host-buffer note: what handing over a HOST batch costs on top of the resident step (never part of `value`)
    h2d = None
    if args.config == 3 and rank == 0:
        host = x  # (32, 1000, 128) float32, pageable -- what net.py:578-587 packs
        for _ in range(2):
            torch.from_numpy(host).to(device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dev = torch.from_numpy(host).to(device)
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t1) / 10 * 1e3
        h2d = {"h2d_ms_per_step": h2d_ms, "bytes": int(host.nbytes),
               "GBps": host.nbytes / (h2d_ms * 1e-3) / 1e9,
               "note": "pageable float32 host batch -> HBM (torch copy, synchronous); the fp32 -> bf16 halo'd repack on "
                       "the GPU (sl_pack_input) is {:.3f} ms and is not in the resident step either; serial upper "
                       "bound of the PCIe-inclusive rate below, speechless_amd/pipeline.py overlaps the copy with the "
                       "previous step".format(live_ms.get("pack_input", 0.0))}
        del dev

    def kernel_pass(tags):
        """ms per launch of the MAIN kernel behind each tag: events recorded by the library immediately around that
        kernel on its stream (sl_profile_next_kernel), nothing else instrumented"""
        eng.kernel_timeline = (set(tags), [])
        n_steps = args.profile_steps if args.config != 5 else len(resident)
        for _ in range(n_steps):
            step()
        torch.cuda.synchronize()
        per_tag = {}
        for tag, start, stop in eng.kernel_timeline[1]:
            per_tag.setdefault(tag, []).append(start.elapsed_time(stop))
        eng.kernel_timeline = None
        return {tag: float(np.mean(v)) for tag, v in per_tag.items()}

    roof_tags = [t for t in live_ms if (t.startswith("wgrad:") and t not in ("wgrad:output_conv", "wgrad:striding_conv"))
                 or t == "fwd:big_conv_1"]
    kernel_ms = kernel_pass(roof_tags) if args.config in (2, 3) else {}
    names = [s.name for s in specs]
    groups = {}
    for prefix in ("fwd", "dgrad", "wgrad"):
        ms = sum(v for t, v in live_ms.items() if t.startswith(prefix + ":"))
        if ms:
            groups[prefix] = {"ms_per_step": ms}
    for other in ("ctc", "softmax", "decode"):
        if other in live_ms:
            groups[other] = {"ms_per_step": live_ms[other]}
    groups["adam_and_repack"] = {"ms_per_step": sum(v for t, v in live_ms.items() if t.startswith("adam"))}
    groups["bias_grad"] = {"ms_per_step": sum(v for t, v in live_ms.items() if t.startswith("bgrad:")),
                           "note": "side stream, concurrent with the wgrad/dgrad kernels of the same layer: these "
                                   "durations are stretched by the overlap and are NOT additive with the other groups"}
    groups["per_launch_ms"] = {t: round(v, 4) for t, v in sorted(live_ms.items())}
    conv_ms = sum(groups[g]["ms_per_step"] for g in ("fwd", "dgrad", "wgrad") if g in groups)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    utt_per_s = batch_per_gpu * world * args.steps / elapsed
    step_tflops = flops_per_step * world * args.steps / elapsed / 1e12
    workloads = {
        3: "BASELINE config 3: Wav2Letter fwd+CTC+bwd+Adam step, random-init, 128-mel x 1000 frames, 32 "
           "utterances/GPU, labels U{20..200}, bf16 storage / fp32 accumulate / fp32 CTC",
        2: "BASELINE config 2: Wav2Letter forward + greedy CTC decode only, random-init, 128-mel x 1000 frames, 32 "
           "utterances/GPU, fp32 storage / exact-fp32 MFMA (the bit-exact-decode path); bf16 path in bf16_path",
        5: "BASELINE config 5: long-form fwd+CTC+bwd+Adam step, 257-bin power spectrograms, 8 utterances/GPU per step, "
           "T ~ U{2000..8000} frames in length-bucketed batches (64 utterances per GPU cycled), bf16 / fp32 CTC",
    }
    metrics = {3: "utterances/sec (fwd+bwd+CTC), 128-mel x 1000-frame batch",
               2: "utterances/sec (fwd-only + greedy decode), 128-mel x 1000-frame batch",
               5: "utterances/sec (fwd+bwd+CTC), 257-bin x 2000..8000-frame bucketed batches"}
    result = {
        "metric": metrics[args.config],
        "value": utt_per_s,
        "unit": "utterances/sec",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": main_dtype,
        "data": "synthetic",
        "config": {"workload": workloads[args.config], "global_batch": batch_per_gpu * world,
                   "frames": FRAMES if args.config != 5 else "2000..8000", "mel": bins,
                   "parallelism": "dp{}".format(world)},
        "final_mean_loss": final_loss,
        "frames_per_sec": frames_per_step * world * args.steps / elapsed,
        "step_mfma_frac": step_tflops / (peak_tflops * world),
        # the 1-D conv stack alone (north_star's 40 % target): algorithmic FLOPs of this rank's step over the summed
        # live durations of its forward / dgrad / wgrad launches
        "conv_stack_mfma_frac": (flops_per_step / 1e12) / (conv_ms * 1e-3) / peak_tflops,
        "kernels": groups,
    }
    if bf16_leg is not None:
        result["bf16_path"] = bf16_leg
    if dp is not None:
        result["data_parallel"] = dp
    if h2d is not None:
        step_ms = elapsed / args.steps * 1e3
        h2d["utterances_per_sec_including_h2d_serial"] = batch_per_gpu / ((step_ms + h2d["h2d_ms_per_step"]) * 1e-3)
        result["host_buffers"] = h2d
    if args.config == 3:
        # Dominant kernel (largest share of main-stream GPU time, profiles/r01k_kernel_stats.csv): wgrad_tn_ilv_kernel,
        # the 8-wave interleaved 256x256-tile weight-gradient kernel.  THREE launches per step use it (the library's
        # measured table picks it for big_conv_1, big_conv_2 and the grouped launch that covers the seven inner_conv_i);
        # algorithmic FLOPs per launch = (sum of those nine layers' wgrad FLOPs) / 3.
        dom_layers = ("big_conv_1", "big_conv_2") + tuple("inner_conv_{}".format(i) for i in range(1, 8))
        dom_tags = [t for t in live_ms if t.startswith("wgrad:") and
                    t not in ("wgrad:output_conv", "wgrad:striding_conv")]
        dom_flops = sum(fl[i] for i, n in enumerate(names) if n in dom_layers) * BATCH_PER_GPU / len(dom_tags)
        dom_ms = sum(kernel_ms[t] for t in dom_tags) / len(dom_tags)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        traffic, traffic_source = None, None
        for name in ("r02_pmc_traffic_wgrad_ilv.json", "r01k_pmc_traffic_wgrad_ilv.json"):
            pmc = ROOT / "profiles" / name
            if pmc.exists():  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh (gfx950 x2 correction)
                traffic = json.loads(pmc.read_text())["traffic_bytes_per_launch_avg"]
                traffic_source = "profiles/" + name
                break
        result["roofline"] = {
            "bound": "mfma", "kernel": "wgrad_tn_ilv_kernel (weight gradient of big_conv_1, big_conv_2 and the grouped "
                                       "inner_conv_1..7 launch; average over its {} launches per "
                                       "step)".format(len(dom_tags)),
            "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
            "traffic_source": traffic_source,
            "traffic_note": "NOT measured in this run (PMC counters need rocprofv3 around the process): bytes per launch "
                            "from the committed file named in traffic_source -- rocprofv3 --pmc FETCH_SIZE*2 + WRITE_SIZE "
                            "in separate passes (tools/pmc_traffic.sh), Infinity-Cache hits included, average of the "
                            "kernel's launches per step",
            "duration_note": "HIP events recorded by the library immediately around the kernel on its launch stream "
                             "(sl_profile_next_kernel) in otherwise un-instrumented steps; compare with the average "
                             "of wgrad_tn_ilv_kernel in profiles/r02k_kernel_stats.csv -- that average is SHORTER (177.6 us "
                             "= 0.55 of peak): the chip is power-limited under this step (1.2 kW, tools/power_probe.py) "
                             "and the profiler's gaps between kernels let every kernel run at a higher clock than in the "
                             "un-instrumented step measured here (DESIGN.md section 4)",
            "flops_per_launch": dom_flops, "avg_launch_ms": dom_ms}
    if args.config in (2, 3):
        nt_flops = fl[names.index("big_conv_1")] * BATCH_PER_GPU
        nt_ms = kernel_ms["fwd:big_conv_1"]
        nt_name = "conv_nt_slab_bf16_kernel<IT=8,WM=2,WN=4,STAGES=2|pipelined,BIAS_RELU,bf16,interleaved>" \
            if main_dtype == "bf16" else "conv_nt_f32_mfma_kernel<BIAS_RELU> (v_mfma_f32_32x32x2_f32, exact fp32)"
        nt = {"bound": "mfma", "kernel": nt_name + " (forward of big_conv_1)",
              "achieved": nt_flops / (nt_ms * 1e-3) / 1e12, "peak": peak_tflops, "unit": "TFLOP/s",
              "frac": nt_flops / (nt_ms * 1e-3) / 1e12 / peak_tflops, "traffic": None,
              "flops_per_launch": nt_flops, "avg_launch_ms": nt_ms}
        result["roofline_nt_256x256" if args.config == 3 else "roofline"] = nt
    if args.config == 5:
        # HBM-bound kernels of the step: output_conv (AI 29 flop/B) in all three passes, softmax,
        # CTC, bias gradients, Adam.  Algorithmic bytes: every operand / result of those kernels moved once.
        t_sum = frames_per_step / 2.0  # output frames per step (all utterances of a batch)
        t_pad = float(np.mean([len(tl) * (-(-int(tl.max()) // 2)) for (_, _, _, _, tl) in host_batches]))
        n_param = sum(s.kernel_size * s.cin * s.cout + s.cout for s in specs)
        ch = {s.name: s.cout for s in specs}
        # algorithmic bytes per launch tag (each operand / result once, padded frames included; bf16 activations)
        bytes_by_tag = {
            "fwd:output_conv": t_pad * (2000 * 2 + 2 * K_CLASSES * 4),          # activation in, probs + log q out
            "dgrad:output_conv": t_pad * (128 * 2 + 2000 * 2 + 2000 * 2),      # g in, mask in, g out
            "wgrad:output_conv": t_pad * (2000 * 2 + 128 * 2),                 # activation + g in (dW is 0.5 MB)
            "softmax": t_pad * K_CLASSES * 4 * 3,
            # CTC: log q + probs in; alpha and beta lattices written by the lattice kernel, read by the gradient kernel;
            # dL/dlogits out (bf16, 128 padded lanes)
            "ctc": t_pad * (K_CLASSES * 4 * 2 + 2 * 2 * 401 * 4 + 128 * 2),
        }
        for n in names:
            bytes_by_tag["bgrad:" + n] = t_pad * ch[n] * 2                     # each layer's g read once
        adam_tags = [t for t in live_ms if t.startswith("adam")]
        for t in adam_tags:                                                    # 4 fp32 reads, 3 fp32 + 2 bf16 writes
            bytes_by_tag[t] = n_param * 32.0 / len(adam_tags)
        per_kernel = []
        for tag in sorted(bytes_by_tag):
            if tag in live_ms and live_ms[tag] > 0:
                gbps = bytes_by_tag[tag] / (live_ms[tag] * 1e-3) / 1e9
                per_kernel.append({"launch": tag, "bytes": bytes_by_tag[tag], "ms": live_ms[tag], "GBps": gbps,
                                   "frac_of_8TBps": gbps / 8000.0})
        hbm_tags = [e["launch"] for e in per_kernel]
        hbm_bytes = sum(e["bytes"] for e in per_kernel)
        hbm_ms = sum(live_ms[t] for t in hbm_tags)
        result["roofline"] = {"bound": "hbm", "kernel": "the step's HBM-bound kernels together: output_conv fwd/dgrad/"
                                                        "wgrad, softmax, CTC lattice+gradient, bias gradients, Adam "
                                                        "(each one listed in per_kernel)",
                              "achieved": hbm_bytes / (hbm_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                              "frac": hbm_bytes / (hbm_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                              "bytes_per_step": hbm_bytes, "ms_per_step": hbm_ms, "per_kernel": per_kernel,
                              "note": "algorithmic bytes (each operand / result once, padded frames included) over "
                                      "the live durations of those launches, averaged over the bucketed batches; the "
                                      "CTC lattice is latency- not bandwidth-bound (up to 4000 sequential frames), "
                                      "which is what holds the group's number down"}
        result["padding_waste"] = waste
    if world == 1 and not args.no_cpu_baseline and args.config == 3:
        from oracle import w2l_oracle as o
        result["cpu_baseline"] = cpu_baseline(o.layer_specs(MEL, K_CLASSES), weights)
    else:
        result["cpu_baseline"] = None
    print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
