#!/usr/bin/env python
"""bench.py -- utterances/sec of the Wav2Letter training hot path (fwd + CTC + bwd + Adam) on N MI355X.

Workload = BASELINE.json config 3 per GPU: synthetic 128-mel x 1000-frame batch of 32 utterances, labels
L_i ~ U{20..200} over the 28 graphemes, random-init (glorot) Wav2Letter, bf16 activations/weights with fp32 accumulate
and fp32 CTC.  A "step" = one full optimisation step over one batch: 11 conv layers forward, softmax, CTC loss and
gradient, 10 dgrads + 11 wgrads + 11 bias grads, [gradient exchange over RCCL when N > 1], Adam, weight repack.
Inputs, labels and lengths are resident in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (it re-executes itself under
torch.distributed.run on 127.0.0.1); started under torch.distributed.run it is one of those ranks.

--config 2 / --config 5 run BASELINE.json's other single-GPU-shardable configurations with the same contract (they are
not the headline line): 2 = forward + greedy decode only on the same batch; 5 = long-form 257-bin power spectrograms, 8
utterances per GPU, lengths U{2000..8000} frames in length-bucketed batches, full training step, reported with the
achieved GB/s of its HBM-bound kernels next to the MFMA fraction.  The default invocation (config 3, one GPU) also runs
short versions of both and attaches them as `also.config2` / `also.config5`, and compares the GPU step with the CPU
step of the `cpu_baseline` leg at the benchmark's own batch size (`parity`).

Rank 0 prints ONE JSON line (see the contract in the task statement) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MEL, FRAMES, BATCH_PER_GPU, K_CLASSES = 128, 1000, 32, 29
BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
F32_MFMA_PEAK_TFLOPS = 157.3     # same guide: v_mfma_f32_32x32x2_f32, exact fp32 = the fp32 vector peak
HBM_PEAK_GBPS = 8000.0           # same guide: HBM3E ~8 TB/s


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


def long_form_steps(world):
    """Config 5 step formation: ONE global corpus of 64 x world utterances, T_i ~ U{2000..8000}
    frames (seed 3), cut into steps of 8 x world utterances from one length bucket and dealt to the ranks round-robin in
    length order (speechless_amd.batching.steps_for_ranks -- the product's cross-rank balancing), every rank of a step
    zero-padding to the step's longest utterance (batching.step_pad_length) so that all ranks launch the same grids.
    Returns (steps, lengths, pad_lengths): steps[s][r] = utterance ids of rank r in step s."""
    from speechless_amd.batching import step_pad_length, steps_for_ranks
    lengths = np.random.RandomState(3).randint(2000, 8001, size=LONG_CORPUS * world)
    steps = steps_for_ranks(list(range(LONG_CORPUS * world)), LONG_BATCH, world, length_of=lambda i: int(lengths[i]),
                            shuffle=False)
    return steps, lengths, [step_pad_length(step, lambda i: int(lengths[i])) for step in steps]


def long_form_batches(rank, world=1):
    """Rank `rank`'s batches of the config-5 steps (long_form_steps): [(x (8, T_pad, 257) float32 zero padded, labels,
    label_lengths, prediction_lengths, true_lengths)], and the padding waste over all ranks.  An utterance's samples and
    labels depend on its global id only (seeds 5000 + id / 7000 + id), L_i <= min(200, T_i / 4)."""
    steps, lengths, pads = long_form_steps(world)
    out = []
    for step, t_pad in zip(steps, pads):
        members = step[rank]
        x = np.zeros((len(members), t_pad, LONG_BINS), dtype=np.float32)
        lab_len = np.zeros(len(members), dtype=np.int32)
        rows = []
        for j, i in enumerate(members):
            x[j, :lengths[i]] = np.random.RandomState(5000 + i).randn(int(lengths[i]), LONG_BINS)
            rng = np.random.RandomState(7000 + i)
            lab_len[j] = rng.randint(20, min(200, int(lengths[i]) // 4) + 1)
            rows.append(rng.randint(0, K_CLASSES - 1, size=lab_len[j]))
        labels = -np.ones((len(members), int(lab_len.max())), dtype=np.int32)
        for j, row in enumerate(rows):
            labels[j, :len(row)] = row
        true_len = np.array([int(lengths[i]) for i in members], dtype=np.int32)
        out.append((x, labels, lab_len, (true_len // 2).astype(np.int32), true_len))
    # waste as convolved: every rank's batch padded to the step's longest utterance
    real = float(sum(int(lengths[i]) for step in steps for b in step for i in b))
    padded = float(sum(t_pad * sum(len(b) for b in step) for step, t_pad in zip(steps, pads)))
    return out, 1.0 - real / padded


def cpu_model_string():
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines():
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(specs_oracle, weights, sample_utts=BATCH_PER_GPU, steps=3):
    """The torch-CPU fp32 port of the same step (oracle/w2l_torch_cpu.py, Keras-form Adam) timed on this node's host
    cores, on the SAME batch as the GPU step (configuration 3, 32 utterances): 1 warm-up + `steps` timed steps (SURVEY.md
    section 8d).  The warm-up step starts from the benchmark's initial weights; what it computed is returned as the
    checker of the `parity` object."""
    import torch
    from oracle import w2l_torch_cpu as tc
    x, labels, lab_len, pred_len = synthetic_batch(0, sample_utts)
    times, first = tc.timed_training_steps(specs_oracle, weights, x, labels, pred_len, lab_len, steps=steps, warmup=1,
                                           record_first=True)
    mean, best = float(np.mean(times)), float(min(times))
    return {"value": sample_utts / mean, "unit": "utterances/sec", "cores": int(torch.get_num_threads()),
            "kind": "port", "host_cpus": os.cpu_count(), "cpu_model": cpu_model_string(),
            "value_best_step": sample_utts / best, "step_seconds": [round(t, 3) for t in times],
            "sample": "{} utterances x {} frames (the GPU step's batch), 1 warm-up + {} timed fwd+CTC+bwd+Adam steps, "
                      "torch-CPU fp32, mean step {:.2f} s, best {:.2f} s".format(sample_utts, FRAMES, steps, mean, best)}, first


def _rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, dtype=np.float64) - b) / max(np.linalg.norm(b), 1e-30))


def gpu_first_step(specs, weights, dtype, device):
    """What the HIP path computes from the benchmark's INITIAL weights on rank 0's batch: per-utterance losses, all
    gradients, frame argmax / greedy decode, the ReLU decisions of every hidden layer (bool tensors in HBM) and the
    weights after one Adam step."""
    import torch
    from speechless_amd.engine import HALO, Engine
    x, labels, lab_len, pred_len = synthetic_batch(0, BATCH_PER_GPU)
    eng = Engine(specs, K_CLASSES, dtype=dtype, device=device)
    eng.set_weights(weights)
    eng.load_input(torch.from_numpy(x).to(device))
    eng.set_labels(labels, lab_len, pred_len)
    eng.forward(training=True)
    losses = eng.ctc()
    eng.backward()
    torch.cuda.synchronize()
    out = {"losses": losses.cpu().numpy().copy(), "grads": eng.get_gradients()}
    buf = eng.cur
    # stored activation > 0 == pre-activation > 0 (bf16x3: the hi plane carries the sign)
    out["masks"] = [(buf.y[i][:, HALO:HALO + buf.t_out, :s.cout] > 0) for i, s in enumerate(specs[:-1])]
    out["decoded"], out["argmax"] = eng.greedy_decode(pred_len)
    eng.adam_step()
    torch.cuda.synchronize()
    out["weights_after"] = eng.get_weights()
    return out


def _mask_flips(a, b):
    """per hidden layer: how many ReLU decisions differ between two runs (bool tensors, any device)"""
    return [int((x.to(y.device) != y).sum().item()) for x, y in zip(a, b)]


def parity_object(specs, weights, names, cpu_first, device):
    """The HIP paths (bf16 = benchmarked, bf16x3 / f32 = parity paths) against the CPU step of the cpu_baseline leg AND,
    since round 4, against a float64 run of the same step (oracle/w2l_float64.py: per-tap dgemm on the GPU as the
    checker): same batch (32 x 1000 frames), same initial weights.  Two float32 implementations can only be compared up
    to their common float32 noise; against float64 each one is measured by itself -- `torch_cpu_f32.vs_f64` is the
    reference side's own distance from the exact result, the yardstick for the HIP parity paths' `vs_f64`
    (north_star: loss / gradients within 1e-3, bit-exact greedy decode)."""
    from oracle import w2l_float64 as f64
    from oracle import w2l_oracle as o
    x, labels, lab_len, pred_len = synthetic_batch(0, BATCH_PER_GPU)
    ref_losses = cpu_first["losses"].astype(np.float64)
    ref_argmax = cpu_first["probs"].argmax(axis=2)
    ref_decoded = o.greedy_decode_indices(cpu_first["probs"], pred_len)
    t0 = time.perf_counter()
    ospecs = o.layer_specs(MEL, K_CLASSES)
    exact = f64.loss_and_gradients(ospecs, weights, x, labels, pred_len, lab_len, device=device)
    f64_seconds = time.perf_counter() - t0
    exact_argmax = exact["probs"].argmax(axis=2)
    exact_decoded = o.greedy_decode_indices(exact["probs"], pred_len)
    hidden = names[:-1]

    def against_f64(losses, grads, masks, argmax, decoded):
        return {"loss_rel_max": float(np.max(np.abs(losses - exact["losses"]) / np.abs(exact["losses"]))),
                "grad_rel_l2": {n: _rel_l2(gw, rw) for n, (gw, _), (rw, _) in zip(names, grads, exact["grads"])},
                "bias_grad_rel_l2_max": max(_rel_l2(gb, rb) for (_, gb), (_, rb) in zip(grads, exact["grads"])),
                "relu_decisions_differing": dict(zip(hidden, _mask_flips(masks, exact["masks"]))),
                "argmax_agreement": float(np.mean(argmax == exact_argmax)),
                "greedy_decode_equal": bool(decoded == exact_decoded)}

    out = {"checker": "first CPU step of the cpu_baseline leg (torch-CPU fp32, oracle/w2l_torch_cpu.py): the GPU step's own "
                      "batch of {} x {} frames from the same initial weights".format(BATCH_PER_GPU, FRAMES),
           "checker_f64": "the same step in float64 (oracle/w2l_float64.py: per-tap dgemm on the GPU, CTC on the host), "
                          "{:.1f} s".format(f64_seconds),
           "relu_decisions_per_layer": {n: int(m.numel()) for n, m in zip(hidden, exact["masks"])}}
    out["torch_cpu_f32"] = {"vs_f64": against_f64(ref_losses, cpu_first["grads"], cpu_first["masks"], ref_argmax,
                                                  ref_decoded)}
    for dtype in ("bf16", "bf16x3", "f16x3", "f32"):
        g = gpu_first_step(specs, weights, dtype, device)
        leg = {"loss_rel_max": float(np.max(np.abs(g["losses"] - ref_losses) / np.abs(ref_losses))),
               "grad_rel_l2": {n: _rel_l2(gw, rw) for n, (gw, _), (rw, _) in zip(names, g["grads"], cpu_first["grads"])},
               "bias_grad_rel_l2_max": max(_rel_l2(gb, rb) for (_, gb), (_, rb) in zip(g["grads"], cpu_first["grads"])),
               "argmax_agreement": float(np.mean(g["argmax"] == ref_argmax)),
               "greedy_decode_equal": bool(g["decoded"] == ref_decoded),
               "sequences_differing": int(sum(a != b for a, b in zip(g["decoded"], ref_decoded))),
               "relu_decisions_differing_vs_torch_cpu_f32": dict(zip(hidden, _mask_flips(g["masks"], cpu_first["masks"])))}
        # the optimizer: weight change of one Adam(1e-4) step against the CPU step's (Keras form on both sides); Adam
        # normalises every element's first update to +-lr, so this counts the gradient SIGNS that differ
        num = sum(float(np.sum((ga.astype(np.float64) - ra) ** 2)) for (ga, _), (ra, _) in
                  zip(g["weights_after"], cpu_first["weights_after"]))
        den = sum(float(np.sum((ra.astype(np.float64) - w0) ** 2)) for (ra, _), (w0, _) in
                  zip(cpu_first["weights_after"], weights))
        leg["first_adam_update_rel_l2"] = float(np.sqrt(num / max(den, 1e-30)))
        leg["vs_f64"] = against_f64(g["losses"].astype(np.float64), g["grads"], g["masks"], g["argmax"], g["decoded"])
        out[dtype] = leg
        del g
    out["note"] = ("bf16 = the benchmarked path (bf16 storage, fp32 accumulate): the loss meets north_star's 1e-3, its "
                   "gradients carry the ReLU sign flips of bf16-rounded activations (DESIGN.md section 1); f32 = the "
                   "parity path (exact-fp32 MFMA), the one held to bit-exact decode and 1e-3 gradients; bf16x3 = the fast "
                   "parity path (hi + lo bf16 planes, three bf16 MFMA terms per product); f16x3 = the same scheme on fp16 "
                   "planes (22 operand bits instead of 16-17, power-of-two scales for range).  vs_f64: each implementation "
                   "against the float64 run -- torch_cpu_f32.vs_f64 is how far the REFERENCE side's float32 arithmetic "
                   "is from exact on this batch; a HIP parity path at or below that level is as right as the CPU path")
    return out


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks ourselves (one process per GPU on this
    node, rendezvous on 127.0.0.1) and hand their output through -- rank 0 prints the one JSON line."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this stack
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + argv
    return subprocess.call(cmd, env=env)


class Bench:
    """One rank's measurement of one BASELINE configuration."""

    def __init__(self, config, args, world, rank, device, reducer_factory=None, dtype=None):
        import torch
        from speechless_amd.engine import Engine, wav2letter_layer_specs
        from speechless_amd.net import Wav2Letter
        self.torch = torch
        self.config, self.args, self.world, self.rank, self.device = config, args, world, rank, device
        self.bins = LONG_BINS if config == 5 else MEL
        self.specs = wav2letter_layer_specs(self.bins, K_CLASSES)
        self.names = [s.name for s in self.specs]
        self.weights = Wav2Letter._glorot_uniform(self.specs, 2)  # Keras default init, same on every rank
        # configuration 2 asks for bit-exact decoded indices against the CPU path: its headline figure is the fp32 path
        # (fp32 storage, exact-fp32 MFMA); the bf16 path is timed next to it and its disagreements are counted
        self.dtype = dtype or ("f32" if config == 2 else "bf16")
        self.peak = F32_MFMA_PEAK_TFLOPS if self.dtype == "f32" else BF16_DENSE_PEAK_TFLOPS
        eng = self.eng = Engine(self.specs, K_CLASSES, dtype=self.dtype, device=device)
        eng.set_weights(self.weights)
        self.reducer = reducer_factory(eng) if (reducer_factory is not None and config != 2) else None
        self.waste = 0.0
        if config == 5:
            eng.max_cached_shapes = 16
            self.host_batches, self.waste = long_form_batches(rank, world)
            self.resident = [(torch.from_numpy(x).to(device), lab, ll, pl, tl) for (x, lab, ll, pl, tl) in self.host_batches]
            self.batch_per_gpu = LONG_BATCH
            self.cursor = 0
            self.frames_per_step = float(np.mean([tl.sum() for (_, _, _, _, tl) in self.host_batches]))
            self.flops_per_step = float(np.mean([
                sum(3 * sum(layer_flops_per_utt(self.specs, -(-int(t) // 2))) - layer_flops_per_utt(self.specs, -(-int(t) // 2))[0]
                    for t in tl) for (_, _, _, _, tl) in self.host_batches]))
        else:
            self.x, labels, lab_len, self.pred_len = synthetic_batch(rank, BATCH_PER_GPU)
            eng.load_input(torch.from_numpy(self.x).to(device))
            eng.set_labels(labels, lab_len, self.pred_len)
            self.batch_per_gpu = BATCH_PER_GPU
            self.fl = layer_flops_per_utt(self.specs, FRAMES // 2)
            self.flops_per_step = ((3 * sum(self.fl) - self.fl[0]) if config == 3 else sum(self.fl)) * BATCH_PER_GPU
            self.frames_per_step = float(BATCH_PER_GPU * FRAMES)

    def step(self):
        eng = self.eng
        if self.config == 5:
            x_dev, lab, ll, pl, _ = self.resident[self.cursor % len(self.resident)]
            self.cursor += 1
            eng.load_input(x_dev)  # fp32 -> bf16 halo'd layout on the GPU (sl_pack_input), part of the step
            eng.set_labels(lab, ll, pl)
            return eng.train_step_resident(self.reducer)
        if self.config == 3:
            return eng.train_step_resident(self.reducer)
        eng.forward()
        return eng.greedy_decode()

    def sync(self):
        self.torch.cuda.synchronize()
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            self.torch.cuda.synchronize()

    def timed(self, steps, warmup, step=None):
        """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; max over ranks (seconds)."""
        step = step or self.step
        out = None
        for _ in range(warmup):
            out = step()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            import torch.distributed as dist
            t = self.torch.tensor([elapsed], dtype=self.torch.float64, device=self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, out

    # ---- roofline legs: a few more steps with HIP events (same stream as the kernels)
    def timeline_pass(self):
        """ms per step and launch tag, events around every C-ABI call (the side-stream bias pass overlaps as in the step)"""
        eng = self.eng
        for _ in range(5 if self.config != 5 else 0):  # (lists recorded, chip in the state of the timed region)
            self.step()
        eng.timeline = []
        n_steps = max(10, self.args.profile_steps) if self.config != 5 else len(self.resident)
        # per-launch durations are taken from the whole-batch sequence: where Engine.split_top runs the CTC of one part of the
        # batch under the top layers of the other, both sides' event-bracketed durations are stretched by the overlap
        split, eng.split_top = eng.split_top, False
        for _ in range(n_steps):
            self.step()
        self.torch.cuda.synchronize()
        eng.split_top = split
        per_tag = {}
        for tag, start, stop in eng.timeline:
            per_tag.setdefault(tag, []).append(start.elapsed_time(stop))
        eng.timeline = None
        return {tag: float(np.sum(v)) / n_steps for tag, v in per_tag.items()}

    def kernel_pass(self, tags):
        """ms per step of the MAIN kernel(s) behind each tag: events recorded by the library immediately around that
        kernel on its stream (sl_profile_next_kernel), nothing else instrumented"""
        eng = self.eng
        # the timeline pass before this one launched eagerly (gaps between the kernels, the chip clocks down in them): a few
        # un-instrumented steps from the recorded lists bring the chip back to the state of the timed region first
        for _ in range(10):
            self.step()
        eng.kernel_timeline = (set(tags), [])
        n_steps = max(10, self.args.profile_steps) if self.config != 5 else len(self.resident)
        for _ in range(n_steps):
            self.step()
        self.torch.cuda.synchronize()
        per_tag = {}
        for tag, start, stop in eng.kernel_timeline[1]:
            per_tag.setdefault(tag, []).append(start.elapsed_time(stop))
        eng.kernel_timeline = None
        # per STEP: the top layers' forward / input-gradient launches come in two half-batch launches per step (Engine.split_top)
        return {tag: float(np.sum(v)) / n_steps for tag, v in per_tag.items()}

    def other_forward_leg(self, dtype):
        """config 2: another storage scheme timed beside the fp32 path, its disagreements with it counted.  bf16 = the
        training path's storage; bf16x3 = hi + lo planes, three bf16 MFMA terms per product (the fast parity path)."""
        from speechless_amd.engine import Engine
        torch = self.torch
        eng16 = Engine(self.specs, K_CLASSES, dtype=dtype, device=self.device)
        eng16.set_weights(self.weights)
        eng16.load_input(torch.from_numpy(self.x).to(self.device))
        eng16.set_input_lengths(self.pred_len)
        self.eng.set_input_lengths(self.pred_len)

        def step16():
            eng16.forward()
            return eng16.greedy_decode()
        el16, out16 = self.timed(self.args_steps, self.args_warmup, step16)
        decoded32, argmax32 = self.eng.greedy_decode()
        decoded16, argmax16 = out16
        notes = {"bf16": "bf16 storage is NOT bit-exact against the fp32 CPU path at random init (near-flat softmax): "
                         "the headline value above is the fp32 path, whose decoded indices are "
                         "(tests/test_gpu_round2.py::test_config2_greedy_decode_bit_exact_at_batch_32)",
                 "bf16x3": "every value as hi + lo bf16 planes, three bf16 MFMA terms per product, fp32 accumulate "
                           "(speechless_amd/csrc/split3.hip): decoded indices bit-exact against the torch-CPU fp32 path "
                           "(tests/test_gpu_round3.py::test_bf16x3_config2_greedy_decode_bit_exact_at_batch_32)"}
        del eng16
        return {"dtype": dtype, "value": self.batch_per_gpu * self.world * self.args_steps / el16, "unit": "utterances/sec",
                "ms_per_step": el16 / self.args_steps * 1e3,
                "mismatching_frames_vs_f32": int((argmax16 != argmax32).sum()), "frames": int(argmax32.size),
                "mismatching_sequences_vs_f32": int(sum(a != b for a, b in zip(decoded16, decoded32))),
                "sequences": len(decoded32), "note": notes[dtype]}

    def run(self, steps, warmup):
        """Times the configuration and assembles its part of the JSON line (every rank runs it; rank 0 keeps it)."""
        self.args_steps, self.args_warmup = steps, warmup
        if self.config == 5:
            # one untimed pass over every length bucket before the W warm-up steps: a bucket's first step allocates its
            # buffer set and records its launch lists (tens of ms), which W = 5 warm-up steps over 8 buckets would leave
            # inside the timed region
            for _ in range(len(self.resident)):
                self.step()
            self.cursor = 0
        elapsed, out = self.timed(steps, warmup)
        world, config = self.world, self.config
        result = {
            "value": self.batch_per_gpu * world * steps / elapsed, "unit": "utterances/sec",
            "ms_per_step": elapsed / steps * 1e3, "dtype": self.dtype,
            "final_mean_loss": float(out.mean().item()) if config != 2 else None,
            "frames_per_sec": self.frames_per_step * world * steps / elapsed,
            "step_mfma_frac": self.flops_per_step * world * steps / elapsed / 1e12 / (self.peak * world),
        }
        if config == 2:
            result["bf16x3_path"] = self.other_forward_leg("bf16x3")
            result["bf16_path"] = self.other_forward_leg("bf16")
        self.elapsed = elapsed
        return result

    def kernel_report(self, result):
        """per-launch durations, groups and the roofline object(s) of this configuration"""
        config = self.config
        live_ms = self.live_ms = self.timeline_pass()
        roof_tags = [t for t in live_ms if (t.startswith("wgrad:") and t not in ("wgrad:output_conv", "wgrad:striding_conv"))
                     or t == "fwd:big_conv_1"]
        kernel_ms = self.kernel_pass(roof_tags) if config in (2, 3) else {}
        groups = {}
        for prefix in ("fwd", "dgrad", "wgrad", "bwd"):  # bwd: = weight and input gradient in one launch (output_conv)
            ms = sum(v for t, v in live_ms.items() if t.startswith(prefix + ":"))
            if ms:
                groups[prefix] = {"ms_per_step": ms}
        for other in ("ctc", "softmax", "decode"):
            if other in live_ms:
                groups[other] = {"ms_per_step": live_ms[other]}
        if config == 5:
            groups["note"] = ("per-launch durations of the WHOLE-BATCH sequence; in the timed region Engine.split_top runs the CTC "
                              "of one part of the batch on a side stream under the top layers of the other part where that takes "
                              "no more rounds of 256 work-groups (the 384 / 640 / 896 / 960-tile buckets): ms_per_step is "
                              "smaller than the sum of these groups by the CTC time hidden there")
        groups["adam_and_repack"] = {"ms_per_step": sum(v for t, v in live_ms.items() if t.startswith("adam"))}
        groups["bias_grad"] = {"ms_per_step": sum(v for t, v in live_ms.items() if t.startswith("bgrad:")),
                               "note": "side stream, concurrent with the wgrad/dgrad kernels of the same layer: these "
                                       "durations are stretched by the overlap and are NOT additive with the other groups"}
        groups["per_launch_ms"] = {t: round(v, 4) for t, v in sorted(live_ms.items())}
        conv_ms = sum(groups[g]["ms_per_step"] for g in ("fwd", "dgrad", "wgrad", "bwd") if g in groups)
        # the 1-D conv stack alone (north_star's 40 % target): algorithmic FLOPs of this rank's step over the summed
        # live durations of its forward / dgrad / wgrad launches
        result["conv_stack_mfma_frac"] = (self.flops_per_step / 1e12) / (conv_ms * 1e-3) / self.peak
        result["kernels"] = groups
        if config == 3:
            result["roofline"] = self.roofline_wgrad(live_ms, kernel_ms)
        if config in (2, 3):
            nt_flops = self.fl[self.names.index("big_conv_1")] * BATCH_PER_GPU
            nt_ms = kernel_ms["fwd:big_conv_1"]
            nt_name = "conv_nt_slab_bf16_kernel<IT=8,WM=2,WN=4,STAGES=2|pipelined,BIAS_RELU,bf16,interleaved>" \
                if self.dtype == "bf16" else "conv_nt_f32_mfma_kernel<BIAS_RELU> (v_mfma_f32_32x32x2_f32, exact fp32)"
            nt = {"bound": "mfma", "kernel": nt_name + " (forward of big_conv_1)",
                  "achieved": nt_flops / (nt_ms * 1e-3) / 1e12, "peak": self.peak, "unit": "TFLOP/s",
                  "frac": nt_flops / (nt_ms * 1e-3) / 1e12 / self.peak, "traffic": None,
                  "flops_per_launch": nt_flops, "avg_launch_ms": nt_ms}
            result["roofline_nt_256x256" if config == 3 else "roofline"] = nt
        if config == 5:
            result["roofline"] = self.roofline_hbm_group(live_ms)
            result["padding_waste"] = self.waste
        return result

    def roofline_wgrad(self, live_ms, kernel_ms):
        """Dominant kernel (largest share of main-stream GPU time, profiles/*_kernel_stats.csv): wgrad_tn_ilv_kernel, the
        8-wave interleaved 256x256-tile weight-gradient kernel, TWO launches per step (big_conv_1, big_conv_2); its loop
        also runs the balanced launch for striding_conv + inner_conv_1..7 (wgrad_tn_ilv_multi_kernel), listed beside
        them in per_launch.  algorithmic FLOPs per launch = (sum of the two layers' wgrad FLOPs) / 2."""
        fl, names = self.fl, self.names
        tags = sorted(t for t in live_ms if t.startswith("wgrad:") and t in kernel_ms)
        per_launch = {}
        for t in tags:
            lo = t.split(":", 1)[1].split("..")
            layers = [lo[0]] if len(lo) == 1 else names[names.index(lo[0]): names.index(lo[1]) + 1]
            flops = sum(fl[names.index(n)] for n in layers) * BATCH_PER_GPU
            per_launch[t] = {"flops": flops, "ms": kernel_ms[t],
                             "frac": flops / (kernel_ms[t] * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS}
        dom_tags = [t for t in tags if t in ("wgrad:big_conv_1", "wgrad:big_conv_2")]
        dom_flops = sum(per_launch[t]["flops"] for t in dom_tags) / len(dom_tags)
        dom_ms = sum(kernel_ms[t] for t in dom_tags) / len(dom_tags)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        traffic, traffic_source, per_launch_traffic, traffic_age = None, None, None, None
        for name in sorted((ROOT / "profiles").glob("r*_pmc_traffic_wgrad_ilv.json"), reverse=True):
            pmc = json.loads(name.read_text())  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh
            per_launch_traffic = {k: v["traffic_bytes"] for k, v in pmc.get("launches", {}).items()}
            dom = [v for k, v in per_launch_traffic.items() if k in ("big_conv_1", "big_conv_2")]
            traffic = sum(dom) / len(dom) if dom else pmc["traffic_bytes_per_launch_avg"]
            traffic_source = "profiles/" + name.name
            # when the constant was measured (stamped into the file by tools/pmc_traffic.sh; there is no .git on the GPU
            # box to ask): a stale file shows as an old stamp
            traffic_age = pmc.get("measured")
            break
        rocprof_ms, rocprof_calls, rocprof_source = tracked_rocprof_average("wgrad_tn_ilv_kernel")
        alg_bytes = {"big_conv_1": BATCH_PER_GPU * (FRAMES // 2) * (256 + 2048) * 2 + 32 * 256 * 2048 * 4,
                     "big_conv_2": BATCH_PER_GPU * (FRAMES // 2) * (2048 + 2048) * 2 + 2048 * 2048 * 4}
        return {
            "bound": "mfma", "kernel": "wgrad_tn_ilv_kernel (weight gradient of big_conv_1 and big_conv_2; average over its {} "
                                       "launches per step)".format(len(dom_tags)),
            "achieved": achieved, "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / BF16_DENSE_PEAK_TFLOPS, "traffic": traffic,
            "traffic_source": traffic_source, "traffic_age": traffic_age, "traffic_per_launch": per_launch_traffic,
            "algorithmic_bytes_per_launch": dict(alg_bytes, note="padded operands read once (bf16) + the fp32 weight "
                                                                 "gradient written once"),
            "traffic_note": "NOT measured in this run (PMC counters need rocprofv3 around the process): bytes per launch "
                            "from the committed file named in traffic_source -- rocprofv3 --pmc FETCH_SIZE*2 + WRITE_SIZE "
                            "in separate passes (tools/pmc_traffic.sh), Infinity-Cache hits included, average of the "
                            "kernel's launches per step",
            "duration_source": "live: HIP events attached by the library to the kernel's own dispatch (sl_profile_next_kernel, "
                               "hipExtLaunchKernelGGL: the kernel's begin / end timestamps) in otherwise un-instrumented "
                               "steps of THIS run",
            "rocprof_avg_launch_ms": rocprof_ms, "rocprof_calls": rocprof_calls, "rocprof_source": rocprof_source,
            "frac_from_rocprof": (dom_flops / (rocprof_ms * 1e-3) / 1e12 / BF16_DENSE_PEAK_TFLOPS) if rocprof_ms else None,
            "duration_note": "avg_launch_ms / frac are measured live in this run; rocprof_avg_launch_ms is the same kernel's "
                             "average in the tracked rocprofv3 --kernel-trace --stats summary named in rocprof_source (a "
                             "different box and clock state: boxes differ by +-5 %, DESIGN.md section 4), and "
                             "frac_from_rocprof the fraction recomputed from it",
            "algorithmic_bytes": sum(alg_bytes.values()) / len(alg_bytes),
            "flops_per_launch": dom_flops, "avg_launch_ms": dom_ms, "per_launch": per_launch}

    def roofline_hbm_group(self, live_ms):
        """config 5: HBM-bound kernels of the step: output_conv (AI 29 flop/B) in all three passes,
        softmax, CTC, bias gradients, Adam.  Algorithmic bytes: every operand / result of those kernels moved once."""
        specs, names = self.specs, self.names
        t_pad = float(np.mean([len(tl) * (-(-int(tl.max()) // 2)) for (_, _, _, _, tl) in self.host_batches]))
        n_param = sum(s.kernel_size * s.cin * s.cout + s.cout for s in specs)
        ch = {s.name: s.cout for s in specs}
        # algorithmic bytes per launch tag (each operand / result once, padded frames included; bf16 activations)
        bytes_by_tag = {
            "fwd:output_conv": t_pad * (2000 * 2 + 2 * K_CLASSES * 4),          # activation in, probs + log q out
            "dgrad:output_conv": t_pad * (128 * 2 + 2000 * 2 + 2000 * 2),      # g in, mask in, g out
            "wgrad:output_conv": t_pad * (2000 * 2 + 128 * 2),                 # activation + g in (dW is 0.5 MB)
            "bwd:output_conv": t_pad * (128 * 2 + 2000 * 2 + 2000 * 2),        # fused dW + dx: x, g in once, g out
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
                                   "frac_of_8TBps": gbps / HBM_PEAK_GBPS})
        hbm_bytes = sum(e["bytes"] for e in per_kernel)
        hbm_ms = sum(e["ms"] for e in per_kernel)
        return {"bound": "hbm", "kernel": "the step's HBM-bound kernels together: output_conv fwd/dgrad/wgrad, softmax, CTC "
                                          "lattice+gradient, bias gradients, Adam (each one listed in per_kernel)",
                "achieved": hbm_bytes / (hbm_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": hbm_bytes / (hbm_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                "bytes_per_step": hbm_bytes, "ms_per_step": hbm_ms, "per_kernel": per_kernel,
                "note": "algorithmic bytes (each operand / result once, padded frames included) over the live durations of "
                        "those launches, averaged over the bucketed batches; the CTC lattice is latency- not "
                        "bandwidth-bound (up to 4000 sequential frames), which is what holds the group's number down"}

    def data_parallel_report(self):
        """N > 1: everything needed to read the first multi-GPU run from its one JSON line"""
        import torch.distributed as dist
        torch, eng, reducer, world, device = self.torch, self.eng, self.reducer, self.world, self.device

        def timed(n):
            self.sync()
            t1 = time.perf_counter()
            for _ in range(n):
                self.step()
            self.sync()
            dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=device)
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            return float(dt.item()) / n * 1e3
        # one more step, then: are the reduced gradients and the updated weights bit-identical on every rank?  (With the
        # sharded optimizer a rank holds the reduced gradient of its own slices only: the weights are what must agree.)
        self.step()
        torch.cuda.synchronize()
        sums = torch.stack([eng.grads.double().sum(), eng.grads.view(torch.int32).long().sum().double(),
                            eng.params.double().sum(), eng.params.view(torch.int32).long().sum().double()])
        gathered = [torch.zeros_like(sums) for _ in range(world)]
        dist.all_gather(gathered, sums)
        cols = (2, 3) if reducer.shard_optimizer else (0, 1, 2, 3)
        same = all(bool(torch.equal(g[list(cols)], gathered[0][list(cols)])) for g in gathered)
        # the exchange alone: all buckets back to back on the communication stream, nothing else running
        ranges = eng.bucket_ranges()
        bucket_bytes = [int((hi - lo) * 4) for lo, hi in ranges]
        scratch = torch.zeros_like(eng.grads)
        per_bucket_ms = []
        # (the shared-GPU test hook moves the bytes through gloo on the host: two repetitions there, not thirteen)
        warm_reps, timed_reps = (1, 1) if os.environ.get("SL_BENCH_SHARE_GPU") == "1" else (3, 10)
        for lo, hi in ranges:
            for _ in range(warm_reps):
                dist.all_reduce(scratch[lo:hi])
            self.sync()
            t1 = time.perf_counter()
            for _ in range(timed_reps):
                dist.all_reduce(scratch[lo:hi])
            self.sync()
            per_bucket_ms.append((time.perf_counter() - t1) / timed_reps * 1e3)
        del scratch
        ar_ms = sum(per_bucket_ms)
        alg = sum(bucket_bytes) / (ar_ms * 1e-3) / 1e9
        with_comm_ms = timed(self.args_steps)
        reducer.skip_collective = True  # same choreography, no bytes on the wire (timing only: weights now differ per rank)
        without_comm_ms = timed(self.args_steps)
        reducer.skip_collective = False
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:  # gloo test hook / CPU build
            rccl = None
        return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": rccl,
                "bucket_bytes": bucket_bytes, "bucket_layers": [[self.names[i] for i in layers]
                                                                 for layers, _ in eng.bucket_plan()],
                "sharded_optimizer": bool(reducer.shard_optimizer),
                "options": {"comm_cus": int(reducer.comm_cus), "compress": reducer.compress,
                            "split_last_bucket": bool(eng.split_last_bucket),
                            "rccl_env": {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}},
                "reduced_gradients_and_weights_identical_on_all_ranks": same,
                "gradient_checksum": float(gathered[0][0].item()), "weight_checksum": float(gathered[0][2].item()),
                "allreduce_alone_ms": ar_ms, "allreduce_alone_ms_per_bucket": per_bucket_ms,
                "allreduce_algbw_GBps": alg, "allreduce_busbw_GBps": alg * 2 * (world - 1) / world,
                "step_ms_with_allreduce": with_comm_ms, "step_ms_without_allreduce": without_comm_ms,
                "exposed_communication_ms": with_comm_ms - without_comm_ms,
                "note": "exposed communication = step time with the bucketed exchange minus the same step with the "
                        "collectives left out (streams and events unchanged); allreduce_alone = the buckets back to back "
                        "with nothing else on the GPU; busbw = algbw * 2(n-1)/n.  Buckets are listed in the order backward "
                        "completes them: only the last one (striding_conv, the last kernel of backward) has no cover"}

    def host_buffer_note(self, ms_per_step):
        """what handing over a HOST batch costs on top of the resident step (never part of `value`)"""
        torch = self.torch
        host = self.x  # (32, 1000, 128) float32, pageable -- what net.py:578-587 packs
        for _ in range(2):
            torch.from_numpy(host).to(self.device)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            dev = torch.from_numpy(host).to(self.device)
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - t1) / 10 * 1e3
        del dev
        return {"h2d_ms_per_step": h2d_ms, "bytes": int(host.nbytes), "GBps": host.nbytes / (h2d_ms * 1e-3) / 1e9,
                "utterances_per_sec_including_h2d_serial": self.batch_per_gpu / ((ms_per_step + h2d_ms) * 1e-3),
                "note": "pageable float32 host batch -> HBM (torch copy, synchronous); the fp32 -> bf16 halo'd repack on "
                        "the GPU (sl_pack_input) is not in the resident step either; serial upper bound of the "
                        "PCIe-inclusive rate, speechless_amd/pipeline.py overlaps the copy with the previous step "
                        "(profiles/*_e2e.txt)"}


def _sig(v, digits=5):
    """numbers of the compact line at `digits` significant figures"""
    if isinstance(v, float):
        return float("{:.{}g}".format(v, digits))
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    return v


def tracked_rocprof_average(kernel_substring, pattern="r*_kernel_stats.csv"):
    """(average launch ms, calls, file) of the kernel in the newest tracked rocprofv3 --kernel-trace --stats summary under
    profiles/ -- printed beside the live HIP-event duration so that `roofline.frac` can be recomputed from profiles/."""
    import csv
    import re
    # r05_kernel_stats.csv, r01k_kernel_stats.csv ... -- not the statistics of other configurations / paths (r05_kernel_stats_config5.csv,
    # r05_kernel_stats_bf16x3.csv)
    files = sorted((f for f in (ROOT / "profiles").glob(pattern) if re.fullmatch(r"r\d+[a-z]?_kernel_stats\.csv", f.name)),
                   reverse=True)
    for f in files:
        try:
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if kernel_substring in row["Name"] and "multi" not in row["Name"]:
                        return float(row["AverageNs"]) * 1e-6, int(row["Calls"]), "profiles/" + f.name
        except (OSError, KeyError, ValueError):
            continue
    return None, None, None


def _parity_summary(leg):
    """one path of the parity object in four numbers (vs the float64 run where there is one)"""
    src = leg.get("vs_f64", leg)
    grads = src["grad_rel_l2"]
    rest = [v for n, v in grads.items() if n != "striding_conv"]
    return {"loss_rel_max": src["loss_rel_max"], "grad_rel_l2_worst": max(grads.values()),
            "grad_rel_l2_worst_excl_striding_conv": max(rest) if rest else None,
            "decode_equal": bool(src["greedy_decode_equal"])}


def compact_line(detail, detail_path):
    """The LAST stdout line: the contract's keys + roofline + cpu_baseline + parity / also summaries in < 4 KB (the
    round-4 line had grown to 20 KB and the driver could not extract it).  Everything else is in `detail_path`."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "step_mfma_frac", "conv_stack_mfma_frac", "final_mean_loss")
    line = {k: detail[k] for k in keys if k in detail}
    roof = detail.get("roofline")
    if roof:
        keep = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_age",
                "algorithmic_bytes", "flops_per_launch", "avg_launch_ms", "duration_source", "rocprof_avg_launch_ms",
                "rocprof_source", "frac_from_rocprof", "bytes_per_step", "ms_per_step")
        line["roofline"] = {k: roof[k] for k in keep if k in roof}
    cpu = detail.get("cpu_baseline")
    line["cpu_baseline"] = None if not cpu else {k: cpu[k] for k in ("value", "unit", "cores", "kind", "cpu_model", "sample")}
    if "parity" in detail:
        line["parity"] = {path: _parity_summary(detail["parity"][path])
                          for path in ("torch_cpu_f32", "f32", "f16x3", "bf16x3", "bf16") if path in detail["parity"]}
        line["parity"]["checker"] = "float64 run of the same 32 x 1000 step (oracle/w2l_float64.py); rel-L2 per weight-gradient tensor"
    if "also" in detail:
        line["also"] = {}
        for name, leg in detail["also"].items():
            short = {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "dtype": leg["dtype"],
                     "step_mfma_frac": leg["step_mfma_frac"]}
            if "roofline" in leg:
                short["roofline_frac"] = leg["roofline"]["frac"]
                short["roofline_bound"] = leg["roofline"]["bound"]
            line["also"][name] = short
    if "data_parallel" in detail:
        dp = detail["data_parallel"]
        line["data_parallel"] = {k: dp[k] for k in (
            "world_size", "backend", "rccl_version", "bucket_bytes", "sharded_optimizer",
            "reduced_gradients_and_weights_identical_on_all_ranks", "gradient_checksum", "weight_checksum", "allreduce_alone_ms", "allreduce_busbw_GBps",
            "step_ms_with_allreduce", "step_ms_without_allreduce", "exposed_communication_ms") if k in dp}
        if "options" in dp:  # (the knobs of tools/first_dp_run.sh; the RCCL environment stays in the detail file)
            line["data_parallel"]["options"] = {k: v for k, v in dp["options"].items() if k != "rccl_env"}
    if "host_buffers" in detail:
        hb = detail["host_buffers"]
        line["host_buffers"] = {k: hb[k] for k in ("h2d_ms_per_step", "utterances_per_sec_including_h2d_serial")}
    line["detail"] = detail_path
    return _sig(line)


def emit(detail):
    """full detail -> gpurun_out/bench_detail*.json (and stderr-free: nothing but the one compact line goes to stdout)"""
    name = "bench_detail.json" if (detail["n_gpus"] == 1 and detail.get("_config_id") == 3) else \
        "bench_detail_config{}_n{}.json".format(detail.get("_config_id"), detail["n_gpus"])
    path = None
    try:
        out_dir = ROOT / "gpurun_out"
        out_dir.mkdir(exist_ok=True)
        (out_dir / name).write_text(json.dumps(detail, indent=1))
        path = "gpurun_out/" + name
    except OSError:
        pass
    print(json.dumps(compact_line(detail, path)), flush=True)


WORKLOADS = {
    3: "BASELINE config 3: Wav2Letter fwd+CTC+bwd+Adam step, random-init, 128-mel x 1000 frames, 32 "
       "utterances/GPU, labels U{20..200}, bf16 storage / fp32 accumulate / fp32 CTC",
    2: "BASELINE config 2: Wav2Letter forward + greedy CTC decode only, random-init, 128-mel x 1000 frames, 32 "
       "utterances/GPU, fp32 storage / exact-fp32 MFMA (the bit-exact-decode path); bf16 path in bf16_path",
    5: "BASELINE config 5: long-form fwd+CTC+bwd+Adam step, 257-bin power spectrograms, 8 utterances/GPU per step, "
       "T ~ U{2000..8000} frames in length-bucketed batches (64 utterances per GPU cycled), bf16 / fp32 CTC",
}
METRICS = {3: "utterances/sec (fwd+bwd+CTC), 128-mel x 1000-frame batch",
           2: "utterances/sec (fwd-only + greedy decode), 128-mel x 1000-frame batch",
           5: "utterances/sec (fwd+bwd+CTC), 257-bin x 2000..8000-frame bucketed batches"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)  # 2 ms steps: 0.8 s timed (the driver passes its own --steps)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5),
                    help="BASELINE.json configuration: 3 = headline training step (default), 2 = forward + greedy "
                         "decode only, 5 = long-form 257-bin x 2000..8000-frame bucketed batches")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU leg (and the parity object it checks)")
    ap.add_argument("--no-also", action="store_true", help="skip the short config-2 / config-5 legs of the default run")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="N > 1: reduce-scatter the gradient buckets, Adam on this rank's slice, all-gather the masters "
                         "(speechless_amd/parallel.py) instead of all-reduce + full Adam on every rank")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra event-instrumented steps for the roofline leg")
    # knobs of the data-parallel exchange for the first real N > 1 run (tools/first_dp_run.sh sweeps them; defaults = the
    # configuration every figure in DESIGN.md was taken with)
    ap.add_argument("--comm-cus", type=int, default=None,
                    help="N > 1: CUs the collectives are expected to own while a bucket is on the wire (0 = no hint; default "
                         "SL_COMM_CUS or 0): backward's grid choosers plan for the rest (sl_set_available_cus)")
    ap.add_argument("--compress", choices=("bf16",), default=None, help="N > 1: gradients travel as bf16 (half the bytes per link)")
    ap.add_argument("--split-last-bucket", action="store_true",
                    help="N > 1: close inner_conv_4..7's gradients as a bucket of their own (Engine.split_last_bucket: the "
                         "fused input-gradient launch and the balanced weight-gradient launch cut in two)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from speechless_amd.parallel import GradBucketReducer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus {} but the launcher started {} rank(s)".format(args.gpus, world))
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

    def reducer_factory(eng):
        if world == 1:
            return None
        if args.split_last_bucket:
            eng.split_last_bucket = True  # (before the bucket plan is read)
        return GradBucketReducer(eng.grads, eng.bucket_ranges(), shard_optimizer=args.shard_optimizer,
                                 compress=args.compress, comm_cus=args.comm_cus)

    bench = Bench(args.config, args, world, rank, device, reducer_factory)
    result = bench.run(args.steps, args.warmup)
    dp = bench.data_parallel_report() if bench.reducer is not None else None
    bench.kernel_report(result)
    h2d = bench.host_buffer_note(result["ms_per_step"]) if (args.config == 3 and rank == 0) else None

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    line = {
        "metric": METRICS[args.config], "value": result.pop("value"), "unit": result.pop("unit"), "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": result.pop("ms_per_step"),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": result.pop("dtype"),
        "data": "synthetic",
        "config": {"workload": WORKLOADS[args.config], "global_batch": bench.batch_per_gpu * world,
                   "frames": FRAMES if args.config != 5 else "2000..8000", "mel": bench.bins,
                   "parallelism": "dp{}".format(world)},
    }
    if args.config == 5:
        line["config"]["step_formation"] = ("one global corpus of 64 x N utterances (seed 3), steps of 8 x N from one "
                                            "length bucket dealt round-robin to the ranks, all ranks of a step padded to "
                                            "the step's longest utterance (batching.steps_for_ranks)")
    line.update(result)
    if dp is not None:
        line["data_parallel"] = dp
    if h2d is not None:
        line["host_buffers"] = h2d
    headline = world == 1 and args.config == 3
    # ---- the other single-GPU configurations, witnessed in the same run (short legs; not the headline)
    if headline and not args.no_also:
        also = {}
        del bench
        torch.cuda.empty_cache()
        for cfg, steps, warm in ((2, 10, 3), (5, 16, 8)):
            t0 = time.perf_counter()
            b = Bench(cfg, args, 1, 0, device)
            r = b.run(steps, warm)
            b.kernel_report(r)
            r.update({"metric": METRICS[cfg], "workload": WORKLOADS[cfg], "steps": steps, "warmup": warm,
                      "leg_seconds": round(time.perf_counter() - t0, 2)})
            also["config{}".format(cfg)] = r
            del b
            torch.cuda.empty_cache()
        # the parity-compliant training rate (gradients within 1e-3, `parity.bf16x3`): config 3 on the bf16x3 path
        t0 = time.perf_counter()
        b = Bench(3, args, 1, 0, device, dtype="bf16x3")
        r = b.run(10, 3)
        live = b.timeline_pass()
        r.update({"step_mfma_frac_executed": 3.0 * r["step_mfma_frac"],
                  "per_launch_ms": {t: round(v, 4) for t, v in sorted(live.items())},
                  "metric": METRICS[3], "steps": 10, "warmup": 3, "leg_seconds": round(time.perf_counter() - t0, 2),
                  "workload": "BASELINE config 3 on the bf16x3 path (every value as hi + lo bf16 planes, three bf16 MFMA "
                              "terms per product, fp32 accumulate: the training path whose gradients meet north_star's 1e-3, "
                              "see parity.bf16x3); step_mfma_frac counts ALGORITHMIC FLOPs, step_mfma_frac_executed the "
                              "3 x it issues"})
        also["config3_bf16x3"] = r
        del b
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        b = Bench(3, args, 1, 0, device, dtype="f16x3")
        r = b.run(10, 3)
        r.update({"step_mfma_frac_executed": 3.0 * r["step_mfma_frac"], "metric": METRICS[3], "steps": 10, "warmup": 3,
                  "leg_seconds": round(time.perf_counter() - t0, 2),
                  "workload": "BASELINE config 3 on the f16x3 path (hi + lo fp16 planes: 22 operand bits, three fp16 MFMA terms "
                              "per product, weights and gradients stored under power-of-two scales; see parity.f16x3)"})
        also["config3_f16x3"] = r
        del b
        torch.cuda.empty_cache()
        line["also"] = also
    # ---- CPU leg: the port timed on this node's host cores, and its first step as the parity checker
    line["cpu_baseline"] = None
    if headline and not args.no_cpu_baseline:
        from oracle import w2l_oracle as o
        from speechless_amd.engine import wav2letter_layer_specs
        from speechless_amd.net import Wav2Letter
        specs = wav2letter_layer_specs(MEL, K_CLASSES)
        weights = Wav2Letter._glorot_uniform(specs, 2)
        line["cpu_baseline"], cpu_first = cpu_baseline(o.layer_specs(MEL, K_CLASSES), weights)
        line["parity"] = parity_object(specs, weights, [s.name for s in specs], cpu_first, device)
    line["_config_id"] = args.config
    emit(line)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
