#!/usr/bin/env python
"""End-to-end utterances/s of Wav2Letter.train_on_batch on HOST batches (List[LabeledSpectrogram], float64 spectrograms
as the reference produces them): the reference's serial loop (pack -> pageable H2D -> step) against the staged pipeline
(speechless_amd/pipeline.py).  BASELINE config-3 shape: 32 utterances x 1000 frames x 128 mel.

    python tools/e2e_train_throughput.py [--steps 40]
"""
import argparse
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000, help="timed steps per repeat")
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--ragged", action="store_true",
                    help="utterances of 600..1000 frames: every batch is padded to its own longest member, as the "
                         "reference's generator does (corpus.py:224-226), so the frame count changes from step to step")
    ap.add_argument("--from-audio", action="store_true",
                    help="train from RAW AUDIO (8 s per utterance = 1001 frames): samples staged, front end on the copy "
                         "stream (pipeline.AudioBatchStager); also times the front end alone")
    ap.add_argument("--switch-interval", type=float, default=None, help="sys.setswitchinterval() for the run (experiment)")
    args = ap.parse_args()
    if args.switch_interval:
        sys.setswitchinterval(args.switch_interval)
    if args.from_audio:
        return from_audio(args)
    import torch
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import LabeledSpectrogram
    from speechless_amd.pipeline import BatchStager
    rng = np.random.RandomState(0)
    chars = "abcdefghijklmnopqrstuvwxyz '"
    pool = []
    for i in range(64):
        label = "".join(rng.choice(list(chars), size=rng.randint(20, 201))).strip() or "a"
        frames = int(rng.randint(600, 1001)) if args.ragged else 1000
        pool.append(LabeledSpectrogram(id=str(i), label=" ".join(label.split())[:frames // 5],
                                       spectrogram=rng.randn(frames, 128)))
    warm = 30  # staged steps consumed before the clock starts: the pipeline is full, every launch list recorded
    batches = [[pool[(j * 7 + i * (3 if args.ragged else 1)) % 64] for i in range(32)] for j in range(args.steps + warm)]
    net = Wav2Letter(128, english_frequent_characters, seed=0)
    eng = net.engine
    for b in batches[:4]:
        net.train_on_batch(b)
    torch.cuda.synchronize()

    def serial_run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for b in batches[:n]:
            loss = net.train_on_batch(b, lazy=True)
        float(loss.item())
        return 32 * n / (time.perf_counter() - t0)

    def staged_run(n):
        """utt/s over n steps after `warm` steps; also where the training thread's time went (waiting for a staged batch /
        enqueueing the step)"""
        stager = BatchStager(batches[:n + warm], net._pack_for_staging, net.engine.device,
                             blank=net.grapheme_encoding.grapheme_set_size - 1, depth=3, workers=3)
        it = iter(stager)
        for _ in range(warm):
            staged = next(it)
            loss = net.train_on_staged_batch(staged, stager)
        float(loss.item())  # the GPU has caught up: the clock starts with an empty queue and a full pipeline
        wait = enqueue = 0.0
        t0 = time.perf_counter()
        for _ in range(n):
            t1 = time.perf_counter()
            staged = next(it)
            t2 = time.perf_counter()
            loss = net.train_on_staged_batch(staged, stager)
            t3 = time.perf_counter()
            wait += t2 - t1
            enqueue += t3 - t2
        float(loss.item())
        dt = time.perf_counter() - t0
        stager.close()
        return 32 * n / dt, wait / n * 1e3, enqueue / n * 1e3

    def resident_run(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            eng.train_step_resident()
        torch.cuda.synchronize()
        return 32 * n / (time.perf_counter() - t0)

    def stats(v):
        v = sorted(v)
        return "min {:8.1f}  median {:8.1f}  max {:8.1f}".format(v[0], v[len(v) // 2], v[-1])
    res, ser, stg = [], [], []
    for _ in range(args.repeats):  # interleaved: resident, serial, staged see the same box state
        res.append(resident_run(args.steps))
        ser.append(serial_run(min(args.steps, 300)))
        stg.append(staged_run(args.steps))
    if args.ragged:
        print("ragged batches: padded lengths {} .. {} frames over the run, {} buffer set(s) allocated".format(
            min(max(e._spectrogram.shape[0] for e in b) for b in batches), max(max(e._spectrogram.shape[0] for e in b)
                                                                               for b in batches), len(eng._buffers)))
    print("{} repeats of {} timed steps each (staged: after {} untimed steps with the pipeline full); utt/s".format(
        args.repeats, args.steps, warm))
    print("resident input (the bench.py regime): {}".format(stats(res)))
    print("serial loop (host packing + pageable H2D on the critical path, {} steps): {}".format(min(args.steps, 300), stats(ser)))
    print("staged (worker threads, copy stream) : {}".format(stats([v[0] for v in stg])))
    med_res = sorted(res)[len(res) // 2]
    print("staged / resident, per repeat: {}  (median of staged / median of resident = {:.1f} %)".format(
        ", ".join("{:.1f} %".format(100 * s[0] / r) for s, r in zip(stg, res)),
        100 * sorted(v[0] for v in stg)[len(stg) // 2] / med_res))
    print("training thread per staged step: {:.3f} ms waiting for the next staged batch, {:.3f} ms enqueueing the step "
          "(the GPU step itself: {:.3f} ms)".format(float(np.median([v[1] for v in stg])), float(np.median([v[2] for v in stg])),
                                                    32e3 / med_res))
    print("HBM: {:.2f} GB allocated now, {:.2f} GB at the peak, {:.2f} GB reserved by the allocator".format(
        torch.cuda.memory_allocated() / 1e9, torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))


def from_audio(args):
    import torch
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.pipeline import AudioBatchStager
    from speechless_amd.spectrogram import LabeledExample
    rng = np.random.RandomState(0)
    chars = "abcdefghijklmnopqrstuvwxyz '"
    pool = []
    for i in range(64):
        label = "".join(rng.choice(list(chars), size=rng.randint(20, 201))).strip() or "a"
        audio = (0.1 * rng.randn(128000)).astype(np.float32)  # 8 s at 16 kHz -> 1001 frames
        pool.append(LabeledExample(lambda a=audio: a, id=str(i), label=" ".join(label.split())[:190]))
    batches = [[pool[(j * 7 + i) % 64] for i in range(32)] for j in range(args.steps + 40)]
    net = Wav2Letter(128, english_frequent_characters, seed=0)
    extractor = net._audio_extractor(pool[0])
    # the front end alone: 32 utterances of audio already in HBM -> (32, 1001, 128) z-normalised mel batch
    flat, offsets, lengths = extractor.flatten([e.get_raw_audio() for e in batches[0]])
    dev = [torch.from_numpy(a).cuda() for a in (flat, offsets, lengths)]
    for _ in range(3):
        extractor.batch_device(dev[0], dev[1], dev[2], lengths)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        extractor.batch_device(dev[0], dev[1], dev[2], lengths)
    torch.cuda.synchronize()
    front = (time.perf_counter() - t0) / 20

    def run(n, on_copy=True):
        stager = AudioBatchStager(batches[:n], net._pack_audio_for_staging, extractor,
                                  net.input_to_prediction_length_ratio, net.engine.device,
                                  blank=net.grapheme_encoding.grapheme_set_size - 1, depth=3, workers=3,
                                  front_end_on_copy_stream=on_copy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for staged in stager:
            loss = net.train_on_staged_batch(staged, stager)
        float(loss.item())
        dt = time.perf_counter() - t0
        stager.close()
        return dt
    warm = 30

    def timed_run(on_copy):
        n = args.steps
        stager = AudioBatchStager(batches[:n + warm], net._pack_audio_for_staging, extractor,
                                  net.input_to_prediction_length_ratio, net.engine.device,
                                  blank=net.grapheme_encoding.grapheme_set_size - 1, depth=3, workers=3,
                                  front_end_on_copy_stream=on_copy)
        it = iter(stager)
        for _ in range(warm):
            staged = next(it)
            loss = net.train_on_staged_batch(staged, stager)
        float(loss.item())
        t0 = time.perf_counter()
        for _ in range(n):
            staged = next(it)
            loss = net.train_on_staged_batch(staged, stager)
        float(loss.item())
        dt = time.perf_counter() - t0
        stager.close()
        return 32 * n / dt
    eng = net.engine
    run(6)

    def resident_run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step_resident()
        torch.cuda.synchronize()
        return 32 * args.steps / (time.perf_counter() - t0)

    def stats(v):
        v = sorted(v)
        return "min {:8.1f}  median {:8.1f}  max {:8.1f}".format(v[0], v[len(v) // 2], v[-1])
    res, copy, comp = [], [], []
    for _ in range(args.repeats):
        res.append(resident_run())
        copy.append(timed_run(True))
        comp.append(timed_run(False))
    med = sorted(res)[len(res) // 2]
    print("{} repeats of {} timed steps each (after {} untimed steps with the pipeline full); utt/s".format(
        args.repeats, args.steps, warm))
    print("resident input (the bench.py regime)                    : {}".format(stats(res)))
    print("from raw audio, staged (front end on the copy stream)   : {}  = {:.1f} % of resident (medians)".format(
        stats(copy), 100 * sorted(copy)[len(copy) // 2] / med))
    print("from raw audio, staged (front end on the COMPUTE stream): {}  = {:.1f} % of resident (medians)".format(
        stats(comp), 100 * sorted(comp)[len(comp) // 2] / med))
    print("front end alone (STFT + level + mel + z-norm, 32 x 8 s in HBM): {:.3f} ms per batch = {:.0f} utt/s = {:.1f} % of a "
          "training step; H2D of the samples: {:.1f} MB per batch".format(
              front * 1e3, 32 / front, 100 * front / (32 / med), flat.nbytes / 1e6))


if __name__ == "__main__":
    main()
