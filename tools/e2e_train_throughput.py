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
    ap.add_argument("--steps", type=int, default=200)
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
    batches = [[pool[(j * 7 + i * (3 if args.ragged else 1)) % 64] for i in range(32)] for j in range(args.steps + 8)]
    net = Wav2Letter(128, english_frequent_characters, seed=0)
    for b in batches[:4]:
        net.train_on_batch(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in batches[4:4 + args.steps]:
        loss = net.train_on_batch(b, lazy=True)
    float(loss.item())
    serial = time.perf_counter() - t0
    stager = BatchStager(batches[4:4 + args.steps], net._pack_for_staging, net.engine.device,
                         blank=net.grapheme_encoding.grapheme_set_size - 1, depth=3, workers=3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for staged in stager:
        loss = net.train_on_staged_batch(staged, stager)
    float(loss.item())
    piped = time.perf_counter() - t0
    stager.close()
    eng = net.engine
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step_resident()
    torch.cuda.synchronize()
    resident = time.perf_counter() - t0
    if args.ragged:
        print("ragged batches: padded lengths {} .. {} frames over the run, {} buffer set(s) allocated".format(
            min(max(e._spectrogram.shape[0] for e in b) for b in batches), max(max(e._spectrogram.shape[0] for e in b)
                                                                               for b in batches), len(eng._buffers)))
    print("resident input (the bench.py regime): {:8.1f} utt/s ({:.2f} ms per batch)".format(
        32 * args.steps / resident, resident / args.steps * 1e3))
    print("serial loop   : {:8.1f} utt/s ({:.2f} ms per batch of 32, host packing + pageable H2D on the critical path)".format(
        32 * args.steps / serial, serial / args.steps * 1e3))
    print("staged (worker thread, copy stream): {:8.1f} utt/s ({:.2f} ms per batch)".format(
        32 * args.steps / piped, piped / args.steps * 1e3))
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
    batches = [[pool[(j * 7 + i) % 64] for i in range(32)] for j in range(args.steps + 8)]
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
    run(6)
    piped = run(args.steps)
    run(6, False)
    piped_c = run(args.steps, False)
    eng = net.engine
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step_resident()
    torch.cuda.synchronize()
    resident = time.perf_counter() - t0
    print("resident input (the bench.py regime)        : {:8.1f} utt/s ({:.2f} ms per batch)".format(
        32 * args.steps / resident, resident / args.steps * 1e3))
    print("from raw audio, staged (front end on the copy stream): {:8.1f} utt/s ({:.2f} ms per batch) = {:.1f} % of resident".format(
        32 * args.steps / piped, piped / args.steps * 1e3, 100 * resident / piped))
    print("from raw audio, staged (front end on the COMPUTE stream) : {:8.1f} utt/s ({:.2f} ms per batch) = {:.1f} % of resident".format(
        32 * args.steps / piped_c, piped_c / args.steps * 1e3, 100 * resident / piped_c))
    print("front end alone (STFT + level + mel + z-norm, 32 x 8 s in HBM): {:.3f} ms per batch = {:.0f} utt/s = {:.1f} % of a "
          "training step; H2D of the samples: {:.1f} MB per batch".format(
              front * 1e3, 32 / front, 100 * front / (resident / args.steps), flat.nbytes / 1e6))


if __name__ == "__main__":
    main()
