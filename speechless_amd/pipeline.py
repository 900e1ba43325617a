"""Host input pipeline of the training step: packing on a worker thread, H2D on a copy stream.

The reference packs every batch serially with numpy on the training thread (speechless/net.py:578-607, fed by
`fit_generator` with one worker, net.py:550) and hands pageable host arrays to the session.  Once the step itself takes
2.5 ms for 32 utterances that serial part (zero-pad 16 MB + pageable H2D) is several times longer than the step, so it
moves off the critical path:

    worker thread : next(batches) -> labels/lengths encoded, spectrograms converted (float64 -> float32) and zero-padded
                    in ONE pass into a host staging buffer -> H2D on a dedicated copy stream -> event
    training loop : waits for the event on the compute stream (no host sync), packs fp32 -> bf16 halo'd layout on the
                    GPU (sl_pack_input) and runs the step while the worker stages the following batches.

Slots are recycled only after (a) their H2D copy finished and (b) the step that read the device tensor has issued its
pack kernel (event recorded on the compute stream), so `depth` batches can be in flight.

The staging buffer is deliberately PAGEABLE: on this platform CPU stores into hipHostMalloc'ed (pinned, coherent) memory
drop to ~3 GB/s while the GPU is busy (5.5 ms to fill 16 MB, against 0.4 ms idle and 0.7 ms for pageable memory), and
the pageable H2D costs the worker 0.7 ms instead of 0.3 -- measured with tools/e2e_train_throughput.py.
"""
import ctypes
import threading
from pathlib import Path

import numpy as np
import torch

from ._host_lib import host_lib  # noqa: E402  (libspeechless_host.so, include/speechless_host.h)


def pack_spectrograms(spectrograms, dst, n_threads=4):
    """Converts + zero-pads a list of (T_i, F) float64/float32 arrays into dst, a C-contiguous (B, Tmax, F) float32
    numpy array (reference net.py:583-586), in native code."""
    b, t_max, f = dst.shape
    dtype = spectrograms[0].dtype
    if dtype not in (np.float64, np.float32) or any(s.dtype != dtype for s in spectrograms):
        dtype = np.float64
    if len(spectrograms) != b:
        raise ValueError("{} spectrograms for a staging buffer of {} rows".format(len(spectrograms), b))
    for i, a in enumerate(spectrograms):  # the native packer reads lengths[i] * f elements through a raw pointer
        if a.ndim != 2 or a.shape[1] != f or a.shape[0] > t_max:
            raise ValueError("spectrogram {} has shape {}; expected (T <= {}, {})".format(i, a.shape, t_max, f))
    arrays = [np.ascontiguousarray(s, dtype=dtype) for s in spectrograms]  # no copy when already in that form
    ptrs = (ctypes.c_void_p * b)(*[a.ctypes.data for a in arrays])
    lengths = (ctypes.c_int32 * b)(*[a.shape[0] for a in arrays])
    rc = host_lib().sl_host_pack_batch(ptrs, lengths, b, f, t_max, 1 if dtype == np.float64 else 0,
                                       dst.ctypes.data, n_threads)
    if rc != 0:
        raise ValueError("sl_host_pack_batch rejected the batch (lengths / shapes)")


class StagedBatch:
    """One batch resident (or arriving) in HBM.  `ready` is recorded on the copy stream behind the H2D copy."""

    def __init__(self, slot, x_dev, labels_dev, label_len_dev, pred_len_dev, ready):
        self.slot = slot
        self.x_dev = x_dev
        self.labels_dev = labels_dev        # int32 (B, Lmax >= 1), validated on the host
        self.label_len_dev = label_len_dev  # int32 (B,)
        self.pred_len_dev = pred_len_dev    # int32 (B,)
        self.ready = ready


class _Slot:
    def __init__(self):
        self.pinned = None   # flat host float32 staging buffer (grown on demand; pageable, see the module docstring)
        self.device = None   # flat device float32 buffer
        self.copied = None   # event: H2D of the last use finished
        self.consumed = None  # event on the compute stream: the last reader has been enqueued


class BatchStager:
    """Iterates `batches` (an iterable of List[LabeledSpectrogram]) ahead of the training loop.

    pack(batch) -> (spectrogram list, label_batch int32 (B, Lmax), label_lengths, prediction_lengths) is the net's own
    packer, so the staged tensors are exactly what train_on_batch would have built.  `depth` batches are in flight on
    `workers` threads (the native packer runs without the GIL); the batch iterable itself is only ever advanced by
    the consuming thread, and batches are delivered in order."""

    def __init__(self, batches, pack, device, blank, depth=3, workers=3, spare_slots=5):
        from concurrent.futures import ThreadPoolExecutor
        self.device = torch.device(device)
        self.pack = pack
        self.blank = blank  # labels must lie in [0, blank)
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.depth = max(1, depth)
        # A slot (host staging buffer + device buffer) may be refilled only after the step that read it has RUN on the
        # GPU.  With just depth + 1 slots every worker spent most of its time blocked on that (7 ms per batch, of which
        # 1 ms packing) and three workers could not quite feed a 2.3 ms step (cadence 2.65 ms); with spare slots the
        # reuse distance exceeds the GPU's backlog and the loop runs at the GPU's own cadence (2.31 ms).
        self.slots = [_Slot() for _ in range(self.depth + 1 + max(0, spare_slots))]
        self.pool = ThreadPoolExecutor(max_workers=max(1, workers), thread_name_prefix="speechless-stager")
        self.source = iter(batches)
        self.pending = []  # futures, oldest first
        self.submitted = 0
        self.exhausted = False
        self.copy_lock = threading.Lock()  # one H2D at a time on the copy stream, in submission order per slot

    # ---- worker threads
    def _stage(self, slot, batch):
        torch.cuda.set_device(self.device)
        spectrograms, labels, label_lengths, prediction_lengths = self.pack(batch)
        b = len(spectrograms)
        t_max = max(s.shape[0] for s in spectrograms)
        f = spectrograms[0].shape[1]
        n = b * t_max * f
        if slot.copied is not None:
            slot.copied.synchronize()      # the previous H2D out of this staging buffer is done
        if slot.consumed is not None:
            slot.consumed.synchronize()    # ... and the previous reader of the device buffer has run
        if slot.pinned is None or slot.pinned.numel() < n:
            slot.pinned = torch.empty((n,), dtype=torch.float32)
            slot.device = torch.empty((n,), dtype=torch.float32, device=self.device)
        host = slot.pinned[:n].view(b, t_max, f)
        pack_spectrograms(spectrograms, host.numpy())  # convert + zero-pad in one native pass (net.py:583-586)
        x_dev = slot.device[:n].view(b, t_max, f)
        # labels and lengths travel on the copy stream too: a pageable H2D copy on the COMPUTE stream blocks the host
        # until everything queued before it has run, which exposes the step's launch overhead (2.9 instead of 2.4 ms)
        labels = np.asarray(labels, dtype=np.int32)
        lab_len = np.asarray(label_lengths, dtype=np.int32).reshape(-1)
        pred_len = np.asarray(prediction_lengths, dtype=np.int32).reshape(-1)
        if labels.ndim != 2 or labels.shape[0] != b:
            raise ValueError("label batch must be (B, Lmax)")
        for i in range(b):
            row = labels[i, :lab_len[i]]
            if row.size and (row.min() < 0 or row.max() >= self.blank):
                raise ValueError("label {} holds an index outside [0, {}) (blank is {})".format(i, self.blank, self.blank))
        if labels.shape[1] == 0:
            labels = np.zeros((b, 1), dtype=np.int32)
        with self.copy_lock, torch.cuda.stream(self.copy_stream):
            x_dev.copy_(host, non_blocking=True)
            labels_dev = torch.from_numpy(np.ascontiguousarray(labels)).to(self.device)
            lab_len_dev = torch.from_numpy(lab_len).to(self.device)
            pred_len_dev = torch.from_numpy(pred_len).to(self.device)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        slot.copied = ready
        slot.consumed = None
        return StagedBatch(slot, x_dev, labels_dev, lab_len_dev, pred_len_dev, ready)

    def _fill(self):
        while not self.exhausted and len(self.pending) < self.depth:
            batch = next(self.source, None)  # errors of the corpus reader surface here, in the training loop
            if batch is None:
                self.exhausted = True
                break
            slot = self.slots[self.submitted % len(self.slots)]
            self.submitted += 1
            self.pending.append(self.pool.submit(self._stage, slot, batch))

    # ---- training loop side
    def __iter__(self):
        return self

    def __next__(self):
        self._fill()
        if not self.pending:
            raise StopIteration
        item = self.pending.pop(0).result()  # re-raises a worker's exception
        self._fill()
        stream = torch.cuda.current_stream(self.device)
        stream.wait_event(item.ready)
        # the label tensors were allocated on the copy stream: tell the caching allocator that the compute stream uses
        # them, so that their memory is not handed out again before the step that reads them has run
        for tensor in (item.labels_dev, item.label_len_dev, item.pred_len_dev):
            tensor.record_stream(stream)
        return item

    def release(self, staged):
        """Call once every kernel reading the staged tensors (input AND labels) has been enqueued on the current stream."""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        staged.slot.consumed = ev

    def close(self):
        for f in self.pending:
            f.cancel()
        self.pool.shutdown(wait=True)
        self.pending = []


class AudioBatchStager(BatchStager):
    """BatchStager for raw audio (training straight from audio).  A batch is a list of
    reference-style LabeledExample objects (labeled_example.py:74-140: `.get_raw_audio()`, `.label`); the worker threads
    read the audio, the H2D copy moves SAMPLES (512 bytes per 128-sample hop: the same bytes per frame as a 128-mel float32
    spectrogram) and the front end -- STFT, power level, mel projection, z-normalisation (speechless_amd/spectrogram.py) --
    runs in HBM: by default on the compute stream in front of the step that consumes it (0.13 ms per 32 x 8 s since round 3:
    92 % of the resident-input rate), optionally on the copy stream right behind the copy, beside the training step of the
    previous batch (90 %: the step's kernels and the front end's then share the CUs and the power budget; it was the better
    choice while the front end took 0.26 ms).  The spectrogram never exists on the host.  pack(batch) -> (raw audio list, label_batch, label_lengths); prediction lengths follow from the frame
    counts."""

    def __init__(self, batches, pack, extractor, length_ratio, device, blank, depth=3, workers=3, spare_slots=5,
                 front_end_on_copy_stream=False):
        super().__init__(batches, pack, device, blank, depth=depth, workers=workers, spare_slots=spare_slots)
        self.extractor = extractor
        self.length_ratio = length_ratio
        # False (default): only the samples travel on the copy stream; the front end runs on the COMPUTE stream in front of
        # the step that consumes it.  True: on the copy stream, under the previous step.
        self.front_end_on_copy_stream = front_end_on_copy_stream
        self.front_end_events = []  # optional (start, stop) timing events per batch, see time_front_end

    time_front_end = False

    def _stage(self, slot, batch):
        torch.cuda.set_device(self.device)
        audios, labels, label_lengths = self.pack(batch)
        b = len(audios)
        # the samples go into the staging buffer as a (B, longest, 1) batch through the native packer (several threads, no
        # GIL: a Python loop of 32 slice copies holds the interpreter for milliseconds, which the training thread feels)
        arrays = [np.asarray(a).reshape(-1, 1) for a in audios]
        arrays = [a if a.dtype in (np.float32, np.float64) else a.astype(np.float32) for a in arrays]
        lengths = np.array([a.shape[0] for a in arrays], dtype=np.int32)
        if lengths.min() <= self.extractor.n_fft // 2:
            raise ValueError("audio shorter than {} samples cannot be reflect-padded".format(self.extractor.n_fft // 2 + 1))
        t_max = int(lengths.max())
        n = b * t_max
        offsets = np.arange(b, dtype=np.int64) * t_max
        if slot.copied is not None:
            slot.copied.synchronize()      # the previous H2D out of this staging buffer is done
        if slot.consumed is not None:
            slot.consumed.synchronize()    # ... and the readers of the slot's device buffers have run
        if slot.pinned is None or slot.pinned.numel() < n:
            slot.pinned = torch.empty((n,), dtype=torch.float32)
            slot.device = torch.empty((n,), dtype=torch.float32, device=self.device)
        pack_spectrograms(arrays, slot.pinned[:n].view(b, t_max, 1).numpy())
        labels = np.asarray(labels, dtype=np.int32)
        lab_len = np.asarray(label_lengths, dtype=np.int32).reshape(-1)
        if labels.ndim != 2 or labels.shape[0] != b:
            raise ValueError("label batch must be (B, Lmax)")
        for i in range(b):
            row = labels[i, :lab_len[i]]
            if row.size and (row.min() < 0 or row.max() >= self.blank):
                raise ValueError("label {} holds an index outside [0, {}) (blank is {})".format(i, self.blank, self.blank))
        if labels.shape[1] == 0:
            labels = np.zeros((b, 1), dtype=np.int32)
        with self.copy_lock, torch.cuda.stream(self.copy_stream):
            # (the device audio buffer of this slot is only ever touched on the copy stream: refills are ordered behind
            # the front-end kernels that read it by the stream itself)
            audio_dev = slot.device[:n]
            audio_dev.copy_(slot.pinned[:n], non_blocking=True)
            off_dev = torch.from_numpy(offsets).to(self.device)
            len_dev = torch.from_numpy(lengths).to(self.device)
            copied = torch.cuda.Event()
            copied.record(self.copy_stream)
            if self.time_front_end:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.copy_stream)
            if self.front_end_on_copy_stream:
                x_dev, frames = self.extractor.batch_device(audio_dev, off_dev, len_dev, lengths, bufs=slot.__dict__.setdefault(
                    "front_end_buffers", {}))
            else:
                x_dev, frames = None, [self.extractor.frame_count(int(m)) for m in lengths]
                frames_dev = torch.tensor(frames, dtype=torch.int32, device=self.device)
            if self.time_front_end:
                t1.record(self.copy_stream)
                self.front_end_events.append((t0, t1))
            pred_len = np.array([f // self.length_ratio for f in frames], dtype=np.int32)
            labels_dev = torch.from_numpy(np.ascontiguousarray(labels)).to(self.device)
            lab_len_dev = torch.from_numpy(lab_len).to(self.device)
            pred_len_dev = torch.from_numpy(pred_len).to(self.device)
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        slot.copied = copied
        slot.consumed = None
        staged = StagedBatch(slot, x_dev, labels_dev, lab_len_dev, pred_len_dev, ready)
        staged.frames = frames
        staged.audio = (audio_dev, off_dev, len_dev, lengths, None if self.front_end_on_copy_stream else frames_dev)
        return staged

    def __next__(self):
        item = super().__next__()
        stream = torch.cuda.current_stream(self.device)
        if item.x_dev is None:  # front end on the compute stream, in front of the step
            audio_dev, off_dev, len_dev, lengths, frames_dev = item.audio
            for tensor in (off_dev, len_dev, frames_dev):
                tensor.record_stream(stream)
            item.x_dev, _ = self.extractor.batch_device(audio_dev, off_dev, len_dev, lengths, frames_dev,
                                                        bufs=item.slot.__dict__.setdefault("front_end_buffers", {}))
        return item
    # release() is the base class's: the slot (audio buffer, front-end buffers incl. the spectrogram the step's
    # sl_pack_input reads) may be refilled once everything enqueued up to the end of the step has run
