"""Audio -> z-normalised (mel) spectrogram on the MI355X (second half).

Mirrors the reference's speechless/labeled_example.py: `LabeledExample` with `get_raw_audio`, `sample_rate`,
`fourier_window_length` (512), `hop_length` (128), `mel_frequency_count` (128) and
`z_normalized_transposed_spectrogram() -> (frames, bins)` (labeled_example.py:74-140), which is what
`Wav2Letter` consumes (net.py:593).  The arithmetic the reference delegates to librosa / numpy runs in three launches:

    sl_stft_power_db   reflect-padded, Hann-windowed frames -> radix-2 FFT in LDS -> |D|^2 -> 10 log10 with the -150 floor
    sl_conv1d_nt       the mel projection of the LEVEL spectrogram (labeled_example.py:106-109, 114-129) as a 1 x 1
                       convolution on the exact-fp32 MFMA kernel (weights = librosa.filters.mel's matrix)
    sl_z_normalize     (a - mean) / std per utterance (labeled_example.py:28-29), zero rows behind short utterances

`SpectrogramExtractor.batch()` keeps the result in HBM as the zero-padded float32 (B, Tmax, F) batch that
`Engine.load_input` takes (net.py:578-587's layout), so a training or prediction step on raw audio never moves a
spectrogram over PCIe; `LabeledExample.z_normalized_transposed_spectrogram()` is the drop-in, numpy-returning form.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import ConvGeom, lib

TIME_TILE = 256


def _round_up(x, m):
    return (x + m - 1) // m * m


def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
    return np.where(m >= min_log_hz / f_sp, min_log_hz * np.exp(logstep * (m - min_log_hz / f_sp)), f_sp * m)


def mel_filter_bank(sample_rate, n_fft, n_mels):
    """The matrix librosa.filters.mel(sr, n_fft, n_mels) returns with its defaults (labeled_example.py:106-109): Slaney's
    mel scale between 0 and sr / 2, triangular filters on the FFT bin frequencies, each scaled by 2 / bandwidth."""
    bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sample_rate / 2.0, bins)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sample_rate / 2.0), n_mels + 2))
    widths = np.diff(edges)
    ramps = edges[:, None] - fftfreqs[None, :]
    bank = np.maximum(0, np.minimum(-ramps[:-2] / widths[:-1, None], ramps[2:] / widths[1:, None]))
    return bank * (2.0 / (edges[2:] - edges[:-2]))[:, None]


class SpectrogramExtractor:
    """Batched front end on one GPU.  mel_frequency_count=None keeps the linear frequency scale (1 + n_fft // 2 bins:
    the 257-bin power-level spectrograms of BASELINE config 5)."""

    def __init__(self, sample_rate=16000, fourier_window_length=512, hop_length=128, mel_frequency_count=128,
                 min_decibel=-150.0, device="cuda:0"):
        if not torch.cuda.is_available():
            raise _lib.HipLibraryError("speechless_amd.spectrogram needs a ROCm GPU; there is no CPU fallback")
        self.lib = lib()
        self.device = torch.device(device)
        self.sample_rate = sample_rate
        self.n_fft = fourier_window_length
        self.hop = hop_length
        self.n_mels = mel_frequency_count
        self.min_decibel = float(min_decibel)
        self.bins = 1 + self.n_fft // 2
        self.bins_pad = _round_up(self.bins, 64)
        self.features = self.bins if mel_frequency_count is None else mel_frequency_count
        self.mel_w = None
        if mel_frequency_count is not None:
            self.mel_pad = _round_up(mel_frequency_count, 128)
            w = np.zeros((self.mel_pad, 1, self.bins_pad), dtype=np.float32)  # packed [cout][taps][cin] (sl_conv1d_nt)
            w[:mel_frequency_count, 0, :self.bins] = mel_filter_bank(sample_rate, self.n_fft, mel_frequency_count)
            self.mel_w = torch.from_numpy(w).to(self.device)

    def frame_count(self, sample_count):
        return 1 + sample_count // self.hop

    def batch(self, raw_audio_list):
        """raw_audio_list: 1-D float arrays (any float dtype).  Returns (x, frames): x float32 (B, Tmax, F) in HBM, every
        utterance z-normalised over its own frames and zero beyond them; frames int list."""
        flat, offsets, lengths = self.flatten(raw_audio_list)
        flat_dev = torch.from_numpy(flat).to(self.device, non_blocking=True)
        off_dev = torch.from_numpy(offsets).to(self.device, non_blocking=True)
        len_dev = torch.from_numpy(lengths).to(self.device, non_blocking=True)
        return self.batch_device(flat_dev, off_dev, len_dev, lengths)

    def flatten(self, raw_audio_list, out=None):
        """Host half of batch(): the utterances back to back as float32 (into `out`, a flat float32 array, if given),
        their start offsets (int64) and sample counts (int32)."""
        audios = [np.asarray(a).reshape(-1) for a in raw_audio_list]
        if not audios:
            raise ValueError("empty batch")
        lengths = np.array([a.shape[0] for a in audios], dtype=np.int32)
        if lengths.min() <= self.n_fft // 2:
            raise ValueError("audio shorter than {} samples cannot be reflect-padded (librosa.stft raises too)".format(
                self.n_fft // 2 + 1))
        offsets = np.zeros(len(audios), dtype=np.int64)
        offsets[1:] = np.cumsum(lengths[:-1])
        total = int(lengths.sum())
        flat = np.empty((total,), dtype=np.float32) if out is None else out[:total]
        for a, off, n in zip(audios, offsets, lengths):
            flat[off:off + n] = a  # (casts float64 / int16-scaled-by-the-reader input to float32)
        return flat, offsets, lengths

    def batch_device(self, flat_dev, off_dev, len_dev, lengths, frames_dev=None, bufs=None):
        """Device half of batch(): audio already in HBM (flat float32, int64 offsets, int32 sample counts; `lengths` the
        same counts on the host).  Everything is enqueued on the CURRENT stream -- the staged input pipeline calls this
        on its copy stream, right behind the H2D copy of the audio, so the front end of batch n + 1 runs under step n.
        bufs: a dict the caller owns (one per staging slot): intermediate and output tensors are kept there and reused, so
        that a steady-state call allocates nothing (allocations on a side stream whose results another stream reads make
        the caching allocator fall back to hipMalloc whenever the reader has not caught up: the staged run then jitters
        between 69 and 86 % of the resident rate)."""
        def tensor(name, shape, dtype):
            n = int(np.prod(shape))
            if bufs is None:
                return torch.empty(shape, dtype=dtype, device=self.device)
            flat = bufs.get(name)
            if flat is None or flat.numel() < n or flat.dtype != dtype:
                flat = bufs[name] = torch.empty((n,), dtype=dtype, device=self.device)
            return flat[:n].view(shape)

        frames = [self.frame_count(int(n)) for n in lengths]
        if frames_dev is None:  # (a pageable H2D copy: callers on the compute stream bring it along from their copy stream)
            frames_dev = torch.tensor(frames, dtype=torch.int32, device=self.device)
        b, t_max = len(frames), max(frames)
        rows = _round_up(t_max, TIME_TILE)  # sl_conv1d_nt reads whole time tiles
        st = torch.cuda.current_stream(self.device).cuda_stream
        level = tensor("level", (b, rows, self.bins_pad), torch.float32)
        self.lib.call("sl_stft_power_db", flat_dev.data_ptr(), off_dev.data_ptr(), len_dev.data_ptr(), level.data_ptr(), b,
                      rows, self.n_fft, self.hop, self.bins_pad, rows * self.bins_pad, self.min_decibel, st)
        src, src_stride = level, self.bins_pad
        if self.mel_w is not None:
            mel = tensor("mel", (b, rows, self.mel_pad), torch.float32)
            g = ConvGeom()
            g.batch, g.t_out, g.taps, g.cin, g.cout = b, t_max, 1, self.bins_pad, self.mel_pad
            g.x_row0, g.x_row_stride, g.x_batch_stride = 0, self.bins_pad, rows * self.bins_pad
            g.y_row0, g.y_row_stride, g.y_batch_stride = 0, self.mel_pad, rows * self.mel_pad
            self.lib.call("sl_conv1d_nt", level.data_ptr(), self.mel_w.data_ptr(), None, None, mel.data_ptr(),
                          ctypes.byref(g), _lib.EPI_NONE, _lib.SL_F32, 0, 0, None, 0, st)
            src, src_stride = mel, self.mel_pad
        out = tensor("out", (b, t_max, self.features), torch.float32)
        ws = tensor("ws", (self.lib.raw("sl_z_normalize_workspace_bytes")(b),), torch.uint8)
        self.lib.call("sl_z_normalize", src.data_ptr(), frames_dev.data_ptr(), out.data_ptr(), b, t_max, self.features,
                      src_stride, rows * src_stride, ws.data_ptr(), ws.numel(), st)
        return out, frames

    def one(self, raw_audio):
        x, frames = self.batch([raw_audio])
        return x[0, :frames[0]].cpu().numpy()


_EXTRACTORS = {}


def shared_extractor(sample_rate, fourier_window_length, hop_length, mel_frequency_count, device="cuda:0"):
    key = (sample_rate, fourier_window_length, hop_length, mel_frequency_count, str(device))
    if key not in _EXTRACTORS:
        _EXTRACTORS[key] = SpectrogramExtractor(sample_rate, fourier_window_length, hop_length, mel_frequency_count,
                                                device=device)
    return _EXTRACTORS[key]


class LabeledExample:
    """Duck type of speechless.labeled_example.LabeledExample (labeled_example.py:74-140) whose spectrogram is computed
    on the GPU: same constructor arguments (the ones the net needs), `.id`, `.label`,
    `.z_normalized_transposed_spectrogram()` -> float32 ndarray (frames, mel_frequency_count)."""

    def __init__(self, get_raw_audio, sample_rate=16000, id=None, label="nolabel", fourier_window_length=512,
                 hop_length=128, mel_frequency_count=128, label_with_tags=None, positional_label=None, device="cuda:0"):
        self.get_raw_audio = get_raw_audio
        self.sample_rate = sample_rate
        self.id = id
        self.label = label
        self.fourier_window_length = fourier_window_length
        self.hop_length = hop_length
        self.mel_frequency_count = mel_frequency_count
        self.label_with_tags = label_with_tags
        self.positional_label = positional_label
        self._device = device

    def z_normalized_transposed_spectrogram(self):
        extractor = shared_extractor(self.sample_rate, self.fourier_window_length, self.hop_length,
                                     self.mel_frequency_count, self._device)
        return extractor.one(self.get_raw_audio())

    @property
    def duration_in_s(self):
        return len(self.get_raw_audio()) / self.sample_rate

    def __str__(self):
        return str(self.id) + (": {}".format(self.label) if self.label else "")
