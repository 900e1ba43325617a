"""CPU restatement of the reference's audio -> z-normalised spectrogram front end (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(speechless_amd/spectrogram.py -> csrc/spectrogram.hip) never does.

What it follows (paths relative to the reference root):
  speechless/labeled_example.py:99-100   _complex_spectrogram = librosa.stft(y, n_fft=512, hop_length=128)
  speechless/labeled_example.py:93-97    amplitude = |D|, power = amplitude ** 2
  speechless/labeled_example.py:150-158  power level: 10 * log10(x), -150 for x == 0 and for anything below -150
  speechless/labeled_example.py:106-109  mel: dot(librosa.filters.mel(sr, n_fft, n_mels), spectrogram) -- applied to the
                                         POWER-LEVEL (dB) matrix, which is what spectrogram(frequency_scale=mel) hands it
                                         (labeled_example.py:114-129)
  speechless/labeled_example.py:136-140, 28-29  z_normalize(spectrogram.T) = (a - mean(a)) / std(a) over the whole matrix

PARITY UNPINNED: the arithmetic lives in librosa (third party, not vendored, not version-pinned: requirements.txt lists a
bare `librosa`), which is not installed here, and the reference's only test of this code
(speechless/test/test_labeled_example.py:12-21) compares against librosa itself after downloading LibriSpeech.  The
functions below restate librosa's published algorithms as of the 0.5 series the reference was written against:
  librosa.stft defaults   : win_length = n_fft, window = scipy.signal.get_window("hann", n_fft, fftbins=True) (the
                            PERIODIC Hann window), center=True with np.pad(mode="reflect") by n_fft // 2 on both sides,
                            frame t = padded[t * hop : t * hop + n_fft], 1 + len(y) // hop frames, 1 + n_fft // 2 bins
  librosa.filters.mel     : fmin = 0, fmax = sr / 2, htk=False (Slaney's scale: linear below 1 kHz, 27 log-spaced steps
                            per factor 6.4 above), triangular filters between n_mels + 2 band edges on the FFT bin
                            frequencies, norm=1 (each filter scaled by 2 / (f_hi - f_lo))
and are cross-checked in tests/test_spectrogram.py against independent implementations available here (torch.stft with
the same conventions; a direct DFT; closed-form properties of the filter bank).
"""
import numpy as np


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True)"""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft(y, n_fft=512, hop_length=128):
    """librosa.stft(y, n_fft, hop_length) with its defaults -> complex (1 + n_fft // 2, 1 + len(y) // hop_length)."""
    y = np.asarray(y, dtype=np.float64)
    if y.ndim != 1 or len(y) <= n_fft // 2:
        raise ValueError("audio must be one-dimensional and longer than n_fft // 2 samples (reflect padding)")
    padded = np.pad(y, n_fft // 2, mode="reflect")
    n_frames = 1 + (len(padded) - n_fft) // hop_length
    window = hann_periodic(n_fft)
    frames = np.stack([padded[t * hop_length: t * hop_length + n_fft] for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames * window[:, None], axis=0)


def power_level_from_power(power, min_decibel=-150.0):
    """labeled_example.py:150-158 (vectorised)."""
    power = np.asarray(power, dtype=np.float64)
    with np.errstate(divide="ignore"):
        level = 10.0 * np.log10(power)
    return np.where((power == 0) | (level < min_decibel), min_decibel, level)


def hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, f / f_sp)


def mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_frequencies(n_mels, fmin=0.0, fmax=8000.0):
    """librosa.mel_frequencies(n_mels, fmin, fmax, htk=False)"""
    return mel_to_hz_slaney(np.linspace(hz_to_mel_slaney(fmin), hz_to_mel_slaney(fmax), n_mels))


def mel_filter_bank(sr=16000, n_fft=512, n_mels=128):
    """librosa.filters.mel(sr, n_fft, n_mels) with its defaults -> (n_mels, 1 + n_fft // 2)."""
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_frequencies(n_mels + 2, 0.0, sr / 2.0)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return weights * enorm[:, None]


def z_normalize(a):
    """labeled_example.py:28-29"""
    return (a - np.mean(a)) / np.std(a)


def z_normalized_transposed_spectrogram(y, sample_rate=16000, n_fft=512, hop_length=128, mel_frequency_count=128):
    """LabeledExample.z_normalized_transposed_spectrogram() (labeled_example.py:136-140): (frames, mel bins) float64.
    mel_frequency_count=None: linear frequency scale (1 + n_fft // 2 bins), i.e. z_normalize(power level).T"""
    level = power_level_from_power(np.abs(stft(y, n_fft, hop_length)) ** 2)
    if mel_frequency_count is not None:
        level = mel_filter_bank(sample_rate, n_fft, mel_frequency_count) @ level
    return z_normalize(level.T)
