"""Audio front end: the CPU restatement of librosa's STFT / mel filter bank / the reference's
power level and z-normalisation (oracle/spectrogram_oracle.py) against independent implementations, and the HIP kernels
(speechless_amd/csrc/spectrogram.hip + the exact-fp32 MFMA projection) against that restatement.

Tolerance of the GPU tests: 2e-3 absolute on the z-normalised output (unit standard deviation), 2e-3 dB on levels --
fp32 arithmetic (the reference's own librosa FFT is single precision too) against a float64 restatement."""
import numpy as np
import pytest

from oracle import spectrogram_oracle as so


def synthetic_audio(seconds, seed, sample_rate=16000, silence=None):
    """A few drifting tones over a broadband noise floor 40 dB down (every bin well above fp32 round-off), optionally
    with a stretch of digital silence (exact zeros -> the -150 dB floor of labeled_example.py:150-158)."""
    rng = np.random.RandomState(seed)
    n = int(seconds * sample_rate)
    t = np.arange(n) / sample_rate
    y = 0.01 * rng.randn(n)
    for _ in range(4):
        f0, f1 = rng.uniform(100, 6000, size=2)
        y += rng.uniform(0.1, 0.4) * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / seconds))
    y *= np.hanning(n) ** 0.25
    if silence is not None:
        y[silence[0]:silence[1]] = 0.0
    return y.astype(np.float32)


# ------------------------------------------------------------------------------------------ oracle, CPU
def test_stft_restatement_against_torch_and_a_direct_dft():
    import torch
    y = synthetic_audio(0.7, 1).astype(np.float64)
    d = so.stft(y)
    assert d.shape == (257, 1 + len(y) // 128)
    ref = torch.stft(torch.from_numpy(y), n_fft=512, hop_length=128, window=torch.hann_window(512, periodic=True,
                     dtype=torch.float64), center=True, pad_mode="reflect", return_complex=True).numpy()
    assert np.abs(d - ref).max() < 1e-10
    # frame 3, directly: reflect-padded samples 3 * 128 - 256 .. + 511, periodic Hann, DFT by definition
    padded = np.pad(y, 256, mode="reflect")
    frame = padded[3 * 128: 3 * 128 + 512] * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512))
    k = np.arange(257)[:, None]
    direct = (frame[None, :] * np.exp(-2j * np.pi * k * np.arange(512)[None, :] / 512)).sum(axis=1)
    assert np.abs(d[:, 3] - direct).max() < 1e-9
    with pytest.raises(ValueError):
        so.stft(np.zeros(200))


def test_mel_filter_bank_properties_and_the_product_copy():
    """Slaney scale / area normalisation of librosa.filters.mel: band edges linear below 1 kHz (200/3 Hz per mel) and
    geometric above; every filter is a non-negative triangle of unit area in Hz; neighbours overlap; the matrix the
    product builds (speechless_amd/spectrogram.py) is the same one."""
    from speechless_amd.spectrogram import mel_filter_bank
    bank = so.mel_filter_bank(16000, 512, 128)
    assert bank.shape == (128, 257) and (bank >= 0).all()
    edges = so.mel_frequencies(130, 0.0, 8000.0)
    assert edges[0] == 0 and abs(edges[-1] - 8000) < 1e-9
    low = edges[edges < 1000]
    assert np.allclose(np.diff(low), low[1] - low[0])                      # linear part
    high = edges[edges > 1000]
    assert np.allclose(high[1:] / high[:-1], high[1] / high[0])            # logarithmic part
    assert abs(so.hz_to_mel_slaney(1000.0) - 15.0) < 1e-12 and abs(so.mel_to_hz_slaney(15.0) - 1000.0) < 1e-9
    df = 8000.0 / 256
    wide = (edges[2:] - edges[:-2]) > 8 * df                               # enough bins under the triangle
    assert np.allclose(bank[wide].sum(axis=1) * df, 1.0, atol=0.03)
    assert np.array_equal(np.argmax(bank, axis=1), np.sort(np.argmax(bank, axis=1)))
    assert np.abs(mel_filter_bank(16000, 512, 128) - bank).max() < 1e-15
    assert np.abs(mel_filter_bank(8000, 256, 40) - so.mel_filter_bank(8000, 256, 40)).max() < 1e-15


def test_power_level_and_z_normalisation():
    p = np.array([0.0, 1e-20, 1e-15, 1.0, 100.0])
    assert np.array_equal(so.power_level_from_power(p), np.array([-150.0, -150.0, -150.0, 0.0, 20.0]))
    a = np.random.RandomState(0).randn(50, 7) * 3 + 5
    z = so.z_normalize(a)
    assert abs(z.mean()) < 1e-12 and abs(z.std() - 1) < 1e-12
    y = synthetic_audio(0.5, 2, silence=(2000, 4000))
    s = so.z_normalized_transposed_spectrogram(y)
    assert s.shape == (1 + len(y) // 128, 128) and abs(s.mean()) < 1e-9 and abs(s.std() - 1) < 1e-9
    lin = so.z_normalized_transposed_spectrogram(y, mel_frequency_count=None)
    assert lin.shape == (1 + len(y) // 128, 257)


# ------------------------------------------------------------------------------------------ HIP kernels
@pytest.mark.gpu
@pytest.mark.parametrize("mel", [128, None, 40])
def test_gpu_front_end_matches_the_restatement(mel):
    """A ragged batch (0.4 .. 2.1 s, one utterance with digital silence) through sl_stft_power_db, the mel projection on
    the exact-fp32 MFMA kernel and sl_z_normalize, against the float64 restatement, utterance by utterance; rows behind a
    short utterance are exactly zero (the zero padding of net.py:583-586)."""
    from speechless_amd.spectrogram import SpectrogramExtractor
    audios = [synthetic_audio(2.1, 3), synthetic_audio(0.4, 4), synthetic_audio(1.3, 5, silence=(6000, 9000)),
              synthetic_audio(1.0, 6)]
    ext = SpectrogramExtractor(mel_frequency_count=mel)
    x, frames = ext.batch(audios)
    x = x.cpu().numpy()
    assert frames == [1 + len(a) // 128 for a in audios] and x.shape == (4, max(frames), 257 if mel is None else mel)
    worst = 0.0
    for i, a in enumerate(audios):
        want = so.z_normalized_transposed_spectrogram(a.astype(np.float64), mel_frequency_count=mel)
        got = x[i, :frames[i]]
        worst = max(worst, float(np.abs(got - want).max()))
        assert not x[i, frames[i]:].any()
    # linear scale: a bin where the noise happens to cancel sits > 120 dB below its frame's peak, at the fp32 round-off of
    # the frame's energy (the reference's own single-precision librosa FFT is no better there); the mel bands average that
    assert worst < (6e-3 if mel is None else 2e-3), worst
    one = ext.one(audios[1])
    assert np.array_equal(one, x[1, :frames[1]])


@pytest.mark.gpu
def test_gpu_power_levels_and_floor():
    """The level spectrogram itself (before mel / z-norm): within 2e-3 dB of the restatement wherever the power is well
    above the fp32 round-off of the frame's energy, exactly -150 on digital silence."""
    import torch
    from speechless_amd import _lib
    y = synthetic_audio(1.0, 7, silence=(4096, 8192))
    n_frames = 1 + len(y) // 128
    rows = 256
    dev = torch.device("cuda:0")
    out = torch.full((1, rows, 320), 7.0, dtype=torch.float32, device=dev)
    audio = torch.from_numpy(y).to(dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    length = torch.tensor([len(y)], dtype=torch.int32, device=dev)
    _lib.lib().call("sl_stft_power_db", audio.data_ptr(), off.data_ptr(), length.data_ptr(), out.data_ptr(), 1, rows,
                    512, 128, 320, rows * 320, -150.0, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().numpy()[0]
    want = so.power_level_from_power(np.abs(so.stft(y.astype(np.float64))) ** 2).T  # (frames, 257)
    assert not got[n_frames:].any() and not got[:, 257:].any()
    silent = [t for t in range(n_frames) if t * 128 - 256 >= 4096 and t * 128 + 256 <= 8192]
    assert len(silent) > 10 and (got[silent, :257] == -150.0).all() and (want[silent] == -150.0).all()
    loud = want > want.max(axis=1, keepdims=True) - 60  # bins within 60 dB of their frame's peak (fp32: eps * sqrt(512)
    assert np.abs(got[:n_frames, :257] - want)[loud].max() < 5e-3  # of the peak amplitude is 1e-3 relative down there)


@pytest.mark.gpu
@pytest.mark.parametrize("n_fft", [64, 128, 256, 1024])
def test_gpu_power_levels_at_other_window_lengths(n_fft):
    """sl_stft_power_db at every window length it accepts besides the reference's 512: the half-length complex transform
    has an odd number of radix-2 stages for 64, 256 and 1024 (a lone last stage behind the paired ones) and an even one
    for 128; levels against the float64 restatement as in the test above."""
    import torch
    from speechless_amd import _lib
    hop = n_fft // 4
    y = synthetic_audio(0.6, 17 + n_fft)
    n_frames = 1 + len(y) // hop
    bins = n_fft // 2 + 1
    stride = (bins + 63) // 64 * 64
    rows = n_frames + 5
    dev = torch.device("cuda:0")
    out = torch.full((1, rows, stride), 7.0, dtype=torch.float32, device=dev)
    audio = torch.from_numpy(y).to(dev)
    off = torch.zeros(1, dtype=torch.int64, device=dev)
    length = torch.tensor([len(y)], dtype=torch.int32, device=dev)
    _lib.lib().call("sl_stft_power_db", audio.data_ptr(), off.data_ptr(), length.data_ptr(), out.data_ptr(), 1, rows,
                    n_fft, hop, stride, rows * stride, -150.0, torch.cuda.current_stream().cuda_stream)
    got = out.cpu().numpy()[0]
    want = so.power_level_from_power(np.abs(so.stft(y.astype(np.float64), n_fft=n_fft, hop_length=hop)) ** 2).T
    assert want.shape == (n_frames, bins)
    assert not got[n_frames:].any() and not got[:, bins:].any()
    loud = want > want.max(axis=1, keepdims=True) - 60
    assert np.abs(got[:n_frames, :bins] - want)[loud].max() < 5e-3


@pytest.mark.gpu
def test_labeled_example_drop_in_feeds_the_net():
    """speechless_amd.spectrogram.LabeledExample (duck type of labeled_example.py:74-140) through Wav2Letter: the same
    transcription and loss whether the net gets the GPU-made spectrogram or the restatement's."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import LabeledSpectrogram
    from speechless_amd.spectrogram import LabeledExample
    small = dict(main_filter_count=20, out_filter_count=40, inner_count=1)
    net = Wav2Letter(128, english_frequent_characters, seed=3, layer_sizes=small, compute_dtype="f32")
    audios = [synthetic_audio(1.2, 8), synthetic_audio(0.9, 9)]
    gpu_examples = [LabeledExample(lambda a=a: a, id="u{}".format(i), label="abc de") for i, a in enumerate(audios)]
    cpu_examples = [LabeledSpectrogram("u{}".format(i), "abc de",
                                       so.z_normalized_transposed_spectrogram(a.astype(np.float64)))
                    for i, a in enumerate(audios)]
    a = net.test_and_predict_batch(gpu_examples)
    b = net.test_and_predict_batch(cpu_examples)
    assert [r.predicted for r in a.results] == [r.predicted for r in b.results]
    assert np.allclose([r.loss for r in a.results], [r.loss for r in b.results], rtol=1e-3)
    assert isinstance(net.predict(gpu_examples[0]), str)
    # raw audio to transcription without the spectrogram leaving HBM
    assert net.predict_batch_greedily_from_audio(audios) == net.predict_batch_greedily(
        [e.z_normalized_transposed_spectrogram() for e in gpu_examples])


@pytest.mark.gpu
@pytest.mark.parametrize("mel", [128, None])
def test_training_from_audio_in_hbm_equals_training_on_the_returned_spectrograms(tmp_path, mel):
    """Wav2Letter.train(..., from_audio=True): raw audio is staged, the front end runs in HBM (on the compute stream in
    front of the step, or on the copy stream beside the previous one: both stager variants checked) and the conv
    stack reads its output there (pipeline.AudioBatchStager) -- labeled_example.py:136-140 feeding net.py:593 without a
    host spectrogram.  The weights after the run must equal, bit for bit, those of the same run fed by
    LabeledExample.z_normalized_transposed_spectrogram() (GPU -> numpy -> host packer -> GPU), for the mel input of
    configuration 3 and the 257-bin linear input of configuration 5."""
    from speechless_amd import Wav2Letter, english_frequent_characters
    from speechless_amd.net import Adam
    from speechless_amd.spectrogram import LabeledExample
    rng = np.random.RandomState(5)
    words = ["she", "wasn't", "three", "abc", "xyz", "a", "it's", "zoo"]

    def example(i):
        audio = synthetic_audio(float(rng.uniform(1.0, 2.2)), 100 + i)
        return LabeledExample(lambda a=audio: a, id="u{}".format(i), mel_frequency_count=mel,
                              label=" ".join(rng.choice(words, size=rng.randint(1, 4))))
    batches = [[example(10 * j + i) for i in range(int(rng.randint(2, 5)))] for j in range(9)]
    small = dict(main_filter_count=250, out_filter_count=256, inner_count=2)
    finals = []
    for from_audio in (False, True):
        net = Wav2Letter(128 if mel else 257, english_frequent_characters, optimizer=Adam(1e-3), seed=4, layer_sizes=small)
        net.train(batches, preview_labeled_spectrogram_batch=batches[0][:2], tensor_board_log_directory=None,
                  net_directory=tmp_path / "n{}".format(int(from_audio)), batches_per_epoch=3, from_audio=from_audio)
        finals.append([w.copy() for w, _ in net.predictive_net.get_weights()])
    moved = 0.0
    for a, b in zip(*finals):
        assert np.array_equal(a, b)
        moved = max(moved, float(np.abs(a).max()))
    assert moved > 0
    with pytest.raises(ValueError, match="prefetch_depth"):
        net.train(batches, preview_labeled_spectrogram_batch=batches[0][:2], tensor_board_log_directory=None,
                  net_directory=tmp_path / "x", batches_per_epoch=3, from_audio=True, prefetch_depth=0)
    # the two places the front end can run: the same spectrograms, bit for bit
    import torch
    from speechless_amd.pipeline import AudioBatchStager
    staged = {}
    for on_copy in (False, True):
        stager = AudioBatchStager(batches[:4], net._pack_audio_for_staging, net._audio_extractor(batches[0][0]),
                                  net.input_to_prediction_length_ratio, net.engine.device,
                                  blank=net.grapheme_encoding.grapheme_set_size - 1, depth=2, workers=1,
                                  front_end_on_copy_stream=on_copy)
        out = []
        for item in stager:
            torch.cuda.current_stream().wait_event(item.ready)
            out.append(item.x_dev.clone())
            stager.release(item)
        torch.cuda.synchronize()
        stager.close()
        staged[on_copy] = out
    assert len(staged[True]) == len(staged[False]) == 4
    for a, b in zip(staged[True], staged[False]):
        assert torch.equal(a, b)
