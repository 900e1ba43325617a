"""Drop-in `Wav2Letter` for speechless.configuration, backed by the MI355X engine.

Mirrors the API surface of the reference's speechless/net.py:117-607 as consumed by speechless/configuration.py
(:96-139,:159-215), main.py (:121,:144,:202) and README.md (:85,:119): same constructor signature, `predict`,
`test_and_predict`, `test_and_predict_batch(es)`, `test_and_predict_grouped_batches`, `predict_batch_greedily`,
`prediction_batch`, `train`, `load_weights`, attributes `predictive_net`, `grapheme_encoding`,
`input_to_prediction_length_ratio`, and the result types `ExpectationVsPrediction*` (net.py:22-114).

All arithmetic (11 x Conv1D, softmax, CTC loss/gradient, greedy decode, Adam) runs in hand-written gfx950 kernels via
speechless_amd.engine; this file is host plumbing (batch packing net.py:578-607, result objects, epoch loop).
"""
import logging
import sys
from collections import OrderedDict
from pathlib import Path

import numpy as np

from .engine import Engine, wav2letter_layer_specs
from .grapheme_encoding import CtcGraphemeEncoding

logger = logging.getLogger("results")
if not logger.handlers:
    logger.setLevel(logging.INFO)
    _handler = logging.StreamHandler(sys.stdout)
    _handler.setLevel(logging.INFO)
    logger.addHandler(_handler)


def log(obj):
    logger.info(str(obj))


def _average_or_nan(numbers):
    return sum(numbers) / len(numbers) if len(numbers) else float("nan")


def edit_distance(a, b):
    """Levenshtein distance between two sequences (the reference uses the `editdistance` package, net.py:31-37)."""
    a, b = list(a), list(b)
    if len(a) < len(b):
        a, b = b, a
    previous = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        current = [i]
        for j, y in enumerate(b, 1):
            current.append(min(previous[j] + 1, current[j - 1] + 1, previous[j - 1] + (x != y)))
        previous = current
    return previous[-1]


class Adam:
    """Hyper-parameters of keras.optimizers.Adam as the reference constructs it (net.py:132: Adam(1e-4))."""

    def __init__(self, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon


class LabeledSpectrogram:
    """Duck type of speechless.labeled_example.LabeledSpectrogram (labeled_example.py:63-71)."""

    def __init__(self, id, label, spectrogram=None):
        self.id = id
        self.label = label
        self._spectrogram = spectrogram

    def z_normalized_transposed_spectrogram(self):
        return self._spectrogram


class ExpectationVsPrediction:
    def __init__(self, expected, predicted, loss):
        self.expected = expected
        self.predicted = predicted
        self.loss = loss
        self.expected_letter_count = len(expected)
        self.expected_words = expected.split()
        self.expected_word_count = len(self.expected_words)
        self.letter_error_count = edit_distance(expected, predicted)
        self.word_error_count = edit_distance(self.expected_words, predicted.split())

    @property
    def letter_error_rate(self):
        return self.letter_error_count / self.expected_letter_count

    @property
    def word_error_rate(self):
        return self.word_error_count / self.expected_word_count

    def __str__(self):
        return 'Expected:  "{}"\nPredicted: "{}"\nErrors: {} letters ({}%), {} words ({}%), loss: {:.2f}.'.format(
            self.expected, self.predicted, self.letter_error_count, round(self.letter_error_rate * 100),
            self.word_error_count, round(self.word_error_rate * 100), self.loss)


class ExpectationsVsPredictions:
    def __init__(self, results):
        self.results = results

    @property
    def average_letter_error_count(self):
        return _average_or_nan([r.letter_error_count for r in self.results])

    @property
    def average_word_error_count(self):
        return _average_or_nan([r.word_error_count for r in self.results])

    @property
    def average_letter_error_rate(self):
        return _average_or_nan([r.letter_error_rate for r in self.results])

    @property
    def average_word_error_rate(self):
        return _average_or_nan([r.word_error_rate for r in self.results])

    @property
    def average_loss(self):
        return _average_or_nan([r.loss for r in self.results])

    def summary_line(self):
        return ("Average over {} examples: {:.1f} letter errors ({:.2f}%), {:.1f} word errors ({:.2f}%), "
                "loss {:.2f}.").format(len(self.results), self.average_letter_error_count,
                                       self.average_letter_error_rate * 100, self.average_word_error_count,
                                       self.average_word_error_rate * 100, self.average_loss)

    def __str__(self):
        return "\n\n".join(str(r) for r in self.results) + "\n\n" + self.summary_line() + "\n\n"


class ExpectationsVsPredictionsInBatches(ExpectationsVsPredictions):
    def __init__(self, result_batches):
        self.result_batches = result_batches
        super().__init__([r for batch in result_batches for r in batch.results])

    def __str__(self):
        return "All batches: {}".format(self.summary_line())


class ExpectationsVsPredictionsInGroupedBatches(ExpectationsVsPredictions):
    def __init__(self, results_by_group_name):
        self.result_batches_by_group_name = results_by_group_name
        super().__init__([r for batches in results_by_group_name.values() for r in batches.results])

    def __str__(self):
        groups = "\n".join("{}: {}".format(name, batches)
                           for name, batches in self.result_batches_by_group_name.items())
        return "\n\n{}\n\nAll corpora: {}\n\n".format(groups, self.summary_line())


class _ConvLayerHandle:
    """What `predictive_net.layers[i]` needs to offer for net.py:237-269 style weight surgery."""

    def __init__(self, net, index, name):
        self._net, self._index, self.name = net, index, name
        self.trainable = True

    def get_weights(self):
        return list(self._net.get_weights()[self._index])

    def set_weights(self, kernel_and_bias):
        weights = self._net.get_weights()
        weights[self._index] = (np.asarray(kernel_and_bias[0], dtype=np.float32),
                                np.asarray(kernel_and_bias[1], dtype=np.float32))
        self._net.set_weights(weights)


class PredictiveNet:
    """Stand-in for the Keras Sequential the reference exposes as `Wav2Letter.predictive_net` (net.py:168)."""

    def __init__(self, engine):
        self._engine = engine
        self.layers = [_ConvLayerHandle(self, i, s.name) for i, s in enumerate(engine.all_specs)]

    def get_weights(self):
        return self._engine.get_weights()

    def set_weights(self, weights):
        self._engine.set_weights(weights)

    def save_weights(self, path):
        """Keras writes HDF5 (net.py:572: `predictive_net.save_weights(...)`): the same file -- root attribute
        `layer_names`, one group per layer with `weight_names`, `<layer>/kernel:0` (k, Cin, Cout) and `<layer>/bias:0`
        below it -- through speechless_amd/h5lite.py (no h5py needed; the real HDF5 library reads it, tests/test_h5lite.py).
        A path ending in .npz keeps the flat numpy form (`<layer>/kernel`, `<layer>/bias`)."""
        path = Path(path)
        weights = self.get_weights()
        if path.suffix == ".npz":
            arrays = {}
            for layer, (w, b) in zip(self.layers, weights):
                arrays[layer.name + "/kernel"] = w
                arrays[layer.name + "/bias"] = b
            np.savez(str(path), **arrays)
            return
        from . import h5lite
        h5lite.write_keras_weights(path, [(layer.name, [("{}/kernel:0".format(layer.name), w),
                                                         ("{}/bias:0".format(layer.name), b)])
                                          for layer, (w, b) in zip(self.layers, weights)])

    def load_weights(self, path):
        """Loads a Keras HDF5 checkpoint of the reference (`weights-epoch{N}.h5` from `save_weights`, net.py:209-212, or a
        full-model file whose tree sits under `model_weights`; Keras-1 weight names `<layer>_W` / `<layer>_b` and
        (k, 1, Cin, Cout) kernels included), or the flat .npz form.  When the .h5 that was asked for does not exist but
        its `.npz` twin does (files written by round-1/2 versions of this class), that one is used -- and logged."""
        path = Path(path)
        npz = path.with_suffix(".npz")
        if path.suffix == ".npz" or (not path.exists() and npz.exists()):
            if path.suffix != ".npz":
                log("{} not found: loading {}".format(path.name, npz.name))
            data = np.load(str(npz))
            self.set_weights([(data[layer.name + "/kernel"], data[layer.name + "/bias"]) for layer in self.layers])
            return
        from . import h5lite
        by_name = dict(h5lite.read_keras_weights(path))
        weights = []
        for layer in self.layers:
            if layer.name not in by_name:
                raise ValueError("{} holds no weights for layer {!r} (layers with weights: {})".format(
                    path.name, layer.name, ", ".join(by_name)))
            entry = by_name[layer.name]
            kernel = [v for n, v in entry.items() if "kernel" in n or n.endswith("W:0") or n.endswith("_W")]
            bias = [v for n, v in entry.items() if "bias" in n or n.endswith("b:0") or n.endswith("_b")]
            if len(kernel) != 1 or len(bias) != 1:
                raise ValueError("layer {!r} of {}: expected one kernel and one bias, found {}".format(
                    layer.name, path.name, list(entry)))
            w = np.asarray(kernel[0])
            if w.ndim == 4:  # Keras-1 style conv kernels (k, 1, Cin, Cout)
                w = w.reshape(w.shape[0], w.shape[2], w.shape[3])
            weights.append((w, np.asarray(bias[0])))
        self.set_weights(weights)


class Wav2Letter:
    """Speech-recognition network based on wav2letter (https://arxiv.org/pdf/1609.03193v2.pdf), MI355X engine."""

    class InputNames:
        input_batch = "input_batch"
        label_batch = "label_batch"
        prediction_lengths = "prediction_lenghts"  # (sic) reference net.py:123
        label_lengths = "label_lenghts"  # (sic) reference net.py:124

    def __init__(self, input_size_per_time_step, allowed_characters, use_raw_wave_input=False, activation="relu",
                 output_activation="softmax", optimizer=None, dropout=None, load_model_from_directory=None,
                 load_epoch=None, allowed_characters_for_loaded_model=None, frozen_layer_count=0,
                 reinitialize_trainable_loaded_layers=False, use_asg=False, asg_transition_probabilities=None,
                 asg_initial_probabilities=None, kenlm_directory=None,
                 # --- extensions of this implementation (keyword-only in spirit) ---
                 compute_dtype=None, device="cuda:0", seed=None, ctc_epsilon=1e-8, layer_sizes=None,
                 load_optimizer_state=False, eval_dtype=None):
        if frozen_layer_count > 0 and load_model_from_directory is None:
            raise ValueError("Layers cannot be frozen if model is trained from scratch.")
        if use_asg:
            raise NotImplementedError("ASG is not yet implemented.")  # as reference net.py:396-399
        if dropout is not None and not 0.0 <= dropout < 1.0:
            raise ValueError("dropout must be a rate in [0, 1)")
        self.kenlm_directory = kenlm_directory
        self.grapheme_encoding = CtcGraphemeEncoding(allowed_characters=allowed_characters)
        self.use_asg = use_asg
        self.frozen_layer_count = frozen_layer_count
        self.output_activation = output_activation
        self.activation = activation
        self.use_raw_wave_input = use_raw_wave_input
        self.input_size_per_time_step = input_size_per_time_step
        self.optimizer = optimizer if optimizer is not None else Adam(1e-4)
        self.load_epoch = load_epoch
        self.dropout = dropout
        # Arithmetic (round 6).  The reference has ONE arithmetic -- fp32 (net.py:389,402-406).  Called with
        # the reference's own signature (compute_dtype not given) this class TRAINS on the benchmarked bf16 engine and
        # EVALUATES -- prediction_batch, predict*, test_and_predict* -- on a second, lazily built engine of a fast PARITY
        # path (greedy decode bit-exact against the fp32 CPU port, loss to 2e-7) that shares the fp32 master weights in HBM
        # and re-packs its operand copies whenever they changed: `f16x3` (hi + lo fp16 planes, 22 operand bits; weights
        # representable up to |w| < 1000 -- eval_engine falls back to bf16x3 beyond that) for spectrogram input, `bf16x3`
        # (hi + lo bf16 planes, fp32's range) for raw-wave input.  compute_dtype="bf16" / "f32" / "bf16x3" / "f16x3" given
        # explicitly: that one engine does everything (eval_dtype overrides the evaluation side alone).
        self.compute_dtype = compute_dtype if compute_dtype is not None else "bf16"
        if eval_dtype is None:
            eval_dtype = ("bf16x3" if use_raw_wave_input else "f16x3") if compute_dtype is None else self.compute_dtype
        self.eval_dtype = eval_dtype
        compute_dtype = self.compute_dtype
        self.device = device
        self.ctc_epsilon = ctc_epsilon
        self._layer_sizes = dict(layer_sizes or {})
        specs = wav2letter_layer_specs(input_size_per_time_step, self.grapheme_encoding.grapheme_set_size,
                                       activation=activation, output_activation=output_activation,
                                       use_raw_wave_input=use_raw_wave_input, **self._layer_sizes)
        self.engine = Engine(specs, self.grapheme_encoding.grapheme_set_size, dtype=compute_dtype, device=device,
                             ctc_epsilon=ctc_epsilon, frozen_layer_count=frozen_layer_count, lr=self.optimizer.lr,
                             beta_1=self.optimizer.beta_1, beta_2=self.optimizer.beta_2,
                             adam_epsilon=self.optimizer.epsilon)
        self.engine.dropout_rate = dropout if dropout else None  # applied by training steps only (learning phase 1)
        # the reference signature has no seed: a plain Wav2Letter(..., dropout=0.1) draws one (Keras does the same)
        self.engine.dropout_seed = int(seed) if seed is not None else \
            int(np.random.SeedSequence().entropy & 0x7fffffff)
        self.engine.set_weights(self._glorot_uniform(specs, seed))
        self._eval_engine = None
        self._eval_weights_version = None
        self.predictive_net = PredictiveNet(self.engine)
        for layer in self.predictive_net.layers[:frozen_layer_count]:
            layer.trainable = False
        if frozen_layer_count > 0:
            log("All but {} layers frozen.".format(len(specs) - frozen_layer_count))
        self.prediction_phase_flag = 0.
        self._beam_decoder = None
        if self.kenlm_directory is not None:  # net.py:171-177
            from .decoder import CtcBeamSearchDecoder, expected_characters
            expected = expected_characters(self.kenlm_directory)
            if list(allowed_characters) != expected:
                raise ValueError("Allowed characters {} differ from those expected by kenlm decoder: {}".format(
                    allowed_characters, expected))
            self._beam_decoder = CtcBeamSearchDecoder.from_kenlm_directory(self.kenlm_directory, allowed_characters,
                                                                           epsilon=ctc_epsilon)
        if load_model_from_directory is not None:
            self.load_weights(allowed_characters_for_loaded_model, load_epoch, load_model_from_directory,
                              loaded_first_layers_count=frozen_layer_count if reinitialize_trainable_loaded_layers
                              else None)
            if load_optimizer_state:
                self.load_optimizer_state(load_model_from_directory, load_epoch)

    @staticmethod
    def _glorot_uniform(specs, seed):
        """Keras default initialisers: glorot_uniform kernels (limit sqrt(6/(fan_in+fan_out))), zero biases."""
        rng = np.random.RandomState(seed)
        weights = []
        for s in specs:
            limit = np.sqrt(6.0 / (s.kernel_size * s.cin + s.kernel_size * s.cout))
            weights.append((rng.uniform(-limit, limit, size=(s.kernel_size, s.cin, s.cout)).astype(np.float32),
                            np.zeros((s.cout,), dtype=np.float32)))
        return weights

    # ------------------------------------------------------------------ weights (net.py:184-269, 558-560)
    @staticmethod
    def model_file_name(epoch):
        return "weights-epoch{}.h5".format(epoch)

    @staticmethod
    def optimizer_state_file_name(epoch):
        return "weights-epoch{}.opt.npz".format(epoch)

    def save_optimizer_state(self, net_directory, epoch):
        """Extension: the reference saves the weights only (net.py:564-572), so a resumed run
        restarts Adam's moments and bias correction.  This writes them next to the weights file of the same epoch."""
        state = self.engine.get_optimizer_state()
        arrays = {"iterations": np.int64(state["iterations"]), "dropout_steps": np.int64(state["dropout_steps"])}
        for layer, (mw, mb), (vw, vb) in zip(self.predictive_net.layers, state["m"], state["v"]):
            arrays[layer.name + "/kernel/m"], arrays[layer.name + "/bias/m"] = mw, mb
            arrays[layer.name + "/kernel/v"], arrays[layer.name + "/bias/v"] = vw, vb
        Path(net_directory).mkdir(parents=True, exist_ok=True)
        np.savez(str(Path(net_directory) / self.optimizer_state_file_name(epoch)), **arrays)

    def load_optimizer_state(self, net_directory, epoch):
        data = np.load(str(Path(net_directory) / self.optimizer_state_file_name(epoch)))
        names = [layer.name for layer in self.predictive_net.layers]
        self.engine.set_optimizer_state({
            "m": [(data[n + "/kernel/m"], data[n + "/bias/m"]) for n in names],
            "v": [(data[n + "/kernel/v"], data[n + "/bias/v"]) for n in names],
            "iterations": int(data["iterations"]), "dropout_steps": int(data["dropout_steps"])})

    @staticmethod
    def indices_to_load_by_target_index(allowed_characters_for_loaded_model, allowed_characters):
        ignored = set(allowed_characters_for_loaded_model) - set(allowed_characters)
        if ignored:
            log("Ignoring characters {} from loaded model.".format(sorted(ignored)))
        extra = set(allowed_characters) - set(allowed_characters_for_loaded_model)
        if extra:
            log("Initializing extra characters {} not found in model.".format(sorted(extra)))
        mapping = []
        for character in allowed_characters:
            hits = [i for i, c in enumerate(allowed_characters_for_loaded_model) if c == character]
            assert len(hits) <= 1
            mapping.append(hits[0] if hits else None)
        log("Character mapping: {}".format(mapping))
        return mapping

    def load_weights(self, allowed_characters_for_loaded_model, load_epoch, load_model_from_directory,
                     loaded_first_layers_count=None):
        path = Path(load_model_from_directory) / self.model_file_name(load_epoch)
        if allowed_characters_for_loaded_model is None:
            self.predictive_net.load_weights(path)
            return
        layer_count = len(self.predictive_net.layers)
        if loaded_first_layers_count is None:
            loaded_first_layers_count = layer_count
        original = Wav2Letter(self.input_size_per_time_step, allowed_characters_for_loaded_model,
                              activation=self.activation, output_activation=self.output_activation,
                              optimizer=self.optimizer, load_model_from_directory=load_model_from_directory,
                              load_epoch=load_epoch, frozen_layer_count=self.frozen_layer_count,
                              compute_dtype=self.compute_dtype, device=self.device, layer_sizes=self._layer_sizes,
                              use_raw_wave_input=self.use_raw_wave_input)
        log("Loading first {} layers of {}, epoch {}, reinitializing the last {}.".format(
            loaded_first_layers_count, load_model_from_directory, load_epoch, layer_count - loaded_first_layers_count))
        source = original.predictive_net.get_weights()
        target = self.predictive_net.get_weights()
        for index in range(loaded_first_layers_count):
            kernel, bias = source[index]
            if index == layer_count - 1:
                mapping = self.indices_to_load_by_target_index(allowed_characters_for_loaded_model,
                                                               self.grapheme_encoding.allowed_characters)
                columns, biases = [], []
                for target_index in range(self.grapheme_encoding.grapheme_set_size):
                    source_index = original.grapheme_encoding.ctc_blank \
                        if target_index == self.grapheme_encoding.ctc_blank else mapping[target_index]
                    # reference quirk kept on purpose (net.py:254,258): `if index` also treats index 0 as missing
                    if source_index:
                        columns.append(kernel[:, :, source_index:source_index + 1])
                        biases.append(bias[source_index])
                    else:
                        columns.append(np.zeros((kernel.shape[0], kernel.shape[1], 1), dtype=kernel.dtype))
                        biases.append(0)
                kernel = np.concatenate(columns, axis=2)
                bias = np.array(biases, dtype=np.float32)
            target[index] = (kernel, bias)
        self.predictive_net.set_weights(target)

    # ------------------------------------------------------------------ packing (net.py:343-348, 578-607)
    @property
    def input_to_prediction_length_ratio(self):
        ratio = 1
        for s in self.engine.all_specs:
            ratio *= s.stride
        return ratio

    def _input_batch_and_prediction_lengths(self, spectrograms):
        input_lengths = [s.shape[0] for s in spectrograms]
        prediction_lengths = [n // self.input_to_prediction_length_ratio for n in input_lengths]
        input_batch = np.zeros((len(spectrograms), max(input_lengths), spectrograms[0].shape[1]), dtype=np.float32)
        for row, s in zip(input_batch, spectrograms):
            row[:s.shape[0], :s.shape[1]] = s
        return input_batch, prediction_lengths

    def _input_dictionary_for_loss_net(self, labeled_spectrogram_batch):
        spectrograms = [x.z_normalized_transposed_spectrogram() for x in labeled_spectrogram_batch]
        labels = [x.label for x in labeled_spectrogram_batch]
        input_batch, prediction_lengths = self._input_batch_and_prediction_lengths(spectrograms)
        n = len(labeled_spectrogram_batch)
        return {
            Wav2Letter.InputNames.input_batch: input_batch,
            Wav2Letter.InputNames.prediction_lengths: np.reshape(np.array(prediction_lengths), (n, 1)),
            Wav2Letter.InputNames.label_batch: self.grapheme_encoding.encode_label_batch(labels),
            Wav2Letter.InputNames.label_lengths: np.reshape(np.array([len(l) for l in labels]), (n, 1)),
        }

    # ------------------------------------------------------------------ inference (net.py:350-357, 461-498)
    @property
    def eval_engine(self):
        """The engine every forward-only entry point runs on (net.py:350-357, 461-498: learning phase 0).  Same object as
        self.engine when training and evaluation share a dtype; otherwise a second Engine of self.eval_dtype over the SAME
        flat fp32 master buffer (no copy: `params` is aliased), whose packed operand copies are refreshed when the training
        engine's weights_version moved (a training step, set_weights, a loaded checkpoint)."""
        if self.eval_dtype == self.compute_dtype:
            return self.engine
        train = self.engine
        if self._eval_engine is None:
            ev = Engine(list(train.all_specs), self.grapheme_encoding.grapheme_set_size, dtype=self.eval_dtype,
                        device=self.device, ctc_epsilon=self.ctc_epsilon, frozen_layer_count=self.frozen_layer_count,
                        forward_only=True)
            ev.params = train.params  # the masters themselves: same plan, same offsets (Engine.__init__ does not depend on dtype)
            assert ev.param_numel == train.param_numel
            self._eval_engine = ev
            self._eval_weights_version = None
        ev = self._eval_engine
        if self._eval_weights_version != train.weights_version:
            if ev.x3_f16 and float(train.params.abs().max().item()) * ev.w_scale >= 6.0e4:
                # fp16 planes hold w_scale * w: a master beyond +-937 would saturate -- evaluate on bf16 planes instead
                log("weights beyond the range of the f16x3 evaluation path: evaluating on bf16x3")
                self.eval_dtype, self._eval_engine = "bf16x3", None
                return self.eval_engine
            ev._packed_dirty = True  # (its next forward re-packs from the shared masters, on the same stream as the step)
            self._eval_weights_version = train.weights_version
        return ev

    def prediction_batch(self, input_batch):
        """Grapheme probabilities (B, T', K) for a (B, T, F) spectrogram batch."""
        return self.eval_engine.forward(np.asarray(input_batch)).cpu().numpy()

    def predict_batch_greedily(self, spectrograms):
        input_batch, prediction_lengths = self._input_batch_and_prediction_lengths(spectrograms)
        engine = self.eval_engine
        engine.forward(input_batch)
        decoded, _ = engine.greedy_decode(prediction_lengths)
        return [self.grapheme_encoding.decode_graphemes(d, merge_repeated=False) for d in decoded]

    def predict_batch_greedily_from_audio(self, raw_audio_batch, sample_rate=16000, fourier_window_length=512,
                                          hop_length=128):
        """Extension: predict_batch_greedily for raw audio (1-D float arrays).  STFT, power level, mel projection and
        z-normalisation (labeled_example.py:99-160, 28-29) run on the GPU (speechless_amd/spectrogram.py) and their result
        feeds the conv stack directly in HBM -- no spectrogram crosses PCIe."""
        from .spectrogram import shared_extractor
        bins = 1 + fourier_window_length // 2
        mel = None if self.input_size_per_time_step == bins else self.input_size_per_time_step
        extractor = shared_extractor(sample_rate, fourier_window_length, hop_length, mel, self.device)
        x, frames = extractor.batch(raw_audio_batch)
        engine = self.eval_engine
        engine.forward(x)
        decoded, _ = engine.greedy_decode([n // self.input_to_prediction_length_ratio for n in frames])
        return [self.grapheme_encoding.decode_graphemes(d, merge_repeated=False) for d in decoded]

    def test_and_predict_batch(self, labeled_spectrogram_batch):
        """ONE forward pass yields both the greedy transcription and the per-utterance CTC loss."""
        inputs = self._input_dictionary_for_loss_net(labeled_spectrogram_batch)
        names = Wav2Letter.InputNames
        engine = self.eval_engine
        engine.forward(inputs[names.input_batch])
        engine.set_labels(inputs[names.label_batch], inputs[names.label_lengths],
                          inputs[names.prediction_lengths])
        losses = engine.ctc().cpu().numpy()
        if self._beam_decoder is not None:  # net.py:444-451: beam search scored by the language model
            decoded, _ = self._beam_decoder.decode(engine.cur.probs.cpu().numpy(),
                                                   inputs[names.prediction_lengths])
        else:
            decoded, _ = engine.greedy_decode()
        predictions = [self.grapheme_encoding.decode_graphemes(d, merge_repeated=False) for d in decoded]
        return ExpectationsVsPredictions(
            [ExpectationVsPrediction(predicted=p, expected=x.label, loss=float(l))
             for p, x, l in zip(predictions, labeled_spectrogram_batch, losses)])

    def test_and_predict(self, labeled_spectrogram):
        # the reference duplicates the example because TF fails on batches of one (net.py:491-495); kept for parity
        return self.test_and_predict_batch([labeled_spectrogram, labeled_spectrogram]).results[0]

    def predict(self, labeled_spectrogram):
        return self.test_and_predict(labeled_spectrogram).predicted

    def test_and_predict_batch_with_log(self, index, batch):
        result = self.test_and_predict_batch(batch)
        log(str(result) + " (batch {})".format(index))
        return result

    def test_and_predict_batches(self, labeled_spectrogram_batches):
        return ExpectationsVsPredictionsInBatches(
            [self.test_and_predict_batch_with_log(i, batch) for i, batch in enumerate(labeled_spectrogram_batches)])

    def test_and_predict_batches_with_log(self, corpus_name, batches):
        result = self.test_and_predict_batches(batches)
        log("{}: {}".format(corpus_name, result))
        return result

    def test_and_predict_grouped_batches(self, grouped_labeled_spectrogram_batches):
        return ExpectationsVsPredictionsInGroupedBatches(OrderedDict(
            (name, self.test_and_predict_batches_with_log(corpus_name=name, batches=batches))
            for name, batches in grouped_labeled_spectrogram_batches.items()))

    # ------------------------------------------------------------------ training (net.py:541-576)
    def train_on_batch(self, labeled_spectrogram_batch, reducer=None, lazy=False):
        """One optimisation step; returns the mean CTC loss of the batch (what Keras' progress bar shows).
        lazy=True returns it as a 0-d tensor still in flight on the GPU, so the host can pack the next batch
        (net.py:578-607 is serial numpy work) while the step runs instead of blocking on `.item()`."""
        inputs = self._input_dictionary_for_loss_net(labeled_spectrogram_batch)
        names = Wav2Letter.InputNames
        losses = self.engine.train_step(inputs[names.input_batch], inputs[names.label_batch],
                                        inputs[names.label_lengths], inputs[names.prediction_lengths],
                                        reducer=reducer)
        mean = losses.mean()  # new tensor: safe against the next step overwriting the loss buffer
        return mean if lazy else float(mean.item())

    def _pack_for_staging(self, labeled_spectrogram_batch):
        """The host half of train_on_batch for pipeline.BatchStager: everything but the padded input array."""
        spectrograms = [np.asarray(x.z_normalized_transposed_spectrogram()) for x in labeled_spectrogram_batch]
        labels = [x.label for x in labeled_spectrogram_batch]
        ratio = self.input_to_prediction_length_ratio
        return (spectrograms, self.grapheme_encoding.encode_label_batch(labels),
                np.array([len(l) for l in labels], dtype=np.int32),
                np.array([s.shape[0] // ratio for s in spectrograms], dtype=np.int32))

    def _pack_audio_for_staging(self, labeled_example_batch):
        """The host half of a training step from raw audio (pipeline.AudioBatchStager): samples and encoded labels."""
        labels = [x.label for x in labeled_example_batch]
        return ([x.get_raw_audio() for x in labeled_example_batch], self.grapheme_encoding.encode_label_batch(labels),
                np.array([len(l) for l in labels], dtype=np.int32))

    def _audio_extractor(self, example):
        """The GPU front end matching this net's input (mel count = input size, or the linear 1 + n_fft / 2 bins) with the
        STFT parameters of the batch's LabeledExample objects (labeled_example.py:74-92)."""
        from .spectrogram import shared_extractor
        n_fft = getattr(example, "fourier_window_length", 512)
        hop = getattr(example, "hop_length", 128)
        rate = getattr(example, "sample_rate", 16000)
        # z_normalized_transposed_spectrogram() is always the MEL spectrogram of the example's own mel_frequency_count
        # (labeled_example.py:136-140): a net whose input size differs from it cannot be fed from this corpus
        # (speechless_amd.spectrogram.LabeledExample(mel_frequency_count=None) asks for the linear 1 + n_fft / 2 bins)
        if hasattr(example, "mel_frequency_count"):
            mel = example.mel_frequency_count
            bins = mel if mel is not None else 1 + n_fft // 2
            if bins != self.input_size_per_time_step:
                raise ValueError("the examples yield {} {} bins per frame, the net expects {} inputs per time step".format(
                    bins, "mel" if mel is not None else "linear", self.input_size_per_time_step))
        else:  # not a LabeledExample: the net's input size decides
            mel = None if self.input_size_per_time_step == 1 + n_fft // 2 else self.input_size_per_time_step
        return shared_extractor(rate, n_fft, hop, mel, self.device)

    def train_on_staged_batch(self, staged, stager, reducer=None):
        """train_on_batch for a pipeline.StagedBatch (input already in HBM, arrival ordered by an event)."""
        self.engine.load_input(staged.x_dev)
        self.engine.set_labels_resident(staged.labels_dev, staged.label_len_dev, staged.pred_len_dev)
        mean = self.engine.train_step_resident(reducer).mean()
        stager.release(staged)  # everything that reads the staged tensors is enqueued now
        return mean

    def train(self, labeled_spectrogram_batches, preview_labeled_spectrogram_batch, tensor_board_log_directory,
              net_directory, batches_per_epoch, max_epochs=100000000, reducer=None, prefetch_depth=3,
              save_optimizer_state=False, from_audio=False):
        """Epoch loop of reference net.py:541-576: preview, then epochs of `batches_per_epoch` steps starting at
        `load_epoch or 0`; after every epoch the preview is logged and (epoch > 0) the weights are saved as
        weights-epoch{N}.  Ends when the batch iterable is exhausted or after max_epochs (Keras: 1e8).
        prefetch_depth > 0: batches are packed on a worker thread and copied to HBM on a side stream
        (speechless_amd/pipeline.py) while the previous steps run; 0 = the reference's serial behaviour.
        save_optimizer_state: also write weights-epoch{N}.opt.npz (Adam moments + step count) with every checkpoint,
        for Wav2Letter(..., load_optimizer_state=True) to resume exactly where the run stopped.
        from_audio: the batches hold reference-style LabeledExample objects (labeled_example.py:74-140) and the spectrograms
        are computed on the GPU from their raw audio (`get_raw_audio()`): the samples arrive on the copy stream under the
        previous step, the front end runs on the compute stream in front of the step that consumes it
        (pipeline.AudioBatchStager, front_end_on_copy_stream=False) -- labeled_example.py:136-140 feeding net.py:593
        without the spectrogram ever existing on the host.  Needs prefetch_depth > 0."""
        def print_preview_batch():
            log(self.test_and_predict_batch(preview_labeled_spectrogram_batch))

        print_preview_batch()
        stager = None
        if from_audio and self.use_raw_wave_input:
            raise ValueError("from_audio=True computes spectrograms on the GPU; a raw-wave net takes the samples themselves "
                             "(batches of (T, {}) arrays)".format(self.input_size_per_time_step))
        if from_audio:
            from .pipeline import AudioBatchStager
            if prefetch_depth <= 0:
                raise ValueError("from_audio=True trains through the staged pipeline: prefetch_depth must be > 0")
            labeled_spectrogram_batches = iter(labeled_spectrogram_batches)
            first = next(labeled_spectrogram_batches, None)

            def with_first():
                if first is not None:
                    yield first
                    yield from labeled_spectrogram_batches
            extractor = self._audio_extractor(first[0]) if first else None
            stager = AudioBatchStager(with_first(), self._pack_audio_for_staging, extractor,
                                      self.input_to_prediction_length_ratio, self.engine.device,
                                      blank=self.grapheme_encoding.grapheme_set_size - 1, depth=prefetch_depth)
            batches = iter(stager)
        elif prefetch_depth > 0:
            from .pipeline import BatchStager
            stager = BatchStager(labeled_spectrogram_batches, self._pack_for_staging, self.engine.device,
                                 blank=self.grapheme_encoding.grapheme_set_size - 1, depth=prefetch_depth)
            batches = iter(stager)
        else:
            batches = iter(labeled_spectrogram_batches)
        epoch = self.load_epoch if self.load_epoch is not None else 0
        log_path = None
        if tensor_board_log_directory is not None:
            Path(tensor_board_log_directory).mkdir(parents=True, exist_ok=True)
            log_path = Path(tensor_board_log_directory) / "loss.csv"
        try:
            while epoch < max_epochs:
                epoch_losses = []
                for _ in range(batches_per_epoch):
                    batch = next(batches, None)
                    if batch is None:
                        break
                    if stager is not None:
                        epoch_losses.append(self.train_on_staged_batch(batch, stager, reducer=reducer))
                    else:
                        epoch_losses.append(self.train_on_batch(batch, reducer=reducer, lazy=True))
                epoch_losses = [float(l.item()) for l in epoch_losses]  # one host<->device sync per epoch, not per step
                if len(epoch_losses) < batches_per_epoch:
                    break
                if log_path is not None:
                    with log_path.open("a") as f:
                        f.write("{},{}\n".format(epoch, _average_or_nan(epoch_losses)))
                log("Epoch {}: loss {:.4f}".format(epoch, _average_or_nan(epoch_losses)))
                print_preview_batch()
                if epoch > 0:
                    Path(net_directory).mkdir(parents=True, exist_ok=True)
                    self.predictive_net.save_weights(Path(net_directory) / self.model_file_name(epoch))
                    if save_optimizer_state:
                        self.save_optimizer_state(net_directory, epoch)
                epoch += 1
        finally:
            if stager is not None:
                stager.close()
