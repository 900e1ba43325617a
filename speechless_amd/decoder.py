"""CTC beam-search decoding with an n-gram language model: host mirror of the reference's `kenlm_directory` branch
(speechless/net.py:171-177, 444-451) over libspeechless_host.so (csrc_host/beam_search.cpp).

A KenLM directory of the reference holds a `vocabulary` file -- one line with the allowed characters, compared (lower-
cased) against the net's alphabet at construction (net.py:171-177) -- and the language model for the patched
TensorFlow.  KenLM's binary format cannot be read without KenLM, so this implementation takes the model as an ARPA
file in the same directory (`lm.arpa`, or the only `*.arpa` there).
"""
import ctypes
from pathlib import Path

import numpy as np

from ._host_lib import host_lib

# net.py:447-450
KENLM_WEIGHT = .8
WORD_COUNT_WEIGHT = 0.
VALID_WORD_COUNT_WEIGHT = 2.3
DEFAULT_BEAM_WIDTH = 100  # tf.nn.ctc_beam_search_decoder's default


def _lib():
    return host_lib()


class NGramLanguageModel:
    """ARPA n-gram model with back-off (what KenLM's FullScore().prob returns, log10)."""

    def __init__(self, arpa_path):
        err = ctypes.create_string_buffer(512)
        self._lib = _lib()
        self._handle = self._lib.sl_host_lm_load_arpa(str(arpa_path).encode("utf8"), err, len(err))
        if not self._handle:
            raise ValueError("cannot load language model: {}".format(err.value.decode("utf8", "replace")))
        self.order = self._lib.sl_host_lm_order(self._handle)

    def score(self, sentence):
        """log10 P(<s> sentence </s>), like kenlm.Model.score."""
        return float(self._lib.sl_host_lm_score_sentence(self._handle, sentence.encode("utf8")))

    def __del__(self):
        if getattr(self, "_handle", None):
            self._lib.sl_host_lm_free(self._handle)
            self._handle = None


def expected_characters(kenlm_directory):
    """net.py:171-174: the single line of <kenlm_directory>/vocabulary, lower-cased, as a character list."""
    lines = (Path(kenlm_directory) / "vocabulary").read_text(encoding="utf8").splitlines()
    if len(lines) != 1:
        raise AssertionError("Expected exactly one line in {}".format(Path(kenlm_directory) / "vocabulary"))
    return list(lines[0].lower())


def find_arpa(kenlm_directory):
    directory = Path(kenlm_directory)
    if (directory / "lm.arpa").exists():
        return directory / "lm.arpa"
    candidates = sorted(directory.glob("*.arpa"))
    if len(candidates) != 1:
        raise ValueError("{} must hold the language model as lm.arpa (or exactly one *.arpa file); KenLM binary models "
                         "cannot be read without KenLM".format(directory))
    return candidates[0]


class CtcBeamSearchDecoder:
    """tf.nn.ctc_beam_search_decoder over the net's probabilities, optionally scored by a language model."""

    def __init__(self, allowed_characters, language_model=None, beam_width=DEFAULT_BEAM_WIDTH, merge_repeated=False,
                 kenlm_weight=KENLM_WEIGHT, word_count_weight=WORD_COUNT_WEIGHT,
                 valid_word_count_weight=VALID_WORD_COUNT_WEIGHT, epsilon=1e-8, threads=8):
        self._lib = _lib()
        self.allowed_characters = list(allowed_characters)
        self.beam_width = beam_width
        self.merge_repeated = merge_repeated
        self.epsilon = epsilon
        self.threads = threads
        self.language_model = language_model
        self._scorer = None
        if language_model is not None:
            alphabet = np.array([ord(c) for c in self.allowed_characters], dtype=np.uint32)
            self._scorer = self._lib.sl_host_scorer_create(language_model._handle, alphabet.ctypes.data, len(alphabet),
                                                           kenlm_weight, word_count_weight, valid_word_count_weight)

    @classmethod
    def from_kenlm_directory(cls, kenlm_directory, allowed_characters, **kw):
        return cls(allowed_characters, NGramLanguageModel(find_arpa(kenlm_directory)), **kw)

    def decode(self, probabilities, prediction_lengths):
        """probabilities: (B, T', K) float array; returns (list of index lists, log-probabilities (B,))."""
        probs = np.ascontiguousarray(probabilities, dtype=np.float32)
        b, t, k = probs.shape
        if k != len(self.allowed_characters) + 1:
            raise ValueError("{} classes for an alphabet of {} characters + blank".format(k, len(self.allowed_characters)))
        lengths = np.ascontiguousarray(np.asarray(prediction_lengths).reshape(-1), dtype=np.int32)
        out = np.empty((b, t), dtype=np.int32)
        out_len = np.empty((b,), dtype=np.int32)
        log_prob = np.empty((b,), dtype=np.float32)
        rc = self._lib.sl_host_ctc_beam_search(probs.ctypes.data, lengths.ctypes.data, b, t, k, k - 1, self.beam_width,
                                               1 if self.merge_repeated else 0, self.epsilon, self._scorer,
                                               out.ctypes.data, out_len.ctypes.data, log_prob.ctypes.data, self.threads)
        if rc != 0:
            raise ValueError("sl_host_ctc_beam_search rejected its arguments")
        return [list(map(int, out[i, :out_len[i]])) for i in range(b)], log_prob

    def __del__(self):
        if getattr(self, "_scorer", None):
            self._lib.sl_host_scorer_free(self._scorer)
            self._scorer = None
