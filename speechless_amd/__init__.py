"""speechless_amd -- MI355X-native implementation of the speechless Wav2Letter hot path (conv stack fwd/bwd + CTC).

Importing the package never touches the GPU; `Wav2Letter` / `Engine` need libspeechless_hip.so (see build.py) and a
ROCm device, and raise instead of falling back to the CPU.
"""
from .grapheme_encoding import CtcGraphemeEncoding, english_frequent_characters, german_frequent_characters  # noqa: F401


def __getattr__(name):
    if name in ("Wav2Letter", "Adam", "LabeledSpectrogram", "ExpectationVsPrediction", "ExpectationsVsPredictions",
                "ExpectationsVsPredictionsInBatches", "ExpectationsVsPredictionsInGroupedBatches"):
        from . import net
        return getattr(net, name)
    if name == "Engine":
        from .engine import Engine
        return Engine
    raise AttributeError(name)
