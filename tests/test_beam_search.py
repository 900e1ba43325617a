"""CTC beam search + n-gram language model: the C++ decoder of libspeechless_host.so
(speechless_amd/decoder.py) against the reference's own known answers, an independent pure-Python restatement
(oracle/beam_search_oracle.py), brute-force enumeration, and hand-computed back-off probabilities.  Host code: CPU tests."""
import itertools
import math
from pathlib import Path

import numpy as np
import pytest

from oracle import beam_search_oracle as bo

TOY = Path(__file__).resolve().parent / "golden" / "toy_kenlm"
ALPHABET = list("acehost ")  # the toy directory's `vocabulary` line


def softmax_rows(z):
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def test_reference_known_answers_of_the_beam_search_decoder():
    """speechless/test/test_ctc_decoders.py:19-41: logits "A A _ A A" (2 classes), beam width 1:
    merge_repeated=True -> [0], merge_repeated=False -> [0, 0].  (The decoder normalises its input, so feeding
    softmax(logits) through log(p + eps) reproduces tf.nn.ctc_beam_search_decoder(inputs=logits).)"""
    from speechless_amd.decoder import CtcBeamSearchDecoder
    logits = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [1.0, 0.0]], dtype=np.float32)
    probs = softmax_rows(logits)[None]
    for merge, want in ((True, [0]), (False, [0, 0])):
        dec = CtcBeamSearchDecoder(["A"], beam_width=1, merge_repeated=merge, epsilon=0.0)
        got, _ = dec.decode(probs, [5])
        assert got == [want], (merge, got)
        assert bo.ctc_beam_search(bo.normalised_log_probs(probs[0], 0.0), 1, merge)[0] == want


@pytest.mark.parametrize("merge", [True, False])
@pytest.mark.parametrize("beam_width", [1, 4, 25])
def test_plain_beam_search_against_the_python_restatement(merge, beam_width):
    from speechless_amd.decoder import CtcBeamSearchDecoder
    rng = np.random.RandomState(10 * beam_width + merge)
    probs = softmax_rows(rng.randn(6, 30, 6) * 2).astype(np.float32)
    lengths = [30, 17, 30, 1, 22, 9]
    dec = CtcBeamSearchDecoder(list("abcde"), beam_width=beam_width, merge_repeated=merge, threads=3)
    got, log_prob = dec.decode(probs, lengths)
    for b in range(6):
        want, want_lp = bo.ctc_beam_search(bo.normalised_log_probs(probs[b, :lengths[b]]), beam_width, merge)
        assert got[b] == want, (b, got[b], want)
        assert abs(log_prob[b] - want_lp) < 1e-3 * max(1.0, abs(want_lp))


def test_wide_beam_finds_the_most_probable_labelling():
    """With a beam wider than the number of prefixes the search is exact: its best log-probability is the maximum over
    ALL labellings of the summed probability of their alignments (brute force over K^T alignments, standard CTC
    collapse -- which is what merge_repeated=True computes)."""
    from speechless_amd.decoder import CtcBeamSearchDecoder
    rng = np.random.RandomState(3)
    t, k = 7, 3
    probs = softmax_rows(rng.randn(1, t, k) * 1.5).astype(np.float32)
    totals = {}
    for path in itertools.product(range(k), repeat=t):
        collapsed = tuple(l for l, _ in itertools.groupby(path) if l != k - 1)
        totals[collapsed] = totals.get(collapsed, 0.0) + float(np.prod([probs[0, i, c] for i, c in enumerate(path)]))
    best = max(totals, key=totals.get)
    dec = CtcBeamSearchDecoder(["a", "b"], beam_width=5000, merge_repeated=True, epsilon=0.0)
    got, log_prob = dec.decode(probs, [t])
    assert abs(log_prob[0] - math.log(totals[best])) < 1e-4
    if all(x != y for x, y in zip(best, best[1:])):  # LabelSeq(merge_repeated=True) also merges repeats of the OUTPUT
        assert tuple(got[0]) == best


def test_arpa_back_off_scores_by_hand():
    from speechless_amd.decoder import NGramLanguageModel
    lm = NGramLanguageModel(TOY / "lm.arpa")
    ref = bo.ArpaModel(str(TOY / "lm.arpa"))
    assert lm.order == 3 and ref.order == 3
    # "the cat sat": trigram hits all the way: P(the|<s>) + P(cat|<s> the) + P(sat|the cat) + P(</s>|cat sat)
    assert abs(lm.score("the cat sat") - (-0.6 - 0.3 - 0.4 - 0.5)) < 1e-6
    # "the cot": (cot | <s> the): no trigram -> back-off(<s> the) = -0.2, then bigram "the cot" = -1.7;
    #            (</s> | the cot): no trigram, back-off(the cot) = -0.1, no bigram "cot </s>", back-off(cot) = -0.2,
    #            unigram </s> = -1.2
    assert abs(lm.score("the cot") - (-0.6 + (-0.2 - 1.7) + (-0.1 - 0.2 - 1.2))) < 1e-6
    # an out-of-vocabulary word is scored as <unk>: (<unk> | <s>): no bigram, back-off(<s>) = -0.4, unigram -1.5;
    #            (</s> | <s> <unk>): back-off of unseen contexts is 0, then back-off(<unk>) = 0, unigram </s>
    assert abs(lm.score("dog") - ((-0.4 - 1.5) + (-1.2))) < 1e-6
    for sentence in ("the cat sat", "a cat", "the cot", "dog", "a cat sat at the cot", ""):
        assert abs(lm.score(sentence) - ref.score_sentence(sentence)) < 1e-5, sentence
    with pytest.raises(ValueError):
        NGramLanguageModel(TOY / "vocabulary")  # not an ARPA file


def _acoustics(text, alphabet, t_per_char=3, confusions=(), sharpness=4.0, seed=0):
    """frame probabilities spelling `text` (each character held t_per_char frames, a blank between characters);
    confusions: {position: (other character, share)} splits that character's mass with another one."""
    rng = np.random.RandomState(seed)
    k = len(alphabet) + 1
    rows = []
    for pos, ch in enumerate(text):
        for _ in range(t_per_char):
            z = rng.randn(k) * 0.3
            z[alphabet.index(ch)] += sharpness
            for p, (other, share) in dict(confusions).items():
                if p == pos:
                    z[alphabet.index(other)] = z[alphabet.index(ch)] + math.log(share / (1 - share))
            rows.append(z)
        z = rng.randn(k) * 0.3
        z[k - 1] += sharpness
        rows.append(z)
    return softmax_rows(np.array(rows)).astype(np.float32)


def test_language_model_scored_search_against_the_python_restatement():
    from speechless_amd.decoder import CtcBeamSearchDecoder, NGramLanguageModel
    lm = NGramLanguageModel(TOY / "lm.arpa")
    ref_lm = bo.ArpaModel(str(TOY / "lm.arpa"))
    rng = np.random.RandomState(5)
    batches = [softmax_rows(rng.randn(24, len(ALPHABET) + 1) * 2.5).astype(np.float32) for _ in range(4)]
    batches.append(_acoustics("the cat sat", ALPHABET, seed=1))
    batches.append(_acoustics("a cot", ALPHABET, seed=2, sharpness=2.0))
    for weights in ((.8, 0., 2.3), (1.5, 1.0, 0.0)):
        for beam_width in (8, 64):
            dec = CtcBeamSearchDecoder(ALPHABET, lm, beam_width=beam_width, merge_repeated=False, kenlm_weight=weights[0],
                                       word_count_weight=weights[1], valid_word_count_weight=weights[2])
            scorer = bo.Scorer(ref_lm, ALPHABET, *weights)
            for probs in batches:
                got, log_prob = dec.decode(probs[None], [len(probs)])
                want, want_lp = bo.ctc_beam_search(bo.normalised_log_probs(probs), beam_width, False, scorer=scorer)
                assert got[0] == want, ("".join(ALPHABET[i] for i in got[0]), "".join(ALPHABET[i] for i in want))
                assert abs(log_prob[0] - want_lp) < 2e-3 * max(1.0, abs(want_lp))


def test_language_model_changes_the_transcription():
    """Acoustics that slightly prefer "the cot" (o 55 % / a 45 % on one character): greedy and the plain beam search say
    "the cot", the language model (P(cat | the) >> P(cot | the)) turns it into "the cat"."""
    from speechless_amd.decoder import CtcBeamSearchDecoder, NGramLanguageModel
    # one frame per character + a blank: with merge_repeated=False (net.py:447) the decoder also counts a label held over
    # two frames WITHOUT a blank between them as a doubled letter -- the reference's documented choice (net.py:441-443)
    probs = _acoustics("the cot", ALPHABET, t_per_char=1, confusions={5: ("a", 0.45)}, seed=4)
    text = lambda labels: "".join(ALPHABET[i] for i in labels)  # noqa: E731
    plain = CtcBeamSearchDecoder(ALPHABET, beam_width=32, merge_repeated=False)
    assert text(plain.decode(probs[None], [len(probs)])[0][0]) == "the cot"
    with_lm = CtcBeamSearchDecoder(ALPHABET, NGramLanguageModel(TOY / "lm.arpa"), beam_width=32, merge_repeated=False)
    assert text(with_lm.decode(probs[None], [len(probs)])[0][0]) == "the cat"


def test_kenlm_directory_conventions():
    from speechless_amd.decoder import CtcBeamSearchDecoder, expected_characters, find_arpa
    assert expected_characters(TOY) == ALPHABET  # net.py:171-174
    assert find_arpa(TOY).name == "lm.arpa"
    dec = CtcBeamSearchDecoder.from_kenlm_directory(TOY, ALPHABET)
    assert dec.language_model.order == 3 and dec.beam_width == 100 and dec.merge_repeated is False


@pytest.mark.gpu
def test_wav2letter_with_a_kenlm_directory_decodes_through_the_language_model(tmp_path):
    """Wav2Letter(kenlm_directory=...) (net.py:171-177, 444-451): the alphabet check against <dir>/vocabulary, and
    test_and_predict_batch decoding = the beam search over the probabilities the HIP path produced."""
    from speechless_amd import Wav2Letter
    from speechless_amd.decoder import CtcBeamSearchDecoder
    from speechless_amd.net import LabeledSpectrogram
    small = dict(main_filter_count=20, out_filter_count=40, inner_count=1)
    with pytest.raises(ValueError):
        Wav2Letter(128, list("abc "), kenlm_directory=TOY, layer_sizes=small)
    net = Wav2Letter(128, ALPHABET, kenlm_directory=TOY, seed=5, layer_sizes=small, compute_dtype="f32")
    rng = np.random.RandomState(2)
    batch = [LabeledSpectrogram("u{}".format(i), "the cat", rng.randn(90 + 10 * i, 128)) for i in range(3)]
    result = net.test_and_predict_batch(batch)
    probs = net.prediction_batch(np.stack([np.pad(e.z_normalized_transposed_spectrogram(), ((0, 110 - 90 - 10 * i), (0, 0)))
                                           for i, e in enumerate(batch)]))
    dec = CtcBeamSearchDecoder.from_kenlm_directory(TOY, ALPHABET)
    want, _ = dec.decode(probs, [(90 + 10 * i) // 2 for i in range(3)])
    assert [r.predicted for r in result.results] == ["".join(ALPHABET[j] for j in w) for w in want]
    assert all(np.isfinite(r.loss) for r in result.results)
