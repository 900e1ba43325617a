"""Label codec at the boundary of the hot path (host logic).

Mirrors the interface of the reference's speechless/grapheme_enconding.py (sic): `CtcGraphemeEncoding` with
`encode`, `encode_label_batch`, `decode_graphemes`, `decode_grapheme_batch`, `decode_prediction_batch`
(grapheme_enconding.py:8-61,121-137).  Behaviour pinned by the reference's own tests
(speechless/test/test_grapheme_encoding.py:9-31) via tests/golden/codec_golden.json.

The ASG half of the reference file is out of scope: the reference's ASG loss raises NotImplementedError
(net.py:396-399), so no model can use it.
"""
import numpy as np

english_frequent_characters = list("abcdefghijklmnopqrstuvwxyz '")  # english_corpus.py:19 (28 characters)
german_frequent_characters = english_frequent_characters + list("äöüß")  # german_corpus.py:14 (32 characters)


class CtcGraphemeEncoding:
    def __init__(self, allowed_characters):
        self.allowed_characters = list(allowed_characters)
        self.allowed_character_count = len(self.allowed_characters)
        self.grapheme_set_size = self.allowed_character_count + 1
        self.ctc_blank = self.grapheme_set_size - 1  # blank is the LAST index (tf.nn.ctc_loss convention)
        self.graphemes_by_character = {c: i for i, c in enumerate(self.allowed_characters)}
        # code point -> index table for the vectorised batch encoder (-1 = not allowed)
        self._table = np.full(max(ord(c) for c in self.allowed_characters) + 1, -1, dtype=np.int32)
        for c, i in self.graphemes_by_character.items():
            self._table[ord(c)] = i

    def encode_character(self, label_char):
        index = self.graphemes_by_character.get(label_char)
        if index is None:
            raise ValueError("Unexpected char: '{}'".format(label_char))
        return index

    def encode(self, label):
        return [self.encode_character(c) for c in label]

    def encode_label_batch(self, labels):
        """int32 (B, Lmax), padded with -1 (never read: the CTC op is told the true lengths)."""
        width = max(len(label) for label in labels)
        batch = np.full((len(labels), width), -1, dtype=np.int32)
        for row, label in zip(batch, labels):
            if not label:
                continue
            codes = np.frombuffer(label.encode("utf-32-le"), dtype=np.uint32)  # one table lookup per label, not per char
            known = codes < self._table.size
            indices = self._table[np.where(known, codes, 0)]
            if not known.all() or (indices < 0).any():
                bad = int(np.argmax(~known | (indices < 0)))
                raise ValueError("Unexpected char: '{}'".format(label[bad]))
            row[:len(label)] = indices
        return batch

    def decode_grapheme(self, grapheme, previous_grapheme=None):
        if 0 <= grapheme < self.allowed_character_count:
            return self.allowed_characters[grapheme]
        if grapheme == self.ctc_blank:
            return ""
        raise ValueError("Unexpected grapheme: '{}'".format(grapheme))

    def decode_graphemes(self, graphemes, merge_repeated=True):
        out = []
        previous = None
        for g in graphemes:
            g = int(g)
            if not (merge_repeated and g == previous):
                out.append(self.decode_grapheme(g))
            previous = g
        return "".join(out)

    def decode_grapheme_batch(self, grapheme_batch, prediction_lengths, merge_repeated=True):
        return [self.decode_graphemes(list(grapheme_batch[i])[:int(prediction_lengths[i])],
                                      merge_repeated=merge_repeated)
                for i in range(len(grapheme_batch))]

    def decode_prediction_batch(self, prediction_batch, prediction_lengths):
        """prediction_batch: (B, T', K) probabilities -> greedy strings (argmax, first max wins)."""
        return self.decode_grapheme_batch(np.argmax(prediction_batch, 2), prediction_lengths)
