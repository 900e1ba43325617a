"""Pure-Python restatement of the reference's language-model decoding branch (TEST INFRASTRUCTURE ONLY).

Only tests/ may import this; the product path is speechless_amd/decoder.py -> csrc_host/beam_search.cpp.

What it follows: speechless/net.py:444-451 calls tf.nn.ctc_beam_search_decoder(log(p + 1e-8), merge_repeated=False,
kenlm_directory_path, kenlm_weight=.8, word_count_weight=0, valid_word_count_weight=2.3) of a patched TensorFlow
(github.com/timediv/tensorflow-with-kenlm) that is not vendored.  Restated here, independently of the C++ (dict-based
trie, tuples instead of entry objects with pointers), from the published algorithms:
  * TensorFlow 1.x CTCBeamSearchDecoder::Step / TopPaths / LabelSeq (tensorflow/core/util/ctc/ctc_beam_search.h);
  * the KenLM beam scorer of that fork's lineage (published later as Mozilla DeepSpeech 0.1's KenLMBeamScorer);
  * ARPA back-off n-gram scoring (= KenLM FullScore().prob).
PINNED by speechless/test/test_ctc_decoders.py:19-41 (beam width 1 on "A A _ A A": [0] with merge_repeated, [0, 0]
without); everything involving the language model is PARITY UNPINNED (the fork is unavailable, the reference holds no
vectors for it).
"""
import math

LOG_ZERO = float("-inf")


def log_sum_exp(a, b):
    if a == LOG_ZERO:
        return b
    if b == LOG_ZERO:
        return a
    return max(a, b) + math.log1p(math.exp(-abs(a - b)))


# ------------------------------------------------------------------------------------------------- ARPA language model
class ArpaModel:
    def __init__(self, path):
        self.grams = {}  # tuple of words -> (log10 prob, log10 back-off)
        self.order = 0
        section = 0
        for line in open(path, encoding="utf8"):
            line = line.strip()
            if not line or line == "\\data\\" or line.startswith("ngram "):
                continue
            if line == "\\end\\":
                break
            if line.startswith("\\"):
                section = int(line[1:].split("-")[0])
                self.order = max(self.order, section)
                continue
            fields = line.split()
            words = tuple(fields[1:1 + section])
            backoff = float(fields[1 + section]) if len(fields) > 1 + section else 0.0
            self.grams[words] = (float(fields[0]), backoff)
        self.unk = self.grams.get(("<unk>",), (-100.0, 0.0))[0]

    def known(self, word):
        return (word,) in self.grams

    def score(self, history, word):
        """log10 P(word | history) -- history: tuple of preceding words."""
        if not self.known(word):
            word = "<unk>"
        ctx = tuple(history)[-(self.order - 1):] if self.order > 1 else ()
        backoff = 0.0
        while True:
            hit = self.grams.get(ctx + (word,))
            if hit is not None:
                return backoff + hit[0]
            if not ctx:
                return backoff + self.unk
            backoff += self.grams.get(ctx, (0.0, 0.0))[1]
            ctx = ctx[1:]

    def advance(self, history, word):
        if not self.known(word):
            word = "<unk>"
        state = tuple(history) + (word,)
        return state[-(self.order - 1):] if self.order > 1 else ()

    def score_sentence(self, sentence):
        state, total = ("<s>",), 0.0
        for w in sentence.split():
            total += self.score(state, w)
            state = self.advance(state, w)
        return total + self.score(state, "</s>")


# ------------------------------------------------------------------------------------------------- scorer
class Scorer:
    """state = (language_model_score, score, delta_score, incomplete_word, trie prefix or None, model history)"""

    def __init__(self, model, alphabet, lm_weight, word_count_weight, valid_word_count_weight):
        self.model, self.alphabet = model, list(alphabet)
        self.space = self.alphabet.index(" ") if " " in self.alphabet else -1
        self.w, self.wc, self.vwc = lm_weight, word_count_weight, valid_word_count_weight
        self.prefix_min = {}  # word prefix -> lowest unigram log-probability of any vocabulary word starting with it
        letters = set(self.alphabet) - {" "}
        for words, (logp, _) in model.grams.items():
            if len(words) == 1 and words[0] not in ("<s>", "</s>", "<unk>") and set(words[0]) <= letters:
                for n in range(1, len(words[0]) + 1):
                    p = words[0][:n]
                    self.prefix_min[p] = min(self.prefix_min.get(p, float("inf")), logp)

    def initial(self):
        return (0.0, 0.0, 0.0, "", "", ("<s>",))

    def _with_lm(self, lm_score, score, delta):
        lm_score = lm_score + delta
        return lm_score, lm_score, lm_score - score

    def expand(self, state, label):
        lm_score, score, _, word, prefix, history = state
        if label != self.space:
            word = word + self.alphabet[label]
            unigram = self.model.unk
            if prefix is not None:
                prefix = prefix + self.alphabet[label]
                if prefix in self.prefix_min:
                    unigram = self.prefix_min[prefix]
                else:
                    prefix = None
            new_score = unigram + lm_score
            return (lm_score, new_score, new_score - score, word, prefix, history)
        delta = self.model.score(history, word)
        if self.model.known(word) and word != "<unk>":
            lm_score += self.vwc
        lm_score += self.wc
        lm_score, new_score, d = self._with_lm(lm_score, score, delta)
        return (lm_score, new_score, d, "", "", self.model.advance(history, word))

    def expand_end(self, state):
        lm_score, score, _, word, prefix, history = state
        delta = 0.0
        if word:
            delta += self.model.score(history, word)
            history = self.model.advance(history, word)
        delta += self.model.score(history, "</s>")
        lm_score, new_score, d = self._with_lm(lm_score, score, delta)
        return (lm_score, new_score, d, "", "", history)

    def expansion_score(self, state, previous):
        return self.w * state[2] + previous

    def end_expansion_score(self, state):
        return self.w * state[2]


# ------------------------------------------------------------------------------------------------- beam search
class _Entry:
    __slots__ = ("parent", "label", "children", "old", "new", "state")

    def __init__(self, parent, label):
        self.parent, self.label, self.children = parent, label, {}
        self.old = [LOG_ZERO, LOG_ZERO, LOG_ZERO]  # total, blank, label
        self.new = [LOG_ZERO, LOG_ZERO, LOG_ZERO]
        self.state = None


def ctc_beam_search(log_probs, beam_width=100, merge_repeated=True, blank=None, scorer=None):
    """log_probs: T x K nested lists of NORMALISED log-probabilities.  Returns (labels, log-probability)."""
    k = len(log_probs[0]) if len(log_probs) else 0
    blank = k - 1 if blank is None else blank
    root = _Entry(None, -1)
    root.new = [0.0, 0.0, LOG_ZERO]
    root.state = scorer.initial() if scorer else None
    leaves = [root]

    def bottom():
        return min(leaves, key=lambda e: e.new[0])

    def candidate(total):
        return total > LOG_ZERO and (len(leaves) < beam_width or total > bottom().new[0])

    def push(e):
        if len(leaves) < beam_width:
            leaves.append(e)
        else:
            worst = bottom()
            if e.new[0] > worst.new[0]:
                leaves[leaves.index(worst)] = e

    for frame in log_probs:
        branches = sorted(leaves, key=lambda e: -e.new[0])
        del leaves[:]
        for b in branches:
            b.old = list(b.new)
        for b in branches:
            if b.parent is not None:
                if b.parent.new[0] != LOG_ZERO:
                    previous = b.parent.old[1] if (merge_repeated and b.label == b.parent.label) else b.parent.old[0]
                    if scorer:
                        previous = scorer.expansion_score(b.state, previous)
                    b.new[2] = log_sum_exp(b.new[2], previous)
                b.new[2] += frame[b.label]
            b.new[1] = b.old[0] + frame[blank]
            b.new[0] = log_sum_exp(b.new[1], b.new[2])
            push(b)
        for b in branches:
            if not candidate(b.old[0]):
                continue
            for ind in range(k):
                if ind == blank:
                    continue
                c = b.children.get(ind)
                if c is None:
                    c = b.children[ind] = _Entry(b, ind)
                if c.new[0] != LOG_ZERO:
                    continue
                c.state = scorer.expand(b.state, ind) if scorer else None
                previous = b.old[1] if (merge_repeated and ind == b.label) else b.old[0]
                if scorer:
                    previous = scorer.expansion_score(c.state, previous)
                c.new = [frame[ind] + previous, LOG_ZERO, frame[ind] + previous]
                if candidate(c.new[0]):
                    if len(leaves) == beam_width:
                        bottom().new = [LOG_ZERO, LOG_ZERO, LOG_ZERO]
                    push(c)
                else:
                    c.old = [LOG_ZERO, LOG_ZERO, LOG_ZERO]
                    c.new = [LOG_ZERO, LOG_ZERO, LOG_ZERO]
    best, best_score = None, LOG_ZERO
    for e in leaves:
        total = e.new[0]
        if scorer:
            total += scorer.end_expansion_score(scorer.expand_end(e.state))
        if best is None or total > best_score:
            best, best_score = e, total
    labels, prev, c = [], -1, best
    while c is not None and c.parent is not None:
        if not merge_repeated or c.label != prev:
            labels.append(c.label)
        prev = c.label
        c = c.parent
    return labels[::-1], best_score


def normalised_log_probs(probs, eps=1e-8):
    """net.py:430 + the decoder's own normalisation: log_softmax(log(p + eps)) per frame."""
    out = []
    for frame in probs:
        logs = [math.log(float(p) + eps) for p in frame]
        m = max(logs)
        norm = m + math.log(sum(math.exp(v - m) for v in logs))
        out.append([v - norm for v in logs])
    return out
