// beam_search.cpp -- CTC beam-search decoding with an n-gram language model, host side.
//
// Replaces the decoding branch the reference takes when `kenlm_directory` is given (speechless/net.py:444-451):
//     tf.nn.ctc_beam_search_decoder(inputs=log(p + 1e-8) time-major, sequence_length, merge_repeated=False,
//                                   kenlm_directory_path=..., kenlm_weight=.8, word_count_weight=0,
//                                   valid_word_count_weight=2.3)
// of a patched TensorFlow (github.com/timediv/tensorflow-with-kenlm, README.md:17) that is neither vendored nor
// installable here.  Restated from the published algorithms (PARITY UNPINNED beyond the vectors named below):
//   * the search itself is TensorFlow 1.x's CTCBeamSearchDecoder (tensorflow/core/util/ctc/ctc_beam_search.h): a trie
//     of prefixes, per entry {blank, label, total} log-probabilities for the old and the new frame, the `previous`
//     rule for repeated labels, growth of new leaves only from candidates, top-N by new total, LabelSeq(merge_repeated).
//     Pinned by the reference's own test_ctc_decoders.py:19-41 ("A A _ A A", beam width 1 -> [0] with merge_repeated,
//     [0, 0] without) -- tests/test_beam_search.py;
//   * the scorer follows the KenLM beam scorer of that TensorFlow fork's lineage (later published as Mozilla
//     DeepSpeech 0.1's KenLMBeamScorer, same three weights): characters extend an incomplete word that is scored by the
//     WORST unigram under its vocabulary-trie node (out-of-vocabulary prefixes by the <unk> unigram); a space scores
//     the finished word with the n-gram model (log10, state = preceding words), adds word_count_weight and, for an
//     in-vocabulary word, valid_word_count_weight; the end of the utterance scores a pending word and </s>;
//     expansion score = previous + lm_weight * (score - previous score);
//   * the language model is read from an ARPA file (KenLM's binary format needs KenLM itself) and queried with
//     standard back-off -- what KenLM's FullScore().prob returns.
// Plain C++ (graph search on the host, not roofline work); utterances of a batch are decoded on separate threads.
#include "../../include/speechless_host.h"  // the C-ABI this file implements (checked by the compiler)

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

constexpr float kLogZero = -std::numeric_limits<float>::infinity();

inline float log_sum_exp(float a, float b) {
    if (a == kLogZero) return b;
    if (b == kLogZero) return a;
    return a > b ? a + log1pf(expf(b - a)) : b + log1pf(expf(a - b));
}

// ---------------------------------------------------------------------------------------------- n-gram model (ARPA)
struct NGramModel {
    int order = 0;
    std::unordered_map<std::string, uint32_t> vocab;      // word -> id; 0 = <unk>
    std::vector<std::string> words;
    // key = the n-gram's word ids as raw bytes
    std::unordered_map<std::string, std::pair<float, float>> grams;  // -> (log10 prob, log10 back-off)
    uint32_t bos = 0, eos = 0;
    float unk_logprob = -100.f;

    static std::string key(const uint32_t* ids, size_t n) { return std::string((const char*)ids, n * sizeof(uint32_t)); }

    uint32_t index(const std::string& w) const {
        auto it = vocab.find(w);
        return it == vocab.end() ? 0u : it->second;
    }
    uint32_t intern(const std::string& w) {
        auto it = vocab.find(w);
        if (it != vocab.end()) return it->second;
        const uint32_t id = (uint32_t)words.size();
        vocab.emplace(w, id);
        words.push_back(w);
        return id;
    }

    // log10 P(word | history) with back-off; history = preceding word ids, oldest first (any length)
    float score(const std::vector<uint32_t>& history, uint32_t word) const {
        std::vector<uint32_t> ctx(history);
        if ((int)ctx.size() > order - 1) ctx.erase(ctx.begin(), ctx.end() - (order - 1));
        float backoff = 0.f;
        for (;;) {
            std::vector<uint32_t> gram(ctx);
            gram.push_back(word);
            auto it = grams.find(key(gram.data(), gram.size()));
            if (it != grams.end()) return backoff + it->second.first;
            if (ctx.empty()) return backoff + unk_logprob;  // a word without a unigram: scored as <unk>
            auto bo = grams.find(key(ctx.data(), ctx.size()));
            if (bo != grams.end()) backoff += bo->second.second;
            ctx.erase(ctx.begin());
        }
    }
    std::vector<uint32_t> advance(const std::vector<uint32_t>& history, uint32_t word) const {
        std::vector<uint32_t> out(history);
        out.push_back(word);
        if ((int)out.size() > order - 1) out.erase(out.begin(), out.end() - (order - 1));
        return out;
    }
};

bool load_arpa(const std::string& path, NGramModel* m, std::string* err) {
    std::ifstream in(path);
    if (!in) {
        *err = "cannot open " + path;
        return false;
    }
    m->intern("<unk>");
    std::string line;
    int section = -1;  // -1 before \data\, 0 in \data\, n in \n-grams:
    bool seen_data = false;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        if (line.empty()) continue;
        if (line == "\\data\\") {
            section = 0;
            seen_data = true;
            continue;
        }
        if (line == "\\end\\") break;
        if (line[0] == '\\') {
            int n = 0;
            if (sscanf(line.c_str(), "\\%d-grams:", &n) == 1 && n > 0) {
                section = n;
                m->order = std::max(m->order, n);
                continue;
            }
            *err = "unexpected ARPA section '" + line + "'";
            return false;
        }
        if (section <= 0) continue;  // "ngram N=count" lines
        std::vector<std::string> fields;
        {
            std::istringstream ss(line);
            std::string f;
            while (ss >> f) fields.push_back(f);
        }
        if ((int)fields.size() < section + 1 || (int)fields.size() > section + 2) {
            *err = "malformed " + std::to_string(section) + "-gram line '" + line + "'";
            return false;
        }
        const float logprob = strtof(fields[0].c_str(), nullptr);
        const float backoff = (int)fields.size() == section + 2 ? strtof(fields[section + 1].c_str(), nullptr) : 0.f;
        std::vector<uint32_t> ids(section);
        for (int i = 0; i < section; ++i) ids[i] = section == 1 ? m->intern(fields[1 + i]) : m->index(fields[1 + i]);
        m->grams[NGramModel::key(ids.data(), ids.size())] = std::make_pair(logprob, backoff);
    }
    if (!seen_data || m->order == 0) {
        *err = path + " is not an ARPA language model (KenLM binary files are not readable without KenLM)";
        return false;
    }
    m->bos = m->index("<s>");
    m->eos = m->index("</s>");
    const uint32_t unk = 0;
    auto it = m->grams.find(NGramModel::key(&unk, 1));
    if (it != m->grams.end()) m->unk_logprob = it->second.first;
    return true;
}

// ---------------------------------------------------------------------------------------------- vocabulary trie
struct TrieNode {
    std::map<int, std::unique_ptr<TrieNode>> children;  // by grapheme index
    float min_unigram_score = std::numeric_limits<float>::max();
};

struct Scorer {
    const NGramModel* lm = nullptr;
    std::vector<uint32_t> alphabet;  // code point per grapheme index (blank excluded)
    int space_label = -1;
    float lm_weight = 0.f, word_count_weight = 0.f, valid_word_count_weight = 0.f;
    TrieNode root;
    float oov_score = -100.f;

    static void append_utf8(std::string* s, uint32_t cp) {
        if (cp < 0x80) {
            s->push_back((char)cp);
        } else if (cp < 0x800) {
            s->push_back((char)(0xC0 | (cp >> 6)));
            s->push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s->push_back((char)(0xE0 | (cp >> 12)));
            s->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s->push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            s->push_back((char)(0xF0 | (cp >> 18)));
            s->push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            s->push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s->push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    static std::vector<uint32_t> decode_utf8(const std::string& s) {
        std::vector<uint32_t> out;
        for (size_t i = 0; i < s.size();) {
            const unsigned char c = (unsigned char)s[i];
            uint32_t cp;
            int n;
            if (c < 0x80) { cp = c; n = 1; }
            else if ((c >> 5) == 6) { cp = c & 0x1F; n = 2; }
            else if ((c >> 4) == 14) { cp = c & 0x0F; n = 3; }
            else { cp = c & 0x07; n = 4; }
            for (int j = 1; j < n && i + j < s.size(); ++j) cp = (cp << 6) | ((unsigned char)s[i + j] & 0x3F);
            out.push_back(cp);
            i += n;
        }
        return out;
    }

    void build_trie() {
        std::unordered_map<uint32_t, int> label_of;
        for (size_t i = 0; i < alphabet.size(); ++i) label_of[alphabet[i]] = (int)i;
        oov_score = lm->unk_logprob;
        for (uint32_t id = 1; id < lm->words.size(); ++id) {
            const std::string& w = lm->words[id];
            if (w == "<s>" || w == "</s>") continue;
            auto it = lm->grams.find(NGramModel::key(&id, 1));
            if (it == lm->grams.end()) continue;
            const float unigram = it->second.first;
            TrieNode* node = &root;
            bool ok = true;
            std::vector<TrieNode*> path;
            for (uint32_t cp : decode_utf8(w)) {
                auto l = label_of.find(cp);
                if (l == label_of.end() || l->second == space_label) { ok = false; break; }
                auto& child = node->children[l->second];
                if (!child) child.reset(new TrieNode());
                node = child.get();
                path.push_back(node);
            }
            if (!ok) continue;  // a word the alphabet cannot spell never matches a prefix
            for (TrieNode* n : path) n->min_unigram_score = std::min(n->min_unigram_score, unigram);
        }
    }
};

struct BeamState {
    float language_model_score = 0.f, score = 0.f, delta_score = 0.f;
    std::string incomplete_word;
    const TrieNode* trie_node = nullptr;
    std::vector<uint32_t> model_state;
};

struct BeamProbability {
    float total = kLogZero, blank = kLogZero, label = kLogZero;
    void reset() { total = blank = label = kLogZero; }
};

struct BeamEntry {
    BeamEntry* parent = nullptr;
    int label = -1;
    std::map<int, BeamEntry*> children;
    BeamProbability oldp, newp;
    BeamState state;
    bool active() const { return newp.total != kLogZero; }
};

struct Decoder {
    const Scorer* scorer;  // may be null: plain CTC beam search
    int num_classes, blank, beam_width;
    bool merge_repeated;
    std::deque<BeamEntry> arena;
    std::vector<BeamEntry*> leaves;
    BeamEntry* root = nullptr;

    void expand_state(const BeamState& from, BeamState* to, int to_label) const {
        *to = from;
        if (!scorer) return;
        if (to_label != scorer->space_label) {
            Scorer::append_utf8(&to->incomplete_word, scorer->alphabet[to_label]);
            float min_unigram = scorer->oov_score;
            const TrieNode* node = from.trie_node;
            if (node != nullptr) {
                auto it = node->children.find(to_label);
                node = it == node->children.end() ? nullptr : it->second.get();
                to->trie_node = node;
                if (node != nullptr) min_unigram = node->min_unigram_score;
            }
            to->score = min_unigram + to->language_model_score;
            to->delta_score = to->score - from.score;
        } else {
            const uint32_t word = scorer->lm->index(to->incomplete_word);
            const float lm_delta = scorer->lm->score(from.model_state, word);
            to->model_state = scorer->lm->advance(from.model_state, word);
            if (word != 0) to->language_model_score += scorer->valid_word_count_weight;
            to->language_model_score += scorer->word_count_weight;
            update_with_lm_score(to, lm_delta);
            to->incomplete_word.clear();
            to->trie_node = &scorer->root;
        }
    }
    static void update_with_lm_score(BeamState* s, float delta) {
        const float previous = s->score;
        s->language_model_score += delta;
        s->score = s->language_model_score;
        s->delta_score = s->language_model_score - previous;
    }
    void expand_state_end(BeamState* s) const {
        if (!scorer) return;
        float delta = 0.f;
        if (!s->incomplete_word.empty()) {
            const uint32_t word = scorer->lm->index(s->incomplete_word);
            delta += scorer->lm->score(s->model_state, word);
            s->model_state = scorer->lm->advance(s->model_state, word);
            s->incomplete_word.clear();
            s->trie_node = &scorer->root;
        }
        delta += scorer->lm->score(s->model_state, scorer->lm->eos);
        update_with_lm_score(s, delta);
    }
    float expansion_score(const BeamState& s, float previous) const {
        return scorer ? scorer->lm_weight * s.delta_score + previous : previous;
    }
    float end_expansion_score(const BeamState& s) const { return scorer ? scorer->lm_weight * s.delta_score : 0.f; }

    BeamEntry* child(BeamEntry* b, int label) {
        auto it = b->children.find(label);
        if (it != b->children.end()) return it->second;
        arena.emplace_back();
        BeamEntry* c = &arena.back();
        c->parent = b;
        c->label = label;
        b->children[label] = c;
        return c;
    }
    BeamEntry* bottom() const {
        BeamEntry* worst = leaves[0];
        for (BeamEntry* e : leaves)
            if (e->newp.total < worst->newp.total) worst = e;
        return worst;
    }
    bool is_candidate(const BeamProbability& p) const {
        return p.total > kLogZero && ((int)leaves.size() < beam_width || p.total > bottom()->newp.total);
    }
    void push(BeamEntry* e) {
        if ((int)leaves.size() < beam_width) {
            leaves.push_back(e);
            return;
        }
        BeamEntry* worst = bottom();
        if (e->newp.total > worst->newp.total) *std::find(leaves.begin(), leaves.end(), worst) = e;
    }

    void reset() {
        arena.clear();
        leaves.clear();
        arena.emplace_back();
        root = &arena.back();
        root->newp.total = 0.f;  // log 1
        root->newp.blank = 0.f;
        if (scorer) {
            root->state.trie_node = &scorer->root;
            root->state.model_state.assign(1, scorer->lm->bos);
        }
        leaves.push_back(root);
    }

    void step(const float* input) {  // input: normalised log-probabilities of one frame
        std::vector<BeamEntry*> branches(leaves);
        std::stable_sort(branches.begin(), branches.end(),
                         [](const BeamEntry* a, const BeamEntry* b) { return a->newp.total > b->newp.total; });
        leaves.clear();
        for (BeamEntry* b : branches) b->oldp = b->newp;
        for (BeamEntry* b : branches) {
            if (b->parent != nullptr) {
                if (b->parent->active()) {
                    const float previous = (merge_repeated && b->label == b->parent->label) ? b->parent->oldp.blank
                                                                                            : b->parent->oldp.total;
                    b->newp.label = log_sum_exp(b->newp.label, expansion_score(b->state, previous));
                }
                b->newp.label += input[b->label];
            }
            b->newp.blank = b->oldp.total + input[blank];
            b->newp.total = log_sum_exp(b->newp.blank, b->newp.label);
            push(b);
        }
        for (BeamEntry* b : branches) {
            if (!is_candidate(b->oldp)) continue;
            for (int ind = 0; ind < num_classes; ++ind) {
                if (ind == blank) continue;
                BeamEntry* c = child(b, ind);
                if (c->active()) continue;
                c->newp.blank = kLogZero;
                expand_state(b->state, &c->state, ind);
                const float previous = (merge_repeated && c->label == b->label) ? b->oldp.blank : b->oldp.total;
                c->newp.label = input[ind] + expansion_score(c->state, previous);
                c->newp.total = c->newp.label;
                if (is_candidate(c->newp)) {
                    if ((int)leaves.size() == beam_width) bottom()->newp.reset();  // it leaves the beam
                    push(c);
                } else {
                    c->oldp.reset();
                    c->newp.reset();
                }
            }
        }
    }

    // best path after the last frame: label sequence (LabelSeq(merge_repeated)) and its log-probability
    float best(std::vector<int>* labels) {
        BeamEntry* top = nullptr;
        float top_score = kLogZero;
        for (BeamEntry* e : leaves) {
            expand_state_end(&e->state);
            e->newp.total += end_expansion_score(e->state);
            if (top == nullptr || e->newp.total > top_score) {
                top = e;
                top_score = e->newp.total;
            }
        }
        labels->clear();
        int prev = -1;
        for (const BeamEntry* c = top; c != nullptr && c->parent != nullptr; c = c->parent) {
            if (!merge_repeated || c->label != prev) labels->push_back(c->label);
            prev = c->label;
        }
        std::reverse(labels->begin(), labels->end());
        return top_score;
    }
};

void decode_range(const float* probs, const int32_t* lengths, int b0, int b1, int t_max, int k, int blank, int beam_width,
                  int merge_repeated, float eps, const Scorer* scorer, int32_t* out, int32_t* out_len,
                  float* out_log_prob) {
    Decoder d{scorer, k, blank, beam_width, merge_repeated != 0};
    std::vector<float> frame(k);
    std::vector<int> labels;
    for (int b = b0; b < b1; ++b) {
        d.reset();
        const int len = std::min(lengths[b], t_max);
        for (int t = 0; t < len; ++t) {
            const float* p = probs + ((size_t)b * t_max + t) * k;
            // net.py:430: log(p + 1e-8), then the decoder's own normalisation (log-softmax of its input)
            float mx = kLogZero;
            for (int j = 0; j < k; ++j) {
                frame[j] = logf(p[j] + eps);
                mx = std::max(mx, frame[j]);
            }
            float sum = 0.f;
            for (int j = 0; j < k; ++j) sum += expf(frame[j] - mx);
            const float norm = mx + logf(sum);
            for (int j = 0; j < k; ++j) frame[j] -= norm;
            d.step(frame.data());
        }
        const float score = d.best(&labels);
        int32_t* row = out + (size_t)b * t_max;
        const int n = std::min<int>((int)labels.size(), t_max);
        for (int i = 0; i < n; ++i) row[i] = labels[i];
        for (int i = n; i < t_max; ++i) row[i] = -1;  // sparse_to_dense default (net.py:436)
        out_len[b] = n;
        if (out_log_prob) out_log_prob[b] = score;
    }
}

void set_err(char* err, int err_len, const std::string& msg) {
    if (err && err_len > 0) {
        strncpy(err, msg.c_str(), err_len - 1);
        err[err_len - 1] = 0;
    }
}

}  // namespace

extern "C" {

// Loads an ARPA n-gram model; returns an opaque handle or NULL (message in err).
void* sl_host_lm_load_arpa(const char* path, char* err, int err_len) {
    std::unique_ptr<NGramModel> m(new NGramModel());
    std::string msg;
    if (!load_arpa(path, m.get(), &msg)) {
        set_err(err, err_len, msg);
        return nullptr;
    }
    return m.release();
}

void sl_host_lm_free(void* lm) { delete static_cast<NGramModel*>(lm); }

int sl_host_lm_order(void* lm) { return static_cast<NGramModel*>(lm)->order; }

// log10 probability of a whitespace-separated sentence between <s> and </s> (what `kenlm.Model.score(sentence)` returns)
double sl_host_lm_score_sentence(void* lm, const char* sentence) {
    const NGramModel* m = static_cast<NGramModel*>(lm);
    std::vector<uint32_t> state(1, m->bos);
    std::istringstream ss(sentence);
    std::string w;
    double total = 0;
    while (ss >> w) {
        const uint32_t id = m->index(w);
        total += m->score(state, id);
        state = m->advance(state, id);
    }
    return total + m->score(state, m->eos);
}

// Scorer over a loaded model: alphabet = code point of every grapheme index (blank excluded; U+0020 marks the word
// boundary), weights as in net.py:447-450.  Returns an opaque handle; the model must outlive it.
void* sl_host_scorer_create(void* lm, const uint32_t* alphabet, int n_graphemes, float lm_weight, float word_count_weight,
                            float valid_word_count_weight) {
    std::unique_ptr<Scorer> s(new Scorer());
    s->lm = static_cast<NGramModel*>(lm);
    s->alphabet.assign(alphabet, alphabet + n_graphemes);
    for (int i = 0; i < n_graphemes; ++i)
        if (alphabet[i] == 0x20) s->space_label = i;
    s->lm_weight = lm_weight;
    s->word_count_weight = word_count_weight;
    s->valid_word_count_weight = valid_word_count_weight;
    s->build_trie();
    return s.release();
}

void sl_host_scorer_free(void* scorer) { delete static_cast<Scorer*>(scorer); }

// CTC beam search over a batch.  probs: float[B][t_max][k] (the net's output probabilities, net.py:350-357);
// lengths[b] frames are decoded; out: int32[B][t_max] filled with -1 behind out_len[b] labels; out_log_prob optional.
// scorer: handle of sl_host_scorer_create or NULL (plain tf.nn.ctc_beam_search_decoder semantics).
int sl_host_ctc_beam_search(const float* probs, const int32_t* lengths, int batch, int t_max, int k, int blank,
                            int beam_width, int merge_repeated, float eps, void* scorer, int32_t* out, int32_t* out_len,
                            float* out_log_prob, int n_threads) {
    if (!probs || !lengths || !out || !out_len || batch <= 0 || t_max <= 0 || k < 2 || blank < 0 || blank >= k ||
        beam_width < 1)
        return -1;
    const Scorer* sc = static_cast<const Scorer*>(scorer);
    if (sc && (int)sc->alphabet.size() != k - 1) return -1;
    n_threads = std::max(1, std::min(n_threads, batch));
    std::vector<std::thread> pool;
    for (int i = 0; i < n_threads; ++i) {
        const int b0 = (int)((long)batch * i / n_threads), b1 = (int)((long)batch * (i + 1) / n_threads);
        pool.emplace_back(decode_range, probs, lengths, b0, b1, t_max, k, blank, beam_width, merge_repeated, eps, sc, out,
                          out_len, out_log_prob);
    }
    for (auto& t : pool) t.join();
    return 0;
}

}  // extern "C"
