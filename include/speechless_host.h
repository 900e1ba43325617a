/*
 * speechless_host.h -- C-ABI of libspeechless_host.so: the plain C++ (no HIP) helpers that sit on the HOST side of the
 * speechless Wav2Letter hot path, either side of the GPU kernels of speechless_hip.h.
 *
 * The reference (juliuskunze/speechless) has no native code; each entry point cites the Python / third-party call it
 * replaces (paths relative to the reference root).  Conventions: extern "C", plain pointers and sizes, no exceptions
 * across the boundary, int status 0 = ok / -1 = bad arguments unless stated otherwise, thread-safe (no global state;
 * handles are immutable once created and may be shared between threads).
 */
#ifndef SPEECHLESS_HOST_H
#define SPEECHLESS_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SL_HOST_VERSION 1
int sl_host_version(void);

/* ---- batch packing (speechless/net.py:578-587: zero-pad every spectrogram of a batch to the longest one, cast) -------
 * src[b]: C-contiguous (lengths[b], f) array of float64 (is_f64 != 0, what labeled_example.py produces) or float32;
 * dst: (batch, t_max, f) float32, rows behind lengths[b] zeroed.  n_threads worker threads (clamped to [1, batch]). */
int sl_host_pack_batch(const void* const* src, const int32_t* lengths, int batch, int f, int t_max, int is_f64,
                       float* dst, int n_threads);

/* ---- n-gram language model (speechless/net.py:171-177, 444-451: `kenlm_directory` handed to a TensorFlow build patched
 * with a KenLM scorer; the model is read here from its ARPA text form, INTEGRATION.md says how a KenLM binary becomes one)
 * sl_host_lm_load_arpa: opaque handle, or NULL with a message in err[err_len].
 * sl_host_lm_score_sentence: log10 P(<s> sentence </s>) with back-off, i.e. what kenlm.Model.score(sentence) returns. */
void* sl_host_lm_load_arpa(const char* path, char* err, int err_len);
void sl_host_lm_free(void* lm);
int sl_host_lm_order(void* lm);
double sl_host_lm_score_sentence(void* lm, const char* sentence);

/* ---- in-beam scorer over a loaded model.  alphabet[i] = Unicode code point of grapheme index i (blank excluded; U+0020
 * marks the word boundary); the three weights are net.py:447-450 (kenlm_weight, word_count_weight,
 * valid_word_count_weight).  The model must outlive the scorer. */
void* sl_host_scorer_create(void* lm, const uint32_t* alphabet, int n_graphemes, float lm_weight,
                            float word_count_weight, float valid_word_count_weight);
void sl_host_scorer_free(void* scorer);

/* ---- CTC beam search over a batch (net.py:444-451 -> tf.nn.ctc_beam_search_decoder(top_paths=1)) ------------------------
 * probs: float[batch][t_max][k], the net's output probabilities (net.py:350-357); lengths[b] frames are decoded;
 * out: int32[batch][t_max], -1 behind out_len[b] labels (sparse_to_dense default, net.py:436); out_log_prob: optional
 * float[batch].  scorer: handle of sl_host_scorer_create, or NULL for the plain decoder.  eps: added to a probability
 * before its logarithm (0 = none). */
int sl_host_ctc_beam_search(const float* probs, const int32_t* lengths, int batch, int t_max, int k, int blank,
                            int beam_width, int merge_repeated, float eps, void* scorer, int32_t* out, int32_t* out_len,
                            float* out_log_prob, int n_threads);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHLESS_HOST_H */
