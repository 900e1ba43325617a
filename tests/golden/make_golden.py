"""Generates tests/golden/*.json|npz.  Run in the BUILD container only (needs /root/reference for the codec vectors):

    python tests/golden/make_golden.py

* codec_golden.json  -- outputs of the REFERENCE's own speechless.grapheme_enconding (importable: numpy only) on the
  inputs of speechless/test/test_grapheme_encoding.py plus a few more, and the known-answer vector of
  speechless/test/test_ctc_decoders.py:19-41 (TF needed to *run* that test; the input/output pair is data).
* stack_golden.npz   -- restatement-generated vectors (oracle/w2l_oracle.py, float64) for a shrunken stack and for the
  real topology at B=2,T=64: inputs, labels, probs, per-utterance losses, decoded indices, gradient slices/norms.
  Conv/CTC numerics are parity-unpinned by the reference; these pin the ORACLE against drift.
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))


def codec_vectors():
    sys.path.insert(0, "/root/reference")
    from speechless.grapheme_enconding import CtcGraphemeEncoding  # the reference itself
    import string
    english = list(string.ascii_lowercase + " '")  # english_corpus.py:19 (that module itself needs `lazy`)
    german = english + list("äöüß")
    out = {"english_frequent_characters": english, "german_frequent_characters": german, "cases": []}
    for name, chars in (("english", english), ("german", german)):
        g = CtcGraphemeEncoding(chars)
        labels = ["she wasn't three abcxyz", "she wasn't", "abc", "a", "", "zz top's"]
        if name == "german":
            labels += ["größe über äpfel"]
        case = {"alphabet": name, "grapheme_set_size": g.grapheme_set_size, "ctc_blank": g.ctc_blank,
                "encode": {l: g.encode(l) for l in labels},
                "encode_label_batch": {"labels": ["abc", "a"],
                                       "result": g.encode_label_batch(["abc", "a"]).tolist()}}
        graphemes = g.encode("sssshhhheeeee      wasn't thre") + [g.ctc_blank] + g.encode("eeeeee")
        case["decode_graphemes"] = [
            {"graphemes": graphemes, "merge_repeated": True, "result": g.decode_graphemes(graphemes)},
            {"graphemes": graphemes, "merge_repeated": False,
             "result": g.decode_graphemes(graphemes, merge_repeated=False)}]
        rng = np.random.RandomState(7)
        preds = rng.rand(3, 9, g.grapheme_set_size)
        lengths = [9, 5, 0]
        case["decode_prediction_batch"] = {"predictions": preds.tolist(), "prediction_lengths": lengths,
                                           "result": g.decode_prediction_batch(preds, prediction_lengths=lengths)}
        predictions = np.zeros((2, 3, g.grapheme_set_size))
        for b in range(2):
            for t, c in enumerate("abc"):
                predictions[b, t, g.encode_character(c)] = 1
        case["test_encode_batch"] = {"result": g.decode_prediction_batch(predictions, prediction_lengths=[3, 2])}
        out["cases"].append(case)
    # test_ctc_decoders.py:22-24,40: logits of "A A _ A A" (2 classes, blank = 1), greedy merge_repeated=True -> [0, 0]
    out["tf_greedy_kat"] = {"logits_t_k": [[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [1.0, 0.0], [1.0, 0.0]],
                            "greedy_merge_repeated": [0, 0], "greedy_no_merge": [0, 0, 0, 0]}
    (HERE / "codec_golden.json").write_text(json.dumps(out, indent=1, ensure_ascii=False), encoding="utf8")


def stack_vectors():
    from oracle import w2l_oracle as o
    arrays = {}
    cases = {
        "toy": dict(specs=o.layer_specs(4, 5, main_filter_count=6, out_filter_count=8, striding_kernel=6,
                                        inner_kernel=3, big_kernel=4, inner_count=2), b=2, t=16, f=4, k=5,
                    labels=[[0, 1, 2], [3, 3]], seed=11),
        "real": dict(specs=o.layer_specs(128, 29), b=2, t=64, f=128, k=29,
                     labels=[list(range(12)), [5, 5, 7, 0, 27, 26, 3, 3, 3]], seed=12),
    }
    for name, c in cases.items():
        specs = c["specs"]
        weights = o.glorot_uniform_weights(specs, seed=2, dtype=np.float64)
        rng = np.random.RandomState(c["seed"])
        weights = [(w, rng.uniform(-0.05, 0.05, size=b.shape)) for (w, b) in weights]
        x = np.random.RandomState(0).randn(c["b"], c["t"], c["f"]).astype(np.float32).astype(np.float64)
        lens = [c["t"] // 2, c["t"] // 2 - 3]
        labels = o.pack_label_batch(c["labels"])
        lab_len = [len(l) for l in c["labels"]]
        r = o.loss_and_gradients(specs, weights, x, labels, lens, lab_len)
        arrays[name + "/x"] = x.astype(np.float32)
        arrays[name + "/labels"] = labels
        arrays[name + "/label_lengths"] = np.array(lab_len)
        arrays[name + "/prediction_lengths"] = np.array(lens)
        arrays[name + "/probs"] = r["probs"]
        arrays[name + "/losses"] = r["losses"]
        dec = o.greedy_decode_indices(r["probs"], lens)
        arrays[name + "/decoded"] = o.pack_label_batch([d if d else [-1] for d in dec])
        arrays[name + "/decoded_lengths"] = np.array([len(d) for d in dec])
        for i, ((dw, db), (w, b)) in enumerate(zip(r["grads"], weights)):
            arrays["{}/bias{}".format(name, i)] = b
            arrays["{}/dw_norm{}".format(name, i)] = np.array(np.linalg.norm(dw))
            arrays["{}/db{}".format(name, i)] = db
            arrays["{}/dw_slice{}".format(name, i)] = dw[0, :4, :4].copy()
    np.savez_compressed(str(HERE / "stack_golden.npz"), **arrays)


if __name__ == "__main__":
    codec_vectors()
    stack_vectors()
    print("golden vectors written to", HERE)
