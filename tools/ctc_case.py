#!/usr/bin/env python
"""ONE case of tools/fuzz_ctc.py (`--seed S`, case index I) across the lattice variants of sl_ctc_select, per utterance: loss,
worst gradient error against the float64 oracle and where it sits; then the raw lattice of utterance U (block exponents of the
first / last blocks, log2 of the first states' alpha / beta at a few frames) from the call's workspace.  With a
-DSL_PROBE_CTC_TOTALS build (SL_LIB_PATH=tools/_probe/lib_TOT.so) the gradient of classes 0..3 carries, per frame: the posterior
total - 1, and the number of states with alpha > 0, beta > 0, both.  How round 6 found the lane that missed its lift
(speechless_amd/csrc/ctc.hip, rescale; profiles/r06_ctc_fix.txt).

    python tools/ctc_case.py SEED INDEX [UTTERANCE]"""
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from fuzz_ctc import replay_case  # noqa: E402
from oracle import w2l_oracle as o  # noqa: E402
from speechless_amd import _lib  # noqa: E402
from test_gpu_parity import run_ctc_kernel  # noqa: E402

seed, target = int(sys.argv[1]), int(sys.argv[2])
lib = _lib.lib()
k, t, input_len, lab_len, labels_list, logits, desc = replay_case(seed, target)
b = logits.shape[0]
print("case", target, "k", k, "t", t, "b", b, "input_len", input_len, "lab_len", lab_len, desc)
labels = o.pack_label_batch([l if l else [-1] for l in labels_list])
ref_p = o.softmax(logits.astype(np.float64))
ref_loss, ref_dp = o.ctc_batch_cost(ref_p, labels, input_len, lab_len)
ref_dl = o.softmax_backward(ref_p, ref_dp)
print("ref loss", ref_loss)
for v in (0, 10, 11, 4, 2, 3, 1, 8):
    lib.call("sl_ctc_select", v)
    _, loss, dl = run_ctc_kernel(lib, logits, labels, lab_len, input_len)
    errs = [float(np.abs(dl[i] - ref_dl[i]).max()) if np.isfinite(ref_loss[i]) else -1 for i in range(b)]
    where = [tuple(int(x) for x in np.unravel_index(np.abs(dl[i] - ref_dl[i]).argmax(), dl[i].shape)) for i in range(b)]
    print("variant %2d loss %s  grad err per utt %s at %s" % (v, loss, ["%.1e" % e for e in errs], where))
    if v == 0:
        errs0 = errs
lib.call("sl_ctc_select", 0)
if os.environ.get("SL_LIB_PATH"):
    lib.call("sl_ctc_select", 11)
    _, loss, dl = run_ctc_kernel(lib, logits, labels, lab_len, input_len)
    for i in range(b):
        dev = dl[i, :input_len[i], 0]
        j = int(np.abs(dev).argmax())
        print("  frames 0..7: total-1, #alpha>0, #beta>0, #both:", [tuple(float(x) for x in dl[i, f, :4]) for f in range(8)])
        print("utt", i, "max |total-1|", float(np.abs(dev).max()), "at frame", j, "of", input_len[i], "; frames over 1e-3:", np.nonzero(np.abs(dev) > 1e-3)[0][:20])
# ---- raw lattice dump (variant 11: no repair)
import torch
def up(n): return (n + 255) // 256 * 256
bb, tt, kk = logits.shape
l_max = labels.shape[1]
sp = ((2 * l_max + 1) + 63) // 64 * 64
rows = bb * tt
off = {}
o_ = 0
for name, size in (("log_alpha", rows * sp * 8), ("log_beta", rows * sp * 8), ("cls", bb * (l_max + 65) * 4),
                   ("lin_alpha", rows * 512 * 4), ("lin_beta", rows * 512 * 4), ("dump", 2 * bb * 512 * 4),
                   ("ea", bb * (tt // 8 + 2) * 64 * 4), ("eb", bb * (tt // 8 + 2) * 64 * 4)):
    off[name] = o_; o_ += up(size)
dev = "cuda:0"
lg = torch.tensor(logits, dtype=torch.float32, device=dev)
probs = torch.zeros((bb, tt, kk), dtype=torch.float32, device=dev); logq = torch.zeros_like(probs)
lab = torch.tensor(labels, dtype=torch.int32, device=dev)
ll = torch.tensor(lab_len, dtype=torch.int32, device=dev); il = torch.tensor(input_len, dtype=torch.int32, device=dev)
loss_t = torch.zeros((bb,), dtype=torch.float32, device=dev); dl_t = torch.zeros((bb, tt, kk), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.call("sl_softmax_logq", lg.data_ptr(), probs.data_ptr(), logq.data_ptr(), bb, tt, kk, kk, tt * kk, 1e-8, st)
need = lib.raw("sl_ctc_workspace_bytes")(bb, tt, l_max)
ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
lib.call("sl_ctc_select", 11)
lib.call("sl_ctc_loss_grad", probs.data_ptr(), logq.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(), loss_t.data_ptr(),
         dl_t.data_ptr(), bb, tt, kk, l_max, 0, kk, tt * kk, _lib.SL_F32, 1e-8, 1.0, ws.data_ptr(), need, st)
torch.cuda.synchronize()
w = ws.cpu().numpy()
def view(name, dtype, count): return w[off[name]: off[name] + count * 4].view(dtype)
u = int(sys.argv[3]) if len(sys.argv) > 3 else int(np.argmax(errs0))
T = input_len[u]
nblk = tt // 16 + 1
ea = view("ea", np.int32, bb * (tt // 8 + 2) * 64)[u * nblk * 64:(u + 1) * nblk * 64].reshape(nblk, 64)
eb = view("eb", np.int32, bb * (tt // 8 + 2) * 64)[u * nblk * 64:(u + 1) * nblk * 64].reshape(nblk, 64)
la = view("lin_alpha", np.uint32, rows * 512).reshape(bb, tt, 512)[u]
lb = view("lin_beta", np.uint32, rows * 512).reshape(bb, tt, 512)[u]
def hi2d(x): return (x.astype(np.uint64) << np.uint64(32)).view(np.float64)
print("T", T, "blocks", (T - 1) // 16)
frames = [int(x) for x in os.environ.get("FRAMES", "0,1,3,4,5,10,11,12").split(",")]
s0, s1 = [int(x) for x in os.environ.get("STATES", "0,16").split(",")]
lanes = slice(s0 // 8, (s1 + 7) // 8)
np.set_printoptions(linewidth=200)
for f in frames:
    ba, bb_ = f // 16, (T - 1 - f) // 16
    print("frame", f, "alpha block", ba, "exponents", ea[ba, lanes], " beta step", T - 1 - f, "block", bb_, "exponents", eb[bb_, lanes])
    print("   alpha log2:", np.round(np.log2(np.maximum(hi2d(la[f, s0:s1]), 1e-320)), 1))
    print("   beta  log2:", np.round(np.log2(np.maximum(hi2d(lb[f, s0:s1]), 1e-320)), 1))
print("loss", loss_t.cpu().numpy()[u], "log2 Z", -ref_loss[u] / np.log(2))
