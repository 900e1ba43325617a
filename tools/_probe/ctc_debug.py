import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from speechless_amd import _lib
lib = _lib.lib()
b, t, k, lmax = 8, 4000, 29, 200
rng = np.random.RandomState(0)
dev = "cuda:0"
lg = rng.randn(b, t, k).astype(np.float32); lg[:, :, k - 1] += float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
logits = torch.tensor(lg, device=dev)
probs = torch.zeros((b, t, k), dtype=torch.float32, device=dev); logq = torch.zeros_like(probs)
lab_len = rng.randint(20, lmax + 1, size=b).astype(np.int32)
labels = np.zeros((b, lmax), dtype=np.int32)
for i, n in enumerate(lab_len): labels[i, :n] = rng.randint(0, k - 1, size=n)
lab = torch.tensor(labels, device=dev); ll = torch.tensor(lab_len, device=dev)
il = torch.full((b,), t, dtype=torch.int32, device=dev)
loss = torch.zeros((b,), dtype=torch.float32, device=dev)
dl = torch.zeros((b, t, 128), dtype=torch.bfloat16, device=dev)
st = torch.cuda.current_stream().cuda_stream
lib.call("sl_softmax_logq", logits.data_ptr(), probs.data_ptr(), logq.data_ptr(), b, t, k, k, t * k, 1e-8, st)
need = lib.raw("sl_ctc_workspace_bytes")(b, t, lmax)
ws = torch.zeros((need,), dtype=torch.uint8, device=dev)
lib.call("sl_ctc_select", 6)
lib.call("sl_ctc_loss_grad", probs.data_ptr(), logq.data_ptr(), lab.data_ptr(), ll.data_ptr(), il.data_ptr(),
         loss.data_ptr(), dl.data_ptr(), b, t, k, lmax, 0, 128, t * 128, _lib.SL_BF16, 1e-8, 1.0 / b, ws.data_ptr(), need, st)
torch.cuda.synchronize()
print("loss", loss.cpu().numpy(), "label lengths", lab_len)
# layout (ctc.hip:ctc_layout)
def up(x): return (x + 255) // 256 * 256
rows = b * t
sp = ((2 * lmax + 1) + 63) // 64 * 64
off = 0
def take(n):
    global off
    at = off; off += up(n); return at
log_alpha = take(rows * sp * 4); log_beta = take(rows * sp * 4); cls = take(b * (lmax + 65) * 4)
lin_alpha = take(rows * 512 * 8); lin_beta = take(rows * 512 * 8); dump = take(2 * b * 512 * 8)
eblocks = b * (t // 8 + 1) * 64
ea = take(eblocks * 4); eb = take(eblocks * 4)
w = ws.cpu().numpy()
al = w[lin_alpha:lin_alpha + rows * 512 * 4].view(np.float32).reshape(b, t, 512)
be = w[lin_beta:lin_beta + rows * 512 * 4].view(np.float32).reshape(b, t, 512)
eaa = w[ea:ea + eblocks * 4].view(np.int32).reshape(b, t // 8 + 1, 64)
eba = w[eb:eb + eblocks * 4].view(np.int32).reshape(b, t // 8 + 1, 64)
for name, rowsarr, ex in (("alpha", al, eaa), ("beta", be, eba)):
    for u in range(b):
        S = 2 * lab_len[u] + 1
        r = rowsarr[u][:, :((S + 7) // 8) * 8]
        bad = ~np.isfinite(r)
        tb = np.where(bad.any(axis=1))[0]
        zero_rows = np.where((r > 0).sum(axis=1) == 0)[0]
        print(name, "utt", u, "S", S, "first non-finite step row", (tb[0] if len(tb) else None) if name == "alpha" else (tb[-1] if len(tb) else None),
              "n non-finite rows", len(tb), "all-zero rows", len(zero_rows), zero_rows[:3])
        if len(tb):
            tt = tb[0] if name == "alpha" else tb[-1]
            for dt in (-2, -1, 0) if name == "alpha" else (2, 1, 0):
                row = r[tt + dt]
                lanes = np.where(~np.isfinite(row).reshape(-1, 8).all(axis=1) | (row.reshape(-1, 8).max(axis=1) > 1e30))[0]
                print("   t", tt + dt, "max per lane (log2)", np.round(np.log2(np.maximum(row.reshape(-1, 8).max(axis=1), 1e-45))[:20], 0),
                      "exp", ex[u, ((tt + dt) if name == "alpha" else (t - 1 - tt - dt)) // 8][:20])
            break
