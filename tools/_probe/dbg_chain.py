import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
import torch
from test_gpu_round2 import make_case, make_engine, run_loss_and_grads, rel_l2
from speechless_amd.engine import HALO
t = 300
case = make_case(b=3, t=t, seed=70 + t)
res = {}
for chain in (True, False):
    eng = make_engine(case, "bf16")
    eng.use_chain = chain
    losses, grads = run_loss_and_grads(eng, case)
    res[chain] = eng
ea, eb = res[True], res[False]
t_out = ea.cur.t_out
print("t_out", t_out)
for i in range(9):
    ga, gb = ea.cur.g[i].float().cpu().numpy(), eb.cur.g[i].float().cpu().numpy()
    d = np.abs(ga - gb)
    rows = np.where(d.max(axis=(0, 2)) > 1e-3 * np.abs(gb).max())[0]
    chans = np.where(d.max(axis=(0, 1)) > 1e-3 * np.abs(gb).max())[0]
    print(i, "rel", rel_l2(ga, gb), "bad rows", rows[:10] - HALO, len(rows), "bad chans", chans[:10], len(chans),
          "nonzero frac a/b", (ga != 0).mean(), (gb != 0).mean())
