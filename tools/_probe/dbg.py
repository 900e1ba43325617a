import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from test_gpu_parity import make_case, make_engine
case = make_case(b=3, t=64, seed=5)
eng = make_engine(case, "bf16x3")
eng.load_input(case["x"]); eng.set_labels(case["labels"], case["label_lengths"], case["prediction_lengths"])
buf = eng.cur
for i, g in enumerate(buf.wgrad_geom):
    print(i, {n: getattr(g, n) for n, _ in g._fields_})
