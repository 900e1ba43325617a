#!/usr/bin/env python
"""Yardstick only (not part of the product path): what the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, bf16) reaches
on plain GEMMs with the shapes of the three big convolution passes, next to which the hand-written kernels' numbers in
DESIGN.md can be read.  The convolutions are NOT computed this way: an im2col'ed big_conv_1 operand alone would be 32x the
activation tensor.    python tools/gemm_yardstick.py"""
import torch


def bench(m, n, k, trans_a=False):
    a = torch.randn((k, m) if trans_a else (m, k), device="cuda", dtype=torch.bfloat16)
    b = torch.randn((k, n), device="cuda", dtype=torch.bfloat16)
    f = (lambda: a.t() @ b) if trans_a else (lambda: a @ b)
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        f()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    return ms, 2.0 * m * n * k / ms / 1e9


if __name__ == "__main__":
    for name, (m, n, k, ta) in {
        "big_conv_1 forward  (16000 x 8000 . 8000 x 2000, NN)": (16384, 2048, 8192, False),
        "big_conv_2 forward  (16000 x 2000 . 2000 x 2000, NN)": (16384, 2048, 2048, False),
        "big_conv_1 wgrad    (8000 x 16000 . 16000 x 2000, TN)": (8192, 2048, 16384, True),
        "inner_conv forward  (16000 x 1750 . 1750 x 250, NN)": (16384, 256, 1792, False),
    }.items():
        ms, tf = bench(m, n, k, ta)
        print("{:58s} {:.4f} ms  {:6.0f} TFLOP/s".format(name, ms, tf))
