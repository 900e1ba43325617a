"""float64 realisation of the same path on torch (TEST INFRASTRUCTURE ONLY, like w2l_oracle.py / w2l_torch_cpu.py).

The third corner of the parity triangle: two float32 implementations of the step -- the HIP parity
paths and the torch-CPU port -- can only be judged against each other up to their common float32 noise (summation
order, and above all the ReLU decisions of pre-activations within float32 rounding of zero).  This module runs the
SAME step in float64, where that noise is 1e-9 of float32's, so that each float32 implementation can be measured against
it separately: `bench.py`'s `parity` object and tests/test_gpu_round4.py print, per tensor, HIP vs float64, torch-CPU
float32 vs float64, and the number of ReLU decisions each float32 run takes differently from the float64 run.

The convolution is a sum of per-tap matmuls (shares nothing with the HIP kernels, nor with F.conv1d's im2col / oneDNN
path the float32 port uses); it runs wherever `device` says -- on the GPU it is rocBLAS dgemm, used as a CHECKER only
(the product path never calls it).  CTC: F.ctc_loss in float64 on the host.

Semantics as in w2l_torch_cpu.py: /root/reference/speechless/net.py:291-341 (stack, SAME padding), :402-406 (Keras
ctc_batch_cost: log(p + 1e-8), re-normalised), :389 (mean over the batch).  Parity unpinned at this size (see
w2l_oracle.py); this file agrees with w2l_oracle.loss_and_gradients (numpy float64) to 1e-12 on small cases
(tests/test_oracle.py::test_float64_torch_realisation_against_the_numpy_oracle).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .w2l_oracle import same_padding


def _conv_same(x, w, bias, stride):
    """x (B, T, Cin), w (k, Cin, Cout) -> (B, T', Cout): sum over taps of a strided row view times W[tap]"""
    k = w.shape[0]
    t_out, pad_l, pad_r = same_padding(x.shape[1], k, stride)
    xp = F.pad(x, (0, 0, pad_l, pad_r))
    y = None
    for tap in range(k):
        term = xp[:, tap: tap + (t_out - 1) * stride + 1: stride] @ w[tap]
        y = term if y is None else y + term
    return y + bias


def loss_and_gradients(specs, weights, input_batch, labels, prediction_lengths, label_lengths, eps=1e-8,
                       device="cpu", dtype=torch.float64, keep_masks=True):
    """Returns dict(probs (B,T',K) numpy, losses (B,), grads [(dW (k,Cin,Cout), db)] numpy float64,
    masks [bool tensor (B,T',C) on `device` per hidden layer: pre-activation > 0])."""
    ws = [(torch.tensor(np.asarray(w), dtype=dtype, device=device, requires_grad=True),
           torch.tensor(np.asarray(b), dtype=dtype, device=device, requires_grad=True)) for w, b in weights]
    x = torch.tensor(np.asarray(input_batch), dtype=dtype, device=device)
    masks = []
    for spec, (w, b) in zip(specs, ws):
        z = _conv_same(x, w, b, spec.stride)
        if spec.activation == "relu":
            if keep_masks:
                masks.append((z > 0).detach())
            x = F.relu(z)
        elif spec.activation == "elu":
            if keep_masks:
                masks.append((z > 0).detach())
            x = F.elu(z)
        elif spec.activation == "softmax":
            x = F.softmax(z, dim=2)
        elif spec.activation != "linear":
            raise ValueError(spec.activation)
    probs = x
    bsz, _, k = probs.shape
    log_q = F.log_softmax(torch.log(probs.cpu() + eps), dim=2).transpose(0, 1)  # (T, B, K) on the host
    flat = torch.cat([torch.as_tensor(np.asarray(labels[i][:int(label_lengths[i])]), dtype=torch.long)
                      for i in range(bsz)])
    losses = F.ctc_loss(log_q, flat, torch.as_tensor(np.asarray(prediction_lengths), dtype=torch.long),
                        torch.as_tensor(np.asarray(label_lengths), dtype=torch.long), blank=k - 1, reduction="none",
                        zero_infinity=False)
    losses.mean().backward()
    grads = [(w.grad.cpu().numpy().astype(np.float64), b.grad.cpu().numpy().astype(np.float64)) for w, b in ws]
    return dict(probs=probs.detach().cpu().numpy(), losses=losses.detach().numpy(), grads=grads, masks=masks)
