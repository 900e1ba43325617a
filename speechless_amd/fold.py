"""Inference-time BatchNorm folding for the fused conv + bias + ReLU epilogue.

BASELINE.json's north_star names "a fused BatchNorm+ReLU"; the reference network has NO normalisation layer
(speechless/net.py:291-341 builds Conv1D layers only).  What the hot path offers instead
is the fused epilogue  y = relu(conv(x, W) + b)  of sl_conv1d_nt, and an inference-mode BatchNorm behind a convolution
is exactly representable in it:

    relu(gamma * (conv(x, W) + b - mean) / sqrt(var + eps) + beta) = relu(conv(x, W * s) + (b - mean) * s + beta),
    s = gamma / sqrt(var + eps)   per output channel.

So a checkpoint with BatchNorm layers (none of the reference's) runs on the same kernels at no extra HBM pass.
Training-mode BatchNorm (batch statistics) would couple the utterances of a batch and is outside the path.
"""
import numpy as np


def fold_batchnorm_into_conv(kernel, bias, gamma, beta, moving_mean, moving_variance, epsilon=1e-3):
    """kernel (k, cin, cout), bias (cout,) -> (kernel', bias') such that conv+bias' == BatchNorm(conv+bias)."""
    kernel = np.asarray(kernel)
    scale = (np.asarray(gamma, dtype=np.float64) / np.sqrt(np.asarray(moving_variance, dtype=np.float64) + epsilon))
    if scale.shape != (kernel.shape[2],):
        raise ValueError("BatchNorm parameters must have one entry per output channel ({})".format(kernel.shape[2]))
    new_kernel = (kernel.astype(np.float64) * scale[None, None, :]).astype(kernel.dtype)
    new_bias = ((np.asarray(bias, dtype=np.float64) - np.asarray(moving_mean, dtype=np.float64)) * scale +
                np.asarray(beta, dtype=np.float64)).astype(np.asarray(bias).dtype)
    return new_kernel, new_bias
