"""torch-CPU fp32 realisation of the same path (TEST INFRASTRUCTURE ONLY, like w2l_oracle.py).

Two jobs:
  1. independent cross-check of the numpy restatement (different conv kernel: F.conv1d on explicitly
     padded input; different CTC implementation: F.ctc_loss; gradients by autograd instead of by hand);
  2. the multi-threaded CPU baseline that bench.py times on the GPU node's host cores
     (``cpu_baseline.kind == "port"``) -- the reference's own Keras/TF CPU path cannot be imported here
     (SURVEY.md section 8c), so the baseline is this port of it.

Semantics follow /root/reference/speechless/net.py:291-341 (stack), :402-406 (Keras ctc_batch_cost: the
op sees log(p + 1e-8) and re-normalises), :389 (mean over batch).  Parity unpinned (see w2l_oracle.py).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .w2l_oracle import same_padding


def to_torch_weights(weights, requires_grad=True):
    """[(W (k,Cin,Cout), b)] -> [(W (Cout,Cin,k) leaf tensor, b leaf tensor)]"""
    out = []
    for w, b in weights:
        wt = torch.tensor(np.ascontiguousarray(np.transpose(w, (2, 1, 0))), dtype=torch.float32,
                          requires_grad=requires_grad)
        bt = torch.tensor(b, dtype=torch.float32, requires_grad=requires_grad)
        out.append((wt, bt))
    return out


def forward_probs(specs, tweights, input_batch, masks=None):
    """input_batch: torch (B,T,F) float32.  Returns probabilities (B,T',K).  masks: optional list that receives, per
    hidden layer, the bool tensor (B,T',C) of pre-activations > 0 (the ReLU decisions; bench.py `parity` counts how many
    of them each float32 implementation takes differently from the float64 run, oracle/w2l_float64.py)."""
    x = input_batch.transpose(1, 2)  # (B,C,T)
    for spec, (w, b) in zip(specs, tweights):
        t_in = x.shape[2]
        _, pad_l, pad_r = same_padding(t_in, spec.kernel_size, spec.stride)
        x = F.conv1d(F.pad(x, (pad_l, pad_r)), w, b, stride=spec.stride)
        if masks is not None and spec.activation in ("relu", "elu"):
            masks.append((x.detach() > 0).transpose(1, 2).contiguous())
        if spec.activation == "relu":
            x = F.relu(x)
        elif spec.activation == "softmax":
            x = F.softmax(x, dim=1)
        elif spec.activation == "elu":
            x = F.elu(x)
        elif spec.activation != "linear":
            raise ValueError(spec.activation)
    return x.transpose(1, 2)


def per_utterance_ctc(probs, labels, prediction_lengths, label_lengths, eps=1e-8):
    bsz, _, k = probs.shape
    log_q = F.log_softmax(torch.log(probs + eps), dim=2).transpose(0, 1)  # (T,B,K)
    flat = torch.cat([torch.as_tensor(np.asarray(labels[i][:int(label_lengths[i])]), dtype=torch.long)
                      for i in range(bsz)]) if bsz else torch.zeros(0, dtype=torch.long)
    return F.ctc_loss(log_q, flat, torch.as_tensor(np.asarray(prediction_lengths), dtype=torch.long),
                      torch.as_tensor(np.asarray(label_lengths), dtype=torch.long), blank=k - 1,
                      reduction="none", zero_infinity=False)


def loss_and_gradients(specs, weights, input_batch, labels, prediction_lengths, label_lengths, eps=1e-8):
    tweights = to_torch_weights(weights)
    x = torch.tensor(np.asarray(input_batch), dtype=torch.float32)
    probs = forward_probs(specs, tweights, x)
    losses = per_utterance_ctc(probs, labels, prediction_lengths, label_lengths, eps)
    mean_loss = losses.mean()
    mean_loss.backward()
    grads = [(np.ascontiguousarray(np.transpose(w.grad.numpy(), (2, 1, 0))), b.grad.numpy().copy())
             for (w, b) in tweights]
    return dict(probs=probs.detach().numpy(), losses=losses.detach().numpy(), mean_loss=float(mean_loss),
                grads=grads)


def keras_adam_update(params, state, step, lr=1e-4, beta_1=0.9, beta_2=0.999, epsilon=1e-8):
    """Keras-2.0 Adam (keras/optimizers.py of the 2.0.x line the reference pins nothing tighter than; net.py:132):
    lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t);  m, v as usual;  p -= lr_t * m / (sqrt(v) + epsilon)
    -- epsilon OUTSIDE the bias correction, unlike torch.optim.Adam.  In place on the leaf tensors `params`
    (their .grad read), state = (list of m, list of v)."""
    ms, vs = state
    lr_t = lr * np.sqrt(1.0 - beta_2 ** step) / (1.0 - beta_1 ** step)
    with torch.no_grad():
        grads = [p.grad for p in params]
        torch._foreach_mul_(ms, beta_1)
        torch._foreach_add_(ms, grads, alpha=1.0 - beta_1)
        torch._foreach_mul_(vs, beta_2)
        torch._foreach_addcmul_(vs, grads, grads, value=1.0 - beta_2)
        denom = torch._foreach_sqrt(vs)
        torch._foreach_add_(denom, epsilon)
        torch._foreach_addcdiv_(params, ms, denom, value=-lr_t)


def timed_training_steps(specs, weights, input_batch, labels, prediction_lengths, label_lengths, steps=1,
                         warmup=0, eps=1e-8, lr=1e-4, record_first=False):
    """fwd + CTC + bwd + Keras-form Adam on the host cores; returns seconds per step (list).  Used by bench.py's
    cpu_baseline leg.  record_first=True: also returns what the FIRST step (the one from `weights`) computed --
    per-utterance losses, probabilities, gradients (Keras layout) and the weights after its update -- so that the
    same leg doubles as the parity checker at the benchmark's own batch size (bench.py `parity`)."""
    import time
    tweights = to_torch_weights(weights)
    params = [p for wb in tweights for p in wb]
    state = ([torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params])
    x = torch.tensor(np.asarray(input_batch), dtype=torch.float32)
    times = []
    record = None
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        for p in params:
            p.grad = None
        masks = [] if (record_first and it == 0) else None
        probs = forward_probs(specs, tweights, x, masks)
        losses = per_utterance_ctc(probs, labels, prediction_lengths, label_lengths, eps)
        losses.mean().backward()
        if record_first and it == 0:
            record = dict(masks=masks, losses=losses.detach().numpy().copy(), probs=probs.detach().numpy().copy(),
                          grads=[(np.ascontiguousarray(np.transpose(w.grad.numpy(), (2, 1, 0))), b.grad.numpy().copy())
                                 for (w, b) in tweights])
        keras_adam_update(params, state, it + 1, lr=lr)
        if record_first and it == 0:
            record["weights_after"] = [(np.ascontiguousarray(np.transpose(w.detach().numpy(), (2, 1, 0))),
                                        b.detach().numpy().copy()) for (w, b) in tweights]
        dt = time.perf_counter() - t0
        if it >= warmup:
            times.append(dt)
    return (times, record) if record_first else times
