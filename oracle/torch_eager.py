"""TEST INFRASTRUCTURE ONLY -- the reference's `WaveRNN.generate()` loop restated as PyTorch *eager* code on the CPU.

What it is for: the north star asks for the GPU number "next to the reference PyTorch CPU generate() timed on the same box's
host cores".  The reference tree does not exist on the GPU box, so `bench.py`'s `cpu_baseline` leg times THIS restatement
there: the same ATen kernels in the same per-step sequence as fatchord/WaveRNN models/fatchord_version.py:192-245 (state
init, `I` -> `nn.GRUCell` -> +res -> `nn.GRUCell` -> +res -> fc1 -> fc2 -> fc3, then `sample_from_discretized_mix_logistic`
(utils/distribution.py:87-123) or softmax + `Categorical.sample()` (:231-237)), drawing from torch's global CPU generator
exactly as the reference does (incl. the two throw-away `nn.GRUCell` constructors of `get_gru_cell`, :178-179, :273-279).
It is `kind: "port"` -- a port to the reference's own framework and op sequence, not the reference's file.

Pinned: tests/test_oracle_golden.py runs it against the committed fixtures made by the reference itself (RAW: identical
pre-decode tensor; MoL: <= 1e-6).  Only tests/ and bench.py's cpu_baseline leg import this module; the product never does."""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F


def _cells(sd, rnn_dims, aux_dims):
    """`get_gru_cell` (:273-279): fresh GRUCells (their constructors draw from the global generator) re-pointed at the GRU weights."""
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k], dtype=np.float32))
    cells = []
    for name, inp in (('rnn1', rnn_dims), ('rnn2', rnn_dims + aux_dims)):
        c = torch.nn.GRUCell(inp, rnn_dims)
        c.weight_ih.data, c.weight_hh.data = t(f'{name}.weight_ih_l0'), t(f'{name}.weight_hh_l0')
        c.bias_ih.data, c.bias_hh.data = t(f'{name}.bias_ih_l0'), t(f'{name}.bias_hh_l0')
        cells.append(c)
    return cells


def _mol_sample(logits):
    """utils/distribution.py:87-123 on the (B, 30) logits of one step; draws (1, B, 10) then (1, B) uniforms."""
    y = logits.unsqueeze(0)                                        # (1, B, 30): B x T x C after the reference's two transposes
    k = y.size(2) // 3
    lp = y[:, :, :k]
    u = lp.new_empty(lp.size()).uniform_(1e-5, 1.0 - 1e-5)
    pick = F.one_hot((lp - torch.log(-torch.log(u))).max(dim=-1)[1], k).float()
    mean = torch.sum(y[:, :, k:2 * k] * pick, dim=-1)
    ls = torch.clamp(torch.sum(y[:, :, 2 * k:3 * k] * pick, dim=-1), min=float(np.log(1e-14)))
    v = mean.new_empty(mean.size()).uniform_(1e-5, 1.0 - 1e-5)
    x = mean + torch.exp(ls) * (torch.log(v) - torch.log(1. - v))
    return torch.clamp(torch.clamp(x, min=-1.), max=1.)            # (1, B)


@torch.no_grad()
def loop(sd, mode, mels, aux, seed=None, steps=None):
    """mels (B, T, feat), aux (B, T, 4 * aux_dims) float32 (numpy or torch, already folded).  `seed` -> torch.manual_seed(seed)
    first (the golden fixtures' protocol).  Runs the first `steps` of the T steps.  Returns (out (B, steps) float32 numpy, seconds)."""
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k], dtype=np.float32))
    mels, aux = torch.as_tensor(mels), torch.as_tensor(aux)
    B, T, _ = mels.shape
    steps = T if steps is None else min(int(steps), T)
    rnn_dims = sd['rnn1.weight_hh_l0'].shape[1]
    d = aux.shape[2] // 4
    n_classes = sd['fc3.weight'].shape[0]
    if seed is not None:
        torch.manual_seed(seed)
    rnn1, rnn2 = _cells(sd, rnn_dims, d)
    WI, bI, W1, b1, W2, b2, W3, b3 = (t(k) for k in ('I.weight', 'I.bias', 'fc1.weight', 'fc1.bias', 'fc2.weight', 'fc2.bias', 'fc3.weight', 'fc3.bias'))
    h1 = torch.zeros(B, rnn_dims)
    h2 = torch.zeros(B, rnn_dims)
    x = torch.zeros(B, 1)
    parts = [aux[:, :, d * i:d * (i + 1)] for i in range(4)]
    out = []
    t0 = time.perf_counter()
    for i in range(steps):
        a1, a2, a3, a4 = (p[:, i, :] for p in parts)
        x = F.linear(torch.cat([x, mels[:, i, :], a1], dim=1), WI, bI)
        h1 = rnn1(x, h1)
        x = x + h1
        h2 = rnn2(torch.cat([x, a2], dim=1), h2)
        x = x + h2
        x = F.relu(F.linear(torch.cat([x, a3], dim=1), W1, b1))
        x = F.relu(F.linear(torch.cat([x, a4], dim=1), W2, b2))
        logits = F.linear(x, W3, b3)
        if mode == 'MOL':
            s = _mol_sample(logits)
            out.append(s.view(-1))
            x = s.transpose(0, 1)
        elif mode == 'RAW':
            s = 2 * torch.distributions.Categorical(F.softmax(logits, dim=1)).sample().float() / (n_classes - 1.) - 1.
            out.append(s)
            x = s.unsqueeze(-1)
        else:
            raise RuntimeError("Unknown model mode value - ", mode)
    dt = time.perf_counter() - t0
    return torch.stack(out).transpose(0, 1).numpy(), dt
