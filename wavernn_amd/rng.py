"""Sampling noise for the loop, drawn in the order the reference consumes its generator.

`source='cpu'` (parity mode): draws from torch's CPU generator exactly as `WaveRNN.generate()` does when the
reference runs on the CPU -- including the two throw-away `nn.GRUCell` constructors of `get_gru_cell`
(models/fatchord_version.py:178-179,273-279), whose `reset_parameters` burn 3,201,024 uniform draws before the
loop starts.  After `torch.manual_seed(s)` the result equals the reference's CPU sample path and the
generator is left in the same state.  CPU `uniform_` / `exponential_` fill serially in memory order, so one big
call equals the reference's per-step calls.  With `generator=g` the same stream is drawn
from a private `torch.Generator` (corpus batches: one independent stream per utterance) and the constructor
draws are burnt as plain uniforms (one 32-bit draw per parameter element, Appendix B.2/B.4).

`source='device'`: the device generator (Philox), like the reference does when it runs on a GPU; not comparable
with any CPU run (nor is the reference's own GPU run).
"""
import torch


def gru_cell_ctor_draws(rnn_dims, aux_dims):
    """32-bit draws `nn.GRUCell(rnn, rnn)` + `nn.GRUCell(rnn+aux, rnn)` consume in `reset_parameters`."""
    h = rnn_dims
    one = lambda inp: 3 * h * inp + 3 * h * h + 2 * 3 * h
    return one(h) + one(h + aux_dims)


def burn_ctor_draws(rnn_dims, aux_dims, source='cpu', generator=None):
    """The RNG side effect of `get_gru_cell` x 2 (reference :178-179, :273-279) -- once per generate() call."""
    if source != 'cpu':
        return
    if generator is None:
        torch.nn.GRUCell(rnn_dims, rnn_dims)                 # same RNG side effect as get_gru_cell(self.rnn1)
        torch.nn.GRUCell(rnn_dims + aux_dims, rnn_dims)      # ... and get_gru_cell(self.rnn2)
    else:
        torch.empty(gru_cell_ctor_draws(rnn_dims, aux_dims), dtype=torch.float32).uniform_(0, 1, generator=generator)


def draw_steps(mode, B, steps, n_classes, device, source='cpu', generator=None):
    """Noise of `steps` consecutive loop steps, continuing the stream: MOL -> (steps, 11*B) U(1e-5, 1-1e-5); RAW -> (steps, B,
    n_classes) Exp(1).  CPU fills are serial in memory order, so chunked draws equal one big draw (= the reference's
    per-step draws); this is what lets long RAW runs upload their noise in slices instead of T*B*C floats at once."""
    if source == 'cpu':
        if mode == 'MOL':
            n = torch.empty(steps, 11 * B, dtype=torch.float32).uniform_(1e-5, 1.0 - 1e-5, generator=generator)
        else:
            n = torch.empty(steps, B, n_classes, dtype=torch.float32).exponential_(1, generator=generator)
        return n.to(device, non_blocking=False)
    if source == 'device':
        if mode == 'MOL':
            return torch.empty(steps, 11 * B, dtype=torch.float32, device=device).uniform_(1e-5, 1.0 - 1e-5)
        return torch.empty(steps, B, n_classes, dtype=torch.float32, device=device).exponential_(1)
    raise ValueError(f'unknown noise source {source!r}')


def draw_noise(mode, B, T, n_classes, rnn_dims, aux_dims, device, source='cpu', generator=None):
    """MOL -> (T, 11*B) U(1e-5, 1-1e-5); RAW -> (T, B, n_classes) Exp(1).  float32 on `device`."""
    if source not in ('cpu', 'device'):
        raise ValueError(f'unknown noise source {source!r}')
    burn_ctor_draws(rnn_dims, aux_dims, source, generator)
    return draw_steps(mode, B, T, n_classes, device, source, generator)
