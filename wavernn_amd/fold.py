"""Batched-generation geometry: fold with overlap, cross-fade + unfold, tail fade, mu-law expansion.

Host-side mirror of the reference helpers (fatchord/WaveRNN models/fatchord_version.py:281-405,
utils/dsp.py:8-9,98-103) with identical semantics, written vectorised.  The loop kernels never materialise
the fold (they index the un-folded conditioning with `fold_geometry`); `fold_with_overlap` exists because it
is part of the reference's public API.
"""
import math
import numpy as np
import torch


def fold_geometry(total_len, target, overlap):
    """(num_folds, padded_len) of `fold_with_overlap` for a (1, total_len, F) input (reference :322-330)."""
    stride = target + overlap
    num_folds = (total_len - overlap) // stride
    remaining = total_len - (num_folds * stride + overlap)
    padded = total_len
    if remaining != 0:
        num_folds += 1
        padded = total_len + target + 2 * overlap - remaining
    return num_folds, padded


def pad_tensor(x, pad, side='both'):
    """(b, t, c) zero padding along t; `side` in {'both','before','after'} (reference :281-291)."""
    b, t, c = x.shape
    total = t + 2 * pad if side == 'both' else t + pad
    out = x.new_zeros(b, total, c)
    start = pad if side in ('before', 'both') else 0
    out[:, start:start + t, :] = x
    return out


def fold_with_overlap(x, target, overlap):
    """(1, L, F) -> (num_folds, target + 2*overlap, F); window i starts at i*(target+overlap), the tail is
    zero padded (reference :293-340)."""
    _, total_len, feats = x.shape
    num_folds, padded = fold_geometry(total_len, target, overlap)
    if padded != total_len:
        x = pad_tensor(x, padded - total_len, side='after')
    T = target + 2 * overlap
    idx = (torch.arange(num_folds, device=x.device)[:, None] * (target + overlap)
           + torch.arange(T, device=x.device)[None, :])
    return x[0][idx]


def xfade_and_unfold(y, target, overlap):
    """(num_folds, T) float64 -> 1-D float64 of length num_folds*(T-overlap)+overlap.

    Equal-power cross-fade over the `overlap` head/tail samples of every fold with the first overlap//2 head
    samples silenced, then overlap-add at stride T-overlap.  `target` is ignored and recomputed and `y` is
    modified in place -- both as in the reference (:374-405)."""
    num_folds, length = y.shape
    target = length - 2 * overlap
    stride = target + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.concatenate([np.zeros(silence_len, np.float64), np.sqrt(0.5 * (1 + t))])
    fade_out = np.concatenate([np.ones(silence_len, np.float64), np.sqrt(0.5 * (1 - t))])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    unfolded = np.zeros(num_folds * stride + overlap, dtype=np.float64)
    # folds i and i+2 never overlap (stride >= overlap + target > length/2), so even and odd folds can be
    # scattered as two disjoint vectorised adds without changing the reference's summation order.
    for par in (0, 1):
        rows = np.arange(par, num_folds, 2)
        if rows.size:
            idx = rows[:, None] * stride + np.arange(length)[None, :]
            unfolded[idx.ravel()] += y[rows].ravel()
    return unfolded


def label_2_float(x, bits):
    return 2 * x / (2 ** bits - 1.) - 1.


def decode_mu_law(y, mu, from_labels=True):
    """mu-law expansion (reference utils/dsp.py:98-103), float64."""
    if from_labels:
        y = label_2_float(y, math.log2(mu))
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


def finish_waveform(output, wave_len, hop_length):
    """truncate to wave_len and fade the last 20 hops linearly to zero (reference :255-258)."""
    fade_out = np.linspace(1, 0, 20 * hop_length)
    output = output[:wave_len]
    output[-20 * hop_length:] *= fade_out      # raises ValueError when wave_len < 20*hop, like the reference
    return output
