"""Block-magnitude pruning of the WaveRNN GRU weights -- the sparsity recipe of BASELINE config 5.

The reference's "Pruning - Scratchpad" notebook (`PruneMask.mask_from_matrix`, notebook JSON :40-125) prunes every
gate matrix of a GRU separately by magnitude: sort |W| of the gate, zero everything below the k-th value,
k = int(rows*cols*z).  The WaveRNN paper prunes in BLOCKS (16x1) so the surviving weights can be packed; config 5
 applies the notebook's per-gate rule to block magnitudes: a block's score is the mean |w| of its
16 rows x 1 column, the lowest-scoring fraction z of the blocks of each gate is zeroed.  Applied to `weight_ih_l0` and
`weight_hh_l0` of rnn1 and rnn2 (the notebook's `prune_rnn_input=True`); biases and the dense layers are untouched.

The loop kernels run the pruned model as masked dense weights (parity: tests/test_gpu_parity.py); a packed block-sparse
kernel is the follow-up that turns the 95 % zeros into speed (the pruned GRU weights are 0.64 MB and fit a few CUs).
"""
import numpy as np

GRU_KEYS = ('rnn1.weight_ih_l0', 'rnn1.weight_hh_l0', 'rnn2.weight_ih_l0', 'rnn2.weight_hh_l0')
LINEAR_KEYS = ('fc1.weight', 'fc2.weight')       # the notebook prunes `[self.rnn, self.fc]` ("Pruning - Scratchpad.ipynb" :199-204; 'Linear': 1 mask, :54)


def block_mask(W, sparsity, block=(16, 1), gates=3):
    """0/1 mask of W (gates*H, K): per gate, the `sparsity` fraction of (block[0] x block[1]) blocks with the smallest
    mean magnitude is zeroed (ties at the threshold survive, as `W_abs >= threshold` does in the notebook)."""
    W = np.asarray(W, dtype=np.float32)
    rows, cols = W.shape
    h = rows // gates
    br, bc = block
    if h % br or cols % bc:
        raise ValueError(f'gate shape ({h},{cols}) is not a multiple of the block {block}')
    mask = np.ones_like(W)
    for g in range(gates):
        Wg = np.abs(W[g * h:(g + 1) * h])
        score = Wg.reshape(h // br, br, cols // bc, bc).mean(axis=(1, 3))
        k = int(score.size * sparsity)
        if k <= 0:
            continue
        thr = np.sort(score.reshape(-1))[min(k, score.size - 1)]
        keep = (score >= thr).astype(np.float32)
        mask[g * h:(g + 1) * h] = np.repeat(np.repeat(keep, br, axis=0), bc, axis=1)
    return mask


def block_prune_state_dict(sd, sparsity=0.95, block=(16, 1), keys=GRU_KEYS, linear=False):
    """Copy of `sd` (numpy arrays) with the GRU matrices block-pruned -- and, `linear=True`, fc1 / fc2 as the reference's pruning notebook prunes its
    Linear layer (one mask over the whole matrix) --; returns (pruned_sd, {key: density})."""
    out = dict(sd)
    density = {}
    for k in tuple(keys) + (LINEAR_KEYS if linear else ()):
        W = np.asarray(sd[k], dtype=np.float32)
        if W.shape[1] % block[1]:
            raise ValueError(f'{k}: {W.shape} does not tile by {block}')
        M = block_mask(W, sparsity, block, gates=1 if k in LINEAR_KEYS else 3)
        out[k] = (W * M).astype(np.float32)
        density[k] = float(M.mean())
    return out, density


# ---------------------------------------------------------------------------------------------------------------------
# The pruning WORKFLOW of the reference ("Pruning - Scratchpad" notebook, JSON :40-186): masks that follow a cubic sparsity
# schedule during training.  Same class names, constructor arguments and method names as the notebook's `PruneMask` /
# `Pruner`, so its training loop (`model.pruner.prune(model.layers2prune, model.t)`) runs unchanged; what is added is the
# BLOCK form (`block=(16, 1)`: the structure `wrnn_sparse_kernel` packs) and a `step_hook` for an ordinary training loop.
# torch only -- works on whatever device the layers live on.
# ---------------------------------------------------------------------------------------------------------------------
def _layer_kind(layer):
    return str(layer).split('(')[0]


_SPLITS = {'Linear': 1, 'GRU': 3, 'LSTM': 4}


class PruneMask:
    """Masks for the weight matrices of one layer (notebook :40-125).  `Linear`: its weight; `GRU` / `LSTM`: weight_ih
    (only if `prune_rnn_input`) and weight_hh, each gate matrix pruned separately."""

    def __init__(self, layer, prune_rnn_input, block=None):
        import torch
        kind = _layer_kind(layer)
        if kind not in _SPLITS:
            raise ValueError(f'cannot prune a {kind} layer (Linear, GRU, LSTM)')
        self.block = block
        self.p_idx = [0] if kind == 'Linear' else ([0, 1] if prune_rnn_input else [1])
        params = self.get_params(layer)
        self.mask = [torch.ones_like(W) for W in params]
        self.total_params = sum(W.size(0) * W.size(1) for W in params)
        self.pruned_params = 0
        self.split_size = self.mask[0].size(0) // _SPLITS[kind]

    def get_params(self, layer):
        ps = list(layer.parameters())
        return [ps[i].data for i in self.p_idx]

    def mask_from_matrix(self, W, z):
        """Per gate: zero the k = int(n * z) entries (or blocks, by mean magnitude) below the k-th smallest magnitude; ties at
        the threshold survive (`>=`), exactly the notebook's rule (:92-113)."""
        import torch
        out = []
        for Wg in torch.split(W, self.split_size):
            score = Wg.abs()
            if self.block is not None:
                br, bc = self.block
                if Wg.size(0) % br or Wg.size(1) % bc:
                    raise ValueError(f'gate shape {tuple(Wg.shape)} is not a multiple of the block {self.block}')
                score = score.reshape(Wg.size(0) // br, br, Wg.size(1) // bc, bc).mean(dim=(1, 3))
            k = int(score.numel() * z)
            thr = torch.sort(score.reshape(-1))[0][min(k, score.numel() - 1)]
            keep = (score >= thr).to(W.dtype)
            if self.block is not None:
                keep = keep.repeat_interleave(self.block[0], dim=0).repeat_interleave(self.block[1], dim=1)
            out.append(keep)
        return torch.cat(out)

    def update_mask(self, layer, z):
        self.mask = [self.mask_from_matrix(W, z) for W in self.get_params(layer)]
        self.update_prune_count()

    def apply_mask(self, layer):
        for M, W in zip(self.mask, self.get_params(layer)):
            W *= M

    def update_prune_count(self):
        self.pruned_params = int(sum(float((1 - M).sum()) for M in self.mask))


class Pruner:
    """Cubic sparsity schedule (notebook :127-186): z(t) = Z * (1 - (1 - (t - t_0) / S)^3), clamped to [0, Z]; masks are
    recomputed every `prune_every` steps after `start_prune` and applied on every step from `start_prune` on."""

    def __init__(self, layers, start_prune, prune_steps, target_sparsity, prune_rnn_input=True, prune_every=500, block=None):
        self.z = 0
        self.t_0 = start_prune
        self.S = prune_steps
        self.Z = target_sparsity
        self.prune_every = prune_every
        self.masks = [PruneMask(layer, prune_rnn_input, block) for layer in layers]
        self.num_pruned = 0
        self.total_params = sum(m.total_params for m in self.masks)
        #: called after every step that edited weights (the masks are applied through `.data`, which bumps nothing PyTorch
        #: exposes): `wavernn_pruner` points it at `WaveRNN.invalidate_engines`, so the next generate() repacks the weights
        self.on_change = None

    @staticmethod
    def _step(t):
        return int(t.detach().reshape(-1)[0].item()) if hasattr(t, 'detach') else int(t)

    def update_sparsity(self, t):
        t = self._step(t)
        z = self.Z * (1 - (1 - (t - self.t_0) / self.S) ** 3)
        self.z = max(0, min(self.Z, z))
        return t

    def prune_or_not(self, t):
        return t % self.prune_every == 0 and t > self.t_0

    def apply_or_not(self, t):
        return t >= self.t_0

    def prune(self, layers, t):
        t = self.update_sparsity(t)
        for layer, m in zip(layers, self.masks):
            if self.prune_or_not(t):
                m.update_mask(layer, self.z)
            if self.apply_or_not(t):
                m.apply_mask(layer)
        self.count_num_pruned()
        if self.apply_or_not(t) and self.on_change is not None:
            self.on_change()

    def restart(self, layers, t):
        """After a training restart: rebuild the masks from the (already pruned) weights at the schedule's sparsity."""
        self.update_sparsity(t)
        for layer, m in zip(layers, self.masks):
            m.update_mask(layer, self.z)
        self.count_num_pruned()

    def count_num_pruned(self):
        self.num_pruned = sum(m.pruned_params for m in self.masks)

    def step_hook(self, layers, step_fn):
        """Training-side hook: returns a callable to invoke after every `optimizer.step()`; `step_fn()` gives the global step
        (e.g. `model.get_step`, reference train_wavernn.py:109)."""
        return lambda: self.prune(layers, step_fn())


def wavernn_pruner(model, start_prune, prune_steps, target_sparsity=0.95, prune_every=500, prune_fc=False, block=(16, 1)):
    """`Pruner` over a WaveRNN's recurrent layers (both weight matrices of rnn1 and rnn2; with `prune_fc` also fc1 / fc2, the
    notebook's `splits['Linear']` case) in the 16x1 block structure the block-sparse loop kernel packs.  Returns
    (pruner, layers): call `pruner.prune(layers, step)` after every optimiser step, then `model.generate()` -- the device weight
    pack is rebuilt after every pruning step (`Pruner.on_change` -> `WaveRNN.invalidate_engines`); `auto` runs `wrnn_sparse_kernel` once
    every block row is sparse enough (`LoopEngine.sparse_blocks`), and its gathered fc stages when `prune_fc` has made fc1 / fc2 block-sparse
    too (`LoopEngine.sparse_fc_blocks`; round 6)."""
    layers = [model.rnn1, model.rnn2] + ([model.fc1, model.fc2] if prune_fc else [])
    if prune_fc and block is not None and (model.fc1.weight.size(1) % block[1] or model.fc1.weight.size(0) % block[0]):
        raise ValueError('fc weights do not tile by the block')
    pruner = Pruner(layers, start_prune, prune_steps, target_sparsity, True, prune_every, block)
    if hasattr(model, 'invalidate_engines'):
        pruner.on_change = model.invalidate_engines
    return pruner, layers
