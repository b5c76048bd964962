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


def block_prune_state_dict(sd, sparsity=0.95, block=(16, 1), keys=GRU_KEYS):
    """Copy of `sd` (numpy arrays) with the GRU matrices block-pruned; returns (pruned_sd, {key: density})."""
    out = dict(sd)
    density = {}
    for k in keys:
        W = np.asarray(sd[k], dtype=np.float32)
        if W.shape[1] % block[1]:
            raise ValueError(f'{k}: {W.shape} does not tile by {block}')
        M = block_mask(W, sparsity, block)
        out[k] = (W * M).astype(np.float32)
        density[k] = float(M.mean())
    return out, density
