import ast, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
CASES = ['raw_unbatched_24f', 'raw_batched_60f', 'mol_unbatched_24f', 'mol_batched_100f', 'mol_batched_ragged_53f']

#: BASELINE configs 2 and 3 (vocoder side) at their stated full-size inputs (T = 12,100); reference outputs, see scripts/make_golden.py
BIG_CASES = ['mol_batched_481f', 'raw_batched_481f', 'mol_tacotron_800f']

#: MoL tolerance (max abs error on samples in [-1,1]) -- BASELINE.md "budget 1e-5"; observed <= 4e-7 CPU-vs-CPU.
MOL_TOL = 1e-5


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = ast.literal_eval(str(g['config']))
    return cfg, g


def case_mel(cfg, g):
    """(feat, N) float32 mel of a golden case: seeded random mel, or the stored one (config 3: the reference Tacotron's)."""
    from wavernn_amd.synthetic import random_mel
    return np.ascontiguousarray(g['mel'], np.float32) if cfg['mseed'] is None else random_mel(cfg['mseed'], cfg['frames'])


def oracle_loop_fn(sd, mode):
    """CPU stand-in for the HIP loop with the `generate_corpus(loop_fn=...)` signature -- TEST ONLY: lets the
    sharding / all-gather / unfold host logic run under gloo on a box without a GPU."""
    import torch
    from oracle import c_oracle as C

    def fn(mels_up, aux, seg_pos, seg_lim, T, noise, hop):
        mu, au, nz = mels_up.cpu().numpy(), aux.cpu().numpy(), noise.cpu().numpy()
        n = len(seg_pos)
        mels_f = np.zeros((n, T, mu.shape[1]), np.float32)
        aux_f = np.zeros((n, T, au.shape[1]), np.float32)
        for b in range(n):
            p = int(seg_pos[b]) + np.arange(T)
            ok = p < int(seg_lim[b])
            mels_f[b, ok] = mu[p[ok]]
            aux_f[b, ok] = au[p[ok] // hop]
        if mode == 'MOL':
            nzo = (np.ascontiguousarray(nz[:, :10 * n].reshape(T, n, 10)), np.ascontiguousarray(nz[:, 10 * n:]))
        else:
            nzo = np.ascontiguousarray(nz.reshape(T, n, -1))
        return torch.from_numpy(C.loop(sd, mode, mels_f, aux_f, nzo))
    return fn


def zero_loop_fn(sd, mode):
    """Loop stand-in that generates silence -- TEST ONLY: for launch-path tests whose workload is too large for the oracle."""
    import torch

    def fn(mels_up, aux, seg_pos, seg_lim, T, noise, hop):
        return torch.zeros(len(seg_pos), T)
    return fn


def probe_loop_fn(sd, mode):
    """Loop stand-in whose output identifies the segment -- TEST ONLY: row b is filled with the first up-sampled mel value of segment
    b's conditioning (distinct per segment for random mels), so a test can tell whether sharded, gathered rows arrive in table order."""
    import torch

    def fn(mels_up, aux, seg_pos, seg_lim, T, noise, hop):
        first = mels_up[torch.as_tensor(np.asarray(seg_pos, dtype=np.int64)), 0].cpu()
        return first[:, None].expand(len(seg_pos), T).contiguous()
    return fn
