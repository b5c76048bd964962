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


def mel_rows_tap_sums(taps, scale):
    """The three tap sums per phase that wrnn_generate_segments builds for `wrnn_options.mel_stage = 1` (csrc/wrnn_abi.hip): with
    q = scale * a + ph the taps j < scale - ph of out(q) = sum_j w[j] rep(q + j - scale) fall on input row a - 1, the next `scale` on row
    a, the last ph + 1 on row a + 1.  Restated here (float64 sums, rounded to float32) so the CPU tests can check the derivation."""
    taps = np.asarray(taps, np.float64).reshape(-1)
    assert taps.shape[0] == 2 * scale + 1
    co = np.zeros((scale, 3), np.float32)
    for ph in range(scale):
        co[ph] = [taps[:scale - ph].sum(), taps[scale - ph:2 * scale - ph].sum(), taps[2 * scale - ph:].sum()]
    return co


def mel_rows_formula(rows, taps, j, scale=11):
    """What wrnn_duo_kernel's `cond_tile_rows` forms for un-cropped positions j: fma(c2, r[a+1], fma(c1, r[a], c0 * r[a-1])) in float32
    (each step rounded; the products are exact in float64).  rows [n_rows, feat] float32, j int array -> [len(j), feat] float32."""
    co = mel_rows_tap_sums(taps, scale).astype(np.float64)
    j = np.asarray(j, np.int64)
    a, ph = j // scale, j % scale
    assert a.min() >= 1 and a.max() + 1 < rows.shape[0]
    r = rows.astype(np.float64)
    acc = (co[ph, 0][:, None] * r[a - 1]).astype(np.float32).astype(np.float64)
    acc = (co[ph, 1][:, None] * r[a] + acc).astype(np.float32).astype(np.float64)
    return (co[ph, 2][:, None] * r[a + 1] + acc).astype(np.float32)


def oracle_stage2_rows(sd, mel, pad=2, scales=(5, 5)):
    """The oracle's UpsampleNetwork (oracle/wavernn_oracle.py `upsample_network`) up to the INPUT of its last stage, as [row][channel]:
    [(N + 2 pad) * 25, 80] -- what `wrnn_pre_upsample_rows` writes."""
    from oracle import wavernn_oracle as O
    x = O.pad_tensor(mel.T[None].astype(np.float32), pad, 'both')[0].T
    for li, s in enumerate(scales):
        x = np.repeat(x, s, axis=1)
        w = sd[f'upsample.up_layers.{2 * li + 1}.weight'].astype(np.float32).reshape(-1)
        xp = np.pad(x, ((0, 0), (s, s)))
        y = np.zeros_like(x)
        for j in range(2 * s + 1):
            y += w[j] * xp[:, j:j + x.shape[1]]
        x = y.astype(np.float32)
    return np.ascontiguousarray(x.T)


# ---------------------------------------------------------------------------------------------------------------------------------
# Full-size oracle references, one utterance at a time, with an on-disk cache (tests/_cache/, git-ignored, filled in the build container
# by scripts/make_oracle_cache.py; it travels to the GPU box with the snapshot).  TEST INFRASTRUCTURE: nothing under wavernn_amd/ reads it.
# ---------------------------------------------------------------------------------------------------------------------------------
CACHE = os.path.join(ROOT, 'tests', '_cache')


def _oracle_src_sha():
    import hashlib
    h = hashlib.sha256()
    for name in ('wrnn_oracle.c', 'wavernn_oracle.py'):
        h.update(open(os.path.join(ROOT, 'oracle', name), 'rb').read())
    return h.hexdigest()[:12]


def pruned_state_dict(mode, wseed, prune, linear=False):
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(wseed, mode=mode)
    if prune > 0:
        from wavernn_amd.prune import block_prune_state_dict
        sd, _ = block_prune_state_dict(sd, prune, (16, 1), linear=linear)
    return sd


def oracle_utterance(mode, wseed, prune, mel_seed, noise_seed, frames, target=11000, overlap=550, nthreads=8, want_cond=True, sd=None, linear=False):
    """The C oracle's loop output for ONE utterance generated the reference's way (random mel `mel_seed` of `frames` frames, weights
    `random_state_dict(wseed)` block-pruned to `prune`, batched fold, `torch.manual_seed(noise_seed)` noise stream): dict(ref = [B, T]
    float32 -- RAW: 2 idx / (C - 1) - 1 --, and with want_cond the oracle-side conditioning: mels_up [L, 80], aux [frames, 128], noise in
    the launch layout).  `ref` comes from tests/_cache when an entry made by the same oracle sources exists."""
    from oracle import c_oracle as C, wavernn_oracle as O
    from wavernn_amd.synthetic import random_mel
    key = f'{mode}_w{wseed}_p{int(round(prune * 1000))}{"L" if linear else ""}_m{mel_seed}_n{noise_seed}_f{frames}_t{target}_o{overlap}_{_oracle_src_sha()}'
    path = os.path.join(CACHE, key + '.npz')
    sd = pruned_state_dict(mode, wseed, prune, linear) if sd is None else sd
    mel = random_mel(mel_seed, frames)
    res = {}
    ref = None
    if os.path.exists(path):
        z = np.load(path)
        ref = z['idx'].astype(np.float32) * np.float32(2) / np.float32(z['classes'] - 1) - np.float32(1) if mode == 'RAW' else z['ref']
    if ref is None or want_cond:
        mels_f, aux_f, _ = O.conditioning(sd, mel, True, target, overlap)
        B, T = mels_f.shape[:2]
        if mode == 'RAW':                  # (torch's CPU generator: the stream the oracle's numpy model restates, 20x faster for 10^8 draws)
            import torch
            from wavernn_amd.rng import draw_noise
            nz = draw_noise('RAW', B, T, 512, 512, 32, 'cpu', 'cpu', generator=torch.Generator(device='cpu').manual_seed(noise_seed)).numpy()
        else:
            nz = O.draw_noise(noise_seed, mode, B, T)
        if ref is None:
            C.build()
            ref = C.loop(sd, mode, mels_f, aux_f, nz, nthreads=nthreads)
            os.makedirs(CACHE, exist_ok=True)
            tmp = path + f'.{os.getpid()}.tmp.npz'
            if mode == 'RAW':
                idx = np.rint((ref.astype(np.float64) + 1.0) * 511.0 / 2.0).astype(np.int16)
                assert np.array_equal(idx.astype(np.float32) * np.float32(2) / np.float32(511) - np.float32(1), ref)
                np.savez_compressed(tmp, idx=idx, classes=512)
            else:
                np.savez(tmp, ref=ref)
            os.replace(tmp, path)
        if want_cond:
            m = O.pad_tensor(mel.T[None], 2, 'both')[0].T
            mu, au = O.upsample_network(sd, m)
            res.update(mels_up=mu, aux=np.ascontiguousarray(au[::275]),
                       noise=nz if mode == 'RAW' else np.concatenate([nz[0].reshape(T, -1), nz[1].reshape(T, -1)], axis=1))
    res['ref'] = ref
    return res
