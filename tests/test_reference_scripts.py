"""INTEGRATION.md section 1 as a test: the reference's OWN `gen_wavernn.py` flow (`gen_from_file`, gen_wavernn.py:38-65) run
with the one-line import swap `from wavernn_amd.model import WaveRNN`, in the build container (needs /root/reference; skipped
on the GPU box).  No GPU here, so the two device entry points of the class are replaced by CPU stand-ins built on the oracle
(test infrastructure) -- what is under test is everything else the reference's script touches: constructor signature,
`load()` of a reference state dict, `get_step()`, the `.npy` checks, `generate()`'s signature / file naming / return
value / RNG consumption / train-eval side effects.  The result must equal the reference class run through the same script."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = os.environ.get('WRNN_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree is only present in the build container')


class _OracleEngine:
    """CPU stand-in with LoopEngine's face (run / plan / last_*): the oracle's C loop."""

    def __init__(self, sd, mode):
        self.sd, self.mode = sd, mode

    def plan(self, n, T, **kw):
        return dict(kernel='oracle')

    def run(self, mels_up, aux, B, T, stride, noise, hop, out=None, t_range=None, **kw):
        from oracle import c_oracle as C
        assert t_range is None
        mu, au, nz = mels_up.numpy(), aux.numpy(), noise.numpy()
        L = mu.shape[0]
        mels_f = np.zeros((B, T, mu.shape[1]), np.float32)
        aux_f = np.zeros((B, T, au.shape[1]), np.float32)
        for b in range(B):
            p = b * stride + np.arange(T)
            ok = p < L
            mels_f[b, ok] = mu[p[ok]]
            aux_f[b, ok] = au[p[ok] // hop]
        nzo = (np.ascontiguousarray(nz[:, :10 * B].reshape(T, B, 10)), np.ascontiguousarray(nz[:, 10 * B:])) if self.mode == 'MOL' \
            else np.ascontiguousarray(nz.reshape(T, B, -1))
        return torch.from_numpy(C.loop(self.sd, self.mode, mels_f, aux_f, nzo))

    def last_loop_ms(self):
        return 0.0

    def last_loop_kernel(self):
        return 'oracle'


@pytest.mark.parametrize('mode', ['RAW', 'MOL'])
def test_reference_gen_wavernn_runs_with_the_import_swap(mode, tmp_path, monkeypatch):
    sys.dont_write_bytecode = True
    monkeypatch.syspath_prepend(REF)
    lib = types.ModuleType('librosa')
    lib.output = types.SimpleNamespace(write_wav=lambda path, x, sr: np.save(str(path) + '.npy', np.asarray(x)))
    monkeypatch.setitem(sys.modules, 'librosa', lib)
    un = types.ModuleType('unidecode'); un.unidecode = lambda s: s                         # text front-end deps of utils.dataset
    inf = types.ModuleType('inflect'); inf.engine = lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'number')
    monkeypatch.setitem(sys.modules, 'unidecode', un)
    monkeypatch.setitem(sys.modules, 'inflect', inf)
    if not hasattr(np, 'cumproduct'):
        monkeypatch.setattr(np, 'cumproduct', np.cumprod, raising=False)
    for m in [k for k in sys.modules if k.split('.')[0] in ('utils', 'models', 'gen_wavernn')]:
        monkeypatch.delitem(sys.modules, m)
    from utils import hparams as hp
    hp.configure(os.path.join(REF, 'hparams.py'))
    import models.fatchord_version as ref_mod
    RefWaveRNN = ref_mod.WaveRNN
    RefWaveRNN.gen_display = lambda self, *a, **k: None
    import wavernn_amd.model as ours
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED

    kw = dict(rnn_dims=hp.voc_rnn_dims, fc_dims=hp.voc_fc_dims, bits=hp.bits, pad=hp.voc_pad, upsample_factors=hp.voc_upsample_factors,
              feat_dims=hp.num_mels, compute_dims=hp.voc_compute_dims, res_out_dims=hp.voc_res_out_dims, res_blocks=hp.voc_res_blocks,
              hop_length=hp.hop_length, sample_rate=hp.sample_rate, mode=mode)            # gen_wavernn.py:112-123
    assert {k: kw[k] for k in SHIPPED} == SHIPPED
    sd = random_state_dict(17, mode=mode)
    ckpt = tmp_path / 'latest_weights.pyt'
    ref_model = RefWaveRNN(**kw)
    ref_model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    ref_model.step += 123456
    ref_model.save(ckpt)                                                                  # a checkpoint as the reference writes it
    np.save(tmp_path / 'utt.npy', random_mel(5, 40))

    def run_script(WaveRNNClass, out_dir):
        # the reference's script with its model import swapped (gen_wavernn.py:3)
        monkeypatch.setattr(ref_mod, 'WaveRNN', WaveRNNClass)
        monkeypatch.delitem(sys.modules, 'gen_wavernn', raising=False)
        import gen_wavernn as G
        assert G.WaveRNN is WaveRNNClass
        model = G.WaveRNN(**kw)
        model.load(ckpt)                                                                  # gen_wavernn.py:129
        os.makedirs(out_dir)
        torch.manual_seed(4242)
        G.gen_from_file(model, tmp_path / 'utt.npy', out_dir, True, 1100, 55)             # gen_wavernn.py:38-65
        assert model.training                                                             # generate() leaves train mode on (:262)
        after = torch.empty(3).uniform_(0, 1).numpy()                                     # the generator moved exactly as far
        (f,) = [p for p in os.listdir(out_dir) if p.endswith('.npy')]
        return f, np.load(os.path.join(out_dir, f)), after

    ref_name, ref_wav, ref_after = run_script(RefWaveRNN, tmp_path / 'ref')

    # ours, with the two device entry points mocked onto the CPU
    monkeypatch.setattr(ours.WaveRNN, '_require_hip_device', staticmethod(lambda device: None))
    monkeypatch.setattr(ours.WaveRNN, '_loop_engine', lambda self: _OracleEngine({k: v.detach().numpy() for k, v in self.state_dict().items()}, self.mode))
    monkeypatch.setattr(ours, 'save_wav', lambda x, path, sr: lib.output.write_wav(path, x.astype(np.float32), sr))

    class Swapped(ours.WaveRNN):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.pre_algo, self.post_algo = 'torch', 'numpy'

    name, wav, after = run_script(Swapped, tmp_path / 'ours')
    assert name == ref_name == '__utt__123k_steps_gen_batched_target1100_overlap55.wav.npy'
    assert np.array_equal(after, ref_after)
    if mode == 'RAW':
        assert np.array_equal(wav, ref_wav)
    else:
        assert np.abs(wav - ref_wav).max() <= 1e-5
    with pytest.raises(ValueError):                                                       # gen_wavernn.py:50-55 still guards the input
        np.save(tmp_path / 'bad.npy', 2 * random_mel(5, 40))
        import gen_wavernn as G
        G.gen_from_file(Swapped(**kw), tmp_path / 'bad.npy', tmp_path / 'ours', True, 1100, 55)
