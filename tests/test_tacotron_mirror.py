"""`wavernn_amd.tacotron.TacotronInference` (the functional, state-dict-driven restatement of the reference's
`Tacotron.generate()`, BASELINE config 3's caller side) against the reference class itself, on the CPU, in the build
container (needs /root/reference; skipped on the GPU box).  Same weights, same ids -> bit-identical mel / linear / attention,
and the mel committed in tests/golden/mol_tacotron_800f.npz (the input of config 3's vocoder parity tests)."""
import os
import sys
import types

import numpy as np
import pytest
import torch

REF = os.environ.get('WRNN_REFERENCE', '/root/reference')
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree is only present in the build container')


def _reference_tacotron(monkeypatch, seed):
    sys.dont_write_bytecode = True
    monkeypatch.syspath_prepend(REF)
    for name, attrs in (('librosa', dict(output=types.SimpleNamespace(write_wav=lambda *a, **k: None))),
                        ('unidecode', dict(unidecode=lambda s: s)),
                        ('inflect', dict(engine=lambda: types.SimpleNamespace(number_to_words=lambda *a, **k: 'number')))):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
    for m in [k for k in sys.modules if k.split('.')[0] in ('utils', 'models')]:
        monkeypatch.delitem(sys.modules, m)
    from utils import hparams as hp
    hp.configure(os.path.join(REF, 'hparams.py'))
    from models.tacotron import Tacotron
    from utils.text.symbols import symbols
    from utils.text import text_to_sequence
    torch.manual_seed(seed)
    tts = Tacotron(embed_dims=hp.tts_embed_dims, num_chars=len(symbols), encoder_dims=hp.tts_encoder_dims, decoder_dims=hp.tts_decoder_dims,
                   n_mels=hp.num_mels, fft_bins=hp.num_mels, postnet_dims=hp.tts_postnet_dims, encoder_K=hp.tts_encoder_K,
                   lstm_dims=hp.tts_lstm_dims, postnet_K=hp.tts_postnet_K, num_highways=hp.tts_num_highways, dropout=hp.tts_dropout,
                   stop_threshold=hp.tts_stop_threshold)
    with open(os.path.join(REF, 'sentences.txt')) as f:
        line = f.readline().strip()
    return tts, text_to_sequence(line, hp.tts_cleaner_names), line


def test_functional_tacotron_equals_the_reference(monkeypatch):
    from wavernn_amd.tacotron import TacotronInference, text_to_ids, tacotron_to_wavernn_mel
    tts, ids, line = _reference_tacotron(monkeypatch, seed=3)
    assert text_to_ids(line) == ids                                 # plain text: basic cleaning == the reference's english_cleaners
    with torch.no_grad():
        ref_mel, ref_lin, ref_attn = tts.generate(ids, steps=60)
    mine = TacotronInference(tts.state_dict())
    mel, lin, attn = mine.generate(ids, steps=60)
    assert mel.shape == ref_mel.shape == (80, 60) and attn.shape == ref_attn.shape == (60, len(ids))
    assert np.array_equal(mel, ref_mel) and np.array_equal(lin, ref_lin) and np.array_equal(attn, ref_attn)
    # r > 1 (the reference's progressive schedule trains with r = 7 .. 2): several frames per decoder step
    tts.r = 3
    with torch.no_grad():
        ref_mel3, _, ref_attn3 = tts.generate(ids, steps=30)
    mel3, _, attn3 = TacotronInference(tts.state_dict()).generate(ids, steps=30)
    assert mel3.shape == ref_mel3.shape == (80, 30) and np.array_equal(mel3, ref_mel3) and np.array_equal(attn3, ref_attn3)
    assert tacotron_to_wavernn_mel(mel).min() >= 0 and tacotron_to_wavernn_mel(mel).max() <= 1


def test_config3_fixture_mel_comes_from_this_tacotron(monkeypatch):
    """the mel stored in tests/golden/mol_tacotron_800f.npz is what the reference's gen_tacotron.py hands its vocoder: the
    SECOND return of `Tacotron.generate` (`_, m, attention = tts_model.generate(x)`, gen_tacotron.py:142 -- the postnet /
    post_proj output, 80 bins because fft_bins = hp.num_mels), rescaled and clipped (:143-145).  The mirror's second output,
    through `tacotron_to_wavernn_mel`, reproduces the stored fixture exactly (round-2 advisor: this test compared the raw
    decoder mel -- the FIRST return -- and hid the 0.04 difference behind a 0.25 tolerance)."""
    from wavernn_amd.tacotron import TacotronInference, tacotron_to_wavernn_mel
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    from helpers import load_case
    cfg, g = load_case('mol_tacotron_800f')
    tts, ids, _ = _reference_tacotron(monkeypatch, seed=cfg['tts_seed'])
    with torch.no_grad():
        ref_mel, ref_lin, _ = tts.generate(ids, steps=cfg['frames'])
    mel, lin, _ = TacotronInference(tts.state_dict()).generate(ids, steps=cfg['frames'])
    assert np.array_equal(mel, ref_mel) and np.array_equal(lin, ref_lin)
    assert lin.shape == (80, cfg['frames'])
    assert np.array_equal(tacotron_to_wavernn_mel(lin).astype(np.float32), g['mel'])
