"""BASELINE config 3 end to end on the GPU (-m gpu): `gen_tacotron.py wavernn` = Tacotron.generate -> (m + 4) / 8, clip ->
WaveRNN.generate batched (reference gen_tacotron.py:139-166).  The Tacotron side is `wavernn_amd.tacotron.TacotronInference`
(functional PyTorch-ROCm restatement, pinned bit-exactly to the reference on the CPU in tests/test_tacotron_mirror.py); here:
its HIP-graph decoder loop equals its eager loop on the device, and the hand-off into the HIP vocoder runs (random-init
weights of the reference's architecture -- the shape table is a committed fixture; vocoder parity for this hand-off is
tests/test_gpu_fullsize.py::...[mol_tacotron_800f])."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_tacotron_graph_loop_and_vocoder_handoff(tmp_path):
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_tacotron_state_dict, SHIPPED
    from wavernn_amd.tacotron import TacotronInference, text_to_ids, tacotron_to_wavernn_mel
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    shapes = json.load(open(os.path.join(HERE, 'golden', 'tacotron_shapes.json')))
    tts = TacotronInference(random_tacotron_state_dict(3, shapes), device=dev)
    ids = text_to_ids('Scientists at the CERN laboratory say they have discovered a new particle.')
    assert len(ids) == 74                                              # first line of sentences.txt
    steps = 800
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mel_e, lin_e, attn_e = tts.generate(ids, steps=steps)
    t_eager = time.perf_counter() - t0
    t0 = time.perf_counter()
    mel_g, lin_g, attn_g = tts.generate(ids, steps=steps, graph=True, stop_check_every=64)
    t_graph = time.perf_counter() - t0
    assert mel_e.shape == mel_g.shape == (80, steps) and attn_e.shape == (steps, 74)
    # the replayed graph runs the eager loop's kernels; hipBLASLt may pick another algorithm under capture, and 800 recurrent steps
    # amplify rounding differences, so: tight on the first frames, loose (same trajectory) on the whole utterance
    d = np.abs(mel_e - mel_g).max(axis=0)
    print('graph vs eager, max |d mel| per frame:', d[:4], '...', d[-4:])
    assert d[:16].max() <= 1e-5, d[:16]
    assert d.max() <= 5e-2 and np.abs(attn_e - attn_g).max() <= 5e-2
    # hand-off (gen_tacotron.py:143-163): L = 800 * 275 = 19 * 11550 + 550 exactly -> 19 folds, no padded fold
    voc = WaveRNN(**SHIPPED, mode='MOL')
    voc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in random_state_dict(0, mode='MOL').items()}, strict=True)
    voc = voc.to(dev)
    voc.noise_source = 'device'
    m = torch.tensor(tacotron_to_wavernn_mel(lin_g)).unsqueeze(0)        # `_, m, attention = tts_model.generate(x)` (:142): the postnet output
    voc.generate(m, tmp_path / 'w.wav', True, 11_000, 550, True)       # warm-up (weight packs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wav = voc.generate(m, tmp_path / 'o.wav', True, 11_000, 550, True)
    t_voc = time.perf_counter() - t0
    assert wav.shape == ((steps - 1) * 275,) and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
    audio_s = wav.shape[0] / 22050
    print(f'config 3 on one MI355X: Tacotron {steps} frames eager {t_eager * 1e3:.0f} ms / HIP-graph decoder loop {t_graph * 1e3:.0f} ms; '
          f'vocoder {t_voc * 1e3:.0f} ms (loop {voc.last_loop_kernel} {voc.last_loop_ms:.0f} ms) for {audio_s:.2f} s of audio = '
          f'{audio_s / (t_graph + t_voc):.1f}x real time end to end')
