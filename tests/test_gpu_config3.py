"""BASELINE config 3 end to end on the GPU (-m gpu): `gen_tacotron.py wavernn` = Tacotron.generate -> (m + 4) / 8, clip ->
WaveRNN.generate batched (reference gen_tacotron.py:139-166).  The Tacotron side is `wavernn_amd.tacotron.TacotronInference`
(functional PyTorch-ROCm restatement, pinned bit-exactly to the reference on the CPU in tests/test_tacotron_mirror.py); here:
its decoder-loop kernels against the REFERENCE's own output for the same weights (tests/golden/tacotron_decoder_200f.npz), the CBHG
GRU kernel, and the hand-off into the HIP vocoder end to end (random-init weights of the reference's architecture -- the shape table
is a committed fixture; vocoder parity for this hand-off is tests/test_gpu_fullsize.py::...[mol_tacotron_800f])."""
import json
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _tts(dev, seed=3, **override):
    from wavernn_amd.synthetic import random_tacotron_state_dict
    from wavernn_amd.tacotron import TacotronInference
    shapes = json.load(open(os.path.join(HERE, 'golden', 'tacotron_shapes.json')))
    sd = random_tacotron_state_dict(seed, shapes)
    sd.update(override)
    return TacotronInference(sd, device=dev)


IDS_TEXT = 'Scientists at the CERN laboratory say they have discovered a new particle.'


#: decoder-kernel tolerance against the REFERENCE's own output over all 200 frames (mel |x| <= 0.016, attention rows sum to 1): the kernels
#: were measured at <= 7.5e-9 (profiles/r03ad_taco_profile.json); 1e-6 leaves room for another summation order, not for a wrong term
TACO_TOL = 1e-6


def _decoder_golden():
    g = np.load(os.path.join(HERE, 'golden', 'tacotron_decoder_200f.npz'))
    return g['mel'], g['linear'], g['attention'], [int(i) for i in g['ids']]


@pytest.mark.parametrize('variant', [2, 1], ids=['resident', 'flag-barrier'])
def test_tacotron_decoder_kernel_matches_the_reference(variant):
    """the decoder loop as ONE persistent kernel (csrc/wrnn_taco.hip, `wrnn_taco_decode`; both forms: the
    register-resident kernel with the tagged exchange that `auto` picks on a >= 128-CU device, and the flag-barrier kernel) against
    what the REFERENCE's `Tacotron.generate` (models/tacotron.py:370-430) itself returned for these weights and this sentence --
    tests/golden/tacotron_decoder_200f.npz, written by scripts/make_golden.py from /root/reference in the build container: decoder
    mel and attention of all 200 frames within TACO_TOL, and the post-net output (encoder / post-net CBHG with `wrnn_bigru`)
    within the same bound.  Also against the CPU mirror and the eager PyTorch-ROCm loop on this box."""
    from wavernn_amd.tacotron import text_to_ids
    dev = torch.device('cuda', 0)
    ref_mel, ref_lin, ref_attn, ref_ids = _decoder_golden()
    ids = text_to_ids(IDS_TEXT)
    assert ids == ref_ids
    steps = ref_mel.shape[1]
    mel_c, lin_c, attn_c = _tts('cpu').generate(ids, steps=steps)                 # the mirror of the reference, on this host
    tts = _tts(dev)
    mel_e, lin_e, attn_e = tts.generate(ids, steps=steps)
    tts.generate(ids, steps=8, kernel=True, kernel_variant=variant)                # warm-up (module load, workspace)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mel_k, lin_k, attn_k = tts.generate(ids, steps=steps, kernel=True, kernel_variant=variant)
    torch.cuda.synchronize()
    t_k = time.perf_counter() - t0
    assert mel_k.shape == ref_mel.shape == (80, steps) and attn_k.shape == ref_attn.shape
    d = np.abs(ref_mel - mel_k).max(axis=0)
    da = np.abs(ref_attn - attn_k).max()
    dl = np.abs(ref_lin - lin_k).max()
    print(f'decoder kernel (variant {variant}): {steps} frames in {t_k * 1e3:.1f} ms incl. encoder + post-net; vs the reference: max |d mel| per frame',
          d[:3], '...', d[-3:], f'attention {da:.2e} post-net {dl:.2e}; CPU mirror on this host vs the reference {np.abs(ref_mel - mel_c).max():.2e}')
    assert d.max() <= TACO_TOL, d.max()
    assert da <= TACO_TOL, da
    assert dl <= TACO_TOL, dl
    # the mirror on this host and the eager loop on the device follow the same trajectory
    assert np.abs(ref_mel - mel_c).max() <= TACO_TOL and np.abs(ref_attn - attn_c).max() <= TACO_TOL
    assert np.abs(mel_e - mel_k).max() <= 1e-5 and np.abs(attn_e - attn_k).max() <= 1e-5
    np.testing.assert_allclose(attn_k.sum(axis=1), 1.0, atol=1e-5)


@pytest.mark.parametrize('variant', [2, 1], ids=['resident', 'flag-barrier'])
def test_tacotron_decoder_kernel_stop_test_and_r(variant):
    """The stop test of models/tacotron.py:411 (`(mel_frames < stop_threshold).all() and t > 10`) is evaluated inside the kernel:
    with a threshold every frame is below, the loop must end at the first t > 10 -- the same frame count as the eager loop and
    the CPU mirror -- and with r = 2 frames per decoder step (the `[:, :, :r]` view of mel_proj, :262) the kernel still agrees."""
    from wavernn_amd.tacotron import text_to_ids
    dev = torch.device('cuda', 0)
    ids = text_to_ids('Hello there.')
    stop = dict(stop_threshold=torch.tensor(1e9))
    mel_c, _, attn_c = _tts('cpu', **stop).generate(ids, steps=100)
    tts = _tts(dev, **stop)
    mel_e, _, _ = tts.generate(ids, steps=100)
    mel_k, _, attn_k = tts.generate(ids, steps=100, kernel=True, kernel_variant=variant)
    assert mel_c.shape == mel_e.shape == mel_k.shape == (80, 12), (mel_c.shape, mel_e.shape, mel_k.shape)      # t = 0 .. 11: the first t > 10
    assert np.abs(mel_c - mel_k).max() <= TACO_TOL and attn_k.shape == attn_c.shape
    r2 = dict(r=torch.tensor(2))
    if 'decoder.r' in _tts('cpu').p:
        r2 = {'decoder.r': torch.tensor(2)}
    mel_c, _, attn_c = _tts('cpu', **r2).generate(ids, steps=40)
    mel_k, _, attn_k = _tts(dev, **r2).generate(ids, steps=40, kernel=True, kernel_variant=variant)
    assert mel_c.shape == mel_k.shape == (80, 40) and attn_c.shape == attn_k.shape == (20, len(ids))
    assert np.abs(mel_c - mel_k).max() <= TACO_TOL and np.abs(attn_c - attn_k).max() <= TACO_TOL
    # more encoder positions than waves (n > 512: the second position of a wave, 16-deep context sums)
    long_ids = text_to_ids(' '.join([IDS_TEXT] * 9))
    assert 512 < len(long_ids) <= 1024
    mel_c, _, attn_c = _tts('cpu').generate(long_ids, steps=12)
    mel_k, _, attn_k = _tts(dev).generate(long_ids, steps=12, kernel=True, kernel_variant=variant)
    assert mel_c.shape == mel_k.shape and attn_c.shape == attn_k.shape == (12, len(long_ids))
    assert np.abs(mel_c - mel_k).max() <= TACO_TOL and np.abs(attn_c - attn_k).max() <= TACO_TOL


@pytest.mark.parametrize('T', [1, 74, 800])
def test_cbhg_bigru_kernel_matches_torch_gru(T):
    """`wrnn_bigru` (one persistent workgroup per direction, W_hh in registers) against `nn.GRU(128, 128, bidirectional=True)`'s
    functional form on the device (MIOpen) and on the CPU (ATen: what the reference runs, models/tacotron.py:95 / :137): another
    summation order, so a tolerance -- 2e-6 on |h| <= 1 over 800 recurrent steps."""
    dev = torch.device('cuda', 0)
    tts = _tts(dev)
    q = 'postnet.rnn.'
    flat = [tts.p[q + n_] for n_ in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse',
                                      'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse')]
    g = torch.Generator().manual_seed(T)
    x = torch.randn(1, T, 128, generator=g)
    hx = torch.zeros(2, 1, 128)
    ref_cpu, _ = torch._VF.gru(x, hx, [f.cpu() for f in flat], True, 1, 0.0, False, True, True)
    ref_dev, _ = torch._VF.gru(x.to(dev), hx.to(dev), flat, True, 1, 0.0, False, True, True)
    out = tts._bigru(x.to(dev), flat)
    torch.cuda.synchronize()
    assert out.shape == ref_cpu.shape == (1, T, 256)
    assert (out.cpu() - ref_cpu).abs().max().item() <= 2e-6, (out.cpu() - ref_cpu).abs().max().item()
    assert (out - ref_dev).abs().max().item() <= 2e-6


def test_config3_end_to_end_with_the_decoder_kernel(tmp_path):
    """BASELINE config 3 end to end on one MI355X with the decoder loop as a persistent kernel: Tacotron (encoder + decoder kernel +
    post-net) -> `_, m, _` (gen_tacotron.py:142) -> (m + 4) / 8, clip -> the HIP vocoder, timed."""
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, SHIPPED
    from wavernn_amd.tacotron import text_to_ids, tacotron_to_wavernn_mel
    dev = torch.device('cuda', 0)
    tts = _tts(dev)
    ids = text_to_ids(IDS_TEXT)
    voc = WaveRNN(**SHIPPED, mode='MOL')
    voc.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in random_state_dict(0, mode='MOL').items()}, strict=True)
    voc = voc.to(dev)
    voc.noise_source = 'device'
    steps = 800

    def run():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, lin, _ = tts.generate(ids, steps=steps, kernel=True)
        t1 = time.perf_counter()
        wav = voc.generate(torch.tensor(tacotron_to_wavernn_mel(lin)).unsqueeze(0), tmp_path / 'o.wav', True, 11_000, 550, True)
        return wav, t1 - t0, time.perf_counter() - t1
    run()
    wav, t_tts, t_voc = run()
    assert wav.shape == ((steps - 1) * 275,) and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
    audio_s = wav.shape[0] / 22050
    print(f'config 3 end to end with the decoder kernel: Tacotron {t_tts * 1e3:.0f} ms ({steps} decoder steps) + vocoder {t_voc * 1e3:.0f} ms '
          f'({voc.last_loop_kernel} {voc.last_loop_ms:.0f} ms) for {audio_s:.2f} s of audio = {audio_s / (t_tts + t_voc):.1f}x real time')
