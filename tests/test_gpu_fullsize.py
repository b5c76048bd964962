"""GPU parity at BASELINE's FULL sizes (-m gpu): the HIP loop against the oracle / the reference's own output at
T = 12,100 steps -- config 2 at its stated inputs (weight seed 0, mel seed 1234, N = 481 -> B = 12, sample
seed 77), config 3's vocoder call (mel from the reference's Tacotron, N = 800 -> B = 19 exact fit), the bench geometry
(8 utterances x 641 frames = 128 segments in ONE launch), an 8-utterance slice of config 4's corpus, and config 5's
block-sparse kernel.  RAW bit-exact, MoL <= MOL_TOL.  /root/reference is never read here: the reference's outputs are the
committed fixtures of scripts/make_golden.py.
"""
import numpy as np
import pytest
import torch

from helpers import BIG_CASES, MOL_TOL, load_case, case_mel

pytestmark = pytest.mark.gpu

HOP, TARGET, OVERLAP = 275, 11000, 550


@pytest.fixture(scope='module')
def gpu():
    assert torch.cuda.is_available(), 'these tests need a HIP device'
    from wavernn_amd import _lib
    _lib.lib()
    return torch.device('cuda', 0)


def _model(sd, mode, gpu):
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import SHIPPED
    model = WaveRNN(**SHIPPED, mode=mode)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return model.to(gpu)


@pytest.mark.parametrize('name', BIG_CASES)
def test_generate_full_size_matches_reference(gpu, name, tmp_path):
    """`WaveRNN.generate()` (all-HIP path, `auto` kernel) vs the waveform the reference itself returned for the same
    weights / mel / `torch.manual_seed` -- BASELINE configs 2 and 3 (vocoder side) at T = 12,100."""
    from wavernn_amd.synthetic import random_state_dict
    cfg, g = load_case(name)
    sd = random_state_dict(cfg['wseed'], mode=cfg['mode'])
    model = _model(sd, cfg['mode'], gpu)
    mel = case_mel(cfg, g)
    torch.manual_seed(cfg['seed'])
    out = model.generate(torch.tensor(mel).unsqueeze(0), tmp_path / 'o.wav', cfg['batched'], cfg['target'], cfg['overlap'], cfg['mu_law'])
    print(f'{name}: loop {model.last_loop_kernel} {model.last_loop_ms:.1f} ms for {g["raw"].shape} segment-steps')
    assert out.dtype == np.float64 and out.shape == g['out'].shape
    if cfg['mode'] == 'RAW':
        assert np.array_equal(out, g['out']), f'{np.count_nonzero(out != g["out"])} of {out.size} samples differ'
    else:
        assert np.abs(out - g['out']).max() <= MOL_TOL, np.abs(out - g['out']).max()


def _corpus_inputs(sd, mode, frames, mel_seeds, noise_seeds):
    """Oracle-side conditioning + noise + reference segments for a batch of utterances (one C.loop call per utterance)."""
    from oracle import c_oracle as C, wavernn_oracle as O
    from wavernn_amd.synthetic import random_mel
    from wavernn_amd.batch import plan_utterances, pack_noise
    ups, auxs, refs, noises = [], [], [], []
    for n, ms, ns in zip(frames, mel_seeds, noise_seeds):
        mel = random_mel(ms, n)
        m = O.pad_tensor(mel.T[None], 2, 'both')[0].T
        mu, au = O.upsample_network(sd, m)
        ups.append(mu)
        auxs.append(np.ascontiguousarray(au[::HOP]))
        mels_f, aux_f, _ = O.conditioning(sd, mel, True, TARGET, OVERLAP)
        nz = O.draw_noise(ns, mode, mels_f.shape[0], mels_f.shape[1])
        noises.append(nz if mode == 'RAW' else np.concatenate([nz[0].reshape(mels_f.shape[1], -1), nz[1].reshape(mels_f.shape[1], -1)], axis=1))
        refs.append(C.loop(sd, mode, mels_f, aux_f, nz))
    plan = plan_utterances([n * HOP for n in frames], TARGET, OVERLAP)
    flat = pack_noise(mode, plan, noises)
    return plan, np.concatenate(ups), np.concatenate(auxs), flat, refs


def test_bench_geometry_matches_oracle(gpu):
    """The bench workload (bench.py: 8 utterances x 641 frames, weight seed 0 -> 128 folded segments x 12,100 steps in ONE
    launch, the kernel `auto` picks) against the C oracle run per utterance on the same conditioning and noise."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(0, mode='MOL')
    frames = [641] * 8
    plan, mels_up, aux, flat, refs = _corpus_inputs(sd, 'MOL', frames, [1234 + u for u in range(8)], [77 + u for u in range(8)])
    assert plan.n_segments == 128 and plan.T == 12100
    eng = LoopEngine(sd, 'MOL', device=gpu)
    out = eng.run_segments(torch.from_numpy(mels_up).to(gpu), torch.from_numpy(aux).to(gpu), plan.seg_pos, plan.seg_lim, plan.T,
                           torch.from_numpy(flat).to(gpu), HOP, algo='auto').cpu().numpy()
    print(f'bench geometry: {eng.last_loop_kernel()} split {eng.last_loop_split()} {eng.last_loop_ms():.1f} ms')
    worst = 0.0
    for u, ref in enumerate(refs):
        got = out[plan.first[u]:plan.first[u] + plan.folds[u]]
        worst = max(worst, float(np.abs(got - ref).max()))
    assert worst <= MOL_TOL, worst


def test_corpus_slice_matches_per_utterance_oracle(gpu):
    """The first 8 utterances of BASELINE config 4's corpus (lens from RandomState(2024), mel seeds 1000+u) through
    `generate_corpus` (one launch, parity noise) vs the oracle's end-to-end `generate` per utterance."""
    from oracle import wavernn_oracle as O, c_oracle as C
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.synthetic import random_state_dict, random_mel
    sd = random_state_dict(0, mode='MOL')
    model = _model(sd, 'MOL', gpu)

    def oracle_generate(mel, seed):                 # O.generate with the C twin of its loop (the numpy loop takes minutes here)
        mels_f, aux_f, wave_len = O.conditioning(sd, mel, True, TARGET, OVERLAP)
        raw = C.loop(sd, 'MOL', mels_f, aux_f, O.draw_noise(seed, 'MOL', mels_f.shape[0], mels_f.shape[1]))
        return O.finish(raw, 'MOL', 30, wave_len, True, TARGET, OVERLAP, True)
    lens = np.random.RandomState(2024).randint(300, 901, 64)[:8]
    mels = [random_mel(1000 + u, int(n)) for u, n in enumerate(lens)]
    seeds = [4000 + u for u in range(8)]
    outs = generate_corpus(model, [torch.from_numpy(m).unsqueeze(0) for m in mels], TARGET, OVERLAP, True, seeds)
    for u, mel in enumerate(mels):
        ref = oracle_generate(mel, seeds[u])
        assert outs[u].shape == ref.shape
        assert np.abs(outs[u] - ref).max() <= MOL_TOL, (u, np.abs(outs[u] - ref).max())


def test_block_sparse_kernel_full_length_matches_oracle(gpu):
    """BASELINE config 5 at T = 12,100: two 641-frame utterances (32 segments) on 95 %-block-pruned GRU weights through
    `wrnn_sparse_kernel` vs the C oracle on the same masked dense weights."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.prune import block_prune_state_dict
    from wavernn_amd.synthetic import random_state_dict
    sd, _ = block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1))
    plan, mels_up, aux, flat, refs = _corpus_inputs(sd, 'MOL', [641, 641], [1234, 1235], [77, 78])
    eng = LoopEngine(sd, 'MOL', device=gpu)
    out = eng.run_segments(torch.from_numpy(mels_up).to(gpu), torch.from_numpy(aux).to(gpu), plan.seg_pos, plan.seg_lim, plan.T,
                           torch.from_numpy(flat).to(gpu), HOP, algo='auto').cpu().numpy()
    assert eng.last_loop_kernel() == 'wrnn_sparse_kernel'
    for u, ref in enumerate(refs):
        got = out[plan.first[u]:plan.first[u] + plan.folds[u]]
        assert np.abs(got - ref).max() <= MOL_TOL, (u, np.abs(got - ref).max())
