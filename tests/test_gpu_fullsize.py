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


@pytest.mark.parametrize('mel_in_loop', [None, True], ids=['mel-default', 'mel-in-loop'])
@pytest.mark.parametrize('name', BIG_CASES)
def test_generate_full_size_matches_reference(gpu, name, mel_in_loop, tmp_path):
    """`WaveRNN.generate()` (all-HIP path, `auto` kernel) vs the waveform the reference itself returned for the same
    weights / mel / `torch.manual_seed` -- BASELINE configs 2 and 3 (vocoder side) at T = 12,100.  `mel-in-loop`: the opt-in form without
    the [L, 80] up-sampled mel (`model.mel_in_loop = True`; MoL cases: RAW is compared class index by class index on the default path)."""
    from wavernn_amd.synthetic import random_state_dict
    cfg, g = load_case(name)
    if mel_in_loop and cfg['mode'] == 'RAW':
        pytest.skip('RAW: the materialised mel (model.mel_in_loop doc; the flip-rate measurement covers the in-loop form)')
    sd = random_state_dict(cfg['wseed'], mode=cfg['mode'])
    model = _model(sd, cfg['mode'], gpu)
    model.mel_in_loop = mel_in_loop
    mel = case_mel(cfg, g)
    torch.manual_seed(cfg['seed'])
    out = model.generate(torch.tensor(mel).unsqueeze(0), tmp_path / 'o.wav', cfg['batched'], cfg['target'], cfg['overlap'], cfg['mu_law'])
    print(f'{name}: loop {model.last_loop_kernel} {model.last_loop_ms:.1f} ms for {g["raw"].shape} segment-steps')
    assert out.dtype == np.float64 and out.shape == g['out'].shape
    if cfg['mode'] == 'RAW':
        assert np.array_equal(out, g['out']), f'{np.count_nonzero(out != g["out"])} of {out.size} samples differ'
    else:
        assert np.abs(out - g['out']).max() <= MOL_TOL, np.abs(out - g['out']).max()


def _pool_map(fn, items, threads_each=8):
    """Run the per-utterance oracle calls side by side (ctypes releases the GIL; every call is its own OpenMP team of
    `threads_each` threads: the C loop is barrier-bound, so 16 small teams beat one wide team by an order of magnitude)."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    width = max(1, min(len(items), (os.cpu_count() or 8) // threads_each))
    with ThreadPoolExecutor(width) as ex:
        return list(ex.map(fn, items))


def _corpus_inputs(sd, mode, frames, mel_seeds, noise_seeds):
    """Oracle-side conditioning + noise + reference segments for a batch of utterances (one C.loop call per utterance)."""
    from oracle import c_oracle as C, wavernn_oracle as O
    from wavernn_amd.synthetic import random_mel
    from wavernn_amd.batch import plan_utterances, pack_noise
    C.build()

    def one(args):
        n, ms, ns = args
        mel = random_mel(ms, n)
        m = O.pad_tensor(mel.T[None], 2, 'both')[0].T
        mu, au = O.upsample_network(sd, m)
        mels_f, aux_f, _ = O.conditioning(sd, mel, True, TARGET, OVERLAP)
        nz = O.draw_noise(ns, mode, mels_f.shape[0], mels_f.shape[1])
        flat = nz if mode == 'RAW' else np.concatenate([nz[0].reshape(mels_f.shape[1], -1), nz[1].reshape(mels_f.shape[1], -1)], axis=1)
        return mu, np.ascontiguousarray(au[::HOP]), flat, C.loop(sd, mode, mels_f, aux_f, nz, nthreads=8)
    res = _pool_map(one, list(zip(frames, mel_seeds, noise_seeds)))
    ups, auxs, noises, refs = (list(x) for x in zip(*res))
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


@pytest.mark.parametrize('algo', ['auto', 'loop', 'duo'])
def test_bench_workload_256_segments_matches_oracle(gpu, algo):
    """THE driver's headline workload (bench.py defaults: 16 utterances x 641 frames, weight seed 0, mel seeds 1234+u ->
    256 folded segments x 12,100 steps in ONE call, whatever kernel and split `auto` picks: today 4 clusters x 4 groups in
    flight, conditioning slabs of 192 steps) with parity noise (seeds 77+u), value-checked against the C oracle run per
    utterance.  Round-2 verdict: the benchmarked split had no oracle parity."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(0, mode='MOL')
    frames = [641] * 16
    if 'bench256' not in _SWEEP:
        _SWEEP.clear()
        _SWEEP['bench256'] = _corpus_inputs(sd, 'MOL', frames, [1234 + u for u in range(16)], [77 + u for u in range(16)])
    plan, mels_up, aux, flat, refs = _SWEEP['bench256']
    assert plan.n_segments == 256 and plan.T == 12100
    eng = LoopEngine(sd, 'MOL', device=gpu)
    want = eng.plan(256, plan.T, algo=algo)
    out = eng.run_segments(torch.from_numpy(mels_up).to(gpu), torch.from_numpy(aux).to(gpu), plan.seg_pos, plan.seg_lim, plan.T,
                           torch.from_numpy(flat).to(gpu), HOP, algo=algo).cpu().numpy()
    info = eng.last_run_info()
    print(f'bench workload [{algo}]: {info} {eng.last_loop_ms():.1f} ms')
    # (the plan does not know the hop: its slab length is an upper bound -- the duo kernel's per-slab aux tables cover 6 hops + 1 steps)
    assert info['kernel'] == want['kernel'] and (info['clusters'], info['depth']) == (want['clusters'], want['depth']) and info['slab_steps'] <= want['slab_steps']
    assert info['clusters'] * info['depth'] * 16 >= 256 and info['rounds'] == 1          # all 256 segments in flight at once
    worst = 0.0
    for u, ref in enumerate(refs):
        got = out[plan.first[u]:plan.first[u] + plan.folds[u]]
        worst = max(worst, float(np.abs(got - ref).max()))
    assert worst <= MOL_TOL, worst


_SWEEP = {}


def _sweep_case(sd, mode, n, T, seed):
    """Synthetic segment table for the split sweep: n segments over random conditioning, segment b starts at 37 b (not a
    multiple of the hop), the last three run into the zero padding of the fold (seg_lim), noise from a seeded CPU generator;
    the reference is the C oracle on the gathered conditioning.  Memoised: the splits of one (mode, n) share it."""
    from helpers import oracle_loop_fn
    key = (mode, n, T, seed)
    if key not in _SWEEP:
        rs = np.random.RandomState(seed)
        stride = 37
        L = (((n - 1) * stride + T) // HOP + 1) * HOP
        mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32))
        aux = torch.from_numpy(rs.uniform(-1, 1, (L // HOP, 128)).astype(np.float32))
        seg_pos = (np.arange(n) * stride).astype(np.int32)
        seg_lim = np.full(n, L, np.int32)
        seg_lim[-3:] = seg_pos[-3:] + np.array([T // 2, T - 1, 1], np.int32)       # ragged tails: zero conditioning from there on
        g = torch.Generator(device='cpu').manual_seed(seed)
        if mode == 'MOL':
            noise = torch.empty(T, 11 * n).uniform_(1e-5, 1 - 1e-5, generator=g)
        else:
            noise = torch.empty(T, n, 512).exponential_(1, generator=g)
        ref = oracle_loop_fn(sd, mode)(mels_up, aux, seg_pos, seg_lim, T, noise, HOP).numpy()
        _SWEEP.clear()                                                              # keep one case resident (RAW noise is ~100 MB)
        _SWEEP[key] = (mels_up, aux, seg_pos, seg_lim, noise, ref)
    return _SWEEP[key]


@pytest.mark.parametrize('mode,algo', [('MOL', 'loop'), ('RAW', 'loop'), ('MOL', 'duo'), ('RAW', 'duo'), ('MOL', 'chain'), ('RAW', 'chain')])
@pytest.mark.parametrize('clusters', [1, 4])
def test_every_depth_the_planner_can_pick(gpu, mode, clusters, algo):
    """Depth 4, 5, 6, 7, 8 groups in flight per cluster (and 3 for wrnn_duo_kernel, the shallowest depth `auto` picks it
    at) x {1, 4} clusters x {MOL, RAW}, value-checked against the C oracle at short T.  Segment counts are chosen so that some clusters run `depth` slots and the others `depth - 1` (both parities of
    the number of active slots in one launch: the `last_i` / alternate-sampling branches of wrnn_loop.hip, incl. last slots
    sampled by role B), the last group is ragged (5 segments), three segments end inside the run, and three conditioning
    slabs are crossed (state saved / restored with role-B-sampled x_t in flight)."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(0, mode=mode)
    eng = LoopEngine(sd, mode, device=gpu)
    T = 264 if mode == 'MOL' else 72
    if algo == 'chain' and clusters == 1:
        pytest.skip('wrnn_chain_kernel always spreads its groups over the 4 clusters')
    for depth in ((2, 3, 4) if algo == 'chain' else ((3,) if algo == 'duo' else ()) + (4, 5, 6, 7, 8)):
        # clusters == 4: groups = 4 (depth - 1) + 2 -> clusters 0, 1 run `depth` slots, clusters 2, 3 `depth - 1`
        groups = depth if clusters == 1 else 4 * (depth - 1) + 2
        n = 16 * (groups - 1) + 5
        mels_up, aux, seg_pos, seg_lim, noise, ref = _sweep_case(sd, mode, n, T, 1000 + n)
        out = eng.run_segments(mels_up.to(gpu), aux.to(gpu), seg_pos, seg_lim, T, noise.to(gpu), HOP, algo=algo, clusters=clusters,
                               depth=depth, slab_steps=T // 3 + 1).cpu().numpy()
        info = eng.last_run_info()
        assert (info['clusters'], info['depth'], info['rounds']) == (clusters, depth, 1), info
        if mode == 'RAW':
            bad = np.argwhere(out != ref)
            assert bad.size == 0, f'depth {depth}: first divergence at (b,t)={bad[0]} of {out.shape}'
        else:
            err = float(np.abs(out - ref).max())
            assert err <= MOL_TOL, (depth, err)


@pytest.mark.parametrize('algo,depth', [('chain', 2), ('duo', 4), ('duo', 8)])
def test_raw_thousand_steps_at_the_depths_the_planner_picks(gpu, algo, depth):
    """Round-5 verdict, "What's weak" 8: the depth sweep above value-checks 9-bit RAW over 72 steps only.  Here the depths `auto` really runs
    RAW at -- two groups per cluster on wrnn_chain_kernel, 4 and 8 on wrnn_duo_kernel (8: the LDS-prefetch form of the ih workgroups) -- run
    1,000 free-running steps across three conditioning slabs, 4 clusters, the last group ragged, against the C oracle: class indices identical."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(0, mode='RAW')
    eng = LoopEngine(sd, 'RAW', device=gpu)
    T = 1000
    n = 16 * (4 * (depth - 1) + 2 - 1) + 5
    mels_up, aux, seg_pos, seg_lim, noise, ref = _sweep_case(sd, 'RAW', n, T, 3000 + n)
    out = eng.run_segments(mels_up.to(gpu), aux.to(gpu), seg_pos, seg_lim, T, noise.to(gpu), HOP, algo=algo, clusters=4, depth=depth,
                           slab_steps=T // 3 + 1).cpu().numpy()
    info = eng.last_run_info()
    assert (info['clusters'], info['depth'], info['rounds']) == (4, depth, 1), info
    bad = np.argwhere(out != ref)
    assert bad.size == 0, f'{algo} depth {depth}: first divergence at (b,t)={bad[0]} of {out.shape}'


def test_corpus_slice_matches_per_utterance_oracle(gpu):
    """The first 16 utterances of BASELINE config 4's corpus (lens from RandomState(2024), mel seeds 1000+u: 245 folded
    segments -> 4 clusters x 4 groups in flight, the split the whole corpus' per-GPU share runs at) through `generate_corpus`
    (one launch, parity noise) vs the oracle's end-to-end `generate` per utterance: a VALUE test of config 4's path."""
    from oracle import wavernn_oracle as O, c_oracle as C
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.synthetic import random_state_dict, random_mel
    sd = random_state_dict(0, mode='MOL')
    model = _model(sd, 'MOL', gpu)

    def oracle_generate(mel, seed):                 # O.generate with the C twin of its loop (the numpy loop takes minutes here)
        mels_f, aux_f, wave_len = O.conditioning(sd, mel, True, TARGET, OVERLAP)
        raw = C.loop(sd, 'MOL', mels_f, aux_f, O.draw_noise(seed, 'MOL', mels_f.shape[0], mels_f.shape[1]), nthreads=8)
        return O.finish(raw, 'MOL', 30, wave_len, True, TARGET, OVERLAP, True)
    NU = 16
    lens = np.random.RandomState(2024).randint(300, 901, 64)[:NU]
    mels = [random_mel(1000 + u, int(n)) for u, n in enumerate(lens)]
    seeds = [4000 + u for u in range(NU)]
    outs = generate_corpus(model, [torch.from_numpy(m).unsqueeze(0) for m in mels], TARGET, OVERLAP, True, seeds)
    info = model._loop_engine().last_run_info()
    print(f'config 4 slice: {info}')
    assert info['depth'] >= 4 and info['rounds'] == 1
    C.build()
    refs = _pool_map(lambda u: oracle_generate(mels[u], seeds[u]), list(range(NU)))
    for u, mel in enumerate(mels):
        ref = refs[u]
        assert outs[u].shape == ref.shape
        assert np.abs(outs[u] - ref).max() <= MOL_TOL, (u, np.abs(outs[u] - ref).max())


def test_block_sparse_kernel_full_length_matches_oracle(gpu):
    """BASELINE config 5 at T = 12,100: two 641-frame utterances (32 segments = 2 clusters) on 95 %-block-pruned GRU weights through
    `wrnn_sparse_kernel` (`auto` for such a pack; oracle-side conditioning, materialised mel) vs the C oracle on the same masked dense
    weights -- and the dense wrnn_duo_kernel on the same masked weights, same bar."""
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.prune import block_prune_state_dict
    from wavernn_amd.synthetic import random_state_dict
    sd, _ = block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1))
    plan, mels_up, aux, flat, refs = _corpus_inputs(sd, 'MOL', [641, 641], [1234, 1235], [77, 78])
    eng = LoopEngine(sd, 'MOL', device=gpu)
    for algo, kernel in (('auto', 'wrnn_sparse_kernel'), ('duo', 'wrnn_duo_kernel')):
        out = eng.run_segments(torch.from_numpy(mels_up).to(gpu), torch.from_numpy(aux).to(gpu), plan.seg_pos, plan.seg_lim, plan.T,
                               torch.from_numpy(flat).to(gpu), HOP, algo=algo).cpu().numpy()
        assert eng.last_loop_kernel() == kernel
        for u, ref in enumerate(refs):
            got = out[plan.first[u]:plan.first[u] + plan.folds[u]]
            assert np.abs(got - ref).max() <= MOL_TOL, (algo, u, np.abs(got - ref).max())


@pytest.mark.parametrize('mel_in_loop', [None, True], ids=['mel-default', 'mel-in-loop'])
@pytest.mark.parametrize('linear', [True, False], ids=['gru+linear', 'gru'])
def test_config5_256_segments_matches_oracle(gpu, linear, mel_in_loop):
    """(`linear`: fc1 / fc2 pruned with the GRUs, the notebook's recipe = `config.config5` since round 6 -- the gathered fc stages; else `config5_gru_only`.)
    BASELINE config 5 AS BENCHMARKED (`bench.py` `config.config5`): the GRU matrices 95 % block-sparse (16x1 blocks), 16 x 641-frame
    utterances = 256 segments x 12,100 steps through `generate_corpus` -- HIP pre-loop kernels; `mel-default`: the shipped default =
    what the bench runs (the materialised mel since the last session of round 6), `mel-in-loop`: the last up-sampling stage formed inside
    the loop (`model.mel_in_loop = True`, wrnn_options.mel_stage = 1: no [L, 80] mel is written) --, `algo = auto` -> wrnn_sparse_kernel: all
    16 clusters, one group each, one round --, parity noise, against the C oracle on the masked dense weights per utterance (round-4
    verdict: 256 segments, was 32).  The loop's workspace must not depend on T."""
    from helpers import oracle_utterance, pruned_state_dict
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.synthetic import random_mel
    sd = pruned_state_dict('MOL', 0, 0.95, linear)
    model = _model(sd, 'MOL', gpu)
    assert model.mel_in_loop is None                  # the shipped default
    model.mel_in_loop = mel_in_loop
    NU = 16
    mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0) for u in range(NU)]
    segs, plan = generate_corpus(model, mels, TARGET, OVERLAP, True, [77 + u for u in range(NU)], return_segments=True)
    eng = model._loop_engine()
    info = eng.last_run_info()
    print(f'config 5: {info} {eng.last_loop_ms():.1f} ms')
    assert plan.n_segments == 256 and plan.T == 12100
    assert info['kernel'] == 'wrnn_sparse_kernel' and (info['clusters'], info['depth'], info['rounds']) == (16, 1, 1)
    assert (eng.sparse_fc_blocks > 0) == linear
    assert model.mel_rows_ok(eng, 256, plan.T) == bool(mel_in_loop)
    ws = [eng.workspace_bytes(256, T, 16 * 641) for T in (12100, 121000)]
    assert ws[0] == ws[1] and ws[0] < 300e6, ws
    refs = _pool_map(lambda u: oracle_utterance('MOL', 0, 0.95, 1234 + u, 77 + u, 641, want_cond=False, sd=sd, linear=linear)['ref'], list(range(NU)))
    for u, ref in enumerate(refs):
        got = segs[plan.first[u]:plan.first[u] + plan.folds[u]].astype(np.float32)
        assert np.abs(got - ref).max() <= MOL_TOL, (u, np.abs(got - ref).max())


@pytest.mark.parametrize('mode,mel_in_loop', [('RAW', None), ('MOL', None), ('MOL', True)], ids=['RAW', 'MOL', 'MOL-mel-in-loop'])
def test_bench_legs_as_benchmarked_match_oracle(gpu, mode, mel_in_loop):
    """The bench's RAW and MoL legs AS BENCHMARKED (round-4 verdict, "What's weak" 1): 16 x 641-frame utterances = 256 segments x
    12,100 steps through `generate_corpus` with the shipped defaults -- HIP pre-loop kernels, the materialised mel (both modes since the
    last session of round 6; scripts/gpu_raw_flips.py, gpu_corpus_ab.py measure both forms) -- and, `MOL-mel-in-loop`, with the opt-in form
    in which the loop kernel forms the last up-sampling stage itself (`model.mel_in_loop = True`, wrnn_options.mel_stage = 1; what rounds
    4-6 benchmarked) --, `algo = auto` (wrnn_duo_kernel, 4 clusters x 4 groups in flight), parity noise
    (per-utterance MT19937 streams, seeds 77 + u) drawn and uploaded in step slices, against the C oracle per utterance.
    RAW: the class indices are BIT-IDENTICAL (3.1 M segment-steps, free-running); MoL: <= MOL_TOL."""
    from helpers import oracle_utterance
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.synthetic import random_state_dict, random_mel
    sd = random_state_dict(0, mode=mode)
    model = _model(sd, mode, gpu)
    assert model.pre_algo == 'native' and model.mel_in_loop is None and model.loop_algo == 'auto'       # the shipped defaults
    model.mel_in_loop = mel_in_loop
    in_loop = bool(mel_in_loop)                                   # default: the materialised mel (wavernn_amd/model.py `mel_in_loop`)
    NU = 16
    mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0) for u in range(NU)]
    segs, plan = generate_corpus(model, mels, TARGET, OVERLAP, True, [77 + u for u in range(NU)], return_segments=True)
    eng = model._loop_engine()
    info = eng.last_run_info()
    print(f'bench leg [{mode}]: {info}, mel rows in loop: {model.mel_rows_ok(eng, 256, plan.T)}')
    assert plan.n_segments == 256 and plan.T == 12100
    assert info['kernel'] == 'wrnn_duo_kernel' and (info['clusters'], info['depth'], info['rounds']) == (4, 4, 1)
    assert model.mel_rows_ok(eng, 256, plan.T) == in_loop         # in-loop: the call above ran with the mel one up-sampling stage short
    if mode == 'RAW':
        assert info['launches'] > 8                               # the noise went up in several step slices (continued launches)
    refs = _pool_map(lambda u: oracle_utterance(mode, 0, 0.0, 1234 + u, 77 + u, 641, want_cond=False, sd=sd)['ref'], list(range(NU)))
    for u, ref in enumerate(refs):
        got = segs[plan.first[u]:plan.first[u] + plan.folds[u]].astype(np.float32)
        if mode == 'RAW':
            bad = np.argwhere(got != ref)
            assert bad.size == 0, f'utterance {u}: {len(bad)} samples differ, first at (segment, step) = {bad[0]}'
        else:
            assert np.abs(got - ref).max() <= MOL_TOL, (u, np.abs(got - ref).max())


FLIP_UTTERANCES = {'raw_flip_u34': 34, 'raw_flip_u46': 46}


@pytest.mark.parametrize('name', list(FLIP_UTTERANCES))
def test_raw_near_ties_one_utterance_matches_the_reference(gpu, name, tmp_path):
    """Round 6 (fatchord_version.py:231-237): the shipped RAW path -- `WaveRNN.generate()`, materialised mel, `auto` = wrnn_chain_kernel for the 16
    segments of one utterance -- reproduces the REFERENCE's class indices at the two utterances where the 12.4 M segment-step measurement saw a
    kernel and the C oracle part ways (tests/golden/raw_flip_u*.npz = the reference's own run; the oracle differs from it at u34 (2, 7399):
    tests/test_oracle_golden.py)."""
    from wavernn_amd.synthetic import random_state_dict
    from wavernn_amd import fold
    cfg, g = load_case(name)
    sd = random_state_dict(cfg['wseed'], mode='RAW')
    model = _model(sd, 'RAW', gpu)
    assert model.mel_in_loop is None                  # the RAW default: materialised mel
    torch.manual_seed(cfg['seed'])
    out = model.generate(torch.tensor(case_mel(cfg, g)).unsqueeze(0), tmp_path / 'o.wav', True, cfg['target'], cfg['overlap'], True)
    # the reference's class indices through the reference's post-loop arithmetic (decode_mu_law, xfade_and_unfold, tail fade: bit-exact
    # restatements, tests/test_oracle_golden.py) -- a sample of `out` differs exactly where a class index of its segment does
    ref_raw = g['cls'].astype(np.float32) * np.float32(2) / np.float32(511) - np.float32(1)
    want = fold.xfade_and_unfold(fold.decode_mu_law(ref_raw.astype(np.float64), 512, False), cfg['target'], cfg['overlap'])
    want = fold.finish_waveform(want, (cfg['frames'] - 1) * HOP, HOP)
    print(f'{name}: {model.last_loop_kernel} {model.last_loop_ms:.1f} ms')
    assert model.last_loop_kernel == 'wrnn_chain_kernel'
    assert out.shape == want.shape and np.array_equal(out, want), f'{np.count_nonzero(out != want)} of {out.size} samples differ'


def test_raw_near_ties_in_the_benchmarked_batch_match_the_reference(gpu):
    """The same two utterances inside the batch they were found in -- utterances 32 .. 47 of the flip-rate corpus = 256 segments x 12,100 steps
    through `generate_corpus` (parity noise per utterance, materialised mel: the RAW default), `auto` = wrnn_duo_kernel at depth 4, the split
    the RAW bench leg runs: the class indices of utterances 34 and 46 are IDENTICAL to the reference's own run."""
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.synthetic import random_state_dict, random_mel
    sd = random_state_dict(0, mode='RAW')
    model = _model(sd, 'RAW', gpu).eval()
    us = list(range(32, 48))
    mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0) for u in us]
    segs, plan = generate_corpus(model, mels, TARGET, OVERLAP, True, [77 + u for u in us], return_segments=True)
    info = model._loop_engine().last_run_info()
    print(f'flip batch: {info["kernel"]} depth {info["depth"]} launches {info["launches"]}')
    assert info['kernel'] == 'wrnn_duo_kernel' and info['depth'] == 4
    for name, u in FLIP_UTTERANCES.items():
        cfg, g = load_case(name)
        i = us.index(u)
        got = segs[plan.first[i]:plan.first[i] + plan.folds[i]].astype(np.float32)
        cls = np.rint((got.astype(np.float64) + 1.0) * 511.0 / 2.0).astype(np.int64)
        ref = g['cls'].astype(np.int64)
        assert cls.shape == ref.shape and np.array_equal(cls, ref), (name, np.argwhere(cls != ref)[:4].tolist())
