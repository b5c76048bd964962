"""CPU tests of the host side: C-ABI exports, fold / cross-fade helpers against the reference's golden waveforms,
the corpus segment table, noise packing and the private-generator noise stream."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import CASES, ROOT, load_case


def test_c_abi_library_exports_every_declared_symbol():
    """The shared object loads without a GPU and exports every function include/wavernn_amd.h declares."""
    from wavernn_amd import _lib
    if not os.path.exists(_lib.SO_PATH):
        _lib.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, 'include', 'wavernn_amd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(wrnn_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.wrnn_abi_version() == 9
    # no GPU here: compute entry points must fail loudly, never fall back to the host
    if not torch.cuda.is_available():
        assert L.wrnn_device_cus(0) < 0
        assert b'no HIP device' in L.wrnn_last_error()


def test_engine_refuses_to_run_without_a_device():
    from wavernn_amd import _lib
    from wavernn_amd.engine import LoopEngine
    from wavernn_amd.synthetic import random_state_dict
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(_lib.WrnnError):
        LoopEngine(random_state_dict(1), 'MOL')


@pytest.mark.parametrize('name', CASES)
def test_post_loop_host_path_matches_reference(name):
    """decode_mu_law -> xfade_and_unfold -> tail fade on the reference's own [B,T] tensor == its returned waveform."""
    from wavernn_amd import fold as F
    cfg, g = load_case(name)
    y = g['raw'].astype(np.float64)
    n_classes = 512 if cfg['mode'] == 'RAW' else 30
    if cfg['mode'] == 'RAW' and cfg['mu_law']:
        y = F.decode_mu_law(y, n_classes, False)
    y = F.xfade_and_unfold(y, cfg['target'], cfg['overlap']) if cfg['batched'] else y[0]
    wave_len = (cfg['frames'] - 1) * 275
    out = F.finish_waveform(y, wave_len, 275)
    assert out.dtype == np.float64 and np.array_equal(out, g['out'])


def test_fold_helpers_match_the_oracle():
    from oracle import wavernn_oracle as O
    from wavernn_amd import fold as F
    rs = np.random.RandomState(0)
    for L, tg, ov in [(22275, 11000, 550), (132275, 11000, 550), (700, 1100, 55), (14575, 2000, 100), (1155 * 3 + 55, 1100, 55)]:
        x = rs.randn(1, L, 3).astype(np.float32)
        ref = O.fold_with_overlap(x, tg, ov)
        got = F.fold_with_overlap(torch.from_numpy(x), tg, ov).numpy()
        assert got.shape == ref.shape and np.array_equal(got, ref)
        assert F.fold_geometry(L, tg, ov)[0] == O.num_folds(L, tg, ov) == ref.shape[0]
    with pytest.raises(ValueError):                       # reference quirk: wave_len < 20*hop
        F.finish_waveform(np.zeros(5000), 5000, 275)


def test_segment_table_and_shards():
    from wavernn_amd.batch import plan_utterances, shard_bounds
    from wavernn_amd import fold as F
    lens = [23 * 275, 40 * 275, 31 * 275, 2 * 275]
    plan = plan_utterances(lens, 550, 55)
    assert plan.T == 660 and plan.stride == 605
    assert list(plan.folds) == [F.fold_geometry(L, 550, 55)[0] for L in lens]
    assert plan.n_segments == plan.folds.sum() and plan.offsets[1] == lens[0]
    for u in range(len(lens)):
        s = slice(plan.first[u], plan.first[u] + plan.folds[u])
        assert np.array_equal(plan.seg_pos[s], plan.offsets[u] + np.arange(plan.folds[u]) * 605)
        assert (plan.seg_lim[s] == plan.offsets[u] + lens[u]).all() and (plan.seg_utt[s] == u).all()
        assert plan.offsets[u] % 275 == 0                 # every utterance starts on a frame boundary
    for world in (1, 2, 3, 8, 64):
        b = shard_bounds(plan.n_segments, world)
        assert b[0][0] == 0 and b[-1][1] == plan.n_segments
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [h - l for l, h in b]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.parametrize('mode', ['MOL', 'RAW'])
def test_pack_noise_layout(mode):
    from wavernn_amd.batch import plan_utterances, pack_noise
    plan = plan_utterances([23 * 275, 40 * 275, 31 * 275], 550, 55)
    T, C = plan.T, 8
    rs = np.random.RandomState(1)
    per = [rs.rand(T, 11 * int(b)).astype(np.float32) if mode == 'MOL' else rs.rand(T, int(b), C).astype(np.float32) for b in plan.folds]
    full = pack_noise(mode, plan, per)
    n = plan.n_segments
    for s in range(n):
        u = int(plan.seg_utt[s]); i = s - int(plan.first[u]); Bu = int(plan.folds[u])
        if mode == 'MOL':
            assert np.array_equal(full[:, 10 * s:10 * s + 10], per[u][:, 10 * i:10 * i + 10])
            assert np.array_equal(full[:, 10 * n + s], per[u][:, 10 * Bu + i])
        else:
            assert np.array_equal(full[:, s], per[u][:, i])
    # a block that starts and ends inside utterances, torch path
    lo, hi = 2, n - 1
    part = pack_noise(mode, plan, [torch.from_numpy(p) for p in per], lo, hi).numpy()
    if mode == 'MOL':
        assert np.array_equal(part[:, :10 * (hi - lo)], full[:, 10 * lo:10 * hi])
        assert np.array_equal(part[:, 10 * (hi - lo):], full[:, 10 * n + lo:10 * n + hi])
    else:
        assert np.array_equal(part, full[:, lo:hi])


@pytest.mark.parametrize('mode', ['MOL', 'RAW'])
def test_private_generator_equals_global_stream(mode):
    """draw_noise(generator=g) == the reference's global-generator consumption (GRUCell ctors + per-step draws),
    which tests/test_oracle_golden.py pins to the reference."""
    from oracle import wavernn_oracle as O
    from wavernn_amd.rng import draw_noise, gru_cell_ctor_draws
    assert gru_cell_ctor_draws(512, 32) == O.gru_cell_ctor_draws() == 3_201_024
    B, T = 3, 5
    torch.manual_seed(77)
    a = draw_noise(mode, B, T, 512, 512, 32, 'cpu')
    b = draw_noise(mode, B, T, 512, 512, 32, 'cpu', generator=torch.Generator().manual_seed(77))
    assert torch.equal(a, b)
    ref = O.draw_noise(77, mode, B, T)
    if mode == 'MOL':
        ref = np.concatenate([ref[0].reshape(T, B * 10), ref[1].reshape(T, B)], axis=1)
    assert np.array_equal(a.numpy().reshape(ref.shape), ref)


def test_block_prune_mask():
    """Config 5 recipe: per gate, the 95 % lowest-magnitude 16x1 blocks are zeroed; survivors keep their values."""
    from wavernn_amd.prune import block_mask, block_prune_state_dict
    from wavernn_amd.synthetic import random_state_dict
    sd = random_state_dict(3, mode='MOL')
    out, density = block_prune_state_dict(sd, 0.95, (16, 1))
    for k in ('rnn1.weight_ih_l0', 'rnn1.weight_hh_l0', 'rnn2.weight_ih_l0', 'rnn2.weight_hh_l0'):
        W, P = sd[k], out[k]
        assert P.shape == W.shape and abs(density[k] - 0.05) < 1.2e-3
        nz = P != 0
        assert np.array_equal(P[nz], W[nz])
        for g in range(3):                                   # every gate is pruned on its own
            blk = nz[g * 512:(g + 1) * 512].reshape(32, 16, -1)
            assert np.all(blk.all(axis=1) | ~blk.any(axis=1))          # blocks are all-or-nothing
            assert abs(blk.any(axis=1).mean() - 0.05) < 2e-3
    assert np.array_equal(out['fc1.weight'], sd['fc1.weight'])
    # the notebook's element-wise rule is the 1x1 special case
    W = sd['rnn1.weight_hh_l0']
    M = block_mask(W, 0.9, (1, 1))
    for g in range(3):
        a = np.abs(W[g * 512:(g + 1) * 512])
        thr = np.sort(a.reshape(-1))[int(a.size * 0.9)]
        assert np.array_equal(M[g * 512:(g + 1) * 512], (a >= thr).astype(np.float32))


def test_header_is_plain_c(tmp_path):
    """include/wavernn_amd.h is the drop-in boundary: it must compile as plain C99 (no C++ in the header, no torch types)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('gcc not available')
    src = tmp_path / 'use_header.c'
    src.write_text('#include "wavernn_amd.h"\n'
                   'int probe(void) { wrnn_weights w; wrnn_geometry g; wrnn_options o; wrnn_run_info ri; wrnn_pre_weights p;\n'
                   '  (void)w; (void)g; (void)o; (void)ri; (void)p;\n'
                   '  return WRNN_ABI_VERSION + WRNN_OK + WRNN_MODE_MOL + WRNN_ALGO_SPARSE + (int)sizeof(wrnn_pre_weights); }\n')
    res = subprocess.run(['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-fsyntax-only', '-I', os.path.join(ROOT, 'include'), str(src)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert res.returncode == 0, res.stdout


def test_gen_corpus_input_checks(tmp_path):
    """the corpus CLI applies the reference's `.npy` mel checks (gen_wavernn.py:48-55) and the 21-frame minimum (:258)."""
    from wavernn_amd import gen_corpus as G
    with pytest.raises(ValueError):
        G.load_mels(tmp_path)                                   # empty directory
    np.save(tmp_path / 'ok.npy', np.random.RandomState(0).rand(80, 30).astype(np.float32))
    paths, mels = G.load_mels(tmp_path)
    assert [p.name for p in paths] == ['ok.npy'] and tuple(mels[0].shape) == (1, 80, 30)
    for name, arr in (('shape.npy', np.zeros((30, 80), np.float32)), ('range.npy', np.full((80, 30), 2.0, np.float32)),
                      ('short.npy', np.zeros((80, 20), np.float32))):
        np.save(tmp_path / name, arr)
        with pytest.raises(ValueError):
            G.load_mels(tmp_path)
        os.remove(tmp_path / name)


def test_pruner_follows_the_reference_schedule_and_rule():
    """`prune.Pruner` / `PruneMask` = the reference notebook's workflow ("Pruning - Scratchpad" :40-186): cubic schedule,
    re-mask every `prune_every` steps, per-gate magnitude rule with `>=` ties; plus the 16x1 block form config 5 uses."""
    import torch.nn as nn
    from wavernn_amd.prune import Pruner, PruneMask, block_mask, wavernn_pruner
    torch.manual_seed(3)
    rnn, fc = nn.GRU(32, 48), nn.Linear(48, 48)
    layers = [rnn, fc]
    pr = Pruner(layers, start_prune=10, prune_steps=200, target_sparsity=0.9375, prune_every=5)
    assert pr.total_params == 3 * 48 * 32 + 3 * 48 * 48 + 48 * 48
    t = torch.zeros(1)
    zs = []
    for step in range(260):
        pr.prune(layers, t)
        zs.append(pr.z)
        t += 1
    expect = [max(0.0, min(0.9375, 0.9375 * (1 - (1 - (s - 10) / 200) ** 3))) for s in range(260)]
    assert np.allclose(zs, expect) and zs[9] == 0 and zs[-1] == 0.9375
    # final masks: per gate exactly k = int(n * Z) entries zeroed (no ties in random weights), weights really zero
    for W, gates in ((rnn.weight_ih_l0, 3), (rnn.weight_hh_l0, 3), (fc.weight, 1)):
        for Wg in torch.split(W.data, W.size(0) // gates):
            assert int((Wg == 0).sum()) == int(Wg.numel() * 0.9375)
    assert pr.num_pruned == sum(int(W.numel() / g * 0.9375) * g for W, g in ((rnn.weight_ih_l0, 3), (rnn.weight_hh_l0, 3), (fc.weight, 1)))
    # the notebook's rule, restated directly, on one gate matrix
    W = torch.randn(3 * 16, 24)
    m = PruneMask(nn.GRU(24, 16), True)
    M = m.mask_from_matrix(W, 0.5)
    for g in range(3):
        Wg = W[16 * g:16 * (g + 1)].abs()
        thr = torch.sort(Wg.reshape(-1))[0][int(Wg.numel() * 0.5)]
        assert torch.equal(M[16 * g:16 * (g + 1)], (Wg >= thr).float())
    # block form == the numpy recipe config 5's fixtures use
    Wb = torch.randn(3 * 64, 40)
    mb = PruneMask(nn.GRU(40, 64), True, block=(16, 1))
    assert np.array_equal(mb.mask_from_matrix(Wb, 0.95).numpy(), block_mask(Wb.numpy(), 0.95, (16, 1)))
    # restart rebuilds the masks from pruned weights; rnn input pruning can be switched off (notebook's prune_rnn_input)
    pr2 = Pruner(layers, 10, 200, 0.9375, prune_rnn_input=False, prune_every=5)
    pr2.restart(layers, torch.tensor([400.]))
    assert pr2.z == 0.9375 and len(pr2.masks[0].mask) == 1 and pr2.num_pruned > 0
    # the WaveRNN convenience wrapper + the step hook
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import SHIPPED
    model = WaveRNN(**SHIPPED, mode='MOL')
    pruner, lay = wavernn_pruner(model, start_prune=0, prune_steps=10, target_sparsity=0.95, prune_every=1)
    hook = pruner.step_hook(lay, model.get_step)
    for _ in range(12):
        model.step += 1
        hook()
    W = model.rnn1.weight_hh_l0.data.numpy()
    dens = (np.abs(W).reshape(3, 32, 16, 512).sum(axis=2) != 0).mean()
    assert abs(dens - 0.05) < 0.002, dens                      # 16x1 blocks, 5 % survive per gate


def test_corpus_chunks_and_seed_check():
    """`generate_corpus` works through a rank's block in chunks of whole utterances (bounded resident conditioning / noise) and
    refuses parity noise without seeds with a clear error (round-1 advisor findings)."""
    from wavernn_amd.batch import plan_utterances, chunk_utterances, generate_corpus
    plan = plan_utterances([n * 275 for n in (23, 40, 31, 26, 55, 21)], 550, 55)
    total = plan.n_segments
    chunks = chunk_utterances(plan, 0, total, 20)
    assert [u for c in chunks for u in c[0]] == list(range(6))                       # every utterance once, in order
    assert chunks[0][1] == 0 and chunks[-1][2] == total
    assert all(a[2] == b[1] for a, b in zip(chunks, chunks[1:]))                     # contiguous segment ranges
    assert all(c[2] - c[1] <= 20 or len(c[0]) == 1 for c in chunks)                  # bounded (one long utterance may exceed)
    lo, hi = total // 3, 2 * total // 3                                               # a middle rank's block: clipped at both ends
    mid = chunk_utterances(plan, lo, hi, 8)
    assert mid[0][1] == lo and mid[-1][2] == hi and all(a[2] == b[1] for a, b in zip(mid, mid[1:]))
    assert chunk_utterances(plan, 0, total, 10 ** 6) == [(list(range(6)), 0, total)]
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import SHIPPED
    m = WaveRNN(**SHIPPED, mode='MOL')
    with pytest.raises(ValueError, match='seeds'):
        generate_corpus(m, [torch.rand(1, 80, 30)], 550, 55, True)


def test_wav_front_end_properties(tmp_path):
    """dsp.py's `.wav` -> mel front-end (librosa restated; PARITY UNPINNED: librosa is not available) -- defining properties only:
    frame count of a centered STFT, range, a pure tone lands in the mel band that contains it, unit-area filters, WAV round trip."""
    from wavernn_amd import dsp
    sr, hop = 22050, 275
    t = np.arange(sr) / sr
    for f0 in (200.0, 1000.0, 4000.0):
        m = dsp.melspectrogram((0.5 * np.sin(2 * np.pi * f0 * t)).astype(np.float32))
        assert m.shape == (80, 1 + sr // hop) and m.dtype == np.float32 and 0.0 <= m.min() and m.max() <= 1.0
        edges = dsp._mel_to_hz(np.linspace(dsp._hz_to_mel(40), dsp._hz_to_mel(sr / 2), 82))
        band = int(m[:, 40].argmax())
        assert edges[band] <= f0 <= edges[band + 2], (f0, band, edges[band:band + 3])
    B = dsp.mel_basis()
    assert B.shape == (80, 1025) and (B >= 0).all()
    np.testing.assert_allclose((B * (sr / 2 / 1024)).sum(axis=1), 1.0, atol=0.08)          # unit area (discretised)
    assert np.allclose(dsp._mel_to_hz(dsp._hz_to_mel([40.0, 999.0, 1000.0, 8000.0])), [40.0, 999.0, 1000.0, 8000.0])
    x = (0.25 * np.sin(2 * np.pi * 440 * t)).astype(np.float32)
    dsp.save_wav(x, tmp_path / 'a.wav', sr)
    assert np.array_equal(dsp.load_wav(tmp_path / 'a.wav', sr), x)
    with pytest.raises(ValueError):
        dsp.load_wav(tmp_path / 'a.wav', 16000)


def test_bench_self_launches_two_ranks_dry_host(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it (how the driver invokes `--gpus 1`): bench.py spawns its own two
    ranks, they rendezvous on 127.0.0.1, shard the segment table, all-gather and unfold; rank 0's JSON line is the last line
    of stdout and reports the world size the process group really had.  `--dry-host` swaps the HIP loop for the test's CPU
    stand-in (gloo instead of RCCL): what is exercised is the launch path, no number is claimed (`value` is null)."""
    import json, subprocess, sys
    from helpers import ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, os.environ.get('PYTHONPATH', '')]))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0', '--utterances', '2',
           '--frames', '30', '--target', '550', '--overlap', '55', '--no-config4-leg', '--dry-host', 'helpers:oracle_loop_fn']      # (the config-4 leg: its own test below)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['dry_host'] is True and line['value'] is None and line['n_gpus'] == 2 and line['scaling'] == 'weak'
    assert line['config']['parallelism'].startswith('2 rank(s)')
    # 4 utterances of 30 frames, target 550 / overlap 55 -> 14 segments each, rank 0 owns the first half
    assert line['config']['segments_rank0'] * 2 == 4 * 14
    # the fixed corpus of BASELINE config 4 is strong scaling; one rank of a launcher-provided world (WORLD_SIZE in the env) runs as is
    env1 = dict(env, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1')
    r = subprocess.run(cmd[:2] + ['--gpus', '1', '--steps', '1', '--warmup', '0', '--corpus', 'config4', '--corpus-limit', '3', '--dry-host', 'helpers:zero_loop_fn'], env=env1, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['scaling'] == 'strong' and line['n_gpus'] == 1 and '3 utterances' in line['config']['workload']


@pytest.mark.parametrize('n', [2, 4, 8])
def test_bench_default_command_carries_the_config4_leg_dry_host(n):
    """What the driver's SCALE run issues -- `bench.py --gpus N` and nothing else about the corpus -- measures weak scaling of a batch per
    GPU AND (round-4 verdict, item 6) a `config.config4` leg: BASELINE config 4's FIXED corpus sharded over the N ranks of the same launch
    (strong scaling), with the world size the group really had, the ranks the planner used, the block sizes, the time blocked on the
    all-gather and the post-loop work done under it.  Dry run on the host (gloo, a loop stand-in, the first 5 utterances of the corpus)."""
    import json, subprocess, sys
    from helpers import ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, os.environ.get('PYTHONPATH', '')]), OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(n), '--steps', '1', '--warmup', '0', '--utterances', '1', '--frames', '30',
           '--target', '550', '--overlap', '55', '--corpus-limit', '5', '--dry-host', 'helpers:probe_loop_fn']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == n and line['scaling'] == 'weak' and line['dry_host'] is True
    leg = line['config']['config4']
    assert leg['world_seen'] == n and 1 <= leg['ranks_used'] <= n and leg['utterances'] == 5 and leg['dry_host'] is True
    per = leg['segments_per_rank']
    assert len(per) == n and sum(per) == leg['segments'] and all(x == 0 for x in per[leg['ranks_used']:])
    assert max(per[:leg['ranks_used']]) - min(per[:leg['ranks_used']]) <= 1
    assert leg['ms_per_pass'] > 0 and leg['gather_wait_ms'] >= 0 and leg['unfold_under_gather_ms'] >= 0


def test_choose_ranks_fills_the_pipeline_not_the_node():
    """`batch.choose_ranks`: the smallest rank count that reaches the best estimated wall time of a pass over a FIXED corpus (the step time
    depends on the pipeline depth a rank's block fills).  BASELINE config 4 (942 segments): all of 1, 2, 4, 8 GPUs; 15 of 16."""
    from wavernn_amd.batch import choose_ranks, shard_bounds, estimate_step_us, DEFAULT_STEP_US_BY_DEPTH as TAB
    assert [choose_ranks(942, w, TAB) for w in (1, 2, 4, 8, 16)] == [1, 2, 4, 8, 15]
    assert choose_ranks(12, 8, TAB) == 1 and choose_ranks(64, 8, TAB) == 1 and choose_ranks(65, 8, TAB) == 2     # one group per cluster is as fast as it gets
    b = shard_bounds(942, 16, 15)
    assert b[15] == (942, 942) and sum(h - l for l, h in b) == 942 and b[0][0] == 0 and b[14][1] == 942
    assert estimate_step_us(118, TAB) < estimate_step_us(135, TAB) < estimate_step_us(236, TAB)
    # a table measured elsewhere changes the plan: were depth 2 as slow as depth 3, five GPUs (189 segments each: depth 3) would do for 942 segments
    flat = dict(TAB); flat[2] = flat[3]
    assert choose_ranks(942, 8, flat) == 5


def test_step_table_is_keyed_to_the_kernel_sources(tmp_path):
    """`batch.step_table`: the record bench.py measured is used only for the kernel sources it was measured on (round-5 verdict: the planner's
    step times were constants in the code that went stale silently); anything else falls back to the built-in numbers and says so."""
    import json
    from wavernn_amd.batch import step_table, kernel_source_sha16, DEFAULT_STEP_US_BY_DEPTH
    f = tmp_path / 'step_us.json'
    tab = {str(d): 10.0 + d for d in range(1, 9)}
    json.dump({'source_sha16': kernel_source_sha16(), 'step_us_by_depth': tab}, open(f, 'w'))
    t, origin = step_table(str(f))
    assert t == {d: 10.0 + d for d in range(1, 9)} and 'measured on these kernel sources' in origin
    json.dump({'source_sha16': '0' * 16, 'step_us_by_depth': tab}, open(f, 'w'))
    t, origin = step_table(str(f))
    assert t == DEFAULT_STEP_US_BY_DEPTH and 'other kernel sources' in origin
    t, origin = step_table(str(tmp_path / 'absent.json'))
    assert t == DEFAULT_STEP_US_BY_DEPTH and 'built-in' in origin


def test_bench_eight_ranks_share_config4_dry_host():
    """BASELINE config 4 as the driver launches it at N = 8 (`bench.py --gpus 8 --corpus config4`: the fixed 64-utterance corpus,
    strong scaling), on the host under gloo with a loop stand-in that tags every segment: 942 segments go to the 8 ranks in contiguous
    blocks of 117 / 118, and the all-gathered table arrives in segment order (row-weighted checksum == the single-process one)."""
    import json, subprocess, sys
    from helpers import ROOT
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, 'tests'), ROOT, os.environ.get('PYTHONPATH', '')]), OMP_NUM_THREADS='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '1', '--warmup', '0', '--corpus', 'config4', '--dry-host', 'helpers:probe_loop_fn']
    r8 = subprocess.run(base + ['--gpus', '8'], env=env, capture_output=True, text=True, timeout=1500)
    assert r8.returncode == 0, r8.stderr[-2000:]
    l8 = json.loads(r8.stdout.strip().splitlines()[-1])
    assert l8['n_gpus'] == 8 and l8['scaling'] == 'strong' and l8['dry_host'] is True
    per = l8['config']['segments_per_rank']
    assert len(per) == 8 and sum(per) == 942 and set(per) == {117, 118} and l8['config']['gathered_rows'] == 942
    r1 = subprocess.run(base + ['--gpus', '1'], env=dict(env, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1'), capture_output=True, text=True, timeout=1500)
    assert r1.returncode == 0, r1.stderr[-2000:]
    l1 = json.loads(r1.stdout.strip().splitlines()[-1])
    assert l1['config']['segments_per_rank'] == [942]
    assert abs(l8['config']['gathered_checksum'] - l1['config']['gathered_checksum']) <= 1e-6 * abs(l1['config']['gathered_checksum'])


def test_sliced_generate_degrades_to_stream_when_the_grid_is_refused(tmp_path, monkeypatch):
    """Round-2 advisor: `auto` -> stream on WRNN_ERR_RESIDENCY only fired for unsliced runs.  A step-sliced generate() (batched
    RAW always is) whose first slice is refused must redo the WHOLE call on the stream kernel from the same point of the
    noise stream, and leave the global generator where an unsliced run leaves it.  (Engine mocked: no GPU here.)"""
    import warnings
    import wavernn_amd.model as M
    from wavernn_amd import _lib
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    calls = []

    class Eng:
        def plan(self, n, T, **kw):
            return dict(kernel='wrnn_loop_kernel')

        def run(self, mels_up, aux, B, T, stride, noise, hop, algo='auto', out=None, t_range=None, progress=None, **kw):
            calls.append((algo, t_range, float(noise.sum())))
            if algo == 'auto':
                raise _lib.ResidencyError('refused')
            assert t_range is None and noise.shape[0] == T
            return torch.zeros(B, T)

        def last_loop_ms(self):
            return 0.0

        def last_loop_kernel(self):
            return 'wrnn_stream_kernel'
    monkeypatch.setattr(M.WaveRNN, '_require_hip_device', staticmethod(lambda device: None))
    monkeypatch.setattr(M.WaveRNN, '_loop_engine', lambda self: Eng())
    monkeypatch.setattr(M, 'save_wav', lambda x, path, sr: None)
    model = M.WaveRNN(**SHIPPED, mode='MOL')
    model.pre_algo, model.post_algo = 'torch', 'numpy'
    model.noise_chunk_bytes = 11 * 3 * 4 * 100          # ~100 steps of noise at a time -> several slices
    mel = torch.from_numpy(random_mel(5, 30)).unsqueeze(0)
    torch.manual_seed(11)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        model.generate(mel, tmp_path / 'x.wav', True, 2200, 55, True)
    assert any('stream kernel' in str(x.message) for x in w)
    assert [c[0] for c in calls] == ['auto', 'stream'] and calls[0][1][0] == 0
    after = torch.empty(3).uniform_(0, 1)
    # reference order of draws: ctor burn, then T x 11 B uniforms -- as one unsliced run
    from wavernn_amd.rng import burn_ctor_draws, draw_steps
    torch.manual_seed(11)
    burn_ctor_draws(512, 32, 'cpu')
    B, T = 4, 2310
    whole = draw_steps('MOL', B, T, 30, 'cpu', 'cpu')
    assert abs(calls[1][2] - float(whole.sum())) < 1e-3
    assert torch.equal(after, torch.empty(3).uniform_(0, 1))


def test_packaged_tacotron_shapes_equal_the_golden_table():
    """bench.py's config-3 leg builds its random-init Tacotron from wavernn_amd/tacotron_shapes.json: the same (key, shape, dtype)
    table scripts/make_golden.py wrote from the reference's module into tests/golden/."""
    import json
    a = json.load(open(os.path.join(ROOT, 'wavernn_amd', 'tacotron_shapes.json')))
    b = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tacotron_shapes.json')))
    assert a == b and len(a) > 100
