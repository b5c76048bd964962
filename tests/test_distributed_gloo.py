"""world_size-2 (and 3) CPU tests of the multi-GPU path: `generate_corpus` shards the folded segments of several
utterances over the ranks, each rank runs its block, ONE all_gather returns the finished audio.  gloo stands in for
RCCL and the oracle's C loop stands in for the HIP loop (injected through `loop_fn`; the product default is the HIP
engine, which refuses to run without a device).  What is under test is the host logic: segment table, per-rank
rebasing of positions, noise addressing by (utterance seed, step, fold), gather order and the per-utterance unfold."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import MOL_TOL, ROOT, oracle_loop_fn

FRAMES = [23, 40, 31, 26]
SEEDS = [900, 901, 902, 903]
TARGET, OVERLAP = 550, 55


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(mode):
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    sd = random_state_dict(51, mode=mode)
    model = WaveRNN(**SHIPPED, mode=mode)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    mels = [torch.from_numpy(random_mel(600 + u, n)).unsqueeze(0) for u, n in enumerate(FRAMES)]
    return sd, model, mels


def _worker(rank, world, port, mode, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from wavernn_amd.batch import generate_corpus
        sd, model, mels = _build(mode)
        outs = generate_corpus(model, mels, TARGET, OVERLAP, True, SEEDS, group=dist.group.WORLD,
                               loop_fn=oracle_loop_fn(sd, mode))
        np.savez(os.path.join(outdir, f'rank{rank}.npz'), *outs)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('mode,world', [('MOL', 2), ('RAW', 2), ('MOL', 3)])
def test_sharded_corpus_equals_single_process(tmp_path, mode, world):
    from wavernn_amd.batch import generate_corpus
    from oracle import wavernn_oracle as O
    from wavernn_amd.synthetic import random_mel
    sd, model, mels = _build(mode)
    single = generate_corpus(model, mels, TARGET, OVERLAP, True, SEEDS, loop_fn=oracle_loop_fn(sd, mode))
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        z = np.load(tmp_path / f'rank{r}.npz')
        got = [z[f'arr_{u}'] for u in range(len(FRAMES))]
        for u in range(len(FRAMES)):
            assert got[u].dtype == np.float64 and got[u].shape == ((FRAMES[u] - 1) * 275,)
            assert np.array_equal(got[u], single[u]), f'rank {r} utterance {u}: sharded != single-process'
    # and the single-process corpus result equals per-utterance generation (the oracle's end-to-end restatement,
    # itself pinned to the reference's golden waveforms): RAW bit-exact, MoL within tolerance
    for u, n in enumerate(FRAMES):
        ref = O.generate(sd, mode, random_mel(600 + u, n), True, TARGET, OVERLAP, True, SEEDS[u])
        if mode == 'RAW':
            assert np.array_equal(single[u], ref), np.abs(single[u] - ref).max()
        else:
            assert np.abs(single[u] - ref).max() <= MOL_TOL


def test_emulated_shard_equals_the_rank_it_stands_for():
    """`generate_corpus(..., shard=(r, w))` (no process group: how bench.py shows one GPU's share of BASELINE config 4 at N = 1) does
    what rank r of a w-rank job does on its own: the utterances lying entirely in its block of the segment table come out equal to the
    single-process run, the others are left to the ranks that own their first segment / need the gather."""
    from wavernn_amd.batch import generate_corpus, plan_utterances, shard_bounds
    sd, model, mels = _build('MOL')
    single = generate_corpus(model, mels, TARGET, OVERLAP, True, SEEDS, loop_fn=oracle_loop_fn(sd, 'MOL'))
    plan = plan_utterances([n * 275 for n in FRAMES], TARGET, OVERLAP)
    seen = 0
    for r in range(2):
        lo, hi = shard_bounds(plan.n_segments, 2)[r]
        outs = generate_corpus(model, mels, TARGET, OVERLAP, True, SEEDS, loop_fn=oracle_loop_fn(sd, 'MOL'), shard=(r, 2))
        for u in range(len(FRAMES)):
            inside = lo <= plan.first[u] and plan.first[u] + plan.folds[u] <= hi
            assert (outs[u] is not None) == bool(inside)
            if inside:
                assert np.array_equal(outs[u], single[u])
                seen += 1
    assert seen >= 2
