"""RCCL on hardware (-m gpu): the multi-GPU code path of `generate_corpus` -- contiguous shard, `dist.all_gather` of the
finished [n_r, T] audio on HIP tensors over the `nccl` (= RCCL) backend, `device_id=`, padding / slicing, finish='own' -- run
with a ONE-rank process group, which is all a one-GPU box admits.  No scaling is claimed from this; it removes the "the
nccl branch never executed" risk before the driver's 8-GPU run (world sizes 2 and 3 are covered under gloo on the CPU,
tests/test_distributed_gloo.py)."""
import numpy as np
import pytest
import torch

from helpers import MOL_TOL

pytestmark = pytest.mark.gpu


def test_generate_corpus_through_a_one_rank_rccl_group(tmp_path):
    import torch.distributed as dist
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    assert torch.cuda.is_available()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29517', rank=0, world_size=1, device_id=dev)
    try:
        sd = random_state_dict(53, mode='MOL')
        model = WaveRNN(**SHIPPED, mode='MOL')
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
        model = model.to(dev)
        frames, seeds = [23, 40, 31], [920, 921, 922]
        mels = [torch.from_numpy(random_mel(620 + u, n)).unsqueeze(0) for u, n in enumerate(frames)]
        ref = generate_corpus(model, mels, 550, 55, True, seeds)                                   # no group
        got = generate_corpus(model, mels, 550, 55, True, seeds, group=dist.group.WORLD, finish='own')
        t = torch.ones(4, device=dev)
        dist.all_reduce(t)                                                                         # the collective bench.py's timing uses
        assert t.sum().item() == 4.0
        for a, b in zip(ref, got):
            assert b is not None and np.abs(a - b).max() <= MOL_TOL
    finally:
        dist.destroy_process_group()
