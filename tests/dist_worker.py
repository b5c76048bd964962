"""Worker of tests/test_gpu_dist.py::test_generate_corpus_over_every_visible_gpu: ONE rank of an N-GPU RCCL group (RANK / WORLD_SIZE / MASTER_* from the
environment, as torch.distributed.run and bench.py's self-launch set them).  Runs a 5-utterance corpus through `generate_corpus` over the group and
saves what this rank holds afterwards to <out>/rank<r>.npz.  Not collected by pytest."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def corpus():
    from wavernn_amd.synthetic import random_mel
    frames, seeds = [23, 40, 31, 55, 28], [920, 921, 922, 923, 924]
    return [torch.from_numpy(random_mel(620 + u, n)).unsqueeze(0) for u, n in enumerate(frames)], seeds


def model_on(dev):
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, SHIPPED
    sd = random_state_dict(53, mode='MOL')
    model = WaveRNN(**SHIPPED, mode='MOL')
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    return model.to(dev)


def main(out_dir):
    import torch.distributed as dist
    from wavernn_amd.batch import generate_corpus
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', rank)))
    torch.cuda.set_device(dev)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
    try:
        mels, seeds = corpus()
        tm = {}
        outs = generate_corpus(model_on(dev), mels, 550, 55, True, seeds, group=dist.group.WORLD, timings=tm)       # finish='all': every rank ends up with every utterance
        t = torch.ones(1, device=dev)
        dist.all_reduce(t)
        np.savez(os.path.join(out_dir, f'rank{rank}.npz'), world_seen=int(tm.get('world_seen', 0)), reduced=float(t.item()),
                 gather_bytes=int(tm.get('gather_bytes', 0)), **{f'u{u}': o for u, o in enumerate(outs)})
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    main(sys.argv[1])
