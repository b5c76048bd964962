"""Batch vocoding of a directory of mel spectrograms on one or several GPUs (BASELINE config 4).

    python -m wavernn_amd.gen_corpus --mels mels_dir --weights latest_weights.pyt --output wavs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m wavernn_amd.gen_corpus ...

Every `*.npy` under --mels is a (80, N) mel in [0, 1] (the reference's `gen_wavernn.py --file x.npy` input,
gen_wavernn.py:38-65).  All utterances are folded into one segment table (`batch.plan_utterances`), the table is cut into
one contiguous block per rank, each rank runs ONE loop launch, the finished audio is all-gathered (RCCL) and every rank
writes the WAVs of the utterances whose first segment it owned.  `--seed S` gives utterance u the parity noise stream of
`torch.manual_seed(S + u)` (equal to a per-utterance `generate()` call with that seed); without it noise is drawn on the
device.
"""
import argparse
import os
from pathlib import Path

import numpy as np
import torch

from .batch import generate_corpus
from .dsp import save_wav
from .model import WaveRNN
from .synthetic import SHIPPED


def load_mels(mel_dir):
    paths = sorted(Path(mel_dir).expanduser().glob('*.npy'))
    if not paths:
        raise ValueError(f'no .npy files under {mel_dir}')
    mels = []
    for p in paths:
        m = np.load(p)
        if m.ndim != 2 or m.shape[0] != 80:
            raise ValueError(f'{p}: expected a numpy array shaped (n_mels, n_hops), but got {m.shape}!')
        if m.max() >= 1.01 or m.min() <= -0.01:
            raise ValueError(f'{p}: expected spectrogram range in [0,1] but was instead [{m.min()}, {m.max()}]')
        if m.shape[1] < 21:
            raise ValueError(f'{p}: {m.shape[1]} frames; the reference needs wave_len >= 20*hop (21 frames)')
        mels.append(torch.from_numpy(m.astype(np.float32)).unsqueeze(0))
    return paths, mels


def main(argv=None):
    ap = argparse.ArgumentParser(description='Vocode a directory of mels with WaveRNN on MI355X (one launch per GPU)')
    ap.add_argument('--mels', required=True, help='directory of (80, N) .npy mel spectrograms')
    ap.add_argument('--weights', '-w', help='state-dict .pyt of the reference WaveRNN (random init if omitted)')
    ap.add_argument('--output', default='model_outputs')
    ap.add_argument('--target', '-t', type=int, default=11000)
    ap.add_argument('--overlap', '-o', type=int, default=550)
    ap.add_argument('--mode', default='MOL', choices=['MOL', 'RAW'])
    ap.add_argument('--seed', type=int, default=None, help='parity noise: utterance u uses torch.manual_seed(seed + u)')
    a = ap.parse_args(argv)
    if not torch.cuda.is_available():
        raise SystemExit('wavernn_amd needs a HIP device; there is no CPU path (use the reference for that)')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)      # nccl == RCCL on ROCm
        group = dist.group.WORLD
    model = WaveRNN(**SHIPPED, mode=a.mode).to(dev)
    if a.weights:
        model.load(a.weights)
    paths, mels = load_mels(a.mels)
    seeds = [a.seed + u for u in range(len(mels))] if a.seed is not None else None
    outs = generate_corpus(model, [m.to(dev) for m in mels], a.target, a.overlap, True, seeds, group=group,
                           noise_source='cpu' if seeds is not None else 'device', finish='own')
    out_dir = Path(a.output)
    out_dir.mkdir(parents=True, exist_ok=True)
    n = 0
    for p, y in zip(paths, outs):
        if y is not None:
            save_wav(y, out_dir / f'__{p.stem}__gen_batched_target{a.target}_overlap{a.overlap}.wav', model.sample_rate)
            n += y.shape[0]
    eng = model._loop_engine()
    print(f'rank {rank}/{world}: wrote {sum(y is not None for y in outs)} of {len(outs)} utterances, {n / model.sample_rate:.1f} s of audio; '
          f'loop {eng.last_loop_kernel()} {eng.last_loop_ms():.0f} ms')
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
