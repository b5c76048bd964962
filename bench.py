#!/usr/bin/env python
"""bench.py -- throughput of the MI355X-native WaveRNN generate path (BASELINE.json metric).

A "step" is ONE full pass of the hot path over one utterance of synthetic input: BASELINE config 2 --
MoL WaveRNN (ljspeech.wavernn.mol hparams, random-init weights), a 481-frame (~6 s) random mel spectrogram,
batched generation with target=11000 overlap=550 -> B=12 folded segments x T=12100 autoregressive steps.
Timed region (inputs already resident in HBM): up-sample network -> hoisted conditioning -> the persistent
loop kernel -> [N>1: RCCL all-gather of the [B,T] audio] -> D2H -> cross-fade/unfold on the host.  The WAV
write is excluded (the CPU baseline excludes it too).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

value = useful audio samples per second over all ranks = N * K * wave_len / max-over-ranks(time).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 22050
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md); measured-achievable 6290 GB/s


def cpu_baseline(sd, mode, frames, target, overlap, steps_sample):
    """The oracle's C restatement ("port") timed on this box's host cores on a bounded sample of the same
    workload: the same B segments, the first `steps_sample` of the T steps."""
    from oracle import wavernn_oracle as O, c_oracle as C
    from wavernn_amd.synthetic import random_mel
    mel = random_mel(1234, frames)
    mels, aux, wave_len = O.conditioning(sd, mel, True, target, overlap)
    B, T, _ = mels.shape
    ts = min(steps_sample, T)
    noise = O.draw_noise(77, mode, B, ts)
    # pick the OpenMP width that is fastest on this host (the layer-by-layer barriers make very wide teams slower)
    ncpu = os.cpu_count() or 1
    best = None
    for nt in (8, 16, 32, 64):
        if nt > ncpu and best is not None:
            break
        nt = min(nt, ncpu)
        C.loop(sd, mode, mels[:, :8], aux[:, :8], (noise[0][:8], noise[1][:8]), nthreads=nt)    # warm-up
        t0 = time.perf_counter()
        C.loop(sd, mode, mels[:, :40], aux[:, :40], (noise[0][:40], noise[1][:40]), nthreads=nt)
        d = time.perf_counter() - t0
        if best is None or d < best[1]:
            best = (nt, d)
    cores = best[0]
    ts = max(40, min(ts, int(15.0 / (best[1] / 40))))        # bound the sample to ~15 s of CPU work
    noise = (noise[0][:ts], noise[1][:ts])
    t0 = time.perf_counter()
    C.loop(sd, mode, mels[:, :ts], aux[:, :ts], noise, nthreads=cores)
    dt = time.perf_counter() - t0
    seg_steps_per_s = B * ts / dt
    useful = seg_steps_per_s * wave_len / (B * T)          # same useful/raw ratio as the full workload
    return dict(value=round(useful, 1), unit='audio samples/s', cores=cores, kind='port',
                sample=f'oracle/wrnn_oracle.c (OpenMP, {cores} of {ncpu} host threads; fastest of 8/16/32/64), B={B} segments x first {ts} of {T} steps '
                       f'({dt:.1f} s); {seg_steps_per_s:.0f} segment-steps/s',
                realtime_factor=round(useful / SAMPLE_RATE, 4))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--frames', type=int, default=481)
    ap.add_argument('--algo', default='auto', choices=['auto', 'persist', 'stream'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-steps', type=int, default=600)
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, f'WORLD_SIZE={world} but --gpus {args.gpus}'
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device: the product path has no CPU fallback')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    from wavernn_amd import fold as F
    from wavernn_amd.rng import draw_noise

    mode, target, overlap, hop = 'MOL', 11000, 550, 275
    sd = random_state_dict(0, mode=mode)
    model = WaveRNN(**SHIPPED, mode=mode)
    model.num_params = lambda *a, **k: 0
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    model.loop_algo = args.algo
    # every rank synthesises its own utterance (weak scaling: fixed work per GPU)
    mel = torch.from_numpy(random_mel(1234 + rank, args.frames)).unsqueeze(0).to(dev)
    eng = model._loop_engine()

    def one_pass():
        with torch.no_grad():
            mels_up, aux, wave_len = model.conditioning(mel)
            L = mels_up.size(0)
            B, _ = F.fold_geometry(L, target, overlap)
            T, stride = target + 2 * overlap, target + overlap
            noise = draw_noise(mode, B, T, 30, 512, 32, dev, 'cpu')     # parity-mode noise (host MT19937 stream)
            out = eng.run(mels_up, aux, B, T, stride, noise, hop, algo=args.algo, check=False)
            if world > 1:
                gathered = [torch.empty_like(out) for _ in range(world)]
                dist.all_gather(gathered, out)                          # RCCL all-gather of the finished audio
                out = gathered[rank]
            y = out.cpu().numpy().astype(np.float64)
            from wavernn_amd import _lib
            _lib.check(eng.lib.wrnn_status(eng._ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'loop kernel')
        y = F.xfade_and_unfold(y, target, overlap)
        y = F.finish_waveform(y, wave_len, hop)
        return y, B, T, wave_len

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    loop_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y, B, T, wave_len = one_pass()
        loop_ms.append(eng.last_loop_ms())
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        value = world * args.steps * wave_len / dt
        W = eng.weight_bytes
        bytes_per_launch = (W + B * 836) * T            # (W + B*836 B) per batch step x T steps
        kms = float(np.mean(loop_ms))
        achieved = bytes_per_launch / (kms * 1e-3) / 1e9
        res = {
            'metric': 'audio samples/sec (real-time factor @22.05 kHz), MoL WaveRNN batched generate',
            'value': round(value, 1), 'unit': 'audio samples/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'realtime_factor': round(value / SAMPLE_RATE, 2),
            'config': {'workload': f'BASELINE config 2: MoL WaveRNN (rnn/fc 512, random-init weights), one {args.frames}-frame '
                                   f'random mel per GPU, batched fold target={target} overlap={overlap} -> B={B} segments x '
                                   f'T={T} steps, wave_len={wave_len}', 'kernel': eng.last_loop_kernel(),
                       'segment_steps_per_s': round(world * B * T * args.steps / dt, 1), 'noise': 'host MT19937 stream (parity mode)',
                       'parallelism': f'{world} x (1 process per GPU, independent utterances, RCCL all-gather of audio)'},
            'roofline': {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': None,
                         'kernel': eng.last_loop_kernel(), 'kernel_ms': round(kms, 3),
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'note': 'weight-streaming-equivalent bandwidth (W + B*836 B per batch step); weights are '
                                 'on-chip resident so true HBM traffic is far lower (DESIGN.md)'},
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                res['cpu_baseline'] = cpu_baseline(sd, mode, args.frames, target, overlap, args.cpu_steps)
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                res['cpu_baseline'] = {'error': repr(e)}
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
