#!/usr/bin/env python
"""bench.py -- throughput of the MI355X-native WaveRNN generate path (BASELINE.json metric).

A "step" is ONE full pass of the hot path over one batch of synthetic input.  Workload (BASELINE config 2 -- MoL
WaveRNN, ljspeech.wavernn.mol hparams, random-init weights, batched generation target=11000 overlap=550 -- as
one GPU's share of a serving batch / of config 4's corpus): `--utterances` random mels of `--frames` frames per
GPU (default 16 x 641 frames = 16 x 8 s of audio -> 16 x 16 = 256 folded segments x T=12100 autoregressive steps; the
same line also carries the 8-utterance batch of round 1 and BASELINE config 2's single-utterance calls, N = 481 / 1001).
Timed region (mels already resident in HBM): up-sample network (HIP pre-loop kernels: MelResNet + the first two up-sampling stages)
-> per slab of steps {aux tables -> the persistent loop kernel, which forms the last up-sampling stage and the I-layer conditioning
itself} -> [N>1: RCCL all-gather of the [n,T] audio] -> cross-fade/unfold on the
device -> D2H of the float64 waveforms.  The WAV write is excluded (the CPU baseline excludes it too).  Sampling noise is
drawn on the device (Philox), as the reference does when it runs on a GPU; `--parity-noise` uses the host
MT19937 stream of the parity tests instead (adds ~25 M host RNG draws per pass).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python bench.py --gpus N --steps K --warmup W          # WORLD_SIZE unset: spawns its own N ranks (one per GPU, 127.0.0.1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W             # ... or is launched as one rank of N (RANK / WORLD_SIZE in the env)
    python bench.py --gpus N --corpus config4              # BASELINE config 4: the fixed 64-utterance corpus (942 segments), STRONG scaling

value = useful audio samples per second over all ranks = N * K * sum(wave_len) / max-over-ranks(time).
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
MFMA_F32_PEAK_TF = 157.3    # v_mfma_f32_16x16x4_f32 dense peak (same guide)


def cpu_baseline(sd, mode, frames, target, overlap, budget_s):
    """The oracle's C restatement ("port") timed on this box's host cores on a bounded sample of the same
    workload: ONE of the utterances (its B segments), the first `ts` of the T steps (~budget_s of CPU work)."""
    from oracle import wavernn_oracle as O, c_oracle as C
    from wavernn_amd.synthetic import random_mel
    mel = random_mel(1234, frames)
    mels, aux, wave_len = O.conditioning(sd, mel, True, target, overlap)
    B, T, _ = mels.shape
    noise = O.draw_noise(77, mode, B, T)
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
    ts = int(max(40, min(noise[1].shape[0], budget_s / (best[1] / 40))))     # ~budget_s of CPU work
    t0 = time.perf_counter()
    C.loop(sd, mode, mels[:, :ts], aux[:, :ts], (noise[0][:ts], noise[1][:ts]), nthreads=cores)
    dt = time.perf_counter() - t0
    seg_steps_per_s = B * ts / dt
    useful = seg_steps_per_s * wave_len / (B * T)          # same useful/raw ratio as the full workload
    res = dict(value=round(useful, 1), unit='audio samples/s', cores=cores, kind='port',
               sample=f'oracle/wrnn_oracle.c (OpenMP, {cores} of {ncpu} host threads; fastest of 8/16/32/64), one utterance '
                      f'of the batch: B={B} segments x first {ts} of {T} steps ({dt:.1f} s); {seg_steps_per_s:.0f} segment-steps/s',
               realtime_factor=round(useful / SAMPLE_RATE, 4))
    try:
        res['pytorch_eager'] = pytorch_eager_baseline(sd, mode, target, overlap, max(3.0, budget_s / 2))
    except Exception as e:
        res['pytorch_eager'] = {'error': repr(e)}
    # the reference ITSELF (unmodified generate(), /root/reference) exists only in the build container: its measured time there
    # (scripts/make_golden.py -> the fixtures' `ref_cpu_seconds`) rides along, box and core count spelled out
    res['reference_pytorch'] = dict(value=7540.0, unit='audio samples/s', realtime_factor=0.342, cores=8,
                                    box='build container (Intel Xeon @ 2.10 GHz, 8 cores, no GPU) -- NOT this GPU box',
                                    sample="the unmodified reference WaveRNN.generate() on BASELINE config 2's stated input (N = 481 -> B = 12 x T = 12100): "
                                           '17.51 s (tests/golden/mol_batched_481f.npz config.ref_cpu_seconds)')
    return res


def pytorch_eager_baseline(sd, mode, target, overlap, budget_s):
    """"The reference PyTorch CPU generate()" of the north star, on THIS box's host cores: oracle/torch_eager.py runs the reference's
    loop as the same PyTorch eager ops in the same order (pinned to the reference's outputs in tests/test_oracle_golden.py; the
    reference tree itself is not on the GPU box) on BASELINE config 2's stated input (N = 481 -> B = 12), for a bounded number
    of steps at the fastest of a few thread counts."""
    from oracle import wavernn_oracle as O, torch_eager as TE
    from wavernn_amd.synthetic import random_mel
    mels, aux, wave_len = O.conditioning(sd, random_mel(1234, 481), True, target, overlap)
    B, T, _ = mels.shape
    ncpu = os.cpu_count() or 1
    was = torch.get_num_threads()
    best = None
    try:
        for nt in sorted({min(8, ncpu), min(32, ncpu), ncpu}):
            torch.set_num_threads(nt)
            TE.loop(sd, mode, mels, aux, seed=77, steps=20)
            _, d = TE.loop(sd, mode, mels, aux, seed=77, steps=100)
            if best is None or d < best[1]:
                best = (nt, d)
        torch.set_num_threads(best[0])
        steps = int(max(200, min(T, budget_s / (best[1] / 100))))
        _, dt = TE.loop(sd, mode, mels, aux, seed=77, steps=steps)
    finally:
        torch.set_num_threads(was)
    useful = (B * steps / dt) * wave_len / (B * T)
    return dict(value=round(useful, 1), unit='audio samples/s', realtime_factor=round(useful / SAMPLE_RATE, 4), cores=best[0], kind='port',
                ms_per_step=round(dt / steps * 1e3, 4),
                sample=f'oracle/torch_eager.py (the reference loop as PyTorch eager CPU ops, torch {torch.__version__}, {best[0]} of {ncpu} host threads: '
                       f'fastest of 8 / 32 / all), N = 481 -> B = {B} segments x first {steps} of {T} steps ({dt:.1f} s)')


def source_sha16():
    """Hash of the kernel sources a measurement belongs to (keys profiles/traffic_latest.json to the code it measured; the
    workload is matched separately: kernel, mode, segments, T): the translation units and headers the vocoder's loop kernels,
    their launcher and the conditioning slabs they read are compiled from -- not the Tacotron / generic / sparse / self-test
    units, which cannot change what those dispatches move."""
    from wavernn_amd.batch import kernel_source_sha16
    return kernel_source_sha16()


def self_launch(n):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves (one process per GPU, rendezvous on
    127.0.0.1), relay rank 0's stdout so that its JSON line is the LAST line this process prints, propagate failures."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0) or None))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit(f'bench.py: ranks failed (rank, exit code): {bad}')


def load_factory(spec):
    """'package.module:function' -> the function (the --dry-host loop stand-in is named by the caller, i.e. by a test)."""
    import importlib
    mod, _, fn = spec.partition(':')
    return getattr(importlib.import_module(mod), fn)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--utterances', type=int, default=16, help='utterances per GPU')
    ap.add_argument('--frames', type=int, default=641, help='mel frames per utterance')
    ap.add_argument('--algo', default='auto', choices=['auto', 'loop', 'duo', 'sparse', 'stream'])
    ap.add_argument('--mode', default='MOL', choices=['MOL', 'RAW'], help="RAW = 9-bit mu-law ('bits') softmax sampling")
    ap.add_argument('--force-dist', action='store_true', help='initialise a torch.distributed nccl (= RCCL) group even at --gpus 1')
    ap.add_argument('--no-single', action='store_true', help="skip the single-utterance generate() entries (BASELINE config 2's N=481 / N=1001)")
    ap.add_argument('--prune', type=float, default=0.0,
                    help='BASELINE config 5: block-prune the GRU matrices (16x1 blocks) to this sparsity, e.g. 0.95')
    ap.add_argument('--prune-linear', action='store_true', help='... and fc1 / fc2 with them, as the reference\'s pruning notebook prunes its Linear layer (the gathered fc stages of wrnn_sparse_kernel)')
    ap.add_argument('--parity-noise', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--corpus', default='batch', choices=['batch', 'config4'],
                    help="batch: --utterances x --frames per GPU (weak scaling).  config4: BASELINE config 4's fixed corpus -- 64 "
                         "utterances of RandomState(2024).randint(300, 901) frames = 942 folded segments, sharded over the ranks "
                         "(strong scaling)")
    ap.add_argument('--corpus-limit', type=int, default=64, help=argparse.SUPPRESS)     # tests: only the first K utterances of config 4
    ap.add_argument('--no-config4-leg', action='store_true', help="skip the `config.config4` leg (BASELINE config 4's fixed corpus shared by the N ranks)")
    ap.add_argument('--target', type=int, default=11000, help=argparse.SUPPRESS)
    ap.add_argument('--overlap', type=int, default=550, help=argparse.SUPPRESS)
    ap.add_argument('--dry-host', default=None, metavar='MODULE:FACTORY',
                    help='LAUNCH-PATH TEST, not a measurement: run the host side (rank spawn, gloo group, sharding, all-gather, '
                         'unfold, the JSON line) on CPU with the loop replaced by FACTORY(state_dict, mode) -> loop_fn; the line '
                         'is marked "dry_host": true and carries no roofline')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        return self_launch(args.gpus)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus and rank == 0:      # the launcher's world size is what runs; --gpus is only what was asked for
        print(f'bench.py: WORLD_SIZE={world} overrides --gpus {args.gpus}', file=sys.stderr)
    dry = args.dry_host is not None
    import torch.distributed as dist
    group = None
    if dry:
        dev = torch.device('cpu')
        torch.set_num_threads(2)
    else:
        if not torch.cuda.is_available():
            raise SystemExit('bench.py needs a HIP device: the product path has no CPU fallback')
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if world > 1 or args.force_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29513')
        if dry:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm
        group = dist.group.WORLD
    world_seen = dist.get_world_size(group) if group is not None else 1      # what the process group actually spans

    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    from wavernn_amd.batch import generate_corpus, plan_utterances
    from wavernn_amd import _lib

    mode, target, overlap, hop = args.mode, args.target, args.overlap, 275
    sd = random_state_dict(0, mode=mode)
    if args.prune > 0:
        from wavernn_amd.prune import block_prune_state_dict
        sd, _ = block_prune_state_dict(sd, args.prune, (16, 1), linear=args.prune_linear)
    model = WaveRNN(**SHIPPED, mode=mode)
    model.num_params = lambda *a, **k: 0
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    model.loop_algo = args.algo
    if args.corpus == 'config4':
        # BASELINE config 4: the corpus is FIXED, the ranks share it -> strong scaling
        frames = [int(n) for n in np.random.RandomState(2024).randint(300, 901, 64)][:args.corpus_limit]
        mel_seeds, seeds = [1000 + u for u in range(len(frames))], [4000 + u for u in range(len(frames))]
    else:
        # the whole job's batch: `utterances` per rank (weak scaling: fixed work per GPU); rank r's share is the r-th block
        frames = [args.frames] * (args.utterances * world)
        mel_seeds, seeds = [1234 + u for u in range(len(frames))], [77 + u for u in range(len(frames))]
    n_utt = len(frames)
    mels = [torch.from_numpy(random_mel(ms, n)).unsqueeze(0).to(dev) for ms, n in zip(mel_seeds, frames)]
    plan = plan_utterances([n * hop for n in frames], target, overlap)
    wave_total = sum((n - 1) * hop for n in frames)
    noise_source = 'cpu' if args.parity_noise else 'device'
    if dry:
        eng, loop_fn = None, load_factory(args.dry_host)(sd, mode)
    else:
        eng, loop_fn = model._loop_engine(), None

    def one_pass():
        outs = generate_corpus(model, mels, target, overlap, True, seeds, group=group, noise_source=noise_source,
                               finish='own', check=False, loop_fn=loop_fn)
        if eng is not None:
            eng.status()
        return outs

    def fence():
        if not dry:
            torch.cuda.synchronize()
        if group is not None:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_pass()
    fence()
    loop_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
        if eng is not None:
            loop_ms.append(eng.last_loop_ms())
    fence()
    dt = time.perf_counter() - t0
    main_info = eng.last_run_info() if eng is not None else None
    if group is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    from wavernn_amd.batch import shard_bounds, choose_ranks, step_table

    def measure_step_table():
        """us per step of the loop kernel `auto` picks at 1 .. 8 groups in flight per cluster (64 .. 512 segments on this GPU), MoL, synthetic
        conditioning, 200 steps each: what `choose_ranks` weighs a split of the fixed corpus with -- measured here, on this device and these
        sources, so that the planner cannot run on stale numbers (round-5 verdict).  Rank 0 also writes it to gpurun_out/step_us.json (copied
        to profiles/step_us.json by the builder: `wavernn_amd.batch.step_table` reads it where nothing can be measured, e.g. dry-host runs)."""
        if dry or eng is None or mode != 'MOL' or args.prune > 0:
            return None
        rs = np.random.RandomState(5)
        T4, stride = 200, 64
        tab = {}
        for d in range(1, 9):
            B4 = 64 * d
            L4 = (B4 * stride + T4 + hop - 1) // hop * hop
            mu = torch.from_numpy(rs.uniform(0, 1, (L4, 80)).astype(np.float32)).to(dev)
            au = torch.from_numpy(rs.uniform(-1, 1, (L4 // hop, 128)).astype(np.float32)).to(dev)
            nz = torch.empty(T4, 11 * B4, device=dev).uniform_(1e-5, 1 - 1e-5)
            for _ in range(2):
                eng.run(mu, au, B4, T4, stride, nz, hop, algo='auto')
            tab[d] = round(eng.last_loop_ms() * 1e3 / T4, 3)
        if rank == 0:
            try:
                os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
                json.dump({'source_sha16': source_sha16(), 'step_us_by_depth': tab, 'device': torch.cuda.get_device_name(dev),
                           'what': 'us per step of the loop kernel `auto` picks for 64 * depth segments (MoL, 200 steps, synthetic conditioning); bench.py'},
                          open(os.path.join(ROOT, 'gpurun_out', 'step_us.json'), 'w'), indent=1)
            except OSError:
                pass
        return tab

    def config4_leg():
        """BASELINE config 4 beside the weak-scaling value (round-4 verdict, item 6): the FIXED corpus -- 64 random mels of 300-900 frames =
        942 folded segments, 479 s of audio -- sharded over the N ranks of THIS launch (strong scaling), timed like the main loop (barrier +
        synchronise on both sides, max over ranks).  `batch.choose_ranks` may leave ranks out when fewer fill the pipeline to the same depth.
        Every rank runs this (collectives inside); rank 0 reports."""
        f4 = [int(n) for n in np.random.RandomState(2024).randint(300, 901, 64)][:args.corpus_limit]
        p4 = plan_utterances([n * hop for n in f4], target, overlap)
        m4 = [torch.from_numpy(random_mel(1000 + u, n)).unsqueeze(0).to(dev) for u, n in enumerate(f4)]
        s4 = [4000 + u for u in range(len(f4))]
        table = measure_step_table()
        if group is not None and table is not None:      # every rank must plan with the SAME table: rank 0's
            tt = torch.tensor([table[d] for d in range(1, 9)], dtype=torch.float64, device=dev)
            dist.broadcast(tt, src=0, group=group)
            table = {d: float(v) for d, v in zip(range(1, 9), tt.tolist())}
        active = choose_ranks(p4.n_segments, world, table)
        tm = {}

        def pass4():
            generate_corpus(model, m4, target, overlap, True, s4, group=group, noise_source=noise_source, finish='own', check=False,
                            loop_fn=loop_fn, ranks=active, timings=tm)
            if eng is not None:
                eng.status()
        reps = 1 if dry else 2
        if not dry:
            pass4()
        tm.clear()
        fence()
        t0 = time.perf_counter()
        for _ in range(reps):
            pass4()
        fence()
        d4 = (time.perf_counter() - t0) / reps
        if group is not None:
            tt = torch.tensor([d4, tm.get('gather_wait_ms', 0.0) / reps, tm.get('unfold_under_gather_ms', 0.0) / reps], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            d4, gw, uu = (float(x) for x in tt.tolist())
        else:
            gw, uu = tm.get('gather_wait_ms', 0.0) / reps, tm.get('unfold_under_gather_ms', 0.0) / reps
        w4 = sum((n - 1) * hop for n in f4)
        out = {'what': "BASELINE config 4: the fixed corpus sharded over the ranks of this launch (STRONG scaling; `value` above is the weak-scaling batch)",
               'utterances': len(f4), 'segments': p4.n_segments, 'steps_per_segment': p4.T, 'world_seen': world_seen, 'ranks_used': active,
               'segments_per_rank': [h - l for l, h in shard_bounds(p4.n_segments, world, active)],
               'ms_per_pass': round(d4 * 1e3, 3), 'gather_wait_ms': round(gw, 3), 'unfold_under_gather_ms': round(uu, 3),
               'group_world_size_seen_by_the_collective': int(tm.get('world_seen', 1)),
               'all_gather_bytes_received_per_rank_per_pass': int(tm.get('gather_bytes', 0) // max(1, reps)),
               'planner_step_us_by_depth': table if table is not None else step_table()[0],
               'planner_table_origin': 'measured in this run (bench.py measure_step_table)' if table is not None else step_table()[1]}
        if dry:
            out['dry_host'] = True
        else:
            out.update(samples_per_s=round(w4 / d4, 1), realtime_factor=round(w4 / d4 / SAMPLE_RATE, 2),
                       split_rank0=eng.last_run_info(), loop_kernel_ms_rank0=round(eng.last_loop_ms(), 3))
        return out
    c4leg = None
    if args.corpus == 'batch' and not args.no_config4_leg:
        try:
            c4leg = config4_leg()
        except Exception as e:
            if group is not None:
                raise                        # (a rank that drops out of a collective would hang the others)
            c4leg = {'error': repr(e)}
    lo0, hi0 = shard_bounds(plan.n_segments, world)[0]
    par = (f'{world_seen} rank(s) in the process group ({"gloo, DRY RUN on the host" if dry else "nccl = RCCL" if group is not None else "no group"}): '
           f'1 process per GPU, contiguous block of the segment table, ONE all-gather of the finished audio')
    if dry:
        # order check of the launch path: the gathered [n_segments, T] table, weighted by row, against what ONE rank computes
        segs, _ = generate_corpus(model, mels, target, overlap, True, seeds, group=group, noise_source=noise_source, loop_fn=loop_fn,
                                  return_segments=True)
        checksum = float((segs[:, 0] * (1.0 + np.arange(segs.shape[0]))).sum())
        if rank == 0:
            print(json.dumps({
                'metric': 'audio samples/sec (real-time factor @22.05 kHz), MoL WaveRNN batched generate', 'value': None,
                'unit': 'audio samples/s', 'n_gpus': world_seen, 'steps': args.steps, 'warmup': args.warmup,
                'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
                'scaling': 'strong' if args.corpus == 'config4' else 'weak', 'vs_baseline': None, 'dtype': 'f32',
                'data': 'synthetic', 'dry_host': True,
                'config': {'workload': f'DRY RUN of the launch path on the host, NOT a measurement: {n_utt} utterances -> '
                                       f'{plan.n_segments} segments x T={plan.T}, loop stand-in {args.dry_host}',
                           'segments_rank0': hi0 - lo0, 'segments_per_rank': [h - l for l, h in shard_bounds(plan.n_segments, world)],
                           'gathered_rows': int(segs.shape[0]), 'gathered_checksum': checksum,
                           'host_samples_per_s': round(args.steps * wave_total / dt, 1),
                           **({'config4': c4leg} if c4leg is not None else {}),
                           'parallelism': par}}), flush=True)
        if group is not None:
            dist.destroy_process_group()
        return
    info = main_info

    def single_utterance(n_frames, model=model, eng=eng):
        """ONE `WaveRNN.generate()` call -- the drop-in API itself -- on BASELINE config 2's stated input (mel seed
        1234, N frames, target 11000 / overlap 550): wall time of the whole call incl. the WAV write, device noise."""
        import tempfile
        mel = torch.from_numpy(random_mel(1234, n_frames)).unsqueeze(0).to(dev)
        was = model.noise_source
        model.noise_source = 'device'
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'o.wav')
            model.generate(mel, path, True, target, overlap, True)                   # warm-up
            times, kms = [], []
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                wav = model.generate(mel, path, True, target, overlap, True)
                times.append(time.perf_counter() - t0)
                kms.append(model.last_loop_ms)
        model.noise_source = was
        model.eval()
        B = plan_utterances([n_frames * hop], target, overlap).n_segments
        T = target + 2 * overlap
        wall, k = float(np.median(times)), float(np.median(kms))
        bytes_eq = (eng.weight_bytes + B * 836) * T
        return {'N_frames': n_frames, 'segments': B, 'steps': T, 'samples': int(wav.shape[0]),
                'samples_per_s': round(wav.shape[0] / wall, 1), 'realtime_factor': round(wav.shape[0] / wall / SAMPLE_RATE, 2),
                'generate_wall_ms': round(wall * 1e3, 2), 'loop_kernel_ms': round(k, 2), 'us_per_step': round(k * 1e3 / T, 3),
                'kernel': model.last_loop_kernel, 'split': eng.last_run_info(),
                'hbm_equivalent_GBps': round(bytes_eq / (k * 1e-3) / 1e9, 1), 'hbm_equivalent_frac': round(bytes_eq / (k * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}

    def config3():
        """BASELINE config 3 end to end (gen_tacotron.py:131-149): text -> Tacotron (encoder, decoder loop as one persistent
        kernel, CBHG post-net with its GRUs as persistent kernels) -> `_, m, _` -> (m + 4) / 8 clipped -> this vocoder's
        `generate()` (batched, target 11000 / overlap 550), random-init Tacotron of the reference's architecture, 800 frames."""
        import tempfile
        from wavernn_amd.synthetic import random_tacotron_state_dict
        from wavernn_amd.tacotron import TacotronInference, text_to_ids, tacotron_to_wavernn_mel
        shapes = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'wavernn_amd', 'tacotron_shapes.json')))
        tts = TacotronInference(random_tacotron_state_dict(3, shapes), device=dev)
        ids = text_to_ids('Scientists at the CERN laboratory say they have discovered a new particle.')
        was = model.noise_source
        model.noise_source = 'device'
        steps = 800
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'o.wav')

            def run():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _, lin, _ = tts.generate(ids, steps=steps, kernel=True)
                t1 = time.perf_counter()
                wav = model.generate(torch.tensor(tacotron_to_wavernn_mel(lin)).unsqueeze(0), path, True, target, overlap, True)
                return wav, t1 - t0, time.perf_counter() - t1
            run()
            runs = [run() for _ in range(3)]
        model.noise_source = was
        model.eval()
        wav = runs[0][0]
        t_tts, t_voc = float(np.median([r[1] for r in runs])), float(np.median([r[2] for r in runs]))
        audio_s = wav.shape[0] / SAMPLE_RATE
        return {'what': 'BASELINE config 3: Tacotron (800 decoder steps, decoder loop + CBHG GRUs as persistent kernels) -> MoL WaveRNN, one sentence, end to end incl. the WAV write',
                'tacotron_ms': round(t_tts * 1e3, 2), 'vocoder_ms': round(t_voc * 1e3, 2), 'loop_kernel_ms': round(float(model.last_loop_ms), 2),
                'kernel': model.last_loop_kernel, 'audio_s': round(audio_s, 3), 'samples': int(wav.shape[0]),
                'realtime_factor': round(audio_s / (t_tts + t_voc), 2)}

    if rank == 0:
        T = plan.T
        n_local = hi0 - lo0                                  # segments this GPU's launches advance (rank 0's block)
        value = args.steps * wave_total / dt
        W = eng.weight_bytes
        kms = float(np.mean(loop_ms))
        # unit = one batch step (all segments resident on the GPU advance one sample);
        # algorithmic bytes per batch step = W + n*836; one launch = T batch steps
        bytes_per_launch = (W + n_local * 836) * T
        achieved = bytes_per_launch / (kms * 1e-3) / 1e9
        wkeys = ('I.weight', 'rnn1.weight_ih_l0', 'rnn1.weight_hh_l0', 'rnn2.weight_ih_l0', 'rnn2.weight_hh_l0', 'fc1.weight',
                 'fc2.weight', 'fc3.weight')
        nnz = int(sum(np.count_nonzero(sd[k]) for k in wkeys))      # 3,825,152 for the dense MoL model
        flops_per_launch = 2.0 * nnz * n_local * T
        tf = flops_per_launch / (kms * 1e-3) / 1e12
        u_per_wg, ncl, depth = info['units_per_wg'], info['clusters'], info['depth']
        # HBM-side traffic of the loop kernel per pass, from the rocprofv3 PMC passes of THIS command line on THIS source tree
        # (scripts/summarize_profile.py writes the file; it is ignored unless kernel, geometry and source hash all match)
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if (tj.get('kernel') == info['kernel'] and tj.get('segments') == n_local and tj.get('T') == T and tj.get('mode') == mode
                    and tj.get('source_sha16') == source_sha16()):
                traffic = tj.get('bytes_per_pass')
        res = {
            'metric': 'audio samples/sec (real-time factor @22.05 kHz), MoL WaveRNN batched generate' if mode == 'MOL' else
                      "audio samples/sec (real-time factor @22.05 kHz), 9-bit mu-law ('bits') WaveRNN batched generate",
            'value': round(value, 1), 'unit': 'audio samples/s', 'n_gpus': world_seen, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'strong' if args.corpus == 'config4' else 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'realtime_factor': round(value / SAMPLE_RATE, 2),
            'config': {'workload': (f'BASELINE config 5 (GRU matrices block-pruned to {args.prune:.0%} zeros, 16x1 blocks) = ' if args.prune > 0 else '') +
                                   f'BASELINE config 2 ({"MoL" if mode == "MOL" else "9-bit mu-law RAW"} WaveRNN, rnn/fc 512, random-init weights, batched fold target={target} '
                                   f'overlap={overlap}) on ' + (f"BASELINE config 4's fixed corpus of {n_utt} random mels of 300-900 frames ({plan.n_segments} segments) shared by {world} GPU(s) "
                                                                 if args.corpus == 'config4' else f'a batch of {args.utterances} random {args.frames}-frame mels per GPU ') +
                                   f'-> {n_local} folded segments x T={T} steps per GPU ({info["launches"]} loop-kernel launches per pass: '
                                   f'{info["rounds"]} round(s) x conditioning slabs of {info["slab_steps"]} steps), '
                                   f'{wave_total // world} output samples per GPU per step',
                       'segments_per_gpu': n_local, 'steps_per_segment': T,
                       'mode': mode, 'kernel': info['kernel'], 'launches_per_pass': info['launches'], 'rounds': info['rounds'],
                       'slab_steps': info['slab_steps'], 'units_per_workgroup': u_per_wg, 'clusters': ncl, 'groups_in_flight_per_cluster': depth,
                       'segment_steps_per_s': round(plan.n_segments * T * args.steps / dt, 1),
                       'noise': 'host MT19937 stream (parity mode)' if args.parity_noise else 'device Philox (as the reference on a GPU)',
                       'mel_last_stage': 'formed inside the loop kernel from the x25 mel (no [L, 80] up-sampled mel is written)'
                                         if model.mel_rows_ok(eng, n_local, T)
                                         else 'materialised by the pre-loop kernels',
                       'parallelism': par},
        }
        # Which roofline bounds the loop: arithmetic intensity = 2n FLOP per 4 weight bytes = n/2 FLOP/B for n
        # segments per weight pass; the f32 ridge of gfx950 is 157.3 TF / 6.3 TB/s = 25 FLOP/B, i.e. weight-bandwidth-bound
        # below ~50 resident segments and f32-MFMA-bound above.  The other figure is reported beside it.
        hbm = {'bound': 'hbm', 'achieved': round(achieved, 2), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
               'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic, 'kernel': info['kernel'],
               'kernel_ms': round(kms, 3), 'algorithmic_bytes_per_launch': bytes_per_launch,
               'note': 'weight-streaming-equivalent bandwidth, SURVEY.md 8(d): (W + n*836 B) per batch step x T steps / kernel '
                       'time (kernel_ms = sum of the loop-kernel launch durations of one pass, HIP events on the launch stream); the weights are on-chip resident, so this is a per-step latency '
                       'figure of merit, not HBM traffic (DESIGN.md)'}
        mfma = {'bound': 'mfma', 'achieved': round(tf, 3), 'peak': MFMA_F32_PEAK_TF, 'unit': 'TFLOP/s',
                'frac': round(tf / MFMA_F32_PEAK_TF, 5), 'traffic': traffic, 'kernel': info['kernel'],
                'kernel_ms': round(kms, 3), 'launches': info['launches'], 'algorithmic_flops_per_pass': flops_per_launch,
                'weights_nnz': nnz,
                'note': 'useful f32 FLOPs of the loop (2 x non-zero loop weights per segment-step x n x T) / kernel time (kernel_ms = '
                        'sum of the loop-kernel launch durations of one pass, HIP events on the launch stream) vs the dense f32 MFMA peak; n >= 50 resident segments puts the loop right of the '
                        'f32 ridge (SURVEY.md 8d)'}
        if c4leg is not None:
            res['config']['config4'] = c4leg
        if n_local >= 50:
            res['roofline'], res['roofline_hbm_equivalent'] = mfma, hbm
        else:
            res['roofline'], res['roofline_mfma'] = hbm, mfma
        if not args.no_single and world == 1 and args.prune == 0 and args.corpus == 'batch':
            try:
                res['config']['single_utterance'] = [single_utterance(481), single_utterance(1001)]
            except Exception as e:
                res['config']['single_utterance'] = {'error': repr(e)}
            if mode == 'MOL':
                try:
                    res['config']['config3'] = config3()
                except Exception as e:
                    res['config']['config3'] = {'error': repr(e)}
            if args.utterances > 8:
                try:    # round 1's workload (8 utterances = 128 segments per GPU), for continuity
                    sub = mels[:8]

                    def small_pass():
                        generate_corpus(model, sub, target, overlap, True, seeds[:8], noise_source=noise_source, finish='own', check=False)
                        eng.status()
                    small_pass()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        small_pass()
                    torch.cuda.synchronize()
                    d8 = (time.perf_counter() - t0) / 2
                    w8 = 8 * (args.frames - 1) * hop
                    k8 = eng.last_loop_ms()
                    i8 = eng.last_run_info()
                    res['config']['batch_of_8_utterances'] = {
                        'segments': 8 * (plan.n_segments // n_utt), 'samples_per_s': round(w8 / d8, 1), 'realtime_factor': round(w8 / d8 / SAMPLE_RATE, 2),
                        'ms_per_pass': round(d8 * 1e3, 3), 'loop_kernel_ms': round(k8, 3), 'split': i8,
                        'mfma_frac': round(2.0 * nnz * 8 * (plan.n_segments // n_utt) * T / (k8 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 5)}
                except Exception as e:
                    res['config']['batch_of_8_utterances'] = {'error': repr(e)}
        if not args.no_single and world == 1 and args.prune == 0 and args.corpus == 'batch' and mode == 'MOL':
            try:    # BASELINE config 4 is strong scaling: at N = 8 a GPU owns 117-118 of the corpus' 942 segments.  What that share runs at,
                    # on this one GPU (rank 0's block of the fixed 64-utterance corpus, no collective): the rate the 8-GPU number is made of
                from wavernn_amd.batch import shard_bounds as _sb
                f4 = [int(n) for n in np.random.RandomState(2024).randint(300, 901, 64)]
                p4 = plan_utterances([n * hop for n in f4], target, overlap)
                lo4, hi4 = _sb(p4.n_segments, 8)[0]
                m4 = [torch.from_numpy(random_mel(1000 + u, n)).unsqueeze(0).to(dev) for u, n in enumerate(f4)]

                def share_pass():
                    generate_corpus(model, m4, target, overlap, True, None, noise_source='device', finish='own', check=False, shard=(0, 8))
                    eng.status()
                share_pass()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(2):
                    share_pass()
                torch.cuda.synchronize()
                d4 = (time.perf_counter() - t0) / 2
                k4, i4 = eng.last_loop_ms(), eng.last_run_info()
                res['config']['config4_share_of_8'] = {
                    'what': "BASELINE config 4 at N = 8, one GPU's share: rank 0's block of the fixed 64-utterance corpus (942 segments), run alone on this GPU",
                    'segments': hi4 - lo4, 'segment_steps_per_s': round((hi4 - lo4) * T / d4, 1), 'ms_per_pass': round(d4 * 1e3, 3),
                    'loop_kernel_ms': round(k4, 3), 'split': i4,
                    'mfma_frac': round(2.0 * nnz * (hi4 - lo4) * T / (k4 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 5),
                    'corpus_realtime_factor_if_8_gpus_ran_at_this_rate': round(sum((n - 1) * hop for n in f4) / d4 / SAMPLE_RATE, 1)}
                del m4
            except Exception as e:
                res['config']['config4_share_of_8'] = {'error': repr(e)}
            # the other two single-GPU configurations of BASELINE.json on the SAME 16-utterance geometry, so that they are in the
            # driver's record too: the bit-exact 9-bit mu-law mode (config 1's model, batched) and config 5 (95 % block-sparse GRUs)
            def side_config(sd2, mode2, label, algo2='auto'):
                try:
                    m2 = WaveRNN(**SHIPPED, mode=mode2)
                    m2.num_params = lambda *a, **k: 0
                    m2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd2.items()}, strict=True)
                    m2 = m2.to(dev).eval()
                    m2.loop_algo = algo2
                    e2 = m2._loop_engine()

                    def p2():
                        generate_corpus(m2, mels, target, overlap, True, seeds, noise_source=noise_source, finish='own', check=False)
                        e2.status()
                    p2()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        p2()
                    torch.cuda.synchronize()
                    d = (time.perf_counter() - t0) / 2
                    k2, i2 = e2.last_loop_ms(), e2.last_run_info()
                    nnz2 = int(sum(np.count_nonzero(sd2[k]) for k in wkeys))
                    out = {'what': label, 'mode': mode2, 'segments': n_local, 'samples_per_s': round(wave_total / d, 1),
                           'realtime_factor': round(wave_total / d / SAMPLE_RATE, 2), 'ms_per_pass': round(d * 1e3, 3),
                           'loop_kernel_ms': round(k2, 3), 'split': i2, 'weights_nnz': nnz2,
                           'mfma_frac': round(2.0 * nnz2 * n_local * T / (k2 * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 5)}
                    del e2
                    m2._engine = None
                    return out
                except Exception as e:
                    return {'what': label, 'error': repr(e)}
            res['config']['raw'] = side_config(random_state_dict(0, mode='RAW'), 'RAW', "9-bit mu-law ('bits', the bit-exact mode) on the same batch")
            try:    # ... and the bit-exact mode's one-utterance call (BASELINE config 2's stated input, N = 481, through WaveRNN.generate())
                mr = WaveRNN(**SHIPPED, mode='RAW')
                mr.num_params = lambda *a, **k: 0
                mr.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in random_state_dict(0, mode='RAW').items()}, strict=True)
                mr = mr.to(dev).eval()
                res['config']['raw']['single_utterance'] = single_utterance(481, mr, mr._loop_engine())
                mr._engine = None
            except Exception as e:
                res['config']['raw']['single_utterance'] = {'error': repr(e)}
            from wavernn_amd.prune import block_prune_state_dict
            # BASELINE config 5 = "the Pruning notebook config": the notebook prunes `[self.rnn, self.fc]` (Pruning - Scratchpad.ipynb :199-204) -- GRU matrices
            # AND Linear layers; `config5_gru_only` is rounds 1-5's pack (fc1 / fc2 dense)
            res['config']['config5'] = side_config(block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1), linear=True)[0], 'MOL',
                                                   'BASELINE config 5: GRU matrices AND fc1 / fc2 95 % block-sparse (16x1 blocks; the pruning notebook prunes the Linear '
                                                   'layers with the GRUs) on the same batch, `auto` kernel (wrnn_sparse_kernel: 16 clusters of 16 CUs, one group of 16 '
                                                   'segments each, gathered gate AND fc stages)')
            if 'samples_per_s' in res['config']['config5']:
                res['config']['config5']['vs_dense_value'] = round(res['config']['config5']['samples_per_s'] / value, 3)
            res['config']['config5_gru_only'] = side_config(block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1))[0], 'MOL',
                                                            'config 5 with ONLY the GRU matrices pruned (rounds 1-5): dense fc1 / fc2 stages on the cluster\'s chain')
            if 'samples_per_s' in res['config']['config5_gru_only']:
                res['config']['config5_gru_only']['vs_dense_value'] = round(res['config']['config5_gru_only']['samples_per_s'] / value, 3)
            res['config']['config5_dense_kernel'] = side_config(block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1))[0], 'MOL',
                                                                'BASELINE config 5 on the DENSE wrnn_duo_kernel (algo = duo: the pruned weights as masked dense matrices)', algo2='duo')
        if not args.no_cpu_baseline and world == 1:
            try:
                res['cpu_baseline'] = cpu_baseline(sd, mode, args.frames, target, overlap, args.cpu_seconds)
            except Exception as e:   # the baseline is a report, never a reason to lose the GPU number
                res['cpu_baseline'] = {'error': repr(e)}
        sys.stdout.flush()
        try:    # RCCL / HIP libraries printf into the C stdio buffer, which is flushed at exit, i.e. AFTER this line: empty it first so
            import ctypes                      # the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if group is not None:
        dist.destroy_process_group()
    # the C-ABI pack must go before the HIP runtime tears down at interpreter exit
    del eng
    model._engine = None


if __name__ == '__main__':
    main()
