#!/usr/bin/env python
"""Whole-pass time of the benchmarked workload (16 x 641-frame utterances through generate_corpus: pre-loop kernels, loop, unfold, audio back on the host) under
the host-side settings of the pass -- model.pre_streams (utterances' pre-loop kernels side by side) and model.pinned_output (audio through page-locked memory):
    python scripts/gpu_pass_ab.py [--prune 0.95 --prune-linear] [--mode RAW]"""
import argparse, sys, os, json, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--mode', default='MOL'); ap.add_argument('--utterances', type=int, default=16); ap.add_argument('--passes', type=int, default=4)
ap.add_argument('--prune', type=float, default=0.0); ap.add_argument('--prune-linear', action='store_true')
ap.add_argument('--noise-chunk-mb', type=int, default=0, help='model.noise_chunk_bytes (0: the default)')
ap.add_argument('--settings', default='1:0:1,8:1:1,8:1:0', help='pre_streams:pinned_output:mel_in_loop triples')
a = ap.parse_args()
from wavernn_amd.model import WaveRNN
from wavernn_amd.batch import generate_corpus
from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
dev = torch.device('cuda', 0)
sd = random_state_dict(0, mode=a.mode)
if a.prune > 0:
    from wavernn_amd.prune import block_prune_state_dict
    sd, _ = block_prune_state_dict(sd, a.prune, (16, 1), linear=a.prune_linear)
m = WaveRNN(**SHIPPED, mode=a.mode); m.num_params = lambda *x, **k: 0
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True); m = m.to(dev).eval()
if a.noise_chunk_mb:
    m.noise_chunk_bytes = a.noise_chunk_mb << 20
mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0).to(dev) for u in range(a.utterances)]
for st in a.settings.split(','):
    ps, pin, mil = (int(x) for x in st.split(':'))
    m.pre_streams, m.pinned_output, m.mel_in_loop = ps, bool(pin), bool(mil)
    best, loop, tot = 1e9, 0.0, []
    for _ in range(a.passes + 1):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = generate_corpus(m, mels, 11000, 550, True, None, noise_source='device', finish='own', check=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        tot.append(dt)
        if dt < best:
            best, loop = dt, m._loop_engine().last_loop_ms()
    n = sum(len(o) for o in outs)
    print(json.dumps(dict(pre_streams=ps, pinned_output=bool(pin), mel_in_loop=bool(mil), mean_pass_ms=round(sum(tot[1:]) / len(tot[1:]), 2), pass_ms=round(best, 2), loop_ms=round(loop, 2), outside_loop_ms=round(best - loop, 2),
                          samples_per_s=round(n / best * 1e3), launches=m._loop_engine().last_run_info()['launches'], kernel=m._loop_engine().last_run_info()['kernel'])), flush=True)
