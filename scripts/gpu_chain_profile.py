#!/usr/bin/env python
"""Per-segment shader clocks of wrnn_chain_kernel (wrnn_options.phase_clocks) + its step time next to wrnn_duo_kernel's at the same batch.
    python scripts/gpu_chain_profile.py [--B 12 --T 2000]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=12); ap.add_argument('--T', type=int, default=2000); ap.add_argument('--tuning', type=int, default=0)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'chain_phase_clocks.json'))
a = ap.parse_args()
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode='MOL'), 'MOL', device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5)
res = {'B': a.B, 'T': a.T}
outs = {}
for algo in ('duo', 'chain'):
    for _ in range(2):
        outs[algo] = eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo=algo, tuning=a.tuning)
    res[algo + '_us_per_step'] = eng.last_loop_ms() * 1e3 / a.T
    res[algo + '_info'] = eng.last_run_info()
res['max_abs_chain_minus_duo'] = float((outs['chain'] - outs['duo']).abs().max())
pc = torch.zeros(256, 32, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='chain', tuning=a.tuning, phase_clocks=pc)
v = pc.cpu().numpy().astype(np.float64)
blocks = np.arange(256)
layer, J = (blocks % 8) & 1, blocks // 8
names1 = ['wait x(t-1)', 'cell+publish', 'wait h1', 'gh stage', 'wait cI', 'Wih.cI stage', 'form cI', 'wait y2', 'fc3+sample']
names2 = ['wait x1', 'gates+cell+publish', 'wait x2', 'fc1', 'wait y1', 'fc2', 'wait h2', 'gh stage']
us = res['chain_us_per_step']
for name, sel, names in (('rnn1 sampler', (layer == 0) & (J == 0), names1), ('rnn1 others', (layer == 0) & (J > 0), names1), ('rnn2', layer == 1, names2)):
    rows = v[sel]
    rows = rows[rows[:, 15] > 0]
    if not len(rows):
        continue
    per = (rows[:, :len(names)] / rows[:, 15:16]).mean(axis=0)
    tot = per.sum()
    res[name] = {n: round(float(x), 1) for n, x in zip(names, per)}
    print(f'{name:13s} ' + ' | '.join(f'{n} {x / tot * us:5.2f}' for n, x in zip(names, per)) + f' | (us of a {us:.2f} us step)')
print(json.dumps({k: res[k] for k in ('duo_us_per_step', 'chain_us_per_step', 'max_abs_chain_minus_duo')}))
json.dump(res, open(a.out, 'w'), indent=1)
