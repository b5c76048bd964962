#!/usr/bin/env python
"""Per-segment shader clocks of wrnn_sparse_kernel (wrnn_options.phase_clocks): where a step of a cluster goes.
    python scripts/gpu_sparse_profile.py [--B 256 --T 1200 --tuning 0]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.prune import block_prune_state_dict
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=256); ap.add_argument('--T', type=int, default=1200); ap.add_argument('--tuning', type=int, default=0)
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'sparse_phase_clocks.json')); ap.add_argument('--linear', action='store_true')
a = ap.parse_args()
dev = torch.device('cuda', 0)
sd, _ = block_prune_state_dict(random_state_dict(0, mode='MOL'), 0.95, (16, 1), linear=a.linear)
eng = LoopEngine(sd, 'MOL', device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5)
for _ in range(2):
    eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='sparse', tuning=a.tuning)
plain = eng.last_loop_ms()
pc = torch.zeros(256, 32, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='sparse', tuning=a.tuning, phase_clocks=pc)
prof_ms = eng.last_loop_ms()
v = pc.cpu().numpy().astype(np.float64)            # [block][16]
# block b: XCD b % 8, q = b / 8: cluster (b % 8) + 8 (q >> 4), CU q & 15: 0-7 rnn1, 8-15 rnn2 (8 = the sampling workgroup)
blocks = np.arange(256)
cu = (blocks // 8) & 15
names1 = ['wait x(t-1)', 'cell+publish', 'wait h1', 'gh tiles', 'wait x2', 'fc1', 'wait y1', 'fc2', 'wait cI', 'Wih.cI tiles', 'form cI']
names2 = ['wait x1', 'gates+cell+publish', 'wait x2', 'fc1', 'wait h2', 'gh tiles', 'wait y1', 'fc2', 'wait y2', 'fc3+sample']
res = {'plain_ms': plain, 'profiled_ms': prof_ms, 'us_per_step': plain * 1e3 / a.T, 'B': a.B, 'T': a.T, 'tuning': a.tuning}
for name, sel, names in (('rnn1', cu < 8, names1), ('rnn2 sampler', cu == 8, names2), ('rnn2 others', cu > 8, names2)):
    rows = v[sel]
    rows = rows[rows[:, 15] > 0]
    if not len(rows):
        continue
    per = (rows[:, :len(names)] / rows[:, 15:16]).mean(axis=0)
    tot = per.sum()
    us = plain * 1e3 / a.T
    res[name] = {n: round(float(x), 1) for n, x in zip(names, per)}
    res[name]['cycles per step'] = round(float(tot), 1)
    print(f'{name:13s} ' + ' | '.join(f'{n} {x / tot * us:5.2f}' for n, x in zip(names, per)) + f' | (us of a {us:.2f} us step; {tot:.0f} clocks per step)')
print(json.dumps({k: res[k] for k in ('plain_ms', 'profiled_ms', 'us_per_step')}))
json.dump(res, open(a.out, 'w'), indent=1)
