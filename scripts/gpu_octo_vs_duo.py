#!/usr/bin/env python
"""wrnn_octo_kernel against wrnn_duo_kernel on the same inputs, several splits: the two kernels run the same arithmetic in the same order, so the
outputs are expected to agree bit for bit; prints the first (segment, step) where they do not.  python scripts/gpu_octo_vs_duo.py [--B 40 --T 400]"""
import argparse, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--B', type=int, default=40); ap.add_argument('--T', type=int, default=400); ap.add_argument('--reps', type=int, default=3); ap.add_argument('--tuning', type=int, default=0); ap.add_argument('--only', default='')
a = ap.parse_args()
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode='MOL'), 'MOL', device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5)
ref = eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='duo').cpu().numpy()
bad = 0
for name, kw in (('octo', {}), ('octo d1', dict(depth=1)), ('octo d2 slabs 97', dict(depth=2, slab_steps=97)), ('octo c1 d3 slabs 160', dict(clusters=1, depth=3, slab_steps=160)),
                 ('octo c1 d1', dict(clusters=1, depth=1)), ('octo c1 d2', dict(clusters=1, depth=2)), ('octo c2 d2', dict(clusters=2, depth=2)), ('octo d1 wt', dict(depth=1, tuning=256)),
                 ('octo d3 slabs 3', dict(depth=3, slab_steps=3))):
    if a.only and a.only not in name:
        continue
    kw = dict(kw); kw['tuning'] = kw.get('tuning', 0) | a.tuning
    for rep in range(a.reps):
        try:
            o = eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo='octo', **kw).cpu().numpy()
        except Exception as e:
            print(f'{name:24s} rep {rep}: {str(e)[:160]}'); bad += 1; break
        d = np.argwhere(o != ref)
        if d.size:
            t_first = d[:, 1].min()
            segs = sorted(set(int(x) for x in d[d[:, 1] == t_first][:, 0]))
            print(f'{name:24s} rep {rep}: {len(d)} values differ; first step {t_first}, segments {segs[:12]}, |diff| there {np.abs(o - ref)[segs[0], t_first]:.3g}; max {np.abs(o - ref).max():.3g}  {eng.last_run_info()}')
            bad += 1
        else:
            print(f'{name:24s} rep {rep}: identical  {eng.last_loop_ms():.2f} ms')
print('FAILED' if bad else 'OK')
