#!/usr/bin/env python
"""One loop-kernel configuration, timed (for rocprofv3 --pmc / --stats passes): python scripts/gpu_probe_one.py --algo duo --depth 8 --B 512"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
ap = argparse.ArgumentParser()
ap.add_argument('--algo', default='duo'); ap.add_argument('--depth', type=int, default=8); ap.add_argument('--B', type=int, default=512)
ap.add_argument('--T', type=int, default=600); ap.add_argument('--reps', type=int, default=2); ap.add_argument('--mode', default='MOL')
ap.add_argument('--tuning', type=int, default=0)
ap.add_argument('--so', default=None, help='A/B builds: load this libwavernn_amd*.so instead of the in-tree one')
a = ap.parse_args()
if a.so:
    from wavernn_amd import _lib as _L
    _L.SO_PATH = os.path.abspath(a.so)
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode=a.mode), a.mode, device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, 64
L = (a.B * stride + a.T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(a.T, 11 * a.B, device=dev).uniform_(1e-5, 1 - 1e-5) if a.mode == 'MOL' else torch.empty(a.T, a.B, 512, device=dev).exponential_(1)
for _ in range(a.reps):
    eng.run(mels_up, aux, a.B, a.T, stride, noise, hop, algo=a.algo, depth=a.depth, tuning=a.tuning)
ms = eng.last_loop_ms()
print(json.dumps(dict(algo=a.algo, depth=a.depth, B=a.B, T=a.T, ms=round(ms, 3), us_per_step=round(ms * 1e3 / a.T, 3), seg_steps_per_s=round(a.B * a.T / (ms * 1e-3)), info=eng.last_run_info())))
