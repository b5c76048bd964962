#!/usr/bin/env python
"""wrnn_duo_kernel: one engine, a list of wrnn_options.tuning values x depths, loop-kernel time per step and the deviation from tuning 0:
    python scripts/gpu_tuning_sweep.py --depths 2,4,8 --tunings 0,1024,2048,16 [--so x.so]"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--depths', default='2,4,8'); ap.add_argument('--tunings', default='0'); ap.add_argument('--T', type=int, default=1500)
ap.add_argument('--reps', type=int, default=2); ap.add_argument('--mode', default='MOL'); ap.add_argument('--algo', default='duo')
ap.add_argument('--so', default=None); ap.add_argument('--out', default=None); ap.add_argument('--stride', type=int, default=64, help='distance of two segments in the conditioning (the corpus: 11550)')
a = ap.parse_args()
if a.so:
    from wavernn_amd import _lib as _L
    _L.SO_PATH = os.path.abspath(a.so)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict
dev = torch.device('cuda', 0)
eng = LoopEngine(random_state_dict(0, mode=a.mode), a.mode, device=dev)
rs = np.random.RandomState(3)
hop, stride = 275, a.stride
rows = []
for d in [int(x) for x in a.depths.split(',')]:
    B = 64 * d
    L = (B * stride + a.T + hop - 1) // hop * hop
    mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
    aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
    noise = torch.empty(a.T, 11 * B, device=dev).uniform_(1e-5, 1 - 1e-5) if a.mode == 'MOL' else torch.empty(a.T, B, 512, device=dev).exponential_(1)
    ref = None
    for tun in [int(x, 0) for x in a.tunings.split(',')]:
        best = 1e9
        for _ in range(a.reps):
            out = eng.run(mels_up, aux, B, a.T, stride, noise, hop, algo=a.algo, depth=d, tuning=tun)
            best = min(best, eng.last_loop_ms())
        o = out.float().cpu().numpy() if torch.is_tensor(out) else np.asarray(out)
        if ref is None:
            ref = o
        r = dict(depth=d, B=B, tuning=tun, us_per_step=round(best * 1e3 / a.T, 3), dev_vs_first=float(np.abs(o - ref).max()), kernel=eng.last_loop_kernel())
        rows.append(r)
        print(json.dumps(r), flush=True)
if a.out:
    json.dump(rows, open(a.out, 'w'), indent=1)
