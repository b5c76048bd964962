#!/usr/bin/env python
"""Loop-kernel timing sweep on one MI355X: every kernel split x segment count, short T (timing only, no parity).

    python scripts/gpu_perf_probe.py [--T 1500] [--out gpurun_out/probe.json]
"""
import argparse, json, os, sys, time
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--T', type=int, default=1500)
ap.add_argument('--mode', default='MOL')
ap.add_argument('--prune', type=float, default=0.0, help='block-prune the GRU matrices to this sparsity (config 5)')
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'probe.json'))
ap.add_argument('--B', default='12,64,120,128,180,192,256,360')
ap.add_argument('--variants', default='u8,p2,p3,auto')
args = ap.parse_args()

dev = torch.device('cuda', 0)
mode, T, hop = args.mode, args.T, 275
sd = random_state_dict(0, mode=mode)
if args.prune > 0:
    from wavernn_amd.prune import block_prune_state_dict
    sd, _ = block_prune_state_dict(sd, args.prune, (16, 1))
eng = LoopEngine(sd, mode, device=dev)
rs = np.random.RandomState(3)
VARS = {'persist': ('persist', {}), 'u2': ('cluster', {'WRNN_CLUSTER_U': '2'}), 'u4': ('cluster', {'WRNN_CLUSTER_U': '4'}),
        'u8': ('cluster', {'WRNN_CLUSTER_U': '8', 'WRNN_CLUSTER_NL': '8'}),
        'u8nl16': ('cluster', {'WRNN_CLUSTER_U': '8', 'WRNN_CLUSTER_NL': '16'}), 'auto': ('auto', {}),
        'p1': ('pipe', {'WRNN_PIPE_G': '1'}), 'p2': ('pipe', {'WRNN_PIPE_G': '2'}), 'p3': ('pipe', {'WRNN_PIPE_G': '3'}),
        's1': ('sparse', {'WRNN_SPARSE_G': '1'}), 's2': ('sparse', {'WRNN_SPARSE_G': '2'}),
        'p3nl8': ('pipe', {'WRNN_PIPE_G': '3', 'WRNN_PIPE_NL': '8'}), 'p2nl8': ('pipe', {'WRNN_PIPE_G': '2', 'WRNN_PIPE_NL': '8'})}
rows = []
for B in [int(x) for x in args.B.split(',')]:
    stride = 64
    L = (B * stride + T + hop - 1) // hop * hop
    N = L // hop
    mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
    aux = torch.from_numpy(rs.uniform(-1, 1, (N, 128)).astype(np.float32)).to(dev)
    if mode == 'MOL':
        noise = torch.empty(T, 11 * B, device=dev).uniform_(1e-5, 1 - 1e-5)
    else:
        noise = torch.empty(T, B, 512, device=dev).exponential_(1)
    ref = None
    for v in args.variants.split(','):
        algo, env = VARS[v]
        if mode == 'RAW' and (v.startswith('u8') or v.startswith('p')) and v != 'persist':
            continue
        if v == 'persist' and B > 64:
            continue
        for k in ('WRNN_CLUSTER_U', 'WRNN_CLUSTER_NL', 'WRNN_PIPE_G', 'WRNN_PIPE_NL', 'WRNN_SPARSE_G'):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            out = eng.run(mels_up, aux, B, T, stride, noise, hop, algo=algo)
            out = eng.run(mels_up, aux, B, T, stride, noise, hop, algo=algo)
            ms = eng.last_loop_ms()
            o = out.cpu().numpy()
            if ref is None:
                ref = o
            err = float(np.abs(o - ref).max())
            u, ncl, gdepth = eng.last_loop_split()
            rows_g = 15 if (gdepth >= 3 and algo != 'sparse') else 16
            groups = -(-B // rows_g)
            rounds = groups if algo == 'persist' else max(1, -(-groups // max(ncl * max(gdepth, 1), 1)))
            row = dict(variant=v, B=B, T=T, ms=round(ms, 3), us_per_round_step=round(ms * 1e3 / (T * rounds), 3),
                       seg_steps_per_s=round(B * T / (ms * 1e-3)), split=[u, ncl, gdepth], max_dev_vs_first=err)
        except Exception as e:
            row = dict(variant=v, B=B, error=str(e)[:200])
        rows.append(row)
        print(json.dumps(row), flush=True)
json.dump(rows, open(args.out, 'w'), indent=0)
