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
ap.add_argument('--prune-linear', action='store_true', help='... and fc1 / fc2 (the notebook prunes the Linear layers too): the gathered fc stages')
ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'probe.json'))
ap.add_argument('--B', default='12,24,64,128,192,256,512')
ap.add_argument('--variants', default='auto,g1,g2,g4,g8')
ap.add_argument('--so', default=None, help='A/B builds: load this libwavernn_amd*.so instead of the in-tree one')
args = ap.parse_args()

if args.so:
    from wavernn_amd import _lib as _L
    _L.SO_PATH = os.path.abspath(args.so)
dev = torch.device('cuda', 0)
mode, T, hop = args.mode, args.T, 275
sd = random_state_dict(0, mode=mode)
if args.prune > 0:
    from wavernn_amd.prune import block_prune_state_dict
    sd, _ = block_prune_state_dict(sd, args.prune, (16, 1), linear=args.prune_linear)
eng = LoopEngine(sd, mode, device=dev)
rs = np.random.RandomState(3)
#: name -> wrnn_options (LoopEngine.run keyword arguments)
VARS = {'auto': dict(algo='auto'), 'stream': dict(algo='stream'), 'g1': dict(algo='loop', depth=1), 'g2': dict(algo='loop', depth=2),
        'g3': dict(algo='loop', depth=3), 'g4': dict(algo='loop', depth=4), 'g6': dict(algo='loop', depth=6), 'g8': dict(algo='loop', depth=8),
        'c1g1': dict(algo='loop', clusters=1, depth=1), 'c2g1': dict(algo='loop', clusters=2, depth=1), 'c1g2': dict(algo='loop', clusters=1, depth=2),
        'nola': dict(algo='loop', tuning=1), 'fence': dict(algo='loop', tuning=2), 'nola-fence': dict(algo='loop', tuning=3),
        'c1': dict(algo='chain', depth=1), 'c2': dict(algo='chain', depth=2), 'c3': dict(algo='chain', depth=3), 'c4': dict(algo='chain', depth=4),
        's1': dict(algo='sparse'), 'swt': dict(algo='sparse', tuning=256),       # wrnn_sparse_kernel (needs --prune); swt: every layer written through
        # wrnn_duo_kernel (round 4): tuning bit 0 = loads first, bit 1 = publish first (default: by depth), bit 8 = every layer written through
        # (no XCD-local plain stores), bit 2 = ring re-filled before every launch
        'd1': dict(algo='duo', depth=1), 'd2': dict(algo='duo', depth=2), 'd3': dict(algo='duo', depth=3), 'd4': dict(algo='duo', depth=4),
        **{f'o{d}': dict(algo='octo', depth=d) for d in range(1, 9)}, 'oauto': dict(algo='octo'),       # wrnn_octo_kernel (round 6)
        'd5': dict(algo='duo', depth=5), 'd6': dict(algo='duo', depth=6), 'd8': dict(algo='duo', depth=8), 'dauto': dict(algo='duo'),
        'd2lf': dict(algo='duo', depth=2, tuning=1), 'd3lf': dict(algo='duo', depth=3, tuning=1), 'd4lf': dict(algo='duo', depth=4, tuning=1),
        'd6lf': dict(algo='duo', depth=6, tuning=1), 'd8lf': dict(algo='duo', depth=8, tuning=1),
        'd2pf': dict(algo='duo', depth=2, tuning=2), 'd4pf': dict(algo='duo', depth=4, tuning=2), 'd6pf': dict(algo='duo', depth=6, tuning=2),
        'd8pf': dict(algo='duo', depth=8, tuning=2),
        **{f'd{d}{o}': dict(algo='duo', depth=d, tuning={'lf': 1, 'pf': 2}[o]) for d in (1, 2, 3, 4, 5, 6, 8) for o in ('lf', 'pf')},
        'd4wt': dict(algo='duo', depth=4, tuning=256), 'd8wt': dict(algo='duo', depth=8, tuning=256), 'd4lfwt': dict(algo='duo', depth=4, tuning=257),
        'd8lfwt': dict(algo='duo', depth=8, tuning=257), 'd1wt': dict(algo='duo', depth=1, tuning=256),
        'g2ns': dict(algo='loop', depth=2, tuning=16), 'g4ns': dict(algo='loop', depth=4, tuning=16), 'g8ns': dict(algo='loop', depth=8, tuning=16),
        'g1nf': dict(algo='loop', depth=1, tuning=4), 'g2nf': dict(algo='loop', depth=2, tuning=4), 'g4nf': dict(algo='loop', depth=4, tuning=4),
        'd3o': dict(algo='duo', depth=3, tuning=128), 'd4o': dict(algo='duo', depth=4, tuning=128), 'd8o': dict(algo='duo', depth=8, tuning=128), 'd6o': dict(algo='duo', depth=6, tuning=128),
        'g1lf': dict(algo='loop', depth=1, tuning=32), 'g2lf': dict(algo='loop', depth=2, tuning=32), 'g3pf': dict(algo='loop', depth=3, tuning=64),
        'g4pf': dict(algo='loop', depth=4, tuning=64), 'g1x': dict(algo='loop', depth=1, tuning=128), 'g1lfx': dict(algo='loop', depth=1, tuning=160),
        'g2na': dict(algo='loop', depth=2, tuning=8), 'g4na': dict(algo='loop', depth=4, tuning=8), 'g4nfna': dict(algo='loop', depth=4, tuning=12)}
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
        opts = dict(VARS[v.split('+')[0]])
        if '+' in v:                                                  # 'd4+0x10AA1000': the variant with this wrnn_options.tuning word
            opts['tuning'] = int(v.split('+')[1], 0)
        try:
            depth = opts.get('depth', 0)
            if depth > 1 and B <= 64 * (depth - 1) and opts['algo'] in ('loop', 'duo', 'chain', 'octo'):
                continue                                              # the extra slots would stay empty: same run as a shallower depth
            out = eng.run(mels_up, aux, B, T, stride, noise, hop, **opts)
            out = eng.run(mels_up, aux, B, T, stride, noise, hop, **opts)
            ms = eng.last_loop_ms()
            o = out.cpu().numpy()
            if ref is None:
                ref = o
            err = float(np.abs(o - ref).max())
            info = eng.last_run_info()
            row = dict(variant=v, B=B, T=T, ms=round(ms, 3), us_per_round_step=round(ms * 1e3 / (T * max(info['rounds'], 1)), 3),
                       seg_steps_per_s=round(B * T / (ms * 1e-3)), info=info, max_dev_vs_first=err)
        except Exception as e:
            row = dict(variant=v, B=B, error=str(e)[:200])
        rows.append(row)
        print(json.dumps(row), flush=True)
json.dump(rows, open(args.out, 'w'), indent=0)
