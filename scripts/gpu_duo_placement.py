#!/usr/bin/env python
"""Where do the 512 workgroups of wrnn_duo_kernel land?  Reads the kernel's placement hook (HW_ID / XCC_ID per block through
wrnn_options.phase_clocks) and prints, per (XCC, SE, SH, CU), the roles of the workgroups sharing that CU."""
import collections, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wavernn_amd.engine import LoopEngine
from wavernn_amd.synthetic import random_state_dict

dev = torch.device('cuda', 0)
sd = random_state_dict(0, mode='MOL')
eng = LoopEngine(sd, 'MOL', device=dev)
rs = np.random.RandomState(3)
B, T, hop, stride = 256, 64, 275, 64
L = (B * stride + T + hop - 1) // hop * hop
mels_up = torch.from_numpy(rs.uniform(0, 1, (L, 80)).astype(np.float32)).to(dev)
aux = torch.from_numpy(rs.uniform(-1, 1, (L // hop, 128)).astype(np.float32)).to(dev)
noise = torch.empty(T, 11 * B, device=dev).uniform_(1e-5, 1 - 1e-5)
pc = torch.zeros(256, 32, dtype=torch.int64, device=dev)
eng.run(mels_up, aux, B, T, stride, noise, hop, algo='duo', depth=4, phase_clocks=pc, tuning=64)
v = pc.cpu().numpy().reshape(-1)[:512].astype(np.uint64)
hw = (v & np.uint64(0xFFFFFFFF)).astype(np.uint32)
xcc = ((v >> np.uint64(32)) & np.uint64(0xFF)).astype(np.uint32)
meta = (v >> np.uint64(40)).astype(np.uint32)
role, J, cl = meta & 3, (meta >> 2) & 63, (meta >> 8) & 15       # role: bit 0 = rnn2, bit 1 = hh
vary = np.bitwise_or.reduce(hw) ^ np.bitwise_and.reduce(hw)
print('HW_ID bits that vary: 0x%08x' % vary)
key = hw & np.uint32(0x0000FF00)                        # cu_id [11:8], sh_id [12], se_id [15:13]
cus = collections.defaultdict(list)
for b in range(512):
    cus[(int(xcc[b]) & 15, int(key[b]) >> 8)].append((int(role[b]), int(J[b]), int(cl[b]), b))
hist = collections.Counter()
for k, lst in sorted(cus.items()):
    hist[tuple(sorted(r for r, _, _, _ in lst))] += 1
xl = collections.defaultdict(set)
for b in range(512):
    xl[(int(cl[b]), int(role[b]) & 1)].add(int(xcc[b]) & 15)
print('XCCs hosting (cluster, layer):', {k: sorted(v) for k, v in sorted(xl.items())})
print('CUs seen:', len(cus), ' roles sharing a CU -> number of CUs:', dict(hist))
for k, lst in list(sorted(cus.items()))[:6]:
    print(k, lst)
json.dump({'vary': int(vary), 'hist': {str(k): v for k, v in hist.items()}, 'n_cus': len(cus)}, open(os.path.join(ROOT, 'gpurun_out', 'duo_placement.json'), 'w'))
