#!/usr/bin/env python
"""Class-index flip rate of the RAW ('bits', 9-bit mu-law) loop against the C oracle, free-running, at the benchmarked geometry
(round-4 verdict, "What's weak" 2): batches of 16 x 641-frame utterances = 256 segments x 12,100 steps through `generate_corpus`
with parity noise (per-utterance MT19937 streams), for three forms of the loop:

    duo_mel_in_loop   wrnn_duo_kernel as shipped: x_{t-1} term after the matrix product (W_ih.cI + x u1), last up-sampling stage formed
                      in the loop from three rows (wrnn_options.mel_stage = 1)
    duo_mel_full      wrnn_duo_kernel fed the materialised [L, 80] mel of the pre-loop kernels (23-tap sum)
    loop_ref_order    wrnn_loop_kernel: x_{t-1} inside the operand (xi = cI + w0 x, the reference's order), materialised mel

A run is free-running, so a segment is "exposed" until its first differing sample: flips = segments that part ways, exposure = the
segment-steps before that; rate = flips / exposure.  The oracle outputs come from tests/_cache (scripts/make_oracle_cache.py) or are
computed on the spot.  Every flip is recorded with the GPU's class, the oracle's and -- where tests/golden holds the reference's own run of
that utterance (raw_flip_u*.npz, scripts/make_golden.py) -- the REFERENCE's.  TEST / MEASUREMENT infrastructure (imports oracle/ through tests/helpers.py).

    python scripts/gpu_raw_flips.py [--batches 4] [--json gpurun_out/raw_flips.json]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batches', type=int, default=4)
    ap.add_argument('--json', default=None)
    ap.add_argument('--variants', default='duo_mel_in_loop,duo_mel_full,loop_ref_order')
    args = ap.parse_args()
    from helpers import oracle_utterance
    from wavernn_amd.batch import generate_corpus
    from wavernn_amd.model import WaveRNN
    from wavernn_amd.synthetic import random_state_dict, random_mel, SHIPPED
    dev = torch.device('cuda', 0)
    sd = random_state_dict(0, mode='RAW')
    model = WaveRNN(**SHIPPED, mode='RAW')
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()}, strict=True)
    model = model.to(dev).eval()
    variants = {'duo_mel_in_loop': dict(mel_in_loop=True, loop_algo='auto'), 'duo_mel_full': dict(mel_in_loop=False, loop_algo='auto'),
                'loop_ref_order': dict(mel_in_loop=False, loop_algo='loop')}
    res = {v: dict(flips=0, exposure=0, segments=0, first=[]) for v in args.variants.split(',')}
    to_cls = lambda x: int(round((float(x) + 1.0) * 511.0 / 2.0))
    for k in range(args.batches):
        us = list(range(16 * k, 16 * k + 16))
        mels = [torch.from_numpy(random_mel(1234 + u, 641)).unsqueeze(0) for u in us]
        refs = [oracle_utterance('RAW', 0, 0.0, 1234 + u, 77 + u, 641, want_cond=False, sd=sd, nthreads=16)['ref'] for u in us]
        for v in res:
            for a, b in variants[v].items():
                setattr(model, a, b)
            t0 = time.time()
            segs, plan = generate_corpus(model, mels, 11000, 550, True, [77 + u for u in us], return_segments=True)
            info = model._loop_engine().last_run_info()
            for i, ref in enumerate(refs):
                got = segs[plan.first[i]:plan.first[i] + plan.folds[i]].astype(np.float32)
                for s in range(got.shape[0]):
                    d = np.flatnonzero(got[s] != ref[s])
                    res[v]['segments'] += 1
                    res[v]['exposure'] += int(d[0]) if d.size else plan.T
                    if d.size:
                        res[v]['flips'] += 1
                        t = int(d[0])
                        rec = dict(utterance=us[i], segment=s, step=t, gpu_class=to_cls(got[s, t]), oracle_class=to_cls(ref[s, t]), reference_class=None)
                        fx = os.path.join(ROOT, 'tests', 'golden', f'raw_flip_u{us[i]}.npz')        # the REFERENCE's own run, where one is held
                        if os.path.exists(fx):
                            rec['reference_class'] = int(np.load(fx)['cls'][s, t])
                        res[v]['first'].append(rec)
            print(f'batch {k} {v}: {info["kernel"]} depth {info["depth"]} launches {info["launches"]}, flips so far {res[v]["flips"]} in '
                  f'{res[v]["exposure"]} segment-steps ({time.time() - t0:.1f} s)', flush=True)
    for v in res:
        res[v]['rate_per_segment_step'] = res[v]['flips'] / max(1, res[v]['exposure'])
    print(json.dumps(res))
    if args.json:
        os.makedirs(os.path.dirname(args.json), exist_ok=True)
        json.dump(res, open(args.json, 'w'), indent=1)


if __name__ == '__main__':
    main()
