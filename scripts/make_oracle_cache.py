#!/usr/bin/env python
"""Pre-compute the C oracle's loop outputs of the full-size GPU parity cases (tests/helpers.py::oracle_utterance) into tests/_cache/.

TEST INFRASTRUCTURE.  The cache is git-ignored but travels to the GPU box with the gpurun snapshot, so a `-m gpu` run there compares
against these arrays instead of spending GPU-box minutes in the CPU oracle; a missing entry is computed on the fly (what the driver's
round-end run does).  Every entry is keyed by everything that determines it (mode, weight seed, pruning, mel seed, noise seed,
frames, fold geometry) and by a hash of the oracle's C source.

    python scripts/make_oracle_cache.py raw64        # 64 RAW utterances (flip-rate measurement + the benchmarked RAW geometry)
    python scripts/make_oracle_cache.py mol16 sparse16
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import oracle_utterance          # noqa: E402

SETS = {
    'raw64': [dict(mode='RAW', wseed=0, prune=0.0, mel_seed=1234 + u, noise_seed=77 + u, frames=641) for u in range(64)],
    'mol16': [dict(mode='MOL', wseed=0, prune=0.0, mel_seed=1234 + u, noise_seed=77 + u, frames=641) for u in range(16)],
    'sparse16': [dict(mode='MOL', wseed=0, prune=0.95, mel_seed=1234 + u, noise_seed=77 + u, frames=641) for u in range(16)],
    'sparse16L': [dict(mode='MOL', wseed=0, prune=0.95, linear=True, mel_seed=1234 + u, noise_seed=77 + u, frames=641) for u in range(16)],
}

if __name__ == '__main__':
    for name in sys.argv[1:]:
        for k in SETS[name]:
            t0 = time.time()
            ref = oracle_utterance(**k, nthreads=int(os.environ.get('ORACLE_THREADS', '8')), want_cond=False)['ref']
            print(name, k, ref.shape, f'{time.time() - t0:.1f} s', flush=True)
