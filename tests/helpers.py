import ast, os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
CASES = ['raw_unbatched_24f', 'raw_batched_60f', 'mol_unbatched_24f', 'mol_batched_100f', 'mol_batched_ragged_53f']

#: MoL tolerance (max abs error on samples in [-1,1]) -- BASELINE.md "budget 1e-5"; observed <= 4e-7 CPU-vs-CPU.
MOL_TOL = 1e-5


def load_case(name):
    g = np.load(os.path.join(GOLDEN, name + '.npz'))
    cfg = ast.literal_eval(str(g['config']))
    return cfg, g
