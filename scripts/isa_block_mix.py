#!/usr/bin/env python3
"""Static view of a HIP kernel's gfx950 ISA: per kernel the instruction mix (MFMA / other VALU / SALU / branches / LDS / VMEM / waits),
code size, and -- with --blocks -- the mix of every basic block of >= --min instructions.  No GPU needed (hipcc -S).  This is the
tool behind DESIGN.md section 6's "what the non-MFMA 60 % is made of": a stage of the loop kernel is a handful of blocks between
two 96-MFMA blocks, and the scalar-select / branch clutter of a run-time phase variable shows up as SALU-heavy blocks.

usage: scripts/isa_block_mix.py [wavernn_amd/csrc/wrnn_loop.hip] [--kernel SUBSTR] [--blocks] [--min 12]
"""
import argparse, os, re, subprocess, sys, tempfile

CATS = ('mfma', 'acc', 'valu', 'salu', 'br', 'lds', 'vmem', 'wait', 'bar', 'oth')


def cat(op):
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('v_accvgpr'): return 'acc'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_barrier'): return 'bar'
    if op.startswith(('s_cbranch', 's_branch')): return 'br'
    if op.startswith('s_'): return 'salu'
    if op.startswith('v_'): return 'valu'
    return 'oth'


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('src', nargs='?', default=os.path.join(os.path.dirname(__file__), '..', 'wavernn_amd', 'csrc', 'wrnn_loop.hip'))
    ap.add_argument('--kernel', default='', help='only kernels whose mangled name contains this')
    ap.add_argument('--blocks', action='store_true')
    ap.add_argument('--min', type=int, default=12)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.check_call([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '--offload-arch=gfx950', '-O3', '-std=c++17', '-S',
                               '--cuda-device-only', '-Wno-unused-value', a.src, '-o', out], stderr=subprocess.DEVNULL)
        lines = open(out).read().split('\n')
    kern, blocks, cur, name = {}, [], None, None
    for n, l in enumerate(lines, 1):
        s = l.strip()
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1); kern[name] = {c: 0 for c in CATS}; kern[name]['blocks'] = blocks = []; cur = None
            continue
        if name is None or not s or s.startswith(';'):
            continue
        if s.startswith('.amdhsa_next_free_vgpr'):
            kern[name]['regs'] = s.split()[-1]; name = None
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if m:
            cur = {'label': m.group(1), 'line': n, 'n': 0, **{c: 0 for c in CATS}}; blocks.append(cur)
            continue
        if s.startswith('.'):
            continue
        c = cat(s.split()[0])
        kern[name][c] += 1
        if cur is not None:
            cur[c] += 1; cur['n'] += 1
    for k, v in kern.items():
        if a.kernel not in k:
            continue
        tot = sum(v[c] for c in CATS)
        print(f"{k}: {tot} instructions, vgpr+agpr {v.get('regs', '?')}: " + ' '.join(f'{c}={v[c]}' for c in CATS if v[c]))
        if a.blocks:
            for b in v['blocks']:
                if b['n'] >= a.min or b['mfma'] or b['bar']:
                    print(f"  {b['label']:>12} @{b['line']:<6} {b['n']:4d}  " + ' '.join(f'{c}={b[c]}' for c in CATS if b[c]))


if __name__ == '__main__':
    sys.exit(main())
