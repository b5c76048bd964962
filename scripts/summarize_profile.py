#!/usr/bin/env python
"""Condense a scripts/gpu_profile.sh run (gpurun_out/prof_<tag>_*) into profiles/<tag>_*.csv|md (tracked).

    python scripts/summarize_profile.py r01a
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, 'gpurun_out')
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)
OURS = ('wrnn',)

# 1) kernel-trace --stats summary: keep the top rows verbatim (names shortened)
stats = os.path.join(src, f'prof_{tag}_stats', 'stats_kernel_stats.csv')
rows = list(csv.reader(open(stats)))
with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(rows[0])
    for r in rows[1:13]:
        r[0] = r[0][:96]
        w.writerow(r)

# 2) PMC passes: per (kernel, counter) mean over dispatches
pmc = collections.OrderedDict()
meta = {}
for d in sorted(glob.glob(os.path.join(src, f'prof_{tag}_pmc_*', 'pmc_counter_collection.csv'))):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if any(o in r['Kernel_Name'] for o in OURS):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            acc[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
            meta[k] = dict(vgpr=r['VGPR_Count'], agpr=r['Accum_VGPR_Count'], sgpr=r['SGPR_Count'],
                           lds=r['LDS_Block_Size'], grid=r['Grid_Size'], wg=r['Workgroup_Size'])
    for (k, c), v in acc.items():
        pmc.setdefault(k, {})[c] = (sum(v) / len(v), len(v))
with open(os.path.join(dst, f'{tag}_pmc.csv'), 'w', newline='') as f:
    w = csv.writer(f)
    w.writerow(['kernel', 'counter', 'mean_per_dispatch', 'dispatches'])
    for k, cs in pmc.items():
        for c, (m, n) in cs.items():
            w.writerow([k, c, f'{m:.6g}', n])

# 3) derived figures for the loop kernel
lines = [f'# rocprofv3 summary `{tag}`', '', 'Source: `scripts/gpu_profile.sh` on one MI355X (separate `--pmc` passes); '
         'raw CSVs condensed by `scripts/summarize_profile.py`.', '']
bench = os.path.join(src, f'bench_{tag}.log')
bj = None
if os.path.exists(bench):
    for ln in open(bench):
        if ln.startswith('{'):
            bj = json.loads(ln)
            lines += ['## bench line', '', '```json', json.dumps(bj, indent=1), '```', '']
# fabric-side traffic of the loop kernel per PASS (= mean per dispatch x launches per pass) -> profiles/traffic_latest.json, read
# back by bench.py's roofline.traffic only when kernel, geometry, mode AND the source hash all match the run it is asked about
if bj is not None:
    sys.path.insert(0, ROOT)
    import bench as _bench
    kname = bj['config'].get('kernel', '')
    launches = int(bj['config'].get('launches_per_pass', 1))
    for k, cs in pmc.items():
        if kname and kname in k and 'FETCH_SIZE' in cs and 'WRITE_SIZE' in cs and 'segments_per_gpu' in bj['config']:
            # FETCH_SIZE / WRITE_SIZE are in KB.  The exchange loads are 16-B-per-lane sc1 buffer loads of 1 KB per wave-instruction,
            # i.e. the wide coalesced pattern the guide calibrates as under-counted 2x on gfx950: both the raw and the doubled
            # fetch figure are kept; bytes_per_pass uses the corrected one
            fetch, write = cs['FETCH_SIZE'][0] * 1024 * launches, cs['WRITE_SIZE'][0] * 1024 * launches
            tr = dict(tag=tag, kernel=kname, kernel_instance=k, mode=bj['config'].get('mode', 'MOL'), segments=bj['config']['segments_per_gpu'],
                      T=bj['config']['steps_per_segment'], launches_per_pass=launches, source_sha16=_bench.source_sha16(),
                      bytes_per_pass=int(2 * fetch + write), fetch_bytes_raw=int(fetch), fetch_bytes_corrected=int(2 * fetch), write_bytes=int(write),
                      note='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per dispatch x launches per pass; fabric-side '
                           '(L2 <-> Infinity Fabric / MALL) bytes = the inter-CU activation exchange + conditioning slabs; FETCH_SIZE doubled per '
                           'MI355X_MICROARCH.md (gfx950 reports 1/2 of wide coalesced reads)')
            # (the default line's figure is what bench.py reads back; a --prune run keeps its own file)
            json.dump(tr, open(os.path.join(dst, 'traffic_config5.json' if 'BASELINE config 5' in bj['config'].get('workload', '') else 'traffic_latest.json'), 'w'), indent=1)
for k, cs in pmc.items():
    g = lambda c: cs.get(c, (None,))[0]
    lines += [f'## {k}', '', f'launch geometry / registers: {meta[k]}', '']
    if g('FETCH_SIZE') is not None and g('WRITE_SIZE') is not None:
        nd = cs['FETCH_SIZE'][1]
        lines.append(f'* fabric-side traffic per dispatch (mean of {nd}): FETCH_SIZE {g("FETCH_SIZE") * 1024 / 1e9:.3f} GB raw (x2 on gfx950 for wide '
                     f'coalesced reads = {2 * g("FETCH_SIZE") * 1024 / 1e9:.3f} GB), WRITE_SIZE {g("WRITE_SIZE") * 1024 / 1e9:.3f} GB')
    if g('SQ_LDS_BANK_CONFLICT') is not None and g('SQ_LDS_IDX_ACTIVE'):
        lines.append(f'* LDS bank-conflict rate SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = '
                     f'{g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"):.4f}')
    if g('SQ_WAVE_CYCLES'):
        wc = g('SQ_WAVE_CYCLES')
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'):
            if g(c) is not None:
                lines.append(f'* {c} / SQ_WAVE_CYCLES = {g(c) / wc:.3f}')
        if g('SQ_VALU_MFMA_BUSY_CYCLES') is not None:
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, SQ_WAVE_CYCLES quad-cycles per WAVE: with n waves per SIMD the SIMD-time
            # is 4 * SQ_WAVE_CYCLES / n (grid threads / 64 lanes / 1024 SIMDs of the chip)
            wps = max(1.0, float(meta[k]['grid']) / 64.0 / 1024.0)
            lines.append(f'* MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_WAVE_CYCLES / {wps:g} waves per SIMD) = '
                         f'{g("SQ_VALU_MFMA_BUSY_CYCLES") * wps / (4 * wc):.3f}')
    if g('TCC_HIT_sum') is not None:
        lines.append(f'* L2 hit rate = {g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")):.4f}')
    lines.append('')
open(os.path.join(dst, f'{tag}_summary.md'), 'w').write('\n'.join(lines))
print('\n'.join(lines))
