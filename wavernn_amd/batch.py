"""Corpus-level batched generation: many utterances -> one segment table -> shards over ranks -> all-gather.

The reference generates one utterance per `WaveRNN.generate()` call (gen_wavernn.py:26-35, :56-65): each call
folds ITS mel into `num_folds` overlapping segments (models/fatchord_version.py:293-340), runs them as one batch
and cross-fades them back (:342-405).  Folded segments are independent from step 0 (zero initial state,
:194-196; no cross-segment term in :201-241), so the segments of SEVERAL utterances can share one launch, and a
corpus can be cut into contiguous blocks of segments, one block per GPU (BASELINE config
4).  This module is the host logic for that:

  plan_utterances   the segment table (`seg_pos`, `seg_lim`) over the concatenated conditioning
  shard_bounds      contiguous, balanced blocks of segments per rank (no data-path collective inside the loop)
  pack_noise        per-utterance reference noise streams -> the launch's [T, 11*n] / [T, n, C] layout
  generate_corpus   upsample (PyTorch) -> loop (libwavernn_amd.so) on this rank's block -> all_gather of the
                    [n_r, T] audio (RCCL over xGMI when the process group is `nccl`) -> per-utterance unfold

Results do not depend on the number of ranks: every segment's noise is addressed by (utterance seed, step, fold
index) exactly as the single-utterance reference call would draw it.
"""
import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import fold as _fold


@dataclass
class Plan:
    target: int
    overlap: int
    T: int                    # steps per segment = target + 2*overlap
    stride: int               # target + overlap
    lengths: List[int]        # un-folded conditioning length (samples) of every utterance
    offsets: np.ndarray       # sample offset of every utterance in the concatenated conditioning
    folds: np.ndarray         # segments per utterance (num_folds of fold_with_overlap)
    first: np.ndarray         # index of every utterance's first segment
    seg_pos: np.ndarray       # int32 [n_segments]: conditioning position of step 0
    seg_lim: np.ndarray       # int32 [n_segments]: first position that is zero padding (the utterance's end)
    seg_utt: np.ndarray       # int32 [n_segments]: owning utterance

    @property
    def n_segments(self):
        return int(self.seg_pos.shape[0])

    @property
    def total_len(self):
        return int(sum(self.lengths))


def plan_utterances(lengths: Sequence[int], target: int, overlap: int) -> Plan:
    """Segment table for utterances whose un-folded conditioning (lengths[u] samples each) is concatenated."""
    lengths = [int(x) for x in lengths]
    if not lengths or min(lengths) < 1:
        raise ValueError('need at least one non-empty utterance')
    stride, T = target + overlap, target + 2 * overlap
    folds = np.array([_fold.fold_geometry(L, target, overlap)[0] for L in lengths], dtype=np.int64)
    # reference quirk: an input shorter than one window still makes one zero-padded fold (:322-330)
    offsets = np.concatenate([[0], np.cumsum(lengths)[:-1]]).astype(np.int64)
    first = np.concatenate([[0], np.cumsum(folds)[:-1]]).astype(np.int64)
    if offsets[-1] + lengths[-1] + T >= 2 ** 31:
        raise ValueError('concatenated conditioning exceeds int32 positions; split the corpus')
    seg_pos, seg_lim, seg_utt = [], [], []
    for u, (L, nf) in enumerate(zip(lengths, folds)):
        seg_pos.append(offsets[u] + np.arange(nf, dtype=np.int64) * stride)
        seg_lim.append(np.full(nf, offsets[u] + L, dtype=np.int64))
        seg_utt.append(np.full(nf, u, dtype=np.int64))
    return Plan(target, overlap, T, stride, lengths, offsets, folds, first,
                np.concatenate(seg_pos).astype(np.int32), np.concatenate(seg_lim).astype(np.int32),
                np.concatenate(seg_utt).astype(np.int32))


def shard_bounds(n_segments: int, world: int, active: Optional[int] = None):
    """[(lo, hi)] * world: contiguous blocks whose sizes differ by at most one (block r = segments [lo, hi)).  active < world: only the
    first `active` ranks get a block (`choose_ranks`); the others own the empty block (n, n) and only take part in the all-gather."""
    active = world if active is None else max(1, min(int(active), world))
    return [((r * n_segments) // active, ((r + 1) * n_segments) // active) if r < active else (n_segments, n_segments) for r in range(world)]


#: LAST-RESORT step times (us) of the dense MoL loop kernels by groups in flight per cluster (4 clusters x depth x 16 segments per GPU, one MI355X,
#: round 6: wrnn_chain_kernel at depth 1-2, wrnn_duo_kernel from 3 on).  `choose_ranks` plans with a table that is MEASURED: bench.py times the
#: depths on the device it runs on and hands the table over (and records it in profiles/step_us.json, keyed by a hash of the kernel sources, like
#: profiles/traffic_latest.json); without one `step_table()` reads that file and falls back to these numbers -- saying so -- only when the file is
#: missing or was measured on other sources.
DEFAULT_STEP_US_BY_DEPTH = {1: 10.4, 2: 13.8, 3: 19.4, 4: 22.9, 5: 27.4, 6: 31.3, 7: 35.5, 8: 38.9}
_STEP_TABLE = None


def kernel_source_sha16() -> str:
    """Hash of the sources the dense loop kernels are compiled from (what a measured step-time / traffic record belongs to)."""
    import hashlib
    import os
    from . import _lib
    h = hashlib.sha256()
    for name in ('wrnn_abi.hip', 'wrnn_cond.hip', 'wrnn_device.h', 'wrnn_duo.hip', 'wrnn_loop.hip', 'wrnn_ring.h', 'wrnn_tiles.h'):
        h.update(open(os.path.join(_lib.CSRC, name), 'rb').read())
    return h.hexdigest()[:16]


def step_table(path: Optional[str] = None):
    """(table {depth: us per step}, origin string).  The record bench.py wrote for THESE kernel sources, else the built-in numbers."""
    global _STEP_TABLE
    if _STEP_TABLE is None or path is not None:
        import json
        import os
        from . import _lib
        f = path or os.path.join(os.path.dirname(os.path.dirname(_lib.CSRC)), 'profiles', 'step_us.json')
        table, origin = dict(DEFAULT_STEP_US_BY_DEPTH), 'built-in defaults (no profiles/step_us.json)'
        try:
            j = json.load(open(f))
            if j.get('source_sha16') == kernel_source_sha16():
                table, origin = {int(k): float(v) for k, v in j['step_us_by_depth'].items()}, f'{f} (measured on these kernel sources)'
            else:
                origin = f'built-in defaults ({f} was measured on other kernel sources)'
        except (OSError, ValueError, KeyError):
            pass
        if path is not None:
            return table, origin
        _STEP_TABLE = (table, origin)
    return _STEP_TABLE


def estimate_step_us(n_segments: int, table: Optional[dict] = None) -> float:
    """Step time of one GPU that advances `n_segments` segments by one sample (rounds of <= 512 segments at the measured depth)."""
    if n_segments <= 0:
        return 0.0
    table = table or step_table()[0]
    groups = -(-n_segments // 16)
    rounds = -(-groups // 32)
    depth = min(8, max(1, -(-(-(-groups // rounds)) // 4)))
    return rounds * table[depth]


def choose_ranks(n_segments: int, world: int, table: Optional[dict] = None) -> int:
    """How many of `world` GPUs a FIXED corpus of `n_segments` folded segments should be sharded over (strong scaling, BASELINE config 4):
    the wall time of a pass is the step time of the largest block, which depends on the pipeline depth that block fills -- not on the rank
    count as such -- so the smallest rank count that reaches the best estimated wall time is taken (942 segments on 8 GPUs: 118 per GPU =
    depth 2, and 7 GPUs would need depth 3: all 8; on 16 GPUs: 15 suffice).  The ranks left out still take part in the all-gather.
    table: {depth: us per step} measured by the caller (bench.py); default: `step_table()`."""
    est = [estimate_step_us(-(-n_segments // r), table) for r in range(1, world + 1)]
    best = min(est)
    return 1 + next(i for i, e in enumerate(est) if e <= best * 1.01)


def pack_noise(mode: str, plan: Plan, per_utt, lo: int = 0, hi: Optional[int] = None, steps: Optional[int] = None):
    """Arrange per-utterance noise into the layout `wrnn_generate_segments` reads for segments [lo, hi).

    per_utt[u] is utterance u's noise exactly as the reference draws it for that utterance alone: MOL
    (T, 11*B_u) = per step 10*B_u mixture uniforms (segment-major) then B_u logistic uniforms; RAW (T, B_u, C).
    Entries for utterances without a segment in [lo, hi) may be None.  numpy in -> numpy out, torch in -> torch out.
    steps: the rows are a slice of `steps` consecutive loop steps instead of all plan.T (a step-sliced run).
    """
    hi = plan.n_segments if hi is None else hi
    T = plan.T if steps is None else int(steps)
    n = hi - lo
    utts = [u for u in range(len(plan.lengths)) if plan.first[u] < hi and plan.first[u] + plan.folds[u] > lo]
    sample = per_utt[utts[0]]
    is_torch = isinstance(sample, torch.Tensor)
    cat = (lambda xs, d: torch.cat(xs, dim=d)) if is_torch else (lambda xs, d: np.concatenate(xs, axis=d))
    mix, logi, raw = [], [], []
    for u in utts:
        Bu, f0 = int(plan.folds[u]), int(plan.first[u])
        a, b = max(lo, f0) - f0, min(hi, f0 + Bu) - f0          # this utterance's segments [a, b) are in the block
        z = per_utt[u]
        if mode == 'MOL':
            z = z.reshape(T, 11 * Bu)
            mix.append(z[:, 10 * a:10 * b])
            logi.append(z[:, 10 * Bu + a:10 * Bu + b])
        else:
            raw.append(z.reshape(T, Bu, -1)[:, a:b])
    out = cat(mix + logi, 1) if mode == 'MOL' else cat(raw, 1)
    assert out.shape[0] == T and (out.shape[1] == 11 * n if mode == 'MOL' else out.shape[1] == n)
    return out.contiguous() if is_torch else np.ascontiguousarray(out)


def chunk_utterances(plan: Plan, lo: int, hi: int, max_segments: int):
    """Cut the utterances that own segments in [lo, hi) into consecutive chunks whose segment count in [lo, hi) stays
    <= max_segments (a single longer utterance makes its own chunk).  Returns [(utterances, seg_lo, seg_hi)]."""
    chunks, cur, cur_lo, cur_hi = [], [], None, None
    for u in range(len(plan.lengths)):
        a, b = max(lo, int(plan.first[u])), min(hi, int(plan.first[u] + plan.folds[u]))
        if a >= b:
            continue
        if cur and (b - cur_lo) > max_segments:
            chunks.append((cur, cur_lo, cur_hi))
            cur, cur_lo = [], None
        if cur_lo is None:
            cur_lo = a
        cur.append(u)
        cur_hi = b
    if cur:
        chunks.append((cur, cur_lo, cur_hi))
    return chunks


def generate_corpus(model, mels: Sequence, target: int, overlap: int, mu_law: bool, seeds: Optional[Sequence[int]] = None,
                    group=None, loop_fn=None, return_segments=False, noise_source='cpu', finish='all', check=True,
                    max_segments_per_launch: int = 4096, shard=None, ranks: Optional[int] = None, timings: Optional[dict] = None):
    """Generate every utterance of `mels` (each (1, feat, N_u) or (feat, N_u)) with `model` (a `wavernn_amd.WaveRNN`),
    batched, sharding the folded segments over the ranks of `group` (None = single process).

    seeds[u] plays the role of `torch.manual_seed(seeds[u])` before the reference's `generate()` call for utterance u
    (parity noise: the CPU MT19937 stream incl. the GRUCell constructor draws).  Returns the list of float64 waveforms
    (on every rank), equal to per-utterance `generate(mel_u, ..., batched=True, target, overlap, mu_law)` calls.

    noise_source='device' draws the block's noise from the device generator instead (what the reference does when it
    runs on a GPU; not comparable with a CPU run; `seeds` unused).  finish='all': every rank unfolds every utterance;
    'own': a rank unfolds only the utterances whose first segment lies in its block (None elsewhere in the list).

    The rank's block is processed in chunks of whole utterances of at most `max_segments_per_launch` segments, so the
    conditioning (320 B per audio sample) and the sampling noise of only one chunk are resident at a time; the loop's own
    workspace does not grow with the corpus (conditioning slabs, `wrnn_workspace_bytes_segments`).

    shard=(r, w), without a group: do what rank r of a w-rank job does on its own -- its block of the segment table, the post-loop
    stage of the utterances lying entirely in it -- with no collective (how bench.py shows a GPU's share of config 4 at N = 1).

    ranks: shard over the first `ranks` ranks of the group only (`choose_ranks`; None = all).  timings: optional dict, filled with
    `gather_wait_ms` (host time blocked on the all-gather) and `unfold_under_gather_ms` (post-loop stage of this rank's own utterances,
    run while the gather is in flight).

    loop_fn(mels_up, aux, seg_pos, seg_lim, T, noise, hop) -> (n, T) tensor replaces the HIP loop (tests inject a CPU
    stand-in to exercise the sharding / gather logic under gloo); the default is the model's LoopEngine.
    """
    import torch.distributed as dist
    from .rng import draw_steps
    if noise_source == 'cpu' and seeds is None:
        raise ValueError("noise_source='cpu' (parity noise) needs `seeds` (one per utterance); "
                         "pass noise_source='device' to draw from the device generator instead")
    world = dist.get_world_size(group) if group is not None else 1
    rank = dist.get_rank(group) if group is not None else 0
    if shard is not None:
        if group is not None:
            raise ValueError('shard=(rank, world) emulates a rank without a process group')
        rank, world = int(shard[0]), int(shard[1])
        finish = 'own'
    device = next(model.parameters()).device
    mode = model.mode
    mu_law = mu_law if mode == 'RAW' else False
    hop = model.hop_length
    was_training = model.training
    model.eval()
    mels = [torch.as_tensor(m) for m in mels]
    mels = [m.unsqueeze(0) if m.dim() == 2 else m for m in mels]
    frames = [int(m.size(-1)) for m in mels]
    plan = plan_utterances([n * hop for n in frames], target, overlap)
    bounds = shard_bounds(plan.n_segments, world, ranks)
    lo, hi = bounds[rank]
    n_max = max(h - l for l, h in bounds)

    out_local = torch.zeros(n_max, plan.T, dtype=torch.float32, device=device)
    native_pre = loop_fn is None and getattr(model, 'pre_algo', 'torch') == 'native' and device.type == 'cuda'
    with torch.no_grad():
        for utts, clo, chi in (chunk_utterances(plan, lo, hi, max_segments_per_launch) if hi > lo else []):
            # conditioning of the utterances this chunk touches, concatenated (each starts on a frame boundary)
            noise = [None] * len(mels)
            local_off, off = {}, 0
            for u in utts:
                local_off[u] = off
                off += frames[u] * hop
            n_seg, T = chi - clo, plan.T
            eng = model._loop_engine() if loop_fn is None else None

            def conditioning(rows):
                """rows: the mel one up-sampling stage short (`engine.MelRows`; wrnn_duo_kernel forms the last stage in the loop)"""
                if native_pre:     # the pre-loop kernels write straight into their slices of the concatenated buffers
                    pre = model._pre_engine()
                    n_rows = [pre.rows_of(frames[u]) for u in utts] if rows else None
                    mels_up = torch.empty(sum(n_rows) if rows else off, mels[utts[0]].size(1), dtype=torch.float32, device=device)
                    aux = torch.empty(off // hop, 4 * model.aux_dims, dtype=torch.float32, device=device)
                    r0, jobs = 0, []
                    for k, u in enumerate(utts):
                        a0 = local_off[u]
                        if rows:
                            jobs.append((mels[u], mels_up[r0:r0 + n_rows[k]], aux[a0 // hop:a0 // hop + frames[u]]))
                            r0 += n_rows[k]
                        else:
                            jobs.append((mels[u], mels_up[a0:a0 + frames[u] * hop], aux[a0 // hop:a0 // hop + frames[u]]))
                    pre.upsample_many(jobs, rows=rows, streams=int(getattr(model, 'pre_streams', 8)))      # (the utterances side by side on side streams)
                    if rows:       # utterance k of the chunk: its rows start 2 * indent * k samples later than its cropped samples do
                        from .engine import MelRows
                        indent, order = pre.pad * hop, {u: k for k, u in enumerate(utts)}
                        seg_off = np.array([indent * (2 * order[int(u)] + 1) for u in plan.seg_utt[clo:chi]], dtype=np.int32)
                        mels_up = MelRows(mels_up, off, pre.scales[2], pre.last_taps, seg_off)
                    return mels_up, aux
                ups, auxs = zip(*[model.conditioning(mels[u])[:2] for u in utts])
                return torch.cat(ups).contiguous(), torch.cat(auxs).contiguous()
            rows = eng is not None and native_pre and model.mel_rows_ok(eng, n_seg, T)
            mels_up, aux = conditioning(rows)
            # this chunk's segment table, rebased onto the conditioning of the utterances it touches
            rebase = np.array([local_off[int(u)] - int(plan.offsets[int(u)]) for u in plan.seg_utt[clo:chi]], dtype=np.int64)
            seg_pos = (plan.seg_pos[clo:chi].astype(np.int64) + rebase).astype(np.int32)
            seg_lim = (plan.seg_lim[clo:chi].astype(np.int64) + rebase).astype(np.int32)
            out_view = out_local[clo - lo:chi - lo]
            # the reference's per-utterance CPU streams (parity noise): one private generator per utterance, the GRUCell constructor
            # draws burnt once, then the loop's draws in step order -- so a slice of steps continues every stream where the last stopped
            gens, pool = {}, None
            if noise_source == 'cpu':
                if len(utts) > 1 and mode == 'RAW':       # (MoL: 11 draws per segment-step -- not worth a thread)
                    import os
                    from concurrent.futures import ThreadPoolExecutor
                    # (the host's cores are shared by the ranks of the node: LOCAL_WORLD_SIZE, as torch.distributed.run and bench.py set it)
                    pool = ThreadPoolExecutor(max(1, min(len(utts), (os.cpu_count() or 2) // (2 * max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))))))
                from .rng import burn_ctor_draws
                for u in utts:
                    gens[u] = torch.Generator(device='cpu').manual_seed(int(seeds[u]))
                    burn_ctor_draws(model.rnn_dims, model.aux_dims, 'cpu', gens[u])

            def draw(steps_):
                """noise rows of the next `steps_` loop steps of this chunk's segments, on the device"""
                if noise_source != 'cpu':
                    return draw_steps(mode, n_seg, steps_, model.n_classes, device, 'device')
                def one(u):        # (ATen releases the GIL inside the fill: the utterances' independent streams are drawn side by side)
                    noise[u] = draw_steps(mode, int(plan.folds[u]), steps_, model.n_classes, 'cpu', 'cpu', generator=gens[u])
                if pool is None:
                    for u in utts:
                        one(u)
                else:
                    list(pool.map(one, utts))
                return pack_noise(mode, plan, noise, clo, chi, steps=steps_).to(device)

            if loop_fn is not None:
                out_view[:] = loop_fn(mels_up, aux, seg_pos, seg_lim, T, draw(T), hop)
                if pool is not None:
                    pool.shutdown()
                continue
            from ._lib import ResidencyError
            from .engine import RESUMABLE_KERNELS
            # the persistent loop kernels continue a call (wrnn_options.t_begin / t_end): the noise -- RAW: n_classes floats per
            # segment-step -- is drawn and uploaded in slices of steps, at most `model.noise_chunk_bytes` of it resident (the bound
            # WaveRNN.generate() keeps); any other kernel takes the whole call's noise at once
            steps = T
            if eng.plan(n_seg, T, algo=model.loop_algo)['kernel'] in RESUMABLE_KERNELS:
                per_step = n_seg * (11 if mode == 'MOL' else model.n_classes) * 4
                steps = max(1, min(T, getattr(model, 'noise_chunk_bytes', 2 << 30) // per_step))
                steps = -(-T // (-(-T // steps)))            # equal slices (no short tail slice with its own launches)
            try:
                for t0 in range(0, T, steps):
                    t1 = min(T, t0 + steps)
                    eng.run_segments(mels_up, aux, seg_pos, seg_lim, T, draw(t1 - t0), hop, algo=model.loop_algo, check=check, out=out_view,
                                     t_range=None if (t0 == 0 and t1 == T) else (t0, t1))
            except ResidencyError as e:
                # the persistent grid was refused (on the first slice: a continuation cannot change kernels).  The other loop kernels read
                # the up-sampled mel in full; the engine then degrades as usual on an unsliced call (one workgroup per CU, then the
                # stream kernel), from the start of every utterance's noise stream
                import warnings
                warnings.warn(f'wavernn_amd: {e}; redoing the chunk unsliced on the next kernel')
                if rows:
                    mels_up, aux = conditioning(False)
                for u in gens:
                    gens[u].manual_seed(int(seeds[u]))
                    burn_ctor_draws(model.rnn_dims, model.aux_dims, 'cpu', gens[u])
                eng.run_segments(mels_up, aux, seg_pos, seg_lim, T, draw(T), hop, algo=model.loop_algo, check=check, out=out_view)
            finally:
                if pool is not None:
                    pool.shutdown()
    # ---- the ONE collective of the path: an all-gather of the finished audio, asynchronous so that the utterances lying entirely in
    #      this rank's block are unfolded (post-loop stage) while the other ranks' segments are still on their way over xGMI
    n_utt = len(frames)
    work, gathered = None, None
    if group is not None:
        gathered = [torch.empty_like(out_local) for _ in range(world)]
        work = dist.all_gather(gathered, out_local, group=group, async_op=True)
        if timings is not None:                         # what this rank receives over the fabric in the ONE collective of the path
            timings['gather_bytes'] = timings.get('gather_bytes', 0) + (world - 1) * out_local.numel() * out_local.element_size()
            timings['world_seen'] = dist.get_world_size(group)

    def gathered_segments():
        nonlocal work
        if group is None:
            if shard is not None:
                raise ValueError('an emulated shard has only its own block of segments')
            return out_local[:plan.n_segments]
        if work is not None:
            t_w = time.perf_counter()
            work.wait()
            work = None
            if timings is not None:
                timings['gather_wait_ms'] = timings.get('gather_wait_ms', 0.0) + (time.perf_counter() - t_w) * 1e3
        return torch.cat([gathered[r][:h - l] for r, (l, h) in enumerate(bounds)])

    if was_training:
        model.train()
    if return_segments:
        return gathered_segments().cpu().numpy().astype(np.float64), plan
    mine_u = [u for u in range(n_utt) if finish != 'own' or (lo <= plan.first[u] < hi)]
    local_u = [u for u in mine_u if lo <= plan.first[u] and plan.first[u] + plan.folds[u] <= hi]      # no segment on another rank
    local_set = set(local_u)
    rest_u = [u for u in mine_u if u not in local_set]
    outs = [None] * n_utt
    native_post = loop_fn is None and getattr(model, 'post_algo', 'numpy') == 'native' and out_local.is_cuda

    def unfold(segs, first_of, us):
        """Post-loop stage (reference :245-260, :342-405) of utterances `us`, whose first segment is row first_of(u) of `segs`."""
        if not us:
            return
        if native_post:       # on the device (float64, reference order): one launch for all of them
            from .post import unfold_on_device
            wav, sl = unfold_on_device(segs, [first_of(u) for u in us], [int(plan.folds[u]) for u in us],
                                       [(frames[u] - 1) * hop for u in us], overlap, hop, model.n_classes, mu_law, True)
            if getattr(model, 'pinned_output', True):
                # the finished audio (8 B per sample) into page-locked memory at the link's rate (a pageable copy is staged through bounce buffers);
                # torch's caching host allocator hands the pages back to the next call; the returned arrays are views that keep the buffer alive
                host = torch.empty(wav.shape, dtype=wav.dtype, pin_memory=True)
                host.copy_(wav, non_blocking=True)
                torch.cuda.current_stream(wav.device).synchronize()
                wav = host.numpy()
            else:
                wav = wav.cpu().numpy()
            for (a, b), u in zip(sl, us):
                outs[u] = wav[a:b]
            return
        host = segs.cpu().numpy().astype(np.float64)
        for u in us:
            y = host[first_of(u):first_of(u) + int(plan.folds[u])].copy()
            if mu_law:
                y = _fold.decode_mu_law(y, model.n_classes, False)
            y = _fold.xfade_and_unfold(y, target, overlap)
            outs[u] = _fold.finish_waveform(y, (frames[u] - 1) * hop, hop)

    t_u = time.perf_counter()
    unfold(out_local, lambda u: int(plan.first[u]) - lo, local_u)          # under the all-gather
    if timings is not None:
        timings['unfold_under_gather_ms'] = timings.get('unfold_under_gather_ms', 0.0) + (time.perf_counter() - t_u) * 1e3
    if shard is not None:
        return outs
    if rest_u:
        unfold(gathered_segments(), lambda u: int(plan.first[u]), rest_u)
    elif work is not None:
        t_w = time.perf_counter()
        work.wait()
        if timings is not None:
            timings['gather_wait_ms'] = timings.get('gather_wait_ms', 0.0) + (time.perf_counter() - t_w) * 1e3
    return outs
