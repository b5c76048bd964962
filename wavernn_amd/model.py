"""`WaveRNN`: drop-in for the reference class of the same name (fatchord/WaveRNN models/fatchord_version.py:92-435)
whose `generate()` runs the per-sample loop on the MI355X-native kernels.

Same constructor signature, same parameter / buffer names (so the reference's `.pyt` checkpoints load with
`load()`), same `generate(mels, save_path, batched, target, overlap, mu_law) -> np.ndarray[float64]`
contract and side effects (eval at entry, train at exit, RNG consumption, WAV written).  What differs:
the loop (reference :192-245) is ONE call into libwavernn_amd.so, conditioning is never folded in memory, and
the model must live on a HIP device -- there is no CPU path here (use the reference for that).
"""
import sys
import time
from pathlib import Path
from typing import Union

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from . import fold as _fold
from .dsp import save_wav
from .engine import LoopEngine, LOOP_KEYS, RESUMABLE_KERNELS, MEL_STAGE_KERNELS
from .rng import burn_ctor_draws, draw_steps


class ResBlock(nn.Module):
    """1x1 conv -> BN -> ReLU -> 1x1 conv -> BN, plus skip (reference :13-28)."""

    def __init__(self, dims):
        super().__init__()
        self.conv1 = nn.Conv1d(dims, dims, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(dims, dims, kernel_size=1, bias=False)
        self.batch_norm1 = nn.BatchNorm1d(dims)
        self.batch_norm2 = nn.BatchNorm1d(dims)

    def forward(self, x):
        y = F.relu(self.batch_norm1(self.conv1(x)))
        return x + self.batch_norm2(self.conv2(y))


class MelResNet(nn.Module):
    """k=2*pad+1 conv (no padding) -> BN -> ReLU -> res blocks -> 1x1 conv (reference :31-48)."""

    def __init__(self, res_blocks, in_dims, compute_dims, res_out_dims, pad):
        super().__init__()
        self.conv_in = nn.Conv1d(in_dims, compute_dims, kernel_size=2 * pad + 1, bias=False)
        self.batch_norm = nn.BatchNorm1d(compute_dims)
        self.layers = nn.ModuleList([ResBlock(compute_dims) for _ in range(res_blocks)])
        self.conv_out = nn.Conv1d(compute_dims, res_out_dims, kernel_size=1)

    def forward(self, x):
        x = F.relu(self.batch_norm(self.conv_in(x)))
        for layer in self.layers:
            x = layer(x)
        return self.conv_out(x)


class Stretch2d(nn.Module):
    """nearest-neighbour repeat along width (x_scale) and height (y_scale) (reference :51-61)."""

    def __init__(self, x_scale, y_scale):
        super().__init__()
        self.x_scale, self.y_scale = x_scale, y_scale

    def forward(self, x):
        return x.repeat_interleave(self.y_scale, dim=2).repeat_interleave(self.x_scale, dim=3)


class UpsampleNetwork(nn.Module):
    """aux = MelResNet(m) repeated hop times; mel = 3 x (repeat by s, box FIR of 2s+1 taps), trimmed (reference :64-89)."""

    def __init__(self, feat_dims, upsample_scales, compute_dims, res_blocks, res_out_dims, pad):
        super().__init__()
        self.total_scale = int(np.prod(upsample_scales))
        self.indent = pad * self.total_scale
        self.resnet = MelResNet(res_blocks, feat_dims, compute_dims, res_out_dims, pad)
        self.resnet_stretch = Stretch2d(self.total_scale, 1)
        self.up_layers = nn.ModuleList()
        for scale in upsample_scales:
            conv = nn.Conv2d(1, 1, kernel_size=(1, 2 * scale + 1), padding=(0, scale), bias=False)
            conv.weight.data.fill_(1. / (2 * scale + 1))
            self.up_layers.append(Stretch2d(scale, 1))
            self.up_layers.append(conv)

    def aux_frames(self, m):
        """(b, feat, N+2*pad) -> (b, res_out, N): the frame-rate aux features, before Stretch2d."""
        return self.resnet(m)

    def upsample_mel(self, m):
        """(b, feat, N+2*pad) -> (b, N*hop, feat)."""
        m = m.unsqueeze(1)
        for f in self.up_layers:
            m = f(m)
        return m.squeeze(1)[:, :, self.indent:-self.indent].transpose(1, 2)

    def forward(self, m):
        aux = self.resnet_stretch(self.aux_frames(m).unsqueeze(1)).squeeze(1)
        return self.upsample_mel(m), aux.transpose(1, 2)


class WaveRNN(nn.Module):
    def __init__(self, rnn_dims, fc_dims, bits, pad, upsample_factors, feat_dims, compute_dims, res_out_dims,
                 res_blocks, hop_length, sample_rate, mode='RAW'):
        super().__init__()
        self.mode = mode
        self.pad = pad
        if self.mode == 'RAW':
            self.n_classes = 2 ** bits
        elif self.mode == 'MOL':
            self.n_classes = 30
        else:
            RuntimeError("Unknown model mode value - ", self.mode)   # constructed, not raised: reference quirk (:104)
        self.rnn_dims = rnn_dims
        self.aux_dims = res_out_dims // 4
        self.hop_length = hop_length
        self.sample_rate = sample_rate

        self.upsample = UpsampleNetwork(feat_dims, upsample_factors, compute_dims, res_blocks, res_out_dims, pad)
        self.I = nn.Linear(feat_dims + self.aux_dims + 1, rnn_dims)
        self.rnn1 = nn.GRU(rnn_dims, rnn_dims, batch_first=True)
        self.rnn2 = nn.GRU(rnn_dims + self.aux_dims, rnn_dims, batch_first=True)
        self.fc1 = nn.Linear(rnn_dims + self.aux_dims, fc_dims)
        self.fc2 = nn.Linear(fc_dims + self.aux_dims, fc_dims)
        self.fc3 = nn.Linear(fc_dims, self.n_classes)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.num_params()

        #: 'cpu' = consume torch's global CPU generator exactly like the reference's CPU run (parity);
        #: 'device' = device Philox generator (what the reference does when it runs on a GPU)
        self.noise_source = 'cpu'
        #: 'auto' | 'loop' | 'sparse' | 'stream'  (WRNN_ALGO_*, include/wavernn_amd.h)
        self.loop_algo = 'auto'
        #: 'native' = the HIP pre-loop kernels (MFMA MelResNet + box-filter up-sampling, wrnn_pre_*);
        #: 'torch' = the nn.Modules below through PyTorch-ROCm (MIOpen)
        self.pre_algo = 'native'
        #: generate_corpus: the pre-loop kernels of a chunk's utterances run side by side on this many side streams (1 = one after the other on the
        #: current stream); the finished float64 audio comes back through page-locked memory (False: a pageable copy)
        self.pre_streams = 8
        self.pinned_output = True
        #: True = when the call runs on wrnn_duo_kernel / wrnn_chain_kernel / wrnn_sparse_kernel (and the pre-loop stage is 'native' with a last stretch
        #: factor of 11), the LAST up-sampling stage and the crop are formed inside the loop from that stage's input (`engine.MelRows`): the [L, feat]
        #: up-sampled mel is never written -- 29 B of conditioning per audio sample resident instead of 320 B (a 64-utterance corpus:
        #: 0.33 GB instead of 3.6 GB).  False = always materialise it (what every other loop kernel reads).  None (default) = False since the last
        #: session of round 6: forming the stage costs the workgroups that form cI three times the loads and ~60 more VALU instructions per slot and
        #: step -- 1.6 % of wrnn_duo_kernel's step at 256 segments, 4-5 % of the latency kernels' (one utterance: 9.26 vs 8.80 us per step;
        #: block-sparse: 8.48 vs 8.15; profiles/r06bo_mel_ab_chain.log, r06be_mel_ab.log) -- and a device with 288 GB of HBM is not short of 0.9 GB per 256
        #: segments.  (For the 9-bit mode it always was the default: the three-row form of that stage is another float32 rounding of the mel (<= 1e-6),
        #: and RAW is compared class index by class index -- against the C oracle the kernel parts ways in 2 of 1,024 segments of 12,100 steps with
        #: the mel formed in the loop and in 1 with the materialised mel, profiles/r05a_raw_flips.json)  `model.mel_in_loop = True`
        #: opts in to the memory-saving form; both are parity-tested at full size.
        self.mel_in_loop = None
        #: 'native' = cross-fade / unfold / mu-law / tail fade on the device in float64 (wrnn_post_unfold);
        #: 'numpy' = the host helpers of fold.py
        self.post_algo = 'native'
        #: upper bound on the sampling noise resident at once (bytes); longer runs draw it slice by slice
        self.noise_chunk_bytes = 2 << 30          # (round 6: 2 GB, was 128 MB -- every slice boundary is a relaunch of the loop kernel behind a draw; 9-bit RAW on 256 segments: 48 -> 9 launches, 369.8 -> 351.0 ms per pass, profiles/r06br_noise_chunk.log)
        #: optional `f(steps_done, T, n_segments, seconds)` called after every noise slice of generate() (row a17: the
        #: reference's gen_display read-out, :241/:267-271); None = silent, no extra synchronisation
        self.progress_callback = None
        self._engine = None
        self._engine_key = None
        self._pre = None
        self._pre_key = None
        self.last_loop_ms = None
        self.last_loop_kernel = None

    # ------------------------------------------------------------------------------------------------
    def forward(self, x, mels):
        """Teacher-forced training forward (reference :131-167); plain PyTorch, not part of the hot path."""
        self.step += 1
        bsize = x.size(0)
        device = x.device
        h1 = torch.zeros(1, bsize, self.rnn_dims, device=device)
        h2 = torch.zeros(1, bsize, self.rnn_dims, device=device)
        mels, aux = self.upsample(mels)
        d = self.aux_dims
        a1, a2, a3, a4 = (aux[:, :, d * i:d * (i + 1)] for i in range(4))
        x = self.I(torch.cat([x.unsqueeze(-1), mels, a1], dim=2))
        res = x
        x, _ = self.rnn1(x, h1)
        x = x + res
        res = x
        x, _ = self.rnn2(torch.cat([x, a2], dim=2), h2)
        x = x + res
        x = F.relu(self.fc1(torch.cat([x, a3], dim=2)))
        x = F.relu(self.fc2(torch.cat([x, a4], dim=2)))
        return self.fc3(x)

    # ------------------------------------------------------------------------------------------------
    def invalidate_engines(self):
        """Drop the cached device weight packs.  They are rebuilt automatically when a weight tensor is replaced or edited
        through autograd-visible in-place ops (`load()`, `load_state_dict()`, an optimiser step: each bumps the tensor's
        `_version`), and by `prune.Pruner` (which calls this).  Edits made behind autograd's back -- `p.data.mul_(mask)`, what
        the reference's pruning notebook does by hand -- bump nothing PyTorch exposes: call this after them.  (Rounds 1-2 keyed
        the packs on a content hash instead: a reduction over all 15 MB of weights plus a device synchronisation on EVERY
        generate() call, which the C ABI's "no hidden synchronisation" rule forbids.)"""
        self._engine = self._pre = self._engine_key = self._pre_key = None

    @staticmethod
    def _weights_key(sd):
        return tuple((k, v.data_ptr(), str(v.device), tuple(v.shape), str(v.dtype), int(v._version)) for k, v in sorted(sd.items()))

    def _loop_engine(self):
        sd = {k: v for k, v in self.state_dict().items() if k in LOOP_KEYS.values()}
        key = self._weights_key(sd)
        if self._engine is None or key != self._engine_key:
            dev = next(self.parameters()).device
            self._engine = LoopEngine(sd, self.mode, device=dev)
            self._engine_key = key
        return self._engine

    def _pre_engine(self):
        from .pre import PreEngine
        sd = {k: v for k, v in self.state_dict().items() if k.startswith('upsample.')}
        key = self._weights_key(sd)
        if self._pre is None or key != self._pre_key:
            self._pre = PreEngine(sd, device=next(self.parameters()).device)
            self._pre_key = key
        return self._pre

    @staticmethod
    def _require_hip_device(device):
        if device.type != 'cuda':
            raise RuntimeError('wavernn_amd.WaveRNN.generate needs the model on a HIP device (model.to("cuda")); '
                               'there is no CPU path in this package')

    def mel_rows_ok(self, eng, n_segments, T):
        """Whether a run over this many segments takes the mel one up-sampling stage short (`mel_in_loop`): it runs on wrnn_duo_kernel / wrnn_sparse_kernel and
        the HIP pre-loop stage ends with the stretch factor that kernel is built for."""
        device = next(self.parameters()).device
        in_loop = bool(self.mel_in_loop)                    # (None = the default = materialised, both modes: see `mel_in_loop`)
        if not (in_loop and self.pre_algo == 'native' and device.type == 'cuda'):
            return False
        if eng.plan(n_segments, T, algo=self.loop_algo)['kernel'] not in MEL_STAGE_KERNELS:
            return False
        try:
            pre = self._pre_engine()
            return pre.scales[2] == 11 and pre.pad >= 1        # (pad 0: the stage's first output would need a row in front of the buffer)
        except _lib.WrnnError:
            return False

    def conditioning(self, mels, rows=False):
        """Pre-loop stage (reference :183-186) without the Stretch2d repeat of aux and without the fold:
        returns mels_up (L, feat), aux frames (N, res_out), wave_len.  rows=True (see `mel_rows_ok`): an `engine.MelRows` -- the input of
        the last up-sampling stage -- in place of mels_up."""
        device = next(self.parameters()).device
        mels = torch.as_tensor(mels, device=device)
        wave_len = (mels.size(-1) - 1) * self.hop_length
        if rows:
            from .engine import MelRows
            pre = self._pre_engine()
            mel_rows, aux = pre.upsample_rows(mels.float())
            return MelRows(mel_rows, mels.size(-1) * self.hop_length, pre.scales[2], pre.last_taps, self.pad * self.hop_length), aux, wave_len
        if self.pre_algo == 'native' and device.type == 'cuda':
            try:
                mels_up, aux = self._pre_engine().upsample(mels.float())
                return mels_up, aux, wave_len
            except _lib.WrnnError as e:
                # the HIP pre-loop kernels take any UpsampleNetwork dims (round 5: wrnn_resnet_generic_kernel beside the MFMA kernel of the
                # shipped ones) up to what one workgroup's LDS holds; a wider network runs through the nn.Modules below on the device
                # (PyTorch-ROCm / MIOpen) -- still no CPU path
                if 'too wide for the pre-loop kernel' not in str(e):
                    raise
                import warnings
                warnings.warn(f'wavernn_amd: {e}; up-sampling with the PyTorch-ROCm modules instead')
                self.pre_algo = 'torch'
        m = _fold.pad_tensor(mels.transpose(1, 2), pad=self.pad, side='both').transpose(1, 2)
        mels_up = self.upsample.upsample_mel(m)[0].contiguous()
        aux = self.upsample.aux_frames(m)[0].transpose(0, 1).contiguous()
        return mels_up, aux, wave_len

    def generate(self, mels, save_path: Union[str, Path], batched, target, overlap, mu_law):
        self.eval()
        device = next(self.parameters()).device
        self._require_hip_device(device)
        if self.mode not in ('RAW', 'MOL'):
            raise RuntimeError("Unknown model mode value - ", self.mode)
        mu_law = mu_law if self.mode == 'RAW' else False

        with torch.no_grad():
            L = int(torch.as_tensor(mels).size(-1)) * self.hop_length
            if batched:
                B, _ = _fold.fold_geometry(L, target, overlap)
                T, stride = target + 2 * overlap, target + overlap
            else:
                B, T, stride = 1, L, 0
            eng = self._loop_engine()
            rows = self.mel_rows_ok(eng, B, T)
            try:
                mels_up, aux, wave_len = self.conditioning(mels, rows=rows)
            except _lib.WrnnError:
                if not rows:
                    raise
                rows = False                                 # (the rows form was refused: the materialised mel, as any other loop kernel reads it)
                mels_up, aux, wave_len = self.conditioning(mels)
            assert (mels_up.L if rows else mels_up.size(0)) == L
            burn_ctor_draws(self.rnn_dims, self.aux_dims, self.noise_source)
            # the sampling noise is drawn and uploaded in slices of steps (RAW: B * n_classes floats per step), each slice
            # continuing the loop where the previous one stopped (wrnn_options.t_begin / t_end)
            per_step = B * (11 if self.mode == 'MOL' else self.n_classes) * 4
            resumable = eng.plan(B, T, algo=self.loop_algo)['kernel'] in RESUMABLE_KERNELS
            chunk = max(1, min(T, self.noise_chunk_bytes // per_step)) if resumable else T
            chunk = -(-T // (-(-T // chunk)))                # equal slices (no short tail slice with its own launches)
            rng_state = torch.get_rng_state() if (self.noise_source == 'cpu' and (chunk < T or rows)) else None
            algo = self.loop_algo
            import warnings
            try:
                out = self._run_sliced(eng, mels_up, aux, B, T, stride, chunk, algo, device)
            except _lib.ResidencyError as e:
                # `auto` picked a persistent kernel but its cooperative launch was refused (CU masking, a smaller partition, another
                # cooperative kernel).  From the same point of the noise stream: with the mel one stage short (only wrnn_duo_kernel reads
                # that) redo the conditioning in full and let the engine degrade as usual (one workgroup per CU, then the stream kernel);
                # otherwise -- the refusal struck the first slice of a sliced run -- redo the call UNSLICED on the stream kernel
                if rng_state is not None:
                    torch.set_rng_state(rng_state)
                if rows:
                    warnings.warn(f'wavernn_amd: {e}; up-sampling the mel in full for the other loop kernels')
                    mels_up, aux, _ = self.conditioning(mels)
                    try:
                        out = self._run_sliced(eng, mels_up, aux, B, T, stride, chunk, algo, device)
                        e = None
                    except _lib.ResidencyError as e2:
                        e = e2
                        if rng_state is not None:
                            torch.set_rng_state(rng_state)
                if e is not None:
                    warnings.warn(f'wavernn_amd: {e}; using the stream kernel')
                    out = self._run_sliced(eng, mels_up, aux, B, T, stride, T, 'stream', device)
            self.last_loop_ms = eng.last_loop_ms()
            self.last_loop_kernel = eng.last_loop_kernel()

        if self.post_algo == 'native':
            from .post import unfold_on_device
            wav, _ = unfold_on_device(out, [0], [B], [wave_len], overlap, self.hop_length, self.n_classes, mu_law, batched)
            output = wav.cpu().numpy()
        else:
            output = out.cpu().numpy().astype(np.float64)
            if mu_law:
                output = _fold.decode_mu_law(output, self.n_classes, False)
            if batched:
                output = _fold.xfade_and_unfold(output, target, overlap)
            else:
                output = output[0]
            output = _fold.finish_waveform(output, wave_len, self.hop_length)
        save_wav(output, save_path, self.sample_rate)
        self.train()
        return output

    def _run_sliced(self, eng, mels_up, aux, B, T, stride, chunk, algo, device):
        """The loop in slices of `chunk` steps: each slice draws its own rows of sampling noise and continues the loop where the
        previous one stopped (wrnn_options.t_begin / t_end), so at most `noise_chunk_bytes` of noise are resident.
        `progress_callback(steps_done, T, B, seconds)`, if set, is driven by wrnn_options.progress: a host function enqueued
        behind every conditioning slab of the loop kernel -- the reference's `gen_display` read-out (:241, :267-271) at slab
        granularity, without a synchronisation and with nothing inside the kernel."""
        out = None
        prog = None
        if self.progress_callback is not None:
            t_start, cb = time.perf_counter(), self.progress_callback
            prog = lambda done, T_, n_: cb(done, T_, n_, time.perf_counter() - t_start)
        for t0 in range(0, T, chunk):
            t1 = min(T, t0 + chunk)
            noise = draw_steps(self.mode, B, t1 - t0, self.n_classes, device, self.noise_source)
            out = eng.run(mels_up, aux, B, T, stride, noise, self.hop_length, algo=algo, out=out,
                          t_range=None if (t0 == 0 and t1 == T) else (t0, t1), progress=prog)
        return out

    def gen_display(self, i, seq_len, b_size, gen_rate):
        """The reference's progress line (:267-271; utils/display.py:9-18 `progbar` + `stream`), for use as

            model.progress_callback = lambda done, T, B, dt: model.gen_display(done, T, B, done / dt * B / 1000)

        The reference calls it every 100 steps from inside its Python loop (:241); here the loop is one persistent kernel per
        conditioning slab, so the read-out exists per slab (a few hundred steps; `wrnn_options.slab_steps`)."""
        bar_len = 16
        done = (i * bar_len) // max(1, seq_len)
        pbar = ''.join(chr(0x2588) if k < done else chr(0x2591) for k in range(bar_len))
        msg = f'| {pbar} {i * b_size}/{seq_len * b_size} | Batch Size: {b_size} | Gen Rate: {gen_rate:.1f}kHz | '
        sys.stdout.write(f'\r{msg}')

    # -- API parity helpers (reference :281-435) -------------------------------------------------------
    def pad_tensor(self, x, pad, side='both'):
        return _fold.pad_tensor(x, pad, side)

    def fold_with_overlap(self, x, target, overlap):
        return _fold.fold_with_overlap(x, target, overlap)

    def xfade_and_unfold(self, y, target, overlap):
        return _fold.xfade_and_unfold(y, target, overlap)

    def get_step(self):
        return self.step.data.item()

    def log(self, path, msg):
        with open(path, 'a') as f:
            print(msg, file=f)

    def load(self, path: Union[str, Path]):
        device = next(self.parameters()).device
        self.load_state_dict(torch.load(path, map_location=device), strict=False)

    def save(self, path: Union[str, Path]):
        torch.save(self.state_dict(), path)

    def num_params(self, print_out=True):
        n = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad) / 1_000_000
        if print_out:
            print('Trainable Parameters: %.3fM' % n)
        return n
