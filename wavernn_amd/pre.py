"""PreEngine: the Python face of the pre-loop C ABI (`wrnn_pre_*`, include/wavernn_amd.h): `UpsampleNetwork.forward`
(reference models/fatchord_version.py:82-89) as hand-written HIP -- MelResNet on f32 MFMA + the Stretch2d/box-filter
mel up-sampling -- producing exactly what the loop ABI consumes: mels_up [L, feat] and aux per FRAME [N, res_out]."""
import ctypes

import numpy as np
import torch

from . import _lib


def _f32(v):
    if isinstance(v, torch.Tensor):
        v = v.detach().to('cpu', torch.float32).numpy()
    return np.ascontiguousarray(v, dtype=np.float32)


class PreEngine:
    """state_dict: mapping with the reference's `upsample.*` keys (torch tensors or numpy arrays)."""

    def __init__(self, state_dict, device=None):
        if not torch.cuda.is_available():
            raise _lib.WrnnError('wavernn_amd needs a HIP device; there is no CPU fallback')
        self.lib = _lib.lib()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        sd = state_dict
        conv_in = _f32(sd['upsample.resnet.conv_in.weight'])
        C, feat, k = conv_in.shape
        blocks = 0
        while f'upsample.resnet.layers.{blocks}.conv1.weight' in sd:
            blocks += 1
        bn = lambda p: np.stack([_f32(sd[p + s]) for s in ('.weight', '.bias', '.running_mean', '.running_var')])
        res_w = np.stack([np.stack([_f32(sd[f'upsample.resnet.layers.{i}.conv{j}.weight'])[:, :, 0] for j in (1, 2)])
                          for i in range(blocks)]) if blocks else np.zeros((0, 2, C, C), np.float32)
        res_bn = np.stack([np.stack([bn(f'upsample.resnet.layers.{i}.batch_norm{j}') for j in (1, 2)])
                           for i in range(blocks)]) if blocks else np.zeros((0, 2, 4, C), np.float32)
        conv_out_w = _f32(sd['upsample.resnet.conv_out.weight'])[:, :, 0]
        ups = [_f32(sd[f'upsample.up_layers.{2 * i + 1}.weight']).reshape(-1) for i in range(3)]
        scales = [(u.shape[0] - 1) // 2 for u in ups]
        host = dict(conv_in_w=np.ascontiguousarray(conv_in.reshape(C, feat * k)), bn_in=np.ascontiguousarray(bn('upsample.resnet.batch_norm')),
                    res_w=np.ascontiguousarray(res_w), res_bn=np.ascontiguousarray(res_bn), conv_out_w=np.ascontiguousarray(conv_out_w),
                    conv_out_b=_f32(sd['upsample.resnet.conv_out.bias']), up_w=np.ascontiguousarray(np.concatenate(ups)))
        w = _lib.PreWeights()
        w.feat_dims, w.compute_dims, w.res_out_dims, w.res_blocks, w.pad = feat, C, conv_out_w.shape[0], blocks, (k - 1) // 2
        for i in range(3):
            w.upsample_factors[i] = scales[i]
        for name, arr in host.items():
            setattr(w, name, arr.ctypes.data)
        self.feat_dims, self.res_out_dims = int(feat), int(conv_out_w.shape[0])
        self.pad, self.scales, self.last_taps = int(w.pad), [int(s_) for s_ in scales], ups[2].copy()
        pre = ctypes.c_void_p()
        rc = self.lib.wrnn_pre_create(ctypes.byref(w), (self.device.index if self.device.index is not None else torch.cuda.current_device()), ctypes.byref(pre))
        if rc != 0:
            raise _lib.WrnnError(f'wrnn_pre_create failed (rc={rc}): {self.lib.wrnn_pre_last_error().decode()}')
        self._pre = pre
        self.hop = int(self.lib.wrnn_pre_hop(pre))
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, '_pre', None):
                self.lib.wrnn_pre_destroy(self._pre)
                self._pre = None
        except Exception:
            pass

    def upsample_rows(self, mel, rows=None, aux=None):
        """The same, one stage short (`wrnn_pre_upsample_rows`): (rows [(N + 2 pad) * s0 * s1, feat] = the input of the last Stretch2d +
        conv stage, aux [N, res_out]) -- what `LoopEngine.run*` takes wrapped in `engine.MelRows`; wrnn_duo_kernel forms the last stage
        and the crop inside the loop."""
        return self.upsample(mel, mels_up=rows, aux=aux, _rows=True)

    def rows_of(self, n_frames):
        return (int(n_frames) + 2 * self.pad) * self.scales[0] * self.scales[1]

    def upsample(self, mel, mels_up=None, aux=None, _rows=False):
        """mel: (feat, N) or (1, feat, N) float32 CUDA tensor -> (mels_up [N*hop, feat], aux [N, res_out]) CUDA tensors.
        `mels_up` / `aux` may be preallocated views (several utterances into one concatenated buffer)."""
        if mel.dim() == 3:
            mel = mel[0]
        mel = mel.to(self.device, torch.float32).contiguous()
        n = int(mel.shape[1])
        if mel.shape[0] != self.feat_dims:
            raise ValueError(f'Expected a mel shaped ({self.feat_dims}, n_hops), but got {tuple(mel.shape)}!')
        n_out = self.rows_of(n) if _rows else n * self.hop
        if mels_up is None:
            mels_up = torch.empty(n_out, self.feat_dims, dtype=torch.float32, device=self.device)
        if aux is None:
            aux = torch.empty(n, self.res_out_dims, dtype=torch.float32, device=self.device)
        assert mels_up.is_contiguous() and aux.is_contiguous() and mels_up.shape == (n_out, self.feat_dims)
        nbytes = int(self.lib.wrnn_pre_workspace_bytes(self._pre, n))
        stream = torch.cuda.current_stream(self.device).cuda_stream
        # one workspace per stream: `upsample_many` runs several utterances side by side on side streams
        if self._ws is None:
            self._ws = {}
        ws = self._ws.get(stream)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[stream] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        fn = self.lib.wrnn_pre_upsample_rows if _rows else self.lib.wrnn_pre_upsample
        rc = fn(self._pre, mel.data_ptr(), n, mels_up.data_ptr(), aux.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        if rc != 0:
            raise _lib.WrnnError(f'wrnn_pre_upsample failed (rc={rc}): {self.lib.wrnn_pre_last_error().decode()}')
        return mels_up, aux

    def upsample_many(self, jobs, rows=False, streams=8):
        """jobs: [(mel, mels_up view, aux view)] -- the utterances of a chunk, each into its slice of the concatenated buffers.  An utterance's three
        kernels fill 41 of the 256 CUs (641 frames) and depend on each other; the utterances do not: they run side by side on up to `streams` side
        streams (own workspace each), forked from and joined to the current stream by events.  streams <= 1: one after the other on the current stream."""
        if streams <= 1 or len(jobs) <= 1:
            for mel, up, aux in jobs:
                self.upsample(mel, mels_up=up, aux=aux, _rows=rows)
            return
        cur = torch.cuda.current_stream(self.device)
        if getattr(self, '_side', None) is None or len(self._side) < streams:
            self._side = [torch.cuda.Stream(device=self.device) for _ in range(streams)]
        mels = [mel[0] if mel.dim() == 3 else mel for mel, _, _ in jobs]
        mels = [m.to(self.device, torch.float32).contiguous() for m in mels]      # (on the current stream, in front of the fork)
        fork = cur.record_event()
        used = self._side[:min(streams, len(jobs))]
        for s in used:
            s.wait_event(fork)
        for k, (m, (_, up, aux)) in enumerate(zip(mels, jobs)):
            with torch.cuda.stream(used[k % len(used)]):
                self.upsample(m, mels_up=up, aux=aux, _rows=rows)
        for s in used:
            cur.wait_stream(s)                   # (the buffers were allocated on `cur` and are next used there: no record_stream needed)

