"""Tacotron inference in front of the vocoder -- BASELINE config 3's caller side (`gen_tacotron.py wavernn`, reference
gen_tacotron.py:139-166; SURVEY.md section 8 row f3's host).

This is NOT the hot path and not a kernel: it is the reference's `Tacotron.generate()` (models/tacotron.py:370-430) restated as
a FUNCTIONAL forward over a reference state dict (`tts_model.state_dict()` / `latest_weights.pyt`), on whatever device the
tensors live on -- PyTorch-ROCm on an MI355X.  No module tree, no training code, eval semantics only (dropout and zoneout
are identities in `generate()`, which calls `self.eval()` first, :371).  On the CPU it reproduces the reference bit for bit
(tests/test_tacotron_mirror.py, build container; tests/golden/tacotron_decoder_200f.npz holds the reference's own output for the
GPU tests' weights); on the GPU the decoder loop -- one sentence = one serial chain of ~40 small ops per mel frame -- runs as ONE
persistent kernel (`generate(..., kernel=True)`: `wrnn_taco_decode`, csrc/wrnn_taco.hip) and the CBHGs'
bidirectional GRUs as `wrnn_bigru`; the eager loop stays as the any-device form.

    tts = TacotronInference(state_dict, device='cuda')
    _, m, attn = tts.generate(ids, steps=800)                 # same returns as the reference: (80, N), (fft, N), (N, chars); the
                                                              # vocoder takes the SECOND one (postnet output; gen_tacotron.py:142)
    m = torch.tensor(np.clip((m + 4) / 8, 0, 1)).unsqueeze(0)     # gen_tacotron.py:143-145
    voc.generate(m, path, True, 11_000, 550, True)               # gen_tacotron.py:161-163
"""
import re

import numpy as np
import torch
import torch.nn.functional as F

#: the character part of the reference's symbol table (utils/text/symbols.py:8-17: pad, '-', punctuation, letters; the ARPAbet
#: symbols follow and are only reachable through {curly brace} input)
SYMBOLS = ['_'] + list('-') + list('!\'(),.:;? ') + list('ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz')
_ID = {s: i for i, s in enumerate(SYMBOLS)}


def text_to_ids(text):
    """`text_to_sequence(text, ['basic_cleaners'])` (utils/text/__init__.py:16-43, cleaners.py:69-73) for plain text: lowercase,
    collapse whitespace, drop unknown symbols and the pad.  The reference's default `english_cleaners` additionally
    transliterates and spells out numbers / abbreviations (needs unidecode + inflect); for text without those both agree."""
    text = re.sub(r'\s+', ' ', text.lower())
    return [_ID[c] for c in text if c in _ID and c not in '_~']


class TacotronInference:
    def __init__(self, state_dict, device=None):
        dev = torch.device(device) if device is not None else None
        self.p = {k: (v.detach().to(dev) if dev is not None else v.detach()) for k, v in state_dict.items() if torch.is_tensor(v)}
        p = self.p
        self.device = p['encoder.embedding.weight'].device
        self.n_mels = p['decoder.prenet.fc1.weight'].shape[1]
        self.decoder_dims = p['decoder.attn_rnn.weight_hh'].shape[1]
        self.lstm_dims = p['decoder.res_rnn1.weight_hh'].shape[1]
        self.max_r = p['decoder.mel_proj.weight'].shape[0] // self.n_mels
        self.r = int(p['decoder.r'].item()) if 'decoder.r' in p else int(p.get('r', torch.tensor(1)).item())
        self.stop_threshold = float(p['stop_threshold'].item()) if 'stop_threshold' in p else -3.4
        self._enc_k = self._count('encoder.cbhg.conv1d_bank.%d.conv.weight')
        self._post_k = self._count('postnet.conv1d_bank.%d.conv.weight')

    def _count(self, pattern):
        n = 0
        while pattern % n in self.p:
            n += 1
        return n

    # -- building blocks (eval mode) ----------------------------------------------------------------------------------
    def _bnconv(self, x, prefix, relu=True):
        """BatchNormConv (models/tacotron.py:43-54): conv (same padding) -> [relu] -> batch norm (running statistics)."""
        w = self.p[prefix + '.conv.weight']
        x = F.conv1d(x, w, None, 1, w.shape[2] // 2)
        if relu:
            x = F.relu(x)
        q = prefix + '.bnorm.'
        return F.batch_norm(x, self.p[q + 'running_mean'], self.p[q + 'running_var'], self.p[q + 'weight'], self.p[q + 'bias'], False, 0.0, 1e-5)

    def _prenet(self, x, prefix):
        """PreNet (:141-155) in eval mode: two linear + relu layers, dropout off."""
        x = F.relu(F.linear(x, self.p[prefix + '.fc1.weight'], self.p[prefix + '.fc1.bias']))
        return F.relu(F.linear(x, self.p[prefix + '.fc2.weight'], self.p[prefix + '.fc2.bias']))

    def _cbhg(self, x, prefix, K):
        """CBHG (:57-139): conv bank 1..K -> max pool -> two projections -> residual -> highways -> bidirectional GRU."""
        p = self.p
        n = x.size(-1)
        res = x
        bank = torch.cat([self._bnconv(x, f'{prefix}.conv1d_bank.{k}')[:, :, :n] for k in range(K)], dim=1)
        x = F.max_pool1d(bank, 2, 1, 1)[:, :, :n]
        x = self._bnconv(x, prefix + '.conv_project1')
        x = self._bnconv(x, prefix + '.conv_project2', relu=False)
        x = (x + res).transpose(1, 2)
        if prefix + '.pre_highway.weight' in p:
            x = F.linear(x, p[prefix + '.pre_highway.weight'])
        h = 0
        while f'{prefix}.highways.{h}.W1.weight' in p:
            q = f'{prefix}.highways.{h}.'
            x1 = F.linear(x, p[q + 'W1.weight'], p[q + 'W1.bias'])
            g = torch.sigmoid(F.linear(x, p[q + 'W2.weight'], p[q + 'W2.bias']))
            x = g * F.relu(x1) + (1. - g) * x
            h += 1
        q = prefix + '.rnn.'
        flat = [p[q + n_] for n_ in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0', 'weight_ih_l0_reverse',
                                     'weight_hh_l0_reverse', 'bias_ih_l0_reverse', 'bias_hh_l0_reverse')]
        if getattr(self, '_bigru_kernel', False) and x.is_cuda and x.size(0) == 1 and flat[1].shape[1] == 128:
            return self._bigru(x, flat)
        hx = torch.zeros(2, x.size(0), flat[1].shape[1], device=x.device, dtype=x.dtype)
        out, _ = torch._VF.gru(x, hx, flat, True, 1, 0.0, False, True, True)
        return out

    def _bigru(self, x, flat):
        """The CBHG's bidirectional GRU (:95, :137) through `wrnn_bigru` (csrc/wrnn_taco.hip: one persistent workgroup per
        direction, W_hh in registers) instead of MIOpen's per-step launches; the input products are two plain GEMMs."""
        import ctypes
        from . import _lib
        L = _lib.lib()
        w_ih, w_hh, b_ih, b_hh, w_ih_r, w_hh_r, b_ih_r, b_hh_r = [t.contiguous() for t in flat]
        gi_f = F.linear(x[0], w_ih, b_ih).contiguous()
        gi_r = F.linear(x[0], w_ih_r, b_ih_r).contiguous()
        out = torch.empty(1, x.size(1), 256, device=x.device, dtype=torch.float32)
        c = _lib.BigruCall()
        c.struct_bytes = ctypes.sizeof(_lib.BigruCall)
        c.T, c.hidden = x.size(1), 128
        c.gi_fwd, c.gi_rev, c.w_hh_fwd, c.w_hh_rev = gi_f.data_ptr(), gi_r.data_ptr(), w_hh.data_ptr(), w_hh_r.data_ptr()
        c.b_hh_fwd, c.b_hh_rev, c.out = b_hh.data_ptr(), b_hh_r.data_ptr(), out.data_ptr()
        c.stream = torch.cuda.current_stream(x.device).cuda_stream
        rc = L.wrnn_bigru((x.device.index if x.device.index is not None else torch.cuda.current_device()), ctypes.byref(c))
        if rc != _lib.WRNN_OK:
            raise _lib.WrnnError(f'wrnn_bigru failed (rc={rc}): {L.wrnn_taco_last_error().decode()}')
        return out

    def encode(self, ids):
        """Encoder (:24-39) + `encoder_proj` (:403-404): ids (n,) -> encoder_seq (1, n, 2C), its projection (1, n, D)."""
        x = torch.as_tensor(ids, dtype=torch.long, device=self.device).unsqueeze(0)
        x = F.embedding(x, self.p['encoder.embedding.weight'])
        x = self._prenet(x, 'encoder.pre_net').transpose(1, 2)
        seq = self._cbhg(x, 'encoder.cbhg', self._enc_k)
        return seq, F.linear(seq, self.p['encoder_proj.weight'])

    def _decoder_step(self, seq, seq_proj, prenet_in, st):
        """Decoder.forward (:218-279) in eval mode with the LSA attention (:181-207); `st` holds the recurrent tensors and is
        updated IN PLACE."""
        p = self.p
        q = 'decoder.'
        pre = self._prenet(prenet_in, q + 'prenet')
        attn_h = torch.gru_cell(torch.cat([st['context'], pre], dim=-1), st['attn_h'], p[q + 'attn_rnn.weight_ih'], p[q + 'attn_rnn.weight_hh'],
                                p[q + 'attn_rnn.bias_ih'], p[q + 'attn_rnn.bias_hh'])
        # location-sensitive attention with sigmoid-normalised ("smooth") scores
        pq = F.linear(attn_h, p[q + 'attn_net.W.weight'], p[q + 'attn_net.W.bias']).unsqueeze(1)
        loc = torch.cat([st['cumulative'].unsqueeze(1), st['attention'].unsqueeze(1)], dim=1)
        kw = p[q + 'attn_net.conv.weight']
        ploc = F.linear(F.conv1d(loc, kw, None, 1, (kw.shape[2] - 1) // 2).transpose(1, 2), p[q + 'attn_net.L.weight'], p[q + 'attn_net.L.bias'])
        u = F.linear(torch.tanh(pq + seq_proj + ploc), p[q + 'attn_net.v.weight']).squeeze(-1)
        scores = torch.sigmoid(u) / torch.sigmoid(u).sum(dim=1, keepdim=True)
        st['attention'].copy_(scores)
        st['cumulative'].add_(scores)
        context = (scores.unsqueeze(-1).transpose(1, 2) @ seq).squeeze(1)
        x = F.linear(torch.cat([context, attn_h], dim=1), p[q + 'rnn_input.weight'], p[q + 'rnn_input.bias'])
        h1, c1 = torch.lstm_cell(x, (st['h1'], st['c1']), p[q + 'res_rnn1.weight_ih'], p[q + 'res_rnn1.weight_hh'], p[q + 'res_rnn1.bias_ih'],
                                 p[q + 'res_rnn1.bias_hh'])
        x = x + h1
        h2, c2 = torch.lstm_cell(x, (st['h2'], st['c2']), p[q + 'res_rnn2.weight_ih'], p[q + 'res_rnn2.weight_hh'], p[q + 'res_rnn2.bias_ih'],
                                 p[q + 'res_rnn2.bias_hh'])
        x = x + h2
        mels = F.linear(x, p[q + 'mel_proj.weight']).view(x.size(0), self.n_mels, self.max_r)[:, :, :self.r]
        for k, v in (('attn_h', attn_h), ('context', context), ('h1', h1), ('c1', c1), ('h2', h2), ('c2', c2)):
            st[k].copy_(v)
        return mels, scores

    def _decode_kernel(self, seq, seq_proj, steps, variant=0):
        """The decoder loop (:396-414) as ONE persistent HIP kernel (csrc/wrnn_taco.hip) through the C ABI
        (`wrnn_taco_decode`; variant 0 = auto, 1 = the flag-barrier kernel, 2 = the register-resident kernel).  Returns (mel (1, n_mels, N) , attention (N / r, n)) device tensors; fails loudly without the
        extension or a HIP device -- there is no host fallback."""
        import ctypes
        from . import _lib
        dev = self.device
        if dev.type != 'cuda':
            raise _lib.WrnnError('the Tacotron decoder kernel needs a HIP device (no CPU fallback)')
        L = _lib.lib()
        p, q = self.p, 'decoder.'
        names = dict(prenet_fc1_w='prenet.fc1.weight', prenet_fc1_b='prenet.fc1.bias', prenet_fc2_w='prenet.fc2.weight',
                     prenet_fc2_b='prenet.fc2.bias', attn_rnn_w_ih='attn_rnn.weight_ih', attn_rnn_w_hh='attn_rnn.weight_hh',
                     attn_rnn_b_ih='attn_rnn.bias_ih', attn_rnn_b_hh='attn_rnn.bias_hh', attn_W_w='attn_net.W.weight',
                     attn_W_b='attn_net.W.bias', attn_conv_w='attn_net.conv.weight', attn_L_w='attn_net.L.weight',
                     attn_L_b='attn_net.L.bias', attn_v_w='attn_net.v.weight', rnn_input_w='rnn_input.weight',
                     rnn_input_b='rnn_input.bias', rnn1_w_ih='res_rnn1.weight_ih', rnn1_w_hh='res_rnn1.weight_hh',
                     rnn1_b_ih='res_rnn1.bias_ih', rnn1_b_hh='res_rnn1.bias_hh', rnn2_w_ih='res_rnn2.weight_ih',
                     rnn2_w_hh='res_rnn2.weight_hh', rnn2_b_ih='res_rnn2.bias_ih', rnn2_b_hh='res_rnn2.bias_hh',
                     mel_proj_w='mel_proj.weight')
        keep = {k: p[q + v].detach().to(dev, torch.float32).contiguous() for k, v in names.items()}
        w = _lib.TacoWeights()
        w.struct_bytes = ctypes.sizeof(_lib.TacoWeights)
        w.n_mels, w.prenet1, w.prenet2 = self.n_mels, keep['prenet_fc1_w'].shape[0], keep['prenet_fc2_w'].shape[0]
        w.decoder_dims, w.encoder_width, w.lstm_dims = self.decoder_dims, seq.size(2), self.lstm_dims
        w.attn_filters, w.attn_kernel = keep['attn_conv_w'].shape[0], keep['attn_conv_w'].shape[2]
        for k, tns in keep.items():
            setattr(w, k, tns.data_ptr())
        n = seq.size(1)
        max_steps = (steps + self.r - 1) // self.r
        seq_c, proj_c = seq[0].contiguous(), seq_proj[0].contiguous()
        mel_out = torch.zeros(max_steps, self.n_mels, self.r, device=dev)
        scores = torch.zeros(max_steps, n, device=dev)
        done = torch.zeros(1, dtype=torch.int32, device=dev)
        ws = torch.empty(int(L.wrnn_taco_workspace_bytes()), dtype=torch.uint8, device=dev)
        c = _lib.TacoCall()
        c.struct_bytes = ctypes.sizeof(_lib.TacoCall)
        c.n, c.r, c.max_r, c.max_steps, c.stop_threshold = n, self.r, self.max_r, max_steps, self.stop_threshold
        c.seq, c.seq_proj, c.mel_out, c.scores_out = seq_c.data_ptr(), proj_c.data_ptr(), mel_out.data_ptr(), scores.data_ptr()
        c.steps_done, c.workspace, c.workspace_bytes = done.data_ptr(), ws.data_ptr(), ws.numel()
        c.stream = torch.cuda.current_stream(dev).cuda_stream
        c.variant = int(variant)
        rc = L.wrnn_taco_decode((dev.index if dev.index is not None else torch.cuda.current_device()), ctypes.byref(w), ctypes.byref(c))
        if rc != _lib.WRNN_OK:
            raise _lib.WrnnError(f'wrnn_taco_decode failed (rc={rc}): {L.wrnn_taco_last_error().decode()}')
        st4 = (ctypes.c_uint32 * 4)()
        rc = L.wrnn_taco_status(ws.data_ptr(), ctypes.byref(st4), c.stream)
        if rc != _lib.WRNN_OK or st4[0] != 0:
            raise _lib.WrnnError(f'Tacotron decoder kernel failed: status {list(st4)} {L.wrnn_taco_last_error().decode()}')
        self._last_taco_ws = ws                                                    # (variant 3: the profile words are its last 192 bytes)
        k = int(done.item())
        mel = mel_out[:k].permute(1, 0, 2).reshape(1, self.n_mels, k * self.r)      # frames of step s at columns [s r, (s+1) r)
        return mel, scores[:k]

    @torch.no_grad()
    def generate(self, ids, steps=2000, kernel=False, kernel_variant=0):
        """`Tacotron.generate(x, steps)` (:370-430).  Returns numpy (mel (n_mels, N), linear (fft, N), attention (N, n_chars)).

        kernel=True (HIP device): the decoder loop runs as ONE persistent kernel (`wrnn_taco_decode`, csrc/wrnn_taco.hip; the stop
        test of :411 is evaluated inside it) and the CBHGs' bidirectional GRUs as `wrnn_bigru`.  kernel=False: the eager loop (any
        device; the CPU form is the mirror tests/test_tacotron_mirror.py pins bit-exactly to the reference).  (Round 2's HIP-graph
        replay of one decoder step -- 555 us per step against the kernel's 27.6 -- is gone; its numbers are in profiles/r03*.)"""
        dev = self.device
        self._bigru_kernel = bool(kernel) and dev.type == 'cuda'                   # the CBHGs' GRUs as persistent kernels too
        seq, seq_proj = self.encode(ids)
        n = seq.size(1)
        z = lambda *s: torch.zeros(*s, device=dev)
        st = dict(attn_h=z(1, self.decoder_dims), h1=z(1, self.lstm_dims), h2=z(1, self.lstm_dims), c1=z(1, self.lstm_dims),
                  c2=z(1, self.lstm_dims), context=z(1, self.decoder_dims), cumulative=z(1, n), attention=z(1, n))
        prenet_in = z(1, self.n_mels)                                              # the <GO> frame
        frames, scores_all = [], []
        if kernel:
            mel, scores_all = self._decode_kernel(seq, seq_proj, steps, kernel_variant)
            frames = [mel]
        else:
            for t in range(0, steps, self.r):
                m_, s_ = self._decoder_step(seq, seq_proj, prenet_in, st)
                frames.append(m_)
                scores_all.append(s_.clone())
                prenet_in = m_[:, :, -1]
                if (m_ < self.stop_threshold).all() and t > 10:
                    break
        mel = torch.cat(frames, dim=2)
        post = self._cbhg(mel, 'postnet', self._post_k)
        linear = F.linear(post, self.p['post_proj.weight']).transpose(1, 2)[0]
        attn = scores_all if kernel else torch.cat([s.unsqueeze(-1).transpose(1, 2) for s in scores_all], 1)[0]
        return mel[0].cpu().numpy(), linear.cpu().numpy(), attn.cpu().numpy()


def tacotron_to_wavernn_mel(mel):
    """gen_tacotron.py:143-145: rescale the Tacotron mel from [-4, 4] to [0, 1] and clip."""
    m = (np.asarray(mel) + 4) / 8
    return np.clip(m, 0, 1, out=m)
