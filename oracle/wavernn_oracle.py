"""numpy restatement of the reference hot path -- TEST INFRASTRUCTURE, never shipped / never measured.

Every function cites the reference `file:line` (relative to the reference repo root) it restates.
Parity pinning: the reference has no tests or golden vectors of its own, so this
oracle is pinned against outputs of the *reference itself* run in the build container
(`scripts/make_golden.py` -> `tests/golden/*.npz`, checked by `tests/test_oracle_golden.py`).

Numeric type: float32 everywhere the reference computes on the device, float64 after the gather
(`models/fatchord_version.py:245`), exactly as the reference does.
"""
import math
import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------------------------------
# RNG: the reference samples from torch's global CPU generator (MT19937).  SURVEY.md Appendix B.
# --------------------------------------------------------------------------------------------------

#: draws burnt by the two `nn.GRUCell` constructors in `get_gru_cell` (fatchord_version.py:178-179,
#: :273-279): GRUCell(512,512) = 3*512*512*2 + 2*3*512, GRUCell(544,512) = 3*512*(544+512) + 2*3*512.
def gru_cell_ctor_draws(rnn_dims=512, aux_dims=32):
    h = rnn_dims
    c1 = 3 * h * h + 3 * h * h + 2 * 3 * h
    c2 = 3 * h * (h + aux_dims) + 3 * h * h + 2 * 3 * h
    return c1 + c2


class TorchCpuStream:
    """Bit-level model of `torch.manual_seed(s)` + CPU `uniform_` / `exponential_` (float32).

    torch's CPU generator is MT19937 whose 32-bit outputs equal numpy's `RandomState(s)` raw stream.
    """

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)

    def raw32(self, n):
        return self.rs.randint(0, 2 ** 32, size=n, dtype=np.uint64)

    def skip(self, n):
        # draw and discard in chunks
        while n > 0:
            m = min(n, 1 << 20)
            self.rs.randint(0, 2 ** 32, size=m, dtype=np.uint64)
            n -= m

    def uniform_(self, n, a, b):
        """`Tensor.uniform_(a, b)` float32: one 32-bit draw per element, 24 random bits,
        `float32(u * double(float32(b) - float32(a)) + double(float32(a)))`."""
        r = self.raw32(n)
        u = (r & np.uint64((1 << 24) - 1)).astype(np.float64) * (2.0 ** -24)
        fa, fb = F32(a), F32(b)
        return (u * np.float64(F32(fb - fa)) + np.float64(fa)).astype(F32)

    def exponential_(self, n):
        """`Tensor.exponential_(1)` float32: two 32-bit draws (hi, lo), 53 random bits,
        `float32(-log1p(-u))` in double."""
        r = self.raw32(2 * n)
        hi = r[0::2]
        lo = r[1::2]
        r64 = (hi << np.uint64(32)) | lo
        u = (r64 & np.uint64((1 << 53) - 1)).astype(np.float64) * (2.0 ** -53)
        return (-np.log1p(-u)).astype(F32)


def draw_noise(seed, mode, B, T, rnn_dims=512, aux_dims=32, n_classes=512):
    """Noise consumed by `generate()` after `torch.manual_seed(seed)`.

    MOL: returns (u1[T,B,10], u2[T,B]) -- `distribution.py:106,118`, both uniform_(1e-5, 1-1e-5).
    RAW: returns q[T,B,n_classes]      -- `Categorical.sample` -> multinomial -> exponential_(1).
    """
    st = TorchCpuStream(seed)
    st.skip(gru_cell_ctor_draws(rnn_dims, aux_dims))
    if mode == 'MOL':
        u = st.uniform_(T * B * 11, 1e-5, 1.0 - 1e-5).reshape(T, B * 11)
        u1 = u[:, :B * 10].reshape(T, B, 10)
        u2 = u[:, B * 10:].reshape(T, B)
        return u1, u2
    q = st.exponential_(T * B * n_classes).reshape(T, B, n_classes)
    return q


# --------------------------------------------------------------------------------------------------
# fold / unfold / dsp  (host-side pieces of the path)
# --------------------------------------------------------------------------------------------------

def pad_tensor(x, pad, side='both'):
    """`WaveRNN.pad_tensor` (fatchord_version.py:281-291).  x: (b, t, c)."""
    b, t, c = x.shape
    total = t + 2 * pad if side == 'both' else t + pad
    padded = np.zeros((b, total, c), dtype=x.dtype)
    if side == 'before' or side == 'both':
        padded[:, pad:pad + t, :] = x
    elif side == 'after':
        padded[:, :t, :] = x
    return padded


def num_folds(total_len, target, overlap):
    """fold count of `fold_with_overlap` (fatchord_version.py:322-330)."""
    nf = (total_len - overlap) // (target + overlap)
    ext = nf * (overlap + target) + overlap
    if total_len - ext != 0:
        nf += 1
    return nf


def fold_with_overlap(x, target, overlap):
    """`WaveRNN.fold_with_overlap` (fatchord_version.py:293-340).  x: (1, L, F) -> (B, T, F)."""
    _, total_len, features = x.shape
    nf = (total_len - overlap) // (target + overlap)
    extended_len = nf * (overlap + target) + overlap
    remaining = total_len - extended_len
    if remaining != 0:
        nf += 1
        padding = target + 2 * overlap - remaining
        x = pad_tensor(x, padding, side='after')
    folded = np.zeros((nf, target + 2 * overlap, features), dtype=x.dtype)
    for i in range(nf):
        start = i * (target + overlap)
        end = start + target + 2 * overlap
        folded[i] = x[0, start:end, :]
    return folded


def xfade_and_unfold(y, target, overlap):
    """`WaveRNN.xfade_and_unfold` (fatchord_version.py:342-405).  y: (B, T) float64, mutated in place
    like the reference (:394-395); the `target` argument is ignored and recomputed (:375)."""
    nf, length = y.shape
    target = length - 2 * overlap
    total_len = nf * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    silence = np.zeros((silence_len), dtype=np.float64)
    linear = np.ones((silence_len), dtype=np.float64)
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.sqrt(0.5 * (1 + t))
    fade_out = np.sqrt(0.5 * (1 - t))
    fade_in = np.concatenate([silence, fade_in])
    fade_out = np.concatenate([linear, fade_out])
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    unfolded = np.zeros((total_len), dtype=np.float64)
    for i in range(nf):
        start = i * (target + overlap)
        end = start + target + 2 * overlap
        unfolded[start:end] += y[i]
    return unfolded


def label_2_float(x, bits):
    """`utils/dsp.py:8-9`."""
    return 2 * x / (2 ** bits - 1.) - 1.


def decode_mu_law(y, mu, from_labels=True):
    """`utils/dsp.py:98-103`."""
    if from_labels:
        y = label_2_float(y, math.log2(mu))
    mu = mu - 1
    return np.sign(y) / mu * ((1 + mu) ** np.abs(y) - 1)


# --------------------------------------------------------------------------------------------------
# pre-loop: UpsampleNetwork (fatchord_version.py:13-89), eval mode
# --------------------------------------------------------------------------------------------------

def _bn(x, sd, prefix, eps=1e-5):
    """eval-mode BatchNorm1d on (C, N): (x-mean)/sqrt(var+eps)*w+b  (fatchord_version.py:18-19,36)."""
    w = sd[prefix + '.weight'].astype(F32)[:, None]
    b = sd[prefix + '.bias'].astype(F32)[:, None]
    m = sd[prefix + '.running_mean'].astype(F32)[:, None]
    v = sd[prefix + '.running_var'].astype(F32)[:, None]
    return ((x - m) / np.sqrt(v + F32(eps)) * w + b).astype(F32)


def mel_resnet(sd, m):
    """`MelResNet.forward` (fatchord_version.py:42-48).  m: (80, N+2*pad) -> (res_out, N)."""
    w = sd['upsample.resnet.conv_in.weight'].astype(F32)          # (C, 80, k)
    C, Fin, k = w.shape
    n = m.shape[1] - (k - 1)
    x = np.zeros((C, n), dtype=F32)
    for j in range(k):
        x += w[:, :, j] @ m[:, j:j + n]
    x = np.maximum(_bn(x, sd, 'upsample.resnet.batch_norm'), 0)
    i = 0
    while f'upsample.resnet.layers.{i}.conv1.weight' in sd:
        p = f'upsample.resnet.layers.{i}'
        r = x
        x = sd[p + '.conv1.weight'].astype(F32)[:, :, 0] @ x
        x = np.maximum(_bn(x, sd, p + '.batch_norm1'), 0)
        x = sd[p + '.conv2.weight'].astype(F32)[:, :, 0] @ x
        x = _bn(x, sd, p + '.batch_norm2')
        x = x + r
        i += 1
    x = sd['upsample.resnet.conv_out.weight'].astype(F32)[:, :, 0] @ x \
        + sd['upsample.resnet.conv_out.bias'].astype(F32)[:, None]
    return x.astype(F32)


def upsample_network(sd, m, upsample_factors=(5, 5, 11), pad=2):
    """`UpsampleNetwork.forward` (fatchord_version.py:82-89).

    m: (80, N+2*pad) zero-padded mel.  Returns mels (N*hop, 80), aux (N*hop, res_out)."""
    total = int(np.prod(upsample_factors))
    indent = pad * total
    aux = mel_resnet(sd, m)                                   # (R, N)
    aux = np.repeat(aux, total, axis=1)                       # Stretch2d(total,1)  (:57-61)
    x = m.astype(F32)
    for li, s in enumerate(upsample_factors):
        x = np.repeat(x, s, axis=1)                           # Stretch2d(s,1)
        w = sd[f'upsample.up_layers.{2 * li + 1}.weight'].astype(F32).reshape(-1)   # (2s+1,)
        xp = np.pad(x, ((0, 0), (s, s)))
        y = np.zeros_like(x)
        for j in range(2 * s + 1):                            # Conv2d = cross-correlation (:77)
            y += w[j] * xp[:, j:j + x.shape[1]]
        x = y.astype(F32)
    x = x[:, indent:-indent]
    return np.ascontiguousarray(x.T), np.ascontiguousarray(aux.T)


# --------------------------------------------------------------------------------------------------
# the loop  (fatchord_version.py:192-241 + utils/distribution.py:87-123)
# --------------------------------------------------------------------------------------------------

def _sigmoid(x):
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    """ATen CPU `gru_cell` (what `nn.GRUCell` runs at fatchord_version.py:210,214):
    r=sig(gh_r+gi_r) z=sig(gh_z+gi_z) n=tanh(gi_n + gh_n*r) h'=(h-n)*z+n, gate rows [r;z;n]."""
    H = h.shape[1]
    gi = (x @ w_ih.T + b_ih).astype(F32)
    gh = (h @ w_hh.T + b_hh).astype(F32)
    r = _sigmoid(gh[:, :H] + gi[:, :H])
    z = _sigmoid(gh[:, H:2 * H] + gi[:, H:2 * H])
    n = np.tanh(gi[:, 2 * H:] + gh[:, 2 * H:] * r, dtype=F32)
    return ((h - n) * z + n).astype(F32)


def sample_mol(logits, u1, u2, log_scale_min=None):
    """`sample_from_discretized_mix_logistic` (utils/distribution.py:87-123) for one step.
    logits (B,30); u1 (B,10); u2 (B,)."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))
    nr = logits.shape[1] // 3
    lp = logits[:, :nr]
    temp = (lp - np.log(-np.log(u1, dtype=F32), dtype=F32)).astype(F32)
    k = np.argmax(temp, axis=1)
    rows = np.arange(logits.shape[0])
    means = logits[rows, nr + k]
    ls = np.maximum(logits[rows, 2 * nr + k], F32(log_scale_min))
    x = means + np.exp(ls, dtype=F32) * (np.log(u2, dtype=F32) - np.log(F32(1.) - u2, dtype=F32))
    return np.clip(x, F32(-1), F32(1)).astype(F32), k


def sample_raw(logits, q):
    """`F.softmax` -> `Categorical(posterior).sample()` (fatchord_version.py:232-235):
    p=softmax; p/=p.sum(); idx=argmax(p/q), q~Exp(1).  Returns (sample, idx)."""
    mx = logits.max(axis=1, keepdims=True)
    e = np.exp(logits - mx, dtype=F32)
    p = (e / e.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    p = (p / p.sum(axis=1, keepdims=True, dtype=F32)).astype(F32)
    idx = np.argmax((p / q).astype(F32), axis=1)
    n_classes = logits.shape[1]
    sample = (F32(2) * idx.astype(F32) / F32(n_classes - 1.) - F32(1.)).astype(F32)
    return sample, idx


def loop(sd, mode, mels, aux, noise, collect=None):
    """The per-sample loop (fatchord_version.py:192-241).

    mels (B,T,80) aux (B,T,128) float32.  noise = (u1,u2) for MOL or q for RAW (see draw_noise).
    Returns out (B,T) float32 (the `torch.stack(output).transpose(0,1)` tensor of :243).
    `collect`, if a dict, receives per-step teacher-forcing records."""
    g = lambda k: np.ascontiguousarray(sd[k], dtype=F32)
    W_I, b_I = g('I.weight'), g('I.bias')
    w_ih1, w_hh1, b_ih1, b_hh1 = g('rnn1.weight_ih_l0'), g('rnn1.weight_hh_l0'), g('rnn1.bias_ih_l0'), g('rnn1.bias_hh_l0')
    w_ih2, w_hh2, b_ih2, b_hh2 = g('rnn2.weight_ih_l0'), g('rnn2.weight_hh_l0'), g('rnn2.bias_ih_l0'), g('rnn2.bias_hh_l0')
    W1, b1, W2, b2, W3, b3 = g('fc1.weight'), g('fc1.bias'), g('fc2.weight'), g('fc2.bias'), g('fc3.weight'), g('fc3.bias')
    B, T, _ = mels.shape
    H = w_hh1.shape[1]
    d = aux.shape[2] // 4
    h1 = np.zeros((B, H), F32)
    h2 = np.zeros((B, H), F32)
    x = np.zeros((B, 1), F32)
    out = np.zeros((B, T), F32)
    if collect is not None:
        collect.update(h1=[], h2=[], logits=[], idx=[])
    for i in range(T):
        m_t = mels[:, i, :]
        a1, a2, a3, a4 = (aux[:, i, d * k:d * (k + 1)] for k in range(4))
        xx = (np.concatenate([x, m_t, a1], axis=1) @ W_I.T + b_I).astype(F32)
        h1 = gru_cell(xx, h1, w_ih1, w_hh1, b_ih1, b_hh1)
        xx = xx + h1
        h2 = gru_cell(np.concatenate([xx, a2], axis=1), h2, w_ih2, w_hh2, b_ih2, b_hh2)
        xx = xx + h2
        xx = np.maximum(np.concatenate([xx, a3], axis=1) @ W1.T + b1, 0).astype(F32)
        xx = np.maximum(np.concatenate([xx, a4], axis=1) @ W2.T + b2, 0).astype(F32)
        logits = (xx @ W3.T + b3).astype(F32)
        if mode == 'MOL':
            s, k = sample_mol(logits, noise[0][i], noise[1][i])
        elif mode == 'RAW':
            s, k = sample_raw(logits, noise[i])
        else:
            raise RuntimeError("Unknown model mode value - ", mode)
        out[:, i] = s
        x = s[:, None]
        if collect is not None:
            collect['h1'].append(h1.copy()); collect['h2'].append(h2.copy())
            collect['logits'].append(logits.copy()); collect['idx'].append(k.copy())
    return out


def conditioning(sd, mel, batched, target, overlap, upsample_factors=(5, 5, 11), pad=2):
    """Pre-loop stage of `generate` (fatchord_version.py:183-190).  mel: (80, N) float32.
    Returns folded mels (B,T,80), aux (B,T,128), wave_len."""
    hop = int(np.prod(upsample_factors))
    wave_len = (mel.shape[1] - 1) * hop
    m = pad_tensor(mel.T[None].astype(F32), pad, 'both')[0].T       # (80, N+2pad)
    mels, aux = upsample_network(sd, m, upsample_factors, pad)
    mels, aux = mels[None], aux[None]
    if batched:
        mels = fold_with_overlap(mels, target, overlap)
        aux = fold_with_overlap(aux, target, overlap)
    return mels, aux, wave_len


def finish(out, mode, n_classes, wave_len, batched, target, overlap, mu_law, hop=275):
    """Post-loop stage of `generate` (fatchord_version.py:243-258), float64."""
    output = out.astype(np.float64)
    mu_law = mu_law if mode == 'RAW' else False
    if mu_law:
        output = decode_mu_law(output, n_classes, False)
    if batched:
        output = xfade_and_unfold(output, target, overlap)
    else:
        output = output[0]
    fade_out = np.linspace(1, 0, 20 * hop)
    output = output[:wave_len]
    output[-20 * hop:] *= fade_out
    return output


def generate(sd, mode, mel, batched, target, overlap, mu_law, seed, bits=9, return_raw=False):
    """`WaveRNN.generate` (fatchord_version.py:169-264) minus the WAV write, under torch.manual_seed(seed)."""
    n_classes = 2 ** bits if mode == 'RAW' else 30
    mels, aux, wave_len = conditioning(sd, mel, batched, target, overlap)
    B, T, _ = mels.shape
    noise = draw_noise(seed, mode, B, T, n_classes=n_classes)
    out = loop(sd, mode, mels, aux, noise)
    res = finish(out.copy(), mode, n_classes, wave_len, batched, target, overlap, mu_law)
    return (res, out) if return_raw else res
