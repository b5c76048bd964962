"""Seeded synthetic inputs (weights, mel spectrograms) for tests, golden fixtures and bench.py.

The reference's pretrained weights are not available (`.MISSING_LARGE_BLOBS`), so every parity and
throughput run uses random-init weights.  They are drawn with numpy's MT19937 `RandomState` so the
same `state_dict` can be rebuilt bit-identically on any box without shipping 15 MB blobs, using the
distributions of torch's default initialisers (U(-1/sqrt(fan), 1/sqrt(fan))) and the state-dict keys
of `WaveRNN` (reference `models/fatchord_version.py:92-129`).
"""
import numpy as np

SHIPPED = dict(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(5, 5, 11), feat_dims=80,
               compute_dims=128, res_out_dims=128, res_blocks=10, hop_length=275, sample_rate=22050)


def random_state_dict(seed, mode='MOL', rnn_dims=512, fc_dims=512, bits=9, pad=2,
                      upsample_factors=(5, 5, 11), feat_dims=80, compute_dims=128, res_out_dims=128,
                      res_blocks=10, nontrivial_bn=True, **_):
    """Return {key: np.ndarray} with the exact key set / shapes of the reference `WaveRNN.state_dict()`."""
    rs = np.random.RandomState(seed)
    f32 = np.float32

    def U(shape, fan):
        b = 1.0 / np.sqrt(fan)
        return rs.uniform(-b, b, size=shape).astype(f32)

    sd = {}
    sd['step'] = np.zeros((1,), np.int64)
    k = 2 * pad + 1

    def bn(prefix, c):
        if nontrivial_bn:
            sd[prefix + '.weight'] = rs.uniform(0.5, 1.5, c).astype(f32)
            sd[prefix + '.bias'] = rs.uniform(-0.2, 0.2, c).astype(f32)
            sd[prefix + '.running_mean'] = rs.uniform(-0.2, 0.2, c).astype(f32)
            sd[prefix + '.running_var'] = rs.uniform(0.5, 1.5, c).astype(f32)
        else:
            sd[prefix + '.weight'] = np.ones(c, f32)
            sd[prefix + '.bias'] = np.zeros(c, f32)
            sd[prefix + '.running_mean'] = np.zeros(c, f32)
            sd[prefix + '.running_var'] = np.ones(c, f32)
        sd[prefix + '.num_batches_tracked'] = np.zeros((), np.int64)

    sd['upsample.resnet.conv_in.weight'] = U((compute_dims, feat_dims, k), feat_dims * k)
    bn('upsample.resnet.batch_norm', compute_dims)
    for i in range(res_blocks):
        p = f'upsample.resnet.layers.{i}'
        sd[p + '.conv1.weight'] = U((compute_dims, compute_dims, 1), compute_dims)
        sd[p + '.conv2.weight'] = U((compute_dims, compute_dims, 1), compute_dims)
        bn(p + '.batch_norm1', compute_dims)
        bn(p + '.batch_norm2', compute_dims)
    sd['upsample.resnet.conv_out.weight'] = U((res_out_dims, compute_dims, 1), compute_dims)
    sd['upsample.resnet.conv_out.bias'] = U((res_out_dims,), compute_dims)
    for li, s in enumerate(upsample_factors):
        w = np.full((1, 1, 1, 2 * s + 1), 1.0 / (2 * s + 1), f32)
        if nontrivial_bn:   # a trained model's box filters are no longer exactly uniform
            w = (w * rs.uniform(0.9, 1.1, w.shape)).astype(f32)
        sd[f'upsample.up_layers.{2 * li + 1}.weight'] = w
    aux = res_out_dims // 4
    n_classes = 2 ** bits if mode == 'RAW' else 30
    sd['I.weight'] = U((rnn_dims, feat_dims + aux + 1), feat_dims + aux + 1)
    sd['I.bias'] = U((rnn_dims,), feat_dims + aux + 1)
    for name, inp in (('rnn1', rnn_dims), ('rnn2', rnn_dims + aux)):
        sd[f'{name}.weight_ih_l0'] = U((3 * rnn_dims, inp), rnn_dims)
        sd[f'{name}.weight_hh_l0'] = U((3 * rnn_dims, rnn_dims), rnn_dims)
        sd[f'{name}.bias_ih_l0'] = U((3 * rnn_dims,), rnn_dims)
        sd[f'{name}.bias_hh_l0'] = U((3 * rnn_dims,), rnn_dims)
    sd['fc1.weight'] = U((fc_dims, rnn_dims + aux), rnn_dims + aux)
    sd['fc1.bias'] = U((fc_dims,), rnn_dims + aux)
    sd['fc2.weight'] = U((fc_dims, fc_dims + aux), fc_dims + aux)
    sd['fc2.bias'] = U((fc_dims,), fc_dims + aux)
    sd['fc3.weight'] = U((n_classes, fc_dims), fc_dims)
    sd['fc3.bias'] = U((n_classes,), fc_dims)
    if mode == 'MOL':
        # a trained MoL head predicts narrow logistics; random init gives scales ~1 and >50 % of the samples
        # clamp at +-1, which would blunt every parity check.  Shift the 10 log-scale rows so samples spread
        # over (-1, 1).
        sd['fc3.bias'][20:30] -= np.float32(3.0)
    return sd


def random_mel(seed, n_frames, n_mels=80):
    """U[0,1) mel of shape (n_mels, n_frames): satisfies the range check of `gen_wavernn.py:52-55`."""
    return np.random.RandomState(seed).uniform(0.0, 1.0, size=(n_mels, n_frames)).astype(np.float32)


def random_tacotron_state_dict(seed, shapes):
    """Seeded random-init Tacotron state dict from a (key, shape, dtype) table (tests/golden/tacotron_shapes.json, written by
    scripts/make_golden.py from the reference's module): Xavier-uniform matrices like `Tacotron.init_model` (models/tacotron.py:
    432-434), zero biases, identity batch norms.  Architecture-only stand-in for the absent pretrained checkpoint."""
    import torch
    rs = np.random.RandomState(seed)
    sd = {}
    for key, shape, dtype in shapes:
        if dtype != 'float32':
            val = {'decoder.r': 1, 'step': 0}.get(key, 0)
            sd[key] = torch.full(shape, val, dtype=getattr(torch, dtype))
        elif key.endswith('running_var') or (key.endswith('.weight') and 'bnorm' in key):
            sd[key] = torch.ones(shape)
        elif key == 'stop_threshold':
            sd[key] = torch.tensor(-3.4)
        elif len(shape) >= 2:
            fan_out, fan_in = shape[0] * int(np.prod(shape[2:])), shape[1] * int(np.prod(shape[2:]))
            a = np.sqrt(6.0 / (fan_in + fan_out))
            sd[key] = torch.from_numpy(rs.uniform(-a, a, shape).astype(np.float32))
        else:
            sd[key] = torch.zeros(shape)
    return sd
