"""Constructor variants of StofNet (models/stofnet.py:11 takes any num_blocks / kernel_sizes / semi_global_scale) used by
tests/golden/make_golden_r4b.py (reference side) and by the parity tests (oracle and gfx950 side): the geometry table and a
numpy-seeded parameter generator, so the fixture holds inputs' seeds and expected outputs only, not the weights."""
import numpy as np

# name -> constructor arguments, input shape, which gradients the fixture keeps ('all' or a tuple of parameter names)
VARIANTS = {
    # even num_blocks (the loop of :52 ends on a leaky-ReLU layer), 5-tap body, SemiGlobalBlock at scale 20 (128 channels)
    'nb8_k5_sgs20_r4': dict(ctor=dict(upsample_factor=4, num_blocks=8, kernel_sizes=[9, 5, 3], semi_global_scale=20),
                            N=3, L=400, grads='all'),
    # the smallest odd num_blocks with a residual layer in the loop, 3-tap body, no SemiGlobalBlock, r = 10
    'nb5_k3_nosgb_r10': dict(ctor=dict(upsample_factor=10, num_blocks=5, kernel_sizes=[9, 3, 3], semi_global_scale=1),
                             N=2, L=250, grads='all'),
    # the smallest num_blocks the reference's forward accepts (:60 needs one pass of the loop)
    'nb4_k7_sgs80_r4': dict(ctor=dict(upsample_factor=4, num_blocks=4, kernel_sizes=[9, 7, 3], semi_global_scale=80),
                            N=2, L=244, grads=('conv1.weight', 'conv2.weight', 'conv3.bias', 'conv_last.weight')),
    # deeper than the shipped network, 7-tap body: the fused sweeps do not serve it
    'nb14_k7_nosgb_r4': dict(ctor=dict(upsample_factor=4, num_blocks=14, kernel_sizes=[9, 7, 3], semi_global_scale=1),
                             N=2, L=300, grads=('conv1.weight', 'conv2.weight', 'conv13.weight', 'conv_last.bias')),
}


def variant_params(shapes: dict, seed: int) -> dict:
    """name -> float32 array for every entry of `shapes` (a state_dict's name -> shape, in its own order): weights
    N(0, 1 / fan_in) so activations keep their scale through the stack, biases N(0, 0.05^2)."""
    rng = np.random.RandomState(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(int(s) for s in shape)
        if name.endswith('.weight'):
            fan_in = shape[1] * shape[2]
            out[name] = (rng.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)
        else:
            out[name] = (0.05 * rng.standard_normal(shape)).astype(np.float32)
    return out


def variant_input(N: int, L: int, seed: int):
    """(x [N, 1, L], t [N, 1, L*r] is made by the caller from the second array's generator): a few damped echoes + noise."""
    rng = np.random.RandomState(seed)
    t = np.arange(L, dtype=np.float64)
    x = 0.02 * rng.standard_normal((N, 1, L))
    for n in range(N):
        for _ in range(3):
            c, f, w = rng.uniform(0.1 * L, 0.9 * L), rng.uniform(0.05, 0.3), rng.uniform(4, 15)
            x[n, 0] += rng.uniform(0.3, 1.0) * np.exp(-0.5 * ((t - c) / w) ** 2) * np.sin(2 * np.pi * f * (t - c))
    return x.astype(np.float32)


def variant_cotangent(N: int, M: int, seed: int):
    """dL/dy for the gradient check: loss = sum(y * t)."""
    return np.random.RandomState(seed + 1000).standard_normal((N, 1, M)).astype(np.float32)


# standalone SemiGlobalBlock(in_channels, out_channels, sample_scale, kernel_size) variants (models/stofnet.py:80-117): L, seed
SGB_VARIANTS = {
    'c32_s20_k3': dict(ctor=(32, 32, 20, 3), N=2, L=244),       # remainder 4: the up-sampled map is shifted by 2 (Q2)
    'c96_s5_k7': dict(ctor=(96, 96, 5, 7), N=2, L=200),         # feat_scale = max(1, 5 // 10) = 1
    'c64_s30_k9': dict(ctor=(64, 64, 30, 9), N=1, L=300),       # 192 contracted channels, the widest kernel the layer kernels take
    'c8_s40_k5': dict(ctor=(8, 8, 40, 5), N=3, L=160),          # rows narrower than one 64-channel block
    'c1_s2_k1': dict(ctor=(1, 1, 2, 1), N=2, L=64),
}


def sgb_input(N, C, L, seed):
    return np.random.RandomState(seed).standard_normal((N, C, L)).astype(np.float32)
