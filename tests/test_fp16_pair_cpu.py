"""Host-side check of the fp16-pair operand scheme of the denoiser convs: a numpy restatement
of what conv_bf3_kernel<SCH = 1> and unet.hip:pack_conv do with the operands -- power-of-two scales from the exponent
bits, a s = h + l in fp16, the three product terms -- against an fp64 dot product.  The GPU side is covered by the
parity suite and tests/test_gpu_full_size.py::test_fp16_pair_convs_agree_with_the_bf16_triple_convs."""
import numpy as np


def act_scale(amax_bits):
    """conv_bf3_kernels.h: exponent field of the group's largest |value| -> (input scale, its inverse's field base)."""
    ef = int(amax_bits) >> 23
    ef = min(max(ef, 87), 167)
    sf = 268 - ef
    return np.uint32(sf << 23).view(np.float32), sf


def weight_exp(w):
    """unet.hip:pack_conv: largest |w| 2^wexp in [2^13, 2^14)."""
    wmax = float(np.abs(w).max())
    if wmax == 0.0 or not np.isfinite(wmax):
        return 0
    _, e = np.frexp(np.float32(wmax))
    return int(min(max(14 - int(e), -10), 40))


def split(v):
    h = v.astype(np.float16)
    l = (v - h.astype(np.float32)).astype(np.float16)
    return h, l


def conv_dot(a, w):
    """(rows, K) x (K,) as the kernel evaluates it: fp16 pair operands, fp32 accumulation per 32-wide k-slice."""
    amax = np.abs(a).max().astype(np.float32)
    s, sf = act_scale(amax.view(np.uint32))
    wexp = weight_exp(w)
    oscale = np.uint32((254 - sf - wexp) << 23).view(np.float32)
    ah, al = split(a * s)
    wh, wl = split(np.ldexp(w, wexp).astype(np.float32))
    acc = np.zeros(a.shape[0], np.float32)
    f = lambda x: x.astype(np.float64)
    for k in range(0, a.shape[1], 32):
        sl = slice(k, k + 32)
        for x, y in ((al, wh), (ah, wl), (ah, wh)):
            acc = (acc.astype(np.float64) + f(x[:, sl]) @ f(y[sl])).astype(np.float32)
    return acc * oscale, float(s), wexp


def test_scales_are_exact_powers_of_two_and_keep_the_operands_in_range():
    rng = np.random.default_rng(0)
    for mag in (1e-30, 1e-12, 1e-6, 1e-3, 1.0, 37.0, 1e4, 1e9, 1e12):
        a = (rng.normal(0, 1, (4, 64)) * mag).astype(np.float32)
        s, sf = act_scale(np.abs(a).max().astype(np.float32).view(np.uint32))
        m, e = np.frexp(s)
        assert m == 0.5                                               # a power of two
        scaled = np.abs(a).max() * s
        if 2.0 ** -40 <= np.abs(a).max() <= 2.0 ** 40:
            assert 2.0 ** 14 <= scaled < 2.0 ** 15                    # below fp16's largest finite value (65504)
        assert np.isfinite((a * s).astype(np.float16)).all() or mag > 2.0 ** 40
    for wm in (1e-4, 0.03, 0.9, 5.0, 300.0):
        w = (rng.normal(0, 1, 1000) * wm).astype(np.float32)
        we = weight_exp(w)
        assert 2.0 ** 13 <= np.abs(w).max() * 2.0 ** we < 2.0 ** 14
    assert act_scale(0)[1] == 268 - 87 and weight_exp(np.zeros(3, np.float32)) == 0      # all-zero tensors


def test_three_fp16_terms_reproduce_an_fp64_dot_product_at_fp32_accuracy():
    rng = np.random.default_rng(1)
    K, N = 1280, 512
    for amag, wmag in ((1.0, 0.05), (1e-4, 1e-3), (30.0, 2.0), (1e3, 1e-2)):
        a = (rng.normal(0, 1, (N, K)) * amag).astype(np.float32)
        a[:, ::7] *= 1e-3                                             # a wide range inside every row
        w = (rng.normal(0, 1, K) * wmag).astype(np.float32)
        ref = a.astype(np.float64) @ w.astype(np.float64)
        got, s, wexp = conv_dot(a, w)
        rms = np.sqrt(np.mean((got - ref) ** 2) / np.mean(ref ** 2))
        # plain fp32 accumulation of the same products, for scale
        acc = np.zeros(N, np.float32)
        for k in range(0, K, 4):
            acc = (acc.astype(np.float64) + a[:, k:k + 4].astype(np.float64) @ w[k:k + 4].astype(np.float64)).astype(np.float32)
        rms32 = np.sqrt(np.mean((acc - ref) ** 2) / np.mean(ref ** 2))
        assert rms < 6e-7, (amag, wmag, rms)
        assert rms < 1.5 * rms32 + 1e-8, (rms, rms32)                 # at least as good as fp32 accumulation


def test_low_terms_degrade_gracefully_below_the_normal_range():
    # elements 2^-20 of the group's maximum: their low term is subnormal, the ABSOLUTE error stays at 2^-25 of the
    # scaled unit, i.e. far below the fp32 rounding of the large elements they are summed with
    a = np.array([[1.0, 2.0 ** -20 * 1.2345678, 2.0 ** -22 * 1.7654321, 0.0] * 8], np.float32)
    w = np.full(32, 0.75, np.float32)
    got, s, _ = conv_dot(a, w)
    ref = a.astype(np.float64) @ w.astype(np.float64)
    assert abs(got[0] - ref[0]) <= 2.0 ** -24 * abs(ref[0])
    h, l = split(a * np.float32(s))
    small = (a[0] > 0) & (a[0] < 1e-3)
    err = np.abs(h.astype(np.float64) + l.astype(np.float64) - a.astype(np.float64) * s)[0]
    assert (err[small] <= 2.0 ** -25).all()
