"""Exhaustive checks of the two exact shortcuts the composite kernel takes (kernels_composite.hip):

  * destination alpha 255:  Pillow's AlphaComposite.c formula == div255(s*sa + d*(255-sa) + 128),
    alpha 255, for every (sa, s, d) in 0..255 (16.7 M triples) -- over_opaque_dst();
  * source alpha 0 / 255:   the formula keeps dst / takes src exactly, for every destination alpha
    -- the select path for binary-alpha cutouts;
  * the packed form (R and B in the two 16-bit halves of one register) never carries.

Pure numpy; pins the identities the HIP code relies on to Pillow's published integer formula
(restated in oracle/mic_oracle.c and itself pinned against the reference's fixtures)."""
import numpy as np


def _d255(t):
    return ((t >> 8) + t) >> 8


def _pillow(s, sa, d, da):
    outa255 = sa * 255 + da * (255 - sa)
    coef1 = np.where(sa > 0, sa * 255 * 255 * 128 // np.maximum(outa255, 1), 0)
    coef2 = 255 * 128 - coef1
    c = _d255(s * coef1 + d * coef2 + (0x80 << 7)) >> 7
    a = _d255(outa255 + 0x80)
    return np.where(sa == 0, d, c), np.where(sa == 0, da, a)


def test_opaque_destination_identity_exhaustive():
    sa = np.arange(256, dtype=np.int64)[:, None, None]
    s = np.arange(256, dtype=np.int64)[None, :, None]
    d = np.arange(256, dtype=np.int64)[None, None, :]
    c, a = _pillow(s, sa, d, 255)
    T = s * sa + d * (255 - sa) + 128
    assert np.array_equal(c, _d255(T))
    assert (a == 255).all()
    assert int((T + (T >> 8)).max()) < 65536  # two channels per 32-bit register never carry


def test_binary_source_alpha_is_a_select_for_any_destination():
    s = np.arange(256, dtype=np.int64)[:, None, None]
    d = np.arange(256, dtype=np.int64)[None, :, None]
    da = np.arange(256, dtype=np.int64)[None, None, :]
    c, a = _pillow(s, np.int64(255), d, da)
    assert (c == s).all() and (a == 255).all()
    c0, a0 = _pillow(s, np.int64(0), d, da)
    assert (c0 == d).all() and (a0 == da).all()


def test_packed_lanes_match_scalar():
    rng = np.random.default_rng(0)
    s = rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64)
    d = rng.integers(0, 2 ** 32, 200_000, dtype=np.uint64) | 0xFF000000
    sa, na = s >> 24, 255 - (s >> 24)
    M = 0x00FF00FF
    rb = ((s & M) * sa + (d & M) * na + 0x00800080) & 0xFFFFFFFF
    g = ((s >> 8) & 0xFF) * sa + ((d >> 8) & 0xFF) * na + 0x80
    rb = ((((rb >> 8) & M) + rb) >> 8) & M
    g = ((g >> 8) + g) >> 8
    packed = rb | (g << 8) | 0xFF000000
    want = np.zeros_like(packed)
    for sh in (0, 8, 16):
        c, _ = _pillow((s >> sh) & 255, sa, (d >> sh) & 255, 255)
        want |= c.astype(np.uint64) << sh
    want |= 0xFF000000
    assert np.array_equal(packed, want)


def test_unpremultiply_reciprocal_table_exhaustive():
    """kernels_resample.hip: 255*c // a == mulhi(c << 8, ceil(255 * 2^24 / a)) for 0 < a < 255."""
    c = np.arange(256, dtype=np.uint64)
    for a in range(1, 255):
        R = -(-(255 << 24) // a)
        assert R < 2 ** 32
        q = ((c << np.uint64(8)) * np.uint64(R)) >> np.uint64(32)
        assert np.array_equal(q, (255 * c) // a), a


def test_unpremultiply_float_factor_exhaustive():
    """kernels_resample.hip (MFMA kernel): min(255, 255*c // a) == min(255, floor(float32(c) * F[a]))
    with F[a] = float32(255 / a) bumped up one ulp, for 0 < a < 255 and every byte c."""
    c = np.arange(256)
    for a in range(1, 255):
        F = np.nextafter(np.float32(255.0) / np.float32(a), np.float32(np.inf))
        assert F.dtype == np.float32
        q = np.floor(c.astype(np.float32) * F).astype(np.int64)
        assert np.array_equal(np.minimum(255, q), np.minimum(255, (255 * c) // a)), a


def test_premultiply_packed_lanes_and_digit_split():
    """kernels_resample.hip premultiply2_hi: two channels in 16-bit lanes of one register give
    Pillow's MULDIV255 per lane (result in the lane's high byte); {G, 255} * a keeps alpha; and
    resample_coeffs.cpp's signed-byte digits rebuild every 22-bit tap."""
    a = np.arange(256, dtype=np.uint64)[:, None]
    c = np.arange(256, dtype=np.uint64)[None, :]
    x = c | (np.uint64(255) << np.uint64(16))               # lanes {c, 255}
    t = (x * a + np.uint64(0x00800080)) & np.uint64(0xFFFFFFFF)
    u = (t + ((t >> np.uint64(8)) & np.uint64(0x00FF00FF))) & np.uint64(0xFFFFFFFF)
    lo, hi = (u >> np.uint64(8)) & np.uint64(255), (u >> np.uint64(24)) & np.uint64(255)
    tt = c * a + np.uint64(128)
    assert np.array_equal(lo, ((tt >> np.uint64(8)) + tt) >> np.uint64(8))  # Convert.c MULDIV255
    assert np.array_equal(hi, np.broadcast_to(a, hi.shape))                   # alpha survives
    k = np.arange(-(5 << 20), (5 << 20) + 1, 97, dtype=np.int64)  # taps are normalised: |k| <= ~2^22
    d0 = ((k + 128) & 255) - 128
    k1 = (k - d0) >> 8
    d1 = ((k1 + 128) & 255) - 128
    d2 = (k1 - d1) >> 8
    assert np.array_equal(d0 + 256 * d1 + 65536 * d2, k)
    assert d2.min() >= -128 and d2.max() <= 127


def test_unpremultiply_fma_round_to_nearest_exhaustive():
    """kernels_resample.hip unpremultiply4 (marching kernel): per channel one fma and one
    v_cvt_pk_u8_f32 (round to nearest even, saturating -- measured, profiles/r02_ubench_isa.txt):
        sat8(rne(float32(c * F[a] - 0.5 + 2^-9))) == min(255, 255*c // a)   for 0 < a < 255,
    and == c for a in {0, 255} with F = 1 (Convert.c rgba2rgbA copies those pixels)."""
    c = np.arange(256, dtype=np.float64)
    K = np.float64(np.float32(-0.5) + np.float32(2.0 ** -9))
    assert K == -0.5 + 2.0 ** -9
    for a in range(256):
        if a in (0, 255):
            F = np.float32(1.0)
            want = np.arange(256)
        else:
            F = np.nextafter(np.float32(255.0) / np.float32(a), np.float32(np.inf))
            want = np.minimum(255, (255 * np.arange(256)) // a)
        # the fma rounds once: c * F is exact in float64 (8 x 24 bits), so is adding K (2^-9 granularity
        # against |product| < 2^16: 25 + 24 bits < 53); float32() is then the fma's single rounding
        t = (c * np.float64(F) + K).astype(np.float32)
        got = np.clip(np.rint(t.astype(np.float64)), 0, 255).astype(np.int64)  # np.rint rounds half to even
        assert np.array_equal(got, want), a


def test_digit_chain_equals_the_full_sum():
    """kernels_resample.hip tile4: the three signed-byte digits of the taps are chained through the
    accumulator with two arithmetic shifts, acc = ((a0 + bias) >> 8 + a1) >> 8 + a2, out = sat8(acc >> 6),
    instead of being recombined: equal to Pillow's clip8((bias + a0 + 256 a1 + 65536 a2) >> 22) for any
    integers (nested floor divisions), checked here on random digit sums of the magnitudes a 64-tap window
    of signed bytes can produce, bias = 2^21 + 128 * sum(taps)."""
    rng = np.random.default_rng(5)
    n = 2_000_000
    lim = 64 * 128 * 128
    a0 = rng.integers(-lim, lim + 1, n, dtype=np.int64)
    a1 = rng.integers(-lim, lim + 1, n, dtype=np.int64)
    a2 = rng.integers(-4 * 16384, 4 * 16384 + 1, n, dtype=np.int64)
    bias = (1 << 21) + 128 * rng.integers((1 << 22) - 64, (1 << 22) + 65, n, dtype=np.int64)
    chain = ((((a0 + bias) >> 8) + a1) >> 8) + a2
    assert np.abs(a0 + bias).max() < 2 ** 31  # int32 accumulators never overflow
    full = (bias + a0 + 256 * a1 + 65536 * a2) >> 22
    assert np.array_equal(np.clip(chain >> 6, 0, 255), np.clip(full, 0, 255))
    assert np.array_equal(chain >> 6, full)
