"""CPU: the arithmetic claim behind the split engines, checked with a NumPy emulation of `split2_bf16x3`
(csrc/wres.hip.h; round-to-nearest at each level, ties away from zero): the three planes are bf16 values,
their sum is the fp32 value EXACTLY (every mantissa checked), every kept plane product is exact in fp32, and the six-term
product differs from the exact product by at most 2^-24 relative -- the rounding of one IEEE fp32 multiply -- and by 3.5e-9
on average (an fp32 multiply: 2.1e-8).  (-DMRL_PRODUCTS8 builds keep two more terms: < 2^-33.)"""
import numpy as np


def round_bf16(x):
    """half a unit of the bf16 last place added to the magnitude bits, then truncated (split2_bf16x3)"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x8000) & 0xffff0000).astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    p0 = round_bf16(x)
    r1 = x - p0
    p1 = round_bf16(r1)
    r2 = r1 - p1
    return p0, p1, round_bf16(r2), r2


def _is_bf16(p):
    return np.all((np.asarray(p, np.float32).view(np.uint32) & np.uint32(0xffff)) == 0)


def _values(rng, n):
    mags = np.exp(rng.uniform(np.log(1e-6), np.log(1e4), n))
    x = (mags * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    x[:8] = [0.0, 1.0, -1.0, 255.0, 1.0 / 255.0, 3.0e38, -1.17549435e-38 * 4, 0.1]
    return x


def test_three_way_split_is_exact_and_bf16():
    rng = np.random.RandomState(0)
    x = _values(rng, 200000)
    p0, p1, p2, r2 = split3(x)
    assert _is_bf16(p0) and _is_bf16(p1) and _is_bf16(p2)
    np.testing.assert_array_equal(p2, r2)                        # the second residual already is a bf16 value
    np.testing.assert_array_equal((p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64)), x.astype(np.float64))
    np.testing.assert_array_equal((p0 + p1) + p2, x)
    assert np.all(np.abs(p1) <= np.abs(x) * 2.0 ** -8) and np.all(np.abs(p2) <= np.abs(x) * 2.0 ** -17)


def test_three_way_split_is_exact_for_every_mantissa():
    """all 2^23 mantissas of one binade (the split is scale-invariant away from the ends of the exponent range)"""
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3f800000)).view(np.float32)
    for sign in (1.0, -1.0):
        p0, p1, p2, r2 = split3(np.float32(sign) * x)
        assert _is_bf16(p0) and _is_bf16(p1) and _is_bf16(p2)
        np.testing.assert_array_equal(p2, r2)
        np.testing.assert_array_equal(p0.astype(np.float64) + p1.astype(np.float64) + p2.astype(np.float64), sign * x.astype(np.float64))
        assert (np.abs(p1) / x).max() <= 2.0 ** -8 and (np.abs(p2) / x).max() <= 2.0 ** -17


def test_u8_times_three_planes_is_exact():
    """bf16 x 3 (first conv layer): integer pixels 0..255 are bf16 values; pixel * plane is exact in fp32"""
    rng = np.random.RandomState(1)
    w = _values(rng, 50000) / np.float32(255.0)
    pix = rng.randint(0, 256, w.size).astype(np.float32)
    for p in split3(w)[:3]:
        prod32 = pix * p
        np.testing.assert_array_equal(prod32.astype(np.float64), pix.astype(np.float64) * p.astype(np.float64))


def test_six_term_product_is_within_one_fp32_rounding():
    rng = np.random.RandomState(2)
    a, b = _values(rng, 1000000)[8:], _values(rng, 1000000)[::-1][8:]
    keep = (np.abs(a.astype(np.float64) * b) < 1e30) & (np.abs(a.astype(np.float64) * b) > 1e-30)
    a, b = a[keep], b[keep]
    a0, a1, a2 = (p.astype(np.float64) for p in split3(a)[:3])
    b0, b1, b2 = (p.astype(np.float64) for p in split3(b)[:3])
    for x, y in ((a0, b0), (a0, b1), (a1, b0), (a0, b2), (a1, b1), (a2, b0)):           # 8 x 8 significant bits: exact in fp32
        np.testing.assert_array_equal((x * y).astype(np.float32).astype(np.float64), x * y)
    six = a0 * b0 + (a0 * b1 + a1 * b0) + (a0 * b2 + a1 * b1 + a2 * b0)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    # dropped: a1 b2 + a2 b1 + a2 b2 <= (2^-25 + 2^-25 + 2^-34) |a b|
    assert rel.max() <= 2.0 ** -24, rel.max()
    fp32_mul = np.abs(exact.astype(np.float32).astype(np.float64) - exact) / np.abs(exact)           # one IEEE fp32 multiply
    assert fp32_mul.max() <= 2.0 ** -24 and rel.max() <= fp32_mul.max()
    assert rel.mean() < 4e-9 and rel.mean() < fp32_mul.mean() / 5                          # 3.5e-9 vs 2.1e-8
    assert abs(((six - exact) / np.abs(exact)).mean()) < 1e-10                             # and no bias
    eight = six + (a1 * b2 + a2 * b1)                                                      # -DMRL_PRODUCTS8
    assert (np.abs(eight - exact) / np.abs(exact)).max() < 2.0 ** -33


def test_split_dot_products_are_fp32_class():
    """a 512-long dot product (conv2's K) through the six-term products with fp32 accumulation vs plain fp32 accumulation:
    both sit within the same distance of the fp64 result"""
    rng = np.random.RandomState(3)
    A = rng.randn(256, 512).astype(np.float32)
    B = (rng.randn(512, 64) * 0.05).astype(np.float32)
    exact = A.astype(np.float64) @ B.astype(np.float64)
    plain = np.zeros((256, 64), np.float32)
    split = np.zeros((256, 64), np.float32)
    a = split3(A)
    b = split3(B)
    for k0 in range(0, 512, 16):                                 # one MFMA k-block at a time, fp32 accumulator
        sl = slice(k0, k0 + 16)
        plain = plain + (A[:, sl].astype(np.float64) @ B[sl].astype(np.float64)).astype(np.float32)
        for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):          # the kernels' order: small terms first
            split = split + (a[i][:, sl].astype(np.float64) @ b[j][sl].astype(np.float64)).astype(np.float32)
    scale = np.abs(exact).max()
    e_plain, e_split = np.abs(plain - exact).max() / scale, np.abs(split - exact).max() / scale
    assert e_split < 1e-6 and e_split < 4 * e_plain + 1e-7, (e_plain, e_split)   # six accumulator roundings per k block instead of one


def test_transposed_accumulator_layout_and_plane_order():
    """Index algebra of baselines_amd/csrc/planes.hip.h, restated in NumPy.
    With the MFMA operands swapped, lane (i, h) of a 32 x 32 accumulator owns row i and the columns 8g + 4h + j, held in
    accumulator register 4g + j (C/D layout of v_mfma_f32_32x32x16: register r of lane l belongs to the register-indexed
    dimension at (r & 3) + 8 (r >> 2) + 4 (l >> 5)).  The 64 lanes x 16 registers cover the tile exactly once, every
    float4 store is 16-byte aligned inside its row, the mask-word bits of the two half-waves are disjoint and complete,
    and perm32 -- the order of a plane tensor inside each block of 32 -- gives every lane 16 consecutive positions."""
    def perm32(c):
        return ((c >> 2) & 1) * 16 + (c >> 3) * 4 + (c & 3)
    assert sorted(perm32(c) for c in range(32)) == list(range(32))                       # a permutation
    seen = np.zeros((32, 32), int)
    for lane in range(64):
        i, h = lane & 31, lane >> 5
        bits = 0
        for r in range(16):
            col = (r & 3) + 8 * (r >> 2) + 4 * h                                         # hardware layout of register r
            g, j = r >> 2, r & 3
            assert col == 8 * g + 4 * h + j                                              # what tr_block_epilogue assumes
            seen[i, col] += 1
            bits |= 1 << col
        assert bits == (0x0f0f0f0f << (4 * h)) & 0xffffffff                              # half-wave h: nibbles 2g + h
        for g in range(4):
            assert (8 * g + 4 * h) % 4 == 0                                              # float4 stores: 16-byte aligned
            assert [perm32(8 * g + 4 * h + j) for j in range(4)] == [16 * h + 4 * g + j for j in range(4)]
        assert sorted(perm32(8 * g + 4 * h + j) for g in range(4) for j in range(4)) == list(range(16 * h, 16 * h + 16))
    assert (seen == 1).all()
    # a dot product does not care in which order k runs: the consumers' weight planes use the same perm32 order
    rng = np.random.RandomState(0)
    a, b = rng.randn(64).astype(np.float64), rng.randn(64).astype(np.float64)
    kp = np.array([(k & ~31) | perm32(k & 31) for k in range(64)])
    ap, bp = np.empty(64), np.empty(64)
    ap[kp], bp[kp] = a, b
    assert abs(ap @ bp - a @ b) < 1e-12


def split3_sg(x, negate):
    """NumPy emulation of `split2_bf16x3_sg` (csrc/wres.hip.h): the planes of s x, s = -1 when `negate`, from the bits of x -- the
    rounding constant carries the sign flip (adding 2^31 flips bit 31), the first residual is one fma s x - plane0"""
    x = np.asarray(x, np.float32)
    k = 0x80008000 if negate else 0x8000
    s = np.float32(-1.0 if negate else 1.0)
    t = (x.view(np.uint32).astype(np.uint64) + k) & 0xffffffff
    p0 = (t & 0xffff0000).astype(np.uint32).view(np.float32)
    r1 = (s.astype(np.float64) * x.astype(np.float64) - p0.astype(np.float64)).astype(np.float32)     # fma: one rounding (exact here)
    p1 = round_bf16(r1)
    r2 = r1 - p1
    return p0, p1, round_bf16(r2), r2


def test_signed_split_yields_exactly_the_planes_of_the_negated_value():
    """the sign alternation of the staged rows (x6_dither) costs no instruction because the split of -x falls out of
    the bits of x: every plane equals the plane of split3(-x), for every mantissa and both signs; s = +1 is the plain split"""
    x = (np.arange(1 << 23, dtype=np.uint32) | np.uint32(0x3f800000)).view(np.float32)
    for sign in (1.0, -1.0):
        v = np.float32(sign) * x
        ref = split3(-v)
        got = split3_sg(v, True)
        for a, b in zip(got, ref):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
        for a, b in zip(split3_sg(v, False), split3(v)):
            np.testing.assert_array_equal(a.view(np.uint32), b.view(np.uint32))
    rng = np.random.RandomState(3)
    v = _values(rng, 200000)
    for a, b in zip(split3_sg(v, True)[:3], split3(-v)[:3]):
        np.testing.assert_array_equal(a, b)
