"""The arithmetic the split-operand kernels rest on, checked in numpy on the host: an fp32
value scaled by a power of two IS the sum of two f16 numbers up to 2^-22 of itself; the four (or, with K ordered as in
rw_dconv.hip, two times two) piece products reproduce the fp32 product to 2^-21; the exponent formulas of the kernels
keep every piece inside f16's range; what falls below f16's normal range costs an ABSOLUTE error that is negligible
beside the map's maximum.  No kernel is involved: this pins the numbers quoted in the headers."""
import numpy


def split(v):
    """(h, l) as the kernels compute them: h = f16(v), l = f16(v - h) (v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32)"""
    v = v.astype(numpy.float32)
    h = v.astype(numpy.float16)
    r = v - h.astype(numpy.float32)                 # exact in fp32: h has 11 of v's 24 significant bits
    return h, r.astype(numpy.float16)


def exponent_above(x):
    """e with x < 2^e, read off the exponent field as the kernels do ((bits >> 23) & 0xff) - 126"""
    bits = numpy.float32(x).view(numpy.uint32)
    return int((bits >> 23) & 0xff) - 126


def test_a_scaled_fp32_value_is_a_pair_of_f16_numbers_to_2_pow_minus_22():
    rs = numpy.random.RandomState(0)
    v = (rs.standard_t(3, size=200000) * numpy.exp(rs.randn(200000))).astype(numpy.float32)
    am = numpy.abs(v).max()
    for top in (14, 8, 12):                          # direct sums / F(4x4,3x3) (gain 100 < 2^7) / F(2,2) (gain 4)
        e = exponent_above(am)
        assert am < 2.0 ** e <= 2 * am * (1 + 1e-7) or am == 2.0 ** (e - 1)
        vs = v * numpy.float32(2.0 ** (top - e))
        h, l = split(vs)
        assert numpy.isfinite(h.astype(numpy.float32)).all() and numpy.abs(h.astype(numpy.float32)).max() <= 2.0 ** top
        back = h.astype(numpy.float64) + l.astype(numpy.float64)
        big = numpy.abs(vs) >= 2.0 ** -3             # h and l both normal f16 numbers (or l exactly 0)
        rel = numpy.abs(back[big] - vs[big].astype(numpy.float64)) / numpy.abs(vs[big])
        assert rel.max() <= 2.0 ** -22
        # below that l (and then h) become f16 denormals: an absolute error of at most 2^-25 in the scaled units --
        # 2^-39 of the map's maximum for the direct sums, where an fp32 accumulation beside that maximum keeps 2^-24
        err = numpy.abs(back - vs.astype(numpy.float64))
        assert err[~big].max() <= 2.0 ** -25
        assert err.max() <= 2.0 ** -22 * 2.0 ** top


def test_piece_products_reproduce_the_fp32_product_to_2_pow_minus_21():
    rs = numpy.random.RandomState(1)
    u = rs.randn(100000).astype(numpy.float32)
    v = (rs.randn(100000) * numpy.exp(rs.randn(100000))).astype(numpy.float32)
    eu, ev = exponent_above(numpy.abs(u).max()), exponent_above(numpy.abs(v).max())
    us, vs = u * numpy.float32(2.0 ** (15 - eu)), v * numpy.float32(2.0 ** (14 - ev))
    uh, ul = (x.astype(numpy.float64) for x in split(us))
    vh, vl = (x.astype(numpy.float64) for x in split(vs))
    assert numpy.abs(uh).max() <= 2.0 ** 15 and numpy.abs(vh).max() <= 2.0 ** 14          # inside f16 (65504)
    # rw_wino4.hip / rw_upwino.hip: A = [Uh, Ul, Uh, Ul], B = [Vh, Vh, Vl, Vl]; rw_dconv.hip: [Vh, Vl] . [Uh, Uh] then
    # [Vh, Vl] . [Ul, Ul] -- the same four products, every one of them exact in the fp32 accumulator's input (22-bit products)
    four = uh * vh + ul * vh + uh * vl + ul * vl
    exact = us.astype(numpy.float64) * vs.astype(numpy.float64)
    big = (numpy.abs(us) >= 2.0 ** -3) & (numpy.abs(vs) >= 2.0 ** -3)
    rel = numpy.abs(four[big] - exact[big]) / numpy.abs(exact[big])
    assert rel.max() <= 2.0 ** -21
    # the scales leave exactly: powers of two
    back = four * 2.0 ** (eu - 15) * 2.0 ** (ev - 14)
    assert numpy.abs(back[big] - u[big].astype(numpy.float64) * v[big].astype(numpy.float64)).max() <= \
        2.0 ** -21 * numpy.abs(u[big].astype(numpy.float64) * v[big].astype(numpy.float64)).max()
    # a direct sum over 576 terms of such products stays far inside fp32's range: |Uh Vh| <= 2^29
    assert 576 * 2.0 ** 29 < 3.4e38


def test_a_bound_that_is_too_large_costs_low_bits_only_and_one_that_is_too_small_overflows():
    rs = numpy.random.RandomState(2)
    v = rs.randn(50000).astype(numpy.float32)
    am = numpy.abs(v).max()
    e = exponent_above(am)
    for k, bits in ((0, 22), (5, 17), (12, 10)):     # a bound 2^k too large: k bits fewer in the pair
        vs = v * numpy.float32(2.0 ** (14 - e - k))
        h, l = split(vs)
        err = numpy.abs(h.astype(numpy.float64) + l.astype(numpy.float64) - vs.astype(numpy.float64)).max() / (am * 2.0 ** (14 - e - k))
        assert err <= 2.0 ** -bits * 1.01
    # ... which is how a STALE bound of another map shows up: finite images, some tiles off by 1e-3 relative (the issue of
    # DESIGN.md section 9, item 0); a bound 8x too small overflows f16 instead
    with numpy.errstate(over='ignore'):
        h, _ = split(v * numpy.float32(2.0 ** (14 - e + 3)))
    assert numpy.isinf(h.astype(numpy.float32)).any()


def test_three_piece_products_are_as_good_as_four():
    """Round 5 (csrc/rw_dconv.hip, DC_PRODUCTS == 3): Ul Vl is dropped for the taps ky = 0 / 1 of every kernel column --
    [Vh(row r) | Vh(row r + 1)] . [Ul(ky 0) | Ul(ky 1)] gives both taps' Vh Ul in ONE instruction where two [Vh | Vl] . [Ul | Ul]
    were issued.  |Ul| <= 2^-11 |Uh| and |Vl| <= 2^-11 |Vh|: the dropped term is <= 2^-22 of the product, beside pieces
    that are themselves rounded at 2^-22 -- the three-product sum holds the SAME 2^-21 bar, and a 576-term dot product of
    such sums differs from the four-product one by less than fp32 accumulation moves it."""
    rs = numpy.random.RandomState(4)
    u = rs.randn(100000).astype(numpy.float32)
    v = (rs.randn(100000) * numpy.exp(rs.randn(100000))).astype(numpy.float32)
    eu, ev = exponent_above(numpy.abs(u).max()), exponent_above(numpy.abs(v).max())
    us, vs = u * numpy.float32(2.0 ** (15 - eu)), v * numpy.float32(2.0 ** (14 - ev))
    uh, ul = (x.astype(numpy.float64) for x in split(us))
    vh, vl = (x.astype(numpy.float64) for x in split(vs))
    exact = us.astype(numpy.float64) * vs.astype(numpy.float64)
    three = uh * vh + uh * vl + ul * vh
    four = three + ul * vl
    big = (numpy.abs(us) >= 2.0 ** -3) & (numpy.abs(vs) >= 2.0 ** -3)
    assert (numpy.abs(ul * vl)[big] / numpy.abs(exact[big])).max() <= 2.0 ** -22
    assert (numpy.abs(three[big] - exact[big]) / numpy.abs(exact[big])).max() <= 2.0 ** -21
    # a dot product of the layer's length: the difference between the two forms against the sum's own fp32 rounding
    n = 576
    t3 = three[:n * 170].reshape(170, n).sum(1)
    t4 = four[:n * 170].reshape(170, n).sum(1)
    scale = numpy.abs(exact[:n * 170].reshape(170, n)).sum(1)
    assert (numpy.abs(t3 - t4) / scale).max() <= 2.0 ** -24
