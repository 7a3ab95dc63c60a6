"""GPU parity: Domain / NTT / UnivariateEval::multiply vs the CPU oracle's serial_fft restatement (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def dev(t):
    return t.cpu().numpy().view(np.uint64)


def test_domain_new(zk):   # domain.rs:154-168
    d = zk.Domain(10)
    assert d.size == 16
    assert zk.Fr.to_ints(d.generator) == [14788168760825820622209131888203028446852016562542525606630160374691593895118]
    assert zk.Fr.to_ints(d.group_gen_inverse) == [26753076894533791554649012143113393549300550745003194222677083919072199473480]
    assert zk.Fr.to_ints(d.group_size_inverse) == [pow(16, -1, zk.Fr.MODULUS)]


@pytest.mark.parametrize("log_n", [0, 1, 2, 5, 9, 10, 11, 12, 14, 16, 18, 21])   # 21 = the transform size of a 2^20 x 2^20 product
def test_fft_ifft_match_oracle(zk, ora, log_n):
    n = 1 << log_n
    x = ora.random_fr(n, 70 + log_n)
    d = zk.Domain(n)
    ev = d.fft(x)
    assert np.array_equal(dev(ev), ora.domain_fft(x, n))
    assert np.array_equal(dev(d.ifft(ev)), x)
    assert np.array_equal(dev(d.ifft(x)), ora.domain_ifft(x, n))


def test_fft_pads_short_input(zk, ora):
    x = ora.random_fr(37, 5)
    d = zk.Domain(37)
    assert d.size == 64
    assert np.array_equal(dev(d.fft(x)), ora.domain_fft(x, 64))
    assert np.array_equal(dev(zk.UnivariateEval.from_coefficients(x).values), ora.domain_fft(x, 64))


def test_multiply_kats(zk):   # dense_univariate.rs:464-497 values through the NTT product
    mul = lambda a, b: zk.Fr.to_ints(dev(zk.UnivariateEval.multiply(zk.DenseUnivariatePolynomial(zk.Fr.from_ints(a)),
                                                                    zk.DenseUnivariatePolynomial(zk.Fr.from_ints(b))).coefficients))
    assert mul([1, 3, 2], [3, 2]) == [3, 11, 12, 4]
    assert mul([6, 5, 3], [5, 4, 2]) == [30, 49, 47, 22, 6]
    assert mul([1, 3, 2], [3]) == [3, 9, 6]
    assert mul([7], [6]) == [42]


@pytest.mark.parametrize("na,nb", [(1, 1), (37, 50), (1000, 1049), (5000, 3000), (1 << 14, 1 << 14)])
def test_multiply_matches_oracle(zk, ora, na, nb):
    a, b = ora.random_fr(na, 11), ora.random_fr(nb, 12)
    got = dev(zk.UnivariateEval.multiply(zk.DenseUnivariatePolynomial(a), zk.DenseUnivariatePolynomial(b)).coefficients)
    assert np.array_equal(got, ora.univariate_multiply(a, b))
    if na * nb <= 40 * 60:
        assert np.array_equal(got, ora.dense_mul(a, b))


def test_multiply_2_20_evaluation_identity(zk, ora):
    """SURVEY 2a size (2 x 2^20 coefficients -> 2^21-point transforms): (a*b)(z) == a(z) * b(z) at a random z,
    with the three evaluations done by the oracle's Horner-free restatement on the downloaded coefficients."""
    import torch
    n = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(3)
    a = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    b = torch.randint(0, 2 ** 62, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    c = zk.UnivariateEval.multiply(zk.DenseUnivariatePolynomial(a), zk.DenseUnivariatePolynomial(b)).coefficients
    assert c.shape[0] == 2 * n - 1
    R = zk.Fr.MODULUS
    z = 0x1234567890ABCDEF1234567890ABCDEF % R

    def horner(t):   # python ints over a strided sample would not be an identity; evaluate fully but vectorised by chunks
        ints = zk.Fr.to_ints(t.cpu().numpy().view(np.uint64))
        acc = 0
        for v in reversed(ints):
            acc = (acc * z + v) % R
        return acc
    # keep the CPU side affordable: check the identity on the low 2^12 x 2^12 sub-product instead of 2^20
    m = 1 << 12
    c_small = zk.UnivariateEval.multiply(zk.DenseUnivariatePolynomial(a[:m].clone()), zk.DenseUnivariatePolynomial(b[:m].clone())).coefficients
    assert horner(c_small) == horner(a[:m]) * horner(b[:m]) % R
    # and tie the big product to the small one: the lowest m coefficients of a*b depend only on a[:m], b[:m]
    assert torch.equal(c[:m], c_small[:m])
    # iNTT(NTT(x)) == x at the full 2^21 size
    d = zk.Domain(2 * n)
    x = torch.cat([a, b])
    assert torch.equal(d.ifft(d.fft(x)), x)


# ---- DenseUnivariatePolynomial::{evaluate, degree, Mul} on device coefficients (dense_univariate.rs) --------------
def test_dense_polynomial_evaluation_degree_multiplication(ora):
    import zk_cryptography_amd as zk
    F = zk.Fr.from_ints
    D = zk.DenseUnivariatePolynomial
    ints = lambda p: zk.Fr.to_ints(p.coefficients.cpu().numpy().view(np.uint64)) if len(p) else []   # noqa: E731
    assert zk.Fr.to_ints(D(F([5, 2, 4])).evaluate(zk.Fr.from_int(2))) == [25]                  # :425-433
    assert zk.Fr.to_ints(D(F([5, 2, 0, 0, 0, 0, 4])).evaluate(zk.Fr.from_int(2))) == [265]     # :185-188
    assert D(F([1, 3, 2])).degree() == 2 and D(F([1, 3, 0, 0])).degree() == 1 and D(F([0, 0])).degree() == 0
    assert ints(D(F([1, 3, 2])) * D(F([3, 2]))) == [3, 11, 12, 4]                               # :464-477
    assert ints(D(F([6, 5, 3])) * D(F([5, 4, 2]))) == [30, 49, 47, 22, 6]                       # :479-497
    assert ints(D(F([1, 3, 2])) * D(F([3]))) == [3, 9, 6]                                        # :500-509
    assert ints(D(F([1, 3, 2, 0, 0])) * D(F([3, 2, 0]))) == [3, 11, 12, 4]                       # degree() ignores zero leading coefficients
    assert ints(D(F([1, 3, 2])) * zk.Fr.from_int(3)) == [3, 9, 6] and ints(D(F([1, 3, 2])) * zk.Fr.from_int(0)) == []


@pytest.mark.parametrize("n", [1, 7, 2048, 2049, 100000])
def test_dense_evaluate_matches_oracle(ora, n):
    import zk_cryptography_amd as zk
    coeffs, z = ora.random_fr(n, 900 + n), ora.random_fr(1, 901 + n)[0]
    got = zk.DenseUnivariatePolynomial(coeffs).evaluate(z)
    ci, zi, acc = zk.Fr.to_ints(coeffs), zk.Fr.to_ints(z)[0], 0
    for c in reversed(ci):
        acc = (acc * zi + c) % zk.Fr.MODULUS
    assert zk.Fr.to_ints(got) == [acc]
    if n <= 2049:
        assert np.array_equal(got, ora.dense_evaluate(coeffs, z))
