"""GPU parity: KZG commit (Pippenger MSM on gfx950) and SRS generation vs the CPU oracle's restatement of the
reference's naive sum of mul_bigint; equality on affine coordinates.  Names follow kzg/src/*_kzg.rs tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def _aff(ora, jac):
    a = ora.g1_to_affine(jac)
    return a[:12].copy(), bool(a[12])


def _same(zk, got, want_xy, want_inf):
    assert got.infinity == want_inf
    if not want_inf:
        assert np.array_equal(got.xy, want_xy)


def test_kzg_1_commitment(zk, ora):   # multilinear_kzg.rs:133-148 data; commit == p(tau) * G = 28 G
    vals = [0, 7, 0, 5, 0, 7, 4, 9]
    tau = zk.Fr.from_ints([2, 3, 4])
    srs = zk.TrustedSetup.setup(tau)
    com = zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints(vals)), srs)
    want = ora.kzg_commitment(zk.Fr.from_ints(vals), ora.kzg_multilinear_srs_g1(tau), True)
    _same(zk, com, *_aff(ora, want))
    x, y = com.coords()
    assert x == 0x16ad11e5d15f77c1143b1697344911b9c590110fdd8dd09df2e58bfd757269169deefe8be3544d4e049fb3776fb0bcfb
    assert y == 0x0f5c8be5f27fc19eee337785e43d18414a8ff04995230f04509800252164cf47887a4a1864f18288652196af6272e7f6


def test_commit_public_g1_multiples(zk, ora):
    """commit([k], SRS) = k * (G * tau^0) = k G: the MSM path against the PUBLISHED compressed encodings of G, 2G, 3G
    (tests/test_oracle_kats.py::PUBLIC_G1_MULTIPLES -- vectors from outside this repository), plain and table paths."""
    from test_oracle_kats import PUBLIC_G1_MULTIPLES, g1_compress
    srs = zk.UnivariateKZG.generate_srs(zk.Fr.from_int(7), 3)
    table = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf).precompute()
    for k, want in PUBLIC_G1_MULTIPLES.items():
        for s in (srs, table):
            com = zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(zk.Fr.from_ints([k])), s)
            assert g1_compress(*com.coords()) == want
    # and the device-generated SRS itself: tau = 2 -> G, 2G, 4G = 2(2G)
    pts = zk.UnivariateKZG.generate_srs(zk.Fr.from_int(2), 1)
    x, y = zk.G1Affine(pts.powers_of_tau_in_g1[1].cpu().numpy().view(np.uint64), False).coords()
    assert g1_compress(x, y) == PUBLIC_G1_MULTIPLES[2]


def test_kzg_2_commitment(zk, ora):   # multilinear_kzg.rs:151-197 data
    vals = [0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4]
    tau = zk.Fr.from_ints([12, 9, 28, 40])
    srs = zk.TrustedSetup.setup(tau)
    com = zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints(vals)), srs)
    want = ora.kzg_commitment(zk.Fr.from_ints(vals), ora.kzg_multilinear_srs_g1(tau), True)
    _same(zk, com, *_aff(ora, want))
    tampered = zk.TrustedSetup.setup(zk.Fr.from_ints([12, 19, 28, 40]))
    assert not (zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints(vals)), tampered) == com)


def test_univariate_kzg_commitment(zk, ora):   # univariate_kzg.rs:111-129 data
    tau = zk.Fr.from_int(10)
    srs = zk.UnivariateKZG.generate_srs(tau, 4)
    coeffs = zk.Fr.from_ints([1, 2, 3, 4, 5])
    com = zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(coeffs), srs)
    want = ora.kzg_commitment(coeffs, ora.kzg_univariate_srs_g1(tau, 4), False)
    _same(zk, com, *_aff(ora, want))
    p_tau = sum(k * 10 ** i for i, k in enumerate([1, 2, 3, 4, 5]))
    _same(zk, com, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), p_tau)))
    # shorter polynomial than the SRS is fine; a longer one indexes out of bounds (univariate_kzg.rs:53)
    zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(coeffs[:3]), srs)
    with pytest.raises(IndexError):
        zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(zk.Fr.from_ints([1, 2, 3, 4, 5, 6])), srs)
    with pytest.raises(AssertionError):    # multilinear_kzg.rs:36-41
        zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints([1, 2, 3, 4])), srs)


@pytest.mark.parametrize("n_vars", [1, 3, 6])
def test_srs_multilinear_matches_oracle(zk, ora, n_vars):
    tau = ora.random_fr(n_vars, 31 + n_vars)
    srs = zk.TrustedSetup.setup(tau)
    want = ora.g1_batch_to_affine(ora.kzg_multilinear_srs_g1(tau))
    assert np.array_equal(srs.powers_of_tau_in_g1.cpu().numpy().view(np.uint64), want[:, :12])
    assert np.array_equal(srs.inf.cpu().numpy(), want[:, 12].astype(np.uint8))


def test_srs_univariate_matches_oracle(zk, ora):
    tau = ora.random_fr(1, 77)[0]
    srs = zk.UnivariateKZG.generate_srs(tau, 40)
    want = ora.g1_batch_to_affine(ora.kzg_univariate_srs_g1(tau, 40))
    assert np.array_equal(srs.powers_of_tau_in_g1.cpu().numpy().view(np.uint64), want[:, :12])


@pytest.mark.parametrize("tau", [2, R - 1, 256])
def test_srs_window_table_single_digit_scalars(zk, ora, tau):
    """SRS points come from a table of the generator's multiples indexed by the scalar's bytes: powers of two (one non-zero
    digit, every window and every bit of a window), powers of 256 (digit 1 per window, reduced mod r beyond 2^255) and +-1
    (r - 1: every digit non-zero) against the oracle's double-and-add."""
    t = zk.Fr.from_ints([tau])[0]
    deg = 254 if tau == 2 else 40
    srs = zk.UnivariateKZG.generate_srs(t, deg)
    want = ora.g1_batch_to_affine(ora.kzg_univariate_srs_g1(t, deg))
    assert np.array_equal(srs.powers_of_tau_in_g1.cpu().numpy().view(np.uint64), want[:, :12])
    assert not srs.inf.any()


def test_srs_with_identity_points_and_zero_scalars(zk, ora):
    # kzg/benches/multilinear_kzg_benchmark.rs:17-22: tau = (0,1,2,...) => eq-scalars that are 0 => G*0 = identity
    tau = zk.Fr.from_ints([0, 1, 2, 3])
    srs = zk.TrustedSetup.setup(tau)
    assert int(srs.inf.sum().item()) > 0
    vals = zk.Fr.from_ints([0, 5, 0, 0, 7, 0, 1, R - 1, 2, 3, 0, 0, 9, 9, 9, 1])
    com = zk.MultilinearKZG.commitment(zk.Multilinear(vals), srs)
    want = ora.kzg_commitment(vals, ora.kzg_multilinear_srs_g1(tau), True)
    _same(zk, com, *_aff(ora, want))
    zero = zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints([0] * 16)), srs)
    assert zero.infinity


def test_commit_duplicate_and_inverse_points(zk, ora):
    # all SRS points equal (P + P doublings inside buckets) and scalars that cancel (P + (-P))
    g = ora.g1_batch_to_affine(np.stack([ora.g1_generator()] * 64))
    srs = zk.TrustedSetup(g[:, :12], g[:, 12].astype(np.uint8))
    sc = zk.Fr.from_ints([5] * 32 + [R - 5] * 32)
    assert zk.MultilinearKZG.commitment(zk.Multilinear(sc), srs).infinity
    sc2 = zk.Fr.from_ints([3] * 64)
    com = zk.MultilinearKZG.commitment(zk.Multilinear(sc2), srs)
    _same(zk, com, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), 192)))


@pytest.mark.parametrize("n", [64, 256, 1024])
def test_commit_duplicate_and_inverse_points_every_pipeline(zk, ora, n):
    """The same degenerate inputs through the bucket pipeline on the plain SRS, the bucket pipeline on the table (ZKHIP_MSM_SMALL has no
    say over zkhip_kzg_commit) and the short path: equal points meet in every tree level (the doubling branch of the four-lane
    addition, csrc/g1u.hpp g1u_add_quad) and cancel (its identity branch)."""
    from zk_cryptography_amd import kzg as K
    g = ora.g1_batch_to_affine(np.stack([ora.g1_generator()] * n))
    srs = zk.TrustedSetup(g[:, :12], g[:, 12].astype(np.uint8))
    table = srs.precompute().table
    for ints, want in (([5] * (n // 2) + [R - 5] * (n // 2), None), ([3] * n, 3 * n), ([1, 2] * (n // 2), 3 * (n // 2)),
                       ([7] + [0] * (n - 1), 7), ([R - 1] * n, (R - 1) * n % R)):
        sc = zk.DenseUnivariatePolynomial(zk.Fr.from_ints(ints)).coefficients
        plain = K._commit(srs.powers_of_tau_in_g1, srs.inf, n, sc, n, True, None)
        short = K._commit(srs.powers_of_tau_in_g1, srs.inf, n, sc, n, True, table)
        assert plain == short
        if want is None:
            assert plain.infinity
        else:
            _same(zk, plain, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), want)))


@pytest.mark.parametrize("log_n", [4, 8, 10, 12])
def test_commit_random_matches_naive_oracle(zk, ora, log_n):
    n = 1 << log_n
    tau = ora.random_fr(log_n, 500 + log_n)
    srs = zk.TrustedSetup.setup(tau)
    sc = ora.random_fr(n, 600 + log_n)
    com = zk.MultilinearKZG.commitment(zk.Multilinear(sc), srs)
    if log_n <= 8:      # the reference algorithm itself (naive double-and-add) is affordable
        want = ora.kzg_commitment(sc, ora.kzg_multilinear_srs_g1(tau), True)
    else:               # commit == p(tau) * G  (and the CPU bucket method agrees)
        p_tau = ora.fr_to_ints(ora.mle_evaluation(sc, tau))[0]
        want = ora.g1_mul_int(ora.g1_generator(), p_tau)
        pts = np.concatenate([srs.powers_of_tau_in_g1.cpu().numpy().view(np.uint64),
                              srs.inf.cpu().numpy().astype(np.uint64)[:, None]], axis=1)
        assert ora.g1_affine_ints(ora.g1_to_affine(ora.msm_pippenger(sc, pts))) == ora.g1_affine_ints(ora.g1_to_affine(want))
    _same(zk, com, *_aff(ora, want))


def test_commit_2_20_identity(zk, ora):
    """BASELINE config 3 size: 2^20-point SRS generated on the device from tau; commit == p(tau) * G where
    p(tau) comes from the oracle's evaluation of the same table and the product from the oracle."""
    import torch
    log_n = 20
    tau = ora.random_fr(log_n, 4242)
    srs = zk.TrustedSetup.setup(tau)
    g = torch.Generator(device="cuda").manual_seed(7)
    t = torch.randint(0, 2 ** 62, (1 << log_n, 4), dtype=torch.int64, device="cuda", generator=g)
    poly = zk.Multilinear(t)
    com = zk.MultilinearKZG.commitment(poly, srs)
    p_tau = ora.fr_to_ints(ora.mle_evaluation(t.cpu().numpy().view(np.uint64), tau))[0]
    _same(zk, com, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), p_tau)))


@pytest.mark.parametrize("log_n", [12, 14, 17])
@pytest.mark.parametrize("kind", ["ones", "bits", "bytes", "minus_one", "uniform"])
def test_commit_skewed_scalars(zk, ora, log_n, kind):
    """Scalars that pile most points into a few buckets (constant / 0-1 / small-valued tables -- what multilinear
    tables usually hold -- and, for `uniform` at 2^14 / 2^17, the sparse top window): the heavy buckets are split into
    chunks (msm_kernels.hpp pass 4b/4c).  commit == p(tau) * G whatever the bucket population is."""
    n = 1 << log_n
    tau = ora.random_fr(log_n, 900 + log_n)
    srs = zk.TrustedSetup.setup(tau)
    rng = np.random.default_rng(log_n)
    if kind == "ones":
        ints = np.ones(n, dtype=np.int64)
    elif kind == "bits":
        ints = rng.integers(0, 2, n)
    elif kind == "bytes":
        ints = rng.integers(0, 256, n)
    elif kind == "minus_one":
        ints = -np.ones(n, dtype=np.int64)
    else:
        ints = None
    sc = ora.random_fr(n, 77 + log_n) if ints is None else zk.Fr.from_ints([int(v) for v in ints])
    poly = zk.Multilinear(sc)
    com = zk.MultilinearKZG.commitment(poly, srs)
    p_tau = ora.fr_to_ints(ora.mle_evaluation(np.ascontiguousarray(sc), tau))[0]     # the oracle's p(tau), not the GPU's
    _same(zk, com, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), p_tau)))


# ---- MultilinearKZG::open (multilinear_kzg.rs:50-88) ---------------------------------------------------------
@pytest.mark.parametrize("vals,tau,z", [
    ([0, 7, 0, 5, 0, 7, 4, 9], [2, 3, 4], [5, 9, 6]),                                                      # test_kzg_1
    ([0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4], [12, 9, 28, 40], [54, 90, 76, 160]),       # test_kzg_2
])
@pytest.mark.parametrize("cached", [True, False])
def test_kzg_open_reference_data(zk, ora, vals, tau, z, cached):
    srs = zk.TrustedSetup.setup(zk.Fr.from_ints(tau))
    proof = zk.MultilinearKZG.open(zk.Multilinear(zk.Fr.from_ints(vals)), zk.Fr.from_ints(z), srs, cache_folded_srs=cached)
    want_ev, want_proofs = ora.kzg_open(zk.Fr.from_ints(vals), zk.Fr.from_ints(z), ora.kzg_multilinear_srs_g1(zk.Fr.from_ints(tau)))
    assert np.array_equal(proof.evaluation, want_ev)
    if vals[1] == 7:
        assert zk.Fr.to_ints(proof.evaluation) == [114]
    assert len(proof.proofs) == len(tau)
    for got, want in zip(proof.proofs, want_proofs):
        _same(zk, got, *_aff(ora, want))


@pytest.mark.parametrize("n_vars", [2, 5, 8])
def test_kzg_open_random_matches_naive_oracle(zk, ora, n_vars):
    tau, z = ora.random_fr(n_vars, 1100 + n_vars), ora.random_fr(n_vars, 1200 + n_vars)
    vals = ora.random_fr(1 << n_vars, 1300 + n_vars)
    srs = zk.TrustedSetup.setup(tau)
    proof = zk.MultilinearKZG.open(zk.Multilinear(vals), z, srs)
    want_ev, want_proofs = ora.kzg_open(vals, z, ora.kzg_multilinear_srs_g1(tau))
    assert np.array_equal(proof.evaluation, want_ev)
    for got, want in zip(proof.proofs, want_proofs):
        _same(zk, got, *_aff(ora, want))


# ---- the same openings against the level tables (TrustedSetup.precompute_open: zkhip_srs_level_tables / zkhip_kzg_open_tables) --------
@pytest.mark.parametrize("vals,tau,z", [
    ([0, 7, 0, 5, 0, 7, 4, 9], [2, 3, 4], [5, 9, 6]),                                                      # test_kzg_1
    ([0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4], [12, 9, 28, 40], [54, 90, 76, 160]),       # test_kzg_2
])
def test_kzg_open_level_tables_reference_data(zk, ora, vals, tau, z):
    srs = zk.TrustedSetup.setup(zk.Fr.from_ints(tau)).precompute_open()
    assert srs.level_tables is not None
    proof = zk.MultilinearKZG.open(zk.Multilinear(zk.Fr.from_ints(vals)), zk.Fr.from_ints(z), srs)
    want_ev, want_proofs = ora.kzg_open(zk.Fr.from_ints(vals), zk.Fr.from_ints(z), ora.kzg_multilinear_srs_g1(zk.Fr.from_ints(tau)))
    assert np.array_equal(proof.evaluation, want_ev)
    for got, want in zip(proof.proofs, want_proofs):
        _same(zk, got, *_aff(ora, want))


@pytest.mark.parametrize("n_vars", [2, 3, 5, 8])
def test_kzg_open_level_tables_random_matches_naive_oracle(zk, ora, n_vars):
    tau, z = ora.random_fr(n_vars, 2100 + n_vars), ora.random_fr(n_vars, 2200 + n_vars)
    vals = ora.random_fr(1 << n_vars, 2300 + n_vars)
    srs = zk.TrustedSetup.setup(tau).precompute_open()
    proof = zk.MultilinearKZG.open(zk.Multilinear(vals), z, srs)
    want_ev, want_proofs = ora.kzg_open(vals, z, ora.kzg_multilinear_srs_g1(tau))
    assert np.array_equal(proof.evaluation, want_ev)
    for got, want in zip(proof.proofs, want_proofs):
        _same(zk, got, *_aff(ora, want))


@pytest.mark.parametrize("n_vars,kind", [(10, "random"), (13, "random"), (16, "random"), (12, "small"), (12, "equal"), (12, "zero_tau"),
                                         (12, "minus_one"), (20, "random"), (21, "random")])
def test_kzg_open_level_tables_agree_with_the_plain_opening(zk, ora, n_vars, kind):
    """Every proof point of an opening against the level tables equals the one the plain batch (whose own parity is pinned by the oracle
    tests above and by the exponent identity below) delivers: dense and degenerate quotients (tiny values: sparse top windows; one
    repeated value: every point of a window in ONE bucket, the heavy-bucket passes; all -1: negative digits throughout), an SRS with
    points at infinity (tau_0 = 0 zeroes half of it), and 2^21, where the first round is a commit of its own beside the tabled batch."""
    n = 1 << n_vars
    tau = ora.random_fr(n_vars, 3100 + n_vars)
    if kind == "zero_tau":
        tau = np.ascontiguousarray(tau)
        tau[0] = zk.Fr.from_int(0)
    z = ora.random_fr(n_vars, 3200 + n_vars)
    rng = np.random.default_rng(3300 + n_vars)
    if kind == "small":
        vals = zk.Fr.from_ints([int(v) for v in rng.integers(0, 4, n)])
    elif kind == "equal":
        # q_i = f(1, .) - f(0, .): a table whose halves differ by one constant gives a first quotient of n/2 equal entries
        half = ora.random_fr(n // 2, 3400)
        lo = zk.Fr.to_ints(half)
        vals = zk.Fr.from_ints(lo + [v + 0x1234567890abcdef1234567890abcdef1234567890abcdef for v in lo])
    elif kind == "minus_one":
        vals = zk.Fr.from_ints([0] * (n // 2) + [-1] * (n // 2))
    else:
        vals = ora.random_fr(n, 3300 + n_vars)
    srs = zk.TrustedSetup.setup(tau)
    plain = zk.TrustedSetup(srs.powers_of_tau_in_g1.clone(), srs.inf.clone())
    srs.precompute_open()
    poly = zk.Multilinear(vals)
    # (an SRS of at most 2^12 points would build its level tables on first use: the plain side then folds per call, which keeps it on the bucket pipeline)
    a, b = zk.MultilinearKZG.open(poly, z, srs), zk.MultilinearKZG.open(poly, z, plain, cache_folded_srs=n > zk.TrustedSetup.SMALL_SRS)
    assert plain.level_tables is None and srs.level_tables is not None
    assert np.array_equal(a.evaluation, b.evaluation)
    assert len(a.proofs) == n_vars and all(p == q for p, q in zip(a.proofs, b.proofs))
    assert np.array_equal(a.evaluation, ora.mle_evaluation(np.ascontiguousarray(vals), z))


@pytest.mark.parametrize("n_vars", [14, 20])   # 20 = BASELINE config 3's SRS size
def test_kzg_open_exponent_identity(zk, ora, n_vars):
    """Beyond what the naive oracle can do in seconds: proof_i == Q_i(tau) * G, Q_i the round's quotient evaluated at
    the remaining tau, and the verifier's equation in the exponent.  Every scalar (quotients, remainders, p(tau), p(z))
    comes from the ORACLE's folds of the same table; the GPU supplies only the proof under test."""
    tau, z = ora.random_fr(n_vars, 71), ora.random_fr(n_vars, 72)
    vals = ora.random_fr(1 << n_vars, 73)
    srs = zk.TrustedSetup.setup(tau)
    poly = zk.Multilinear(vals)
    proof = zk.MultilinearKZG.open(poly, z, srs)
    assert np.array_equal(proof.evaluation, ora.mle_evaluation(vals, z))
    tau_i, z_i = zk.Fr.to_ints(tau), zk.Fr.to_ints(z)
    one, zero = zk.Fr.from_int(1), zk.Fr.from_int(0)
    f, acc = vals, 0
    for i in range(n_vars):
        # the quotient q = f(1, .) - f(0, .) (kzg/src/utils.rs:5-10) evaluated at the remaining tau; evaluation is linear
        hi, lo = ora.mle_partial_evaluation(f, one, 0), ora.mle_partial_evaluation(f, zero, 0)
        ev = (lambda t: ora.fr_to_ints(ora.mle_evaluation(t, tau[i + 1:]) if i + 1 < n_vars else t)[0])
        q_tau = (ev(hi) - ev(lo)) % R
        _same(zk, proof.proofs[i], *_aff(ora, ora.g1_mul_int(ora.g1_generator(), q_tau)))
        acc = (acc + q_tau * (tau_i[i] - z_i[i])) % R
        f = ora.mle_partial_evaluation(f, z[i], 0)
    p_tau = ora.fr_to_ints(ora.mle_evaluation(vals, tau))[0]
    assert (p_tau - zk.Fr.to_ints(proof.evaluation)[0]) % R == acc


def test_kzg_open_shape_panics(zk, ora):
    srs = zk.TrustedSetup.setup(zk.Fr.from_ints([2, 3, 4]))
    poly = zk.Multilinear(zk.Fr.from_ints([0, 7, 0, 5, 0, 7, 4, 9]))
    with pytest.raises(AssertionError):    # evaluation_form.rs:163-167
        zk.MultilinearKZG.open(poly, zk.Fr.from_ints([5, 9]), srs)
    with pytest.raises(AssertionError):    # multilinear_kzg.rs:36-41
        zk.MultilinearKZG.open(zk.Multilinear(zk.Fr.from_ints([0, 7, 0, 5])), zk.Fr.from_ints([5, 9]), srs)
    with pytest.raises(AssertionError):    # one variable: `variable_index - 1` underflows in the reference (:73)
        zk.MultilinearKZG.open(zk.Multilinear(zk.Fr.from_ints([3, 4])), zk.Fr.from_ints([5]),
                               zk.TrustedSetup.setup(zk.Fr.from_ints([2])))


# ---- UnivariateKZG::open (univariate_kzg.rs:60-81) -----------------------------------------------------------
def test_univariate_kzg_open_reference_data(zk, ora):   # univariate_kzg.rs:111-129: tau = 10, poly 1..5, opened at 2
    tau, z, coeffs = zk.Fr.from_int(10), zk.Fr.from_int(2), zk.Fr.from_ints([1, 2, 3, 4, 5])
    srs = zk.UnivariateKZG.generate_srs(tau, 4)
    proof = zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(coeffs), z, srs)
    want_ev, want_proof = ora.univariate_kzg_open(coeffs, z, ora.kzg_univariate_srs_g1(tau, 4))
    assert zk.Fr.to_ints(proof.evaluation) == [129] and np.array_equal(proof.evaluation, want_ev)
    _same(zk, proof.proof, *_aff(ora, want_proof))
    with pytest.raises(IndexError):   # quotient longer than the SRS (univariate_kzg.rs:75)
        zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(zk.Fr.from_ints([1, 2, 3, 4, 5, 6, 7])), z, srs)
    const = zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(zk.Fr.from_ints([7])), z, srs)   # degree 0: quotient zero
    assert zk.Fr.to_ints(const.evaluation) == [7] and const.proof.infinity


@pytest.mark.parametrize("n", [2, 9, 64, 300])
def test_univariate_kzg_open_random_matches_naive_oracle(zk, ora, n):
    tau, z = ora.random_fr(1, 2100 + n)[0], ora.random_fr(1, 2200 + n)[0]
    coeffs = ora.random_fr(n, 2300 + n)
    srs = zk.UnivariateKZG.generate_srs(tau, n - 1)
    proof = zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(coeffs), z, srs)
    want_ev, want_proof = ora.univariate_kzg_open(coeffs, z, ora.kzg_univariate_srs_g1(tau, n - 1))
    assert np.array_equal(proof.evaluation, want_ev)
    _same(zk, proof.proof, *_aff(ora, want_proof))


@pytest.mark.parametrize("n,z_int", [(5000, None), (1 << 16, None), ((1 << 16) + 123, None), (4097, 0), (4097, 1)])
def test_univariate_kzg_open_exponent_identity(zk, ora, n, z_int):
    """Sizes spanning several workgroups of the Horner scan: proof == q(tau) G with q(tau) = (p(tau) - p(z)) / (tau - z),
    p(.) from the oracle's Horner-free evaluate on the host (python ints)."""
    tau = ora.random_fr(1, 31)[0]
    z = ora.random_fr(1, 32)[0] if z_int is None else zk.Fr.from_int(z_int)
    coeffs = ora.random_fr(n, 33 + n)
    srs = zk.UnivariateKZG.generate_srs(tau, n - 1)
    proof = zk.UnivariateKZG.open(zk.DenseUnivariatePolynomial(coeffs), z, srs)
    ci, t, zi = zk.Fr.to_ints(coeffs), zk.Fr.to_ints(tau)[0], zk.Fr.to_ints(z)[0]

    def horner(x):
        acc = 0
        for c in reversed(ci):
            acc = (acc * x + c) % R
        return acc
    assert zk.Fr.to_ints(proof.evaluation) == [horner(zi)]
    q_tau = (horner(t) - horner(zi)) * pow(t - zi, -1, R) % R
    _same(zk, proof.proof, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), q_tau)))


@pytest.mark.parametrize("kind", ["uniform", "bits"])
@pytest.mark.parametrize("sizes", [[1, 2, 4, 8], [4096, 2048, 1024, 512, 256, 128, 64, 32, 16, 8, 4, 2, 1], [300, 17, 5000], [16384, 3],
                                   [131072, 5, 70000, 0, 1], [5] * 64, [700] * 50 + [9000]])
def test_commit_batch_matches_single_commits(zk, ora, sizes, kind):
    """zkhip_kzg_commit_batch (the rounds of MultilinearKZG::open share one pass, every problem with window widths of its own --
    here problems from 0 to 2^17 entries in one batch): every slice's commitment equals the stand-alone commitment of that slice
    (itself checked against the oracle above).  `bits`: 0 / 1 scalars, the heavy-bucket passes inside a batch."""
    from zk_cryptography_amd.kzg import commit_batch
    import torch
    total = sum(sizes)
    tau = ora.random_fr(1, 3100 + len(sizes))[0]
    srs = zk.UnivariateKZG.generate_srs(tau, total - 1)
    if kind == "bits":
        ints = np.random.default_rng(total).integers(0, 2, total)
        sc = torch.from_numpy(np.ascontiguousarray(zk.Fr.from_ints([int(v) for v in ints])).view(np.int64)).cuda()
    else:
        sc = torch.from_numpy(ora.random_fr(total, 3200 + total).view(np.int64)).cuda()
    offsets = [0]
    for s in sizes:
        offsets.append(offsets[-1] + s)
    got = commit_batch(srs.powers_of_tau_in_g1, srs.inf, sc, offsets)
    for j, s in enumerate(sizes):
        lo, hi = offsets[j], offsets[j + 1]
        sub = zk.TrustedSetup(srs.powers_of_tau_in_g1[lo:hi], srs.inf[lo:hi])
        want = zk.UnivariateKZG.commitment(zk.DenseUnivariatePolynomial(sc[lo:hi]), sub)
        assert got[j] == want, (j, s)


# ---- shifted-SRS table (zkhip_srs_precompute / zkhip_kzg_commit_table): same commitments, fewer bucket additions ----
def test_commit_table_reference_data(zk, ora):   # multilinear_kzg.rs:133-148 data through the table path
    vals = [0, 7, 0, 5, 0, 7, 4, 9]
    srs = zk.TrustedSetup.setup(zk.Fr.from_ints([2, 3, 4])).precompute()
    com = zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints(vals)), srs)
    x, y = com.coords()
    assert x == 0x16ad11e5d15f77c1143b1697344911b9c590110fdd8dd09df2e58bfd757269169deefe8be3544d4e049fb3776fb0bcfb
    assert y == 0x0f5c8be5f27fc19eee337785e43d18414a8ff04995230f04509800252164cf47887a4a1864f18288652196af6272e7f6
    with pytest.raises(AssertionError):
        zk.MultilinearKZG.commitment(zk.Multilinear(zk.Fr.from_ints([1, 2, 3, 4])), srs)


@pytest.mark.parametrize("log_n,kind", [(1, "uniform"), (2, "ones"), (4, "uniform"), (7, "bytes"), (10, "uniform"), (13, "bytes"), (16, "uniform"), (16, "ones"),
                                        (17, "uniform"), (18, "uniform"), (19, "uniform")])   # the table's window widths change at 2^17, 2^18, 2^19
def test_commit_table_matches_plain_commit(zk, ora, log_n, kind):
    n = 1 << log_n
    tau = ora.random_fr(log_n, 4100 + log_n)
    plain = zk.TrustedSetup.setup(tau)
    table = zk.TrustedSetup(plain.powers_of_tau_in_g1, plain.inf).precompute()
    rng = np.random.default_rng(log_n)
    if kind == "uniform":
        sc = ora.random_fr(n, 4200 + log_n)
    elif kind == "ones":
        sc = zk.Fr.from_ints([1] * n)
    else:
        sc = zk.Fr.from_ints([int(v) for v in rng.integers(0, 256, n)])
    poly = zk.Multilinear(sc)
    a, b = zk.MultilinearKZG.commitment(poly, plain), zk.MultilinearKZG.commitment(poly, table)
    assert a == b
    if log_n <= 10:   # and both equal p(tau) G
        p_tau = ora.fr_to_ints(ora.mle_evaluation(sc, tau))[0]
        _same(zk, b, *_aff(ora, ora.g1_mul_int(ora.g1_generator(), p_tau)))


@pytest.mark.parametrize("n_points,n_scalars,kind", [
    (1, 1, "uniform"), (2, 2, "uniform"), (8, 8, "zeros"), (8, 8, "minus_one"), (64, 64, "uniform"), (256, 256, "uniform"), (256, 256, "ones"),
    (256, 100, "uniform"), (300, 300, "bytes"), (1024, 1024, "uniform"), (2048, 1000, "bits"), (4096, 4096, "uniform"), (4096, 4095, "minus_one"),
    (4096, 1, "uniform")])
def test_short_commits_cross_the_bucket_pipeline(zk, ora, n_points, n_scalars, kind):
    """zkhip_kzg_commit_table's short path (<= 2^12 scalars: one plain sum per digit bit, csrc/msm.hip msm_commit_small) against the
    bucket pipeline on the plain SRS (zkhip_kzg_commit: sort, buckets, reductions -- no table, no short path) on the same inputs:
    univariate SRS with n_scalars <= n_points (the table's stride is the SRS size), a point at infinity in the SRS, scalars of every
    kind; the naive oracle as a third opinion where it is quick."""
    from zk_cryptography_amd import kzg as K
    srs = zk.UnivariateKZG.generate_srs(zk.Fr.from_int(3 + n_points), n_points - 1)
    assert len(srs) == n_points
    if n_points >= 8:
        srs.inf[5] = 1                                       # a point at infinity: both paths must skip it
        srs.invalidate()
    table = zk.TrustedSetup(srs.powers_of_tau_in_g1, srs.inf).precompute().table
    rng = np.random.default_rng(n_points * 7 + n_scalars)
    if kind == "uniform":
        sc = ora.random_fr(n_scalars, 5100 + n_scalars)
    elif kind == "zeros":
        sc = zk.Fr.from_ints([0] * n_scalars)
    elif kind == "ones":
        sc = zk.Fr.from_ints([1] * n_scalars)
    elif kind == "minus_one":
        sc = zk.Fr.from_ints([R - 1] * n_scalars)
    elif kind == "bits":
        sc = zk.Fr.from_ints([int(v) for v in rng.integers(0, 2, n_scalars)])
    else:
        sc = zk.Fr.from_ints([int(v) for v in rng.integers(0, 256, n_scalars)])
    d_sc = zk.DenseUnivariatePolynomial(sc).coefficients
    short = K._commit(srs.powers_of_tau_in_g1, srs.inf, n_points, d_sc, n_scalars, False, table)
    buckets = K._commit(srs.powers_of_tau_in_g1, srs.inf, n_points, d_sc, n_scalars, False, None)
    assert short == buckets
    if n_scalars <= 300:
        jac = ora.kzg_univariate_srs_g1(zk.Fr.from_int(3 + n_points), n_points - 1).reshape(-1, 18).copy()
        if n_points >= 8:
            jac[5] = ora.g1_identity()
        _same(zk, short, *_aff(ora, ora.kzg_commitment(sc, jac, False)))


def test_commit_table_with_identity_points_and_short_polynomial(zk, ora):
    tau = zk.Fr.from_ints([0, 1, 2, 3])     # SRS with points at infinity (kzg benches' tau)
    plain = zk.TrustedSetup.setup(tau)
    table = zk.TrustedSetup(plain.powers_of_tau_in_g1, plain.inf).precompute()
    vals = zk.Fr.from_ints([0, 5, 0, 0, 7, 0, 1, R - 1, 2, 3, 0, 0, 9, 9, 9, 1])
    assert zk.MultilinearKZG.commitment(zk.Multilinear(vals), plain) == zk.MultilinearKZG.commitment(zk.Multilinear(vals), table)
    usrs = zk.UnivariateKZG.generate_srs(zk.Fr.from_int(10), 40)
    utab = zk.TrustedSetup(usrs.powers_of_tau_in_g1, usrs.inf).precompute()
    coeffs = zk.DenseUnivariatePolynomial(ora.random_fr(17, 5))     # shorter than the SRS: table stride != n_scalars
    assert zk.UnivariateKZG.commitment(coeffs, usrs) == zk.UnivariateKZG.commitment(coeffs, utab)


def test_srs_caches_follow_in_place_edits(zk, ora):
    """The shifted-SRS table and the folded levels are derived from the SRS tensors once -- and again when those tensors are edited in
    place or replaced (torch's version counters are part of the cache key): never a stale table."""
    tau = ora.random_fr(8, 321)
    srs = zk.TrustedSetup.setup(tau).precompute()
    poly = zk.Multilinear(ora.random_fr(256, 322))
    before = zk.MultilinearKZG.commitment(poly, srs)
    z = ora.random_fr(8, 323)
    zk.MultilinearKZG.open(poly, z, srs)                                   # caches the folded levels too
    stale = srs.table
    assert stale is not None
    srs.powers_of_tau_in_g1[[0, 1]] = srs.powers_of_tau_in_g1[[1, 0]]      # in place: the first two points swapped
    plain = zk.TrustedSetup(srs.powers_of_tau_in_g1.clone(), srs.inf.clone())
    assert srs.table is None                                               # the stale table was dropped ...
    after = zk.MultilinearKZG.commitment(poly, srs)                        # ... (a small SRS builds a fresh one on its next commitment)
    assert srs.table is not stale and not (after == before) and after == zk.MultilinearKZG.commitment(poly, plain)
    assert zk.MultilinearKZG.commitment(poly, srs.precompute()) == after   # and a table rebuilt from the edited points agrees
    a, b = zk.MultilinearKZG.open(poly, z, srs), zk.MultilinearKZG.open(poly, z, plain, cache_folded_srs=False)
    assert all(p == q for p, q in zip(a.proofs, b.proofs))
    # the level tables of `open` follow the same rule
    srs.precompute_open()
    assert srs.level_tables is not None
    srs.powers_of_tau_in_g1[[2, 3]] = srs.powers_of_tau_in_g1[[3, 2]]
    plain2 = zk.TrustedSetup(srs.powers_of_tau_in_g1.clone(), srs.inf.clone())
    stale_levels = srs._level_tables
    assert srs.level_tables is None                                        # the stale level tables were dropped ...
    a, b = zk.MultilinearKZG.open(poly, z, srs), zk.MultilinearKZG.open(poly, z, plain2, cache_folded_srs=False)
    assert srs.level_tables is not stale_levels and all(p == q for p, q in zip(a.proofs, b.proofs))   # ... (a small SRS builds fresh ones on its next opening)
    a = zk.MultilinearKZG.open(poly, z, srs.precompute_open())
    assert srs.level_tables is not None and all(p == q for p, q in zip(a.proofs, b.proofs))


def test_srs_caches_follow_raw_pointer_writes(zk, ora):
    """A write into the SRS tensors through their raw pointers (what the library's own kernels do: no torch version bump) is caught by
    the content fingerprint of the cache key; TrustedSetup.invalidate() is the explicit form."""
    import ctypes as C
    from zk_cryptography_amd import _native as N
    tau = ora.random_fr(8, 421)
    srs = zk.TrustedSetup.setup(tau).precompute()
    poly = zk.Multilinear(ora.random_fr(256, 422))
    before = zk.MultilinearKZG.commitment(poly, srs)
    zk.MultilinearKZG.open(poly, ora.random_fr(8, 423), srs)
    version = srs.powers_of_tau_in_g1._version
    swapped = srs.powers_of_tau_in_g1[[1, 0]].cpu().numpy().copy()
    ctx = N.Context.get(srs.powers_of_tau_in_g1.device.index)
    N.check(N.lib().zkhip_memcpy_h2d(ctx.handle, C.c_void_p(srs.powers_of_tau_in_g1.data_ptr()), swapped.ctypes.data_as(C.c_void_p),
                                     C.c_size_t(swapped.nbytes)), "memcpy")
    assert srs.powers_of_tau_in_g1._version == version                     # torch saw nothing
    stale = srs._table
    plain = zk.TrustedSetup(srs.powers_of_tau_in_g1.clone(), srs.inf.clone())
    assert srs.table is None                                               # the fingerprint did: the stale table was dropped ...
    after = zk.MultilinearKZG.commitment(poly, srs)                        # ... (a small SRS builds a fresh one on its next commitment)
    assert srs.table is not stale and not (after == before) and after == zk.MultilinearKZG.commitment(poly, plain)
    srs.precompute()
    assert srs.table is not None
    srs.invalidate()
    assert srs.table is None


# ---- commits in flight (zkhip_kzg_commit_begin / _end) ------------------------------------------------------------------
@pytest.mark.parametrize("table", [False, True])
def test_commits_in_flight_match_synchronous_commits(zk, ora, table):
    from zk_cryptography_amd import _native as N
    log_n = 14
    srs = zk.TrustedSetup.setup(ora.random_fr(log_n, 880))
    if table:
        srs.precompute()
    polys = [zk.Multilinear(ora.random_fr(1 << log_n, 890 + i)) for i in range(5)]
    want = [zk.MultilinearKZG.commitment(p, srs) for p in polys]
    got, pending = [], []
    for depth in (2, 3):                  # depth-3 pipeline, as bench.py issues them
        got, pending = [], []
        for p in polys:
            pending.append(zk.MultilinearKZG.commitment_begin(p, srs))
            if len(pending) == depth:
                got.append(pending.pop(0).wait())
        got += [h.wait() for h in pending]
        assert got == want
    # three in flight: a fourth begin, and anything else that needs the workspace, is refused until one has ended
    a, b, c3 = (zk.MultilinearKZG.commitment_begin(polys[i], srs) for i in range(3))
    with pytest.raises(N.ZkhipError, match="split-phase|lent"):
        zk.MultilinearKZG.commitment_begin(polys[3], srs)
    with pytest.raises(N.ZkhipError, match="split-phase|lent"):
        zk.MultilinearKZG.commitment(polys[3], srs)
    assert a.wait() == want[0]
    d4 = zk.MultilinearKZG.commitment_begin(polys[3], srs)     # the freed slot is taken again while two are still in flight
    assert c3.wait() == want[2] and d4.wait() == want[3]
    del b                                  # an abandoned commitment is drained and its slot released
    assert zk.MultilinearKZG.commitment(polys[2], srs) == want[2]
