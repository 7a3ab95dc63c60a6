"""GPU parity: composed and multi-composed sumcheck provers vs the CPU oracle (round polynomials, challenges,
proof bytes bit-exact).  Names follow sumcheck/src/composed/*.rs and composed_multilinear.rs tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def zk():
    import zk_cryptography_amd as z
    return z


def F(zk, v):
    return zk.Fr.from_ints(v)


def test_composed_multilinear_evaluation(zk):   # composed_multilinear.rs:134-155
    polys = zk.ComposedMultilinear([F(zk, [0, 1, 2, 3]), F(zk, [0, 0, 0, 1])])
    assert polys.evaluation(F(zk, [2, 3])) == 42
    assert polys.partial_evaluation(F(zk, [2])[0], 0).evaluation(F(zk, [3])) == 42
    assert polys.n_vars() == 2 and polys.max_degree() == 2


def _host(t):
    return t.cpu().numpy().view(np.uint64)


def test_element_wise_product_and_add(zk):   # composed_multilinear.rs:159-184
    polys = zk.ComposedMultilinear([F(zk, [0, 1, 2, 3]), F(zk, [0, 0, 0, 1])])
    assert zk.Fr.to_ints(_host(polys.element_wise_product())) == [0, 0, 0, 3]
    assert zk.Fr.to_ints(_host(polys.element_wise_add())) == [0, 1, 2, 4]


@pytest.mark.parametrize("k,log_n", [(1, 0), (2, 3), (3, 12), (5, 16), (11, 10)])
def test_element_wise_random_matches_host_arithmetic(zk, ora, k, log_n):
    """Any number of tables (the reference's fold over `polys` has no cap): 11 tables take two launches."""
    n = 1 << log_n
    tabs = [ora.random_fr(n, 6100 + 17 * k + q) for q in range(k)]
    polys = zk.ComposedMultilinear(tabs)
    R = zk.Fr.MODULUS
    ints = [zk.Fr.to_ints(t) for t in tabs]            # python integers: an independent check of the kernels' field arithmetic
    prod, add = list(ints[0]), list(ints[0])
    for t in ints[1:]:
        prod = [a * b % R for a, b in zip(prod, t)]
        add = [(a + b) % R for a, b in zip(add, t)]
    assert zk.Fr.to_ints(_host(polys.element_wise_product())) == prod
    assert zk.Fr.to_ints(_host(polys.element_wise_add())) == add
    # sum_over_boolean_hypercube (sumcheck/src/utils.rs:45-51) = the sum of the product vector
    if k <= 5:
        assert zk.Fr.to_ints(zk.ComposedSumcheck.calculate_poly_sum(polys))[0] == sum(prod) % R


def test_sum_calculation(zk):   # composed_sumcheck.rs:108-140, multi_composed_sumcheck.rs:195-214
    cs = zk.ComposedSumcheck.calculate_poly_sum
    assert zk.Fr.to_ints(cs(zk.ComposedMultilinear([F(zk, [0, 1, 2, 3]), F(zk, [0, 0, 0, 1])]))) == [3]
    assert zk.Fr.to_ints(cs(zk.ComposedMultilinear([F(zk, [3, 3, 5, 5]), F(zk, [0, 0, 0, 1])]))) == [5]
    assert zk.Fr.to_ints(cs(zk.ComposedMultilinear([F(zk, [0, 1, 2, 3])]))) == [6]
    assert zk.Fr.to_ints(cs(zk.ComposedMultilinear([F(zk, [0, 0, 0, 2, 2, 2, 2, 4])]))) == [12]
    ms = zk.MultiComposedSumcheckProver.calculate_poly_sum
    assert zk.Fr.to_ints(ms([zk.ComposedMultilinear([F(zk, [0, 1, 2, 3])]), zk.ComposedMultilinear([F(zk, [0, 0, 0, 1])])])) == [7]
    assert zk.Fr.to_ints(ms([zk.ComposedMultilinear([F(zk, [0, 0, 0, 2])]), zk.ComposedMultilinear([F(zk, [0, 3, 0, 3])])])) == [8]


COMPOSED_CASES = [
    [[3, 3, 5, 5], [0, 0, 0, 1]],
    [[0, 0, 2, 7, 3, 3, 6, 11]],
    [[0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0]],
    [[1, 3, 5, 7, 2, 4, 6, 8, 3, 5, 7, 9, 4, 6, 8, 10]],
]


@pytest.mark.parametrize("tables", COMPOSED_CASES)
def test_sum_check_proof(zk, ora, tables):   # composed_sumcheck.rs:143-241
    t = np.stack([F(zk, v) for v in tables])
    proof, ch = zk.ComposedSumcheck(zk.ComposedMultilinear(list(t))).prove()
    rp, och = ora.composed_prove(t)
    assert np.array_equal(proof.round_polys, rp) and np.array_equal(ch, och)
    assert ora.composed_verify(t, ora.composed_sum(t), proof.round_polys)


@pytest.mark.parametrize("k,log_n", [(1, 1), (2, 5), (2, 12), (3, 9), (5, 8), (5, 13), (4, 10),
                                     (2, 20), (3, 18), (2, 21), (2, 22), (5, 20)])   # (2, 20) .. (2, 22): grids above the workgroup cap (grid-stride rounds); (2, 21) / (2, 22): one / two rounds on the unreduced K = 2 sums; (5, 20): the first round's last factor on the matrix cores (composed_dot.hpp), four steps per workgroup
def test_composed_prove_random(zk, ora, k, log_n):   # benches: 2 and 5 tables (composed_sumcheck_benchmark.rs)
    t = np.stack([ora.random_fr(1 << log_n, 900 + 10 * k + q) for q in range(k)])
    poly = zk.ComposedMultilinear(list(t))
    proof, ch = zk.ComposedSumcheck(poly).prove()
    rp, och = ora.composed_prove(t)
    assert np.array_equal(proof.round_polys, rp)
    assert np.array_equal(ch, och)
    assert np.array_equal(zk.ComposedSumcheck.calculate_poly_sum(poly), ora.composed_sum(t))


def _gkr_example(zk, ora):   # multi_composed_sumcheck.rs:266-311
    add_i, mul_i = F(zk, [4, 4, 7, 7, 4, 4, 7, 9]), F(zk, [3, 3, 3, 4, 3, 3, 5, 6])
    w_b, w_c = zk.Multilinear(F(zk, [0, 4])), zk.Multilinear(F(zk, [0, 3]))
    two = F(zk, [2])[0]
    lhs = [zk.Multilinear(add_i).partial_evaluation(two, 0), w_b.add_distinct(w_c)]
    rhs = [zk.Multilinear(mul_i).partial_evaluation(two, 0), w_b.mul_distinct(w_c)]
    return [lhs, rhs]


def _multi_cases(zk, ora):
    p1, p2 = F(zk, [0, 0, 0, 2]), F(zk, [0, 3, 0, 3])
    mk = lambda tabs: [zk.Multilinear(t) for t in tabs]
    return [[mk([p1]), mk([p2])], [mk([p1]), mk([p2]), mk([p2])], [mk([p1, p2]), mk([p2, p1])], _gkr_example(zk, ora)]


def _check_multi(zk, ora, terms, partial):
    poly = [zk.ComposedMultilinear(t) for t in terms]
    flat = np.stack([m.to_numpy() for t in terms for m in t])
    sizes = [len(t) for t in terms]
    s = zk.MultiComposedSumcheckProver.calculate_poly_sum(poly)
    assert np.array_equal(s, ora.multi_composed_sum(flat, sizes))
    fn = zk.MultiComposedSumcheckProver.prove_partial if partial else zk.MultiComposedSumcheckProver.prove
    proof, ch = fn(poly, s)
    orps, och = ora.multi_composed_prove(flat, sizes, s, partial=partial)
    assert [rp.monomials() for rp in proof.round_polys] == [o.monomials() for o in orps]
    assert proof.to_bytes() == ora.multi_composed_proof_bytes(orps)
    assert np.array_equal(ch, och)
    return flat, sizes, s, orps


@pytest.mark.parametrize("stage", ["0", "1"])
def test_composed_forms_forced(stage):
    """Large claims of two-table terms take two rounds per pass by default (csrc/composed_stage.hpp, cross sums on the matrix cores);
    ZKHIP_STAGE=0 forces the round form (its unreduced K = 2 kernel is otherwise idle), ZKHIP_STAGE=1 the stage form from 2^15 entries on
    (the VALU cross sums below 2^12 indices per block): both must give the oracle's proof."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys; sys.path.insert(0, %r)
import numpy as np
import zk_cryptography_amd as zk
from oracle import oracle as ora
for k, log_n in ((2, 21), (2, 16)):
    t = np.stack([ora.random_fr(1 << log_n, 4000 + 10 * log_n + q) for q in range(k)])
    proof, ch = zk.ComposedSumcheck(zk.ComposedMultilinear(list(t))).prove()
    rp, och = ora.composed_prove(t)
    assert np.array_equal(proof.round_polys, rp) and np.array_equal(ch, och), (k, log_n)
flat = np.stack([ora.random_fr(1 << 17, 4500 + q) for q in range(4)])
s = ora.multi_composed_sum(flat, [2, 2])
terms = [zk.ComposedMultilinear([zk.Multilinear(flat[0]), zk.Multilinear(flat[1])]), zk.ComposedMultilinear([zk.Multilinear(flat[2]), zk.Multilinear(flat[3])])]
proof, ch = zk.MultiComposedSumcheckProver.prove_partial(terms, s)
orps, och = ora.multi_composed_prove(flat, [2, 2], s, partial=True)
assert proof.to_bytes() == ora.multi_composed_proof_bytes(orps) and np.array_equal(ch, och)
print("forms ok")
""" % root
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ZKHIP_STAGE=stage), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "forms ok" in res.stdout, res.stdout[-2000:] + res.stderr[-2000:]


@pytest.mark.parametrize("case", range(4))
@pytest.mark.parametrize("partial", [False, True])
def test_multi_composed_sumcheck_proof(zk, ora, case, partial):   # multi_composed_sumcheck.rs:217-311
    flat, sizes, s, orps = _check_multi(zk, ora, _multi_cases(zk, ora)[case], partial)
    if not partial:
        assert ora.multi_composed_verify(flat, sizes, s, orps) == 1


@pytest.mark.parametrize("sizes,log_n", [([2, 3], 8), ([2, 2], 12), ([1, 5], 6), ([3], 10), ([2, 2, 1, 1], 7), ([2, 2], 20)])
@pytest.mark.parametrize("partial", [False, True])
def test_multi_composed_random(zk, ora, sizes, log_n, partial):   # bench shape: 2 + 3 tables x 2^8
    terms, seed = [], 1000
    for k in sizes:
        terms.append([zk.Multilinear(ora.random_fr(1 << log_n, seed + q)) for q in range(k)])
        seed += 17
    _check_multi(zk, ora, terms, partial)


def test_multi_composed_zero_coefficient_semantics(zk, ora):
    # a term whose round polynomial has vanishing coefficients (dropped at interpolation, sparse_univariate.rs:55)
    # next to one that cancels it (zero kept after addition, :159-203)
    R = zk.Fr.MODULUS
    a = zk.Multilinear(F(zk, [1, 2, 3, 4]))
    neg_a = zk.Multilinear(F(zk, [R - 1, R - 2, R - 3, R - 4]))
    const = zk.Multilinear(F(zk, [5, 5, 5, 5]))
    _check_multi(zk, ora, [[a], [neg_a]], True)
    _check_multi(zk, ora, [[const], [a]], True)
    _check_multi(zk, ora, [[const, const], [neg_a]], False)


@pytest.mark.parametrize("grid", [None, "2"])
def test_five_table_rounds_on_the_matrix_cores_at_every_size(grid):
    """composed_round_dot_kernel (K = 5: the last factor of every index as a byte GEMM, csrc/composed_dot.hpp) serves rounds of >= 2^18
    output pairs by default.  Here the prover tests above run once more in a process where it takes every round of >= 256 pairs
    (ZKHIP_ROUND_DOT_MIN_LOG=8), first and folding rounds, one step per workgroup and -- with two workgroups per round -- up to eight:
    bit-identical to the oracle and the golden vectors."""
    import os, subprocess, sys
    env = dict(os.environ, ZKHIP_ROUND_DOT_MIN_LOG="8")
    if grid:
        env["ZKHIP_ROUND_GRID"] = grid
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_composed.py"), os.path.join(here, "test_golden.py"), "-m", "gpu", "-x", "-q",
                          "-k", "(test_composed_prove_random and 5-) or test_multi_composed_random or test_hip_path_reproduces_golden or test_sum_check_proof"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1200)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
