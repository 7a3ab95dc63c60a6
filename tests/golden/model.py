"""An INDEPENDENT model of the hot path: python ints + hashlib, nothing else.

This file imports neither `oracle/` nor the package under test.  It was written from the reference's Rust
(citations are path:line under /root/reference) and is the generator of tests/golden/hot_path_vectors.json, so
that three implementations must agree on every vector: this model, the C oracle (-m "not gpu") and the HIP
path (-m gpu).  Field elements are canonical python ints mod r, group elements affine (x, y) tuples or None.

Third-party semantics used (ark-ff / ark-ec / sha2, not under /root/reference): `into_bigint().to_bytes_be()`
= 32-byte big-endian canonical integer; `from_be_bytes_mod_order` = that integer mod r; `Ord for Fp` compares
canonical integers; `get_root_of_unity(n)` squares the 2^32-th root down; `mul_bigint` / `+` are the group law.

`self_check()` pins the model on value KATs the reference's own tests hold -- it is run by make_golden.py
before anything is written and by tests/test_golden.py.
"""
import hashlib

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001          # BLS12-381 scalar field
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
TWO_ADICITY = 32
TWO_ADIC_ROOT = 10238227357739495823651030575849232062558860180284477541189508159991286009131
G1 = (3685416753713387016781088315183077757961620795782546409894578378688607592378376318836054947676345821548104185464507,
      1339506544944476473020471379941921221584933875938349620426543736416511423956333506472724655353366534992391756441569)


def fr(v):
    return v % R


def inv(v):
    return pow(v, R - 2, R)


def be32(v):
    """convert_field_to_byte (sumcheck/src/utils.rs:7-9)"""
    return (v % R).to_bytes(32, "big")


# ---------------------------------------------------------------------------------------------------------------
# transcript: transcripts/fiat-shamir/src/fiat_shamir.rs:10-40
class Transcript:
    def __init__(self):
        self.h = hashlib.sha256()

    def commit(self, data):                       # :17-19
        self.h.update(data)

    def challenge(self):                          # :21-25  finalize_reset, then the fresh hasher absorbs the digest
        d = self.h.digest()
        self.h = hashlib.sha256()
        self.h.update(d)
        return d

    def field(self):                              # :27-29
        return int.from_bytes(self.challenge(), "big") % R

    def n_fields(self, n):                        # :31-39
        return [self.field() for _ in range(n)]


# ---------------------------------------------------------------------------------------------------------------
# multilinear, evaluation form: polynomial/src/multilinear/evaluation_form.rs, polynomial/src/utils.rs:26-53
def pairs(n, k):
    """pick_pairs_with_random_index (utils.rs:26-53), written as the loop the reference runs"""
    assert n % 2 == 0 and k < n // 2
    out = []
    iters = 1 << k
    for _ in range(iters):
        rnd = []
        half = (n // iters) // 2
        for y in range(half):
            rnd.append((y + len(out) * 2, half + y + len(out) * 2))
        out.extend(rnd)
    return out


def n_vars(n):
    nv = n.bit_length() - 1
    assert 1 << nv == n, "Number of evaluations must be a power of 2"     # evaluation_form.rs:12-26
    return nv


def fold(ev, r, k=0):
    """partial_evaluation (evaluation_form.rs:123-141): r*y2 + (1-r)*y1"""
    return [(r * ev[j] + (1 - r) * ev[i]) % R for i, j in pairs(len(ev), k)]


def folds(ev, pts, ks):                           # partial_evaluations :143-159
    assert len(pts) == len(ks)
    for p, k in zip(pts, ks):
        ev = fold(ev, p, k)
    return ev


def evaluate(ev, pts):                            # evaluation :162-175
    assert len(pts) == n_vars(len(ev))
    for p in pts:
        ev = fold(ev, p, 0)
    return ev[0]


def half_sums(ev):                                # split_poly_into_two_and_sum_each_part :68-74
    m = len(ev) // 2
    return [sum(ev[:m]) % R, sum(ev[m:]) % R]


def add_distinct(a, b):                           # :28-39
    return [(x + y) % R for x in a for y in b]


def mul_distinct(a, b):                           # :41-52
    return [(x * y) % R for x in a for y in b]


def add_to_front(ev, variable_length):            # :86-96
    return list(ev) * (2 << variable_length)


def table_bytes(ev):                              # to_bytes :54-62
    return b"".join(be32(v) for v in ev)


# ---------------------------------------------------------------------------------------------------------------
# sparse univariate: polynomial/src/univariate/sparse_univariate.rs, lagrange_basis polynomial/src/utils.rs:78-100
def lagrange_basis(xs, i):
    l = [1]
    for j, xj in enumerate(xs):
        if j != i:
            nl = [0] * (len(l) + 1)
            for k, c in enumerate(l):
                nl[k] = (nl[k] - c * xj) % R
                nl[k + 1] = (nl[k + 1] + c) % R
            l = nl
    den = 1
    for j, xj in enumerate(xs):
        if j != i:
            den = den * (xs[i] - xj) % R
    di = inv(den)
    return [c * di % R for c in l]


def interpolation(ys, xs=None):
    """SparseUnivariatePolynomial::interpolation (:40-63); the provers interpolate over x = 0..d
    (convert_round_poly_to_uni_poly_format, sumcheck/src/utils.rs:29-35).  List of (coeff, pow), ZERO coefficients dropped (:55)"""
    xs = list(range(len(ys))) if xs is None else [x % R for x in xs]
    res = [0] * len(ys)
    for i, y in enumerate(ys):
        for k, c in enumerate(lagrange_basis(xs, i)):
            res[k] = (res[k] + c * y) % R
    return [(c, p) for p, c in enumerate(res) if c != 0]


def sparse_add(a, b):
    """impl Add (:159-203): ordered merge on pow; an equal pow keeps the SUM, even a zero one"""
    out, i, j = [], 0, 0
    while i < len(a) or j < len(b):
        if i < len(a) and j < len(b):
            if a[i][1] == b[j][1]:
                out.append(((a[i][0] + b[j][0]) % R, a[i][1])); i += 1; j += 1
            elif a[i][1] < b[j][1]:
                out.append(a[i]); i += 1
            else:
                out.append(b[j]); j += 1
        elif i < len(a):
            out.append(a[i]); i += 1
        else:
            out.append(b[j]); j += 1
    return out


def sparse_bytes(p):                              # to_bytes :27-34: coeff || pow per monomial
    return b"".join(be32(c) + be32(w) for c, w in p)


def sparse_eval(p, x):                            # evaluate :90-106
    return sum(c * pow(x, w, R) for c, w in p) % R


# ---------------------------------------------------------------------------------------------------------------
# provers
def sumcheck_prove(ev):
    """Sumcheck::poly_sum + prove (sumcheck/src/sumcheck.rs:25-61) -> (sum, round polys, challenges)"""
    s = sum(ev) % R
    t = Transcript()
    t.commit(be32(s))
    rps, chs = [], []
    cur = list(ev)
    for _ in range(n_vars(len(ev))):
        u = half_sums(cur)
        t.commit(table_bytes(u))
        rps.append(u)
        r = t.field()
        chs.append(r)
        cur = fold(cur, r, 0)
    return s, rps, chs


def product_sums(tables):
    """Σ_x Π_k f_k(x): element_wise_product().iter().sum() (composed_multilinear.rs:105-111)"""
    tot = 0
    for i in range(len(tables[0])):
        pr = 1
        for t in tables:
            pr = pr * t[i] % R
        tot += pr
    return tot % R


def round_evals(tables):
    """evaluations at t = 0..=max_degree of Σ Π fold(f_k, t) (composed_sumcheck.rs:41-49)"""
    return [product_sums([fold(t, x, 0) for t in tables]) for x in range(len(tables) + 1)]


def composed_prove(tables):
    """ComposedSumcheck::prove (composed_sumcheck.rs:32-67): raw evaluations committed (vec_to_bytes)"""
    t = Transcript()
    cur = [list(x) for x in tables]
    rps, chs = [], []
    for _ in range(n_vars(len(tables[0]))):
        rp = round_evals(cur)
        t.commit(table_bytes(rp))
        r = t.field()
        chs.append(r)
        rps.append(rp)
        cur = [fold(x, r, 0) for x in cur]
    return rps, chs


def multi_composed_sum(terms):                    # multi_composed_sumcheck.rs:36-45
    return sum(product_sums(term) for term in terms) % R


def multi_composed_internal(terms, s, t):
    """prove_internal (multi_composed_sumcheck.rs:64-121) on transcript t -> (sparse round polys, challenges)"""
    t.commit(be32(s))
    cur = [[list(x) for x in term] for term in terms]
    rps, chs = [], []
    for _ in range(n_vars(len(terms[0][0]))):
        rp = []
        for term in cur:
            rp = sparse_add(rp, interpolation(round_evals(term)))
        t.commit(sparse_bytes(rp))
        r = t.field()
        cur = [[fold(x, r, 0) for x in term] for term in cur]
        chs.append(r)
        rps.append(rp)
    return rps, chs


def multi_composed_prove(terms, s, partial):
    t = Transcript()
    if not partial:                               # prove :47-54 absorbs composed_poly_to_bytes first
        t.commit(b"".join(table_bytes(x) for term in terms for x in term))
    return multi_composed_internal(terms, s, t)


def proof_bytes(rps):                             # ComposedSumcheckProof::to_bytes :24-31
    return b"".join(sparse_bytes(rp) for rp in rps)


def multi_composed_verify_partial(s, rps):
    """verify_internal (:149-181) on a fresh transcript -> (final claim, challenges) or None"""
    t = Transcript()
    t.commit(be32(s))
    claim, chs = s, []
    for rp in rps:
        t.commit(sparse_bytes(rp))
        c = t.field()
        chs.append(c)
        if claim != (sparse_eval(rp, 0) + sparse_eval(rp, 1)) % R:
            return None
        claim = sparse_eval(rp, c)
    return claim, chs


# ---------------------------------------------------------------------------------------------------------------
# circuit + GKR: circuit/src/circuit.rs:31-97, circuit/src/utils.rs:1-34, gkr/src/protocol.rs:21-116, gkr/src/utils.rs:8-56
def circuit_evaluation(layers, inp):
    """layers[0] = output layer; gates (type, in0, in1).  Returns output layer first, input last"""
    out = [list(inp)]
    cur = list(inp)
    for layer in reversed(layers):
        cur = [(cur[a] + cur[b]) % R if g == "add" else cur[a] * cur[b] % R for g, a, b in layer]
        out.append(cur)
    out.reverse()
    return out


def mle_size(layer_index):                        # size_of_mle_n_var_at_each_layer
    return 8 if layer_index == 0 else 1 << (layer_index + 2 * (layer_index + 1))


def label(layer_index, a, b, c):                  # transform_label_to_binary_and_to_decimal
    def bits(v, n):
        n = n or 1
        s = format(v, "b")
        return "0" * max(0, n - len(s)) + s
    return int(bits(a, layer_index) + bits(b, layer_index + 1) + bits(c, layer_index + 1), 2)


def add_mult_mle(layers, layer_index):
    add, mul = [0] * mle_size(layer_index), [0] * mle_size(layer_index)
    for gi, (g, a, b) in enumerate(layers[layer_index]):
        (add if g == "add" else mul)[label(layer_index, gi, a, b)] = 1
    return add, mul


def gkr_prove(layers, evaluation):
    """GKRProtocol::prove -> dict(w0, layers = [dict(sum, rps, challenges, wb, wc)])"""
    t = Transcript()
    w0 = list(evaluation[0]) + [0]
    n_vars(len(w0))
    t.commit(table_bytes(w0))
    n_r = t.n_fields(n_vars(len(w0)))
    claimed = evaluate(w0, n_r)
    out = []
    r_b = r_c = None
    alpha = beta = None
    for li in range(1, len(evaluation)):
        add, mul = add_mult_mle(layers, li - 1)
        w = list(evaluation[li])
        if li == 1:                               # generate_layer_one_prove_sumcheck (gkr/src/utils.rs:12-56)
            a_t = folds(add, n_r, [0] * len(n_r))
            m_t = folds(mul, n_r, [0] * len(n_r))
        else:                                     # protocol.rs:61-80
            zeros = [0] * len(r_b)
            a_b, m_b = folds(add, r_b, zeros), folds(mul, r_b, zeros)
            a_c, m_c = folds(add, r_c, zeros), folds(mul, r_c, zeros)
            a_t = [(x * alpha + y * beta) % R for x, y in zip(a_b, a_c)]
            m_t = [(x * alpha + y * beta) % R for x, y in zip(m_b, m_c)]
        terms = [[a_t, add_distinct(w, w)], [m_t, mul_distinct(w, w)]]
        rps, chs = multi_composed_prove(terms, claimed, True)
        t.commit(proof_bytes(rps))
        b, c = chs[:len(chs) // 2], chs[len(chs) // 2:]
        wb, wc = evaluate(w, b), evaluate(w, c)
        out.append(dict(sum=claimed, rps=rps, challenges=chs, wb=wb, wc=wc))
        r_b, r_c = b, c
        alpha, beta = t.field(), t.field()
        claimed = (alpha * wb + beta * wc) % R
    return dict(w0=w0, layers=out)


def gkr_verify(layers, inp, proof):
    """GKRProtocol::verify (protocol.rs:118-195)"""
    t = Transcript()
    t.commit(table_bytes(proof["w0"]))
    n_r = t.n_fields(n_vars(len(proof["w0"])))
    claimed = evaluate(proof["w0"], n_r)
    r_b = r_c = []
    alpha = beta = 0
    for i, lp in enumerate(proof["layers"]):
        if claimed != lp["sum"]:
            return False
        t.commit(proof_bytes(lp["rps"]))
        sub = multi_composed_verify_partial(lp["sum"], lp["rps"])
        if sub is None:
            return False
        claim, chs = sub
        if i == 0:
            add, mul = add_mult_mle(layers, 0)
            rbc = n_r + chs
            if (evaluate(add, rbc) * (lp["wb"] + lp["wc"]) + evaluate(mul, rbc) * lp["wb"] * lp["wc"]) % R != claim:
                return False
        r_b, r_c = chs[:len(chs) // 2], chs[len(chs) // 2:]
        alpha, beta = t.field(), t.field()
        claimed = (alpha * lp["wb"] + beta * lp["wc"]) % R
    return claimed == (alpha * evaluate(list(inp), r_b) + beta * evaluate(list(inp), r_c)) % R


# ---------------------------------------------------------------------------------------------------------------
# G1 (affine, python ints) and KZG: kzg/src/{trusted_setup,univariate_kzg,multilinear_kzg,utils}.rs
def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, P - 2, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, P - 2, P) % P
    x3 = (lam * lam - x1 - x2) % P
    return x3, (lam * (x1 - x3) - y1) % P


def g1_mul(pt, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, pt)
        pt = g1_add(pt, pt)
        k >>= 1
    return acc


def on_curve(pt):
    return pt is None or (pt[1] * pt[1] - pt[0] ** 3 - 4) % P == 0


def hypercube(n):                                 # boolean_hypercube (polynomial/src/utils.rs:141-157): MSB first
    return [[(i >> j) & 1 for j in range(n - 1, -1, -1)] for i in range(1 << n)]


def eq_points(tau):                               # generate_array_of_points / check_for_zero_and_one (kzg/src/utils.rs:19-40)
    out = []
    for bh in hypercube(len(tau)):
        acc = 1
        for b, e in zip(bh, tau):
            acc = acc * (e if b else 1 - e) % R
        out.append(acc)
    return out


def multilinear_srs(tau):                         # trusted_setup.rs:25-35
    return [g1_mul(G1, v) for v in eq_points(tau)]


def univariate_srs(tau, max_degree):              # univariate_kzg.rs:18-35
    return [g1_mul(G1, pow(tau, i, R)) for i in range(max_degree + 1)]


def commit(scalars, srs, require_equal_len):
    """MultilinearKZG::commitment (multilinear_kzg.rs:33-48, lengths asserted) / UnivariateKZG::commitment
    (univariate_kzg.rs:37-58, SRS may be longer)"""
    if require_equal_len:
        assert len(scalars) == len(srs)
    acc = None
    for s, pt in zip(scalars, srs):
        acc = g1_add(acc, g1_mul(pt, s))
    return acc


def kzg_open(ev, z, srs):
    """MultilinearKZG::open (multilinear_kzg.rs:50-88)"""
    evaluation = evaluate(ev, z)
    proofs, poly, final = [], list(ev), 0
    for vi, pt in enumerate(z):
        f1, f0 = fold(poly, 1, 0), fold(poly, 0, 0)               # get_poly_quotient (kzg/src/utils.rs:12-17)
        q = [(a - b) % R for a, b in zip(f1, f0)]
        if vi != len(z) - 1:
            rem = fold(poly, pt, 0)
            blown = add_to_front(q, vi)
        else:
            final = evaluate(poly, [pt])
            rem = [0] * (1 << vi)                                 # additive_identity(variable_index); never used again
            blown = add_to_front(q + q, vi - 1)                   # duplicate_evaluation, then add_to_front(vi - 1)
        proofs.append(commit(blown, srs, True))
        poly = rem
    assert evaluation == final
    return evaluation, proofs


def dense_degree(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return max(0, len(c) - 1)


def dense_divide(num, den):
    """divide_with_q_and_r (dense_univariate.rs:88-124) -> quotient coefficients"""
    if not num:
        return []
    if dense_degree(num) < dense_degree(den):
        return []
    q = [0] * (dense_degree(num) - dense_degree(den) + 1)
    rem = list(num)
    li = inv(den[-1])
    while rem and dense_degree(rem) >= dense_degree(den):
        c = rem[-1] * li % R
        d = dense_degree(rem) - dense_degree(den)
        q[d] = c
        for i, dc in enumerate(den):
            rem[d + i] = (rem[d + i] - c * dc) % R
        while rem and rem[-1] == 0:
            rem.pop()
    return q


def univariate_open(coeffs, z, srs):
    """UnivariateKZG::open (univariate_kzg.rs:60-81); the numerator is `poly - z` (:70), as the reference writes it"""
    evaluation = sum(c * pow(z, i, R) for i, c in enumerate(coeffs)) % R
    num = list(coeffs)
    num[0] = (num[0] - z) % R
    q = dense_divide(num, [(-z) % R, 1])
    return evaluation, commit(q, srs, False)


# ---------------------------------------------------------------------------------------------------------------
# NTT: polynomial/src/utils.rs:281-324, polynomial/src/univariate/domain.rs:31-48,108-133, evaluation.rs:59-86
def root_of_unity(size):
    log = (size - 1).bit_length() if size > 1 else 0
    w = TWO_ADIC_ROOT
    for _ in range(log, TWO_ADICITY):
        w = w * w % R
    return w


def bitreverse(n, l):
    r = 0
    for _ in range(l):
        r = (r << 1) | (n & 1)
        n >>= 1
    return r


def serial_fft(lst, w, size_log):
    n = len(lst)
    assert n == 1 << size_log
    for k in range(n):
        rk = bitreverse(k, size_log)
        if k < rk:
            lst[k], lst[rk] = lst[rk], lst[k]
    m = 1
    for _ in range(size_log):
        w_m = pow(w, n // (2 * m), R)
        k = 0
        while k < n:
            ww = 1
            for j in range(m):
                t = lst[k + j + m] * ww % R
                lst[k + j + m] = (lst[k + j] - t) % R
                lst[k + j] = (lst[k + j] + t) % R
                ww = ww * w_m % R
            k += 2 * m
        m *= 2


def domain_size(n):
    return 1 << ((n - 1).bit_length() if n > 1 else 0)


def domain_fft(coeffs, num_of_coeffs):
    size = domain_size(num_of_coeffs)
    v = [c % R for c in coeffs] + [0] * (size - len(coeffs))
    serial_fft(v, root_of_unity(size), size.bit_length() - 1)
    return v


def domain_ifft(evals, num_of_coeffs):
    size = domain_size(num_of_coeffs)
    v = [c % R for c in evals] + [0] * (size - len(evals))
    serial_fft(v, inv(root_of_unity(size)), size.bit_length() - 1)
    ni = inv(size)
    return [x * ni % R for x in v]


def univariate_multiply(a, b):                    # UnivariateEval::multiply (evaluation.rs:59-86)
    ln = len(a) + len(b) - 1
    size = domain_size(ln)
    fa, fb = domain_fft(a, size), domain_fft(b, size)
    return domain_ifft([x * y % R for x, y in zip(fa, fb)], size)[:ln]


# ---------------------------------------------------------------------------------------------------------------
def self_check():
    """Value KATs the reference's own tests hold (file:line) -- the model must reproduce each before it may generate anything"""
    F = lambda vs: [v % R for v in vs]
    # evaluation_form.rs:315-325, :328-359, :362-387, :390-405, :408-438, :441-462
    assert fold(F([3, 1, 2, 5]), 5, 0) == F([-2, 21])
    ev = F([3, 9, 7, 13, 6, 12, 10, 18])
    assert [evaluate(fold(ev, 2, 0), [3, 2]), evaluate(fold(ev, 3, 1), [3, 2]), evaluate(fold(ev, 1, 2), [3, 2])] == [57, 72, 38]
    assert evaluate(F([3, 1, 2, 5]), [5, 6]) == 136 and evaluate(ev, [2, 3, 1]) == 39
    assert evaluate(F([0, 0, 0, 3, 0, 0, 2, 5]), [2, 3, 4]) == 48
    assert half_sums(F([0, 0, 0, 2, 2, 2, 2, 4])) == [2, 10] and half_sums(F([0, 0, 2, 7, 3, 3, 6, 11])) == [9, 23]
    assert sum(range(1, 9)) % R == 36
    # pick_pairs: variable k pairs (i, i + n >> (k+1)) for every i with that bit clear
    assert pairs(8, 0) == [(0, 4), (1, 5), (2, 6), (3, 7)] and pairs(8, 1) == [(0, 2), (1, 3), (4, 6), (5, 7)]
    assert pairs(8, 2) == [(0, 1), (2, 3), (4, 5), (6, 7)]
    # sumcheck/src/utils.rs:70-93: Fr(1) -> 31 x 0x00, 0x01
    assert be32(1) == bytes(31) + b"\x01"
    # sumcheck.rs:108-122 (12); composed_sumcheck.rs:108-140 (3, 5, 6, 12); multi_composed_sumcheck.rs:195-214 (7, 8)
    assert product_sums([F([0, 0, 0, 2, 2, 2, 2, 4])]) == 12 and product_sums([F([0, 1, 2, 3])]) == 6
    assert product_sums([F([0, 1, 2, 3]), F([0, 0, 0, 1])]) == 3 and product_sums([F([3, 3, 5, 5]), F([0, 0, 0, 1])]) == 5
    assert multi_composed_sum([[F([0, 1, 2, 3])], [F([0, 0, 0, 1])]]) == 7
    assert multi_composed_sum([[F([0, 0, 0, 2])], [F([0, 3, 0, 3])]]) == 8
    # sparse_univariate.rs:232-247 (265), :250-299 (Add), :361-384, :387-448 (interpolation)
    assert sparse_eval([(5, 0), (2, 1), (4, 6)], 2) == 265
    assert sparse_add([(5, 0)], [(2, 1)]) == [(5, 0), (2, 1)]
    assert sparse_add([(5, 0), (5, 2)], [(2, 1), (2, 2)]) == [(5, 0), (2, 1), (7, 2)]
    assert interpolation([2, 3, 11], [1, 2, 4]) == [(3, 0), (R - 2, 1), (1, 2)]
    assert interpolation([6, 11, 18, 27, 38], [1, 2, 3, 4, 5]) == [(3, 0), (2, 1), (1, 2)]     # zero x^3, x^4 dropped
    assert interpolation([0, 2]) == [(2, 1)] and sparse_eval(interpolation([5, 7, 13]), 2) == 13
    assert sparse_eval(interpolation([12, 48, 3150, 11772, 33452, 315020], [0, 1, 3, 4, 5, 8]), 1) == 48
    # kzg/src/utils.rs:73-103: eq points of (2, 3, 4)
    assert eq_points([2, 3, 4]) == F([-6, 8, 9, -12, 12, -16, -18, 24])
    # domain.rs:154-168
    assert root_of_unity(16) == 14788168760825820622209131888203028446852016562542525606630160374691593895118
    assert inv(root_of_unity(16)) == 26753076894533791554649012143113393549300550745003194222677083919072199473480
    # circuit.rs:139-166, :209-260; protocol.rs:280 (224)
    assert circuit_evaluation([[("mul", 0, 1)], [("add", 0, 1), ("mul", 2, 3)]], [2, 3, 4, 5]) == [[100], [5, 20], [2, 3, 4, 5]]
    assert [mle_size(i) for i in range(4)] == [8, 32, 256, 2048]
    # G1: generator on the curve, r G = identity; multilinear_kzg.rs:133-148 data: commit == p(tau) G with p(2,3,4) = 28
    assert on_curve(G1) and g1_mul(G1, R - 1) == (G1[0], P - G1[1])
    vals = F([0, 7, 0, 5, 0, 7, 4, 9])
    assert evaluate(vals, [2, 3, 4]) == 28 and commit(vals, multilinear_srs([2, 3, 4]), True) == g1_mul(G1, 28)
    # NTT: inverse of forward is the identity; product equals schoolbook (dense_univariate.rs:464-497 style)
    v = list(range(1, 17))
    assert domain_ifft(domain_fft(v, 16), 16) == v
    a, b = [1, 2, 3], [4, 5]
    assert univariate_multiply(a, b) == [4, 13, 22, 15]
    # prove -> verify round trips of the reference's own tests (protocol.rs:209-232)
    layers = [[("mul", 0, 1)], [("add", 0, 1), ("mul", 2, 3)]]
    assert gkr_verify(layers, [2, 3, 4, 5], gkr_prove(layers, circuit_evaluation(layers, [2, 3, 4, 5])))
    return True


if __name__ == "__main__":
    print("model self-check:", self_check())
