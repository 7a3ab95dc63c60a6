// test_mirror.cpp -- the reference's own unit tests (same names, same inputs, same expected values) run against the
// C++ host mirror (include/zkhip.hpp) on the GPU, plus bit-exact comparisons with the CPU oracle on random inputs.
// Built and run by tests/test_cpp_mirror.py.  Exit code 0 = all passed.
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <mutex>
#include <random>
#include <thread>

#include "../../include/zkhip.hpp"
extern "C" {
#include "../../oracle/zkoracle.h"
}

using namespace zkc;
static int g_failed = 0, g_run = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { std::printf("  FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond); ++g_failed; } \
    } while (0)
#define TEST(name) static void name(); static struct name##_reg { name##_reg() { tests().push_back({#name, name}); } } name##_inst; static void name()
struct T { const char* n; void (*f)(); };
static std::vector<T>& tests() { static std::vector<T> v; return v; }

static std::vector<Fr> F(std::initializer_list<long> v) { std::vector<Fr> o; for (long x : v) o.push_back(Fr::from(x)); return o; }
static std::vector<Fr> random_fr(size_t n, uint64_t seed) {
    std::mt19937_64 g(seed);
    std::vector<Fr> v(n);
    for (auto& e : v) { for (int i = 0; i < 4; ++i) e.l[i] = g(); e.l[3] &= 0x3FFFFFFFFFFFFFFFULL; }
    return v;
}
static const fr_t* O(const std::vector<Fr>& v) { return reinterpret_cast<const fr_t*>(v.data()); }
static fr_t* O(std::vector<Fr>& v) { return reinterpret_cast<fr_t*>(v.data()); }
template <class Fn> static bool panics(Fn f) { try { f(); } catch (const std::exception&) { return true; } return false; }

// ---- polynomial/src/multilinear/evaluation_form.rs `mod tests` ------------------------------------------------
TEST(test_add_mul_distinct) {
    Multilinear p1(F({0, 0, 2, 2})), p2(F({0, 3, 0, 3}));
    EXPECT(p1.add_distinct(p2) == Multilinear(F({0, 3, 0, 3, 0, 3, 0, 3, 2, 5, 2, 5, 2, 5, 2, 5})));
    EXPECT(p1.mul_distinct(p2) == Multilinear(F({0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 0, 6, 0, 6, 0, 6})));
}
TEST(test_partial_evaluation_1) {
    Multilinear poly(F({3, 1, 2, 5}));
    EXPECT(poly.partial_evaluation(Fr::from(5), 0) == Multilinear(F({-2, 21})));
}
TEST(test_partial_evaluation_2) {
    Multilinear poly(F({3, 9, 7, 13, 6, 12, 10, 18}));
    EXPECT(poly.partial_evaluation(Fr::from(2), 0).evaluation(F({3, 2})) == Fr::from(57));
    EXPECT(poly.partial_evaluation(Fr::from(3), 1).evaluation(F({3, 2})) == Fr::from(72));
    EXPECT(poly.partial_evaluation(Fr::from(1), 2).evaluation(F({3, 2})) == Fr::from(38));
}
TEST(test_evaluation_1) {
    EXPECT(Multilinear(F({3, 1, 2, 5})).evaluation(F({5, 6})) == Fr::from(136));
    EXPECT(Multilinear(F({3, 1, 2, 5})).evaluation(F({5, 6})) != Fr::from(3));
    EXPECT(Multilinear(F({3, 9, 7, 13, 6, 12, 10, 18})).evaluation(F({2, 3, 1})) == Fr::from(39));
}
TEST(test_evaluation_2) { EXPECT(Multilinear(F({0, 0, 0, 3, 0, 0, 2, 5})).evaluation(F({2, 3, 4})) == Fr::from(48)); }
TEST(test_split_poly_into_two_and_sum_each_part) {
    EXPECT(Multilinear(F({0, 0, 0, 2, 2, 2, 2, 4})).split_poly_into_two_and_sum_each_part() == Multilinear(F({2, 10})));
    EXPECT(Multilinear(F({0, 0, 2, 7, 3, 3, 6, 11})).split_poly_into_two_and_sum_each_part() == Multilinear(F({9, 23})));
}
TEST(test_sum_over_boolean_hypercube) { EXPECT(Multilinear(F({1, 2, 3, 4, 5, 6, 7, 8})).sum_over_the_boolean_hypercube() == Fr::from(36)); }
TEST(test_add_to_front_and_back) {   // evaluation_form.rs:465-507, :548-556
    EXPECT(Multilinear(F({0, 0, 4, 4})).add_to_front(0) == Multilinear(F({0, 0, 4, 4, 0, 0, 4, 4})));
    EXPECT(Multilinear(F({0, 4})).add_to_front(1) == Multilinear(F({0, 4, 0, 4, 0, 4, 0, 4})));
    EXPECT(Multilinear(F({0, 0, 4, 4})).add_to_back(1) == Multilinear(F({0, 0, 0, 0, 4, 4, 4, 4})));
    EXPECT(Multilinear::duplicate_evaluation(F({11})) == Multilinear(F({11, 11})));
}
TEST(test_dense_polynomial_evaluation) {   // dense_univariate.rs:425-433, :199-207
    EXPECT(DenseUnivariatePolynomial(F({5, 2, 4})).evaluate(Fr::from(2)) == Fr::from(25));
    EXPECT(DenseUnivariatePolynomial(F({1, 3, 0, 0})).degree() == 1 && DenseUnivariatePolynomial(F({0, 0})).degree() == 0);
}
TEST(test_poly_subtraction) {
    EXPECT(Multilinear(F({0, 0, 0, 5, 4, 4, 7, 12})) - Multilinear(F({0, 0, 0, 2, 0, 0, 1, 3})) == Multilinear(F({0, 0, 0, 3, 4, 4, 6, 9})));
}
TEST(test_panics_like_the_reference) {
    EXPECT(panics([] { Multilinear m(F({1, 2, 3})); }));                                        // evaluation_form.rs:16-20
    EXPECT(panics([] { Multilinear(F({1, 2, 3, 4})).evaluation(F({5})); }));                      // :163-167
    EXPECT(panics([] { Multilinear(F({1, 2, 3, 4})).partial_evaluation(Fr::from(5), 2); }));      // utils.rs:31-34
    EXPECT(panics([] { Multilinear(F({1, 2, 3, 4})).partial_evaluations(F({5, 6}), {0}); }));     // :146-152
}
// ---- sumcheck/src/utils.rs:70-93 ------------------------------------------------------------------------------------
TEST(test_convert_field_to_byte) {
    auto one = Fr::from(1).to_bytes_be(), hundred = Fr::from(100).to_bytes_be();
    std::vector<uint8_t> e1(32, 0), e100(32, 0);
    e1[31] = 1; e100[31] = 100;
    EXPECT(one == e1 && hundred == e100);
    EXPECT(Multilinear(F({1, 100})).to_bytes() == [&] { auto v = e1; v.insert(v.end(), e100.begin(), e100.end()); return v; }());
}
// ---- sumcheck/src/sumcheck.rs `mod tests` ---------------------------------------------------------------------------
TEST(test_sum_calculation) {
    Sumcheck prover(Multilinear(F({0, 0, 0, 2, 2, 2, 2, 4})));
    prover.poly_sum();
    EXPECT(prover.sum() == Fr::from(12));
}
static void sum_check_proof(const std::vector<Fr>& evals) {
    Sumcheck sc{Multilinear(evals)};
    sc.poly_sum();
    auto [proof, challenges] = sc.prove();
    const size_t nv = proof.poly.n_vars;
    std::vector<Fr> rp(2 * nv), och(nv), gpu_rp;
    Fr osum;
    ora_sumcheck_prove(O(evals), evals.size(), (fr_t*)&osum, O(rp), O(och));
    for (auto& u : proof.univariate_poly) for (auto& e : u.evaluations()) gpu_rp.push_back(e);
    EXPECT(proof.sum == osum && gpu_rp == rp && challenges == och);
    EXPECT(ora_sumcheck_verify(O(evals), evals.size(), (const fr_t*)&proof.sum, O(gpu_rp)) == 1);   // `verify == true`
}
TEST(test_sum_check_proof) { sum_check_proof(F({0, 0, 2, 7, 3, 3, 6, 11})); }
TEST(test_sum_check_proof_2) { sum_check_proof(F({0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0})); }
TEST(test_sum_check_proof_3) { sum_check_proof(F({1, 3, 5, 7, 2, 4, 6, 8, 3, 5, 7, 9, 4, 6, 8, 10})); }
TEST(test_sum_check_proof_random_2_16) { sum_check_proof(random_fr(1 << 16, 7)); }
// ---- composed ----------------------------------------------------------------------------------------------------------
TEST(test_composed_sum_calculation) {
    EXPECT(ComposedSumcheck::calculate_poly_sum(ComposedMultilinear({Multilinear(F({0, 1, 2, 3})), Multilinear(F({0, 0, 0, 1}))})) == Fr::from(3));
    EXPECT(ComposedSumcheck::calculate_poly_sum(ComposedMultilinear({Multilinear(F({3, 3, 5, 5})), Multilinear(F({0, 0, 0, 1}))})) == Fr::from(5));
    EXPECT(ComposedSumcheck::calculate_poly_sum(ComposedMultilinear({Multilinear(F({0, 1, 2, 3}))})) == Fr::from(6));
    EXPECT(ComposedMultilinear({Multilinear(F({0, 1, 2, 3})), Multilinear(F({0, 0, 0, 1}))}).evaluation(F({2, 3})) == Fr::from(42));
}
TEST(test_element_wise_product) {   // composed_multilinear.rs:159-170
    ComposedMultilinear polys({Multilinear(F({0, 1, 2, 3})), Multilinear(F({0, 0, 0, 1}))});
    EXPECT(polys.element_wise_product() == F({0, 0, 0, 3}));
}
TEST(test_element_wise_add) {       // composed_multilinear.rs:172-184
    ComposedMultilinear polys({Multilinear(F({0, 1, 2, 3})), Multilinear(F({0, 0, 0, 1}))});
    EXPECT(polys.element_wise_add() == F({0, 1, 2, 4}));
}
TEST(test_composed_sum_check_proof) {
    std::vector<std::vector<Fr>> tabs = {F({3, 3, 5, 5}), F({0, 0, 0, 1})};
    ComposedSumcheck sc(ComposedMultilinear({Multilinear(tabs[0]), Multilinear(tabs[1])}));
    auto [proof, ch] = sc.prove();
    std::vector<Fr> flat = tabs[0]; flat.insert(flat.end(), tabs[1].begin(), tabs[1].end());
    std::vector<Fr> rp(3 * 2), och(2), grp;
    ora_composed_prove(O(flat), 2, 4, O(rp), O(och));
    for (auto& r : proof.round_polys) grp.insert(grp.end(), r.begin(), r.end());
    EXPECT(grp == rp && ch == och);
    Fr s = ComposedSumcheck::calculate_poly_sum(sc.poly);
    EXPECT(ora_composed_verify(O(flat), 2, 4, (const fr_t*)&s, O(grp)) == 1);
}
TEST(test_multi_composed_sum_check_proof_2_on_gkr_example) {   // multi_composed_sumcheck.rs:266-311
    Multilinear add_i(F({4, 4, 7, 7, 4, 4, 7, 9})), mul_i(F({3, 3, 3, 4, 3, 3, 5, 6})), w_b(F({0, 4})), w_c(F({0, 3}));
    ComposedMultilinear lhs({add_i.partial_evaluation(Fr::from(2), 0), w_b.add_distinct(w_c)});
    ComposedMultilinear rhs({mul_i.partial_evaluation(Fr::from(2), 0), w_b.mul_distinct(w_c)});
    std::vector<ComposedMultilinear> multi = {lhs, rhs};
    Fr sum = MultiComposedSumcheckProver::calculate_poly_sum(multi);
    auto [proof, ch] = MultiComposedSumcheckProver::prove(multi, sum);
    std::vector<Fr> flat;
    for (auto& t : multi) for (auto& p : t.polys) { auto e = p.evaluations(); flat.insert(flat.end(), e.begin(), e.end()); }
    size_t sizes[2] = {2, 2};
    std::vector<ora_sparse_t> orp(2);
    std::vector<Fr> och(2);
    Fr osum;
    ora_multi_composed_sum((fr_t*)&osum, O(flat), sizes, 2, 4);
    EXPECT(osum == sum);
    ora_multi_composed_prove(O(flat), sizes, 2, 4, (const fr_t*)&sum, 0, orp.data(), O(och));
    std::vector<uint8_t> obytes;
    for (auto& r : orp) { uint8_t b[64 * ORA_SPARSE_MAX]; size_t n = ora_sparse_to_bytes(b, &r); obytes.insert(obytes.end(), b, b + n); }
    EXPECT(proof.to_bytes() == obytes && ch == och);
    EXPECT(ora_multi_composed_verify(O(flat), sizes, 2, 4, (const fr_t*)&sum, orp.data(), 2) == 1);
    auto [pp, chp] = MultiComposedSumcheckProver::prove_partial(multi, sum);
    EXPECT(chp != ch);
}
// ---- kzg ---------------------------------------------------------------------------------------------------------------
static bool same_point(const G1Affine& g, const g1_jac_t& want) {
    g1_affine_t a;
    ora_g1_to_affine(&a, &want);
    return (g.infinity == (a.inf != 0)) && (g.infinity || std::memcmp(g.xy, &a, 96) == 0);
}
TEST(test_kzg_1) {   // multilinear_kzg.rs:133-148 (commitment half)
    auto vals = F({0, 7, 0, 5, 0, 7, 4, 9}), tau = F({2, 3, 4});
    TrustedSetup srs = TrustedSetup::setup(tau);
    G1Affine commit = MultilinearKZG::commitment(Multilinear(vals), srs);
    std::vector<g1_jac_t> osrs(8);
    ora_kzg_multilinear_srs_g1(osrs.data(), O(tau), 3);
    g1_jac_t want;
    EXPECT(ora_kzg_commitment(&want, O(vals), 8, osrs.data(), 8, 1) == 0);
    EXPECT(same_point(commit, want));
    EXPECT(panics([&] { MultilinearKZG::commitment(Multilinear(F({1, 2, 3, 4})), srs); }));     // assert_eq! :36-41
}
static void kzg_open_case(const std::vector<Fr>& vals, const std::vector<Fr>& tau, const std::vector<Fr>& z) {
    TrustedSetup srs = TrustedSetup::setup(tau);
    MultilinearKZGProof proof = MultilinearKZG::open(Multilinear(vals), z, srs);
    std::vector<g1_jac_t> osrs(vals.size()), want(tau.size());
    ora_kzg_multilinear_srs_g1(osrs.data(), O(tau), tau.size());
    Fr ev;
    EXPECT(ora_kzg_open((fr_t*)&ev, want.data(), O(vals), vals.size(), O(z), z.size(), osrs.data(), osrs.size()) == 0);
    EXPECT(proof.evaluation == ev && proof.proofs.size() == tau.size());
    for (size_t i = 0; i < want.size() && i < proof.proofs.size(); ++i) EXPECT(same_point(proof.proofs[i], want[i]));
    // the same opening against the level tables (TrustedSetup::precompute_open)
    srs.precompute_open();
    EXPECT(srs.level_tables() != nullptr);
    MultilinearKZGProof tabled = MultilinearKZG::open(Multilinear(vals), z, srs);
    EXPECT(tabled.evaluation == ev && tabled.proofs.size() == tau.size());
    for (size_t i = 0; i < want.size() && i < tabled.proofs.size(); ++i) EXPECT(same_point(tabled.proofs[i], want[i]));
}
TEST(test_kzg_1_open) {   // multilinear_kzg.rs:131-155 (open half; the pairing verifier is out of scope)
    kzg_open_case(F({0, 7, 0, 5, 0, 7, 4, 9}), F({2, 3, 4}), F({5, 9, 6}));
    EXPECT(MultilinearKZG::open(Multilinear(F({0, 7, 0, 5, 0, 7, 4, 9})), F({5, 9, 6}), TrustedSetup::setup(F({2, 3, 4}))).evaluation == Fr::from(114));
    EXPECT(panics([&] { MultilinearKZG::open(Multilinear(F({0, 7, 0, 5, 0, 7, 4, 9})), F({5, 9}), TrustedSetup::setup(F({2, 3, 4}))); }));
}
TEST(test_kzg_2_open) {   // multilinear_kzg.rs:157-197
    kzg_open_case(F({0, 0, 0, 2, 0, 0, 10, 12, 0, -12, 4, -6, 0, -12, 14, 4}), F({12, 9, 28, 40}), F({54, 90, 76, 160}));
}
TEST(test_univariate_kzg) {   // univariate_kzg.rs:111-129 (commitment half)
    auto coeffs = F({1, 2, 3, 4, 5});
    TrustedSetup srs = UnivariateKZG::generate_srs(Fr::from(10), 4);
    G1Affine commit = UnivariateKZG::commitment(DenseUnivariatePolynomial(coeffs), srs);
    std::vector<g1_jac_t> osrs(5);
    Fr tau = Fr::from(10);
    ora_kzg_univariate_srs_g1(osrs.data(), (const fr_t*)&tau, 4);
    g1_jac_t want;
    ora_kzg_commitment(&want, O(coeffs), 5, osrs.data(), 5, 0);
    EXPECT(same_point(commit, want));
    EXPECT(panics([&] { UnivariateKZG::commitment(DenseUnivariatePolynomial(F({1, 2, 3, 4, 5, 6})), srs); }));   // index out of bounds :53
}
TEST(test_univariate_kzg_open) {   // univariate_kzg.rs:111-129 (open half)
    auto coeffs = F({1, 2, 3, 4, 5});
    Fr tau = Fr::from(10), z = Fr::from(2);
    TrustedSetup srs = UnivariateKZG::generate_srs(tau, 4);
    UnivariateKZGProof proof = UnivariateKZG::open(DenseUnivariatePolynomial(coeffs), z, srs);
    std::vector<g1_jac_t> osrs(5);
    ora_kzg_univariate_srs_g1(osrs.data(), (const fr_t*)&tau, 4);
    Fr ev; g1_jac_t want;
    EXPECT(ora_univariate_kzg_open((fr_t*)&ev, &want, O(coeffs), 5, (const fr_t*)&z, osrs.data(), 5) == 0);
    EXPECT(proof.evaluation == ev && proof.evaluation == Fr::from(129) && same_point(proof.proof, want));
    EXPECT(panics([&] { UnivariateKZG::open(DenseUnivariatePolynomial(F({1, 2, 3, 4, 5, 6, 7})), z, srs); }));
}
TEST(test_kzg_random_2_10) {
    auto tau = random_fr(10, 3), sc = random_fr(1 << 10, 4);
    G1Affine commit = MultilinearKZG::commitment(Multilinear(sc), TrustedSetup::setup(tau));
    Fr p_tau;
    ora_mle_evaluation((fr_t*)&p_tau, O(sc), sc.size(), O(tau), 10);
    uint64_t canon[4];
    ora_fr_to_canonical(canon, (const fr_t*)&p_tau);
    g1_jac_t g, want;
    ora_g1_generator(&g);
    ora_g1_mul_bigint(&want, &g, canon, 4);          // commit == p(tau) * G
    EXPECT(same_point(commit, want));
    TrustedSetup with_table = TrustedSetup::setup(tau);
    EXPECT(same_point(MultilinearKZG::commitment(Multilinear(sc), with_table.precompute()), want));   // shifted-SRS table path
}
// ---- circuit / GKR ------------------------------------------------------------------------------------------------------
static Circuit make_circuit(const std::vector<std::vector<std::array<int, 3>>>& layers) {   // {type (0 add, 1 mul), in0, in1}
    std::vector<CircuitLayer> ls;
    for (auto& l : layers) {
        CircuitLayer cl;
        for (auto& g : l) cl.layer.push_back(Gate{g[0] ? GateType::Mul : GateType::Add, {(size_t)g[1], (size_t)g[2]}});
        ls.push_back(cl);
    }
    return Circuit(ls);
}
static void gkr_case(const Circuit& circuit, const std::vector<Fr>& input, int64_t expected_output) {
    auto ev = circuit.evaluation(input);
    EXPECT(ev.layer(0)[0] == Fr::from(expected_output));
    GKRProof proof = GKRProtocol::prove(circuit, ev);
    // the oracle's restatement of the same circuit: identical proof, accepted by the restated verifier
    std::vector<size_t> n_gates; std::vector<uint8_t> gt; std::vector<uint32_t> i0, i1;
    for (auto& l : circuit.layers) { n_gates.push_back(l.layer.size()); Circuit::arrays(l, gt, i0, i1); }
    size_t total = 0;
    for (auto n : ev.lens) total += n;
    std::vector<Fr> flat(total);
    std::vector<size_t> lens(ev.lens.size());
    EXPECT(ora_circuit_evaluation(circuit.layers.size(), n_gates.data(), gt.data(), i0.data(), i1.data(), O(input), input.size(), (fr_t*)flat.data(), lens.data()) == 0);
    auto want = std::make_unique<ora_gkr_proof_t>();
    EXPECT(ora_gkr_prove(circuit.layers.size(), n_gates.data(), gt.data(), i0.data(), i1.data(), (const fr_t*)flat.data(), lens.data(), want.get()) == 0);
    EXPECT(want->n_proofs == proof.sumcheck_proofs.size());
    auto got = std::make_unique<ora_gkr_proof_t>();
    std::memset(got.get(), 0, sizeof(ora_gkr_proof_t));
    got->n_proofs = proof.sumcheck_proofs.size();
    for (size_t k = 0; k < proof.sumcheck_proofs.size(); ++k) {
        auto& sp = proof.sumcheck_proofs[k];
        std::memcpy(&got->sums[k], sp.sum.l, 32);
        got->n_rounds[k] = sp.round_polys.size();
        for (size_t r = 0; r < sp.round_polys.size(); ++r) {
            got->round_polys[k][r].len = sp.round_polys[r].monomial.size();
            for (size_t m = 0; m < sp.round_polys[r].monomial.size(); ++m) {
                std::memcpy(&got->round_polys[k][r].coeff[m], sp.round_polys[r].monomial[m].coeff.l, 32);
                std::memcpy(&got->round_polys[k][r].pow[m], sp.round_polys[r].monomial[m].pow.l, 32);
            }
            EXPECT(want->round_polys[k][r].len == got->round_polys[k][r].len &&
                   std::memcmp(want->round_polys[k][r].coeff, got->round_polys[k][r].coeff, 32 * got->round_polys[k][r].len) == 0);
        }
        std::memcpy(&got->wb[k], proof.wb_s[k].l, 32);
        std::memcpy(&got->wc[k], proof.wc_s[k].l, 32);
        EXPECT(std::memcmp(&want->wb[k], &got->wb[k], 32) == 0 && std::memcmp(&want->wc[k], &got->wc[k], 32) == 0);
    }
    std::memcpy(got->w0, proof.w_0_mle[0].l, 64);
    EXPECT(ora_gkr_verify(circuit.layers.size(), n_gates.data(), gt.data(), i0.data(), i1.data(), O(input), input.size(), got.get()) == 1);
}
TEST(test_gkr_protocol_1) {   // gkr/src/protocol.rs:209-232
    gkr_case(make_circuit({{{1, 0, 1}}, {{0, 0, 1}, {1, 2, 3}}}), F({2, 3, 4, 5}), 100);
}
TEST(test_gkr_protocol_2) {   // gkr/src/protocol.rs:234-286
    gkr_case(make_circuit({{{0, 0, 1}},
                           {{1, 0, 1}, {0, 2, 3}},
                           {{0, 0, 1}, {1, 2, 3}, {1, 4, 5}, {1, 6, 7}},
                           {{1, 0, 1}, {1, 2, 3}, {1, 4, 5}, {0, 6, 7}, {1, 8, 9}, {0, 10, 11}, {1, 12, 13}, {1, 14, 15}}}),
             F({2, 1, 3, 1, 4, 1, 2, 2, 3, 3, 4, 4, 2, 3, 3, 4}), 224);
}
TEST(test_gkr_device_circuit_reused) {   // one resident circuit, two inputs: each proof equals the one-shot prover's
    Circuit c = make_circuit({{{0, 0, 1}}, {{1, 0, 1}, {0, 2, 3}}, {{0, 0, 1}, {1, 2, 3}, {1, 4, 5}, {1, 6, 7}}});
    DeviceCircuit dc(c);
    for (uint64_t seed : {11, 12}) {
        auto ev = c.evaluation(random_fr(8, seed));
        GKRProof a = dc.prove(ev), b = GKRProtocol::prove(c, ev);
        EXPECT(a.wb_s == b.wb_s && a.wc_s == b.wc_s && a.w_0_mle == b.w_0_mle && a.sumcheck_proofs.size() == b.sumcheck_proofs.size());
        for (size_t k = 0; k < a.sumcheck_proofs.size(); ++k) EXPECT(a.sumcheck_proofs[k].to_bytes() == b.sumcheck_proofs[k].to_bytes());
    }
}
TEST(test_gkr_prove_batch) {   // gkr/benches/gkr_benchmark.rs:11-27: many inputs, one circuit -- one call, every proof the single prover's
    Circuit c = make_circuit({{{0, 0, 1}}, {{1, 0, 1}, {0, 2, 3}}, {{0, 0, 1}, {1, 2, 3}, {1, 4, 5}, {1, 6, 7}}});
    DeviceCircuit dc(c);
    std::vector<Circuit::Evaluation> evs;
    for (uint64_t seed = 21; seed < 21 + 11; ++seed) evs.push_back(c.evaluation(random_fr(8, seed)));
    for (int rep = 0; rep < 2; ++rep) {             // the second batch replays the lanes' recorded launch chains
        std::vector<GKRProof> got = dc.prove_batch(evs, rep ? 3 : 0);
        EXPECT(got.size() == evs.size());
        for (size_t b = 0; b < got.size(); ++b) {
            GKRProof want = dc.prove(evs[b]);
            EXPECT(got[b].wb_s == want.wb_s && got[b].wc_s == want.wc_s && got[b].w_0_mle == want.w_0_mle && got[b].sumcheck_proofs.size() == want.sumcheck_proofs.size());
            for (size_t k = 0; k < want.sumcheck_proofs.size(); ++k) EXPECT(got[b].sumcheck_proofs[k].to_bytes() == want.sumcheck_proofs[k].to_bytes());
        }
    }
    EXPECT(dc.prove_batch({}).empty());
}
// ---- the sharded provers through the C ABI's one-call entry points (include/zkhip.h): ranks = host threads with a context each,
// the all-gather a barrier + a shared host buffer.  Every rank must return the single-GPU prover's proof of the WHOLE input.
namespace {
struct ThreadExchange {
    uint32_t world;
    std::vector<uint8_t> buf;
    std::mutex m;
    std::condition_variable cv;
    uint32_t waiting = 0;
    uint64_t generation = 0;
    explicit ThreadExchange(uint32_t w) : world(w), buf((size_t)w << 20) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(m);
        const uint64_t g = generation;
        if (++waiting == world) { waiting = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != g; });
    }
};
struct RankEnd { ThreadExchange* x; uint32_t rank; zkhip_ctx* c; };
// zkhip_all_gather_fn: completes before it returns (zkhip_memcpy_* wait for the context's stream), as the contract allows
int thread_all_gather(void* user, const void* d_send, void* d_recv, size_t bytes, void*) {
    RankEnd* e = (RankEnd*)user;
    if (bytes * e->x->world > e->x->buf.size()) return 1;
    if (zkhip_memcpy_d2h(e->c, e->x->buf.data() + (size_t)e->rank * bytes, d_send, bytes) != 0) return 1;
    e->x->barrier();
    if (zkhip_memcpy_h2d(e->c, d_recv, e->x->buf.data(), bytes * e->x->world) != 0) return 1;
    e->x->barrier();
    return 0;
}
}  // namespace
TEST(test_sharded_provers_with_threads_as_ranks) {
    const uint32_t world = 2;
    // whole inputs and the single-GPU provers' results, on the main thread's context
    auto full = random_fr(1 << 14, 4141);
    Sumcheck sc{Multilinear(full)};
    sc.poly_sum();
    auto want_sc = sc.prove();
    std::vector<std::vector<Fr>> tabs;
    for (int k = 0; k < 4; ++k) tabs.push_back(random_fr(1 << 12, 4150 + k));
    std::vector<ComposedMultilinear> poly;
    poly.push_back(ComposedMultilinear({Multilinear(tabs[0]), Multilinear(tabs[1])}));
    poly.push_back(ComposedMultilinear({Multilinear(tabs[2]), Multilinear(tabs[3])}));
    const Fr claimed = MultiComposedSumcheckProver::calculate_poly_sum(poly);
    auto want_mc = MultiComposedSumcheckProver::prove_partial(poly, claimed);
    Circuit circuit = make_circuit({{{0, 0, 1}},
                                    {{1, 0, 1}, {0, 2, 3}},
                                    {{0, 0, 1}, {1, 2, 3}, {1, 4, 5}, {1, 6, 7}},
                                    {{1, 0, 1}, {1, 2, 3}, {1, 4, 5}, {0, 6, 7}, {1, 8, 9}, {0, 10, 11}, {1, 12, 13}, {1, 14, 15}}});
    auto gkr_in = F({2, 1, 3, 1, 4, 1, 2, 2, 3, 3, 4, 4, 2, 3, 3, 4});
    GKRProof want_gkr = GKRProtocol::prove(circuit, circuit.evaluation(gkr_in));
    auto tau = random_fr(8, 4160), scal = random_fr(256, 4161);
    TrustedSetup srs = TrustedSetup::setup(tau);
    G1Affine want_c = MultilinearKZG::commitment(Multilinear(scal), srs);
    std::vector<uint64_t> srs_xy(12 * 256);
    std::vector<uint8_t> srs_inf(256);
    srs.pts_->download(srs_xy.data(), 96 * 256);
    srs.inf_->download(srs_inf.data(), 256);

    ThreadExchange x(world);
    std::vector<int> ok(world, 0);
    std::vector<uint64_t> n_ex(world, 0);
    auto body = [&](uint32_t rank) {
        try {
            RankEnd end{&x, rank, ctx()};                                   // this thread's own context
            Comm comm(rank, world, thread_all_gather, &end);
            bool good = true;
            auto got = ShardedSumcheck::prove(Multilinear(shard_interleaved(full, rank, world)), comm);
            good = good && got.first.sum == want_sc.first.sum && got.second == want_sc.second && got.first.univariate_poly.size() == want_sc.first.univariate_poly.size();
            for (size_t i = 0; good && i < got.first.univariate_poly.size(); ++i) good = got.first.univariate_poly[i] == want_sc.first.univariate_poly[i];
            std::vector<ComposedMultilinear> sh;
            sh.push_back(ComposedMultilinear({Multilinear(shard_interleaved(tabs[0], rank, world)), Multilinear(shard_interleaved(tabs[1], rank, world))}));
            sh.push_back(ComposedMultilinear({Multilinear(shard_interleaved(tabs[2], rank, world)), Multilinear(shard_interleaved(tabs[3], rank, world))}));
            auto mc = MultiComposedSumcheckProver::prove_partial_sharded(sh, claimed, comm);
            good = good && mc.first.to_bytes() == want_mc.first.to_bytes() && mc.second == want_mc.second;
            GKRProof g = GKRProtocol::prove_sharded(circuit, circuit.evaluation(gkr_in), comm);
            good = good && g.wb_s == want_gkr.wb_s && g.wc_s == want_gkr.wc_s && g.w_0_mle == want_gkr.w_0_mle && g.sumcheck_proofs.size() == want_gkr.sumcheck_proofs.size();
            for (size_t k = 0; good && k < g.sumcheck_proofs.size(); ++k) good = g.sumcheck_proofs[k].to_bytes() == want_gkr.sumcheck_proofs[k].to_bytes();
            TrustedSetup my_srs(256 / world);
            std::vector<uint64_t> my_xy;
            auto my_scal = shard_interleaved(scal, rank, world);
            std::vector<uint8_t> my_inf;
            for (size_t i = rank; i < 256; i += world) { my_xy.insert(my_xy.end(), srs_xy.begin() + 12 * i, srs_xy.begin() + 12 * (i + 1)); my_inf.push_back(srs_inf[i]); }
            my_srs.pts_->upload(my_xy.data(), 8 * my_xy.size());
            my_srs.inf_->upload(my_inf.data(), my_inf.size());
            G1Affine c = MultilinearKZG::commitment_sharded(Multilinear(my_scal), my_srs, comm);
            good = good && c == want_c;
            n_ex[rank] = comm.exchanges();
            ok[rank] = good ? 1 : 0;
        } catch (const std::exception& e) {
            std::printf("  rank %u: %s\n", rank, e.what());
        }
    };
    std::vector<std::thread> ts;
    for (uint32_t r = 0; r < world; ++r) ts.emplace_back(body, r);
    for (auto& t : ts) t.join();
    for (uint32_t r = 0; r < world; ++r) EXPECT(ok[r] == 1);
    EXPECT(n_ex[0] == n_ex[1] && n_ex[0] >= 4);
    // one rank, no transport: the same entry points give the same proofs
    Comm solo;
    auto got1 = ShardedSumcheck::prove(Multilinear(full), solo);
    EXPECT(got1.second == want_sc.second && got1.first.sum == want_sc.first.sum);
    EXPECT(panics([&] { Comm bad(0, 3, thread_all_gather, nullptr); }));       // world must be a power of two
}
// ---- domain / NTT ---------------------------------------------------------------------------------------------------------
TEST(test_domain_new) {   // domain.rs:154-168 (the decimal strings are checked through the oracle's KAT-pinned root)
    Domain d(10);
    EXPECT(d.size == 16);
    Fr w;
    ora_fr_get_root_of_unity((fr_t*)&w, 16);
    EXPECT(d.generator == w && d.generator * d.group_gen_inverse == Fr::one());
}
TEST(test_multiply) {
    auto c = UnivariateEval::multiply(DenseUnivariatePolynomial(F({6, 5, 3})), DenseUnivariatePolynomial(F({5, 4, 2}))).coefficients();
    EXPECT(c == F({30, 49, 47, 22, 6}));                                                       // dense_univariate.rs:480-496 values
    auto a = random_fr(300, 5), b = random_fr(211, 6);
    std::vector<Fr> want(510);
    ora_univariate_multiply(O(want), O(a), 300, O(b), 211);
    EXPECT(UnivariateEval::multiply(DenseUnivariatePolynomial(a), DenseUnivariatePolynomial(b)).coefficients() == want);
    Domain d(64);
    auto x = random_fr(64, 8);
    EXPECT(d.ifft(d.fft(x)) == x);
}

int main() {
    for (auto& t : tests()) {
        int before = g_failed;
        try { t.f(); } catch (const std::exception& e) { std::printf("  EXCEPTION in %s: %s\n", t.n, e.what()); ++g_failed; }
        std::printf("%s %s\n", g_failed == before ? "ok    " : "FAILED", t.n);
        ++g_run;
    }
    std::printf("%d tests, %d failed\n", g_run, g_failed);
    return g_failed ? 1 : 0;
}
