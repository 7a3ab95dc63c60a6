// zkhip.hpp -- C++ host mirror of the reference's Rust surfaces for the hot path, header-only over the C ABI
// (include/zkhip.h).  No Rust toolchain exists in the build image, so this C++ layer plays the role of the Rust
// shim: same type and method names, same argument meaning, and the reference's assert!/panic! surface as C++
// exceptions (zkc::Panic carries the reference's message).  Device memory comes from zkhip_malloc; nothing here
// needs PyTorch.
//
//   polynomial::Multilinear / MultilinearTrait         polynomial/src/multilinear/evaluation_form.rs, interface.rs:9-13
//   polynomial::ComposedMultilinear                    polynomial/src/composed/composed_multilinear.rs
//   sumcheck::{Sumcheck, ComposedSumcheck, MultiComposedSumcheckProver}   sumcheck/src/**
//   kzg::{TrustedSetup, MultilinearKZG, UnivariateKZG} kzg/src/{trusted_setup,multilinear_kzg,univariate_kzg}.rs
//   polynomial::univariate::{Domain, UnivariateEval, DenseUnivariatePolynomial}
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "zkhip.h"

namespace zkc {

struct Panic : std::runtime_error { using std::runtime_error::runtime_error; };

// a sharded prover: ANOTHER rank failed and said so through the exchange (ZKHIP_ERR_PEER); this rank's outputs are void
struct PeerFailed : Panic { using Panic::Panic; };

inline void check(int st, const char* what) {
    if (st == ZKHIP_OK) return;
    if (st == ZKHIP_ERR_PEER) throw PeerFailed(std::string(what) + ": " + zkhip_status_string(st));
    throw Panic(std::string(what) + ": " + zkhip_status_string(st));
}

// ---- Fr (ark_test_curves::bls12_381::Fr as the reference's tests use it) -----------------------------------
struct Fr {
    uint64_t l[4];
    static Fr from(int64_t v) { Fr r; zkhip_fr_from_i64(v, r.l); return r; }
    static Fr zero() { return from(0); }
    static Fr one() { return from(1); }
    bool operator==(const Fr& o) const { return std::memcmp(l, o.l, 32) == 0; }
    bool operator!=(const Fr& o) const { return !(*this == o); }
    Fr operator+(const Fr& o) const { Fr r; zkhip_fr_add(l, o.l, r.l); return r; }
    Fr operator-(const Fr& o) const { Fr r; zkhip_fr_sub(l, o.l, r.l); return r; }
    Fr operator*(const Fr& o) const { Fr r; zkhip_fr_mul(l, o.l, r.l); return r; }
    std::vector<uint8_t> to_bytes_be() const {   // into_bigint().to_bytes_be()
        uint64_t c[4];
        zkhip_fr_to_canonical(l, c);
        std::vector<uint8_t> b(32);
        for (int i = 0; i < 4; ++i) for (int k = 0; k < 8; ++k) b[31 - (8 * i + k)] = (uint8_t)(c[i] >> (8 * k));
        return b;
    }
};
static_assert(sizeof(Fr) == 32, "Fr must be 4 x u64");

// ---- context + device memory -------------------------------------------------------------------------------
class Context {
  public:
    explicit Context(int device = 0) { check(zkhip_ctx_create(&c_, device, nullptr), "zkhip_ctx_create"); }
    ~Context() { if (c_) zkhip_ctx_destroy(c_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    zkhip_ctx* raw() const { return c_; }
    // one context per HOST THREAD (include/zkhip.h): threads that prove at once never share one
    static Context& instance() { static thread_local Context ctx(0); return ctx; }
  private:
    zkhip_ctx* c_ = nullptr;
};

class DeviceBuffer {   // owns `bytes` of device memory
  public:
    DeviceBuffer() = default;
    explicit DeviceBuffer(size_t bytes) : bytes_(bytes) { check(zkhip_malloc(Context::instance().raw(), &p_, bytes ? bytes : 1), "zkhip_malloc"); }
    ~DeviceBuffer() { if (p_) zkhip_free(Context::instance().raw(), p_); }
    DeviceBuffer(DeviceBuffer&& o) noexcept : p_(o.p_), bytes_(o.bytes_) { o.p_ = nullptr; o.bytes_ = 0; }
    DeviceBuffer& operator=(DeviceBuffer&& o) noexcept { std::swap(p_, o.p_); std::swap(bytes_, o.bytes_); return *this; }
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    uint64_t* u64() const { return (uint64_t*)p_; }
    uint8_t* u8() const { return (uint8_t*)p_; }
    size_t bytes() const { return bytes_; }
    void upload(const void* h, size_t n) { check(zkhip_memcpy_h2d(Context::instance().raw(), p_, h, n), "h2d"); }
    void download(void* h, size_t n) const { check(zkhip_memcpy_d2h(Context::instance().raw(), h, p_, n), "d2h"); }
  private:
    void* p_ = nullptr;
    size_t bytes_ = 0;
};

inline zkhip_ctx* ctx() { return Context::instance().raw(); }

// ---- polynomial::Multilinear ------------------------------------------------------------------------------------
class Multilinear {
  public:
    size_t n_vars = 0;
    // Multilinear::new (evaluation_form.rs:12-26)
    explicit Multilinear(const std::vector<Fr>& evaluations) : n_(evaluations.size()), dev_(std::make_shared<DeviceBuffer>(32 * evaluations.size())) {
        if (n_ == 0 || (n_ & (n_ - 1))) throw Panic("Number of evaluations must be a power of 2");
        while (((size_t)1 << n_vars) < n_) ++n_vars;
        dev_->upload(evaluations.data(), 32 * n_);
    }
    size_t len() const { return n_; }
    const uint64_t* device() const { return dev_->u64(); }
    std::vector<Fr> evaluations() const { std::vector<Fr> v(n_); dev_->download(v.data(), 32 * n_); return v; }
    bool operator==(const Multilinear& o) const { return n_vars == o.n_vars && evaluations() == o.evaluations(); }

    // MultilinearTrait (interface.rs:9-13)
    Multilinear partial_evaluation(const Fr& eval_point, size_t variable_index) const {   // evaluation_form.rs:123-141
        if (n_ % 2 != 0) throw Panic("n must be even");                                     // utils.rs:30
        if (!(variable_index < n_ / 2)) throw Panic("variable_index must be less than n/2"); // utils.rs:31-34
        // 2^variable_index >= n: the reference's pair list is empty (utils.rs:37-50) -> an empty table with n_vars - 1
        Multilinear out(zkhip_mle_partial_evaluation_len(n_, (uint32_t)variable_index), n_vars - 1);
        check(zkhip_mle_partial_evaluation(ctx(), device(), n_, eval_point.l, nullptr, (uint32_t)variable_index, out.dev_->u64()),
              "variable_index must be less than n/2");
        return out;
    }
    Multilinear partial_evaluations(const std::vector<Fr>& points, const std::vector<size_t>& variable_indices) const {   // :143-159
        if (points.size() != variable_indices.size())
            throw Panic("The length of evaluation_points and variable_indices should be the same");
        std::vector<uint32_t> idx(variable_indices.begin(), variable_indices.end());
        Multilinear out(n_ >> points.size(), n_vars - points.size());
        check(zkhip_mle_partial_evaluations(ctx(), device(), n_, points.empty() ? nullptr : points[0].l, idx.data(), points.size(), out.dev_->u64()),
              "partial_evaluations");
        return out;
    }
    Fr evaluation(const std::vector<Fr>& evaluation_points) const {   // :162-175
        if (evaluation_points.size() != n_vars) throw Panic("Number of evaluation points must match the number of variables");
        Fr r;
        check(zkhip_mle_evaluation(ctx(), device(), n_, evaluation_points.empty() ? nullptr : evaluation_points[0].l, evaluation_points.size(), r.l),
              "evaluation");
        return r;
    }
    Multilinear split_poly_into_two_and_sum_each_part() const {   // :68-74
        uint64_t h[12];
        check(zkhip_mle_half_sums(ctx(), device(), n_, h), "half_sums");
        Fr lo, hi;
        std::memcpy(lo.l, h, 32); std::memcpy(hi.l, h + 4, 32);
        return Multilinear(std::vector<Fr>{lo, hi});
    }
    Fr sum_over_the_boolean_hypercube() const {   // :80-84
        if (n_ == 1) return evaluations()[0];
        uint64_t h[12];
        check(zkhip_mle_half_sums(ctx(), device(), n_, h), "half_sums");
        Fr t; std::memcpy(t.l, h + 8, 32); return t;
    }
    Multilinear add_distinct(const Multilinear& rhs) const { return distinct(rhs, false); }   // :28-39
    Multilinear mul_distinct(const Multilinear& rhs) const { return distinct(rhs, true); }    // :41-52
    Multilinear add_to_front(size_t variable_length) const {                                  // :86-96
        Multilinear out(n_ * ((size_t)2 << variable_length), n_vars + variable_length + 1);
        check(zkhip_mle_add_to_front(ctx(), device(), n_, (uint32_t)variable_length, out.dev_->u64()), "add_to_front");
        return out;
    }
    Multilinear add_to_back(size_t variable_length) const {                                   // :98-110
        Multilinear out(n_ << variable_length, n_vars + variable_length);
        check(zkhip_mle_add_to_back(ctx(), device(), n_, (uint32_t)variable_length, out.dev_->u64()), "add_to_back");
        return out;
    }
    static Multilinear duplicate_evaluation(const std::vector<Fr>& value) {                   // :112-119
        std::vector<Fr> v(value);
        v.insert(v.end(), value.begin(), value.end());
        return Multilinear(v);
    }
    std::vector<uint8_t> to_bytes() const {   // :54-62
        DeviceBuffer b(32 * n_);
        check(zkhip_mle_to_bytes(ctx(), device(), n_, b.u8()), "to_bytes");
        std::vector<uint8_t> out(32 * n_);
        b.download(out.data(), out.size());
        return out;
    }
    Multilinear operator+(const Multilinear& o) const { return elementwise(0, &o, nullptr); }   // :178-194
    Multilinear operator-(const Multilinear& o) const { return elementwise(1, &o, nullptr); }   // :208-224
    Multilinear operator*(const Fr& s) const { return elementwise(2, nullptr, &s); }            // :235-251

  private:
    Multilinear(size_t n, size_t nv) : n_vars(nv), n_(n), dev_(std::make_shared<DeviceBuffer>(32 * (n ? n : 1))) {}
    Multilinear distinct(const Multilinear& rhs, bool mul) const {
        Multilinear out(n_ * rhs.n_, n_vars + rhs.n_vars);
        check((mul ? zkhip_mle_mul_distinct : zkhip_mle_add_distinct)(ctx(), device(), n_, rhs.device(), rhs.n_, out.dev_->u64()), "distinct");
        return out;
    }
    Multilinear elementwise(int op, const Multilinear* o, const Fr* s) const {
        Multilinear out(n_, n_vars);
        check(zkhip_mle_elementwise(ctx(), op, device(), o ? o->device() : nullptr, s ? s->l : nullptr, n_, o ? o->n_ : 0, out.dev_->u64()), "elementwise");
        return out;
    }
    size_t n_ = 0;
    std::shared_ptr<DeviceBuffer> dev_;
    friend class Sumcheck;
    friend class ComposedMultilinear;
};

// ---- sumcheck::Sumcheck (sumcheck/src/sumcheck.rs) --------------------------------------------------------------
struct SumcheckProof {
    Multilinear poly;
    Fr sum;
    std::vector<Multilinear> univariate_poly;   // one 2-evaluation polynomial per round (:11-15)
};
class Sumcheck {
  public:
    explicit Sumcheck(Multilinear poly) : poly_(std::move(poly)), sum_(Fr::zero()) {}   // Sumcheck::new :18-23
    void poly_sum() { sum_ = poly_.sum_over_the_boolean_hypercube(); }                  // :25-27
    const Fr& sum() const { return sum_; }
    std::pair<SumcheckProof, std::vector<Fr>> prove() const {                           // :29-61
        const size_t nv = poly_.n_vars;
        std::vector<Fr> rp(2 * (nv ? nv : 1)), ch(nv ? nv : 1);
        Fr s;
        check(zkhip_sumcheck_prove(ctx(), poly_.device(), poly_.len(), sum_.l, nullptr, nullptr, 0, s.l, rp[0].l, ch[0].l), "sumcheck_prove");
        SumcheckProof proof{poly_, s, {}};
        for (size_t i = 0; i < nv; ++i) proof.univariate_poly.emplace_back(std::vector<Fr>{rp[2 * i], rp[2 * i + 1]});
        ch.resize(nv);
        return {std::move(proof), std::move(ch)};
    }
  private:
    Multilinear poly_;
    Fr sum_;
};

// ---- the sharded provers (include/zkhip.h, "the sharded provers behind ONE call each") ----------------------------------------
// Comm = this rank's end of the exchange (zkhip_comm): a caller-supplied all-gather (a Rust host: rccl-sys; the tests: a barrier
// between threads), the library's own RCCL communicator, or none (one rank).
class Comm {
  public:
    Comm() { check(zkhip_comm_create(ctx(), 0, 1, nullptr, nullptr, &m_), "comm_create"); }                         // one rank
    Comm(uint32_t rank, uint32_t world, zkhip_all_gather_fn fn, void* user) : rank_(rank), world_(world) {
        int st = zkhip_comm_create(ctx(), rank, world, fn, user, &m_);
        if (st == ZKHIP_ERR_SHAPE) throw Panic("world size must be a power of two");
        check(st, "comm_create");
    }
    static Comm rccl(const uint8_t id[128], uint32_t rank, uint32_t world) {                                         // collective over the ranks
        Comm c(nullptr);
        c.rank_ = rank; c.world_ = world;
        check(zkhip_comm_create_rccl(ctx(), id, rank, world, &c.m_), "comm_create_rccl");
        return c;
    }
    ~Comm() { if (m_) zkhip_comm_destroy(m_); }
    Comm(Comm&& o) noexcept : m_(o.m_), rank_(o.rank_), world_(o.world_) { o.m_ = nullptr; }
    Comm(const Comm&) = delete;
    Comm& operator=(const Comm&) = delete;
    zkhip_comm* raw() const { return m_; }
    uint32_t rank() const { return rank_; }
    uint32_t world() const { return world_; }
    uint64_t exchanges() const { uint64_t e = 0; zkhip_comm_stats(m_, &e, nullptr); return e; }
  private:
    explicit Comm(std::nullptr_t) {}
    zkhip_comm* m_ = nullptr;
    uint32_t rank_ = 0, world_ = 1;
};
// rank `rank`'s shard of a table / vector sharded by its low index bits: entries rank, rank + world, ...
template <class T>
inline std::vector<T> shard_interleaved(const std::vector<T>& full, uint32_t rank, uint32_t world) {
    std::vector<T> out;
    for (size_t i = rank; i < full.size(); i += world) out.push_back(full[i]);
    return out;
}
// Sumcheck::prove (sumcheck.rs:29-61) of the table whose shard this rank holds; every rank gets the proof of the whole table
// (SumcheckProof::poly is this rank's shard: the whole table lives on no single GPU)
struct ShardedSumcheck {
    static std::pair<SumcheckProof, std::vector<Fr>> prove(const Multilinear& shard, const Comm& comm, const Fr* claimed_sum = nullptr, uint32_t* exchanges = nullptr) {
        size_t nv = shard.n_vars;
        for (uint32_t w = comm.world(); w > 1; w >>= 1) ++nv;
        std::vector<Fr> rp(2 * (nv ? nv : 1)), ch(nv ? nv : 1);
        Fr s;
        check(zkhip_sumcheck_prove_sharded(comm.raw(), shard.device(), shard.len(), claimed_sum ? claimed_sum->l : nullptr, s.l, rp[0].l, ch[0].l, exchanges), "sumcheck_prove_sharded");
        SumcheckProof proof{shard, s, {}};
        for (size_t i = 0; i < nv; ++i) proof.univariate_poly.emplace_back(std::vector<Fr>{rp[2 * i], rp[2 * i + 1]});
        ch.resize(nv);
        return {std::move(proof), std::move(ch)};
    }
};

// ---- polynomial::ComposedMultilinear + composed provers ---------------------------------------------------------
class ComposedMultilinear {
  public:
    explicit ComposedMultilinear(std::vector<Multilinear> polys) : polys(std::move(polys)) {   // composed_multilinear.rs:12-18
        for (auto& p : this->polys) if (p.n_vars != this->polys[0].n_vars) throw Panic("assertion failed: polys.iter().all(|p| p.n_vars == n_vars)");
    }
    std::vector<Multilinear> polys;
    size_t n_vars() const { return polys[0].n_vars; }
    size_t max_degree() const { return polys.size(); }                                         // :101-103
    ComposedMultilinear partial_evaluation(const Fr& r, size_t k) const {                      // :63-75
        std::vector<Multilinear> out;
        for (auto& p : polys) out.push_back(p.partial_evaluation(r, k));
        return ComposedMultilinear(std::move(out));
    }
    Fr evaluation(const std::vector<Fr>& pts) const {                                          // :51-61
        Fr acc = Fr::one();
        for (auto& p : polys) acc = acc * p.evaluation(pts);
        return acc;
    }
    std::vector<const uint64_t*> ptrs() const { std::vector<const uint64_t*> v; for (auto& p : polys) v.push_back(p.device()); return v; }
    // ComposedMultilinearTrait (interface.rs:15-19): the materialised vectors of composed_multilinear.rs:105-119
    std::vector<Fr> element_wise_product() const { return element_wise(0); }
    std::vector<Fr> element_wise_add() const { return element_wise(1); }
  private:
    std::vector<Fr> element_wise(int op) const {
        if (polys.empty()) throw Panic("index out of bounds: the len is 0 but the index is 0");
        const size_t n = polys[0].len();
        for (auto& p : polys) if (p.len() < n) throw Panic("index out of bounds");
        DeviceBuffer out(32 * n);
        auto pv = ptrs();
        check(zkhip_composed_element_wise(ctx(), op, pv.data(), (uint32_t)pv.size(), n, out.u64()), "element_wise");
        std::vector<Fr> v(n);
        out.download(v.data(), 32 * n);
        return v;
    }
};

struct ComposedSumcheckProof { std::vector<std::vector<Fr>> round_polys; };   // composed_sumcheck.rs:15-18 (poly omitted: caller holds it)
class ComposedSumcheck {
  public:
    explicit ComposedSumcheck(ComposedMultilinear poly) : poly(std::move(poly)) {}
    ComposedMultilinear poly;
    static Fr calculate_poly_sum(const ComposedMultilinear& p) {                               // composed_sumcheck.rs:28-30
        Fr s; auto ptrs = p.ptrs();
        check(zkhip_composed_sum(ctx(), ptrs.data(), (uint32_t)ptrs.size(), p.polys[0].len(), s.l), "composed_sum");
        return s;
    }
    std::pair<ComposedSumcheckProof, std::vector<Fr>> prove() const {                          // :32-67
        const size_t nv = poly.n_vars(), k = poly.polys.size();
        std::vector<Fr> rp((k + 1) * (nv ? nv : 1)), ch(nv ? nv : 1);
        auto ptrs = poly.ptrs();
        check(zkhip_composed_prove(ctx(), ptrs.data(), (uint32_t)k, poly.polys[0].len(), rp[0].l, ch[0].l), "composed_prove");
        ComposedSumcheckProof proof;
        for (size_t r = 0; r < nv; ++r) proof.round_polys.emplace_back(rp.begin() + r * (k + 1), rp.begin() + (r + 1) * (k + 1));
        ch.resize(nv);
        return {std::move(proof), std::move(ch)};
    }
};

struct UnivariateMonomial { Fr coeff, pow; };                                                  // sparse_univariate.rs:11-15
struct SparseUnivariatePolynomial {
    std::vector<UnivariateMonomial> monomial;
    std::vector<uint8_t> to_bytes() const {                                                    // :27-34
        std::vector<uint8_t> out;
        for (auto& m : monomial) { auto c = m.coeff.to_bytes_be(), p = m.pow.to_bytes_be(); out.insert(out.end(), c.begin(), c.end()); out.insert(out.end(), p.begin(), p.end()); }
        return out;
    }
};
struct MultiComposedSumcheckProof {                                                            // multi_composed_sumcheck.rs:12-16
    std::vector<SparseUnivariatePolynomial> round_polys;
    Fr sum;
    std::vector<uint8_t> to_bytes() const { std::vector<uint8_t> o; for (auto& r : round_polys) { auto b = r.to_bytes(); o.insert(o.end(), b.begin(), b.end()); } return o; }
};
class MultiComposedSumcheckProver {
  public:
    static Fr calculate_poly_sum(const std::vector<ComposedMultilinear>& poly) {               // :36-45
        std::vector<const uint64_t*> ptrs; std::vector<uint32_t> sizes;
        flatten(poly, ptrs, sizes);
        Fr s;
        check(zkhip_multi_composed_sum(ctx(), ptrs.data(), sizes.data(), (uint32_t)sizes.size(), poly[0].polys[0].len(), s.l), "multi_composed_sum");
        return s;
    }
    static std::pair<MultiComposedSumcheckProof, std::vector<Fr>> prove(const std::vector<ComposedMultilinear>& poly, const Fr& sum) { return run(poly, sum, 0); }          // :47-54
    static std::pair<MultiComposedSumcheckProof, std::vector<Fr>> prove_partial(const std::vector<ComposedMultilinear>& poly, const Fr& sum) { return run(poly, sum, 1); }  // :56-62
    // prove_partial over rank-interleaved shards of every table (sum = the claimed sum of the WHOLE tables); the proof of the whole claim on every rank
    static std::pair<MultiComposedSumcheckProof, std::vector<Fr>> prove_partial_sharded(const std::vector<ComposedMultilinear>& shards, const Fr& sum, const Comm& comm,
                                                                                         int use_stages = -1, uint32_t* exchanges = nullptr) {
        std::vector<const uint64_t*> ptrs; std::vector<uint32_t> sizes;
        flatten(shards, ptrs, sizes);
        size_t nv = shards[0].n_vars();
        for (uint32_t w = comm.world(); w > 1; w >>= 1) ++nv;
        std::vector<uint32_t> lens(nv ? nv : 1);
        std::vector<uint64_t> rp(7 * 8 * (nv ? nv : 1));
        std::vector<Fr> ch(nv ? nv : 1);
        check(zkhip_multi_composed_prove_sharded(comm.raw(), ptrs.data(), sizes.data(), (uint32_t)sizes.size(), shards[0].polys[0].len(), sum.l, use_stages, lens.data(),
                                                 rp.data(), ch[0].l, exchanges), "multi_composed_prove_sharded");
        return unpack(sum, nv, lens, rp, ch);
    }
  private:
    static std::pair<MultiComposedSumcheckProof, std::vector<Fr>> unpack(const Fr& sum, size_t nv, const std::vector<uint32_t>& lens, const std::vector<uint64_t>& rp, std::vector<Fr>& ch) {
        MultiComposedSumcheckProof proof;
        proof.sum = sum;
        for (size_t r = 0; r < nv; ++r) {
            SparseUnivariatePolynomial sp;
            for (uint32_t m = 0; m < lens[r]; ++m) {
                UnivariateMonomial mono;
                std::memcpy(mono.coeff.l, &rp[(r * 7 + m) * 8], 32);
                std::memcpy(mono.pow.l, &rp[(r * 7 + m) * 8 + 4], 32);
                sp.monomial.push_back(mono);
            }
            proof.round_polys.push_back(std::move(sp));
        }
        ch.resize(nv);
        return {std::move(proof), std::move(ch)};
    }
    static void flatten(const std::vector<ComposedMultilinear>& poly, std::vector<const uint64_t*>& ptrs, std::vector<uint32_t>& sizes) {
        for (auto& t : poly) { for (auto& p : t.polys) ptrs.push_back(p.device()); sizes.push_back((uint32_t)t.polys.size()); }
    }
    static std::pair<MultiComposedSumcheckProof, std::vector<Fr>> run(const std::vector<ComposedMultilinear>& poly, const Fr& sum, int partial) {
        std::vector<const uint64_t*> ptrs; std::vector<uint32_t> sizes;
        flatten(poly, ptrs, sizes);
        const size_t nv = poly[0].n_vars();
        std::vector<uint32_t> lens(nv ? nv : 1);
        std::vector<uint64_t> rp(7 * 8 * (nv ? nv : 1));
        std::vector<Fr> ch(nv ? nv : 1);
        check(zkhip_multi_composed_prove(ctx(), ptrs.data(), sizes.data(), (uint32_t)sizes.size(), poly[0].polys[0].len(), sum.l, partial, lens.data(), rp.data(), ch[0].l),
              "multi_composed_prove");
        return unpack(sum, nv, lens, rp, ch);
    }
};

// ---- kzg ----------------------------------------------------------------------------------------------------------
struct G1Affine {   // the commitment, affine Montgomery Fq limbs (x[6], y[6]) + infinity
    uint64_t xy[12];
    bool infinity;
    bool operator==(const G1Affine& o) const { return infinity == o.infinity && (infinity || std::memcmp(xy, o.xy, 96) == 0); }
};
struct DenseUnivariatePolynomial {                                                            // dense_univariate.rs:15-17
    explicit DenseUnivariatePolynomial(const std::vector<Fr>& c) : n(c.size()), dev(std::make_shared<DeviceBuffer>(32 * (c.size() ? c.size() : 1))) { if (n) dev->upload(c.data(), 32 * n); }
    DenseUnivariatePolynomial(size_t n_, std::shared_ptr<DeviceBuffer> d) : n(n_), dev(std::move(d)) {}
    std::vector<Fr> coefficients() const { std::vector<Fr> v(n); if (n) dev->download(v.data(), 32 * n); return v; }
    Fr evaluate(const Fr& point) const {                                                       // :184-196
        Fr out;
        check(zkhip_dense_evaluate(ctx(), n ? dev->u64() : nullptr, n, point.l, out.l), "dense_evaluate");
        return out;
    }
    size_t degree() const {                                                                    // :199-207
        size_t d = 0;
        check(zkhip_dense_degree(ctx(), n ? dev->u64() : nullptr, n, &d), "dense_degree");
        return d;
    }
    size_t n;
    std::shared_ptr<DeviceBuffer> dev;
};
class TrustedSetup {                                                                           // trusted_setup.rs:9-13 (G1 side)
  public:
    static TrustedSetup setup(const std::vector<Fr>& eval_points) {                            // :15-35
        TrustedSetup s((size_t)1 << eval_points.size());
        check(zkhip_srs_multilinear_g1(ctx(), eval_points.empty() ? nullptr : eval_points[0].l, (uint32_t)eval_points.size(), s.pts_->u64(), s.inf_->u8()), "setup");
        return s;
    }
    size_t len() const { return n_; }
    // shifted-SRS table (zkhip_srs_precompute): built once per SRS, then every commitment uses it
    TrustedSetup& precompute() {
        if (!table_) {
            table_ = std::make_shared<DeviceBuffer>(zkhip_srs_table_bytes(n_));
            check(zkhip_srs_precompute(ctx(), points(), inf(), n_, table_->u8()), "srs_precompute");
        }
        return *this;
    }
    const void* table() const { return table_ ? table_->u8() : nullptr; }
    // folded levels + their shifted tables (zkhip_srs_fold_levels, zkhip_srs_level_tables): built once per SRS, then every `open` uses them
    TrustedSetup& precompute_open() {
        if (!level_tables_ && n_ >= 4 && !(n_ & (n_ - 1))) {
            folded_xy_ = std::make_shared<DeviceBuffer>(96 * (n_ - 1));
            folded_inf_ = std::make_shared<DeviceBuffer>(n_ - 1);
            check(zkhip_srs_fold_levels(ctx(), points(), inf(), n_, folded_xy_->u64(), folded_inf_->u8()), "srs_fold_levels");
            level_tables_ = std::make_shared<DeviceBuffer>(zkhip_srs_level_tables_bytes(n_));
            check(zkhip_srs_level_tables(ctx(), folded_xy_->u64(), folded_inf_->u8(), n_, level_tables_->u8()), "srs_level_tables");
        }
        return *this;
    }
    const uint64_t* folded_xy() const { return folded_xy_ ? folded_xy_->u64() : nullptr; }
    const uint8_t* folded_inf() const { return folded_inf_ ? folded_inf_->u8() : nullptr; }
    const void* level_tables() const { return level_tables_ ? level_tables_->u8() : nullptr; }
    const uint64_t* points() const { return pts_->u64(); }
    const uint8_t* inf() const { return inf_->u8(); }
    explicit TrustedSetup(size_t n) : pts_(std::make_shared<DeviceBuffer>(96 * n)), inf_(std::make_shared<DeviceBuffer>(n)), n_(n) {}
    std::shared_ptr<DeviceBuffer> pts_, inf_, table_, folded_xy_, folded_inf_, level_tables_;
  private:
    size_t n_;
};
inline G1Affine commit_impl(const TrustedSetup& srs, const uint64_t* d_scalars, size_t n, int require_equal) {
    G1Affine g; uint8_t inf = 0;
    int st = srs.table() ? zkhip_kzg_commit_table(ctx(), srs.table(), srs.inf(), srs.len(), d_scalars, n, require_equal, g.xy, &inf)
                         : zkhip_kzg_commit(ctx(), srs.points(), srs.inf(), srs.len(), d_scalars, n, require_equal, g.xy, &inf);
    if (st == ZKHIP_ERR_SHAPE) throw Panic("The length of powers_of_tau_in_g1 and the length of the evaluations of the polynomial should tally!");
    if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: the len of powers_of_tau_in_g1 is smaller than the polynomial");
    check(st, "kzg_commit");
    g.infinity = inf != 0;
    return g;
}
// MultilinearKZG::commitment over (scalars, SRS) sharded by low index bits: this rank's shards in, the whole commitment out on every rank
inline G1Affine commit_sharded(const TrustedSetup& srs_shard, const uint64_t* d_scalars, size_t n, int require_equal, const Comm& comm) {
    G1Affine g; uint8_t inf = 0;
    int st = zkhip_kzg_commit_sharded(comm.raw(), srs_shard.table() ? nullptr : srs_shard.points(), srs_shard.table(), srs_shard.inf(), srs_shard.len(), d_scalars, n,
                                      require_equal, g.xy, &inf);
    if (st == ZKHIP_ERR_SHAPE) throw Panic("The length of powers_of_tau_in_g1 and the length of the evaluations of the polynomial should tally!");
    if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: the len of powers_of_tau_in_g1 is smaller than the polynomial");
    check(st, "kzg_commit_sharded");
    g.infinity = inf != 0;
    return g;
}
struct MultilinearKZGProof {                                                                   // multilinear_kzg.rs:17-21
    Fr evaluation;
    std::vector<G1Affine> proofs;
};
struct MultilinearKZG {
    static G1Affine commitment(const Multilinear& poly, const TrustedSetup& srs) { return commit_impl(srs, poly.device(), poly.len(), 1); }   // multilinear_kzg.rs:33-48
    static G1Affine commitment_sharded(const Multilinear& poly_shard, const TrustedSetup& srs_shard, const Comm& comm) { return commit_sharded(srs_shard, poly_shard.device(), poly_shard.len(), 1, comm); }
    static MultilinearKZGProof open(const Multilinear& poly, const std::vector<Fr>& evaluation_points, const TrustedSetup& srs) {             // :50-88
        MultilinearKZGProof pr;
        const size_t nv = evaluation_points.size();
        std::vector<uint64_t> xy(12 * (nv ? nv : 1));
        std::vector<uint8_t> inf(nv ? nv : 1);
        int st = zkhip_kzg_open_tables(ctx(), poly.device(), poly.len(), nv ? evaluation_points[0].l : nullptr, nv, srs.points(), srs.inf(), srs.len(),
                                       srs.folded_xy(), srs.folded_inf(), srs.level_tables(), pr.evaluation.l, xy.data(), inf.data());
        if (st == ZKHIP_ERR_SHAPE) throw Panic("open: evaluation points / SRS length must match the polynomial (n_vars >= 2)");
        check(st, "kzg_open");
        for (size_t i = 0; i < nv; ++i) {
            G1Affine g;
            std::memcpy(g.xy, &xy[12 * i], 96);
            g.infinity = inf[i] != 0;
            pr.proofs.push_back(g);
        }
        return pr;
    }
};
struct UnivariateKZGProof {                                                                    // univariate_kzg.rs:11-15
    Fr evaluation;
    G1Affine proof;
};
struct UnivariateKZG {
    static UnivariateKZGProof open(const DenseUnivariatePolynomial& poly, const Fr& evaluation_point, const TrustedSetup& srs) {   // :60-81
        UnivariateKZGProof pr;
        uint8_t inf = 0;
        int st = zkhip_univariate_kzg_open(ctx(), poly.n ? poly.dev->u64() : nullptr, poly.n, evaluation_point.l, srs.points(), srs.inf(), srs.len(),
                                           pr.evaluation.l, pr.proof.xy, &inf);
        if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: the len of powers_of_tau_in_g1 is smaller than the quotient");
        check(st, "univariate_kzg_open");
        pr.proof.infinity = inf != 0;
        return pr;
    }
    static TrustedSetup generate_srs(const Fr& tau, size_t max_degree) {                       // univariate_kzg.rs:18-35
        TrustedSetup s(max_degree + 1);
        check(zkhip_srs_univariate_g1(ctx(), tau.l, max_degree, s.pts_->u64(), s.inf_->u8()), "generate_srs");
        return s;
    }
    static G1Affine commitment(const DenseUnivariatePolynomial& poly, const TrustedSetup& srs) { return commit_impl(srs, poly.dev->u64(), poly.n, 0); }   // :37-58
};

// ---- circuit::{Gate, CircuitLayer, Circuit}, gkr::GKRProtocol::prove -----------------------------------------------------
enum class GateType { Add, Mul };                                                              // circuit/src/gate.rs:1-5
struct Gate { GateType gate_type; size_t inputs[2]; };                                         // gate.rs:7-17
struct CircuitLayer { std::vector<Gate> layer; };                                              // circuit.rs:8-11
class Circuit {                                                                                // circuit.rs:13-16
  public:
    std::vector<CircuitLayer> layers;
    explicit Circuit(std::vector<CircuitLayer> l) : layers(std::move(l)) {}
    // device-resident layer values, output layer first, input last (Circuit::evaluation, circuit.rs:31-57)
    struct Evaluation {
        std::vector<std::shared_ptr<DeviceBuffer>> tables;
        std::vector<size_t> lens;
        std::vector<Fr> layer(size_t k) const { std::vector<Fr> v(lens[k]); tables[k]->download(v.data(), 32 * lens[k]); return v; }
    };
    Evaluation evaluation(const std::vector<Fr>& input) const {
        Evaluation ev;
        auto cur = std::make_shared<DeviceBuffer>(32 * input.size());
        cur->upload(input.data(), 32 * input.size());
        ev.tables.push_back(cur); ev.lens.push_back(input.size());
        for (size_t li = layers.size(); li-- > 0;) {
            std::vector<uint8_t> gt; std::vector<uint32_t> i0, i1;
            arrays(layers[li], gt, i0, i1);
            auto out = std::make_shared<DeviceBuffer>(32 * gt.size());
            int st = zkhip_circuit_layer_eval(ctx(), ev.tables.back()->u64(), ev.lens.back(), gt.data(), i0.data(), i1.data(), gt.size(), out->u64());
            if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: gate input");
            check(st, "circuit_layer_eval");
            ev.tables.push_back(out); ev.lens.push_back(gt.size());
        }
        std::reverse(ev.tables.begin(), ev.tables.end());
        std::reverse(ev.lens.begin(), ev.lens.end());
        return ev;
    }
    static void arrays(const CircuitLayer& l, std::vector<uint8_t>& gt, std::vector<uint32_t>& i0, std::vector<uint32_t>& i1) {
        for (auto& g : l.layer) { gt.push_back(g.gate_type == GateType::Add ? 0 : 1); i0.push_back((uint32_t)g.inputs[0]); i1.push_back((uint32_t)g.inputs[1]); }
    }
};
struct GKRProof {                                                                              // gkr/src/protocol.rs:10-15
    std::vector<MultiComposedSumcheckProof> sumcheck_proofs;
    std::vector<Fr> wb_s, wc_s;
    std::vector<Fr> w_0_mle;
};
// A circuit resident in HBM (zkhip_circuit): gate arrays and their groupings validated, built and uploaded once, for provers
// that prove many inputs on one circuit.  GKRProtocol::prove builds one for the call.
class DeviceCircuit {
  public:
    explicit DeviceCircuit(const Circuit& circuit) : n_layers_((uint32_t)circuit.layers.size()) {
        std::vector<size_t> n_gates;
        std::vector<uint8_t> gt; std::vector<uint32_t> i0, i1;
        for (auto& l : circuit.layers) { n_gates.push_back(l.layer.size()); Circuit::arrays(l, gt, i0, i1); }
        check(zkhip_circuit_create(ctx(), n_layers_, n_gates.data(), gt.data(), i0.data(), i1.data(), &handle_), "circuit_create");
    }
    ~DeviceCircuit() { zkhip_circuit_destroy(handle_); }
    DeviceCircuit(const DeviceCircuit&) = delete;
    DeviceCircuit& operator=(const DeviceCircuit&) = delete;
    GKRProof prove(const Circuit::Evaluation& ev) const { return prove_impl(ev, nullptr, -1, nullptr); }   // protocol.rs:21-117
    // the same proof with every layer's sumcheck tables sharded over the ranks of `comm` (the layer values stay whole on every rank)
    GKRProof prove_sharded(const Circuit::Evaluation& ev, const Comm& comm, int use_stages = -1, uint32_t* exchanges = nullptr) const {
        return prove_impl(ev, &comm, use_stages, exchanges);
    }
  private:
    GKRProof prove_impl(const Circuit::Evaluation& ev, const Comm* comm, int use_stages, uint32_t* exchanges) const {
        const uint32_t nl = n_layers_, stride = 2 * nl;
        if (ev.tables.size() != (size_t)nl + 1) throw Panic("circuit evaluation does not match the circuit");
        std::vector<const uint64_t*> ptrs;
        for (auto& t : ev.tables) ptrs.push_back(t->u64());
        std::vector<Fr> sums(nl), wb(nl), wc(nl), w0(2);
        std::vector<uint32_t> n_rounds(nl), lens((size_t)nl * stride);
        std::vector<uint64_t> rps((size_t)nl * stride * 7 * 8);
        int st = comm ? zkhip_gkr_prove_sharded(handle_, comm->raw(), ptrs.data(), ev.lens.data(), use_stages, sums[0].l, n_rounds.data(), lens.data(), rps.data(),
                                                wb[0].l, wc[0].l, w0[0].l, nullptr, exchanges)
                      : zkhip_gkr_prove_circuit(handle_, ptrs.data(), ev.lens.data(), sums[0].l, n_rounds.data(), lens.data(), rps.data(), wb[0].l,
                                                wc[0].l, w0[0].l, nullptr);
        if (st == ZKHIP_ERR_SHAPE) throw Panic("Number of evaluations must be a power of 2");
        if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: gate input");
        check(st, "gkr_prove");
        return unpack(nl, sums.data(), n_rounds.data(), lens.data(), rps.data(), wb.data(), wc.data(), w0.data());
    }
    // the C ABI's packed proof (zkhip_gkr_prove_circuit's arrays) -> GKRProof
    static GKRProof unpack(uint32_t nl, const Fr* sums, const uint32_t* n_rounds, const uint32_t* lens, const uint64_t* rps, const Fr* wb, const Fr* wc, const Fr* w0) {
        const uint32_t stride = 2 * nl;
        GKRProof proof;
        for (uint32_t k = 0; k < nl; ++k) {
            MultiComposedSumcheckProof sp;
            sp.sum = sums[k];
            for (uint32_t r = 0; r < n_rounds[k]; ++r) {
                SparseUnivariatePolynomial poly;
                for (uint32_t m = 0; m < lens[(size_t)k * stride + r]; ++m) {
                    UnivariateMonomial mono;
                    std::memcpy(mono.coeff.l, &rps[(((size_t)k * stride + r) * 7 + m) * 8], 32);
                    std::memcpy(mono.pow.l, &rps[(((size_t)k * stride + r) * 7 + m) * 8 + 4], 32);
                    poly.monomial.push_back(mono);
                }
                sp.round_polys.push_back(poly);
            }
            proof.sumcheck_proofs.push_back(sp);
        }
        proof.wb_s.assign(wb, wb + nl); proof.wc_s.assign(wc, wc + nl); proof.w_0_mle.assign(w0, w0 + 2);
        return proof;
    }
  public:
    // one GKRProtocol::prove per evaluation in ONE call (zkhip_gkr_prove_batch: the proofs run side by side on the context's internal lanes;
    // gkr/benches/gkr_benchmark.rs:11-27 proves input after input); every proof equals prove()'s
    std::vector<GKRProof> prove_batch(const std::vector<Circuit::Evaluation>& evs, uint32_t max_lanes = 0) const {
        const uint32_t nl = n_layers_, stride = 2 * nl, B = (uint32_t)evs.size();
        if (!B) return {};
        std::vector<const uint64_t*> ptrs;
        for (auto& ev : evs) {
            if (ev.tables.size() != (size_t)nl + 1 || ev.lens != evs[0].lens) throw Panic("circuit evaluation does not match the circuit");
            for (auto& t : ev.tables) ptrs.push_back(t->u64());
        }
        std::vector<Fr> sums((size_t)B * nl), wb((size_t)B * nl), wc((size_t)B * nl), w0((size_t)B * 2);
        std::vector<uint32_t> n_rounds((size_t)B * nl), lens((size_t)B * nl * stride);
        std::vector<uint64_t> rps((size_t)B * nl * stride * 7 * 8);
        const int st = zkhip_gkr_prove_batch(handle_, B, max_lanes, ptrs.data(), evs[0].lens.data(), sums[0].l, n_rounds.data(), lens.data(), rps.data(), wb[0].l, wc[0].l,
                                             w0[0].l, nullptr, nullptr);
        if (st == ZKHIP_ERR_SHAPE) throw Panic("Number of evaluations must be a power of 2");
        if (st == ZKHIP_ERR_INDEX) throw std::out_of_range("index out of bounds: gate input");
        check(st, "gkr_prove_batch");
        std::vector<GKRProof> out;
        for (uint32_t b = 0; b < B; ++b)
            out.push_back(unpack(nl, &sums[(size_t)b * nl], &n_rounds[(size_t)b * nl], &lens[(size_t)b * nl * stride], &rps[(size_t)b * nl * stride * 7 * 8],
                                 &wb[(size_t)b * nl], &wc[(size_t)b * nl], &w0[(size_t)b * 2]));
        return out;
    }
  private:
    uint32_t n_layers_;
    zkhip_circuit* handle_ = nullptr;
};
struct GKRProtocol {
    static GKRProof prove(const Circuit& circuit, const Circuit::Evaluation& ev) {             // protocol.rs:21-117
        return DeviceCircuit(circuit).prove(ev);
    }
    static GKRProof prove_sharded(const Circuit& circuit, const Circuit::Evaluation& ev, const Comm& comm) {
        return DeviceCircuit(circuit).prove_sharded(ev, comm);
    }
};

// ---- polynomial::univariate::{Domain, UnivariateEval} -------------------------------------------------------------------
class Domain {
  public:
    uint64_t size;
    Fr generator, group_gen_inverse, group_size_inverse;
    explicit Domain(size_t num_of_coeffs) {                                                     // domain.rs:31-48
        size = 1;
        while (size < num_of_coeffs) size <<= 1;
        check(zkhip_domain_params(size, generator.l, group_gen_inverse.l, group_size_inverse.l), "called `Option::unwrap()` on a `None` value");
    }
    std::vector<Fr> fft(const std::vector<Fr>& coeffs) const { return transform(coeffs, 0); }   // :108-112
    std::vector<Fr> ifft(const std::vector<Fr>& evals) const { return transform(evals, 1); }    // :114-118
  private:
    std::vector<Fr> transform(const std::vector<Fr>& v, int inverse) const {
        const size_t n_src = v.size() < size ? v.size() : (size_t)size;
        DeviceBuffer src(32 * (n_src ? n_src : 1)), d(32 * size);
        if (n_src) src.upload(v.data(), 32 * n_src);
        uint32_t lg = 0; while (((uint64_t)1 << lg) < size) ++lg;
        check(zkhip_domain_transform(ctx(), src.u64(), n_src, d.u64(), lg, inverse), "domain_transform");   // resize(size, zero) inside
        std::vector<Fr> buf(size);
        d.download(buf.data(), 32 * size);
        return buf;
    }
};
struct UnivariateEval {
    static DenseUnivariatePolynomial multiply(const DenseUnivariatePolynomial& a, const DenseUnivariatePolynomial& b) {   // evaluation.rs:59-86
        if (a.n == 0 || b.n == 0) throw Panic("attempt to subtract with overflow");
        auto out = std::make_shared<DeviceBuffer>(32 * (a.n + b.n - 1));
        check(zkhip_univariate_multiply(ctx(), a.dev->u64(), a.n, b.dev->u64(), b.n, out->u64()), "multiply");
        return DenseUnivariatePolynomial(a.n + b.n - 1, out);
    }
};

}  // namespace zkc
