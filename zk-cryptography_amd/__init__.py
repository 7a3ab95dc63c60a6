"""zk-cryptography_amd -- MI355X (gfx950) proving hot path of aagbotemi/zk-cryptography.

Host-side mirror (Python over ctypes) of the reference's Rust surfaces for the hot path:
    polynomial::Multilinear / MultilinearTrait      -> zk_cryptography_amd.polynomial
    sumcheck::{Sumcheck, ComposedSumcheck, MultiComposedSumcheckProver} -> .sumcheck
    kzg::{MultilinearKZG, UnivariateKZG}::commitment, MultilinearKZG::open -> .kzg
    circuit::Circuit, gkr::GKRProtocol::prove -> .gkr
    polynomial::univariate::{Domain, UnivariateEval} -> .univariate
Every operation runs hand-written HIP kernels in csrc/libzkhip.so through the C ABI of
include/zkhip.h; there is no CPU fallback (a missing library or GPU raises).
"""
from zk_cryptography_amd import _native  # noqa: F401
from zk_cryptography_amd.field import Fr  # noqa: F401
from zk_cryptography_amd.polynomial import Multilinear  # noqa: F401
from zk_cryptography_amd.sumcheck import Sumcheck, SumcheckProof  # noqa: F401
from zk_cryptography_amd.kzg import (DenseUnivariatePolynomial, G1Affine, MultilinearKZG, MultilinearKZGProof, TrustedSetup,  # noqa: F401
                                     UnivariateKZG, UnivariateKZGProof)
from zk_cryptography_amd.composed import (ComposedMultilinear, ComposedSumcheck, ComposedSumcheckProof,  # noqa: F401
                                          MultiComposedSumcheckProof, MultiComposedSumcheckProver,
                                          SparseUnivariatePolynomial)
from zk_cryptography_amd.univariate import Domain, UnivariateEval  # noqa: F401
from zk_cryptography_amd.gkr import (Circuit, CircuitLayer, DeviceFiatShamirTranscript, FiatShamirTranscript, Gate, GKRProof, GKRProtocol,  # noqa: F401
                                     SuccintGKRProof, SuccintGKRProtocol)
