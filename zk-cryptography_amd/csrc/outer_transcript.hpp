// outer_transcript.hpp -- the device-resident state of an OUTER Fiat-Shamir transcript fed beside the rounds of the composed provers
// (composed_kernels.hpp "an OUTER transcript fed beside the rounds"; used by gkr.hip: GKRProtocol::prove's own transcript,
// gkr/src/protocol.rs:25,91,104-105).
#pragma once
#include "transcript.hpp"

namespace zk {

constexpr int CMP_OUTER_ROUNDS = 48;    // >= ZK_MAX_ROUNDS (composed.hip asserts it)
constexpr int CMP_OUTER_MONO = 7;       // = CMP_MAX_MONO: monomials of a round polynomial
struct OuterDev {                       // device-resident, one per proof in flight
    Sha256State state;                  // the outer transcript between kernels
    uint32_t error;                     // a hasher gave up waiting for a flag (never on a healthy run): 1 + round
    uint32_t pad_[3];
    uint32_t flag[CMP_OUTER_ROUNDS];    // == the session's token once round r's items lie in items[r]
    uint32_t items[CMP_OUTER_ROUNDS][1 + CMP_OUTER_MONO * 9];   // n, then per monomial the canonical coefficient (8 words, little endian) and the power
};
struct OuterPub {
    OuterDev* dev;                      // nullptr: no outer transcript
    uint32_t token;
};

}  // namespace zk
