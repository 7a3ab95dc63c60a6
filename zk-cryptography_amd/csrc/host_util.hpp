// host_util.hpp -- small host helpers shared by the translation units of libzkhip
#pragma once
#include <stddef.h>
#include <stdint.h>
static inline uint32_t log2_exact(size_t n) {
    uint32_t k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}
static inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
