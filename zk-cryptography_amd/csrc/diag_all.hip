// diag_all.hip -- libzkhip_diag.so (make libzkhip_diag.so): the prover units in one translation unit, -DZK_STAMPS
#include "zkhip.hip"
#include "composed.hip"
#include "gkr.hip"
#include "ntt.hip"
#include "shard.hip"
// the MSM is not part of the diagnostic build: the few entry points the units above reference
extern "C" int zkhip_kzg_commit(zkhip_ctx*, const uint64_t*, const uint8_t*, size_t, const uint64_t*, size_t, int, uint64_t*, uint8_t*) { return ZKHIP_ERR_ARG; }
extern "C" int zkhip_kzg_commit_table(zkhip_ctx*, const void*, const uint8_t*, size_t, const uint64_t*, size_t, int, uint64_t*, uint8_t*) { return ZKHIP_ERR_ARG; }
extern "C" int zkhip_kzg_commit_end(zkhip_ctx*, uint32_t, uint64_t*, uint8_t*) { return ZKHIP_ERR_ARG; }
extern "C" int zkhip_g1_sum_affine(const uint64_t*, const uint8_t*, size_t, uint64_t*, uint8_t*) { return ZKHIP_ERR_ARG; }
