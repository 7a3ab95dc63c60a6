// diag_all.hip -- libzkhip_diag.so (make libzkhip_diag.so): the prover units in one translation unit, -DZK_STAMPS
#include "zkhip.hip"
#include "composed.hip"
#include "gkr.hip"
#include "ntt.hip"
