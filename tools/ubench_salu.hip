// ubench_salu.hip -- issue cadence of ONE wave alone on a CU (gfx950): scalar vs vector instructions, dependent vs
// independent, and alternating.  The serial transcript of every prover is one such wave; this says whether its SHA-256
// rounds would run faster on the scalar unit.  Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_salu tools/ubench_salu.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP 512
#define KERNEL(name, body)                                                    \
    __global__ void name(uint32_t* out, int iters) {                          \
        uint32_t s0 = out[0], s1 = out[1], s2 = out[2], s3 = out[3];          \
        uint32_t v0 = threadIdx.x, v1 = s1, v2 = s2, v3 = s3;                 \
        for (int i = 0; i < iters; ++i) {                                     \
            asm volatile(".rept " #body "\n.endr" ::: "memory");              \
        }                                                                     \
        out[4] = s0 + s1 + s2 + s3 + v0 + v1 + v2 + v3;                       \
    }

// hand-written bodies with fixed registers (clobbers declared) so that the compiler cannot interfere
#define BODY_KERNEL(name, asm_body)                                                              \
    __global__ void name(uint32_t* out, int iters) {                                             \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(".rept 512\n" asm_body "\n.endr" ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "v20", "v21", "v22", "v23", "v24", "v25", "scc", "memory"); \
        }                                                                                        \
        if (threadIdx.x == 1000) out[0] = 1;                                                     \
    }
BODY_KERNEL(k_sadd_dep, "s_add_u32 s20, s20, s21")
BODY_KERNEL(k_sadd_ind, "s_add_u32 s20, s20, s24\ns_add_u32 s21, s21, s24\ns_add_u32 s22, s22, s24\ns_add_u32 s23, s23, s24")
BODY_KERNEL(k_sxor_dep, "s_xor_b32 s20, s20, s21")
BODY_KERNEL(k_slshr64_dep, "s_lshr_b64 s[20:21], s[20:21], 7")
BODY_KERNEL(k_vadd_dep, "v_add_u32 v20, v20, v21")
BODY_KERNEL(k_vadd_ind, "v_add_u32 v20, v20, v24\nv_add_u32 v21, v21, v24\nv_add_u32 v22, v22, v24\nv_add_u32 v23, v23, v24")
BODY_KERNEL(k_valign_dep, "v_alignbit_b32 v20, v20, v20, 7")
BODY_KERNEL(k_mix_sv, "s_add_u32 s20, s20, s21\nv_add_u32 v20, v20, v21")
BODY_KERNEL(k_mix_svv, "s_add_u32 s20, s20, s21\nv_add_u32 v20, v20, v21\nv_add_u32 v22, v22, v21")
BODY_KERNEL(k_vadd3_dep, "v_add3_u32 v20, v20, v21, v22")
BODY_KERNEL(k_vxor3_dep, "v_bitop3_b32 v20, v20, v21, v22 bitop3:0x96")
// DPP forms (the six-lane SHA-256 rounds of csrc/transcript.hpp): the chain runs through src1, the DPP source is a register nobody writes
BODY_KERNEL(k_vadd_dpp_quad, "v_add_u32_dpp v20, v22, v20 quad_perm:[1,2,0,3] row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_vxor_dpp_quad, "v_xor_b32_dpp v20, v22, v20 quad_perm:[2,0,1,3] row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_vadd_dpp_shr, "v_add_u32_dpp v20, v22, v20 row_shr:4 row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_vadd_dpp_ror, "v_add_u32_dpp v20, v22, v20 row_ror:4 row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_vadd_dpp_masked, "v_add_u32_dpp v20, v22, v20 quad_perm:[0,1,2,3] row_mask:0xf bank_mask:0x1")
BODY_KERNEL(k_vmov_dpp, "v_mov_b32_dpp v20, v22 row_shr:4 row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_valign_vgpr, "v_alignbit_b32 v20, v20, v20, v21")
BODY_KERNEL(k_snop0, "s_nop 0")
BODY_KERNEL(k_vadd_snop, "v_add_u32 v20, v20, v21\ns_nop 0")
BODY_KERNEL(k_vadd_dpp_after_write, "v_add_u32 v22, v22, v21\nv_add_u32 v23, v23, v21\nv_add_u32 v24, v24, v21\nv_add_u32_dpp v20, v22, v20 row_shr:4 row_mask:0xf bank_mask:0xf")
BODY_KERNEL(k_vperm, "v_perm_b32 v20, v20, v20, v21")
BODY_KERNEL(k_vbfi, "v_bfi_b32 v20, v20, v21, v22")
BODY_KERNEL(k_vlshl_or, "v_lshl_or_b32 v20, v20, v21, v22")
BODY_KERNEL(k_vlshr, "v_lshrrev_b32 v20, 7, v20")

template <class K> static void run(const char* name, K k, int instr_per_rep, uint32_t* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 10); hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double n = (double)iters * 512 * instr_per_rep;
    std::printf("%-22s %8.3f ns/instr  %6.2f cycles @2.4GHz\n", name, 1e6 * ms / n, 1e6 * ms / n * 2.4);
}
int main() {
    uint32_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run("s_add dep", k_sadd_dep, 1, d);
    run("s_add indep x4", k_sadd_ind, 4, d);
    run("s_xor dep", k_sxor_dep, 1, d);
    run("s_lshr_b64 dep", k_slshr64_dep, 1, d);
    run("v_add dep", k_vadd_dep, 1, d);
    run("v_add indep x4", k_vadd_ind, 4, d);
    run("v_alignbit dep", k_valign_dep, 1, d);
    run("v_add3 dep", k_vadd3_dep, 1, d);
    run("v_bitop3 dep", k_vxor3_dep, 1, d);
    run("s,v alternating", k_mix_sv, 2, d);
    run("s,v,v", k_mix_svv, 3, d);
    run("v_add_dpp quad_perm", k_vadd_dpp_quad, 1, d);
    run("v_xor_dpp quad_perm", k_vxor_dpp_quad, 1, d);
    run("v_add_dpp row_shr:4", k_vadd_dpp_shr, 1, d);
    run("v_add_dpp row_ror:4", k_vadd_dpp_ror, 1, d);
    run("v_add_dpp bank-masked", k_vadd_dpp_masked, 1, d);
    run("v_mov_dpp row_shr:4", k_vmov_dpp, 1, d);
    run("v_alignbit vgpr amt", k_valign_vgpr, 1, d);
    run("s_nop 0", k_snop0, 1, d);
    run("v_add, s_nop 0", k_vadd_snop, 2, d);
    run("3 v_add, v_add_dpp", k_vadd_dpp_after_write, 4, d);
    run("v_perm dep", k_vperm, 1, d);
    run("v_bfi dep", k_vbfi, 1, d);
    run("v_lshl_or dep", k_vlshl_or, 1, d);
    run("v_lshrrev dep", k_vlshr, 1, d);
    return 0;
}
