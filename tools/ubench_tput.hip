// ubench_tput.hip -- per-SIMD THROUGHPUT of the multiply-class instructions of gfx950 with 1, 2, 4 and 8 waves per SIMD
// (one workgroup on one CU; every wave runs four independent chains of the instruction).  Decides what the unreduced
// accumulation of the k-variable fold should be built from: v_mad_u64_u32 (what it uses), 24-bit multiplies, or f64 FMAs.
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/ubench_tput tools/ubench_tput.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define BODY_KERNEL(name, asm_body)                                                              \
    __global__ __launch_bounds__(1024) void name(uint32_t* out, int iters) {                     \
        for (int i = 0; i < iters; ++i) {                                                        \
            asm volatile(".rept 128\n" asm_body "\n.endr" ::: "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "vcc", "memory"); \
        }                                                                                        \
        if (threadIdx.x == 5000) out[0] = 1;                                                     \
    }
BODY_KERNEL(k_add, "v_add_u32 v20, v20, v36\nv_add_u32 v21, v21, v36\nv_add_u32 v22, v22, v36\nv_add_u32 v23, v23, v36")
BODY_KERNEL(k_mad64, "v_mad_u64_u32 v[20:21], vcc, v36, v37, v[20:21]\nv_mad_u64_u32 v[22:23], vcc, v36, v37, v[22:23]\nv_mad_u64_u32 v[24:25], vcc, v36, v37, v[24:25]\nv_mad_u64_u32 v[26:27], vcc, v36, v37, v[26:27]")
BODY_KERNEL(k_mullo, "v_mul_lo_u32 v20, v20, v36\nv_mul_lo_u32 v21, v21, v36\nv_mul_lo_u32 v22, v22, v36\nv_mul_lo_u32 v23, v23, v36")
BODY_KERNEL(k_mulhi, "v_mul_hi_u32 v20, v20, v36\nv_mul_hi_u32 v21, v21, v36\nv_mul_hi_u32 v22, v22, v36\nv_mul_hi_u32 v23, v23, v36")
BODY_KERNEL(k_mul24, "v_mul_u32_u24 v20, v20, v36\nv_mul_u32_u24 v21, v21, v36\nv_mul_u32_u24 v22, v22, v36\nv_mul_u32_u24 v23, v23, v36")
BODY_KERNEL(k_mulhi24, "v_mul_hi_u32_u24 v20, v20, v36\nv_mul_hi_u32_u24 v21, v21, v36\nv_mul_hi_u32_u24 v22, v22, v36\nv_mul_hi_u32_u24 v23, v23, v36")
BODY_KERNEL(k_mad24, "v_mad_u32_u24 v20, v20, v36, v37\nv_mad_u32_u24 v21, v21, v36, v37\nv_mad_u32_u24 v22, v22, v36, v37\nv_mad_u32_u24 v23, v23, v36, v37")
BODY_KERNEL(k_fma64, "v_fma_f64 v[20:21], v[36:37], v[38:39], v[20:21]\nv_fma_f64 v[22:23], v[36:37], v[38:39], v[22:23]\nv_fma_f64 v[24:25], v[36:37], v[38:39], v[24:25]\nv_fma_f64 v[26:27], v[36:37], v[38:39], v[26:27]")
BODY_KERNEL(k_add64, "v_add_f64 v[20:21], v[36:37], v[20:21]\nv_add_f64 v[22:23], v[36:37], v[22:23]\nv_add_f64 v[24:25], v[36:37], v[24:25]\nv_add_f64 v[26:27], v[36:37], v[26:27]")
BODY_KERNEL(k_addco, "v_add_co_u32 v20, vcc, v20, v36\nv_addc_co_u32 v21, vcc, v21, v37, vcc\nv_add_co_u32 v22, vcc, v22, v36\nv_addc_co_u32 v23, vcc, v23, v37, vcc")
BODY_KERNEL(k_lshl_add64, "v_lshl_add_u64 v[20:21], v[20:21], 0, v[36:37]\nv_lshl_add_u64 v[22:23], v[22:23], 0, v[36:37]\nv_lshl_add_u64 v[24:25], v[24:25], 0, v[36:37]\nv_lshl_add_u64 v[26:27], v[26:27], 0, v[36:37]")
BODY_KERNEL(k_pkfma32, "v_pk_fma_f32 v[20:21], v[36:37], v[38:39], v[20:21]\nv_pk_fma_f32 v[22:23], v[36:37], v[38:39], v[22:23]\nv_pk_fma_f32 v[24:25], v[36:37], v[38:39], v[24:25]\nv_pk_fma_f32 v[26:27], v[36:37], v[38:39], v[26:27]")

template <class K> static void run(const char* name, K k, uint32_t* d) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 400;
    std::printf("%-20s", name);
    for (int wps : {1, 2, 4, 8}) {          // waves per SIMD
        const int threads = 64 * 4 * wps;
        if (threads > 1024) {               // two workgroups of 1024 cannot be forced onto one CU: use 2 WGs x 1024 on a 1-CU-wide grid is not possible; report n/a
            std::printf("   %dw:    n/a", wps);
            continue;
        }
        hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, 4); hipDeviceSynchronize();
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(1), dim3(threads), 0, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double n = (double)iters * 128 * 4 * wps;   // wave-instructions per SIMD
        std::printf("   %dw: %6.2f cyc", wps, 1e6 * ms / n * 2.4);
    }
    std::printf("   (cycles per wave-instruction per SIMD @2.4 GHz)\n");
}
int main() {
    uint32_t* d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    run("v_add_u32", k_add, d);
    run("v_add_co+addc", k_addco, d);
    run("v_lshl_add_u64", k_lshl_add64, d);
    run("v_mad_u64_u32", k_mad64, d);
    run("v_mul_lo_u32", k_mullo, d);
    run("v_mul_hi_u32", k_mulhi, d);
    run("v_mul_u32_u24", k_mul24, d);
    run("v_mul_hi_u32_u24", k_mulhi24, d);
    run("v_mad_u32_u24", k_mad24, d);
    run("v_fma_f64", k_fma64, d);
    run("v_add_f64", k_add64, d);
    run("v_pk_fma_f32", k_pkfma32, d);
    return 0;
}
