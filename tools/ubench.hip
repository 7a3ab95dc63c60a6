// ubench.hip -- gfx950 integer-ALU micro-benchmarks that size the field-arithmetic design
// (which instruction mix a 256/384-bit Montgomery product should be built from).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o tools/ubench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITER 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each kernel: ITER iterations x 8 independent instructions per lane
#define BODY8(INS) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)

__global__ void k_mad_u64_u32(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
#define I(n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[n]) : "v"(x), "v"(y) : "vcc");
        BODY8(I)
#undef I
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_addc(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint32_t hi[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int i = 0; i < 8; ++i) { acc[i] = i; hi[i] = 0; }
    for (int it = 0; it < ITER; ++it) {
#define I(n) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc[n]), "+v"(hi[n]) : "v"(x), "v"(y) : "vcc");
        BODY8(I)
#undef I
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i] + hi[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_addc_chain(uint64_t* out, uint32_t a, uint32_t b) {   // ONE dependent accumulator (latency)
    uint64_t acc = 1; uint32_t hi = 0; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#define I(n) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(acc), "+v"(hi) : "v"(x), "v"(y) : "vcc");
        BODY8(I)
#undef I
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + hi;
}
#define K32(NAME, ASM)                                                                      \
__global__ void NAME(uint64_t* out, uint32_t a, uint32_t b) {                                \
    uint32_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;                     \
    for (int i = 0; i < 8; ++i) acc[i] = i + y;                                             \
    for (int it = 0; it < ITER; ++it) {                                                     \
        for (int n = 0; n < 8; ++n) asm volatile(ASM : "+v"(acc[n]) : "v"(x), "v"(y));     \
    }                                                                                       \
    uint32_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];                                \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                         \
}
K32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
K32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
K32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1")
K32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2")
K32(k_add_u32, "v_add_u32 %0, %0, %1")
K32(k_addc, "v_add_co_u32 %0, vcc, %0, %1\n\tv_addc_co_u32 %0, vcc, %0, %2, vcc")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7")
K32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %1, %2, %0")
K32(k_mov, "v_mov_b32 %0, %1")
__global__ void k_lshl_add_u64(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint64_t x = ((uint64_t)a << 32) + threadIdx.x;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int n = 0; n < 8; ++n) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc[n]) : "v"(x));
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma_f64(uint64_t* out, uint32_t a, uint32_t b) {
    double acc[8]; double x = 1.0 + a * 1e-9 + threadIdx.x * 1e-12, y = 1e-7 * b;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int n = 0; n < 8; ++n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[n]) : "v"(x), "v"(y));
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_fma_f32(uint64_t* out, uint32_t a, uint32_t b) {
    float acc[8]; float x = 1.0f + a * 1e-9f + threadIdx.x * 1e-7f, y = 1e-7f * b;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int n = 0; n < 8; ++n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[n]) : "v"(x), "v"(y));
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint64_t)s;
}
__global__ void k_mad_i64_i32(uint64_t* out, uint32_t a, uint32_t b) {
    uint64_t acc[8]; uint32_t x = a + threadIdx.x, y = b ^ threadIdx.x;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int n = 0; n < 8; ++n) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc[n]) : "v"(x), "v"(y) : "vcc");
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad_u64_sgpr(uint64_t* out, uint32_t a, uint32_t b) {   // SGPR multiplier (constant bus)
    uint64_t acc[8]; uint32_t y = b ^ threadIdx.x;
    for (int i = 0; i < 8; ++i) acc[i] = i;
    for (int it = 0; it < ITER; ++it) {
        for (int n = 0; n < 8; ++n) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[n]) : "s"(a), "v"(y) : "vcc");
    }
    uint64_t s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef void (*kern_t)(uint64_t*, uint32_t, uint32_t);
struct Case { const char* name; kern_t k; int inst_per_slot; };

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    const int n_cu = prop.multiProcessorCount;
    uint64_t* d; CHK(hipMalloc(&d, (size_t)n_cu * 8 * 1024 * 8 * 4));
    std::vector<Case> cases = {
        {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_u64_u32(sgpr)", k_mad_u64_sgpr, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
        {"mad_u64+addc pair", k_mad_addc, 1}, {"mad_u64+addc DEPENDENT", k_mad_addc_chain, 1},
        {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1}, {"v_mad_u32_u24", k_mad_u32_u24, 1},
        {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1}, {"v_add3_u32", k_add3_u32, 1}, {"v_add_u32", k_add_u32, 1},
        {"add_co+addc pair", k_addc, 1}, {"v_alignbit_b32", k_alignbit, 1}, {"v_dot4_u32_u8", k_dot4_u32_u8, 1},
        {"v_mov_b32", k_mov, 1}, {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_fma_f64", k_fma_f64, 1}, {"v_fma_f32", k_fma_f32, 1},
    };
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int waves_per_simd : {1, 2, 4, 8}) {
        printf("---- %d wave(s) per SIMD, every CU busy ----\n", waves_per_simd);
        for (auto& c : cases) {
            dim3 grid(n_cu * waves_per_simd), block(256);   // 256 threads = 4 waves = one per SIMD
            hipLaunchKernelGGL(c.k, grid, block, 0, 0, d, 12345u, 6789u);   // warm
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(c.k, grid, block, 0, 0, d, 12345u, 6789u);
            CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
            double slots = (double)ITER * 8;                       // instruction slots per wave
            double cyc_per_slot_per_simd = ms * 1e-3 * 2.4e9 / (slots * waves_per_simd);   // at nominal 2.4 GHz
            double gops = (double)n_cu * 4 * waves_per_simd * 64 * slots / (ms * 1e-3) / 1e9;
            printf("%-26s %8.3f ms  %7.2f cyc/wave-slot/SIMD (@2.4GHz)  %9.1f Glane-ops/s\n", c.name, ms, cyc_per_slot_per_simd, gops);
        }
    }
    return 0;
}
