// VALU issue-rate probe (gfx950): cycles per wave64 instruction and SIMD for the conversions
// and arithmetic the sparse passes use.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, UN = 16;
template <int OP>
__global__ __launch_bounds__(256) void probe(uint32_t seed, double* out) {
    uint32_t a[UN];
    double d[UN];
    float f[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) { a[u] = seed + threadIdx.x * 17 + u; d[u] = 1.0 + u; f[u] = 1.0f + u; }
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(d[u]));
            if (OP == 1) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[u]) : "v"(a[u]));
            if (OP == 2) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[u]) : "v"(f[u]));
            if (OP == 3) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[u]) : "v"(a[u]));
            if (OP == 4) asm volatile("v_bfe_u32 %0, %1, 11, 11" : "=v"(a[u]) : "v"(a[u]));
            if (OP == 5) asm volatile("v_mad_u32_u24 %0, %1, %1, %1" : "=v"(a[u]) : "v"(a[u]));
            if (OP == 6) asm volatile("v_add_f64 %0, %0, %0" : "+v"(d[u]));
            if (OP == 7) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[u]) : "v"(a[u]));
            if (OP == 8) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(d[u]));
            if (OP == 9) asm volatile("v_fmac_f64 %0, %1, %1" : "+v"(d[u]) : "v"(d[(u + 1) % UN]));
        }
    }
    double s = 0;
#pragma unroll
    for (int u = 0; u < UN; ++u) s += d[u] + a[u] + f[u];
    if (s == 12345.678) out[0] = s;
}
template <int OP>
int run(const char* name, double* out, double clk_ghz) {
    const int blocks = 256 * 8;  // 8 blocks x 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<OP><<<blocks, 256>>>(1, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<OP><<<blocks, 256>>>(1, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)blocks * 4 / (256 * 4) * ITER * UN;  // wave-instructions per SIMD
    printf("%-18s %.3f ms  -> %.2f clk per wave64 instruction per SIMD (at %.1f GHz)\n", name, ms,
           ms * 1e-3 * clk_ghz * 1e9 / inst_per_simd, clk_ghz);
    return 0;
}
int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    run<0>("v_fma_f64", out, ghz); run<9>("v_fmac_f64 (2 src)", out, ghz); run<6>("v_add_f64", out, ghz); run<8>("v_mul_f64", out, ghz);
    run<1>("v_cvt_f64_u32", out, ghz); run<2>("v_cvt_f64_f32", out, ghz); run<3>("v_cvt_f32_ubyte1", out, ghz);
    run<7>("v_cvt_f32_u32", out, ghz); run<4>("v_bfe_u32", out, ghz); run<5>("v_mad_u32_u24", out, ghz);
    return 0;
}
