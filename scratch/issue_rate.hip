// Does a SIMD issue a scalar / wait / LDS instruction of one wave in the same slot as a vector
// instruction of another?  Loop body: NV v_fma_f64 interleaved with NS scalar-side instructions;
// 4 waves per SIMD (1024-thread workgroup per CU).  If time ~ 4 clk x (NV + NS) the SIMD is
// effectively single-issue for this mix; if ~ 4 clk x NV the scalar side is free.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;
// KIND 0: no extra; 1: s_add_u32; 2: s_waitcnt lgkmcnt(0) (nothing outstanding); 3: s_nop 0;
// 4: v_mov_b32 (cheap VALU) ; 5: s_cmp + s_cbranch (never taken)
template <int KIND>
__global__ __launch_bounds__(1024) void probe(double* out, int z) {
    double a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = 1.0 + i + threadIdx.x;
    double x = 1.0000001;
    int sc = z;
    uint32_t vm = threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(a[i % 16]) : "v"(x));
            if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
            if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
            if (KIND == 3) asm volatile("s_nop 0");
            if (KIND == 4) asm volatile("v_mov_b32 %0, %0" : "+v"(vm));
            if (KIND == 5) asm volatile("s_cmp_eq_u32 %0, -7\n\ts_cbranch_scc1 1f\n1:" : : "s"(sc));
        }
    }
    double s = sc + vm;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 12345.678) out[0] = s;
}
template <int KIND>
int run(const char* name, double* out, double ghz, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    probe<KIND><<<256, threads>>>(out, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    probe<KIND><<<256, threads>>>(out, 0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e-3 * ghz * 1e9 / ITER / 32 / (threads / 256);   // per (fma + extra) pair and wave on a SIMD
    printf("%-34s waves/SIMD=%d : %5.2f clk per fma(+extra) per wave\n", name, threads / 256, clk);
    return 0;
}
int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    for (int threads : {1024, 512, 256}) {
        run<0>("v_fma_f64 alone", out, ghz, threads);
        run<1>("v_fma_f64 + s_add_u32", out, ghz, threads);
        run<2>("v_fma_f64 + s_waitcnt", out, ghz, threads);
        run<3>("v_fma_f64 + s_nop", out, ghz, threads);
        run<4>("v_fma_f64 + v_mov_b32", out, ghz, threads);
        run<5>("v_fma_f64 + s_cmp + s_cbranch", out, ghz, threads);
    }
    return 0;
}
