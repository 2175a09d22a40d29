// Do LDS reads and fp64 VALU work overlap on a gfx950 CU, or do their cycles add?
// One 1024-thread workgroup per CU (16 waves, 4 per SIMD, like vrx_spmm_lds); per loop
// iteration every wave issues NL ds_read_b128 (conflict-free, 16 B per lane) and NV v_fma_f64.
//   hipcc --offload-arch=gfx950 -O3 lds_valu_overlap.hip -o lds_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 2048;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// MODE 0: independent streams (inline asm; loaded data never consumed)
// MODE 1: interleaved issue order: one ds_read then NV/NL fmas, ...
// WIDTH: 128 / 64 bit reads
template <int NL, int NV, int MODE, int WIDTH>
__global__ __launch_bounds__(1024) void probe(double* out, int stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    // lane l reads 16 B at quad (l & 15) of row (something): conflict-free for every lane grouping
    uint32_t addr = (uint32_t)((lane & 15) * 16 + (lane >> 4) * 256 + (threadIdx.x >> 6) * 1024);
    double acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = 1.0 + i + lane;
    double x = 1.000001;
    u32x4 sink[16];
    u32x2 sink2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sink[i] = 0; sink2[i] = 0; }
    for (int i = threadIdx.x; i < 32768 / 8; i += 1024) reinterpret_cast<double*>(smem)[i] = 1.0;
    __syncthreads();
    for (int it = 0; it < ITER; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                if (WIDTH == 128)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[l % 16]) : "v"(addr), "n"((l % 4) * 4096));
                else
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(sink2[l % 16]) : "v"(addr), "n"((l % 4) * 4096));
            }
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(acc[v % 32]) : "v"(x));
        } else {
            constexpr int per = NL > 0 ? NV / NL : NV;
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[l % 16]) : "v"(addr), "n"((l % 4) * 4096));
#pragma unroll
                for (int v = 0; v < per; ++v)
                    asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(acc[(l * per + v) % 32]) : "v"(x));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        addr ^= (uint32_t)stride;  // (keeps the address live; stride = 0)
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += sink[i].x + sink[i].y + sink2[i].x;
    if (s == 12345.678) out[0] = s;
}

template <int NL, int NV, int MODE, int WIDTH>
int run(double* out, double ghz, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = probe<NL, NV, MODE, WIDTH>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    k<<<256, threads, 128 * 1024>>>(out, 0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<<<256, threads, 128 * 1024>>>(out, 0);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e-3 * ghz * 1e9 / ITER;  // CU clocks per loop iteration
    const int waves = threads / 64;
    const double lds_ideal = (double)NL * waves * (WIDTH == 128 ? 4 : 2), valu_ideal = (double)NV * waves / 4 * 4;
    printf("NL=%2d NV=%2d mode=%d b%d waves=%2d : %8.1f clk/iter   (LDS array %6.0f, VALU %6.0f, sum %6.0f)\n", NL, NV,
           MODE, WIDTH, waves, clk, lds_ideal, valu_ideal, lds_ideal + valu_ideal);
    return 0;
}

int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, clock %.2f GHz\n", p.name, p.multiProcessorCount, ghz);
    for (int threads : {1024, 512, 256}) {
        run<8, 0, 0, 128>(out, ghz, threads);
        run<16, 0, 0, 128>(out, ghz, threads);
        run<0, 32, 0, 128>(out, ghz, threads);
        run<8, 32, 0, 128>(out, ghz, threads);
        run<8, 32, 1, 128>(out, ghz, threads);
        run<16, 32, 0, 128>(out, ghz, threads);
        run<16, 32, 1, 128>(out, ghz, threads);
        run<8, 16, 0, 128>(out, ghz, threads);
        run<4, 32, 0, 128>(out, ghz, threads);
        run<16, 0, 0, 64>(out, ghz, threads);
        run<16, 32, 0, 64>(out, ghz, threads);
    }
    return 0;
}
