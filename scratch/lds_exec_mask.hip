// Does the LDS skip the lane groups of a ds_read_b128 whose lanes are all masked off by EXEC?
// (MI355X_MICROARCH.md: a wave64 ds_read_b128 is serviced in four fixed groups of 16 lanes, one
// LDS cycle each.)  If it does, padded stream slots could be made free for the LDS array by
// masking their lanes -- the LDS-resident passes are bound by that array (DESIGN_HISTORY.md 4.2).
// One 1024-thread workgroup per CU, every wave issues 8 ds_read_b128 per iteration under a mask.
//   hipcc --offload-arch=gfx950 -O3 scratch/lds_exec_mask.hip -o scratch/lds_exec_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(1024) void probe(double* out, unsigned long long mask, int nfma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)((lane & 15) * 16 + (lane >> 4) * 256 + (threadIdx.x >> 6) * 1024);
    for (int i = threadIdx.x; i < 65536 / 8; i += 1024) reinterpret_cast<double*>(smem)[i] = 1.0;
    __syncthreads();
    u32x4 s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 1.0 + i + lane;
    double x = 1.000001;
    for (int it = 0; it < ITER; ++it) {
        unsigned long long keep;
        asm volatile(
            "s_mov_b64 %[keep], exec\n\t"
            "s_mov_b64 exec, %[m]\n\t"
            "ds_read_b128 %[a0], %[ad]\n\t"
            "ds_read_b128 %[a1], %[ad] offset:4096\n\t"
            "ds_read_b128 %[a2], %[ad] offset:8192\n\t"
            "ds_read_b128 %[a3], %[ad] offset:12288\n\t"
            "ds_read_b128 %[a4], %[ad] offset:16384\n\t"
            "ds_read_b128 %[a5], %[ad] offset:20480\n\t"
            "ds_read_b128 %[a6], %[ad] offset:24576\n\t"
            "ds_read_b128 %[a7], %[ad] offset:28672\n\t"
            "s_mov_b64 exec, %[keep]\n\t"
            : [keep] "=&s"(keep), [a0] "+v"(s0), [a1] "+v"(s1), [a2] "+v"(s2), [a3] "+v"(s3), [a4] "+v"(s4),
              [a5] "+v"(s5), [a6] "+v"(s6), [a7] "+v"(s7)
            : [m] "s"(mask), [ad] "v"(addr)
            : "memory");
        for (int v = 0; v < nfma; ++v) asm volatile("v_fma_f64 %0, %1, %0, %0" : "+v"(acc[v & 15]) : "v"(x));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    s += s0.x + s1.x + s2.x + s3.x + s4.x + s5.x + s6.x + s7.x;
    if (s == 12345.678) out[0] = s;
}

static unsigned long long lanes(std::initializer_list<std::pair<int, int>> ranges) {
    unsigned long long m = 0;
    for (auto r : ranges)
        for (int l = r.first; l <= r.second; ++l) m |= 1ull << l;
    return m;
}

int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    printf("%s, %d CUs, clock %.2f GHz; 16 waves per CU, 8 ds_read_b128 per wave and iteration\n", p.name,
           p.multiProcessorCount, ghz);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    struct Case { const char* name; unsigned long long m; };
    const unsigned long long A = lanes({{0, 3}, {12, 15}, {20, 27}}), B = lanes({{4, 11}, {16, 19}, {28, 31}});
    const Case cases[] = {
        {"all 64 lanes", ~0ull},
        {"service groups A+B (lanes 0-31)", A | B},
        {"service group A only (16 lanes)", A},
        {"service groups A and C (A + 32)", A | (A << 32)},
        {"lanes 0-15 (half of A, half of B)", 0xffffull},
        {"one 4-lane row group per service group (16 lanes)", lanes({{0, 3}, {4, 7}, {32, 35}, {36, 39}})},
        {"3 of 4 service groups", A | B | (A << 32)},
        {"one lane", 1ull},
    };
    for (int nfma : {0, 16}) {
        for (const Case& c : cases) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            probe<<<256, 1024, 128 * 1024>>>(out, c.m, nfma);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            probe<<<256, 1024, 128 * 1024>>>(out, c.m, nfma);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("fma %2d  %-52s mask %016llx : %7.1f clk per iteration (full rate would be %d)\n", nfma, c.name, c.m,
                   ms * 1e-3 * ghz * 1e9 / ITER, 8 * 16 * 4);
        }
    }
    return 0;
}
