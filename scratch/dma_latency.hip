// Latency of streaming 1-KiB chunks per wave, as the entry-stream ring of vrx_spmm_lds does:
// every wave of a 1024-thread workgroup per CU walks its own contiguous region and keeps D
// loads in flight (wait for the oldest, issue the next).  Reports clk per chunk per wave =
// latency / D when latency-bound.  KIND 0: LDS-DMA (global_load_lds_dwordx4);
// KIND 1: plain global_load_dwordx4 into VGPRs.  span: bytes per wave (small -> L2 resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int KIND, int D>
__global__ __launch_bounds__(1024) void probe(const uint32_t* buf, int64_t words_per_wave, int iters, int wrap, double* out) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t* base = buf + ((int64_t)blockIdx.x * 16 + wave) * words_per_wave;
    u4 r[D];
#pragma unroll
    for (int d = 0; d < D; ++d) r[d] = 0;
    uint32_t acc = 0;
    auto issue = [&](int i, int d) {
        const uint32_t* src = base + (int64_t)(i % wrap) * 256 + 4 * lane;
        if (KIND == 0) {
            const uint32_t dst = __builtin_amdgcn_readfirstlane((uint32_t)(wave * 2048 + (i & 1) * 1024));
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[d]) : "v"(src) : "memory");
        }
    };
#pragma unroll
    for (int d = 0; d < D; ++d) issue(d, d);
    for (int i = D; i < iters; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            // oldest of the D outstanding loads
            if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            if (KIND == 1) acc += r[d].x;
            issue(i + d, d);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (acc == 0x12345678u) out[0] = acc + smem[lane];
}

template <int KIND, int D>
int run(const uint32_t* buf, int64_t wpw, int iters, int wrap, double ghz, double* out, const char* tag) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = probe<KIND, D>;
    k<<<256, 1024, 64 * 1024>>>(buf, wpw, iters, wrap, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<<<256, 1024, 64 * 1024>>>(buf, wpw, iters, wrap, out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double clk = ms * 1e-3 * ghz * 1e9 / iters;
    const double gbs = 256.0 * 16 * iters * 1024 / (ms * 1e-3) / 1e9;
    printf("%-8s %s D=%d : %7.0f clk per chunk and wave  (=> latency ~%6.0f clk = %.2f us), %7.1f GB/s\n", tag,
           KIND == 0 ? "lds-dma " : "vgpr    ", D, clk, clk * D, clk * D / ghz / 1e3, gbs);
    fflush(stdout);
    return 0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    const int iters = 512;                       // 512 KiB per wave, 2 GiB in total
    const int64_t wpw = (int64_t)iters * 256;
    uint32_t* buf; CK(hipMalloc(&buf, (size_t)256 * 16 * wpw * 4));
    CK(hipMemset(buf, 1, (size_t)256 * 16 * wpw * 4));
    double* out; CK(hipMalloc(&out, 8));
    CK(hipDeviceSynchronize()); printf("buffers ready\n"); fflush(stdout);
    run<1, 1>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<1, 2>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<1, 4>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<1, 8>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<1, 1>(buf, wpw, iters, 4, ghz, out, "L2");
    run<1, 4>(buf, wpw, iters, 4, ghz, out, "L2");
    run<0, 1>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<0, 2>(buf, wpw, iters, iters, ghz, out, "HBM");
    run<0, 4>(buf, wpw, iters, iters, ghz, out, "HBM");
    // L2-resident: every wave re-reads 4 chunks (16 MiB in total over the chip)
    run<0, 1>(buf, wpw, iters, 4, ghz, out, "L2");
    run<0, 2>(buf, wpw, iters, 4, ghz, out, "L2");
    run<1, 1>(buf, wpw, iters, 4, ghz, out, "L2");
    run<1, 4>(buf, wpw, iters, 4, ghz, out, "L2");
    return 0;
}
