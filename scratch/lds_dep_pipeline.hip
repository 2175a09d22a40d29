// Follow-up to lds_valu_overlap.hip: the FMAs now CONSUME the ds_read_b128 results, as in the
// sparse passes.  Per loop iteration (one "trip" of 4 entries, AD/BD form): 8 ds_read_b128,
// 16 v_fma_f64 on the loaded data, 20 filler VALU (unpack / address / convert).
//   MODE 0  reads, then FMAs behind progressive lgkmcnt waits          (what hipcc emits)
//   MODE 1  half-trip software pipeline: 4 reads -> A | FMAs(B) | 4 reads -> B | FMAs(A),
//           the FMAs of a half run while the other half's reads are in flight
//   MODE 2  full-trip pipeline: 8 reads -> next buffer | 16 FMAs on the previous buffer
// 1024 / 512 threads per CU.   hipcc --offload-arch=gfx950 -O3 lds_dep_pipeline.hip -o lds_dep_pipeline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096;
typedef double d2 __attribute__((ext_vector_type(2)));

#define RD(dst, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr))
#define FILL5() asm volatile("v_and_b32 %0, 0x1ff80, %0\n\tv_add_u32 %1, %0, %1\n\tv_add_u32 %1, %0, %1\n\t" \
                             "v_ashrrev_i32 %1, 17, %1\n\tv_cvt_f64_i32 %2, %1" : "+v"(t0), "+v"(t1), "=v"(val))
#define WAIT(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory")
#define FMA2(B_) asm volatile("v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %4" : "+v"(a0), "+v"(a1) : "v"(val), "v"(B_[0]), "v"(B_[1]))
#define FMA2b(B_) asm volatile("v_fmac_f64 %0, %2, %3\n\tv_fmac_f64 %1, %2, %4" : "+v"(a2), "+v"(a3) : "v"(val), "v"(B_[0]), "v"(B_[1]))

template <int MODE>
__global__ __launch_bounds__(1024) void probe(double* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    uint32_t addr = (uint32_t)((lane & 15) * 16 + (lane >> 4) * 256 + (threadIdx.x >> 6) * 1024);
    for (int i = threadIdx.x; i < 65536 / 8; i += blockDim.x) reinterpret_cast<double*>(smem)[i] = 1.0;
    __syncthreads();
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0, val = 1.0;
    uint32_t t0 = lane, t1 = lane * 3;
    d2 x0, x1, x2, x3, x4, x5, x6, x7, y0, y1, y2, y3, y4, y5, y6, y7;
    y0 = y1 = y2 = y3 = y4 = y5 = y6 = y7 = 0.0;
    x0 = x1 = x2 = x3 = x4 = x5 = x6 = x7 = 0.0;
    if (MODE == 0) {
        for (int it = 0; it < ITER; ++it) {
            FILL5(); FILL5();
            RD(x0, 0); RD(x1, 4096); RD(x2, 8192); RD(x3, 12288);
            RD(x4, 16384); RD(x5, 20480); RD(x6, 24576); RD(x7, 28672);
            FILL5(); FILL5();
            WAIT(7); FMA2(x0); WAIT(6); FMA2b(x1); WAIT(5); FMA2(x2); WAIT(4); FMA2b(x3);
            WAIT(3); FMA2(x4); WAIT(2); FMA2b(x5); WAIT(1); FMA2(x6); WAIT(0); FMA2b(x7);
        }
    } else if (MODE == 1) {
        for (int it = 0; it < ITER; ++it) {
            FILL5(); FILL5();
            RD(x0, 0); RD(x1, 4096); RD(x2, 8192); RD(x3, 12288);       // half A of this trip
            WAIT(4);                                                     // half B of the previous trip
            FMA2(y0); FMA2b(y1); FMA2(y2); FMA2b(y3);
            FILL5(); FILL5();
            RD(y0, 16384); RD(y1, 20480); RD(y2, 24576); RD(y3, 28672);  // half B of this trip
            WAIT(4);                                                     // half A
            FMA2(x0); FMA2b(x1); FMA2(x2); FMA2b(x3);
        }
    } else if (MODE == 3) {  // mode 0 + the entry words read from the LDS ring at the top of the trip
        uint32_t raddr = (uint32_t)(65536 + (lane >> 2) * 4 + (threadIdx.x >> 6) * 2048);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 wa, wb;
        for (int it = 0; it < ITER; ++it) {
            asm volatile("ds_read2_b32 %0, %2 offset1:16\n\tds_read2_b32 %1, %2 offset0:32 offset1:48"
                         : "=v"(wa), "=v"(wb) : "v"(raddr));
            WAIT(0);
            t0 += wa[0] + wb[1];
            FILL5(); FILL5();
            RD(x0, 0); RD(x1, 4096); RD(x2, 8192); RD(x3, 12288);
            RD(x4, 16384); RD(x5, 20480); RD(x6, 24576); RD(x7, 28672);
            FILL5(); FILL5();
            WAIT(7); FMA2(x0); WAIT(6); FMA2b(x1); WAIT(5); FMA2(x2); WAIT(4); FMA2b(x3);
            WAIT(3); FMA2(x4); WAIT(2); FMA2b(x5); WAIT(1); FMA2(x6); WAIT(0); FMA2b(x7);
        }
    } else if (MODE == 4) {  // mode 3 with the words of the NEXT trip requested behind the slices
        uint32_t raddr = (uint32_t)(65536 + (lane >> 2) * 4 + (threadIdx.x >> 6) * 2048);
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 wa = 0, wb = 0;
        for (int it = 0; it < ITER; ++it) {
            t0 += wa[0] + wb[1];
            FILL5(); FILL5();
            RD(x0, 0); RD(x1, 4096); RD(x2, 8192); RD(x3, 12288);
            RD(x4, 16384); RD(x5, 20480); RD(x6, 24576); RD(x7, 28672);
            asm volatile("ds_read2_b32 %0, %2 offset1:16\n\tds_read2_b32 %1, %2 offset0:32 offset1:48"
                         : "=v"(wa), "=v"(wb) : "v"(raddr));
            FILL5(); FILL5();
            WAIT(9); FMA2(x0); WAIT(8); FMA2b(x1); WAIT(7); FMA2(x2); WAIT(6); FMA2b(x3);
            WAIT(5); FMA2(x4); WAIT(4); FMA2b(x5); WAIT(3); FMA2(x6); WAIT(2); FMA2b(x7);
            WAIT(0);
        }
    } else if (MODE == 5 || MODE == 6) {
        // mode 3 with the rows taken from the ring words (random 128-B half rows inside 64 KiB):
        // MODE 5 random halves (bank conflicts between the two groups that share a rotation),
        // MODE 6 halves fixed per lane group (g % 8 < 4 -> half 0, else half 1: conflict-free)
        const int wave = threadIdx.x >> 6, g = lane >> 2;
        uint32_t* ringw = reinterpret_cast<uint32_t*>(smem + 65536) + wave * 512;
        uint32_t seed = threadIdx.x * 2654435761u + 12345u;
        for (int i = lane; i < 512; i += 64) {
            seed = seed * 1664525u + 1013904223u;
            uint32_t w = (seed >> 8) & 0xff00u;                 // 256-B row
            const int gg = i & 15;                              // the lane group this word feeds
            if (MODE == 5) w |= (seed >> 3) & 0x80u; else w |= (gg % 8 >= 4) ? 0x80u : 0u;
            ringw[i] = w;
        }
        __syncthreads();
        typedef uint32_t u2 __attribute__((ext_vector_type(2)));
        u2 wa, wb;
        const uint32_t lo = ((lane & 3) * 32 + (g & 1) * 16), lo2 = lo ^ 16;
        const uint32_t rbase = (uint32_t)(65536 + wave * 2048 + g * 4);
        for (int it = 0; it < ITER; ++it) {
            const uint32_t raddr = rbase + (uint32_t)((it & 7) * 256);
            asm volatile("ds_read2_b32 %0, %2 offset1:16\n\tds_read2_b32 %1, %2 offset0:32 offset1:48"
                         : "=v"(wa), "=v"(wb) : "v"(raddr));
            WAIT(0);
            uint32_t ad[8];
            const uint32_t ww[4] = {wa[0], wa[1], wb[0], wb[1]};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                uint32_t r, a, b;
                asm volatile("v_and_b32 %0, 0xff80, %3\n\tv_add_u32 %1, %0, %4\n\tv_add_u32 %2, %0, %5"
                             : "=&v"(r), "=&v"(a), "=&v"(b) : "v"(ww[u]), "v"(lo), "v"(lo2));
                ad[2 * u] = a;
                ad[2 * u + 1] = b;
            }
#define RDA(dst, a) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(a))
            RDA(x0, ad[0]); RDA(x1, ad[1]); RDA(x2, ad[2]); RDA(x3, ad[3]);
            RDA(x4, ad[4]); RDA(x5, ad[5]); RDA(x6, ad[6]); RDA(x7, ad[7]);
            FILL5(); FILL5();  // (values: shift + convert; 8 ops stand in for 2 x 4)
            WAIT(7); FMA2(x0); WAIT(6); FMA2b(x1); WAIT(5); FMA2(x2); WAIT(4); FMA2b(x3);
            WAIT(3); FMA2(x4); WAIT(2); FMA2b(x5); WAIT(1); FMA2(x6); WAIT(0); FMA2b(x7);
        }
    } else {
        for (int it = 0; it < ITER; it += 2) {
            FILL5(); FILL5(); FILL5(); FILL5();
            RD(x0, 0); RD(x1, 4096); RD(x2, 8192); RD(x3, 12288);
            RD(x4, 16384); RD(x5, 20480); RD(x6, 24576); RD(x7, 28672);
            WAIT(8);
            FMA2(y0); FMA2b(y1); FMA2(y2); FMA2b(y3); FMA2(y4); FMA2b(y5); FMA2(y6); FMA2b(y7);
            FILL5(); FILL5(); FILL5(); FILL5();
            RD(y0, 0); RD(y1, 4096); RD(y2, 8192); RD(y3, 12288);
            RD(y4, 16384); RD(y5, 20480); RD(y6, 24576); RD(y7, 28672);
            WAIT(8);
            FMA2(x0); FMA2b(x1); FMA2(x2); FMA2b(x3); FMA2(x4); FMA2b(x5); FMA2(x6); FMA2b(x7);
        }
    }
    WAIT(0);
    double s = a0 + a1 + a2 + a3 + t0 + t1 + y0.x + y7.y + x0.x + x7.y;
    if (s == 12345.678) out[0] = s;
}

template <int MODE>
int run(double* out, double ghz, int threads) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto k = probe<MODE>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    k<<<256, threads, 128 * 1024>>>(out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    k<<<256, threads, 128 * 1024>>>(out);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = threads / 64;
    printf("mode %d waves=%2d : %7.1f clk per trip-round   (LDS array %d, VALU %d per SIMD)\n", MODE, waves,
           ms * 1e-3 * ghz * 1e9 / ITER, 8 * 4 * waves, (16 + 20) * 4 * waves / 4);
    return 0;
}

int main() {
    double* out; CK(hipMalloc(&out, 8));
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate * 1e-6;
    for (int threads : {1024, 512}) { run<0>(out, ghz, threads); run<1>(out, ghz, threads); run<2>(out, ghz, threads); run<3>(out, ghz, threads); run<4>(out, ghz, threads); run<5>(out, ghz, threads); run<6>(out, ghz, threads); }
    return 0;
}
