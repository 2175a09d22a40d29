// How fast is a ~30-us streaming kernel with 8-B against 16-B accesses per lane on MI355X?
// (the dense kernels of an iteration move 56-123 MB each at 2.7-3.5 TB/s with 8-B accesses)
//   hipcc --offload-arch=gfx950 -O3 scratch/stream_probe.hip -o scratch/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int W>  // doubles per access
__global__ __launch_bounds__(256) void probe(const double* __restrict__ a, double* __restrict__ b, long n_read, int ratio) {
    // thread reads `ratio` accesses of W doubles (strided by the grid like the planes of a pass) and writes one
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i * W < n_read / ratio; i += stride) {
        double s = 0.0;
        double v[8][W];
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < ratio) {
                if (W == 2) { const double2 x = reinterpret_cast<const double2*>(a)[i + (long)r * (n_read / ratio / 2)]; v[r][0] = x.x; v[r][W - 1] = x.y; }
                else v[r][0] = a[i + (long)r * (n_read / ratio)];
            }
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < ratio)
#pragma unroll
                for (int w = 0; w < W; ++w) s += v[r][w];
        if (W == 2) reinterpret_cast<double2*>(b)[i] = make_double2(s, s + 1.0);
        else b[i] = s;
    }
}
int main() {
    const long n_read = 12l << 20, ratio = 4;      // 12 Mi doubles = 100 MB read, 25 MB written
    double *a, *b;
    CK(hipMalloc(&a, n_read * 8)); CK(hipMalloc(&b, n_read / ratio * 8));
    CK(hipMemset(a, 0, n_read * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1024, 2048, 4096, 12288}) {
        for (int w = 1; w <= 2; ++w) {
            float best = 1e9;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                if (w == 1) probe<1><<<blocks, 256>>>(a, b, n_read, (int)ratio);
                else probe<2><<<blocks, 256>>>(a, b, n_read, (int)ratio);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            }
            printf("blocks %5d  %2d B per lane : %6.1f us  %5.2f TB/s\n", blocks, 8 * w, best * 1e3, (n_read * 8.0 * (1 + 1.0 / ratio)) / (best * 1e-3) / 1e12);
        }
    }
    return 0;
}
