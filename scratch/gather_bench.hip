// Microbenchmark: how fast can gfx950 gather fp64 rows per sparse entry?
// Explores the design space of the two sparse passes (see vireo_amd/csrc/vrx_kernels.h).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/gather_bench.hip -o scratch/gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// ---- variant A: global gather, UN gathers in flight per batch, W waves/SIMD via launch bounds
template <int VW, int UN, int MINW>  // VW = doubles per lane per gather (1: 8B, 2: 16B)
__global__ __launch_bounds__(256, MINW) void gather_global(int64_t nnz, int per_wave,
                                                           const int* __restrict__ idx,
                                                           const long long* __restrict__ val,
                                                           const double* __restrict__ X, int K,
                                                           double* __restrict__ out) {
    constexpr int KP = 16, G = 4;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    if (b >= nnz) return;
    const int g = lane / KP, kl = lane % KP;
    double a1 = 0, a2 = 0;
    for (int off = 0; off < per_wave; off += 64) {
        const int id = __builtin_nontemporal_load(idx + b + off + lane);
        const long long pv = __builtin_nontemporal_load(val + b + off + lane);
        const int ad = (int)(pv & 0xffffffffll), dp = (int)(pv >> 32);
#pragma unroll
        for (int j0 = 0; j0 < KP; j0 += UN) {
            double x0[UN], x1[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const uint32_t r = (uint32_t)__shfl(id, (j0 + u) * G + g, 64);
                if (VW == 1) {
                    x0[u] = X[r * (uint32_t)K + kl];
                } else {
                    const double2 w = reinterpret_cast<const double2*>(X)[r * (uint32_t)K + kl];
                    x0[u] = w.x; x1[u] = w.y;
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int src = (j0 + u) * G + g;
                const double a = (double)__shfl(ad, src, 64), d = (double)__shfl(dp, src, 64);
                if (VW == 1) { a1 += a * x0[u]; a2 += d * x0[u]; }
                else { a1 += a * x0[u] + d * x1[u]; }
            }
        }
    }
    out[wave * 64 + lane] = a1 + a2;
}

// ---- variant S: stream only (no gather): the HBM floor
__global__ __launch_bounds__(256) void stream_only(int64_t nnz, int per_wave, const int* __restrict__ idx,
                                                   const long long* __restrict__ val, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    if (b >= nnz) return;
    long long acc = 0;
    for (int off = 0; off < per_wave; off += 64) {
        acc += __builtin_nontemporal_load(idx + b + off + lane);
        acc += __builtin_nontemporal_load(val + b + off + lane);
    }
    out[wave * 64 + lane] = (double)acc;
}

// ---- variant L: slab staged in LDS, gathers from LDS (rows < slab_rows), 16 B or 8 B per lane
template <int VW>
__global__ __launch_bounds__(1024) void gather_lds(int64_t nnz, int per_wave, int slab_rows,
                                                   const int* __restrict__ idx,
                                                   const long long* __restrict__ val,
                                                   const double* __restrict__ X, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* S = reinterpret_cast<double*>(smem);
    constexpr int KP = 16, G = 4, K = 16;
    const int n_el = slab_rows * K * VW;
    for (int i = threadIdx.x; i < n_el; i += blockDim.x) S[i] = X[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    if (b >= nnz) return;
    const int g = lane / KP, kl = lane % KP;
    double a1 = 0, a2 = 0;
    int id = __builtin_nontemporal_load(idx + b + lane);
    long long pv = __builtin_nontemporal_load(val + b + lane);
    for (int off = 0; off < per_wave; off += 64) {
        const int idc = id; const long long pvc = pv;
        if (off + 64 < per_wave) {
            id = __builtin_nontemporal_load(idx + b + off + 64 + lane);
            pv = __builtin_nontemporal_load(val + b + off + 64 + lane);
        }
        const int ad = (int)(pvc & 0xffffffffll), dp = (int)(pvc >> 32);
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            const int src = u * G + g;
            const uint32_t r = (uint32_t)__shfl(idc, src, 64) % (uint32_t)slab_rows;
            const double a = (double)__shfl(ad, src, 64), d = (double)__shfl(dp, src, 64);
            if (VW == 1) {
                const double x = S[r * K + kl];
                a1 += a * x; a2 += d * x;
            } else {
                const double2 w = reinterpret_cast<const double2*>(S)[r * K + kl];
                a1 += a * w.x + d * w.y;
            }
        }
    }
    out[wave * 64 + lane] = a1 + a2;
}


// ---- variant L2: LDS slab, pre-reduced indices, entry packed in ONE 32-bit word
// (row:20 | ad:6 | dp:6): one cross-lane permute per step.  VW=2: 16-lane groups read 256 B rows
// (ds_read_b128); VW=1: 8-lane groups read 128 B rows (ds_read_b128, 8 entries per step).
template <int VW, int SW>  // SW=1: use ds_swizzle-free path (bpermute); SW=0 same (placeholder)
__global__ __launch_bounds__(1024) void gather_lds2(int64_t nnz, int per_wave, int slab_rows,
                                                    const uint32_t* __restrict__ ent,
                                                    const double* __restrict__ X, double* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* S = reinterpret_cast<double*>(smem);
    constexpr int K = 16;
    constexpr int KP = VW == 2 ? 16 : 8;   // lanes per entry, each reading 16 B
    constexpr int G = 64 / KP;
    const int n_el = slab_rows * K * VW;
    for (int i = threadIdx.x; i < n_el; i += blockDim.x) S[i] = X[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    if (b >= nnz) return;
    const int g = lane / KP, kl = lane % KP;
    double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    uint32_t e = __builtin_nontemporal_load(ent + b + lane);
    for (int off = 0; off < per_wave; off += 64) {
        const uint32_t ec = e;
        if (off + 64 < per_wave) e = __builtin_nontemporal_load(ent + b + off + 64 + lane);
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            const uint32_t w = (uint32_t)__shfl((int)ec, u * G + g, 64);
            const uint32_t r = w >> 12;
            const double a = (double)(int)((w >> 6) & 63u), d = (double)(int)(w & 63u);
            const double2 x = reinterpret_cast<const double2*>(S)[r * (K * VW / 2) + kl];
            if (VW == 2) { a1 += a * x.x + d * x.y; }
            else { a1 += a * x.x; a2 += d * x.x; a3 += a * x.y; a4 += d * x.y; }
        }
    }
    out[wave * 64 + lane] = a1 + a2 + a3 + a4;
}

// ---- variant G2: global gather with packed 32-bit entries (one permute), 16 B per lane
template <int VW>
__global__ __launch_bounds__(256) void gather_global2(int64_t nnz, int per_wave,
                                                      const uint32_t* __restrict__ ent,
                                                      const double* __restrict__ X, double* __restrict__ out) {
    constexpr int K = 16;
    constexpr int KP = VW == 2 ? 16 : 8;
    constexpr int G = 64 / KP;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t b = wave * per_wave;
    if (b >= nnz) return;
    const int g = lane / KP, kl = lane % KP;
    double a1 = 0, a2 = 0, a3 = 0, a4 = 0;
    uint32_t e = __builtin_nontemporal_load(ent + b + lane);
    for (int off = 0; off < per_wave; off += 64) {
        const uint32_t ec = e;
        if (off + 64 < per_wave) e = __builtin_nontemporal_load(ent + b + off + 64 + lane);
        double2 x[KP];
        uint32_t w[KP];
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            w[u] = (uint32_t)__shfl((int)ec, u * G + g, 64);
            x[u] = reinterpret_cast<const double2*>(X)[(w[u] >> 12) * (K * VW / 2) + kl];
        }
#pragma unroll
        for (int u = 0; u < KP; ++u) {
            const double a = (double)(int)((w[u] >> 6) & 63u), d = (double)(int)(w[u] & 63u);
            if (VW == 2) { a1 += a * x[u].x + d * x[u].y; }
            else { a1 += a * x[u].x; a2 += d * x[u].x; a3 += a * x[u].y; a4 += d * x[u].y; }
        }
    }
    out[wave * 64 + lane] = a1 + a2 + a3 + a4;
}

template <typename F>
float time_it(F&& launch, int reps = 5) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char** argv) {
    const int64_t nnz = 1 << 27;  // 134M entries
    const int K = 16;
    std::vector<int> rows_list = {6250, 25000, 100000};
    int* idx; long long* val; double* X; double* out;
    CK(hipMalloc(&idx, nnz * 4)); CK(hipMalloc(&val, nnz * 8));
    CK(hipMalloc(&X, (size_t)100000 * K * 16)); CK(hipMalloc(&out, nnz + 65536));
    std::vector<double> hx((size_t)100000 * K * 2);
    for (auto& v : hx) v = 0.5;
    CK(hipMemcpy(X, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
    std::vector<long long> hv(nnz, (1ll << 32) | 1);
    CK(hipMemcpy(val, hv.data(), nnz * 8, hipMemcpyHostToDevice));
    std::vector<int> hi(nnz);
    std::mt19937 rng(1);
    const int per_wave_list[] = {1024, 2048};
    {
        const int pw = 2048; const int64_t nw = nnz / pw;
        float ms = time_it([&] { stream_only<<<nw / 4, 256>>>(nnz, pw, idx, val, out); });
        printf("stream_only                         : %.3f ms  (%.0f GB/s)\n", ms, nnz * 12.0 / ms / 1e6);
    }
    for (int rows : rows_list) {
        for (auto& v : hi) v = (int)(rng() % (uint32_t)rows);
        CK(hipMemcpy(idx, hi.data(), nnz * 4, hipMemcpyHostToDevice));
        for (int pw : per_wave_list) {
            const int64_t nw = nnz / pw;
#define RUN(VW, UN, MW)                                                                          \
    {                                                                                            \
        float ms = time_it([&] { gather_global<VW, UN, MW><<<nw / 4, 256>>>(nnz, pw, idx, val, X, K, out); }); \
        printf("global rows=%6d pw=%4d VW=%d UN=%2d minw=%d : %.3f ms  %.1f Gentries/s  gather %.1f TB/s\n", rows, pw, \
               VW, UN, MW, ms, nnz / ms / 1e6, nnz * (VW * 128.0) / ms / 1e9);                   \
    }
            RUN(1, 16, 1) RUN(1, 16, 4) RUN(1, 8, 1) RUN(1, 8, 8) RUN(1, 4, 8)
            RUN(2, 16, 1) RUN(2, 16, 4) RUN(2, 8, 1) RUN(2, 8, 6) RUN(2, 4, 8)
        }
    }
    // LDS variant: slab of 512 rows (128 KB at VW=2, 64 KB at VW=1)
    for (auto& v : hi) v = (int)(rng() % 512u);
    CK(hipMemcpy(idx, hi.data(), nnz * 4, hipMemcpyHostToDevice));
    {
        const int pw = 8192; const int64_t nw = nnz / pw;
        CK(hipFuncSetAttribute((const void*)gather_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        CK(hipFuncSetAttribute((const void*)gather_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int threads : {512, 1024}) {
            float ms = time_it([&] { gather_lds<2><<<nw / (threads / 64), threads, 512 * 16 * 16>>>(nnz, pw, 512, idx, val, X, out); });
            printf("lds VW=2 slab=512 rows threads=%4d   : %.3f ms  %.1f Gentries/s\n", threads, ms, nnz / ms / 1e6);
            ms = time_it([&] { gather_lds<1><<<nw / (threads / 64), threads, 1024 * 16 * 8>>>(nnz, pw, 1024, idx, val, X, out); });
            printf("lds VW=1 slab=1024 rows threads=%4d  : %.3f ms  %.1f Gentries/s\n", threads, ms, nnz / ms / 1e6);
        }
    }

    // packed-entry variants
    {
        uint32_t* ent; CK(hipMalloc(&ent, nnz * 4));
        std::vector<uint32_t> he(nnz);
        for (int rows : {512, 1024, 6250, 25000, 100000}) {
            for (auto& v : he) v = ((uint32_t)(rng() % (uint32_t)rows) << 12) | (1u << 6) | 2u;
            CK(hipMemcpy(ent, he.data(), nnz * 4, hipMemcpyHostToDevice));
            const int pw = 2048; const int64_t nw = nnz / pw;
            if (rows >= 6250) {
                float ms = time_it([&] { gather_global2<2><<<nw / 4, 256>>>(nnz, pw, ent, X, out); });
                printf("global2 packed VW=2 (16 lanes x16B) rows=%6d : %.3f ms  %.1f Gentries/s\n", rows, ms, nnz / ms / 1e6);
                ms = time_it([&] { gather_global2<1><<<nw / 4, 256>>>(nnz, pw, ent, X, out); });
                printf("global2 packed VW=1 ( 8 lanes x16B) rows=%6d : %.3f ms  %.1f Gentries/s\n", rows, ms, nnz / ms / 1e6);
            } else {
                const int pw2 = 8192; const int64_t nw2 = nnz / pw2;
                CK(hipFuncSetAttribute((const void*)gather_lds2<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                CK(hipFuncSetAttribute((const void*)gather_lds2<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                for (int threads : {512, 1024}) {
                    if (rows == 512) {
                        float ms = time_it([&] { gather_lds2<2, 0><<<nw2 / (threads / 64), threads, 512 * 256>>>(nnz, pw2, 512, ent, X, out); });
                        printf("lds2 packed VW=2 slab=512  threads=%4d : %.3f ms  %.1f Gentries/s\n", threads, ms, nnz / ms / 1e6);
                    } else {
                        float ms = time_it([&] { gather_lds2<1, 0><<<nw2 / (threads / 64), threads, 1024 * 128>>>(nnz, pw2, 1024, ent, X, out); });
                        printf("lds2 packed VW=1 slab=1024 threads=%4d : %.3f ms  %.1f Gentries/s\n", threads, ms, nnz / ms / 1e6);
                    }
                }
            }
        }
    }
    return 0;
}
