// Scratch-build probes of vrx_spmm_lds (-DVRX_PROBE_BUILD; never part of the product build).
// Every wave times the bracketed statements with s_memtime and writes ONE record of its own
// (no atomics: they would serialise the ends of the workgroups):
//   [0] start  [1] end (before the output stores)  [2] barrier 1  [3] barrier 2  [4] slab store
//   [5] waits of the walk for its stream (ring_event)  [6] end of the first barrier 2
//   [7] xcc_id << 32 | hw_id
// s_memtime counters are per XCD: only differences inside one XCD mean anything.
#pragma once
constexpr int VRX_PROBE_MAXW = 16384;
__device__ unsigned long long vrx_probe_rec[2][VRX_PROBE_MAXW][8];
// per-visit log of ONE workgroup (VRX_PROBE_WG): [wave][visit] = (trips start, trips end, trips so far)
#ifndef VRX_PROBE_WG
#define VRX_PROBE_WG 100
#endif
constexpr int VRX_PROBE_VISITS = 128;
__device__ unsigned long long vrx_probe_visit[2][VRX_LDS_WAVES][VRX_PROBE_VISITS][3];
#define VRX_PROBE_BEGIN                                                             \
    unsigned long long tm_bar1 = 0, tm_bar2 = 0, tm_stage = 0, tm_dma = 0, tm_first = 0; \
    int pv_ = 0;                                                                     \
    unsigned long long ptrips_ = 0;                                                  \
    const bool plog_ = blockIdx.x == VRX_PROBE_WG && (threadIdx.x & 63) == 0;        \
    const unsigned long long tm_start = __builtin_amdgcn_s_memtime();
#define VRX_PROBE(var, stmt)                                          \
    {                                                                 \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();  \
        stmt;                                                         \
        const unsigned long long u_ = __builtin_amdgcn_s_memtime();  \
        var += u_ - t_;                                               \
        if (&var == &tm_bar2 && tm_first == 0) tm_first = u_;         \
        if (plog_ && &var == &tm_bar1 && pv_ > 0 && pv_ <= VRX_PROBE_VISITS) {       \
            vrx_probe_visit[MODE][wave][pv_ - 1][1] = t_;             \
            vrx_probe_visit[MODE][wave][pv_ - 1][2] = ptrips_;        \
        }                                                             \
        if (plog_ && &var == &tm_bar2 && pv_ < VRX_PROBE_VISITS) {    \
            vrx_probe_visit[MODE][wave][pv_][0] = u_;                 \
            ++pv_;                                                    \
        }                                                             \
    }
#define VRX_PROBE_END                                                                  \
    {                                                                                  \
        const unsigned long long tm_end = __builtin_amdgcn_s_memtime();               \
        const int w_ = (blockIdx.y * gridDim.x + blockIdx.x) * VRX_LDS_WAVES + wave;   \
        if (plog_ && pv_ > 0 && pv_ <= VRX_PROBE_VISITS) {                             \
            vrx_probe_visit[MODE][wave][pv_ - 1][1] = tm_end;                          \
            vrx_probe_visit[MODE][wave][pv_ - 1][2] = ptrips_;                         \
        }                                                                              \
        if (lane == 0 && w_ < VRX_PROBE_MAXW) {                                        \
            unsigned hw_, xcc_;                                                        \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));          \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));        \
            unsigned long long* r_ = vrx_probe_rec[MODE][w_];                          \
            r_[0] = tm_start, r_[1] = tm_end, r_[2] = tm_bar1, r_[3] = tm_bar2;        \
            r_[4] = tm_stage, r_[5] = tm_dma, r_[6] = tm_first;                        \
            r_[7] = ((unsigned long long)xcc_ << 32) | hw_;                            \
        }                                                                              \
    }
#define VRX_PROBE_TRIP ++ptrips_;
