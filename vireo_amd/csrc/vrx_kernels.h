// Device code of libvireo_hip.so -- hand-written HIP for gfx950 (MI355X / CDNA4).
//
// The per-iteration work of the reference (vireoSNP/utils/vireo_model.py:251-264) is
//   13 scipy SpMMs + 2 sparse subtractions + dense softmaxes + an ELBO reduction.
// Here it is TWO streaming passes over the (ad,dp) matrix plus a few dense kernels:
//
//   variant pass  S[n,k]   = ( sum_m ad*ID[m,k] , sum_m dp*ID[m,k] )
//   cell pass     LID[m,k] =   sum_n ad*W1[n,k] + dp*W2[n,k]
//        with  W1 = sum_t GT[n,k,t](psi1_t - psi2_t),  W2 = sum_t GT[n,k,t](psi2_t - psis_t)
//        (AD^T(GT psi1) + BD^T(GT psi2) - DP^T(GT psis)  regrouped by ad and dp)
//
// Each pass exists twice: vrx_spmm (gathers the dense rows from global memory; any K, any
// counts; bound by the L1-miss path) and vrx_spmm_lds (streams the dense operand through
// LDS; large problems, K a multiple of 4 up to 16).  The passes are integer streams (4-12 B
// per non-zero) with an fp64 row read per non-zero; nothing here is GEMM-shaped enough for
// MFMA (the K x T contraction is a length-3 dot product, fp64 MFMA runs at the vector rate).
// All arithmetic is fp64; all reductions are fixed-order (no atomics), so results are
// run-to-run deterministic.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

constexpr int VRX_BLOCK = 256;  // 4 wavefronts
constexpr int VRX_WAVES = VRX_BLOCK / 64;
constexpr int VRX_MAXT = 8;  // max genotype classes handled by the dense kernels

// ------------------------------------------------------------------------------------
// Device-side loop control.  The host enqueues several iterations at a time; the stop rule of
// _fit_VB (vireo_model.py:266-274) is evaluated on the device by the kernel that forms the
// ELBO, and once it fires every later kernel of the batch returns at once.
//   ctl[0] stop flag   ctl[1] iteration the loop stopped at   ctl[2] warn flags (1: lower
//   bound decreased, 2: not converged)
// (Measured and not adopted: running the one-block final reductions inside the last block of
//  their producer kernel.  The agent-scope release fence every producer block then needs costs
//  more than the two small launches it saves: dense kernels 0.136 -> 0.223 ms per iteration at
//  c3, 41 -> 49 us per iteration at c2.)
// ------------------------------------------------------------------------------------
enum { VRX_CTL_STOP = 0, VRX_CTL_IT = 1, VRX_CTL_WARN = 2, VRX_CTL_WORDS = 4 };  // per restart
// PRECONDITION (also of wave_sum and the vrx_group_* DPP reductions below): called with ALL 64
// lanes of the wave active and converged -- the R <= 64 form reads restart r's word on lane r
// and ballots, and DPP moves read 0 from inactive lanes (bound_ctrl).  Every call site is at kernel
// entry / outside divergent control flow; a call behind a divergent early return would report
// "all stopped" for restarts whose lanes have left.
__device__ __forceinline__ bool vrx_all_stopped(const int32_t* ctl, int R) {
    if (R == 1) return ctl[VRX_CTL_STOP] != 0;
    if (R <= 64) {  // lane r reads restart r's word: one round trip for the whole batch
        const int r = (int)(threadIdx.x & 63);
        const int v = r < R ? ctl[r * VRX_CTL_WORDS + VRX_CTL_STOP] : 1;
        return __builtin_amdgcn_ballot_w64(v == 0) == 0;
    }
    bool all = true;
    for (int r = 0; r < R; ++r) all = all && ctl[r * VRX_CTL_WORDS + VRX_CTL_STOP] != 0;
    return all;
}
// Restart batches.  A model may hold R restarts of the same problem (vireo_wrap.py:64-87 runs
// them one after the other): the dense operands then carry R * K columns -- ID_prob [M][R][K],
// GT_prob [N][R][K][T], S / W [N][R * K] -- so ONE sparse pass serves every restart of the batch
// (the entry stream is read once for R * K columns), and every dense kernel runs once with
// blockIdx.y = restart.  Each restart has its own control words, theta, partial sums and trace.
struct VrxBatch {  // by value
    int R, K, Kt;  // restarts, donors per restart, R * K
};
// position of (variant or cell n, restart r, donor k) given i = n * K + k
__device__ __forceinline__ int64_t vrx_col(const VrxBatch& b, int64_t i, int r) {
    if (b.R == 1) return i;
    const int64_t n = i / b.K;
    return n * b.Kt + (int64_t)r * b.K + (i - n * b.K);
}

struct VrxStopRule {  // by value to the ELBO kernel
    int it, min_iter, max_iter, active;
    double eps;
};

// ------------------------------------------------------------------------------------
// wave / block reductions (fixed butterfly order => deterministic)
// ------------------------------------------------------------------------------------
typedef int vrx_i2 __attribute__((ext_vector_type(2)));
// value of lane ^ S for the steps DPP can express EXACTLY -- 1, 2: quad permutations; 8: rotation
// of the 16-lane row by 8; 4: lane 7 - l of the half row (= l ^ 7), then the quad reversed
// (^ 3) -- VALU moves instead of a trip through the LDS crossbar (ds_bpermute).  16 and 32
// cross rows: __shfl_xor.
template <int S>
__device__ __forceinline__ double vrx_lane_xor_exact(double v) {
    if constexpr (S >= 16) {
        return __shfl_xor(v, S, 64);
    } else {
        constexpr int ctrl = S == 1 ? 0xB1 : S == 2 ? 0x4E : S == 4 ? 0x141 : 0x128;
        const vrx_i2 b = __builtin_bit_cast(vrx_i2, v);
        vrx_i2 r;
        r.x = __builtin_amdgcn_update_dpp(0, b.x, ctrl, 0xf, 0xf, true);
        r.y = __builtin_amdgcn_update_dpp(0, b.y, ctrl, 0xf, 0xf, true);
        if constexpr (S == 4) {
            r.x = __builtin_amdgcn_update_dpp(0, r.x, 0x1B, 0xf, 0xf, true);  // quad_perm [3,2,1,0]
            r.y = __builtin_amdgcn_update_dpp(0, r.y, 0x1B, 0xf, 0xf, true);
        }
        return __builtin_bit_cast(double, r);
    }
}

__device__ __forceinline__ double wave_sum(double v) {  // butterfly 32, 16, ... 1: the order is part of the results
    v += vrx_lane_xor_exact<32>(v);
    v += vrx_lane_xor_exact<16>(v);
    v += vrx_lane_xor_exact<8>(v);
    v += vrx_lane_xor_exact<4>(v);
    v += vrx_lane_xor_exact<2>(v);
    v += vrx_lane_xor_exact<1>(v);
    return v;
}

// Partner value of an ASCENDING butterfly step (s = 1, 2, 4, ...) by DPP -- a VALU move instead of
// a trip through the LDS crossbar (ds_bpermute).  Steps 1 and 2 are exact lane ^ S permutations
// of a quad; steps 4 and 8 take lane 7 - l / 15 - l of the row, which holds the same value as lane
// l ^ S once the steps below have run (all lanes of an S-group are equal by then) -- so the results
// are bit for bit those of the __shfl_xor butterfly.  S >= 16 crosses rows: __shfl_xor.
template <int S>
__device__ __forceinline__ double vrx_butterfly_partner(double v) {
    if constexpr (S >= 16) {
        return __shfl_xor(v, S, 64);
    } else {
        constexpr int ctrl = S == 1 ? 0xB1 : S == 2 ? 0x4E : S == 4 ? 0x141 : 0x140;
        const vrx_i2 b = __builtin_bit_cast(vrx_i2, v);
        vrx_i2 r;
        r.x = __builtin_amdgcn_update_dpp(0, b.x, ctrl, 0xf, 0xf, true);
        r.y = __builtin_amdgcn_update_dpp(0, b.y, ctrl, 0xf, 0xf, true);
        return __builtin_bit_cast(double, r);
    }
}
// max / sum over the W (a power of two <= 64) lanes of an aligned group, every lane gets the result
template <int W>
__device__ __forceinline__ double vrx_group_max(double v) {
    if constexpr (W > 1) v = fmax(v, vrx_butterfly_partner<1>(v));
    if constexpr (W > 2) v = fmax(v, vrx_butterfly_partner<2>(v));
    if constexpr (W > 4) v = fmax(v, vrx_butterfly_partner<4>(v));
    if constexpr (W > 8) v = fmax(v, vrx_butterfly_partner<8>(v));
    if constexpr (W > 16) v = fmax(v, vrx_butterfly_partner<16>(v));
    if constexpr (W > 32) v = fmax(v, vrx_butterfly_partner<32>(v));
    return v;
}
template <int W>
__device__ __forceinline__ double vrx_group_sum(double v) {
    if constexpr (W > 1) v += vrx_butterfly_partner<1>(v);
    if constexpr (W > 2) v += vrx_butterfly_partner<2>(v);
    if constexpr (W > 4) v += vrx_butterfly_partner<4>(v);
    if constexpr (W > 8) v += vrx_butterfly_partner<8>(v);
    if constexpr (W > 16) v += vrx_butterfly_partner<16>(v);
    if constexpr (W > 32) v += vrx_butterfly_partner<32>(v);
    return v;
}

// Sum NV values per thread over the 256-thread block; thread 0 writes out[0..NV).  `period`,
// `live`: only the values with i % period < live are summed (the others are written as 0) --
// the theta sums are laid out [2][VRX_MAXT] and T = 3 of the VRX_MAXT classes exist, and a
// wave reduction is 12 cross-lane moves of latency per value.
template <int NV>
__device__ __forceinline__ void block_sum_store(const double (&v)[NV], double* out, int period = NV,
                                                int live = NV) {
    __shared__ double sm[NV * VRX_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (i % period >= live) continue;  // (uniform)
        double r = wave_sum(v[i]);
        if (lane == 0) sm[i * VRX_WAVES + wave] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double r = 0.0;
            if (i % period < live) {
#pragma unroll
                for (int w = 0; w < VRX_WAVES; ++w) r += sm[i * VRX_WAVES + w];
            }
            out[i] = r;
        }
    }
}

// ------------------------------------------------------------------------------------
// special functions
// ------------------------------------------------------------------------------------
// digamma for x > 0: upward recurrence to x >= 10, then the asymptotic series
// ln x - 1/(2x) - sum B_2k / (2k x^2k)  (the classic Cephes scheme scipy.special.digamma
// uses away from its root; absolute accuracy ~1e-15 on the Beta shapes met here).
__device__ __forceinline__ double vrx_digamma(double x) {
    if (!(x > 0.0)) return __builtin_nan("");
    double w = 0.0;
    while (x < 10.0) {
        w += 1.0 / x;
        x += 1.0;
    }
    const double z = 1.0 / (x * x);
    double y = 8.33333333333333333333e-2;
    y = y * z - 2.10927960927960927961e-2;
    y = y * z + 7.57575757575757575758e-3;
    y = y * z - 4.16666666666666666667e-3;
    y = y * z + 3.96825396825396825397e-3;
    y = y * z - 8.33333333333333333333e-3;
    y = y * z + 8.33333333333333333333e-2;
    return log(x) - 0.5 / x - y * z - w;
}

__device__ __forceinline__ double vrx_betaln(double a, double b) {
    return lgamma(a) + lgamma(b) - lgamma(a + b);
}

// KL( Beta(p1,p2) || Beta(q1,q2) ), term order of vireoSNP/utils/vireo_base.py:96-125
// (cross(p,q) - cross(p,p)); d1,d2,ds are digamma(p1), digamma(p2), digamma(p1+p2).
__device__ __forceinline__ double vrx_beta_kl(double p1, double p2, double q1, double q2,
                                              double d1, double d2, double ds) {
    const double cq = vrx_betaln(q1, q2) - (q1 - 1.0) * d1 - (q2 - 1.0) * d2 +
                      ((q1 + q2) - 2.0) * ds;
    const double cp = vrx_betaln(p1, p2) - (p1 - 1.0) * d1 - (p2 - 1.0) * d2 +
                      ((p1 + p2) - 2.0) * ds;
    return cq - cp;
}

// ------------------------------------------------------------------------------------
// the sparse passes
// ------------------------------------------------------------------------------------
// Entry formats (chosen per orientation when the problem is uploaded):
//   VRX_FMT_P32   one 32-bit word   index:20 | ad:6 | dp:6      ( 4 B / non-zero)
//   VRX_FMT_P64   two words         index:32 , ad:16 | dp:16    ( 8 B / non-zero)
//   VRX_FMT_WIDE  three words       index , ad , dp             (12 B / non-zero)
// Narrower formats cut the HBM stream and, more importantly, the cross-lane permutes
// per step (1 / 2 / 3).
enum { VRX_FMT_P32 = 0, VRX_FMT_P64 = 1, VRX_FMT_WIDE = 2 };

template <int FMT>
struct VrxWords {
    uint32_t w[FMT + 1];
};

template <int FMT>
__device__ __forceinline__ VrxWords<FMT> vrx_load_words(const uint32_t* __restrict__ ent,
                                                        int64_t at, bool ok) {
    VrxWords<FMT> e;
#pragma unroll
    for (int i = 0; i <= FMT; ++i) e.w[i] = 0;  // padded entries: index 0, ad = dp = 0
    if (ok) {  // streamed once per pass: non-temporal, keep L2 for the dense rows
        if (FMT == VRX_FMT_P32) {
            e.w[0] = __builtin_nontemporal_load(ent + at);
        } else if (FMT == VRX_FMT_P64) {
            const unsigned long long v =
                __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(ent) + at);
            e.w[0] = (uint32_t)v;
            e.w[1] = (uint32_t)(v >> 32);
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) e.w[i] = __builtin_nontemporal_load(ent + at * 3 + i);
        }
    }
    return e;
}

template <int FMT>
__device__ __forceinline__ void vrx_unpack(const VrxWords<FMT>& e, uint32_t& id, int& ad, int& dp) {
    if (FMT == VRX_FMT_P32) {
        id = e.w[0] >> 12;
        ad = (int)((e.w[0] >> 6) & 63u);
        dp = (int)(e.w[0] & 63u);
    } else if (FMT == VRX_FMT_P64) {
        id = e.w[0];
        ad = (int)(e.w[FMT >= 1 ? 1 : 0] & 0xffffu);
        dp = (int)(e.w[FMT >= 1 ? 1 : 0] >> 16);
    } else {
        id = e.w[0];
        ad = (int)e.w[FMT >= 1 ? 1 : 0];
        dp = (int)e.w[FMT >= 2 ? 2 : 0];
    }
}

// One wavefront per segment.  The wave reads 64 entries with one coalesced load (prefetched
// one batch ahead), then walks them G = 64/LPE at a time: an LPE-lane group takes one entry
// and each lane of the group owns CPL adjacent dense columns, so the gather of one dense row
// is a single contiguous read of 16 B per lane wherever the layout allows:
//   MODE 0 (variant pass): X = ID_prob (rows x K doubles), CPL = 2 for even K (double2 per
//           lane) else 1;  out = S[row][k] = double2 (sum ad*x, sum dp*x)
//   MODE 1 (cell pass)   : X = W (rows x K double2 (w1,w2)), CPL = 1;
//           out = LID[row][k] = sum ad*w1 + dp*w2
// Entries reach the groups through cross-lane permutes, never through memory.  The L1-miss
// path (one 128-B line every ~3.6 clk per CU) bounds this kernel, which is why the segment
// table is tiled over the contracted dimension and launched XCD-aware (vrx_engine.hip).
// LPE*CPL <= 16 columns per block; wider K is covered by blockIdx.y column chunks.
template <int LPE, int CPL, int MODE, int FMT, bool TAIL>
__device__ __forceinline__ void vrx_spmm_batch(const VrxWords<FMT>& e, int nh, int g, uint32_t kc,
                                               uint32_t K, const double* __restrict__ X,
                                               double (&a1)[CPL], double (&a2)[CPL]) {
    constexpr int G = 64 / LPE;
    double x0[LPE][CPL], x1[MODE == 1 ? LPE : 1][CPL];
    int ads[LPE], dps[LPE];
#pragma unroll
    for (int u = 0; u < LPE; ++u) {
        if (TAIL && u * G >= nh) break;  // wave-uniform
        VrxWords<FMT> q;
#pragma unroll
        for (int i = 0; i <= FMT; ++i) q.w[i] = (uint32_t)__shfl((int)e.w[i], u * G + g, 64);
        uint32_t r;
        vrx_unpack<FMT>(q, r, ads[u], dps[u]);
        if (MODE == 0) {
            if (CPL == 2) {
                const double2 v = *reinterpret_cast<const double2*>(X + (r * K + kc));
                x0[u][0] = v.x;
                x0[u][CPL - 1] = v.y;
            } else {
                x0[u][0] = X[r * K + kc];
            }
        } else {
            const double2 v = reinterpret_cast<const double2*>(X)[r * K + kc];
            x0[u][0] = v.x;
            x1[u][0] = v.y;
        }
    }
#pragma unroll
    for (int u = 0; u < LPE; ++u) {
        if (TAIL && u * G >= nh) break;
        const double ad = (double)ads[u], dp = (double)dps[u];
        const bool live = !TAIL || (u * G + g < nh);  // 0 * (inf|nan) must not leak in
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (MODE == 0) {
                a1[c] += live ? ad * x0[u][c] : 0.0;
                a2[c] += live ? dp * x0[u][c] : 0.0;
            } else {
                a1[c] += live ? ad * x0[u][c] + dp * x1[u][c] : 0.0;
            }
        }
    }
}

// FUSE = 1 (cell pass of ONE restart, K <= 16, every row a single segment): the wave that holds a
// cell's complete logLik_ID row also takes its softmax and the cell's ELBO terms -- vrx_cell_softmax's
// arithmetic on the LPE lanes of group 0 -- so a small problem's iteration is one launch shorter
// (each launch costs it ~4-5 us) and logLik_ID is not re-read.
struct VrxCellFuse {  // by value
    const double* logq;  // id_mode 1: one row of K; 2: (M, K)
    int id_mode;
    double logq_uni;
    double* ID;
    double* part;  // [gridDim.x][2]: (sum L p, sum p (log p - log q)) of the block's cells
};

template <int LPE, int CPL, int MODE, int FMT, int FUSE = 0>
__global__ __launch_bounds__(VRX_BLOCK) void vrx_spmm(
    int64_t n_seg, const int64_t* __restrict__ seg_begin, const int32_t* __restrict__ seg_len,
    const int32_t* __restrict__ seg_dst, const uint32_t* __restrict__ ent,
    const double* __restrict__ X, int K, double* __restrict__ out, double* __restrict__ partial,
    const int32_t* __restrict__ ctl, int n_batch, VrxCellFuse F) {
    static_assert(LPE * CPL <= 16 && (MODE == 0 || CPL == 1), "layout");
    static_assert(FUSE == 0 || (MODE == 1 && CPL == 1), "the fused epilogue is the cell pass's");
    // (the stop words and the segment record are read TOGETHER: a kernel of a small problem is a
    //  chain of dependent memory round trips of ~1 us each, and an early return on the stop
    //  word alone would put one more in front of every launch)
    const int lane = threadIdx.x & 63;
    const int64_t seg = (int64_t)blockIdx.x * VRX_WAVES + (threadIdx.x >> 6);
    const bool in_range = seg < n_seg;
    const int len_raw = in_range ? seg_len[seg] : -1;
    const int64_t b = (in_range ? seg_begin[seg] : 0) + lane;
    const int d = in_range ? seg_dst[seg] : 0;
    const bool stopped = vrx_all_stopped(ctl, n_batch);
    // (len < 0: padding of the XCD-aware launch order; the fused epilogue ends in a block sum,
    //  so its idle waves stay until then)
    if (stopped || (FUSE == 0 && len_raw < 0)) return;
    const bool act = len_raw >= 0;
    const int len = act ? len_raw : 0;
    const int g = lane / LPE, kl = lane % LPE;
    const int k = (blockIdx.y * LPE + kl) * CPL;
    const bool kok = k < K;
    const uint32_t kc = kok ? k : K - CPL;  // padded lanes re-read the last columns, never store
    double a1[CPL], a2[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) a1[c] = a2[c] = 0.0;
    VrxWords<FMT> nxt = vrx_load_words<FMT>(ent, b, lane < len);
    int off = 0;
    for (; off + 64 <= len; off += 64) {
        const VrxWords<FMT> cur = nxt;
        nxt = vrx_load_words<FMT>(ent, b + off + 64, off + 64 + lane < len);
        vrx_spmm_batch<LPE, CPL, MODE, FMT, false>(cur, 64, g, kc, (uint32_t)K, X, a1, a2);
    }
    if (off < len)
        vrx_spmm_batch<LPE, CPL, MODE, FMT, true>(nxt, len - off, g, kc, (uint32_t)K, X, a1, a2);
    auto fold = [&](auto s_tag) {  // (steps 1 ... 8 by DPP: vrx_lane_xor_exact)
        constexpr int S = decltype(s_tag)::value;
        if constexpr (S >= LPE) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                a1[c] += vrx_lane_xor_exact<S>(a1[c]);
                if (MODE == 0) a2[c] += vrx_lane_xor_exact<S>(a2[c]);
            }
        }
    };
    fold(std::integral_constant<int, 1>());
    fold(std::integral_constant<int, 2>());
    fold(std::integral_constant<int, 4>());
    fold(std::integral_constant<int, 8>());
    fold(std::integral_constant<int, 16>());
    fold(std::integral_constant<int, 32>());
    const bool own = act && g == 0 && kok;
    if (own) {
        double* base = d >= 0 ? out : partial;
        const int64_t row = d >= 0 ? d : -(int64_t)d - 1;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            if (MODE == 0)
                reinterpret_cast<double2*>(base)[row * K + k + c] = make_double2(a1[c], a2[c]);
            else
                base[row * K + k + c] = a1[c];
        }
    }
    if (FUSE == 1) {  // (d >= 0 for every segment: the host fuses only when no row is split)
        const double L = a1[0];
        const double lq = !own || F.id_mode == 0 ? F.logq_uni
                                                 : (F.id_mode == 2 ? F.logq[(int64_t)d * K + k] : F.logq[k]);
        const double x0 = own ? L + lq : -__builtin_inf();
        double mx = x0;
        mx = vrx_group_max<LPE>(mx);
        const double e0 = own ? exp(x0 - mx) : 0.0;
        double sum = e0;
        sum = vrx_group_sum<LPE>(sum);
        double acc[2] = {0.0, 0.0};
        if (own) {
            const double x = x0 - mx;
            const double p = e0 / sum;
            const double lp = x - log(sum);
            F.ID[(int64_t)d * K + k] = p;
            acc[0] = L * p;
            if (p > 0.0) acc[1] = p * (lp - lq);
        }
        block_sum_store<2>(acc, F.part + (int64_t)blockIdx.x * 2);
    }
}

// ------------------------------------------------------------------------------------
// LDS-resident variant of the sparse passes (large problems, counts < 2048; K <= 16 columns
// per launch, wider operands in column blocks)
// ------------------------------------------------------------------------------------
// The global-gather kernel above is bound by the L1-miss path: every non-zero pulls one or
// two 128-B lines through the vector cache.  Here the dense operand is streamed through
// LDS instead (ds_read_b128: 4 clk per KiB against ~29 clk for a gathering
// global_load_dwordx4), which needs BOTH dimensions tiled:
//   * a workgroup (16 waves) owns a tile of 16*RW output rows; wave w owns RW of them;
//   * the contracted dimension is cut into slabs of `slab_rows` dense rows (<= 128 KiB);
//     the workgroup walks the slabs of its range in order, staging each slab into LDS with
//     coalesced loads issued one slab ahead (register prefetch);
//   * inside a wave, LPE = 4 lanes form a group that owns ONE output row at a time
//     (G = 64/LPE rows per round, RW/G rounds) and each lane owns 4 columns of it, so the
//     per-entry overhead (unpack, convert, address) is paid by 4 lanes instead of 16 and a
//     row's sums never leave its lanes: no cross-lane traffic at all;
//   * entries are one 32-bit word each (slab-local index:10 | ad:11 | dp:11), stored in
//     tiled order (tile, wave, slab, round) and TRIP-MAJOR inside a round: word j*G + g is
//     the j-th entry of group g's row, zero words padding every row to the round's longest
//     (the stream is rounded up to whole trips of U entries per row; the zero words of the
//     last trip are skipped).  Every wave thus reads ONE contiguous stream in lock-step with its
//     compute, landing in a 512-word LDS ring in 256-word chunks by LDS-DMA
//     (global_load_lds_dwordx4), one chunk ahead of the walk; bnd[] holds, per wave, the
//     stream offset of every (slab, round) and, in its low bits, the entries of the round's
//     last trip.  With ~10 entries per row and slab the padding costs ~1.6 slots per entry,
//     which the 4x lower per-slot instruction count more than repays;
//   * the 4 columns of a lane are visited in a group-dependent rotation so that the
//     16 lanes serviced together by ds_read_b128 hit 16 different 16-B bank slots.
// Output: one partial array per contracted range (summed in fixed order afterwards).
typedef double vrx_d2 __attribute__((ext_vector_type(2)));
typedef uint32_t vrx_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t vrx_u4 __attribute__((ext_vector_type(4)));
#ifndef VRX_LDS_RING
#define VRX_LDS_RING 512
#endif
constexpr int VRX_RING = VRX_LDS_RING;   // entries per wave
constexpr int VRX_CHUNK = 256;  // entries per refill (64 lanes x 16 B of one LDS-DMA load)
#ifndef VRX_LDS_U_DEF
#define VRX_LDS_U_DEF 4
#endif
// waves per workgroup of the LDS-resident passes (16: one 160-KiB workgroup per CU; 8: two
// 80-KiB workgroups per CU whose barrier / staging phases overlap -- experimental builds)
#ifndef VRX_LDS_WAVES
#define VRX_LDS_WAVES 16
#endif
#ifndef VRX_LDS_PRECISE
#define VRX_LDS_PRECISE 1
#endif
// LDS of a pass = one 2-KiB entry ring per wave + the slab; a slab is staged through PF 16-B
// registers per thread: 16 waves: 32 KiB + 8 x 16 KiB = 160 KiB
#ifdef VRX_LDS_PF_DEF
constexpr int VRX_LDS_PF = VRX_LDS_PF_DEF;
#else
constexpr int VRX_LDS_PF = (160 * 1024 - VRX_LDS_WAVES * VRX_RING * 4) / (VRX_LDS_WAVES * 64 * 16);
#endif
constexpr int VRX_LDS_SLAB_BYTES = VRX_LDS_PF * VRX_LDS_WAVES * 64 * 16;
// Word order inside a trip (U entries for each of the G lane groups).  Pair words (FORM 0):
// entry-major, word j of group g at j * G + g.  AD/BD words (FORM 1, 2): group-major, the U
// words of a group adjacent (g * U + u), so that a lane takes its trip with ONE ds_read_b128
// instead of two ds_read2_b32.  Both builders place the words with vrx_trip_slot.
// stream position, relative to the round's base, of entry n (0, 1, ...) of lane group g
__host__ __device__ inline int64_t vrx_trip_slot(int64_t n, int g, int G, int U, int form) {
    if (form != 0) return (n / U) * ((int64_t)U * G) + (int64_t)g * U + n % U;
    return n * G + g;
}
// output rows per wave (tile = 16 x this), per pass: the tallest tile the 128 registers of a
// 1024-thread workgroup hold without spilling (accumulators: RW / 2 registers in the AD/BD cell
// pass, RW in the AD/BD variant pass and the pair-word cell pass) -- every slab is staged once
// per tile, so a taller tile means less staging per entry (c3 cell pass: 0.389 / 0.362 / 0.349 /
// 0.341 ms at 48 / 64 / 80 / 96 rows)
#ifndef VRX_LDS_LPE_DEF
#define VRX_LDS_LPE_DEF 4
#define VRX_LDS_RWV_DEF 32
#define VRX_LDS_RWC_DEF 96
#endif
constexpr int VRX_LDS_RW_VARIANT = VRX_LDS_RWV_DEF, VRX_LDS_RW_CELL = VRX_LDS_RWC_DEF;
constexpr int VRX_LDS_RW_CELL_PAIR = 64;  // cell pass on (ad, dp) pair words (FORM 0): twice the accumulators
constexpr int VRX_LDS_RW_CELL_SHORT = VRX_LDS_LPE_DEF == 1 ? 64 : 32;  // cell pass with one or two slabs, see vrx_problem_create
constexpr int VRX_LDS_LPE = VRX_LDS_LPE_DEF;  // lanes per output row (16 / this columns per lane)
constexpr int VRX_LDS_U = VRX_LDS_U_DEF;    // entries per trip and group; rows are padded to it

// PADK: K is not a multiple of 4 (the slab is staged element-wise into zero-padded rows).
// SPLIT: with K <= 8 (<= 4) a row needs only 2 (1) of the group's 4 lanes, so the group's lanes
// take SPLIT = 2 (4) consecutive entries of the row at once, each into its own partial sums;
// the partial sums are added across the lanes once, after the last slab (fixed order).
//
// FORM 1 (cell pass, K = 16): the (ad, dp) pair of an entry is split into SINGLE-VALUED entries
// of the two matrices the reference multiplies with (AD and BD = DP - AD,
// vireo_model.py:190-196): LID = AD^T Wa + BD^T Wb with Wa = W1 + W2, Wb = W2.  On sequencing
// data most entries have ad = 0 or ad = dp, so the split stream is only ~1.14x longer while
// every entry reads 128 B of LDS instead of 256 B and costs 4 FMAs per lane instead of 8.
// The dense operand is PLANAR: row n = [Wa[n][0..16) | Wb[n][0..16)] (256 B), an entry names
// one half.  Word = top 14 bits of the value's IEEE double (sign, exponent, two mantissa bits:
// every integer below 8 and m * 2^e with m < 8; other values are sums of such entries) | LDS
// byte address of the half row:18 (slab base included, a multiple of 128): the value needs no
// conversion (high dword = word & 0xfffc0000, low dword 0) and the LDS address of a slice is
// one v_and_or_b32.  The two lane groups of a ds_read_b128 service
// group that share a slice rotation read different halves whenever one walks AD entries and the
// other BD entries, which the stream builder arranges (AD-first / BD-first segments).
//
// FORM 2 (variant pass): the same single-valued words against the 128-B rows of ID_prob, in
// two PHASES per round -- first the AD entries of the round's rows, accumulated into S1 = AD @ ID,
// then their BD entries, accumulated into S2 = BD @ ID (SS = S1 + S2 at the store); each phase is
// padded to its own longest row.  ~22 % more slots than the (ad, dp) pair words, but 7 instead of
// ~15 vector instructions per slot, no conversions, and counts of any size.
// Probe hooks: the product build defines them away.  A scratch build with -DVRX_PROBE_BUILD includes
// scratch/vrx_probe.h, which times the bracketed statements with s_memtime (DESIGN_HISTORY.md 4.2).
#ifdef VRX_PROBE_BUILD
#include "../../scratch/vrx_probe.h"
#else
#define VRX_PROBE_BEGIN
#define VRX_PROBE(var, stmt) stmt;
#define VRX_PROBE_TRIP
#define VRX_PROBE_END
#endif
// PADK: 0 = rows of exactly 16 contiguous columns (flat slab copy); 1 = any K / row stride
// (element-wise staging into zero-padded rows, masked stores); 2 = AD/BD forms with even K and
// even row stride (column blocks of wider operands, restart batches, K = 2 ... 14): as 1, but
// a 16-B unit is either whole or absent, so it is staged with one load.
template <int LPE, int MODE, int RW, int PADK, int SPLIT, int FORM = 0>
__global__ __launch_bounds__(VRX_LDS_WAVES * 64)
    void vrx_spmm_lds(
    const uint32_t* __restrict__ ent, const int64_t* __restrict__ wave_start,
    const int32_t* __restrict__ bnd, const int32_t* __restrict__ rowmap,
    const int32_t* __restrict__ items, const int32_t* __restrict__ wg_first, int n_slab,
    int slab_rows, int64_t n_contract, int64_t n_rows, const double* __restrict__ X, int K,
    int ld, double* __restrict__ out, const int32_t* __restrict__ ctl, int n_batch,
    const int32_t* __restrict__ perm) {
    // perm != null (balanced slabs, TiledStream::perm; flat AD/BD instances only): slab s of tile t
    // holds the contracted rows perm[(t * n_slab + s) * slab_rows + p], p = its slab-local position
    if (vrx_all_stopped(ctl, n_batch)) return;
    // K <= 16 columns of this launch; ld = columns per row of X and out (ld > K: one block of a
    // wider operand, always with PADK = true: the flat slab copy needs contiguous rows)
    constexpr int G = 64 / LPE;            // rows per round
    constexpr int NR = RW / G;             // rounds
    constexpr int XD = MODE == 1 ? 2 : 1;  // doubles per (contracted row, column)
    constexpr int CP = 16 / LPE;           // dense columns per lane (LPE lanes cover K <= 16)
    constexpr int NQ = FORM == 1 ? CP / 2 : CP * XD / 2;  // 16-B reads per lane per entry
    constexpr int PF = VRX_LDS_PF;         // 16-B prefetch registers per thread (a slab / the workgroup)
    constexpr int NV = MODE == 0 ? 2 : 1;  // accumulated values per column
    constexpr int U = VRX_LDS_U;           // entries per trip and group
    static_assert(RW % G == 0 && 2 * (RW / G) < 63, "rows per wave");
    static_assert(FORM == 0 || (SPLIT == 1 && ((MODE == 1 && FORM == 1) || (MODE == 0 && FORM == 2))),
                  "AD/BD forms");
    constexpr int PH = FORM == 2 ? 2 : 1;  // phases of a round (FORM 2: AD entries, then BD entries)
    constexpr int NRV = NR * PH;           // (round, phase) pairs per slab
    // AD/BD forms staged in whole 16-B units: the slab prefetch is exactly PF vector loads per
    // thread, every one of them unconditional (lanes outside the slab re-read a valid unit; the
    // rows / columns they fill are never referenced by a word resp. never stored), so the walk
    // can leave the prefetch in flight while it waits for a chunk of its stream
    constexpr bool PRECISE = FORM != 0 && VRX_LDS_PRECISE;
    // (element-wise: 2 loads per unit) + the bnd words of the next slab (+ AD/BD instances: the rows the wave
    //  stages for the slab behind it, TiledStream::perm -- issued whether or not the stream is balanced)
    constexpr bool GATHER = PRECISE;
    constexpr int NPF = (PADK == 1 ? 2 : 1) * PF + 1 + (GATHER ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) char vrx_smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // (wave-uniform: scalar)
    // LDS rows are padded to a multiple of CP columns (zeros); FORM 1: two halves of 16 columns
    const int KP = FORM != 0 ? 16 : (K + CP - 1) / CP * CP;  // (whole lanes: CP columns each)
    const int slab_doubles = slab_rows * KP * XD;
    // LDS = [16 entry rings][slab]: the rings first, so that the LDS-DMA destinations stay
    // below 64 KiB
    double* slab = reinterpret_cast<double*>(vrx_smem + VRX_LDS_WAVES * VRX_RING * 4);
    uint32_t* ring = reinterpret_cast<uint32_t*>(vrx_smem) + wave * VRX_RING;
    // One persistent workgroup per CU walks its work items (TiledStream::items): a contiguous run
    // of one tile's slabs each, whose sums go to partial array `slot`.
    const int item_end = wg_first[blockIdx.x + 1];
    VRX_PROBE_BEGIN
    for (int item = wg_first[blockIdx.x]; item < item_end; ++item) {
    const int tile = __builtin_amdgcn_readfirstlane(items[4 * item]);
    const int s_lo = __builtin_amdgcn_readfirstlane(items[4 * item + 1]);
    const int s_hi = __builtin_amdgcn_readfirstlane(items[4 * item + 2]);
    const int slot = __builtin_amdgcn_readfirstlane(items[4 * item + 3]);
    if (s_lo >= s_hi) continue;
    constexpr int LPR = LPE / SPLIT;  // lanes per entry: they cover LPR*CP >= K columns
    constexpr int US = U / SPLIT;     // entries per trip and lane
    static_assert(LPE % SPLIT == 0 && U % SPLIT == 0, "split");
    const int g = lane / LPE, sub = (lane % LPE) / LPR, kl = lane % LPR;
    const bool kok = kl * CP < K;  // a lane's 4 columns may start (or run) past K
    const int64_t wid = (int64_t)tile * VRX_LDS_WAVES + wave;
    const int32_t* bw = bnd + wid * ((int64_t)n_slab * NRV + 1);
    const uint32_t* stream = ent + wave_start[wid];
    // byte offset, inside a dense row, of the q-th 16-B slice this lane reads (rotated by g)
    uint32_t qoff[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) qoff[q] = (kok ? kl : 0) * (NQ * 16) + ((q + g) % NQ) * 16;
    const int row_bytes = KP * XD * 8;
    const uint32_t* ring_g = ring + g + sub * G;  // entries sub, sub + SPLIT, ... of a trip
    static_assert(VRX_RING % (U * G) == 0, "a trip must not wrap the ring");
    static_assert((U & (U - 1)) == 0 && U <= U * G, "tail count lives in the low bits of bnd");

    double acc[NR][NQ][2];  // [round][slice][half]: cell pass (w1,w2)->1 value; variant: 2 cols x2
    double acc2[NR][NQ][2];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int q = 0; q < NQ; ++q) acc[r][q][0] = acc[r][q][1] = acc2[r][q][0] = acc2[r][q][1] = 0.0;

    // ---- slab prefetch (global -> registers), one slab ahead ---------------------------
    // PADK: K columns become KP columns per LDS row (zero filled).  Thread t stages the 16-B
    // unit j0 = t % upr of rows r0 + i*rstep (t below the largest multiple T of upr), so the
    // unit's column never changes and nothing is divided inside the loop.
    vrx_d2 pf[PF];  // (a native vector type: struct copies would keep the array in scratch)
    const int upr = KP * XD / 2;  // 16-B units per LDS row
    constexpr int NT = VRX_LDS_WAVES * 64;  // threads of the workgroup
    const int padT = NT / upr * upr, j0 = threadIdx.x % upr, r0 = threadIdx.x / upr;
    const int rstep = padT / upr;
    const bool pad_act = (int)threadIdx.x < padT;
    auto slab_fetch = [&](int s, int rvec) {
        const int64_t row0 = (int64_t)s * slab_rows;
        const int64_t rows = min((int64_t)slab_rows, n_contract - row0);
        if (!PADK && GATHER && perm) {
            // balanced slabs: a wave's load i covers the four slab-local rows 4 * wave + 64 i + 0..3; their
            // contracted rows sit in lanes 32 + 4 i + 0..3 of the vector rows_load fetched a slab ahead
            const vrx_d2* src = reinterpret_cast<const vrx_d2*>(X);
            int tid_f = threadIdx.x;
            asm volatile("" : "+v"(tid_f));
            const int upr_g = K * XD / 2;  // 16-B units per contracted row (16)
            const int sel = (32 + ((tid_f & 63) >> 4)) << 2, u15 = tid_f & 15;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                // (the list holds a valid row at every position: unused ones name row 0)
                const int rowg = __builtin_amdgcn_ds_bpermute(sel + 16 * i, rvec);
                pf[i] = src[(uint32_t)(rowg * upr_g + u15)];
                // two loads' address temporaries at a time: hoisted together they would cost the walk
                // registers it has not got
                if (i & 1) __builtin_amdgcn_sched_barrier(0);
            }
        } else if (!PADK) {  // rows are contiguous 16-B units: flat copy
            const int n16 = (int)(rows * K * XD / 2);
            const vrx_d2* src = reinterpret_cast<const vrx_d2*>(X + row0 * K * XD);
            // (offsets formed again for every slab from an opaque copy of the thread index: hoisted
            //  out of the walk they would hold registers the walk has not got)
            int tid_f = threadIdx.x;
            asm volatile("" : "+v"(tid_f));
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int at = tid_f + i * NT;
                if (PRECISE)
                    pf[i] = src[(uint32_t)min(at, n16 - 1)];  // (unsigned: scalar base + 32-bit lane offset)
                else
                    pf[i] = at < n16 ? src[at] : vrx_d2{0.0, 0.0};
            }
        } else {
            // 32-bit offsets from one wave-uniform base (scalar base + vector offset loads): the
            // unit's column, its masks and the row stride are per-thread constants
            const double* src = X + row0 * ld * XD;  // (FORM 1: planar rows of 2 * ld doubles)
            const int rows32 = (int)rows;
            // even K and even row stride (column blocks of a wider operand, restart batches,
            // K = 2, 4, ... 14): a unit is either whole or absent and 16-B aligned -> one load
            if (PADK == 2) {
                const int cc = FORM == 1 ? 2 * (j0 & 7) : 2 * j0;
                const bool m0 = cc < K;
                const int off0 = FORM == 1 ? r0 * 2 * ld + (j0 >> 3) * ld + cc : r0 * ld + cc;
                const int step = FORM == 1 ? rstep * 2 * ld : rstep * ld;
                if (PRECISE) {
                    // unconditional loads: rows past the slab's last re-read it, units past K
                    // re-read unit 0 (never referenced resp. never stored).  The offsets are formed
                    // again for every slab from an opaque copy of the thread index: hoisted
                    // out of the walk they would hold PF registers the walk has not got.
                    int tid_f = threadIdx.x;
                    asm volatile("" : "+v"(tid_f));
                    const int j0 = tid_f % upr, r0v = tid_f / upr;
                    const int cc = FORM == 1 ? 2 * (j0 & 7) : 2 * j0;
                    const bool m0 = cc < K;
                    const int col = m0 ? (FORM == 1 ? (j0 >> 3) * ld + cc : cc) : 0;
                    const int pitch = FORM == 1 ? 2 * ld : ld;
                    if (perm) {  // balanced slabs: the rows come through the tile's list (see the flat path)
                        const int sel = (32 + ((tid_f & 63) >> 4)) << 2;
#pragma unroll
                        for (int i = 0; i < PF; ++i) {
                            const int rowg = __builtin_amdgcn_ds_bpermute(sel + 16 * i, rvec);
                            pf[i] = *reinterpret_cast<const vrx_d2*>(X + (uint32_t)(rowg * pitch + col));
                            if (i & 1) __builtin_amdgcn_sched_barrier(0);
                        }
                    } else
#pragma unroll
                    for (int i = 0; i < PF; ++i)
                        pf[i] = *reinterpret_cast<const vrx_d2*>(
                            src + (uint32_t)(min(r0v + i * rstep, rows32 - 1) * pitch + col));
                } else {
#pragma unroll
                    for (int i = 0; i < PF; ++i) {
                        const bool in = pad_act && m0 && r0 + i * rstep < rows32;
                        pf[i] = in ? *reinterpret_cast<const vrx_d2*>(src + off0 + i * step) : vrx_d2{0.0, 0.0};
                    }
                }
            } else if (FORM != 0) {
                // any K / row stride, element-wise (two 8-B loads per unit), unconditional like the
                // 16-B units above: unit j0 = columns cc, cc + 1 (FORM 1: of half j0 >> 3)
                int tid_f = threadIdx.x;
                asm volatile("" : "+v"(tid_f));
                const int j0 = tid_f % upr, r0v = tid_f / upr;
                const int cc = FORM == 1 ? 2 * (j0 & 7) : 2 * j0;
                const int half = FORM == 1 ? (j0 >> 3) * ld : 0, pitch = FORM == 1 ? 2 * ld : ld;
                const int c0 = half + (cc < K ? cc : 0), c1 = half + (cc + 1 < K ? cc + 1 : 0);
                const int sel = (32 + ((tid_f & 63) >> 4)) << 2;
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const int at = perm ? __builtin_amdgcn_ds_bpermute(sel + 16 * i, rvec) * pitch
                                        : min(r0v + i * rstep, rows32 - 1) * pitch;
                    const double* base = perm ? X : src;  // (the list names rows of the whole operand)
                    vrx_d2 v;
                    v.x = base[at + c0];
                    v.y = base[at + c1];
                    pf[i] = v;
                }
            } else if (MODE == 1) {  // unit j0 = (w1, w2) of column j0
                // (pair words: zero-filled rows / columns, conditional loads; the offsets are
                //  formed per slab from an opaque copy of the thread index, as above)
                int tid_f = threadIdx.x;
                asm volatile("" : "+v"(tid_f));
                const int j0 = tid_f % upr, r0 = tid_f / upr;
                const bool pad_act = tid_f < padT;
                const bool m0 = j0 < K;
                const int off0 = r0 * ld + j0, step = rstep * ld;
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const bool in = pad_act && m0 && r0 + i * rstep < rows32;
                    pf[i] = in ? reinterpret_cast<const vrx_d2*>(src)[off0 + i * step] : vrx_d2{0.0, 0.0};
                }
            } else {  // unit j0 = columns 2*j0, 2*j0 + 1
                int tid_f = threadIdx.x;
                asm volatile("" : "+v"(tid_f));
                const int j0 = tid_f % upr, r0 = tid_f / upr;
                const bool pad_act = tid_f < padT;
                const bool m0 = 2 * j0 < K, m1 = 2 * j0 + 1 < K;
                const int off0 = r0 * ld + 2 * j0, step = rstep * ld;
#pragma unroll
                for (int i = 0; i < PF; ++i) {
                    const bool in = pad_act && r0 + i * rstep < rows32;
                    vrx_d2 v;
                    v.x = in && m0 ? src[off0 + i * step] : 0.0;
                    v.y = in && m1 ? src[off0 + i * step + 1] : 0.0;
                    pf[i] = v;
                }
            }
        }
    };
    auto slab_store = [&]() {
        vrx_d2* dst = reinterpret_cast<vrx_d2*>(slab);
        if (!PADK) {
            const int n16 = slab_doubles / 2;
            int tid_s = threadIdx.x;
            asm volatile("" : "+v"(tid_s));
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int at = tid_s + i * NT;
                if (at < n16) dst[at] = pf[i];
            }
        } else {
            int tid_s = threadIdx.x;  // (as above: nothing of this survives the walk in a register)
            asm volatile("" : "+v"(tid_s));
            const int j0s = tid_s % upr, r0s = tid_s / upr;
#pragma unroll
            for (int i = 0; i < PF; ++i) {
                const int row = r0s + i * rstep;
                if (tid_s < padT && row < slab_rows) dst[row * upr + j0s] = pf[i];
            }
        }
    };

    // ---- entry stream: global -> LDS directly (LDS-DMA: no staging registers, no ds_write).
    // A chunk is 256 words (one global_load_lds_dwordx4 per wave) on the absolute 256-word grid
    // of the wave's stream; the ring holds two.  When the walk enters a chunk it waits for that
    // chunk and then issues the next one into the slot it has just left, so exactly one chunk
    // is in flight, a chunk ahead of the walk.  The compiler does not see these loads: its own
    // vmcnt waits only become stricter.  The walk's wait is exact where the instruction count
    // of the slab prefetch is fixed (PRECISE): the prefetch of the next slab, issued after the
    // chunk, stays in flight (s_waitcnt vmcnt(NPF)) instead of being drained with it.
    const int stream_lo = __builtin_amdgcn_readfirstlane(bw[(int64_t)s_lo * NRV]) & ~(U * G - 1);
    const int stream_end = __builtin_amdgcn_readfirstlane(bw[(int64_t)s_hi * NRV]) & ~(U * G - 1);
    const int base0 = stream_lo & ~(VRX_CHUNK - 1);
    const int clamp_last = max(stream_end - 4, 0);  // lanes past the range re-read its last 16 B
    const uint32_t ring_lds = (uint32_t)(wave * VRX_RING * 4);  // (dynamic LDS starts at 0)
    auto dma_issue = [&](int pos) {
        // (4 * lane from the hardware lane counter, formed again at every issue: kept live across
        //  the walk it would cost a register the walk has not got)
        int lane4;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0\n\tv_lshlrev_b32 %0, 2, %0"
                     : "=v"(lane4));
        const uint32_t* gsrc = stream + min(pos + lane4, clamp_last);
        const uint32_t dst =
            __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)((pos & (VRX_RING - 1)) * 4));
        unsigned keep;  // M0 = LDS destination of lane 0; written and restored in one statement
        asm volatile(
            "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
            "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(gsrc), "s"(dst)
            : "memory");
    };
    // ND chunks are in flight ahead of the walk (the ring holds ND + 1): when the walk enters
    // the chunk at ring_evt it waits for that chunk -- the ND - 1 younger chunks, and the slab
    // prefetch if it was issued after the chunk, stay in flight -- and issues the chunk ND
    // ahead into the slot it has just left (positions past the wave's range re-read its last
    // words).
    constexpr int ND = VRX_RING / VRX_CHUNK - 1;
    if (base0 < stream_end)
#pragma unroll
        for (int i = 0; i < ND; ++i) dma_issue(base0 + i * VRX_CHUNK);
    int ring_evt = base0;     // the chunk boundary the walk services next (multiple of CHUNK)
    int since_fetch = ND;     // chunks issued since the last slab prefetch (ND: none in flight)
    auto ring_event = [&]() {
        constexpr int YOUNGER = ND - 1;  // chunks issued behind it
        if (PRECISE && since_fetch < ND)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER + NPF) : "memory");
        else
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(YOUNGER) : "memory");
        since_fetch = min(since_fetch + 1, ND);
        dma_issue(ring_evt + ND * VRX_CHUNK);
        ring_evt += VRX_CHUNK;
    };
    // one entry of this group's segment: word -> 4 column slices -> FMAs
    auto entry = [&](uint32_t w, double (&a)[NQ][2], double (&a2)[NQ][2]) {
        const double ad = (double)((w >> 11) & 2047u), dp = (double)(w & 2047u);
        const uint32_t idx = w >> 22;  // < 1024 and row_bytes <= 256: 24-bit multiply-add
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const double2 x = *reinterpret_cast<const double2*>(
                reinterpret_cast<const char*>(slab) + (__umul24(idx, (uint32_t)row_bytes) + qoff[q]));
            if (MODE == 1) {  // x = (w1, w2) of one column
                a[q][0] = fma(ad, x.x, a[q][0]);
                a[q][0] = fma(dp, x.y, a[q][0]);
            } else {  // x = ID_prob of two adjacent columns
                a[q][0] = fma(ad, x.x, a[q][0]);
                a[q][1] = fma(ad, x.y, a[q][1]);
                a2[q][0] = fma(dp, x.x, a2[q][0]);
                a2[q][1] = fma(dp, x.y, a2[q][1]);
            }
        }
    };

    // bnd words of a slab: stream offset (a multiple of U*G) | entries in the round's last trip
    // (0 = a full trip) for each (round, phase), then the first word of the next slab: lane i
    // holds word i, ONE vector load a slab ahead (it is part of the prefetch the walk counts)
    // the record of a slab: lanes 0 .. NRV its bnd words; balanced slabs: lanes 32 .. 63 the contracted rows
    // this wave stages for the slab BEHIND it (they are needed a slab earlier, by the prefetch)
    // (scalar bases + 32-bit lane offsets formed from an opaque copy of the lane index: per-lane 64-bit
    //  pointers kept across the walk would cost registers it has not got)
    auto bnd_load = [&](int s_b) {
        int lane_o = threadIdx.x & 63;
        asm volatile("" : "+v"(lane_o));
        const int32_t* a = bw + (int64_t)s_b * NRV;
        return a[(uint32_t)min(lane_o, NRV)];
    };
    // balanced slabs: lane 32 + 4 i + q holds the contracted row of slab-local row 4 * wave + 64 i + q of slab
    // s_p -- the 32 rows this wave stages.  A load of its own (loaded a slab before the prefetch that uses it,
    // pending until then like the bnd words); issued whether or not the stream is balanced: the number of
    // loads in flight behind a slab prefetch is one constant (NPF)
    auto rows_load = [&](int s_p) {
        int lane_o = threadIdx.x & 63;
        asm volatile("" : "+v"(lane_o));
        const int32_t* b = perm ? perm + ((int64_t)tile * n_slab + min(s_p, n_slab - 1)) * slab_rows : bw;
        const int j = max(lane_o - 32, 0);
        const int pos = perm ? min(4 * wave + 64 * (j >> 2) + (j & 3), slab_rows - 1) : 0;
        return b[(uint32_t)pos];
    };
    int bvec = bnd_load(s_lo);
    int rvec = 0;
    if (GATHER) rvec = rows_load(s_lo);
    slab_fetch(s_lo, rvec);
    if (GATHER) rvec = rows_load(s_lo + 1);
    for (int s = s_lo; s < s_hi; ++s) {
#ifndef VRX_PROBE_NOBAR    // TIMING PROBES ONLY (scratch builds): barriers / slab staging compiled out (racy, wrong results)
        VRX_PROBE(tm_bar1, __syncthreads())  // every wave is done reading the previous slab
#endif
        // The slab's bnd words go to scalar registers BEFORE the next slab's prefetch is issued: the
        // compiler guards the readlanes with a full `s_waitcnt vmcnt(0)` (it cannot count across the loop),
        // which up to round 5 sat BEHIND the prefetch and drained it on the spot -- one exposed memory round
        // trip per visit, the 10-14 % a pass spent "staging" (profiles/r06_pass_structure_probes.txt).  Here
        // it only meets loads of the previous visit, which landed long ago.
        int bcur[NRV + 1];
#pragma unroll
        for (int i = 0; i <= NRV; ++i) bcur[i] = __builtin_amdgcn_readlane(bvec, i);
#ifndef VRX_PROBE_NOSTAGE
        VRX_PROBE(tm_stage, slab_store())
        if (s + 1 < s_hi) {  // (with the two loads below: NPF vector loads)
            slab_fetch(s + 1, rvec);
            since_fetch = 0;
        }
#endif
        if (s + 1 < s_hi) {
            bvec = bnd_load(s + 1);
            if (GATHER) rvec = rows_load(s + 2);
        }
#ifndef VRX_PROBE_NOBAR
        VRX_PROBE(tm_bar2, __syncthreads())
#endif
#pragma unroll
        for (int rv = 0; rv < NRV; ++rv) {
            const int r = rv / PH;
            // the round's entries are stored trip-major; the zero words that pad the round's
            // last trip are not executed
            const int braw = bcur[rv];
            const int base = braw & ~(U * G - 1), tail = braw & (U - 1);
            const int end = bcur[rv + 1] & ~(U * G - 1);
            const int full_end = tail ? end - U * G : end;
            if (FORM != 0) {
                // FORM 2: the AD entries of the round's rows feed S1 (acc), the BD entries S2 (acc2)
                double (&ac)[NQ][2] = FORM == 2 && (rv & 1) ? acc2[r] : acc[r];
                // A trip of NE <= U entries: every slice is requested (ds_read_b128, integer
                // addresses: word offset bits | lane offset, the slab base is part of the word)
                // before the first FMA; the FMAs of the first half wait for their four reads
                // only.  The value is the high dword of an IEEE double (low dword 0), so there
                // is no conversion: 3 VALU instructions of overhead per entry.
                auto trip = [&](int at, auto ne_tag) {
                    constexpr int NE = decltype(ne_tag)::value;
                    uint32_t w[NE];
                    VRX_PROBE_TRIP
                    {
#ifndef VRX_PROBE_NODMA   // TIMING PROBE ONLY (scratch builds): the stream's LDS-DMA and its waits compiled out
                        if (at >= ring_evt) VRX_PROBE(tm_dma, ring_event())
#endif
                        // the group's U words are adjacent (padding words fill a short last trip)
#ifdef VRX_PROBE_NORING   // TIMING PROBE ONLY: no ring read -- every word is the slab's first half row, value 0
                        uint32_t pw = (uint32_t)VRX_LDS_WAVES * VRX_RING * 4u + ((uint32_t)at & 0x380u);
                        asm volatile("" : "+v"(pw));
                        const uint4 q4 = make_uint4(pw, pw ^ 128u, pw, pw ^ 128u);
#else
                        const uint4 q4 = *reinterpret_cast<const uint4*>(ring + (at & (VRX_RING - 1)) + g * U);
#endif
                        const uint32_t qq[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
                        for (int u = 0; u < NE; ++u) w[u] = qq[u];
                    }
                    if constexpr (NQ == 1) {
                        // EXPERIMENT BUILDS ONLY (-DVRX_LDS_LPE_DEF=8: rounds of 8 rows, 8 lanes and 2
                        // columns each, DESIGN_HISTORY.md 4.2 r4): one 16-B slice per word and lane
                        vrx_d2 y[NE];
                        if (NE == 4) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[NE - 1]));
                        if (NE == 3) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[NE - 1]));
                        if (NE == 2) asm volatile("" : "+v"(w[0]), "+v"(w[NE - 1]));
                        uint32_t b0, b1;
#define VRX_R1(X0, W, A) "v_and_or_b32 " A ", " W ", %[msk], %[q0]\n\tds_read_b128 " X0 ", " A "\n\t"
                        if constexpr (NE == 4)
                            asm volatile(VRX_R1("%[x0]", "%[w0]", "%[a0]") VRX_R1("%[x1]", "%[w1]", "%[a1]")
                                         VRX_R1("%[x2]", "%[w2]", "%[a0]") VRX_R1("%[x3]", "%[w3]", "%[a1]")
                                         : [x0] "=&v"(y[0]), [x1] "=&v"(y[1]), [x2] "=&v"(y[2]), [x3] "=&v"(y[NE - 1]),
                                           [a0] "=&v"(b0), [a1] "=&v"(b1)
                                         : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[NE - 1]),
                                           [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                         : "memory");
                        else {
#pragma unroll
                            for (int u = 0; u < NE; ++u)
                                asm volatile(VRX_R1("%[x0]", "%[w0]", "%[a0]")
                                             : [x0] "=&v"(y[u]), [a0] "=&v"(b0)
                                             : [w0] "v"(w[u]), [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                             : "memory");
                        }
#undef VRX_R1
                        if (NE == 4) {
                            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(y[0]), "+v"(y[1]));
                        } else {
#pragma unroll
                            for (int u = 0; u < NE; ++u) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[u]));
                        }
#pragma unroll
                        for (int u = 0; u < NE; ++u) {
                            if (NE == 4 && u == 2)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(y[2]), "+v"(y[NE - 1]), "+v"(ac[0][0]), "+v"(ac[0][1]));
                            const double v = __builtin_bit_cast(double, vrx_u2{0u, w[u] & 0xfffc0000u});
                            ac[0][0] = fma(v, y[u][0], ac[0][0]);
                            ac[0][1] = fma(v, y[u][1], ac[0][1]);
                        }
                    } else {
                    vrx_d2 x[NE][2];
                    // (all words have landed before the first slice is requested: the compiler's
                    //  own waits do not count the reads issued from inline assembly)
                    if (NE == 4) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[NE - 1]));
                    if (NE == 3) asm volatile("" : "+v"(w[0]), "+v"(w[1]), "+v"(w[NE - 1]));
                    if (NE == 2) asm volatile("" : "+v"(w[0]), "+v"(w[NE - 1]));
                    // ONE statement per trip: between separate asm statements the compiler pads
                    // with an s_nop.  Address = word offset bits | lane offset; two address
                    // registers alternate (an LDS instruction reads its address when it issues).
                    {
                        uint32_t a0, a1;
#define VRX_RD(X0, X1, W)                                    \
    "v_and_or_b32 %[a0], " W ", %[msk], %[q0]\n\t"            \
    "ds_read_b128 " X0 ", %[a0]\n\t"                         \
    "v_xor_b32 %[a1], 16, %[a0]\n\t"                         \
    "ds_read_b128 " X1 ", %[a1]\n\t"
                        if constexpr (NE == 4)
                            asm volatile(VRX_RD("%[x00]", "%[x01]", "%[w0]") VRX_RD("%[x10]", "%[x11]", "%[w1]")
                                         VRX_RD("%[x20]", "%[x21]", "%[w2]") VRX_RD("%[x30]", "%[x31]", "%[w3]")
                                         : [x00] "=&v"(x[0][0]), [x01] "=&v"(x[0][1]), [x10] "=&v"(x[1][0]),
                                           [x11] "=&v"(x[1][1]), [x20] "=&v"(x[2][0]), [x21] "=&v"(x[2][1]),
                                           [x30] "=&v"(x[NE - 1][0]), [x31] "=&v"(x[NE - 1][1]), [a0] "=&v"(a0), [a1] "=&v"(a1)
                                         : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[2]), [w3] "v"(w[NE - 1]),
                                           [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                         : "memory");
                        else if constexpr (NE == 3)
                            asm volatile(VRX_RD("%[x00]", "%[x01]", "%[w0]") VRX_RD("%[x10]", "%[x11]", "%[w1]")
                                         VRX_RD("%[x20]", "%[x21]", "%[w2]")
                                         : [x00] "=&v"(x[0][0]), [x01] "=&v"(x[0][1]), [x10] "=&v"(x[1][0]),
                                           [x11] "=&v"(x[1][1]), [x20] "=&v"(x[NE - 1][0]), [x21] "=&v"(x[NE - 1][1]),
                                           [a0] "=&v"(a0), [a1] "=&v"(a1)
                                         : [w0] "v"(w[0]), [w1] "v"(w[1]), [w2] "v"(w[NE - 1]),
                                           [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                         : "memory");
                        else if constexpr (NE == 2)
                            asm volatile(VRX_RD("%[x00]", "%[x01]", "%[w0]") VRX_RD("%[x10]", "%[x11]", "%[w1]")
                                         : [x00] "=&v"(x[0][0]), [x01] "=&v"(x[0][1]), [x10] "=&v"(x[NE - 1][0]),
                                           [x11] "=&v"(x[NE - 1][1]), [a0] "=&v"(a0), [a1] "=&v"(a1)
                                         : [w0] "v"(w[0]), [w1] "v"(w[NE - 1]),
                                           [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                         : "memory");
                        else
                            asm volatile(VRX_RD("%[x00]", "%[x01]", "%[w0]")
                                         : [x00] "=&v"(x[0][0]), [x01] "=&v"(x[0][1]), [a0] "=&v"(a0), [a1] "=&v"(a1)
                                         : [w0] "v"(w[0]), [msk] "s"(0x3ff80u), [q0] "v"(qoff[0])
                                         : "memory");
#undef VRX_RD
                    }
                    constexpr int H = NE > 2 ? 2 : NE;  // entries of the first half
                    if (NE > 2) {
                        if (NE == 4)
                            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]));
                        else
                            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]));
                    }
#pragma unroll
                    for (int u = 0; u < NE; ++u) {
                        if (u == H || NE <= 2) {
                            // (the accumulators tie this wait behind the first half's FMAs)
                            if (u == H && NE == 4)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[2][0]), "+v"(x[2][1]), "+v"(x[3][0]), "+v"(x[3][1]),
                                             "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[1][0]), "+v"(ac[1][1]));
                            else if (u == H && NE == 3)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[2][0]), "+v"(x[2][1]),
                                             "+v"(ac[0][0]), "+v"(ac[0][1]), "+v"(ac[1][0]), "+v"(ac[1][1]));
                            else if (u == 0 && NE == 2)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1]));
                            else if (u == 0 && NE == 1)
                                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x[0][0]), "+v"(x[0][1]));
                        }
                        const double v = __builtin_bit_cast(double, vrx_u2{0u, w[u] & 0xfffc0000u});
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            ac[q][0] = fma(v, x[u][q][0], ac[q][0]);
                            ac[q][1] = fma(v, x[u][q][1], ac[q][1]);
                        }
                    }
                    }  // NQ == 2
                };
                for (int at = base; at < full_end; at += U * G) trip(at, std::integral_constant<int, 4>());
                if (tail == 1) trip(full_end, std::integral_constant<int, 1>());
                if (tail == 2) trip(full_end, std::integral_constant<int, 2>());
                if (tail == 3) trip(full_end, std::integral_constant<int, 3>());
                continue;
            }
            for (int at = base; at < full_end; at += U * G) {
                if (at >= ring_evt) ring_event();
                // trips start at multiples of U*G = 64 words and the ring is a multiple of
                // that, so a trip never wraps: one address, constant offsets
                const uint32_t* rp = ring_g + (at & (VRX_RING - 1));
                uint32_t w[US];
#pragma unroll
                for (int u = 0; u < US; ++u) w[u] = rp[u * SPLIT * G];
#pragma unroll
                for (int u = 0; u < US; ++u) entry(w[u], acc[r], acc2[r]);
            }
            if (tail) {  // (entries past the tail are zero words: harmless where SPLIT > 1)
                if (full_end >= ring_evt) ring_event();
                const uint32_t* rp = ring_g + (full_end & (VRX_RING - 1));
                uint32_t w[US];
#pragma unroll
                for (int u = 0; u < US; ++u) w[u] = rp[u * SPLIT * G];
#pragma unroll
                for (int u = 0; u < US; ++u)
                    if (u * SPLIT < tail) entry(w[u], acc[r], acc2[r]);
            }
        }
    }
    // (every issued chunk has been awaited by the walk; this only guards the invariant that no
    //  LDS-DMA write is in flight when the workgroup's LDS is released)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VRX_PROBE_END
    // ---- every group holds the complete sums of its rows: store them ------------------------
    if (SPLIT > 1) {  // partial sums of the SPLIT entry streams: butterfly over the lanes
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int m = LPR; m < LPE; m <<= 1) {
                        acc[r][q][h] += __shfl_xor(acc[r][q][h], m, 64);
                        if (MODE == 0) acc2[r][q][h] += __shfl_xor(acc2[r][q][h], m, 64);
                    }
    }
    double* dst = out + (int64_t)slot * n_rows * ld * NV;
    // (the lane's coordinates are derived again, from an opaque copy of the thread index: kept
    //  live across the walk they would cost registers the walk has not got)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int g_e = lane_e / LPE, sub_e = (lane_e % LPE) / LPR, kl_e = lane_e % LPR;
    const bool kok_e = kl_e * CP < K;
    {
        const int g = g_e, kl = kl_e;
        if (kok_e && sub_e == 0)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int64_t row = rowmap[((int64_t)tile * VRX_LDS_WAVES + wave_e) * RW + r * G + g];
            if (row >= 0) {  // tile position -> row (rows are dealt to rounds by length)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int slice = kl * NQ + (q + g) % NQ;  // 16-B slice of the dense row
                    if (FORM == 1) {  // columns 2*slice, 2*slice + 1 of logLik_ID
                        double* o = dst + row * ld + 2 * slice;
#ifdef VRX_PROBE_WT_PARTIALS  // TIMING PROBE ONLY (scratch builds): the partial planes written through
                        if (!PADK) {  //  to memory (sc0 sc1), as a cross-XCD fold inside the pass would need
                            const vrx_d2 v2 = {acc[r][q][0], acc[r][q][1]};
                            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(v2) : "memory");
                        } else
#endif
                        if (!PADK)
                            *reinterpret_cast<double2*>(o) = make_double2(acc[r][q][0], acc[r][q][1]);
                        else if (PADK == 2) {  // even K and stride: the pair is whole or absent
                            if (2 * slice < K)
                                *reinterpret_cast<double2*>(o) = make_double2(acc[r][q][0], acc[r][q][1]);
                        } else {
                            if (2 * slice < K) o[0] = acc[r][q][0];
                            if (2 * slice + 1 < K) o[1] = acc[r][q][1];
                        }
                    } else if (MODE == 1) {
                        if (!PADK || slice < K) dst[row * ld + slice] = acc[r][q][0];
                    } else {  // columns 2*slice, 2*slice+1; S[row][k] = (s1, ss)
                        if (FORM == 2) {  // acc2 holds S2 = BD @ ID_prob: ss = s1 + s2
                            acc2[r][q][0] += acc[r][q][0];
                            acc2[r][q][1] += acc[r][q][1];
                        }
                        double2* o = reinterpret_cast<double2*>(dst) + row * ld + 2 * slice;
                        if (!PADK || 2 * slice < K)
                            o[0] = make_double2(acc[r][q][0], acc2[r][q][0]);
                        if (!PADK || 2 * slice + 1 < K)
                            o[1] = make_double2(acc[r][q][1], acc2[r][q][1]);
                    }
                }
            }
        }
    }
    }  // work items
}

// out[row][c] = sum over ranges (outer) and the row's pieces (inner), fixed order
__global__ __launch_bounds__(VRX_BLOCK) void vrx_sum_pieces(
    int64_t n_rows, int width, int n_range, int64_t n_vrows, const int32_t* __restrict__ vptr,
    const uint16_t* __restrict__ npiece, const double* __restrict__ partial, double* __restrict__ out,
    const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= n_rows * width) return;
    const int64_t row = i / width;
    const int c = (int)(i - row * width);
    const int v0 = vptr[row], v1 = vptr[row + 1];
    // the terms are numbered t = r * (v1 - v0) + (v - v0) and added in that order; 16 loads are
    // in flight at a time (a plain loop pays one memory round trip per term)
    const int np = v1 - v0, nt = n_range * np;
    double s = 0.0;
    for (int t0 = 0; t0 < nt; t0 += 16) {
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int t = t0 + j, r = t / np, v = v0 + (t - r * np);
            // (partial array r holds a term of piece v only if the piece's tile was cut that often)
            x[j] = t < nt && r < npiece[v] ? partial[((int64_t)r * n_vrows + v) * width + c] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s += x[j];
    }
    out[i] = s;
}

// the same sum when a row has many terms (few long rows cut into many pieces, many ranges):
// one wavefront per output, lane l adds terms l, l + 64, ... in order, then the fixed
// butterfly of wave_sum
__global__ __launch_bounds__(VRX_BLOCK) void vrx_sum_pieces_wave(
    int64_t n_rows, int width, int n_range, int64_t n_vrows, const int32_t* __restrict__ vptr,
    const uint16_t* __restrict__ npiece, const double* __restrict__ partial, double* __restrict__ out,
    const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = ((int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (i >= n_rows * width) return;
    const int64_t row = i / width;
    const int c = (int)(i - row * width);
    const int v0 = vptr[row], np = vptr[row + 1] - v0, nt = n_range * np;
    double s = 0.0;
    for (int t = lane; t < nt; t += 64) {
        const int r = t / np, v = v0 + (t - r * np);
        if (r < npiece[v]) s += partial[((int64_t)r * n_vrows + v) * width + c];
    }
    s = wave_sum(s);
    if (lane == 0) out[i] = s;
}

// Rows that were cut into pieces (heavy-tailed data: ~10 % of the rows at c3 size), ahead of a
// consumer that sums the partial arrays itself (vrx_theta_partial, vrx_cell_softmax): EIGHT lanes
// per (split row, column) add the row's terms -- lane l the terms l, l + 8, ... in (slot, piece)
// order, then three butterfly steps -- and leave the sum in slot 0 of the row's first piece, where
// the consumer reads ONE value for the row.  A typical split row has 2-3 pieces x <= 3 slots
// (one load per lane), the longest ~100 terms.  (Measured at c3 size, dense kernels per iteration:
// the consumers' own per-thread loop over the terms 0.20 ms; one wavefront per element 0.27 ms --
// 400 k waves of a few dependent loads each; this form: see DESIGN_HISTORY.md 4.2.)
// Two properties a caller must know:
//  * ORDER.  The terms are added strided over 8 lanes and then by a butterfly; the consumers that
//    form S / logLik_ID explicitly (vrx_s_from_virtual, vrx_sum_pieces: the step-wise API,
//    vrx_model_step) add the same terms sequentially.  A fit and the same iterations driven step by
//    step therefore round differently on split rows: 1e-12 on the ELBO, 3e-9 on small posteriors
//    after six iterations (tests/test_gpu_parity.py::test_fit_loop_and_stepwise_api_agree_on_split_rows).
//  * NOT IDEMPOTENT.  The sum overwrites slot 0 of the row's first piece, which is one of its
//    terms: a second call on the same partial array would count the other terms twice.  The host
//    calls it once per pass output, right before the one consumer that reads the array, and clears
//    s_pending / l_pending there (theta_step, softmax_step in vrx_engine.hip).
__global__ __launch_bounds__(VRX_BLOCK) void vrx_fold_split(
    int64_t n_split, const int32_t* __restrict__ split_rows, int width, int64_t n_vrows,
    const int32_t* __restrict__ vptr, const uint16_t* __restrict__ npiece, double* partial,
    const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = ((int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x) >> 3;
    const int sub = threadIdx.x & 7;
    const bool live = i < n_split * width;  // (idle groups stay: the group sum below wants whole waves)
    const int64_t j = live ? i / width : 0;
    const int c = (int)(i - j * width);
    const int64_t row = live ? split_rows[j] : 0;
    const int v0 = live ? vptr[row] : 0, np = live ? vptr[row + 1] - v0 : 0;
    int most = 0;
    for (int q = sub; q < np; q += 8) most = max(most, (int)npiece[v0 + q]);
    most = max(most, __shfl_xor(most, 1, 64));
    most = max(most, __shfl_xor(most, 2, 64));
    most = max(most, __shfl_xor(most, 4, 64));
    const int nt = most * np;
    double s = 0.0;
    for (int t0 = sub; t0 < nt; t0 += 32) {  // four loads in flight per lane
        double x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 8 * u, r = t / max(np, 1), v = v0 + (t - r * np);
            x[u] = t < nt && r < npiece[v] ? partial[((int64_t)r * n_vrows + v) * width + c] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) s += x[u];
    }
    s = vrx_group_sum<8>(s);
    if (live && sub == 0) partial[(int64_t)v0 * width + c] = s;
}

// partial[slot][row][width] summed in slot order over the npiece[row] partial arrays that hold a
// term of the row (the pieces its tile was cut into, TiledStream::items)
__global__ __launch_bounds__(VRX_BLOCK) void vrx_sum_ranges(int64_t n, int width,
                                                            const uint16_t* __restrict__ npiece,
                                                            const double* __restrict__ partial,
                                                            double* __restrict__ out,
                                                            const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= n) return;
    const int n_range = npiece[i / width];
    double s = 0.0;
    for (int r0 = 0; r0 < n_range; r0 += 16) {  // 16 loads in flight, added in slot order
        double x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = r0 + j < n_range ? partial[(int64_t)(r0 + j) * n + i] : 0.0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += x[j];
    }
    out[i] = s;
}

// S = (S1, S1 + S2) from the planar partial sums of the virtual rows (TiledStream::virt):
// partial[slot][piece v][Kt]; row 2 n + kind of the virtual matrix is the pieces
// [vptr[2 n + kind], vptr[2 n + kind + 1]) (vptr == null: one piece each, v = the row), piece v
// has a term in the first npiece[v] arrays.  Terms in (slot, piece) order, like vrx_sum_pieces.
__global__ __launch_bounds__(VRX_BLOCK) void vrx_s_from_virtual(
    int64_t N, int Kt, int64_t n_vrows, const int32_t* __restrict__ vptr, const uint16_t* __restrict__ npiece,
    const double* __restrict__ partial, double2* __restrict__ S, const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= N * Kt) return;
    const int64_t n = i / Kt;
    const int c = (int)(i - n * Kt);
    double sk[2];
    for (int kind = 0; kind < 2; ++kind) {
        const int64_t row = 2 * n + kind;
        const int64_t v0 = vptr ? vptr[row] : row, v1 = vptr ? vptr[row + 1] : row + 1;
        double t = 0.0;
        if (v1 - v0 == 1) {  // one piece (every row of uniform data): its slots, 8 loads in flight
            const int nr = npiece[v0];
            const double* src = partial + v0 * Kt + c;
            for (int r0 = 0; r0 < nr; r0 += 8) {
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = r0 + u < nr ? src[(int64_t)(r0 + u) * n_vrows * Kt] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + u < nr) t += x[u];
            }
        } else {
            int most = 0;
            for (int64_t v = v0; v < v1; ++v) most = max(most, (int)npiece[v]);
            for (int r = 0; r < most; ++r)
                for (int64_t v = v0; v < v1; ++v)
                    if (r < npiece[v]) t += partial[((int64_t)r * n_vrows + v) * Kt + c];
        }
        sk[kind] = t;
    }
    S[i] = make_double2(sk[0], sk[0] + sk[1]);
}

// Second stage for rows that were split over several segments: in-order sum of the slots.
// VPE = values per element (2 for the variant pass, 1 for the cell pass).
template <int VPE>
__global__ __launch_bounds__(VRX_BLOCK) void vrx_sum_slots(int64_t n_multi, int K,
                                                           const int32_t* __restrict__ multi_row,
                                                           const int32_t* __restrict__ multi_ptr,
                                                           const double* __restrict__ partial,
                                                           double* __restrict__ out,
                                                           const int32_t* __restrict__ ctl, int n_batch) {
    if (vrx_all_stopped(ctl, n_batch)) return;
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    const int64_t per = (int64_t)K * VPE;
    if (i >= n_multi * per) return;
    const int64_t j = i / per, c = i - j * per;
    double s = 0.0;
    for (int q = multi_ptr[j]; q < multi_ptr[j + 1]; ++q) s += partial[(int64_t)q * per + c];
    out[(int64_t)multi_row[j] * per + c] = s;
}

// ------------------------------------------------------------------------------------
// ELBO finalisation of one restart (get_ELBO, vireo_model.py:222-248; the stop rule of _fit_VB,
// :266-274) as a block-level routine: the body of vrx_elbo_final further down, and -- on
// launch-bound problems -- of the extra block that rides in the NEXT iteration's
// vrx_theta_partial (VrxElboRide).
// ------------------------------------------------------------------------------------
struct VrxElboIn {  // by value; per restart r: partial arrays r * n_*_part on, trace r * trace_stride on
    const double *cell_part, *gt_part, *th_part;
    int n_cell_part, n_gt_part, n_th_part;
    double *elbo, *parts;  // elbo: the trace (slot rule.it is written)
    int64_t trace_stride;
};

// All first loads -- the stop word, the previous ELBO, and the first sweep of each partial array --
// are requested together: the block is otherwise five dependent memory round trips long.
__device__ __forceinline__ void vrx_elbo_final_block(const double* cell_part, int n_cell_part,
                                                     const double* gt_part, int n_gt_part,
                                                     const double* th_part, int n_th_part,
                                                     double* elbo_out, double* parts_out,
                                                     const VrxStopRule& rule, int32_t* ctl) {
    __shared__ double tot[4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const int stop = ctl[VRX_CTL_STOP];
    const bool judge = rule.active && rule.it > rule.min_iter;
    double prev = 0.0;
    if (threadIdx.x == 0 && judge) prev = elbo_out[rule.it - 1];
    // the loads of 8 strides are issued together (one memory round trip instead of 8), the
    // additions keep the order of the plain strided loop
    constexpr int UN = 8;
    const int tid = threadIdx.x;
    double2 vc[UN];
    double vg[UN], vt[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        const int b = tid + u * VRX_BLOCK;
        vc[u] = b < n_cell_part ? reinterpret_cast<const double2*>(cell_part)[b] : make_double2(0.0, 0.0);
        vg[u] = b < n_gt_part ? gt_part[b] : 0.0;
        vt[u] = b < n_th_part ? th_part[b] : 0.0;
    }
    if (stop) return;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
        acc[0] += vc[u].x;
        acc[1] += vc[u].y;
    }
    for (int b0 = tid + UN * VRX_BLOCK; b0 < n_cell_part; b0 += UN * VRX_BLOCK) {
        double2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int b = b0 + u * VRX_BLOCK;
            v[u] = b < n_cell_part ? reinterpret_cast<const double2*>(cell_part)[b]
                                   : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc[0] += v[u].x;
            acc[1] += v[u].y;
        }
    }
    auto strided = [&](const double* p, int n, const double (&v0)[UN], double& a) {
#pragma unroll
        for (int u = 0; u < UN; ++u) a += v0[u];
        for (int b0 = tid + UN * VRX_BLOCK; b0 < n; b0 += UN * VRX_BLOCK) {
            double v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int b = b0 + u * VRX_BLOCK;
                v[u] = b < n ? p[b] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) a += v[u];
        }
    };
    strided(gt_part, n_gt_part, vg, acc[2]);
    strided(th_part, n_th_part, vt, acc[3]);
    block_sum_store<4>(acc, tot);
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cur = tot[0] - tot[1] - tot[2] - tot[3];
        elbo_out[rule.it] = cur;
#pragma unroll
        for (int i = 0; i < 4; ++i) parts_out[i] = tot[i];
        // the stop rule of _fit_VB / _fit_BV (vireo_model.py:266-274, bmm_model.py:190-199)
        if (judge) {
            if (cur < prev - 1e-6) {
                ctl[VRX_CTL_WARN] |= 1;
            } else if (rule.it == rule.max_iter - 1) {
                ctl[VRX_CTL_WARN] |= 2;
            } else if (cur - prev < rule.eps) {
                ctl[VRX_CTL_IT] = rule.it;
                __threadfence();
                ctl[VRX_CTL_STOP] = 1;
            }
        }
    }
}

struct VrxElboRide {  // by value to vrx_theta_partial: on = 1 adds the ELBO block of the previous iteration
    int on;
    VrxElboIn in;
    VrxStopRule rule;
};

// ------------------------------------------------------------------------------------
// theta  (Vireo.update_theta_size, vireoSNP/utils/vireo_model.py:165-185)
// ------------------------------------------------------------------------------------
// Beta update + digammas + KL for ONE theta row.  psi is laid out [3][rows][T].
// update == 0: only derive psi / KL from the current beta_mu, beta_sum.
__device__ __forceinline__ double vrx_theta_row(int T, int update, int fix_sum, const double* add1,
                                                const double* add2, const double* p1,
                                                const double* p2, double* mu, double* sm,
                                                double* psi1, double* psi2, double* psis) {
    double kl = 0.0;
    for (int t = 0; t < T; ++t) {
        double m = mu[t], s = sm[t];
        if (update) {
            const double t1 = p1[t] + add1[t];
            const double t2 = p2[t] + add2[t];
            m = t1 / (t1 + t2);
            if (!fix_sum) s = t1 + t2;
            mu[t] = m;
            sm[t] = s;
        }
        const double s1 = m * s, s2 = (1.0 - m) * s;  // theta_s1/theta_s2 properties
        const double d1 = vrx_digamma(s1), d2 = vrx_digamma(s2), ds = vrx_digamma(s1 + s2);
        psi1[t] = d1;
        psi2[t] = d2;
        psis[t] = ds;
        kl += vrx_beta_kl(s1, s2, p1[t], p2[t], d1, d2, ds);
    }
    return kl;
}

// stage 2 (shared theta): one block sums the stage-1 partials in a fixed order.
__device__ __forceinline__ void vrx_theta_final_block(int n_part, int T, int update, int fix_sum,
                                                      const double* part, const double* prior1,
                                                      const double* prior2, double* mu, double* sm,
                                                      double* psi, double* kl_out, int stop) {
// (as in vrx_gt_update, which inlines this finalisation for small problems: the two paths must
//  round alike whichever one nb_theta selects -- no FMA contraction, like the reference's NumPy)
#pragma clang fp contract(off)
    __shared__ double tot[2 * VRX_MAXT];
    double acc[2 * VRX_MAXT];
#pragma unroll
    for (int t = 0; t < 2 * VRX_MAXT; ++t) acc[t] = 0.0;
    // (the partials -- four strides in flight together -- and the Beta parameters are requested
    //  before the caller's stop word is tested: one memory round trip instead of three)
    if (update)
        for (int b0 = threadIdx.x; b0 < n_part; b0 += 4 * VRX_BLOCK) {
            double v[4][2 * VRX_MAXT];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 2 * VRX_MAXT; ++t)
                    v[u][t] = b0 + u * VRX_BLOCK < n_part && t % VRX_MAXT < T
                                  ? part[(int64_t)(b0 + u * VRX_BLOCK) * 2 * VRX_MAXT + t] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (b0 + u * VRX_BLOCK < n_part)
#pragma unroll
                    for (int t = 0; t < 2 * VRX_MAXT; ++t) acc[t] += v[u][t];
        }
    double in_mu = 0.0, in_sm = 0.0, in_p1 = 0.0, in_p2 = 0.0;
    if (threadIdx.x < T) {
        in_mu = mu[threadIdx.x];
        in_sm = sm[threadIdx.x];
        in_p1 = prior1[threadIdx.x];
        in_p2 = prior2[threadIdx.x];
    }
    if (stop) return;
    block_sum_store<2 * VRX_MAXT>(acc, tot, VRX_MAXT, T);
    __syncthreads();
    // The 9 special-function values per genotype class (3 digammas, 6 log-gammas) are the
    // whole cost of this kernel: spread them over 9*T threads, combine in thread 0 in the
    // same order as vrx_theta_row / vrx_beta_kl (identical arithmetic, ~9x shorter).
    __shared__ double sh[VRX_MAXT][2];   // updated (mu, sum)
    __shared__ double sf[VRX_MAXT][9];   // psi(s1) psi(s2) psi(s12) lg(q1) lg(q2) lg(q12) lg(s1) lg(s2) lg(s12)
    if (threadIdx.x < T) {
        const int t = threadIdx.x;
        double m = in_mu, s = in_sm;
        if (update) {
            const double t1 = in_p1 + tot[t];
            const double t2 = in_p2 + tot[VRX_MAXT + t];
            m = t1 / (t1 + t2);
            if (!fix_sum) s = t1 + t2;
            mu[t] = m;
            sm[t] = s;
        }
        sh[t][0] = m;
        sh[t][1] = s;
    }
    __syncthreads();
    // wave 0 evaluates the 3*T digammas, wave 1 the 6*T log-gammas: every lane of a wave runs
    // the SAME function (a switch over 9 functions inside one wave would run them one after
    // the other), and the two waves run side by side
    {
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        const int per = wv == 0 ? 3 : 6;
        if (wv < 2 && ln < per * T) {
            const int t = ln / per, j = ln % per;
            const double m = sh[t][0], s = sh[t][1];
            const double s1 = m * s, s2 = (1.0 - m) * s, q1 = prior1[t], q2 = prior2[t];
            if (wv == 0) {
                const double x = j == 0 ? s1 : j == 1 ? s2 : s1 + s2;
                sf[t][j] = vrx_digamma(x);
            } else {
                const double x = j == 0 ? q1 : j == 1 ? q2 : j == 2 ? q1 + q2 : j == 3 ? s1 : j == 4 ? s2 : s1 + s2;
                sf[t][3 + j] = lgamma(x);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double kl = 0.0;
        for (int t = 0; t < T; ++t) {
            const double m = sh[t][0], s = sh[t][1];
            const double s1 = m * s, s2 = (1.0 - m) * s, q1 = prior1[t], q2 = prior2[t];
            const double d1 = sf[t][0], d2 = sf[t][1], ds = sf[t][2];
            psi[t] = d1;
            psi[T + t] = d2;
            psi[2 * T + t] = ds;
            const double cq = (sf[t][3] + sf[t][4] - sf[t][5]) - (q1 - 1.0) * d1 -
                              (q2 - 1.0) * d2 + ((q1 + q2) - 2.0) * ds;
            const double cp = (sf[t][6] + sf[t][7] - sf[t][8]) - (s1 - 1.0) * d1 -
                              (s2 - 1.0) * d2 + ((s1 + s2) - 2.0) * ds;
            kl += cq - cp;
        }
        kl_out[0] = kl;
    }
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_theta_final(int n_part, int T, int update,
                                                             int fix_sum, const double* part,
                                                             const double* prior1,
                                                             const double* prior2, double* mu,
                                                             double* sm, double* psi,
                                                             double* kl_out,
                                                             const int32_t* __restrict__ ctl) {
    const int r = blockIdx.x;  // one block per restart of the batch
    const int stop = ctl[r * VRX_CTL_WORDS + VRX_CTL_STOP];
    vrx_theta_final_block(n_part, T, update, fix_sum, part + (int64_t)r * n_part * 2 * VRX_MAXT, prior1,
                          prior2, mu + r * T, sm + r * T, psi + r * 3 * T, kl_out + r, stop);
}

// stage 1 (shared theta): per-block partial sums of S1*GT_t and S2*GT_t over all (n,k).
// npiece != null: S has not been formed yet -- it is the in-order sum of the partial arrays the
// LDS-resident variant pass left in `ranges` (npiece[variant] of them; fused here to save a launch).
template <int TT>  // genotype classes: 3 exactly (no per-class branches), or VRX_MAXT = any T
__global__ __launch_bounds__(VRX_BLOCK) void vrx_theta_partial(
    int64_t NK, int T, double2* S, const uint16_t* __restrict__ npiece,
    const double2* __restrict__ ranges, int64_t n_virtual, const int32_t* __restrict__ vptr,
    const double* __restrict__ GT, double* __restrict__ part, VrxBatch B,
    int32_t* ctl, VrxElboRide E) {
    const int rb = blockIdx.y;
    const int Tn = TT == VRX_MAXT ? T : TT;
    // Launch-bound problems: the ELBO (and stop rule) of the PREVIOUS iteration is finalised here by
    // one extra block at the end of the grid instead of by a kernel of its own between two
    // iterations -- vrx_elbo_final is a one-block latency chain of ~4.5 us plus a kernel boundary, a
    // fifth of a c2-sized iteration.  Nothing that ran since that iteration's last kernel has
    // touched the model's state (the variant pass and this kernel write S and partial sums only),
    // and the next kernel that does, vrx_gt_update, starts after this one has ended and sees the
    // stop word; blocks of this launch that see it early just return.
    const int nb = (int)gridDim.x - E.on;
    if (E.on && (int)blockIdx.x == nb) {
        vrx_elbo_final_block(E.in.cell_part + (int64_t)rb * E.in.n_cell_part * 2, E.in.n_cell_part,
                             E.in.gt_part + (int64_t)rb * E.in.n_gt_part, E.in.n_gt_part,
                             E.in.th_part + (int64_t)rb * E.in.n_th_part, E.in.n_th_part,
                             E.in.elbo + rb * E.in.trace_stride, E.in.parts + rb * 4, E.rule,
                             ctl + rb * VRX_CTL_WORDS);
        return;
    }
    const int stop = ctl[rb * VRX_CTL_WORDS + VRX_CTL_STOP];
    const int64_t NKt = NK * B.R;
    double acc[2 * VRX_MAXT];
#pragma unroll
    for (int t = 0; t < 2 * VRX_MAXT; ++t) acc[t] = 0.0;
    // (the variant of an element without a 64-bit division per element, as in vrx_gt_update)
    const int64_t stride = (int64_t)nb * VRX_BLOCK;
    const int64_t step_n = stride / B.K;
    const int step_k = (int)(stride - step_n * B.K);
    const int64_t i0 = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    int64_t n = i0 / B.K;
    int k = (int)(i0 - n * B.K);
    // the first element's operands travel together with the stop word (one round trip, not two)
    const bool pre = npiece == nullptr && i0 < NK;
    double2 s_pre = make_double2(0.0, 0.0);
    double g_pre[VRX_MAXT];
    if (pre) {
        const int64_t j = B.R == 1 ? i0 : n * B.Kt + (int64_t)rb * B.K + k;
        s_pre = S[j];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < Tn) g_pre[t] = GT[j * T + t];
    }
    if (stop) return;
    for (int64_t i = i0; i < NK; i += stride, n += step_n, k += step_k) {
        if (k >= B.K) {
            k -= B.K;
            ++n;
        }
        const int64_t j = B.R == 1 ? i : n * B.Kt + (int64_t)rb * B.K + k;
        const bool first = pre && i == i0;
        double2 s;
        if (npiece && n_virtual > 0) {
            // virtual rows (TiledStream::virt): planar partial sums [slot][2 n + kind][Kt], the AD
            // row's into S1, the BD row's into S2, each over its own number of slots; SS = S1 + S2
            const double* P = reinterpret_cast<const double*>(ranges);
            const int64_t col = B.R == 1 ? k : (int64_t)rb * B.K + k;
            double sk[2];
#pragma unroll
            for (int kind = 0; kind < 2; ++kind) {
                // vptr != null: long rows are cut into pieces (heavy-tailed data); virtual row
                // 2n + kind is the pieces [vptr[.], vptr[. + 1]), whose terms vrx_fold_split has
                // summed beforehand (one wave per row and column: a long row has ~100 terms)
                const int64_t vr = 2 * n + kind;
                const int64_t v = vptr ? vptr[vr] : vr;
                const int np = vptr ? vptr[vr + 1] - (int)v : 1;
                double t = 0.0;
                if (np == 1) {
#ifdef VRX_PROBE_ONE_PLANE  // TIMING PROBE ONLY (scratch builds): as if the pass had left ONE plane per tile
                    const int nr = 1;
#else
                    const int nr = npiece[v];
#endif
                    const double* src = P + v * B.Kt + col;
                    for (int r0 = 0; r0 < nr; r0 += 8) {
                        double x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = r0 + u < nr ? src[(int64_t)(r0 + u) * n_virtual * B.Kt] : 0.0;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (r0 + u < nr) t += x[u];
                    }
                } else if (np > 1) {
                    // (a row cut into pieces: vrx_fold_split has left its whole sum in slot 0 of
                    //  its first piece)
                    t = P[v * B.Kt + col];
                }
                sk[kind] = t;
            }
            s = make_double2(sk[0], sk[0] + sk[1]);
            S[j] = s;
        } else if (npiece) {  // S is the in-order sum of the partial arrays that hold the variant
            const int n_range = npiece[n];
            s = make_double2(0.0, 0.0);
            for (int r0 = 0; r0 < n_range; r0 += 8) {  // (loads of 8 ranges in flight together)
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    v[u] = r0 + u < n_range ? ranges[(int64_t)(r0 + u) * NKt + j] : make_double2(0.0, 0.0);
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + u < n_range) {
                        s.x += v[u].x;
                        s.y += v[u].y;
                    }
            }
            S[j] = s;
        } else {
            s = first ? s_pre : S[j];
        }
        const double s1 = s.x, s2 = s.y - s.x;
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < Tn) {
                const double g = first ? g_pre[t] : GT[j * T + t];
                acc[t] += s1 * g;
                acc[VRX_MAXT + t] += s2 * g;
            }
    }
    block_sum_store<2 * VRX_MAXT>(acc, part + ((int64_t)rb * nb + blockIdx.x) * 2 * VRX_MAXT, VRX_MAXT, T);
}

// ASE mode: one theta row per variant (vireo_model.py:82,:177 axis=1).  Thread per variant.
__global__ __launch_bounds__(VRX_BLOCK) void vrx_theta_ase(int64_t N, int K, int T, int update,
                                                           int fix_sum,
                                                           const double2* __restrict__ S,
                                                           const double* __restrict__ GT,
                                                           const double* __restrict__ prior1,
                                                           const double* __restrict__ prior2,
                                                           int prior_rows, double* mu, double* sm,
                                                           double* psi, double* kl_part, VrxBatch B,
                                                           const int32_t* __restrict__ ctl) {
    const int rb = blockIdx.y;
    if (ctl[rb * VRX_CTL_WORDS + VRX_CTL_STOP]) return;
    mu += (int64_t)rb * N * T;
    sm += (int64_t)rb * N * T;
    psi += (int64_t)rb * 3 * N * T;
    const int64_t n = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    double kl[1] = {0.0};
    if (n < N) {
        double a1[VRX_MAXT], a2[VRX_MAXT];
#pragma unroll
        for (int t = 0; t < VRX_MAXT; ++t) a1[t] = a2[t] = 0.0;
        if (update)
            for (int k = 0; k < K; ++k) {
                const int64_t j = n * B.Kt + (int64_t)rb * K + k;
                const double2 s = S[j];
                const double s1 = s.x, s2 = s.y - s.x;
#pragma unroll
                for (int t = 0; t < VRX_MAXT; ++t)
                    if (t < T) {
                        const double g = GT[j * T + t];
                        a1[t] += s1 * g;
                        a2[t] += s2 * g;
                    }
            }
        const int64_t pr = prior_rows == 1 ? 0 : n;
        kl[0] = vrx_theta_row(T, update, fix_sum, a1, a2, prior1 + pr * T, prior2 + pr * T,
                              mu + n * T, sm + n * T, psi + n * T, psi + (N + n) * T,
                              psi + (2 * N + n) * T);
    }
    block_sum_store<1>(kl, kl_part + (int64_t)rb * gridDim.x + blockIdx.x);
}

// One entry of the cell pass's dense operand.  wform 0: W[n][k] = (W1, W2) interleaved, used
// with (ad, dp) pairs; wform 1: planar rows [Wa[n][0..K) | Wb[n][0..K)] with Wa = W1 + W2 (the
// factor of AD) and Wb = W2 (the factor of BD = DP - AD), used by the AD/BD stream (FORM 1).
__device__ __forceinline__ void vrx_store_w(double* W, int wform, int64_t n, int k, int K,
                                            double w1, double w2, double wa) {
    if (wform) {
        W[n * 2 * K + k] = wa;
        W[n * 2 * K + K + k] = w2;
    } else {
        reinterpret_cast<double2*>(W)[n * K + k] = make_double2(w1, w2);
    }
}

// ------------------------------------------------------------------------------------
// genotype posterior  (Vireo.update_GT_prob, vireo_model.py:204-219) fused with the
// W1/W2 tables of the cell pass and the KL(GT || prior) partial of get_ELBO (:238).
// Thread per (variant, donor).  learn == 0: GT is fixed, only W (and the KL) are derived.
// gt_mode: 0 uniform prior (scalar log 1/T), 1 one (K,T) slab, 2 full (N,K,T).
// ------------------------------------------------------------------------------------
// The shared-theta finalisation folded into this kernel (on != 0; never in ASE mode): every block
// sums the stage-1 partials of vrx_theta_partial in the same fixed order -- identical values in
// every block -- and takes psi from its own shared memory; block 0 also writes beta_mu,
// beta_sum, psi and KL_theta to global memory.  One launch (vrx_theta_final: one block of pure
// latency) and one kernel boundary less per iteration.
struct VrxThetaFuse {  // by value
    int on, n_part, fix_sum;
    const double *part, *prior1, *prior2;
    double *mu, *sm, *psi, *kl_out;
};

// Grid-stride over the (variant, donor) pairs: gridDim.x blocks, gridDim.x KL partials.
template <int TT>  // genotype classes: 3 exactly (no per-class branches), or VRX_MAXT = any T
__global__ __launch_bounds__(VRX_BLOCK) void vrx_gt_update(
    int64_t NK, int K, int T, int learn, int ase, int64_t N, const double2* __restrict__ S,
    const double* psi, const double* __restrict__ logq, int gt_mode, double logq_uni,
    double* __restrict__ GT, double* __restrict__ W, int wform, double* __restrict__ kl_part,
    VrxThetaFuse F, VrxBatch B, const int32_t* __restrict__ ctl) {
#pragma clang fp contract(off)
    const int rb = blockIdx.y;
    const int Tn = TT == VRX_MAXT ? T : TT;
    const int stop = ctl[rb * VRX_CTL_WORDS + VRX_CTL_STOP];
    __shared__ double th_tot[2 * VRX_MAXT], th_ms[VRX_MAXT][2], th_psi[3 * VRX_MAXT], th_lg[VRX_MAXT][6];
    const double* psi_r = psi + (int64_t)rb * 3 * (ase ? N : 1) * T;
    // (variant, donor) of this thread's elements without a 64-bit division per element: one at
    // the start, then steps of the grid stride
    const int64_t stride = (int64_t)gridDim.x * VRX_BLOCK;
    const int64_t step_n = stride / K;
    const int step_k = (int)(stride - step_n * K);
    const int64_t i0 = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    int64_t n = i0 / K;
    int k = (int)(i0 - n * K);
    // Everything the kernel reads first is requested before the stop word is tested (a small
    // problem's kernel is a chain of ~1 us memory round trips: this makes them one): the
    // first element's S, and for the fused theta finalisation the stage-1 partials and the
    // current / prior Beta parameters.
    const bool pre = learn && i0 < NK;
    double2 s_pre = make_double2(0.0, 0.0);
    if (pre) s_pre = S[n * B.Kt + (int64_t)rb * K + k];
    double th_acc[2 * VRX_MAXT], th_in[4] = {0.0, 0.0, 0.0, 0.0};
    if (F.on) {
        const double* part = F.part + (int64_t)rb * F.n_part * 2 * VRX_MAXT;
#pragma unroll
        for (int t = 0; t < 2 * VRX_MAXT; ++t) th_acc[t] = 0.0;
        for (int b = threadIdx.x; b < F.n_part; b += VRX_BLOCK)
#pragma unroll
            for (int t = 0; t < 2 * VRX_MAXT; ++t)
                if (t % VRX_MAXT < T) th_acc[t] += part[(int64_t)b * 2 * VRX_MAXT + t];
        if ((int)threadIdx.x < T) {
            th_in[0] = F.mu[rb * T + threadIdx.x];
            th_in[1] = F.sm[rb * T + threadIdx.x];
            th_in[2] = F.prior1[threadIdx.x];
            th_in[3] = F.prior2[threadIdx.x];
        }
    }
    if (stop) return;
    if (F.on) {
        // vrx_theta_final_block's arithmetic, identical in every block (same sums in the same
        // order); only block 0 needs KL_theta -- its log-gammas wait until after the element loop
        block_sum_store<2 * VRX_MAXT>(th_acc, th_tot, VRX_MAXT, T);
        __syncthreads();
        if ((int)threadIdx.x < T) {
            const int t = threadIdx.x;
            const double t1 = th_in[2] + th_tot[t];
            const double t2 = th_in[3] + th_tot[VRX_MAXT + t];
            const double mn = t1 / (t1 + t2);
            const double sn = F.fix_sum ? th_in[1] : t1 + t2;
            th_ms[t][0] = mn;
            th_ms[t][1] = sn;
            if (blockIdx.x == 0) {  // the one writer of the new state
                F.mu[rb * T + t] = mn;
                F.sm[rb * T + t] = sn;
            }
        }
        __syncthreads();
        if ((int)threadIdx.x < 3 * T) {
            const int t = threadIdx.x / 3, jj = threadIdx.x % 3;
            const double mn = th_ms[t][0], sn = th_ms[t][1];
            const double s1 = mn * sn, s2 = (1.0 - mn) * sn;
            const double d = vrx_digamma(jj == 0 ? s1 : jj == 1 ? s2 : s1 + s2);
            th_psi[jj * T + t] = d;
            if (blockIdx.x == 0) F.psi[rb * 3 * T + jj * T + t] = d;
        }
        __syncthreads();
        psi_r = th_psi;
    }
    double kl[1] = {0.0};
    for (int64_t i = i0; i < NK; i += stride, n += step_n, k += step_k) {
        if (k >= K) {
            k -= K;
            ++n;
        }
        const bool first = pre && i == i0;
        const int64_t j = n * B.Kt + (int64_t)rb * K + k;  // this restart's column of S / GT / W
        const int64_t rows = ase ? N : 1, pr = ase ? n : 0;
        const double* pb = psi_r;
        const double* p1 = pb + pr * T;
        const double* p2 = pb + (rows + pr) * T;
        const double* ps = pb + (2 * rows + pr) * T;
        double g[VRX_MAXT], lq[VRX_MAXT];
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < Tn)
                lq[t] = gt_mode == 0 ? logq_uni
                                     : (gt_mode == 1 ? logq[k * T + t] : logq[i * T + t]);
        if (learn) {
            const double2 s = first ? s_pre : S[j];
            const double s1 = s.x, ss = s.y, s2 = ss - s1;
            double L[VRX_MAXT];
            double mx = -__builtin_inf();
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < Tn) {
                    L[t] = (s1 * p1[t] + s2 * p2[t] - ss * ps[t]) + lq[t];
                    mx = fmax(mx, L[t]);
                }
            double sum = 0.0;
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < Tn) {
                    L[t] -= mx;
                    g[t] = exp(L[t]);
                    sum += g[t];
                }
            const double lsum = log(sum);
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < Tn) {
                    g[t] = g[t] / sum;
                    GT[j * T + t] = g[t];
                    if (g[t] > 0.0) kl[0] += g[t] * ((L[t] - lsum) - lq[t]);
                }
        } else {
#pragma unroll
            for (int t = 0; t < TT; ++t)
                if (t < Tn) {
                    g[t] = GT[j * T + t];
                    if (g[t] > 0.0) kl[0] += g[t] * (log(g[t]) - lq[t]);
                }
        }
        double w1 = 0.0, w2 = 0.0, wa = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t)
            if (t < Tn) {
                w1 += g[t] * (p1[t] - p2[t]);
                w2 += g[t] * (p2[t] - ps[t]);
                wa += g[t] * (p1[t] - ps[t]);
            }
        vrx_store_w(W, wform, n, rb * K + k, B.Kt, w1, w2, wa);
    }
    if (F.on && blockIdx.x == 0) {  // KL_theta: 6 log-gammas per class on wave 1, combined below
        const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
        if (wv == 1 && ln < 6 * T) {
            const int t = ln / 6, jj = ln % 6;
            const double mn = th_ms[t][0], sn = th_ms[t][1];
            const double s1 = mn * sn, s2 = (1.0 - mn) * sn, q1 = F.prior1[t], q2 = F.prior2[t];
            th_lg[t][jj] = lgamma(jj == 0 ? q1 : jj == 1 ? q2 : jj == 2 ? q1 + q2 : jj == 3 ? s1 : jj == 4 ? s2 : s1 + s2);
        }
    }
    block_sum_store<1>(kl, kl_part + (int64_t)rb * gridDim.x + blockIdx.x);  // (synchronises the block)
    if (F.on && blockIdx.x == 0 && threadIdx.x == 0) {
        double klt = 0.0;  // same order as vrx_theta_final_block
        for (int t = 0; t < T; ++t) {
            const double mn = th_ms[t][0], sn = th_ms[t][1];
            const double s1 = mn * sn, s2 = (1.0 - mn) * sn, q1 = F.prior1[t], q2 = F.prior2[t];
            const double d1 = th_psi[t], d2 = th_psi[T + t], ds = th_psi[2 * T + t];
            const double cq = (th_lg[t][0] + th_lg[t][1] - th_lg[t][2]) - (q1 - 1.0) * d1 -
                              (q2 - 1.0) * d2 + ((q1 + q2) - 2.0) * ds;
            const double cp = (th_lg[t][3] + th_lg[t][4] - th_lg[t][5]) - (s1 - 1.0) * d1 -
                              (s2 - 1.0) * d2 + ((s1 + s2) - 2.0) * ds;
            klt += cq - cp;
        }
        F.kl_out[rb] = klt;
    }
}

// ------------------------------------------------------------------------------------
// doublet tables  (add_doublet_GT, vireoSNP/utils/vireo_doublet.py:105-136, folded into the
// W1/W2 tables of the cell pass).  Column c < K is donor c (the T genotype classes, zero
// mixed classes); column K + j is the j-th donor pair (a < b) over T + T(T-1)/2 classes:
//   both[t]       = p_t q_t                                   (same genotype)
//   both[T + m]   = p_{g1} q_{g2} + p_{g2} q_{g1}             (m-th genotype pair g1 < g2)
// normalised over the classes.  The N x C x (T + T(T-1)/2) tensor the reference
// materialises (653 MB at N=100k, K=16) never exists: each thread forms its classes in
// registers and writes W[n][c] = (sum_g both_g (psi1_g - psi2_g), sum_g both_g (psi2_g - psis_g)).
// psi*: [rows][G] with G = T + T(T-1)/2 classes (rows = N in ASE mode, else 1).  T <= 3.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_doublet_w(
    int64_t N, int K, int T, int C, int ase, const double* __restrict__ GT,
    const int2* __restrict__ pair, const double* __restrict__ psi1,
    const double* __restrict__ psi2, const double* __restrict__ psis, double* __restrict__ W,
    int wform) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= N * C) return;
    const int64_t n = i / C;
    const int c = (int)(i - n * C);
    const int G = T + T * (T - 1) / 2;
    const int64_t pr = ase ? n : 0;
    const double* p1 = psi1 + pr * G;
    const double* p2 = psi2 + pr * G;
    const double* ps = psis + pr * G;
    double both[6];
#pragma unroll
    for (int g = 0; g < 6; ++g) both[g] = 0.0;
    if (c < K) {
        for (int t = 0; t < T; ++t) both[t] = GT[(n * K + c) * T + t];
    } else {
        const int2 ab = pair[c - K];
        const double* p = GT + (n * K + ab.x) * T;
        const double* q = GT + (n * K + ab.y) * T;
        double sum = 0.0;
        for (int t = 0; t < T; ++t) {
            both[t] = p[t] * q[t];
            sum += both[t];
        }
        int m = T;
        for (int g1 = 0; g1 < T; ++g1)
            for (int g2 = g1 + 1; g2 < T; ++g2) {
                both[m] = p[g1] * q[g2] + p[g2] * q[g1];
                sum += both[m];
                ++m;
            }
        for (int g = 0; g < G; ++g) both[g] = both[g] / sum;
    }
    double w1 = 0.0, w2 = 0.0, wa = 0.0;
    for (int g = 0; g < G; ++g) {
        w1 += both[g] * (p1[g] - p2[g]);
        w2 += both[g] * (p2[g] - ps[g]);
        wa += both[g] * (p1[g] - ps[g]);
    }
    vrx_store_w(W, wform, n, c, C, w1, w2, wa);
}

// ------------------------------------------------------------------------------------
// BinomMixtureVB theta  (bmm_model.py:133-144) fused with the digamma tables of
// get_E_logLik (:118-130) and the KL_theta partial of get_ELBO (:166-172).
// Thread per (variant, clone).  update == 0: derive W / KL from the current beta only.
// ------------------------------------------------------------------------------------
// npiece != null: S has not been formed yet -- it is the in-order sum of the npiece[variant]
// partial arrays the LDS-resident variant pass left in `ranges` (as in vrx_theta_partial; clone
// mode has few variants and long rows: c5 cuts its one tile into 196 pieces).
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bmm_theta(int64_t NK, int update, int fix_sum,
                                                           double2* __restrict__ S,
                                                           const uint16_t* __restrict__ npiece,
                                                           const double2* __restrict__ ranges,
                                                           const double* __restrict__ prior1,
                                                           const double* __restrict__ prior2,
                                                           int prior_full, double* mu, double* sm,
                                                           double* __restrict__ W, int K, int wform,
                                                           double* __restrict__ kl_part, VrxBatch B,
                                                           int32_t* ctl, VrxElboRide E) {
#pragma clang fp contract(off)
    const int rb = blockIdx.y;
    // (the previous iteration's ELBO + stop rule as one extra block, as in vrx_theta_partial; its
    //  KL_theta partials sit in the half of the buffer this launch does not write)
    const int nb = (int)gridDim.x - E.on;
    if (E.on && (int)blockIdx.x == nb) {
        vrx_elbo_final_block(E.in.cell_part + (int64_t)rb * E.in.n_cell_part * 2, E.in.n_cell_part,
                             E.in.gt_part + (int64_t)rb * E.in.n_gt_part, E.in.n_gt_part,
                             E.in.th_part + (int64_t)rb * E.in.n_th_part, E.in.n_th_part,
                             E.in.elbo + rb * E.in.trace_stride, E.in.parts + rb * 4, E.rule,
                             ctl + rb * VRX_CTL_WORDS);
        return;
    }
    if (ctl[rb * VRX_CTL_WORDS + VRX_CTL_STOP]) return;
    mu += (int64_t)rb * NK;
    sm += (int64_t)rb * NK;
    // npiece: LP = 16 lanes per element, lane `sub` adds the pieces sub, sub + 16, ... (all its
    // loads in flight at once), then a fixed butterfly over the 16 lanes: a chain of 196 pieces
    // is 13 loads deep instead of 196
    const int LP = npiece ? 16 : 1;
    const int64_t gt = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    const int64_t i = gt / LP;
    const int sub = (int)(gt - i * LP);
    double kl[1] = {0.0};
    const bool live = i < NK;
    double2 a = make_double2(0.0, 0.0);
    if (update && npiece) {  // (whole 16-lane groups take this branch together)
        const int64_t ic = live ? i : 0;
        const int64_t j = vrx_col(B, ic, rb);
        const int n_range = live ? npiece[ic / K] : 0;
        const int64_t NKt = NK * B.R;
        for (int r0 = sub; r0 < n_range; r0 += 16 * 16) {
            double2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = r0 + u * 16 < n_range ? ranges[(int64_t)(r0 + u * 16) * NKt + j] : make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (r0 + u * 16 < n_range) {
                    a.x += v[u].x;
                    a.y += v[u].y;
                }
        }
        a.x = vrx_group_sum<16>(a.x);
        a.y = vrx_group_sum<16>(a.y);
        if (live && sub == 0) S[j] = a;
    }
    if (live && sub == 0) {
        const double q1 = prior1[prior_full ? i : 0], q2 = prior2[prior_full ? i : 0];
        double m = mu[i], s = sm[i];
        if (update) {
            if (!npiece) a = S[vrx_col(B, i, rb)];
            const double t1 = a.x + q1;
            const double t2 = (a.y - a.x) + q2;
            m = t1 / (t1 + t2);
            if (!fix_sum) s = t1 + t2;
            mu[i] = m;
            sm[i] = s;
        }
        const double s1 = m * s, s2 = (1.0 - m) * s;
        const double d1 = vrx_digamma(s1), d2 = vrx_digamma(s2), ds = vrx_digamma(s1 + s2);
        vrx_store_w(W, wform, i / K, rb * K + (int)(i % K), B.Kt, d1 - d2, d2 - ds, d1 - ds);
        kl[0] = vrx_beta_kl(s1, s2, q1, q2, d1, d2, ds);
    }
    block_sum_store<1>(kl, kl_part + (int64_t)rb * nb + blockIdx.x);
}

// ------------------------------------------------------------------------------------
// ELBO  = LB_p - KL_ID - KL_GT - KL_theta  (vireo_model.py:247-248, bmm_model.py:175)
// One block; each term is the fixed-order sum of a partial array.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_elbo_final(VrxElboIn e, VrxStopRule rule, int32_t* ctl) {
    const int r = blockIdx.x;  // one block per restart of the batch
    vrx_elbo_final_block(e.cell_part + (int64_t)r * e.n_cell_part * 2, e.n_cell_part,
                         e.gt_part + (int64_t)r * e.n_gt_part, e.n_gt_part,
                         e.th_part + (int64_t)r * e.n_th_part, e.n_th_part,
                         e.elbo + r * e.trace_stride, e.parts + r * 4, rule, ctl + r * VRX_CTL_WORDS);
}

// ------------------------------------------------------------------------------------
// cell posterior  (Vireo.update_ID_prob vireo_model.py:198-199, bmm_model.py:153-154) fused
// with the LB_p and KL(ID || prior) partials of get_ELBO (vireo_model.py:236-237).
// A KP-lane group per cell; K > KP loops.  update == 0: ID_prob is left alone and only the
// ELBO partials are formed from the stored ID_prob (get_ELBO on user-supplied state).
// id_mode: 0 uniform prior, 1 one row of K, 2 full (M,K).
// npiece != null: logLik_ID has not been formed yet -- it is the in-order sum of the npiece[cell]
// partial arrays the LDS-resident cell pass left in `ranges` (fused here to save a launch;
// every lane sums, stores and later re-reads only its own columns).
// ------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(VRX_BLOCK) void vrx_cell_softmax(
    int64_t M, int K, int update, double* LID, const uint16_t* __restrict__ npiece,
    const double* __restrict__ ranges, const int32_t* __restrict__ vptr, int64_t n_vrows,
    const double* __restrict__ logq, int id_mode, double logq_uni, double* __restrict__ ID,
    double* __restrict__ part, VrxBatch B, const int32_t* __restrict__ ctl) {
    const int rb = blockIdx.y;
    const int stop = ctl[rb * VRX_CTL_WORDS + VRX_CTL_STOP];
    const int kl = threadIdx.x % KP;
    double acc[2] = {0.0, 0.0};
    // Grid-stride over the cells: the host caps the grid (8 blocks per CU), a block takes the cell
    // groups b, b + grid, ... and ends in ONE block sum -- large M paid a prologue (stop word, first
    // load) and a block reduction per 256 threads, three waves of blocks deep at c5.  Every thread
    // runs every sweep (`live` guards the work: the group reductions want whole groups).
    const int64_t cpg = (int64_t)gridDim.x * VRX_BLOCK / KP;  // cells per sweep of the grid
    const int64_t cell0 = ((int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x) / KP;
    // (the lane's first column travels together with the stop word: one round trip, not two)
    const bool pre0 = cell0 < M && npiece == nullptr && kl < K;
    const double L_pre0 = pre0 ? LID[cell0 * (int64_t)B.Kt + (int64_t)rb * K + kl] : 0.0;
    if (stop) return;
    for (int64_t cell = cell0; cell - cell0 + (int64_t)blockIdx.x * VRX_BLOCK / KP < M; cell += cpg) {
    const bool live = cell < M;
    const int64_t row0 = (live ? cell : 0) * (int64_t)B.Kt + (int64_t)rb * K;  // this restart's K columns
    double* Lr = LID + row0;
    const bool pre = pre0 && cell == cell0;
    const double L_pre = L_pre0;
    // npiece != null: logLik_ID still sits in the pass's partial arrays [slot][piece][Kt] (n_vrows
    // pieces; vptr == null: piece = cell).  A cell that was cut into several pieces (heavy-tailed
    // data) has been summed by vrx_fold_split.
    const int64_t pv0 = live && npiece ? (vptr ? vptr[cell] : cell) : 0;
    const int np = live && npiece ? (vptr ? vptr[cell + 1] - (int)pv0 : 1) : 0;
    const int64_t col0 = (int64_t)rb * K;
    if (np == 1) {
#ifdef VRX_PROBE_ONE_PLANE  // TIMING PROBE ONLY (scratch builds)
        const int n_range = 1;
#else
        const int n_range = npiece[pv0];
#endif
        for (int k = kl; k < K; k += KP) {
            // the loads of 8 ranges are issued together (one memory round trip instead of 8);
            // the additions keep the range order
            double t = 0.0;
            const double* src = ranges + pv0 * B.Kt + col0 + k;
            const int64_t stride = n_vrows * B.Kt;
            for (int r0 = 0; r0 < n_range; r0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = r0 + u < n_range ? src[(int64_t)(r0 + u) * stride] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (r0 + u < n_range) t += v[u];
            }
            Lr[k] = t;
        }
    } else if (np > 1) {
        // (a cell cut into pieces: vrx_fold_split has left its whole sum in slot 0 of its first piece)
        for (int k = kl; k < K; k += KP) Lr[k] = ranges[pv0 * B.Kt + col0 + k];
    }
    const double* qr = id_mode == 2 ? logq + (live ? cell : 0) * (int64_t)K : logq;
    double mx = -__builtin_inf();
    if (live)
        for (int k = kl; k < K; k += KP) mx = fmax(mx, (pre && k == kl ? L_pre : Lr[k]) + (id_mode ? qr[k] : logq_uni));
    mx = vrx_group_max<KP>(mx);
    double sum = 0.0, e_first = 0.0;  // (the lane's first exponential is reused below: same argument)
    if (live && update)
        for (int k = kl; k < K; k += KP) {
            const double e = exp((pre && k == kl ? L_pre : Lr[k]) + (id_mode ? qr[k] : logq_uni) - mx);
            if (k == kl) e_first = e;
            sum += e;
        }
    sum = vrx_group_sum<KP>(sum);
    if (live) {
        const double lsum = update ? log(sum) : 0.0;
        double* Ir = ID + row0;
        for (int k = kl; k < K; k += KP) {
            const double L = pre && k == kl ? L_pre : Lr[k];
            const double lq = id_mode ? qr[k] : logq_uni;
            double p, lp;
            if (update) {
                const double x = (L + lq) - mx;
                p = (k == kl ? e_first : exp(x)) / sum;
                lp = x - lsum;
                Ir[k] = p;
            } else {
                p = Ir[k];
                lp = log(p);
            }
            acc[0] += L * p;
            if (p > 0.0) acc[1] += p * (lp - lq);
        }
    }
    }
    block_sum_store<2>(acc, part + ((int64_t)rb * gridDim.x + blockIdx.x) * 2);
}

// ------------------------------------------------------------------------------------
// priors: row-normalised logs  (scipy.stats.entropy normalises q; np.log(prior) in the
// softmax is shift-invariant, so one table serves both uses)
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_log_rows(int64_t rows, int C,
                                                          const double* __restrict__ p,
                                                          double* __restrict__ logq) {
    const int64_t r = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (r >= rows) return;
    double s = 0.0;
    for (int c = 0; c < C; ++c) s += p[r * C + c];
    for (int c = 0; c < C; ++c) logq[r * C + c] = log(p[r * C + c] / s);
}

// ------------------------------------------------------------------------------------
// normalize(X) = X / X.sum(axis=-1, keepdims=True)  (vireo_base.py:44-55) of a freshly drawn
// initial state, on the device.  The row sum follows NumPy's pairwise summation (the add-reduce
// inner loop over a contiguous axis: a plain loop below 8 terms, eight running sums combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail up to 128 terms), so the result
// is bit-identical to the host's.  Thread per row; C <= 128.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_normalize_rows(int64_t rows, int C,
                                                                double* __restrict__ X) {
#pragma clang fp contract(off)
    const int64_t r = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (r >= rows) return;
    double* a = X + r * C;
    double s;
    if (C < 8) {
        s = 0.0;
        for (int i = 0; i < C; ++i) s += a[i];
    } else {
        double q[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = a[j];
        int i = 8;
        for (; i < C - (C % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] += a[i + j];
        s = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        for (; i < C; ++i) s += a[i];
    }
    for (int i = 0; i < C; ++i) a[i] = a[i] / s;
}

// ------------------------------------------------------------------------------------
// binomial-coefficient constant  (get_binom_coeff, vireo_base.py:7-22)
//   float32( min( log C(dp, ad), 700 ) ) for every entry with dp > 0; summed on the host in
//   NumPy's float32 pairwise order (vrx_problem_binom_const)
// ------------------------------------------------------------------------------------
// per-entry terms in storage order; entries with dp == 0 (never indexed by the reference's
// DP > 0 mask) are marked with a NaN
// NumPy's float32 sum of one full iterator buffer (8192 elements): pairwise_sum splits in
// halves down to blocks of 128, a block is eight running sums combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) (numpy/_core/src/umath/loops_utils.h.src; the host
// restatement is np_pairwise_sum_f32 in vrx_host.cpp).  One wave per buffer: lane = block of
// 128, then the binary tree over the 64 block sums.  Also flags NaN marks (dp == 0 entries).
__global__ __launch_bounds__(64) void vrx_np_chunk_sums_f32(int64_t n_chunk, const float* __restrict__ a,
                                                            float* __restrict__ sums, int32_t* has_nan) {
#pragma clang fp contract(off)
    __shared__ float sh[64];
    const int64_t c = blockIdx.x;
    if (c >= n_chunk) return;
    const float* p = a + c * 8192 + (int64_t)threadIdx.x * 128;
    float r[8];
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r[j] = p[j];
        bad = bad || r[j] != r[j];
    }
    for (int i = 8; i < 128; i += 8)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = p[i + j];
            bad = bad || v != v;
            r[j] += v;
        }
    sh[threadIdx.x] = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    __syncthreads();
    for (int w = 1; w < 64; w <<= 1) {
        if ((threadIdx.x & (2 * w - 1)) == 0) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[c] = sh[0];
    if (bad) atomicOr(has_nan, 1);
}

template <int FMT>
__global__ __launch_bounds__(VRX_BLOCK) void vrx_binom_terms(int64_t nnz,
                                                             const uint32_t* __restrict__ ent,
                                                             float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    uint32_t id;
    int ad, dp;
    vrx_unpack<FMT>(vrx_load_words<FMT>(ent, e, true), id, ad, dp);
    float t = __builtin_nanf("");
    if (dp > 0) {
        const double n = (double)dp, k = (double)ad;
        double c = lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
        if (c > 700.0) c = 700.0;
        t = (float)c;
    }
    out[e] = t;
}
