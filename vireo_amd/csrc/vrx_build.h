// Device-side construction of a problem (f4): from the merged CSC arrays to both orientations and
// their tiled entry streams, in HIP kernels.  The reference's counterpart is
// `mmread(...).tocsc()` plus the implicit CSR<->CSC conversions inside SciPy's products
// (vireoSNP/utils/io_utils.py:57, vireo_model.py:167-170); the host builder of vrx_engine.hip is
// the specification: every stream built here is bit-identical to the one it builds
// (tests/test_gpu_parity.py::test_device_builder_equals_host_builder).
//
//   validate        thread per entry: column of the entry (binary search in colptr), row order,
//                   range, sign; per-cell count of covered variants; largest count
//   transpose       stable radix sort of (variant, entry id) pairs (hipCUB): cells stay
//                   increasing inside a variant row; row pointers by binary search
//   pack            the 4 / 8 / 12-byte entry arrays of both orientations
//   tiled streams   count: thread per (wave, slab, tile position): the segment of that row in
//                          that slab (two binary searches), its entry count (AD/BD chunks in
//                          form 1), the round's length by a 16-lane max
//                   offsets: thread per wave: running sum of the padded round lengths
//                   fill:  thread per (wave, slab, round, pair of lane groups): the parity /
//                          half pairing of build_tiled, emitted straight to the words'
//                          positions (two running counters per group, no temporary lists)
// Row pieces, their sort by length and the snake order (O(rows) work) stay on the host.
#pragma once

#include <hipcub/hipcub.hpp>
#include "vrx_balance.h"

#include "vrx_common.h"
#include "vrx_kernels.h"

struct VrxTileArgs {  // one orientation's tiled-stream geometry, by value to the kernels
    const int64_t* ptr;
    const int32_t* idx;
    const int2* val;
    const int32_t *rowmap, *vptr, *vrow_row;
    int RW, NR, G, U, n_slab, slab_rows, form, PH, bit_shift, pairing, xor_partner;
    uint32_t f1_base, pad_word;
    int64_t n_wave;
#ifdef VRX_CAP_PROBE
    int cap;  // TIMING PROBE ONLY (scratch builds, -DVRX_CAP_PROBE): words kept per (row, slab); the rest is DROPPED
#endif
};

// ------------------------------------------------------------------------------------
// Virtual rows of the variant pass (TiledStream::virt).  The AD/BD variant pass needs two sums
// per (variant, column): S1 over the AD counts and S2 over the BD counts of the variant's cells.
// As 2N single-sum rows -- row 2n = the AD counts of variant n, row 2n + 1 its BD counts --
// against the operand read as DOUBLE rows (cells 2j, 2j + 1 = the two 128-B halves of one 256-B
// row) it is exactly the AD/BD cell pass (FORM 1): an entry of double row j with
// (ad', dp') = (count at cell 2j, count at cell 2j + count at cell 2j + 1) gives a word against
// the first half for cell 2j and one against the second half for cell 2j + 1.  One accumulator
// per row instead of two, so the tiles are three times as tall as the two-phase variant pass's.
// These kernels derive that matrix from the variant-major arrays (thread per variant).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_virt_count(int64_t n_var, const int64_t* __restrict__ rptr,
                                                            const int32_t* __restrict__ ridx,
                                                            const int2* __restrict__ rval,
                                                            int64_t* __restrict__ cnt) {
    const int64_t n = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (n >= n_var) return;
    int64_t ca = 0, cb = 0;
    int32_t ja = -1, jb = -1;  // last double row counted for the AD / BD row
    for (int64_t e = rptr[n]; e < rptr[n + 1]; ++e) {
        const int32_t j = ridx[e] >> 1;
        const int2 x = rval[e];
        if (x.x != 0 && j != ja) ++ca, ja = j;   // (BD = DP - AD may be negative on bad input:
        if (x.y - x.x != 0 && j != jb) ++cb, jb = j;  //  the reference subtracts all the same)
    }
    cnt[2 * n] = ca;
    cnt[2 * n + 1] = cb;
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_virt_fill(int64_t n_var, const int64_t* __restrict__ rptr,
                                                           const int32_t* __restrict__ ridx,
                                                           const int2* __restrict__ rval,
                                                           const int64_t* __restrict__ vptr2,
                                                           int32_t* __restrict__ vidx, int2* __restrict__ vval) {
    const int64_t n = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (n >= n_var) return;
    int64_t oa = vptr2[2 * n] - 1, ob = vptr2[2 * n + 1] - 1;  // last written position of each row
    int32_t ja = -1, jb = -1;
    for (int64_t e = rptr[n]; e < rptr[n + 1]; ++e) {
        const int32_t c = ridx[e], j = c >> 1;
        const int2 x = rval[e];
        const int a = x.x, b = x.y - x.x;
        if (a != 0) {
            if (j != ja) {
                ++oa, ja = j;
                vidx[oa] = j;
                vval[oa] = make_int2(0, 0);
            }
            if (c & 1) vval[oa].y += a; else vval[oa] = make_int2(a, vval[oa].y + a);
        }
        if (b != 0) {
            if (j != jb) {
                ++ob, jb = j;
                vidx[ob] = j;
                vval[ob] = make_int2(0, 0);
            }
            if (c & 1) vval[ob].y += b; else vval[ob] = make_int2(b, vval[ob].y + b);
        }
    }
}

// number of FORM 1 entries a count becomes (build_tiled's push_value: top three significant bits
// at a time)
__device__ __forceinline__ int vrx_chunks(int64_t v) {
    int n = 0;
    while (v != 0) {
        const uint64_t mag = (uint64_t)(v < 0 ? -v : v);
        const int len = 64 - __clzll((long long)mag), sh = len > 3 ? len - 3 : 0;
        const int64_t c = (int64_t)((mag >> sh) << sh);
        v -= v < 0 ? -c : c;
        ++n;
    }
    return n;
}

__device__ __forceinline__ int64_t vrx_lower_bound(const int32_t* a, int64_t lo, int64_t hi, int64_t key) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// ---- validation -----------------------------------------------------------------------------
// status[0] = smallest offending column (INT_MAX if none), status[1] = its kind, status[2] = max count
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_validate(
    int64_t nnz, int64_t n_var, int64_t n_cell, const int64_t* __restrict__ colptr,
    const int32_t* __restrict__ rowidx, const int32_t* __restrict__ ad, const int32_t* __restrict__ dp,
    int32_t* __restrict__ ecol, int32_t* __restrict__ n_vars, int32_t* status) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    // column of entry e: the last c with colptr[c] <= e
    int64_t lo = 0, hi = n_cell;
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (colptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int32_t c = (int32_t)lo;
    ecol[e] = c;
    const int32_t r = rowidx[e], a = ad[e], d = dp[e];
    int kind = 0;
    if (r < 0 || r >= n_var || (e > colptr[c] && rowidx[e - 1] >= r)) kind = 2;
    else if (a < 0 || d < 0) kind = 3;
    if (kind) {
        if (atomicMin(&status[0], c) > c) status[1] = kind;  // (a benign race picks some kind)
    }
    atomicMax(&status[2], a > d ? a : d);
    if (d > 0) atomicAdd(&n_vars[c], 1);
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_iota_keys(int64_t nnz, const int32_t* __restrict__ rowidx,
                                                                 uint32_t* __restrict__ keys,
                                                                 uint32_t* __restrict__ vals) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    keys[e] = (uint32_t)rowidx[e];
    vals[e] = (uint32_t)e;
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_rptr(int64_t n_var, int64_t nnz,
                                                            const uint32_t* __restrict__ keys_sorted,
                                                            int64_t* __restrict__ rptr) {
    const int64_t r = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (r > n_var) return;
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)keys_sorted[mid] < r)
            lo = mid + 1;
        else
            hi = mid;
    }
    rptr[r] = lo;
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_gather(int64_t nnz, const uint32_t* __restrict__ perm,
                                                              const int32_t* __restrict__ ecol,
                                                              const int32_t* __restrict__ ad,
                                                              const int32_t* __restrict__ dp,
                                                              int32_t* __restrict__ ridx, int2* __restrict__ rval,
                                                              int2* __restrict__ cval) {
    const int64_t q = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (q >= nnz) return;
    const uint32_t e = perm[q];
    ridx[q] = ecol[e];
    rval[q] = make_int2(ad[e], dp[e]);
    cval[q] = make_int2(ad[q], dp[q]);
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_pack(int64_t nnz, int fmt, const int32_t* __restrict__ idx,
                                                            const int2* __restrict__ val, uint32_t* __restrict__ ent) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    const uint32_t id = (uint32_t)idx[e], a = (uint32_t)val[e].x, d = (uint32_t)val[e].y;
    if (fmt == VRX_FMT_P32) {
        ent[e] = (id << 12) | (a << 6) | d;
    } else if (fmt == VRX_FMT_P64) {
        ent[e * 2] = id;
        ent[e * 2 + 1] = a | (d << 16);
    } else {
        ent[e * 3] = id;
        ent[e * 3 + 1] = a;
        ent[e * 3 + 2] = d;
    }
}

// ---- balanced slabs: relabel the contracted indices of every row by its tile's permutation -------
// key[e] = row << pbits | posmap[tile_of_row[row]][idx[e]]  (row by binary search in ptr), val[e] = e
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_relabel(int64_t nnz, int64_t n_rows, int64_t n_contract,
                                                               const int64_t* __restrict__ ptr,
                                                               const int32_t* __restrict__ idx,
                                                               const int32_t* __restrict__ tile_of_row,
                                                               const int32_t* __restrict__ posmap, int pbits,
                                                               uint64_t* __restrict__ keys,
                                                               uint32_t* __restrict__ vals) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    int64_t lo = 0, hi = n_rows;  // the last row with ptr[row] <= e
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int32_t t = tile_of_row[lo];
    const uint32_t pos = t >= 0 ? (uint32_t)posmap[(int64_t)t * n_contract + idx[e]] : (uint32_t)idx[e];
    keys[e] = ((uint64_t)lo << pbits) | pos;
    vals[e] = (uint32_t)e;
}

// Rows that were cut into pieces, as rows of their own: entry k of a row with P pieces belongs to piece
// k % P and is that piece's entry k / P (`pptr` = the pieces' row pointer, from the host).  Balanced slabs
// relabel the entries of a piece by the permutation of the piece's TILE, and the pieces of one row may sit
// in different tiles.
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_pieces(int64_t nnz, int64_t n_rows,
                                                              const int64_t* __restrict__ ptr,
                                                              const int32_t* __restrict__ vptr,
                                                              const int64_t* __restrict__ pptr,
                                                              const int32_t* __restrict__ idx,
                                                              const int2* __restrict__ val,
                                                              int32_t* __restrict__ idx_p, int2* __restrict__ val_p) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    int64_t lo = 0, hi = n_rows;  // the last row with ptr[row] <= e
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int64_t k = e - ptr[lo];
    const int32_t v0 = vptr[lo], P = vptr[lo + 1] - v0;
    const int64_t at = pptr[v0 + (int32_t)(k % P)] + k / P;
    idx_p[at] = idx[e];
    val_p[at] = val[e];
}

// FORM 1 words of every entry (what the host-side balancing needs besides the index)
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_words(int64_t nnz, const int2* __restrict__ val,
                                                             uint8_t* __restrict__ words) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    const int2 x = val[e];
    words[e] = (uint8_t)min(vrx_chunks(x.x) + vrx_chunks((int64_t)x.y - x.x), 255);
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_relabel_gather(int64_t nnz, const uint64_t* __restrict__ keys,
                                                                      const uint32_t* __restrict__ perm,
                                                                      const int2* __restrict__ val, int pbits,
                                                                      int32_t* __restrict__ idx2,
                                                                      int2* __restrict__ val2) {
    const int64_t q = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (q >= nnz) return;
    idx2[q] = (int32_t)(keys[q] & (((uint64_t)1 << pbits) - 1));
    val2[q] = val[perm[q]];
}

// ---- balanced slabs: the per-tile greedy on the device (rule: vrx_balance.h; specification: the host's
// vrx_balance_tile, against which VIREO_BALANCE_CHECK=1 compares it bit for bit) ----------------------
// (1) one key per entry: tile << cbits | contracted index; value = position of its unit (row or piece) inside
//     the tile | words << 16.  Entries without words, or of a unit in no tile, sort behind everything
//     (tile index n_tile).
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_keys(int64_t nnz, int64_t n_units,
                                                          const int64_t* __restrict__ ptr,
                                                          const int32_t* __restrict__ idx,
                                                          const int2* __restrict__ val,
                                                          const int32_t* __restrict__ tpos_of_unit, int tile_pos,
                                                          int n_tile, int cbits, uint64_t* __restrict__ keys,
                                                          uint32_t* __restrict__ vals) {
    const int64_t e = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (e >= nnz) return;
    int64_t lo = 0, hi = n_units;  // the last unit with ptr[unit] <= e
    while (lo < hi) {
        const int64_t mid = (lo + hi + 1) >> 1;
        if (ptr[mid] <= e)
            lo = mid;
        else
            hi = mid - 1;
    }
    const int2 x = val[e];
    const int w = min(vrx_chunks(x.x) + vrx_chunks((int64_t)x.y - x.x), 255);
    const int32_t tp = tpos_of_unit[lo];
    if (tp < 0 || w == 0) {
        keys[e] = (uint64_t)n_tile << cbits;  // (the tile behind the last one)
        vals[e] = 0;
        return;
    }
    keys[e] = ((uint64_t)(tp / tile_pos) << cbits) | (uint32_t)idx[e];
    vals[e] = (uint32_t)(tp % tile_pos) | (uint32_t)w << 16;
}

// (2) first sorted entry of every (tile, column); cptr[n_tile * n_contract] = the entries that count
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_cptr(int64_t n_cols_all, int64_t n_contract, int64_t nnz,
                                                          const uint64_t* __restrict__ keys, int cbits,
                                                          uint32_t* __restrict__ cptr) {
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i > n_cols_all) return;
    const uint64_t key = ((uint64_t)(i / n_contract) << cbits) | (uint64_t)(i % n_contract);
    int64_t lo = 0, hi = nnz;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (keys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    cptr[i] = (uint32_t)lo;
}

// (3) processing order: (tile, block) | 4095 - degree | column, sorted ascending = degree descending, then column
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_order_keys(int64_t n_cols_all, int64_t n_contract,
                                                                const uint32_t* __restrict__ cptr, int slab_rows,
                                                                int bs, int nb, int cbits,
                                                                uint64_t* __restrict__ okeys, int32_t* too_deep) {
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= n_cols_all) return;
    const int64_t t = i / n_contract, c = i % n_contract;
    const uint32_t deg = cptr[i + 1] - cptr[i];
    if (deg > 4095u) *too_deep = 1;
    const uint64_t g = (uint64_t)t * (uint64_t)nb + (uint64_t)((c / slab_rows) / bs);
    okeys[i] = (g << (12 + cbits)) | ((uint64_t)(4095u - min(deg, 4095u)) << cbits) | (uint64_t)c;
}

// (4) first column (in processing order) of every (tile, block); seg[n_groups] = all columns
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_groups(int64_t n_groups, int64_t n_cols_all,
                                                            const uint64_t* __restrict__ okeys, int shift,
                                                            int64_t* __restrict__ seg) {
    const int64_t g = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (g > n_groups) return;
    const uint64_t key = (uint64_t)g << shift;
    int64_t lo = 0, hi = n_cols_all;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (okeys[mid] < key)
            lo = mid + 1;
        else
            hi = mid;
    }
    seg[g] = lo;
}

// (5) degrees in processing order (scanned into stream offsets by the host's hipCUB call), (6) the entries
// copied into that order: the greedy reads ONE sequential stream, no dependent address in its loop
__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_degrees(int64_t n_cols_all, const uint64_t* __restrict__ okeys,
                                                             int cbits, uint32_t* __restrict__ deg) {
    const int64_t k = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (k < n_cols_all) deg[k] = 4095u - (uint32_t)((okeys[k] >> cbits) & 4095u);
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_bal_stream(int64_t n_cols_all, int64_t n_contract,
                                                            const uint64_t* __restrict__ okeys, int cbits, int nb,
                                                            const uint32_t* __restrict__ cptr,
                                                            const uint32_t* __restrict__ vals,
                                                            const uint32_t* __restrict__ ostart,
                                                            uint32_t* __restrict__ ovals) {
    const int64_t k = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (k >= n_cols_all) return;
    const uint64_t ok = okeys[k];
    const int64_t t = (int64_t)(ok >> (12 + cbits)) / nb, c = (int64_t)(ok & (((uint64_t)1 << cbits) - 1));
    const uint32_t a = cptr[t * n_contract + c], b = cptr[t * n_contract + c + 1];
    uint32_t o = ostart[k];
    for (uint32_t e = a; e < b; ++e) ovals[o++] = vals[e];
}

// minimum over the wave: DPP inside the rows of 16 lanes (pairs, quads, half-row mirror, row mirror: a
// minimum does not care which way the lanes are folded), the four row results by readlane
__device__ __forceinline__ int vrx_wave_min(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xf, 0xf, false));  // row_half_mirror
    v = min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xf, 0xf, false));  // row_mirror
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// (7) the greedy: one wave per (tile, block), lane = slab of the block, the block's load matrix
// [tile position][64 slabs] of bytes in LDS.  The group's entries arrive through a ring of two 1024-entry
// batches in LDS: the next batch waits in 16 registers per lane (loaded a batch -- some thirty columns --
// ahead, so its latency is never waited for) and is written behind the ring when the columns reach it.  A
// column of <= 64 entries is one LDS read (lane i = entry i); the score of a slab is the sum of the loads of
// the column's rows in it (every lane reads its own byte of each row, 16 reads in flight), the winner a
// wave-wide minimum of score << 6 | lane among the lanes with room; the column's words are added
// lane-parallel (saturating).
constexpr int VRX_BAL_BATCH = 1024;

// One wave per workgroup: the LDS executes a wave's instructions in order, so a write is seen by every later
// read of any lane without a barrier; what is needed is only that the COMPILER keeps the order.  (A
// __syncthreads() here also waits for the column's global stores and the batch on its way: ~1 us per column.)
__device__ __forceinline__ void vrx_bal_lds_order() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void vrx_balance_greedy(const uint64_t* __restrict__ okeys,
                                                         const int64_t* __restrict__ seg,
                                                         const uint32_t* __restrict__ ostart,
                                                         const uint32_t* __restrict__ ovals, int64_t n_stream,
                                                         int64_t n_contract, int cbits, int n_slab, int slab_rows,
                                                         int bs, int nb, int tile_pos, int32_t* __restrict__ posmap,
                                                         int32_t* __restrict__ perm, int32_t* __restrict__ fail) {
    extern __shared__ __attribute__((aligned(16))) uint8_t vrx_bal_lds[];
    uint32_t* ring = reinterpret_cast<uint32_t*>(vrx_bal_lds);        // [2 * VRX_BAL_BATCH] entries
    uint8_t* load = vrx_bal_lds + 2 * VRX_BAL_BATCH * sizeof(uint32_t);  // [tile_pos + 1][64], the last row zero
    const int g = blockIdx.x, lane = threadIdx.x;
    const int64_t t = g / nb;
    const int w0 = (g % nb) * bs, wn = min(bs, n_slab - w0);
    for (int i = lane; i < (tile_pos + 1) * 16; i += 64) reinterpret_cast<uint32_t*>(load)[i] = 0u;
    __syncthreads();
    const int64_t k0 = seg[g], k1 = seg[g + 1];
    if (k0 >= k1) return;
    const int64_t slots = (int64_t)n_slab * slab_rows;
    const uint64_t cmask = ((uint64_t)1 << cbits) - 1;
    int cap = lane < wn ? slab_rows : 0, fill = 0;
    const int64_t q0 = ostart[k0];
    const int64_t q_end = k1 < seg[(int64_t)gridDim.x] ? (int64_t)ostart[k1] : n_stream;
    int64_t q = q0;            // the next column's first entry
    int64_t resident = q0;     // entries [.., resident) are in the ring (at (entry - q0) mod 2 batches)
    uint32_t nxt[16];          // batch [resident, resident + VRX_BAL_BATCH), on its way
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t e = resident + i * 64 + lane;
        nxt[i] = e < q_end ? ovals[e] : 0u;
    }
    for (int64_t kb = k0; kb < k1; kb += 64) {
        const uint64_t okl = kb + lane < k1 ? okeys[kb + lane] : 0;
        uint32_t okl_lo = (uint32_t)okl, okl_hi = (uint32_t)(okl >> 32);
        // (the wait for this load belongs HERE: left to the compiler it becomes an `s_waitcnt vmcnt(0)` at the
        //  head of the column loop, which then waits for the previous column's two stores every time: ~1 us)
        asm volatile("" : "+v"(okl_lo), "+v"(okl_hi));
        const int nk = (int)min((int64_t)64, k1 - kb);
        for (int j = 0; j < nk; ++j) {
            const uint32_t ok_lo = __builtin_amdgcn_readlane(okl_lo, j), ok_hi = __builtin_amdgcn_readlane(okl_hi, j);
            const uint64_t ok = (uint64_t)ok_hi << 32 | ok_lo;
            const int64_t c = (int64_t)(ok & cmask);
            const int deg = 4095 - (int)((ok >> cbits) & 4095u);
            int best = -1;
            uint32_t mine = 0;
            if (deg > 0 && deg <= 64) {
                while (q + 64 > resident && resident < q_end) {  // the waiting batch goes behind the ring
                    const uint32_t at = (uint32_t)((resident - q0) & (2 * VRX_BAL_BATCH - 1));
#pragma unroll
                    for (int i = 0; i < 16; ++i) ring[at + i * 64 + lane] = nxt[i];
                    resident += VRX_BAL_BATCH;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int64_t e = resident + i * 64 + lane;
                        nxt[i] = e < q_end ? ovals[e] : 0u;
                    }
                    vrx_bal_lds_order();
                }
                mine = lane < deg ? ring[(uint32_t)((q - q0 + lane) & (2 * VRX_BAL_BATCH - 1))] : (uint32_t)tile_pos;
                // Scores, four entries a step: the 16 lanes of row r of the wave take entry 4 * step + r, each
                // lane one dword = four slabs of that entry's load row, summed in 16-bit fields (a single wave
                // issues one instruction every few clocks: the instruction count per column IS the run time;
                // one byte read per entry and slab cost four times as many).
                const uint32_t maddr = (mine & 0xffffu) << 6;  // byte address of the entry's load row
                const uint32_t from = (uint32_t)(lane >> 4) << 2, quad = (uint32_t)(lane & 15) << 2;
                uint32_t even = 0, odd = 0;  // slabs (4q, 4q + 2) and (4q + 1, 4q + 3) of quad q = lane & 15
                for (int e0 = 0; e0 < deg; e0 += 16) {
                    uint32_t x[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t row = __builtin_amdgcn_ds_bpermute((int)(from + (uint32_t)(e0 + 4 * i) * 4u), (int)maddr);
                        x[i] = *reinterpret_cast<const uint32_t*>(load + row + quad);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        even += x[i] & 0x00ff00ffu;
                        odd += (x[i] >> 8) & 0x00ff00ffu;
                    }
                }
                // the four rows of the wave added up (every lane then holds its quad's totals), and lane = slab
                // picks its field out of the lane that holds its quad
                even += (uint32_t)__shfl_xor((int)even, 16, 64);
                odd += (uint32_t)__shfl_xor((int)odd, 16, 64);
                even += (uint32_t)__shfl_xor((int)even, 32, 64);
                odd += (uint32_t)__shfl_xor((int)odd, 32, 64);
                const uint32_t ev = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 2) << 2, (int)even);
                const uint32_t od = (uint32_t)__builtin_amdgcn_ds_bpermute((lane >> 2) << 2, (int)odd);
                const uint32_t both = (lane & 1) ? od : ev;
                const int score = (int)((lane & 2) ? both >> 16 : both & 0xffffu);
                const int key = vrx_wave_min(cap > 0 ? (score << 6 | lane) : 0x7fffffff);
                best = key == 0x7fffffff ? -1 : (key & 63);
            } else if (deg > 64) {  // a long column: straight from memory, 64 entries at a time
                int score = 0;
                for (int64_t e0 = q; e0 < q + deg; e0 += 64) {
                    const int n = (int)min((int64_t)64, q + deg - e0);
                    const uint32_t v = lane < n ? ovals[e0 + lane] : 0u;
                    for (int i = 0; i < n; ++i) score += load[(__builtin_amdgcn_readlane(v, i) & 0xffffu) * 64u + lane];
                }
                const int key = vrx_wave_min(cap > 0 ? (score << 6 | lane) : 0x7fffffff);
                best = key == 0x7fffffff ? -1 : (key & 63);
            }
            if (best < 0) {  // no entry in this tile (they come last): the first slab of the block with room
                const unsigned long long room = __ballot(cap > 0);
                if (room == 0ull) {
                    if (lane == 0) *fail = 1;
                    return;
                }
                best = __ffsll((long long)room) - 1;
            }
            const int f = __builtin_amdgcn_readlane(fill, best);
            if (lane == best) {
                --cap;
                ++fill;
            }
            if (lane == 0) {
                const int64_t pos = (int64_t)(w0 + best) * slab_rows + f;
                posmap[t * n_contract + c] = (int32_t)pos;
                perm[t * slots + pos] = (int32_t)c;
            }
            if (deg > 0 && deg <= 64) {
                if (lane < deg) {
                    uint8_t& L = load[(mine & 0xffffu) * 64u + (uint32_t)best];
                    L = (uint8_t)min((int)L + (int)(mine >> 16), VRX_BAL_LOAD_MAX);
                }
            } else if (deg > 64) {
                for (int64_t e = q + lane; e < q + deg; e += 64) {
                    const uint32_t v = ovals[e];
                    uint8_t& L = load[(v & 0xffffu) * 64u + (uint32_t)best];
                    L = (uint8_t)min((int)L + (int)(v >> 16), VRX_BAL_LOAD_MAX);
                }
            }
            q += deg;
            vrx_bal_lds_order();  // this column's LDS updates before the next column's reads
        }
    }
}

// ---- tiled streams: segment of every (wave, slab, tile position), round lengths ------------------
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_count(VrxTileArgs A, uint32_t* __restrict__ seg_lo,
                                                             uint32_t* __restrict__ seg_hi,
                                                             int32_t* __restrict__ rlen) {
    const int64_t t = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    const int64_t total = A.n_wave * A.n_slab * A.RW;
    int n = 0, n2 = 0;  // entries of phase 0 (all forms) and phase 1 (form 2)
    if (t < total) {
        const int pos = (int)(t % A.RW);
        const int64_t ws = t / A.RW;
        const int sl = (int)(ws % A.n_slab);
        const int64_t w = ws / A.n_slab;
        const int32_t v = A.rowmap[w * A.RW + pos];
        uint32_t lo = 0, hi = 0;  // (entry offsets: up to 2^32 - 1 entries per orientation)
        if (v >= 0) {
            const int32_t row = A.vrow_row[v];
            const int64_t r0 = A.ptr[row], r1 = A.ptr[row + 1];
            const int64_t base = (int64_t)sl * A.slab_rows;
            const int64_t seg = vrx_lower_bound(A.idx, r0, r1, base);
            const int64_t end = vrx_lower_bound(A.idx, seg, r1, base + A.slab_rows);
            const int step = A.vptr[row + 1] - A.vptr[row];
            const int off = (int)(((int64_t)(v - A.vptr[row]) + sl) % step);
            lo = (uint32_t)(seg + off);
            hi = (uint32_t)end;
            for (int64_t e = seg + off; e < end; e += step) {
                if (A.form == 0) {
                    ++n;
                } else {
                    const int2 x = A.val[e];
                    const int ca = vrx_chunks(x.x), cb = vrx_chunks((int64_t)x.y - x.x);
                    if (A.form == 1) {
                        n += ca + cb;
#ifdef VRX_CAP_PROBE
                        if (A.cap > 0 && n > A.cap) n = A.cap;
#endif
                    } else {
                        n += ca;
                        n2 += cb;
                    }
                }
            }
        }
        seg_lo[t] = lo;
        seg_hi[t] = hi;
    }
    // longest segment of the round: the G positions of a round are G consecutive threads
    int L = n, L2 = n2;
    for (int m = 1; m < A.G; m <<= 1) {
        L = max(L, __shfl_xor(L, m, 64));
        L2 = max(L2, __shfl_xor(L2, m, 64));
    }
    if (t < total && (t % A.G) == 0) {
        rlen[t / A.G * A.PH] = L;
        if (A.PH == 2) rlen[t / A.G * 2 + 1] = L2;
    }
}

// per wave: stream offset of every (slab, round) | entries in its last trip; total length
__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_offsets(VrxTileArgs A, const int32_t* __restrict__ rlen,
                                                               int32_t* __restrict__ bnd,
                                                               int64_t* __restrict__ wave_len, int32_t* too_long) {
    const int64_t w = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (w >= A.n_wave) return;
    const int64_t nsr = (int64_t)A.n_slab * A.NR * A.PH;
    int64_t rel = 0;
    for (int64_t i = 0; i < nsr; ++i) {
        if (rel >= INT32_MAX - 4096) {
            *too_long = 1;
            return;
        }
        const int L = rlen[w * nsr + i];
        bnd[w * (nsr + 1) + i] = (int32_t)(rel | (L % A.U));
        rel += (int64_t)((L + A.U - 1) / A.U * A.U) * A.G;
    }
    bnd[w * (nsr + 1) + nsr] = (int32_t)rel;
    wave_len[w] = rel;
}

// the words of one segment, in the order of build_tiled: f(word) for every entry / chunk
template <class F>
__device__ __forceinline__ void vrx_segment_words(const VrxTileArgs& A, int64_t lo, int64_t hi, int step,
                                                  int64_t base, int ph, F&& f0) {
#ifdef VRX_CAP_PROBE
    int left = A.cap > 0 ? A.cap : INT32_MAX;
    auto f = [&](uint32_t wd) {
        if (left > 0) {
            --left;
            f0(wd);
        }
    };
#else
    F& f = f0;
#endif
    for (int64_t e = lo; e < hi; e += step) {
        const int2 x = A.val[e];
        const uint32_t loc = (uint32_t)(A.idx[e] - base);
        if (A.form == 0) {
            f((loc << 22) | ((uint32_t)x.x << 11) | (uint32_t)x.y);
        } else {
            int64_t parts[2] = {x.x, (int64_t)x.y - x.x};
            for (int h = 0; h < 2; ++h) {
                if (A.form == 2 && h != ph) continue;  // form 2: one of the two per phase
                int64_t v = parts[h];
                const uint32_t off = A.form == 2 ? A.f1_base + loc * 128u
                                                 : A.f1_base + loc * 256u + (uint32_t)h * 128u;
                while (v != 0) {
                    const uint64_t mag = (uint64_t)(v < 0 ? -v : v);
                    const int len = 64 - __clzll((long long)mag), sh = len > 3 ? len - 3 : 0;
                    const int64_t c = (int64_t)((mag >> sh) << sh) * (v < 0 ? -1 : 1);
                    const uint64_t bits = (uint64_t)__double_as_longlong((double)c);
                    f((uint32_t)(bits >> 50) << 18 | off);
                    v -= c;
                }
            }
        }
    }
}

__global__ __launch_bounds__(VRX_BLOCK) void vrx_build_fill(VrxTileArgs A, const uint32_t* __restrict__ seg_lo,
                                                            const uint32_t* __restrict__ seg_hi,
                                                            const int32_t* __restrict__ rlen,
                                                            const int32_t* __restrict__ bnd,
                                                            const int64_t* __restrict__ wave_start,
                                                            uint32_t* __restrict__ ent) {
    const int half = A.G / 2;
    const int64_t t = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    const int64_t nsr = (int64_t)A.n_slab * A.NR * A.PH;
    if (t >= A.n_wave * nsr * half) return;
    const int p = (int)(t % half);
    const int64_t wr = t / half;
    const int64_t i = wr % nsr;  // (slab, round, phase)
    const int64_t w = wr / nsr;
    const int ph = (int)(i % A.PH);
    const int sl = (int)(i / A.PH / A.NR), r = (int)(i / A.PH % A.NR);
    const int L = rlen[w * nsr + i];
    if (L == 0) return;
    const int Lr = (L + A.U - 1) / A.U * A.U;
    // the p-th lane group whose partner (g ^ xor_partner) has the larger index
    int hb = 1;
    while (hb * 2 <= A.xor_partner) hb *= 2;
    const int g0 = ((p / hb) * hb * 2) | (p % hb), g1 = g0 ^ A.xor_partner;
    uint32_t* dst = ent + wave_start[w] + (bnd[w * (nsr + 1) + i] & ~(A.U * A.G - 1));
    const int64_t base = (int64_t)sl * A.slab_rows;
    int64_t lo[2], hi[2];
    int step[2];
    const int gs[2] = {g0, g1};
    for (int m = 0; m < 2; ++m) {
        const int64_t at = (w * A.n_slab + sl) * A.RW + r * A.G + gs[m];
        lo[m] = seg_lo[at];
        hi[m] = seg_hi[at];
        const int32_t v = A.rowmap[w * A.RW + r * A.G + gs[m]];
        step[m] = 1;
        if (v >= 0) {
            const int32_t row = A.vrow_row[v];
            step[m] = A.vptr[row + 1] - A.vptr[row];
        }
    }
    const bool pair = A.pairing && A.bit_shift >= 0;
    if (!pair) {  // natural order, zero-valued words behind the segment
        for (int m = 0; m < 2; ++m) {
            int n = 0;
            vrx_segment_words(A, lo[m], hi[m], step[m], base, ph, [&](uint32_t wd) { dst[vrx_trip_slot(n++, gs[m], A.G, A.U, A.form)] = wd; });
            for (; n < Lr; ++n) dst[vrx_trip_slot(n, gs[m], A.G, A.U, A.form)] = A.pad_word;
        }
        return;
    }
    // counts per bank bit: [group][bit]
    int cnt[2][2] = {{0, 0}, {0, 0}};
    for (int m = 0; m < 2; ++m)
        vrx_segment_words(A, lo[m], hi[m], step[m], base, ph, [&](uint32_t wd) { ++cnt[m][(wd >> A.bit_shift) & 1u]; });
    const int p0 = cnt[0][0], p1 = cnt[0][1], q0 = cnt[1][0], q1 = cnt[1][1];
    const int zlo = max(p0, q1), zhi = min(L - p1, L - q0);
    const bool fits = zlo <= zhi;
    const int z = (zlo + zhi) / 2;
    const uint32_t pad0 = A.form != 0 ? A.f1_base : 0u, pad1 = pad0 | 1u << A.bit_shift;
    // group g0: bit-0 words from position 0, bit-1 words from z (or right behind when it does not fit)
    {
        int c0 = 0, c1 = fits ? z : p0;
        vrx_segment_words(A, lo[0], hi[0], step[0], base, ph, [&](uint32_t wd) {
            const int at = ((wd >> A.bit_shift) & 1u) ? c1++ : c0++;
            dst[vrx_trip_slot(at, g0, A.G, A.U, A.form)] = wd;
        });
        if (fits) {
            for (int n = p0; n < z; ++n) dst[vrx_trip_slot(n, g0, A.G, A.U, A.form)] = pad0;
            for (int n = z + p1; n < L; ++n) dst[vrx_trip_slot(n, g0, A.G, A.U, A.form)] = pad1;
            for (int n = L; n < Lr; ++n) dst[vrx_trip_slot(n, g0, A.G, A.U, A.form)] = A.pad_word;
        } else {
            for (int n = p0 + p1; n < Lr; ++n) dst[vrx_trip_slot(n, g0, A.G, A.U, A.form)] = A.pad_word;
        }
    }
    // its partner: bit-1 words first
    {
        int c1 = 0, c0 = fits ? z : q1;
        vrx_segment_words(A, lo[1], hi[1], step[1], base, ph, [&](uint32_t wd) {
            const int at = ((wd >> A.bit_shift) & 1u) ? c1++ : c0++;
            dst[vrx_trip_slot(at, g1, A.G, A.U, A.form)] = wd;
        });
        if (fits) {
            for (int n = q1; n < z; ++n) dst[vrx_trip_slot(n, g1, A.G, A.U, A.form)] = pad1;
            for (int n = z + q0; n < L; ++n) dst[vrx_trip_slot(n, g1, A.G, A.U, A.form)] = pad0;
            for (int n = L; n < Lr; ++n) dst[vrx_trip_slot(n, g1, A.G, A.U, A.form)] = A.pad_word;
        } else {
            for (int n = q0 + q1; n < Lr; ++n) dst[vrx_trip_slot(n, g1, A.G, A.U, A.form)] = A.pad_word;
        }
    }
}
