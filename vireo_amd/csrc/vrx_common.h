// Internal declarations shared by the translation units of libvireo_hip.so.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts are assumed everywhere.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vireo_hip.h"

void vrx_set_error(const char* fmt, ...);

#define VRX_HIP(expr)                                                                   \
    do {                                                                                \
        hipError_t e__ = (expr);                                                        \
        if (e__ != hipSuccess) {                                                        \
            vrx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),       \
                          __FILE__, __LINE__);                                          \
            return VRX_ERR_HIP;                                                         \
        }                                                                               \
    } while (0)

#define VRX_REQUIRE(cond, ...)                                                          \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            vrx_set_error(__VA_ARGS__);                                                 \
            return VRX_ERR_ARG;                                                         \
        }                                                                               \
    } while (0)

// Owning device buffer (freed with the handle).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    hipError_t alloc(size_t count) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    void swap(DevBuf& o) {
        std::swap(p, o.p);
        std::swap(n, o.n);
    }
    hipError_t upload(const T* src, size_t count, hipStream_t s) {
        hipError_t e = alloc(count);
        if (e != hipSuccess || count == 0) return e;
        return hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, s);
    }
};

// One orientation of the (ad, dp) matrix plus its segment table.
//   rows       : the OUTPUT dimension of a sparse pass over this orientation
//   ent        : the entries, (index along the CONTRACTED dimension, ad, dp) packed in
//                fmt + 1 32-bit words each (VRX_FMT_* in vrx_kernels.h)
// A segment is a contiguous run of one row's entries inside ONE tile of the contracted
// dimension, at most seg_cap long; one wavefront reduces one segment.  Tiles bound the
// slab of the dense operand a pass gathers from (so that it stays in an XCD's 4 MiB L2);
// the segment arrays are stored in LAUNCH ORDER: workgroup b runs on XCD b % 8 (observed
// dispatch rule, used for speed only), so the segments of tile t are laid out to land on
// XCD t % 8.  seg_len < 0 marks padding.  Rows with exactly one segment write their result
// in place (seg_dst >= 0: the row); rows with several write partial sums into slots
// (seg_dst = -(slot+1)) that `multi_*` lists for the in-order second stage.
// Tiled copy of one orientation for the LDS-resident pass (vrx_spmm_lds): entries ordered
// (row tile, wave, slab, row, index) with slab-local indices; bnd holds per wave the stream
// offset of every (slab, row) segment start (+ one end marker).
struct TiledStream {
    bool ready = false;
    int form = 0;  // 0: (ad, dp) pairs; 1: single-valued AD / BD entries (cell pass, FORM 1)
    int rw = 0, slab_rows = 0, n_slab = 0, n_tile = 0;
    // The pass runs as n_wg persistent workgroups (one per CU); workgroup b walks the work items
    // [wg_first[b], wg_first[b + 1]): item = (tile, first slab, end slab, slot) -- a contiguous
    // run of one tile's slabs whose partial output goes to partial array `slot` (0, 1, ... in
    // slab order inside a tile).  The (tile, slab) visits are cut into n_wg runs of equal cost.
    // virt: the stream's rows are VIRTUAL rows of the variant pass (vrx_build.h): row 2n = the AD
    // counts of variant n, row 2n + 1 its BD counts, against the operand read as double rows
    // (n_contract of them); the pass runs the cell pass's kernel form and leaves planar sums
    bool virt = false;
    int64_t n_contract = 0;   // contracted rows of THIS stream (double rows when virt)
    int n_range = 1;  // partial arrays = the most pieces any tile is cut into
    int n_wg = 0;
    DevBuf<int32_t> items;     // 4 words per item
    DevBuf<int32_t> wg_first;  // n_wg + 1
    DevBuf<uint16_t> npiece;    // per piece row (n_vrows): partial arrays that hold a term of it
    DevBuf<uint32_t> ent;
    DevBuf<int64_t> wave_start;
    DevBuf<int32_t> bnd;
    DevBuf<int32_t> rowmap;   // tile position -> piece (-1 = padding), pieces sorted by length
    // Balanced slabs (r6): which contracted rows share a slab is chosen PER TILE so that the rows of the
    // tile carry about the same number of words in every slab (the lock-step padding of a round is the
    // maximum over its 16 rows).  perm[tile][slab * slab_rows + p] = contracted row staged at slab-local
    // position p (unused positions name row 0); the stream's indices are positions.  Empty: slabs are consecutive rows.
    DevBuf<int32_t> perm;
    bool balanced = false, want_balance = false;
    double balance_seconds = 0.0;
    DevBuf<int32_t> vptr;     // row -> its pieces [vptr[r], vptr[r+1]) (only when split)
    DevBuf<int32_t> split_rows;  // the rows with more than one piece (vrx_fold_split)
    int64_t n_split = 0;
    int64_t n_vrows = 0;      // pieces (== rows unless split)
    bool split = false;       // some row is cut into several pieces
    double pad_ratio = 0.0;   // stream words per entry
    double imbalance = 1.0;   // longest wave stream / mean wave stream
};

struct Orient {
    TiledStream tiled;
    int64_t n_rows = 0, n_contract = 0, nnz = 0;
    int fmt = 2;
    int n_tiles = 1;
    DevBuf<uint32_t> ent;
    int64_t n_seg = 0;  // including padding
    DevBuf<int64_t> seg_begin;
    DevBuf<int32_t> seg_len;
    DevBuf<int32_t> seg_dst;
    int64_t n_multi = 0, n_slots = 0;
    int64_t n_empty = 0;  // rows without entries (no segment: their output is the zero fill)
    DevBuf<int32_t> multi_row;
    DevBuf<int32_t> multi_ptr;  // n_multi + 1 slot offsets
};

struct vrx_problem {
    int device = 0;
    int n_cu = 0;
    int64_t n_var = 0, n_cell = 0, nnz = 0;
    hipStream_t stream = nullptr;
    Orient by_var;   // rows = variants, contracted = cells   (CSR of the N x M matrix)
    Orient by_cell;  // rows = cells,    contracted = variants (CSC of the N x M matrix)
    bool binom_done = false;
    double binom_sum = 0.0;
    std::vector<int32_t> n_vars;
    bool want_balance = false;     // vrx_problem_create2 flag VRX_PROBLEM_BALANCED (TiledStream::perm)
    double balance_seconds = 0.0;  // what the balanced slabs added to the build (both orientations)
};

// vrx_comm.hip broadcasts a model's variational state in place (vrx_comm_bcast_model): the four
// state arrays of a model -- ID_prob, GT_prob (n = 0 in clone mode), beta_mu, beta_sum -- after the
// model's stream has drained; will_write: the caller overwrites them (derived tables are then stale).
// Internal to the library (C++ linkage: not part of the C ABI).
struct VrxModelBuffers {
    double* p[4];
    size_t n[4];
    int device;
};
int vrx_model_state_buffers(vrx_model* m, bool will_write, VrxModelBuffers* out);
