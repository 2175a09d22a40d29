// Balanced slabs: the rule both implementations of the per-tile greedy follow -- vrx_balance_tile (host,
// vrx_host.cpp: the specification) and vrx_balance_greedy (device, vrx_build.h: what a device build runs).
// With VIREO_BALANCE_CHECK=1 the device build runs both and compares them bit for bit.
//
//  * the slabs of a tile are cut into `nb` blocks of `bs` consecutive slabs (the last one may be shorter);
//    a contracted row ("column" of the tile's sub-matrix) may only move inside its own block.  Blocks of at
//    most 64 slabs: measured at c3 (196 slabs in the cell orientation) the whole range gives 1.185 executed
//    slots per word, blocks of 64 give 1.225 -- and the same pass time, because the workgroups of a launch
//    then stage their slabs out of one stretch of the operand at the same time (blocks of 32 / 16 / 8:
//    1.26 / 1.30 / 1.35, passes 1-5 % slower); the search is linear in the problem at any size, and a
//    block's load matrix (tile rows x 64 slabs, one byte each) fits a CU's LDS.
//  * columns in the order (degree in the tile descending, column ascending); each goes to the slab of its
//    block -- among those with room -- where the sum of the present loads of the tile rows it touches is
//    smallest (ties: the lowest slab); a load is the words placed so far, saturating at 127; columns
//    without an entry in the tile take the first slab of their block with room, after all others.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define VRX_BAL_HD __host__ __device__
#else
#define VRX_BAL_HD
#endif

constexpr int VRX_BAL_LOAD_MAX = 127;

struct VrxBalBlocks {
    int nb, bs;  // number of blocks, slabs per block (the last block: n_slab - (nb - 1) * bs)
};

VRX_BAL_HD inline VrxBalBlocks vrx_bal_blocks(int n_slab, int max_block) {
    if (max_block <= 0 || n_slab <= max_block) return VrxBalBlocks{1, n_slab > 0 ? n_slab : 1};
    const int nb = (n_slab + max_block - 1) / max_block;
    return VrxBalBlocks{nb, (n_slab + nb - 1) / nb};
}
