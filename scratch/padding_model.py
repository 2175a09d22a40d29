"""Executed stream slots of the LDS-resident CELL pass on the c3 matrix, counted exactly from the
data for the current lock-step layout and for the alternatives VERDICT r3 (item 6) names --
no Poisson approximation: words per (cell, slab of variants) come from the matrix itself.

  python scratch/padding_model.py            (CPU only; ~6 min, ~10 GB)

Layout model (vrx_engine.hip build_tiled, cell orientation, AD/BD words): a tile is 16 waves x RW
rows; inside a tile rows are sorted by total length and dealt to the waves; a wave walks its RW
rows in rounds of G rows (one per lane group); inside a slab a round executes max over its G rows
of the row's words in that slab (the last trip of a round executes only its live words).
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from vireo_amd import synth


def words_of(v):
    """AD/BD words a count needs: chunks of three significant bits (45 = 40 + 5)"""
    v = v.astype(np.int64).copy()
    n = np.zeros(v.shape, dtype=np.int64)
    while True:
        live = v > 0
        if not live.any():
            return n
        ln = np.zeros(v.shape, dtype=np.int64)
        ln[live] = np.floor(np.log2(v[live])).astype(np.int64) + 1
        sh = np.maximum(ln - 3, 0)
        v[live] -= (v[live] >> sh[live]) << sh[live]
        n[live] += 1


def main():
    t0 = time.time()
    N, M, K, d = synth.CONFIGS["c3"]
    w = synth.donor_workload(N, M, K, d, seed=0)
    nnz = w["rowidx"].size
    cell = np.repeat(np.arange(M, dtype=np.int64), np.diff(w["colptr"]))
    wd = words_of(w["ad"]) + words_of(w["dp"] - w["ad"])
    n_words = int(wd.sum())
    print("c3: nnz %d, AD/BD words %d (%.3f per non-zero)  [%.0f s]" % (nnz, n_words, n_words / nnz, time.time() - t0))

    def counts(slab_rows):
        n_slab = -(-N // slab_rows)
        key = cell * n_slab + w["rowidx"] // slab_rows
        return np.bincount(key, weights=wd, minlength=M * n_slab).reshape(M, n_slab).astype(np.int32)

    def tile_order(C, RW):
        """rows of a tile sorted by total length, dealt to the 16 waves in snake order ->
        [tile][wave][RW] row ids (-1 = padding)"""
        tile_rows = 16 * RW
        n_tile = -(-M // tile_rows)
        tot = C.sum(1)
        out = np.full((n_tile, 16, RW), -1, dtype=np.int64)
        for t in range(n_tile):
            rows = np.arange(t * tile_rows, min(M, (t + 1) * tile_rows))
            rows = rows[np.argsort(-tot[rows], kind="stable")]
            pos = np.arange(rows.size)
            lap, lane = pos // 16, pos % 16
            wave = np.where(lap % 2 == 0, lane, 15 - lane)
            out[t, wave, lap] = rows
        return out

    def slots(C, order, G, chain=1, cap_sigma=None):
        """executed slots: rounds of G rows x `chain` rows chained per group; optional cap"""
        Cz = np.vstack([C, np.zeros((1, C.shape[1]), dtype=C.dtype)])     # row -1 -> zeros
        n_tile, _, RW = order.shape
        NR = RW // G
        total = 0
        overflow = 0
        for t in range(n_tile):
            X = Cz[order[t]]                                   # [16][RW][n_slab]
            X = X.reshape(16, NR, G, -1)                        # rounds of G rows (sorted neighbours)
            if chain > 1:
                X = X.reshape(16, NR // chain, chain, G, -1).sum(2)
            if cap_sigma is not None:
                mean = X.mean()
                cap = int(np.ceil(mean + cap_sigma * np.sqrt(mean)))
                overflow += int(np.maximum(X - cap, 0).sum())
                X = np.minimum(X, cap)
            total += int(X.max(2).sum()) * G                  # a round executes its longest row's words on all G groups
        return total, overflow

    base = None
    rows_out = []
    for slab_rows in (512, 640, 768, 1024):
        C = counts(slab_rows)
        for RW in ((96,) if slab_rows != 512 else (96, 48)):
            order = tile_order(C, RW)
            for G in (16, 8, 32):
                if RW % G:
                    continue
                s, _ = slots(C, order, G)
                if base is None:
                    base = s
                rows_out.append(("slab %4d rows, %2d rows/wave, rounds of %2d rows" % (slab_rows, RW, G), s))
            if slab_rows == 512 and RW == 96:
                for chain in (2, 3, 6):
                    s, _ = slots(C, order, 16, chain=chain)
                    rows_out.append(("slab  512, 96 rows/wave, rounds of 16, %d rows chained per group" % chain, s))
                for sig in (0.5, 1.0, 1.5):
                    s, ov = slots(C, order, 16, cap_sigma=sig)
                    rows_out.append(("slab  512, 96 rows/wave, rounds of 16, cap mean+%.1f sigma (overflow %d words = %.2f %%)"
                                     % (sig, ov, 100.0 * ov / n_words), s))
        print("  slab %d done [%.0f s]" % (slab_rows, time.time() - t0), flush=True)
    print("%-100s %12s %8s %8s" % ("layout", "slots", "/word", "/nnz"))
    for name, s in rows_out:
        print("%-100s %12d %8.3f %8.3f   (%+.1f %% vs current)" % (name, s, s / n_words, s / nnz, 100.0 * (s - base) / base))


if __name__ == "__main__":
    main()
