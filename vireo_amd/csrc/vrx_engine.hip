// Host side of libvireo_hip.so: the C ABI of include/vireo_hip.h.
// Owns the device state, schedules the kernels of vrx_kernels.h on one HIP stream per
// problem and runs the coordinate-ascent loop of the reference
// (vireoSNP/utils/vireo_model.py:251-276, vireoSNP/utils/bmm_model.py:178-201).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "vrx_common.h"
#include "vrx_kernels.h"
#include "vrx_build.h"

// (vrx_set_error / vrx_last_error: vrx_host.cpp, so that the host-only translation unit links on
//  its own for the sanitizer build of tests/test_host_sanitizers_cpu.py)

extern "C" int vrx_device_count(int* n) {
    VRX_REQUIRE(n, "vrx_device_count: null output");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        c = 0;
    }
    *n = c;
    return VRX_OK;
}

extern "C" int vrx_device_info(int device, char* name, int name_len, int* n_cu,
                               int64_t* hbm_bytes) {
    hipDeviceProp_t prop;
    VRX_HIP(hipGetDeviceProperties(&prop, device));
    if (name && name_len > 0) {
        // (some boxes report an empty marketing name: say what the arch is then)
        snprintf(name, name_len, "%s (%s)", prop.name[0] ? prop.name : "AMD GPU",
                 prop.gcnArchName);
    }
    if (n_cu) *n_cu = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return VRX_OK;
}

// "0000:c1:00.0" of a device: which physical GPU a rank of the restart shard sits on
// (bench.py's `comm` block lists it per rank)
extern "C" int vrx_device_pci_bus_id(int device, char* out, int out_len) {
    VRX_REQUIRE(out && out_len >= 16, "vrx_device_pci_bus_id: buffer of >= 16 bytes needed");
    VRX_HIP(hipDeviceGetPCIBusId(out, out_len, device));
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// problem
// ------------------------------------------------------------------------------------
static constexpr int kSegCap = 4096;     // entries per segment (one wavefront each)
static constexpr int kXcd = 8;           // XCDs per MI355X; workgroup b is observed on XCD b % 8
static constexpr double kSlabBytes = 1.6e6;  // dense-operand slab per tile (fits a 4 MiB L2)

// Host-side build of a problem (validation, transposition, packing, tiling) is plain loops
// over the non-zeros; they are spread over host threads (VIREO_HOST_THREADS, default <= 64).
static int host_threads() {
    static const int n = [] {
        const char* v = getenv("VIREO_HOST_THREADS");
        int t = v && *v ? atoi(v) : (int)std::min(64u, std::max(1u, std::thread::hardware_concurrency()));
        return std::max(1, t);
    }();
    return n;
}

// the per-tile greedy of the balanced-slab build is compute in a core's own L2: it takes more threads
static int balance_threads() {
    const char* v = getenv("VIREO_HOST_THREADS");
    if (v && *v) return host_threads();
    return (int)std::min(128u, std::max(1u, std::thread::hardware_concurrency()));
}

// uninitialised host array (a std::vector would zero hundreds of MB on one thread first)
template <class T>
struct RawArray {
    T* p = nullptr;
    explicit RawArray(size_t n) : p(static_cast<T*>(std::malloc(std::max<size_t>(n, 1) * sizeof(T)))) {}
    RawArray(const RawArray&) = delete;
    RawArray& operator=(const RawArray&) = delete;
    ~RawArray() { std::free(p); }
    T* data() { return p; }
    T& operator[](size_t i) { return p[i]; }
};

// f(begin, end, tid) over [0, n) cut into contiguous chunks, one per thread
template <class F>
static void parallel_chunks(int64_t n, int n_threads, F&& f) {
    n_threads = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads, n));
    if (n_threads == 1) {
        f((int64_t)0, n, 0);
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)n_threads);
    for (int t = 0; t < n_threads; ++t) {
        const int64_t b = n * t / n_threads, e = n * (t + 1) / n_threads;
        pool.emplace_back([&f, b, e, t] { f(b, e, t); });
    }
    for (auto& th : pool) th.join();
}

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// number of tiles over the contracted dimension for a dense operand of `row_bytes` per
// contracted index (nominal K = 16): 1, 2, 4, 8 or a multiple of 8
static int pick_tiles(int64_t n_contract, double row_bytes, const char* env) {
    int t = env_int(env, 0);
    // (an operand of up to two slabs still sits in one XCD's 4 MiB L2: one tile, and no rows
    //  split over tiles whose slots need a second kernel to sum -- c2: 45.1 -> 42.7 us)
    if (t <= 0) t = n_contract * row_bytes <= 2.0 * kSlabBytes ? 1 : (int)std::lround(n_contract * row_bytes / kSlabBytes);
    if (t <= 1) return 1;
    if (t <= kXcd) {
        int p = 1;
        while (p < t) p <<= 1;
        return p;
    }
    return (t + kXcd - 1) / kXcd * kXcd;
}

// Pack the entries of one orientation, build its tiled / XCD-ordered segment table, upload.
//   contract_count[i] = number of entries with contracted index i (for equal-nnz tiles)
static int build_orient(Orient& o, int64_t n_rows, int64_t n_contract, const int64_t* ptr,
                        const int32_t* idx, const int2* val, const int64_t* contract_ptr,
                        int n_tiles, int fmt, hipStream_t s) {
    o.n_rows = n_rows;
    o.n_contract = n_contract;
    o.nnz = ptr[n_rows];
    o.fmt = fmt;
    o.n_tiles = n_tiles;
    // ---- entries ------------------------------------------------------------------
    const int ew = fmt + 1;
    RawArray<uint32_t> ent((size_t)o.nnz * ew);
    VRX_REQUIRE(ent.p, "out of host memory");
    parallel_chunks(o.nnz, host_threads(), [&](int64_t b, int64_t e_end, int) {
        for (int64_t e = b; e < e_end; ++e) {
            const uint32_t id = (uint32_t)idx[e], ad = (uint32_t)val[e].x, dp = (uint32_t)val[e].y;
            if (fmt == VRX_FMT_P32) {
                ent[(size_t)e] = (id << 12) | (ad << 6) | dp;
            } else if (fmt == VRX_FMT_P64) {
                ent[(size_t)e * 2] = id;
                ent[(size_t)e * 2 + 1] = ad | (dp << 16);
            } else {
                ent[(size_t)e * 3] = id;
                ent[(size_t)e * 3 + 1] = ad;
                ent[(size_t)e * 3 + 2] = dp;
            }
        }
    });
    // ---- tile boundaries: equal entry counts ------------------------------------------
    std::vector<int64_t> bound((size_t)n_tiles + 1, n_contract);
    bound[0] = 0;
    for (int t = 1; t < n_tiles; ++t) {
        const int64_t want = o.nnz * t / n_tiles;
        bound[(size_t)t] = std::lower_bound(contract_ptr, contract_ptr + n_contract + 1, want) -
                           contract_ptr;
        if (bound[(size_t)t] > n_contract) bound[(size_t)t] = n_contract;
        if (bound[(size_t)t] < bound[(size_t)t - 1]) bound[(size_t)t] = bound[(size_t)t - 1];
    }
    // ---- segments, grouped per tile -----------------------------------------------------
    struct Seg {
        int64_t begin;
        int32_t len, dst;
    };
    std::vector<std::vector<Seg>> per_tile((size_t)n_tiles);
    std::vector<int32_t> multi_row, multi_ptr;
    multi_ptr.push_back(0);
    int64_t slots = 0;
    std::vector<Seg> row_segs;
    std::vector<int> row_tile;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int64_t lo = ptr[r], hi = ptr[r + 1];
        if (hi <= lo) {  // empty row: its output stays 0 (buffers are zero-filled)
            ++o.n_empty;
            continue;
        }
        row_segs.clear();
        row_tile.clear();
        int64_t at = lo;
        for (int t = 0; t < n_tiles && at < hi; ++t) {
            const int64_t end = n_tiles == 1 ? hi
                                             : std::lower_bound(idx + at, idx + hi,
                                                                (int32_t)std::min<int64_t>(bound[(size_t)t + 1], INT32_MAX)) - idx;
            int64_t len = end - at;
            if (len <= 0) continue;
            const int64_t parts = (len + kSegCap - 1) / kSegCap;
            int64_t chunk = (len + parts - 1) / parts;
            if (parts > 1) chunk = (chunk + 63) / 64 * 64;
            for (int64_t b = 0; b < len; b += chunk) {
                row_segs.push_back({at + b, (int32_t)std::min(chunk, len - b), 0});
                row_tile.push_back(t);
            }
            at = end;
        }
        if (row_segs.size() == 1) {
            row_segs[0].dst = (int32_t)r;
        } else {
            for (auto& sg : row_segs) sg.dst = (int32_t)(-(++slots));
            multi_row.push_back((int32_t)r);
            multi_ptr.push_back((int32_t)slots);
        }
        for (size_t i = 0; i < row_segs.size(); ++i) per_tile[(size_t)row_tile[i]].push_back(row_segs[i]);
    }
    if (slots >= INT32_MAX) {
        vrx_set_error("too many segments");
        return VRX_ERR_UNSUPPORTED;
    }
    // ---- launch order: tile t -> XCD t % 8 (tiles < 8: each tile shared by 8/n_tiles XCDs)
    std::vector<std::vector<Seg>> per_xcd(kXcd);
    if (n_tiles >= kXcd) {
        for (int t = 0; t < n_tiles; ++t) {
            auto& dst = per_xcd[(size_t)(t % kXcd)];
            dst.insert(dst.end(), per_tile[(size_t)t].begin(), per_tile[(size_t)t].end());
        }
    } else {
        const int share = kXcd / n_tiles;  // XCDs per tile
        for (int t = 0; t < n_tiles; ++t) {
            const auto& src = per_tile[(size_t)t];
            for (size_t i = 0; i < src.size(); ++i) {
                const int x = t + n_tiles * (int)((i / VRX_WAVES) % share);
                per_xcd[(size_t)x].push_back(src[i]);
            }
        }
    }
    size_t longest = 0;
    for (auto& v : per_xcd) longest = std::max(longest, v.size());
    const size_t blocks_per_xcd = (longest + VRX_WAVES - 1) / VRX_WAVES;
    const size_t total = blocks_per_xcd * kXcd * VRX_WAVES;
    std::vector<int64_t> seg_begin(total, 0);
    std::vector<int32_t> seg_len(total, -1), seg_dst(total, 0);
    for (int x = 0; x < kXcd; ++x)
        for (size_t i = 0; i < per_xcd[(size_t)x].size(); ++i) {
            const size_t pos = ((i / VRX_WAVES) * kXcd + (size_t)x) * VRX_WAVES + i % VRX_WAVES;
            seg_begin[pos] = per_xcd[(size_t)x][i].begin;
            seg_len[pos] = per_xcd[(size_t)x][i].len;
            seg_dst[pos] = per_xcd[(size_t)x][i].dst;
        }
    if (total >= (size_t)INT32_MAX) {
        vrx_set_error("too many segments");
        return VRX_ERR_UNSUPPORTED;
    }
    o.n_seg = (int64_t)total;
    o.n_multi = (int64_t)multi_row.size();
    o.n_slots = slots;
    VRX_HIP(o.ent.upload(ent.data(), (size_t)o.nnz * ew, s));
    VRX_HIP(o.seg_begin.upload(seg_begin.data(), seg_begin.size(), s));
    VRX_HIP(o.seg_len.upload(seg_len.data(), seg_len.size(), s));
    VRX_HIP(o.seg_dst.upload(seg_dst.data(), seg_dst.size(), s));
    VRX_HIP(o.multi_row.upload(multi_row.data(), multi_row.size(), s));
    VRX_HIP(o.multi_ptr.upload(multi_ptr.data(), multi_ptr.size(), s));
    VRX_HIP(hipStreamSynchronize(s));  // the host vectors die at return
    return VRX_OK;
}

// Tiled entry stream for vrx_spmm_lds (see vrx_kernels.h): RW rows per wave handled
// 16 at a time (a round), 16 waves per tile, slabs of slab_rows contracted indices; inside a
// round the words are trip-major and zero-padded to the round's longest row.
//
// Ragged data (heavy-tailed coverage / depth): a round costs its LONGEST row and a workgroup
// its slowest wave, so (1) a row much longer than the mean is cut into P interleaved PIECES
// (entry j of a slab segment -> piece (j + slab) % P) that accumulate separately and are
// summed afterwards in piece order (vrx_sum_pieces), (2) pieces are sorted by length so that
// a round holds 16 similar ones, (3) the sorted rounds are dealt to the waves in snake order
// so that every wave (and tile) carries about the same number of entries.
//
// form 1 (cell pass): every (ad, dp) entry becomes single-valued entries of AD and of
// BD = DP - AD (none for a zero, several for a value outside 15 signed bits), see FORM 1 in
// vrx_kernels.h.  Word = value:15 | (2 * slab-local index + half) * 128.
// `dev` != nullptr: the rows of this orientation live on the device (idx / val are then unused
// host pointers) and the stream is built there (vrx_build.h); ptr is always the host copy.
struct DevRows {
    const int64_t* ptr;
    const int32_t* idx;
    const int2* val;
};
// the same rows on the host, where the caller still has them (the cell orientation IS the CSC input)
struct HostCounts {
    const int32_t* idx;
    const int32_t *ad, *dp;
};
// ... or what the balancing needs of them, already brought back from the device by the caller
struct HostWords {
    const int32_t* idx;
    const uint8_t* words;
};

// Work list of an LDS-resident pass (TiledStream::items).  The (tile, slab) visits, tile-major,
// are cut into n_wg contiguous runs of equal cost -- cost of a visit = the longest wave's trips in
// that slab + the staging of the slab, in trips -- so that every CU is busy for the whole launch
// and a tile is cut into few pieces (tiles + n_wg pieces at most: the partial outputs the
// consumers add up).  bnd = the per-wave (slab, round) offsets on the host.
static int plan_items(TiledStream& t, const int32_t* bnd, int64_t n_wave, int nrv, int n_cu,
                      const int32_t* rowmap, int mode, hipStream_t s) {
    constexpr int UG = VRX_LDS_U * (64 / VRX_LDS_LPE);
    const int64_t per_wave = (int64_t)t.n_slab * nrv + 1, visits = (int64_t)t.n_tile * t.n_slab;
    const int want = env_int(mode == 1 ? "VIREO_LDS_BLOCKS_CELL" : "VIREO_LDS_BLOCKS_VAR",
                             env_int("VIREO_LDS_BLOCKS", std::max(1, n_cu)));
    const int n_wg = (int)std::max<int64_t>(1, std::min<int64_t>(want, visits));
    const double stage = (double)env_int("VIREO_LDS_STAGE_TRIPS_X10", 100) / 10.0;
    std::vector<double> cost((size_t)visits);
    parallel_chunks(t.n_tile, host_threads(), [&](int64_t t0, int64_t t1, int) {
        for (int64_t tl = t0; tl < t1; ++tl)
            for (int sl = 0; sl < t.n_slab; ++sl) {
                int32_t longest = 0;
                for (int w = 0; w < VRX_LDS_WAVES; ++w) {
                    const int32_t* bw = bnd + (tl * VRX_LDS_WAVES + w) * per_wave;
                    longest = std::max(longest, (bw[(int64_t)(sl + 1) * nrv] & ~(UG - 1)) -
                                                    (bw[(int64_t)sl * nrv] & ~(UG - 1)));
                }
                cost[(size_t)(tl * t.n_slab + sl)] = (double)longest / UG + stage;
            }
    });
    (void)n_wave;
    double total = 0.0;
    for (double c : cost) total += c;
    std::vector<int32_t> items, first((size_t)n_wg + 1, 0), pieces((size_t)t.n_tile, 0);
    double acc = 0.0;
    int64_t u = 0;
    for (int b = 0; b < n_wg; ++b) {
        first[(size_t)b] = (int32_t)(items.size() / 4);
        // run b ends at the visit where the accumulated cost passes (b + 1) / n_wg of the total
        // (every run gets at least one visit while visits remain for the runs behind it)
        const double goal = total * (double)(b + 1) / (double)n_wg;
        int64_t end = u;
        while (end < visits && (end == u || acc + cost[(size_t)end] * 0.5 <= goal) &&
               visits - (end + 1) >= n_wg - 1 - b)
            acc += cost[(size_t)end++];
        if (b == n_wg - 1)
            while (end < visits) acc += cost[(size_t)end++];
        while (u < end) {  // cut the run at tile boundaries
            const int64_t tl = u / t.n_slab, s0 = u % t.n_slab;
            const int64_t s1 = std::min<int64_t>(t.n_slab, s0 + (end - u));
            items.insert(items.end(), {(int32_t)tl, (int32_t)s0, (int32_t)s1, pieces[(size_t)tl]++});
            u += s1 - s0;
        }
    }
    first[(size_t)n_wg] = (int32_t)(items.size() / 4);
    // Which workgroup walks which run.  The runs start at every phase of the slab cycle, so in
    // launch order the workgroups of one XCD would stage different slabs at any time and share
    // nothing in their L2 (measured: the passes' L2-miss traffic 1.1 -> 1.6 GB per launch).
    // Workgroup b runs on XCD b % 8 (observed dispatch rule; used for speed only): the runs are
    // sorted by the slab they start at and dealt XCD by XCD, so that the ~32 workgroups of an
    // XCD walk neighbouring slabs together and every slab is fetched into that L2 about once.
    if (env_int("VIREO_LDS_XCD_PHASE", 1) && n_wg > kXcd) {
        std::vector<int> run((size_t)n_wg);
        for (int b = 0; b < n_wg; ++b) run[(size_t)b] = b;
        auto phase = [&](int b) {
            return first[(size_t)b] < first[(size_t)b + 1] ? items[(size_t)(4 * first[(size_t)b] + 1)] : INT32_MAX;
        };
        std::stable_sort(run.begin(), run.end(), [&](int a, int b) { return phase(a) < phase(b); });
        std::vector<int> run_of_wg((size_t)n_wg, -1);
        int next = 0;
        for (int x = 0; x < kXcd; ++x)
            for (int b = x; b < n_wg; b += kXcd) run_of_wg[(size_t)b] = run[(size_t)next++];
        std::vector<int32_t> it2, f2((size_t)n_wg + 1, 0);
        for (int b = 0; b < n_wg; ++b) {
            const int r = run_of_wg[(size_t)b];
            f2[(size_t)b] = (int32_t)(it2.size() / 4);
            it2.insert(it2.end(), items.begin() + 4 * first[(size_t)r], items.begin() + 4 * first[(size_t)r + 1]);
        }
        f2[(size_t)n_wg] = (int32_t)(it2.size() / 4);
        items.swap(it2);
        first.swap(f2);
    }
    t.n_wg = n_wg;
    t.n_range = 1;
    for (int32_t c : pieces) t.n_range = std::max(t.n_range, (int)c);
    VRX_REQUIRE(t.n_range < 65536, "tiled stream: a tile is cut into too many pieces");
    std::vector<uint16_t> npiece((size_t)t.n_vrows, 0);
    const int64_t tile_pos = (int64_t)VRX_LDS_WAVES * t.rw;
    for (int64_t pos = 0; pos < (int64_t)t.n_tile * tile_pos; ++pos)
        if (rowmap[(size_t)pos] >= 0) npiece[(size_t)rowmap[(size_t)pos]] = (uint16_t)pieces[(size_t)(pos / tile_pos)];
    VRX_HIP(t.items.upload(items.data(), items.size(), s));
    VRX_HIP(t.wg_first.upload(first.data(), first.size(), s));
    VRX_HIP(t.npiece.upload(npiece.data(), npiece.size(), s));
    VRX_HIP(hipStreamSynchronize(s));
    return VRX_OK;
}

// the virtual rows of the variant pass on the host (vrx_build.h: vrx_virt_count / vrx_virt_fill
// are the same derivation on the device)
static void derive_virtual_rows(int64_t n_var, const int64_t* rptr, const int32_t* ridx, const int2* rval,
                                std::vector<int64_t>& vptr2, std::vector<int32_t>& vidx, std::vector<int2>& vval) {
    vptr2.assign((size_t)(2 * n_var + 1), 0);
    parallel_chunks(n_var, host_threads(), [&](int64_t n0, int64_t n1, int) {
        for (int64_t n = n0; n < n1; ++n) {
            int64_t ca = 0, cb = 0;
            int32_t ja = -1, jb = -1;
            for (int64_t e = rptr[n]; e < rptr[n + 1]; ++e) {
                const int32_t j = ridx[e] >> 1;
                if (rval[e].x != 0 && j != ja) ++ca, ja = j;
                if (rval[e].y - rval[e].x != 0 && j != jb) ++cb, jb = j;
            }
            vptr2[(size_t)(2 * n + 1)] = ca;   // (counts: scanned below)
            vptr2[(size_t)(2 * n + 2)] = cb;
        }
    });
    for (int64_t r = 0; r < 2 * n_var; ++r) vptr2[(size_t)r + 1] += vptr2[(size_t)r];
    vidx.resize((size_t)vptr2[(size_t)(2 * n_var)]);
    vval.resize(vidx.size());
    parallel_chunks(n_var, host_threads(), [&](int64_t n0, int64_t n1, int) {
        for (int64_t n = n0; n < n1; ++n) {
            int64_t oa = vptr2[(size_t)(2 * n)] - 1, ob = vptr2[(size_t)(2 * n + 1)] - 1;
            int32_t ja = -1, jb = -1;
            for (int64_t e = rptr[n]; e < rptr[n + 1]; ++e) {
                const int32_t c = ridx[e], j = c >> 1;
                const int a = rval[e].x, b = rval[e].y - rval[e].x;
                if (a != 0) {
                    if (j != ja) ++oa, ja = j, vidx[(size_t)oa] = j, vval[(size_t)oa] = make_int2(0, 0);
                    if (c & 1) vval[(size_t)oa].y += a; else vval[(size_t)oa] = make_int2(a, vval[(size_t)oa].y + a);
                }
                if (b != 0) {
                    if (j != jb) ++ob, jb = j, vidx[(size_t)ob] = j, vval[(size_t)ob] = make_int2(0, 0);
                    if (c & 1) vval[(size_t)ob].y += b; else vval[(size_t)ob] = make_int2(b, vval[(size_t)ob].y + b);
                }
            }
        }
    });
}

// ---- balanced slabs (r6; TiledStream::perm) -------------------------------------------------------
// The padding of a round is the maximum over its 16 rows of their words in ONE slab; which contracted
// rows share a slab is free per tile.  Greedy, per tile: the contracted rows ("columns" of the tile's
// sub-matrix) most-covered first, each to the slab -- among those with room -- where the sum of the
// present loads of the rows it touches is smallest (ties: the lowest slab); columns without an entry in
// the tile fill what is left.  Counted on the c3 matrix: 1.62 -> 1.17 executed slots per word.
// Deterministic (the result is part of the stream both builders must agree on).
static inline int words_of_count(int64_t v) {  // FORM 1 words of one count (push_value / vrx_chunks)
    int n = 0;
    while (v != 0) {
        const uint64_t mag = (uint64_t)(v < 0 ? -v : v);
        const int len = 64 - __builtin_clzll(mag), sh = std::max(0, len - 3);
        const int64_t c = (int64_t)((mag >> sh) << sh);
        v -= v < 0 ? -c : c;
        ++n;
    }
    return n;
}

// (the greedy itself: vrx_host.cpp, vrx_balance_tile -- host only, built with AVX2 clones of its inner loops)
void vrx_balance_tile(const int32_t* rows, int64_t n_rows_tile, const int64_t* ptr, const int32_t* idx,
                      const uint8_t* words, int64_t n_contract, int n_slab, int slab_rows, int max_block,
                      int32_t* posmap, int32_t* perm);

// The host half of a tiled stream that needs nothing but the row pointer: the pieces long rows are cut
// into, the tile / slab geometry, and which piece sits at which tile position.  A function of its own so
// that device_build can run it -- and the balanced-slab greedy behind it -- on a helper thread while the
// counts are still being uploaded and transposed.
struct TileLayout {
    // what it was computed for
    const int64_t* ptr = nullptr;
    int64_t n_rows = -1, n_contract = -1, nnz = -1;
    int RW = 0, slab_rows_in = 0, form = -1, n_cu = -1;
    // the layout
    std::vector<int32_t> vptr, vrow_row, split_rows, rowmap;
    int64_t n_vrows = 0;
    bool split = false;
    int n_tile = 0, n_slab = 0, slab_rows = 0;
    // balanced slabs, when the greedy already ran (cell orientation, from the caller's arrays)
    bool greedy_done = false;
    std::vector<int32_t> posmap, perm, tile_of_row;
    double greedy_seconds = 0.0;
    bool matches(const int64_t* p, int64_t nr, int64_t nc, int64_t nz, int rw, int sr, int f, int cu) const {
        return ptr == p && n_rows == nr && n_contract == nc && nnz == nz && RW == rw && slab_rows_in == sr &&
               form == f && n_cu == cu;
    }
};

static int tile_layout(TileLayout& L, const int64_t* ptr, int64_t o_n_rows, int64_t o_n_contract, int64_t o_nnz,
                       int RW, int slab_rows, int form, int n_cu) {
    constexpr int G = 64 / VRX_LDS_LPE;
    L.ptr = ptr;
    L.n_rows = o_n_rows;
    L.n_contract = o_n_contract;
    L.nnz = o_nnz;
    L.RW = RW;
    L.slab_rows_in = slab_rows;
    L.form = form;
    L.n_cu = n_cu;
    L.slab_rows = slab_rows;
    L.n_slab = (int)((o_n_contract + slab_rows - 1) / slab_rows);
    // ---- pieces ------------------------------------------------------------------------
    const double mean = (double)o_nnz / (double)std::max<int64_t>(o_n_rows, 1);
    const int64_t cap = std::max<int64_t>(
        64, (int64_t)(mean * (double)env_int("VIREO_LDS_SPLIT_X10", 20) / 10.0 + 0.5));
    const bool reorder = env_int("VIREO_LDS_SORT", 1) != 0;
    std::vector<int32_t>& vptr = L.vptr;
    vptr.assign((size_t)o_n_rows + 1, 0);
    for (int64_t r = 0; r < o_n_rows; ++r) {
        const int64_t len = ptr[r + 1] - ptr[r];
        const int64_t P = reorder ? std::max<int64_t>(1, (len + cap - 1) / cap) : 1;
        if ((int64_t)vptr[(size_t)r] + P >= INT32_MAX) {
            vrx_set_error("tiled stream: too many row pieces");
            return VRX_ERR_UNSUPPORTED;
        }
        vptr[(size_t)r + 1] = vptr[(size_t)r] + (int32_t)P;
    }
    const int64_t n_vrows = vptr[(size_t)o_n_rows];
    L.n_vrows = n_vrows;
    L.split = n_vrows != o_n_rows;
    std::vector<int32_t>& split_rows = L.split_rows;  // rows cut into several pieces: folded by vrx_fold_split
    split_rows.clear();
    for (int64_t r = 0; r < o_n_rows; ++r)
        if (vptr[(size_t)r + 1] - vptr[(size_t)r] > 1) split_rows.push_back((int32_t)r);
    std::vector<int32_t>& vrow_row = L.vrow_row;
    vrow_row.assign((size_t)n_vrows, 0);
    for (int64_t r = 0; r < o_n_rows; ++r)
        for (int32_t v = vptr[(size_t)r]; v < vptr[(size_t)r + 1]; ++v) vrow_row[(size_t)v] = (int32_t)r;
    const int64_t tile_rows = VRX_LDS_WAVES * (int64_t)RW;
    L.n_tile = (int)((n_vrows + tile_rows - 1) / tile_rows);
    // Coarse shapes (clone mode: a few hundred variants, 10^5 cells): the (tile, slab) visits
    // are what the work list deals to the CUs, and a visit is not divisible.
    //  * fewer visits than CUs: shorter slabs, until every CU has one (c5 variant pass: ONE tile
    //    of 200 variants x 196 slabs of 1024 cells left 60 CUs idle -> 256 slabs of 784);
    //  * one or two slabs: the rows are spread over as many tiles as make whole rounds of CUs --
    //    the pieces are dealt round-robin to the waves anyway, only the number of tiles changes
    //    (c5 cell pass: 196 full tiles of 1024 cells -> 256 tiles of 782).
    if (n_cu > 0 && env_int("VIREO_LDS_FILL_CUS", 1) != 0) {
        auto spread_tiles = [&]() {  // (tiles must keep >= 4 rows per wave on average)
            const int64_t visits = (int64_t)L.n_tile * L.n_slab, rounds = (visits + n_cu - 1) / n_cu;
            const int64_t nt = rounds * n_cu / L.n_slab;
            if (nt > L.n_tile && nt * VRX_LDS_WAVES * G <= std::max<int64_t>(n_vrows, 1) * 4) L.n_tile = (int)nt;
        };
        if (L.n_slab <= 2) spread_tiles();
        if ((int64_t)L.n_tile * L.n_slab < n_cu) {
            const int64_t want = (n_cu + L.n_tile - 1) / L.n_tile;
            const int64_t sr = std::min<int64_t>(slab_rows, std::max<int64_t>(64, ((o_n_contract + want - 1) / want + 15) / 16 * 16));
            slab_rows = (int)sr;
            L.slab_rows = slab_rows;
            L.n_slab = (int)((o_n_contract + slab_rows - 1) / slab_rows);
            if ((int64_t)L.n_tile * L.n_slab < n_cu) spread_tiles();
        }
    }
    const int64_t n_wave = (int64_t)L.n_tile * VRX_LDS_WAVES;
    // ---- tile position -> piece (-1 = padding position) ---------------------------------
    std::vector<int32_t>& rowmap = L.rowmap;
    rowmap.assign((size_t)(n_wave * RW), -1);
    {
        std::vector<int32_t> order((size_t)n_vrows);
        for (int64_t v = 0; v < n_vrows; ++v) order[(size_t)v] = (int32_t)v;
        if (reorder) {
            auto piece_len = [&](int32_t v) {
                const int32_t r = vrow_row[(size_t)v];
                return (ptr[r + 1] - ptr[r]) / (vptr[(size_t)r + 1] - vptr[(size_t)r]);
            };
            std::stable_sort(order.begin(), order.end(),
                             [&](int32_t a, int32_t b) { return piece_len(a) > piece_len(b); });
        }
        const int64_t n_unit = (n_vrows + G - 1) / G;
        for (int64_t u = 0; u < n_unit; ++u) {
            const int64_t j = u / n_wave, i = u % n_wave;  // stratum j -> round j of the wave
            const int64_t w = reorder && (j & 1) ? n_wave - 1 - i : i;
            for (int g = 0; g < G && u * G + g < n_vrows; ++g)
                rowmap[(size_t)(w * RW + j * G + g)] = order[(size_t)(u * G + g)];
        }
    }
    L.slab_rows = slab_rows;
    return VRX_OK;
}

// The greedy of every tile (vrx_balance_tile), tiles in parallel: the tile of every unit (row, or piece where
// rows are cut), posmap / perm per tile.  `unit_ptr`, `idx`, `words`: the units' entries on the host.
static void greedy_tiles(const TileLayout& L, bool pieces, const int64_t* unit_ptr, const int32_t* idx,
                         const uint8_t* words, int64_t n_unit_rows, std::vector<int32_t>& posmap,
                         std::vector<int32_t>& perm, std::vector<int32_t>& tile_of_row) {
    const int64_t tile_pos = (int64_t)VRX_LDS_WAVES * L.RW, slots = (int64_t)L.n_slab * L.slab_rows;
    const int64_t n_contract = L.n_contract;
    const int max_block = env_int("VIREO_BALANCE_BLOCK", 64);
    posmap.resize((size_t)(L.n_tile * n_contract));
    perm.resize((size_t)(L.n_tile * slots));
    tile_of_row.assign((size_t)n_unit_rows, -1);
    auto unit_of = [&](int32_t v) { return pieces ? v : L.vrow_row[(size_t)v]; };
    for (int64_t pos = 0; pos < (int64_t)L.n_tile * tile_pos; ++pos)
        if (L.rowmap[(size_t)pos] >= 0) tile_of_row[(size_t)unit_of(L.rowmap[(size_t)pos])] = (int32_t)(pos / tile_pos);
    std::atomic<int64_t> next_tile{0};  // (tiles cost about the same: first come, first served)
    parallel_chunks(std::min<int64_t>(L.n_tile, balance_threads()), balance_threads(), [&](int64_t, int64_t, int) {
        std::vector<int32_t> rows;
        for (int64_t tl = next_tile++; tl < L.n_tile; tl = next_tile++) {
            rows.clear();
            for (int64_t pos = tl * tile_pos; pos < (tl + 1) * tile_pos; ++pos)
                if (L.rowmap[(size_t)pos] >= 0) rows.push_back(unit_of(L.rowmap[(size_t)pos]));
            vrx_balance_tile(rows.data(), (int64_t)rows.size(), unit_ptr, idx, words, n_contract, L.n_slab,
                             L.slab_rows, max_block, posmap.data() + tl * n_contract, perm.data() + tl * slots);
        }
    });
}

// one byte per entry: the FORM 1 words of the caller's (ad, dp), on all host threads; false if an index is
// outside [0, n_contract) (the caller's arrays may not have been validated yet)
static bool host_words(const HostCounts& hc, int64_t nnz, int64_t n_contract, std::vector<uint8_t>& words) {
    words.resize((size_t)nnz);
    std::atomic<bool> bad{false};
    parallel_chunks(nnz, host_threads(), [&](int64_t e0, int64_t e1, int) {
        bool b = false;
        for (int64_t e = e0; e < e1; ++e) {
            b |= hc.idx[e] < 0 || hc.idx[e] >= n_contract;
            words[(size_t)e] = (uint8_t)std::min(words_of_count(hc.ad[e]) + words_of_count((int64_t)hc.dp[e] - hc.ad[e]), 255);
        }
        if (b) bad = true;
    });
    return !bad;
}

// may this layout be balanced on the host from the caller's arrays alone?  (the same conditions build_tiled
// applies; pieces need the device's copy of the rows)
static bool host_balance_applies(const TileLayout& L) {
    return L.form == 1 && L.n_slab > 1 && L.n_contract < ((int64_t)1 << 24) && L.n_vrows < ((int64_t)1 << 31) && !L.split;
}

static int build_tiled(Orient& o, const int64_t* ptr, const int32_t* idx, const int2* val,
                       int RW, int slab_rows, bool guard, int form, int mode, hipStream_t s,
                       int n_cu, const DevRows* dev = nullptr, int64_t virt_rows = -1,
                       int64_t virt_contract = -1, int64_t virt_nnz = -1, const HostCounts* hc = nullptr,
                       const HostWords* hw = nullptr, TileLayout* pre = nullptr) {
    constexpr int G = 64 / VRX_LDS_LPE, U = VRX_LDS_U;
    const int NR = RW / G;
    TiledStream& t = o.tiled;
    // (virt_rows >= 0: ptr / idx / val are the VIRTUAL rows of the variant pass, vrx_build.h)
    t.virt = virt_rows >= 0;
    const int64_t o_n_rows = t.virt ? virt_rows : o.n_rows;
    const int64_t o_n_contract = t.virt ? virt_contract : o.n_contract;
    const int64_t o_nnz = t.virt ? virt_nnz : o.nnz;
    t.n_contract = o_n_contract;
    t.form = form;
    t.rw = RW;
    t.slab_rows = slab_rows;
    t.n_slab = (int)((o_n_contract + slab_rows - 1) / slab_rows);
    // ---- pieces, tile / slab geometry, tile position -> piece (tile_layout) ---------------------
    TileLayout own_layout;
    const bool have_pre = pre && pre->matches(ptr, o_n_rows, o_n_contract, o_nnz, RW, slab_rows, form, n_cu);
    TileLayout& L = have_pre ? *pre : own_layout;
    if (!have_pre) {
        const int rc_layout = tile_layout(L, ptr, o_n_rows, o_n_contract, o_nnz, RW, slab_rows, form, n_cu);
        if (rc_layout) return rc_layout;
    }
    std::vector<int32_t>&vptr = L.vptr, &vrow_row = L.vrow_row, &split_rows = L.split_rows, &rowmap = L.rowmap;
    const int64_t n_vrows = L.n_vrows;
    t.n_vrows = n_vrows;
    t.split = L.split;
    t.n_split = (int64_t)split_rows.size();
    t.n_tile = L.n_tile;
    slab_rows = L.slab_rows;
    t.slab_rows = slab_rows;
    t.n_slab = L.n_slab;
    const int64_t n_wave = (int64_t)t.n_tile * VRX_LDS_WAVES;
    const int PH = form == 2 ? 2 : 1;  // phases of a round (form 2: AD entries, then BD entries)
    const int64_t per_wave = (int64_t)t.n_slab * NR * PH + 1;
    std::vector<int64_t> wave_start((size_t)n_wave), wave_len((size_t)n_wave, 0);
    std::vector<int32_t> bnd((size_t)(n_wave * per_wave));
    std::atomic<bool> too_long{false};
    // One wave's stream: walks its RW pieces slab by slab, records the (slab, round) offsets
    // and appends the words to the wave's own buffer.  Waves are independent.
    const bool parity_order = env_int("VIREO_LDS_PARITY", 1) != 0;
    // the entry bit that selects the LDS bank half of a 128-B dense row: half (form 1), parity
    // of the slab-local index (variant pass); none for the 256-B rows of the (ad, dp) cell pass
    const int bit_shift = form != 0 ? 7 : (mode == 0 ? 22 : -1);
    // form 1 words carry the LDS address of their half row: the slab starts behind the rings
    const uint32_t f1_base = (uint32_t)VRX_LDS_WAVES * VRX_RING * 4u, pad_word = form != 0 ? f1_base : 0u;
    // FORM 1 value field: the top 14 bits of the IEEE double (sign, exponent, 2 mantissa bits).
    // A value with more than three significant bits becomes several entries (9 = 8 + 1, ...).
    auto push_value = [](std::vector<uint32_t>& out, int64_t v, uint32_t off) {
        while (v != 0) {
            const uint64_t mag = (uint64_t)(v < 0 ? -v : v);
            const int len = 64 - __builtin_clzll(mag), sh = std::max(0, len - 3);
            const int64_t c = (int64_t)((mag >> sh) << sh) * (v < 0 ? -1 : 1);
            const double d = (double)c;
            uint64_t bits;
            std::memcpy(&bits, &d, 8);
            out.push_back((uint32_t)(bits >> 50) << 18 | off);
            v -= c;
        }
    };
    if (dev) {  // ---- the stream is built on the device --------------------------------------
        DevBuf<int32_t> d_rowmap, d_vptr, d_vrow, rlen, d_too;
        DevBuf<uint32_t> seg_lo, seg_hi;  // entry offsets (< 2^32: device_build's limit)
        VRX_REQUIRE(o_nnz < (int64_t)UINT32_MAX, "tiled stream: more than 2^32 - 1 entries");
        VRX_HIP(d_rowmap.upload(rowmap.data(), rowmap.size(), s));
        VRX_HIP(d_vptr.upload(vptr.data(), vptr.size(), s));
        VRX_HIP(d_vrow.upload(vrow_row.data(), vrow_row.size(), s));
        VrxTileArgs A;
        A.ptr = dev->ptr;
        A.idx = dev->idx;
        A.val = dev->val;
        A.rowmap = d_rowmap.p;
        A.vptr = d_vptr.p;
        A.vrow_row = d_vrow.p;
        A.RW = RW;
        A.NR = NR;
        A.G = G;
        A.U = U;
        A.n_slab = t.n_slab;
        A.slab_rows = slab_rows;
        A.form = form;
        A.PH = PH;
        A.bit_shift = bit_shift;
        A.pairing = parity_order ? 1 : 0;
        A.xor_partner = 24 / VRX_LDS_LPE;
        A.f1_base = f1_base;
        A.pad_word = pad_word;
        A.n_wave = n_wave;
#ifdef VRX_CAP_PROBE
        A.cap = t.virt ? env_int("VIREO_CAP_PROBE_VAR", 0) : env_int("VIREO_CAP_PROBE_CELL", 0);
#endif
        // ---- balanced slabs: per-tile permutation of the contracted rows, entries relabelled + re-sorted
        DevBuf<int32_t> d_idx2, d_tile_of_row, d_posmap;
        DevBuf<int2> d_val2;
        t.balanced = false;
        t.perm.release();
        DevBuf<int64_t> d_pptr;       // (split rows: the pieces as rows of their own, see vrx_build_pieces)
        DevBuf<int32_t> d_pidx, d_iota;
        DevBuf<int2> d_pval;
        if (form == 1 && t.want_balance && t.n_slab > 1 && o_n_contract < ((int64_t)1 << 24) &&
            n_vrows < ((int64_t)1 << 31) && (!t.split || env_int("VIREO_BALANCE_SPLIT", 1) != 0)) {
            const int64_t tile_pos = (int64_t)VRX_LDS_WAVES * RW, slots = (int64_t)t.n_slab * slab_rows;
            const bool timing = env_int("VIREO_BUILD_TIMING", 0) != 0;
            auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
            double tm0 = now();
            const double tm_begin = tm0;
            auto lap = [&](const char* what) {
                if (!timing) return;
                (void)hipStreamSynchronize(s);
                const double t1 = now();
                fprintf(stderr, "[vrx build] balanced slabs (mode %d): %-28s %.3f s\n", mode, what, t1 - tm0);
                tm0 = t1;
            };
            // what the greedy reads: every entry's contracted index and its FORM 1 word count.  The cell
            // orientation's rows are the caller's CSC arrays; otherwise the indices and one byte per entry
            // (vrx_build_words) come back from the device
            std::vector<int32_t> h_idx;
            std::vector<uint8_t> h_words;
            // Rows cut into pieces (heavy-tailed data): the unit of everything below is the PIECE -- its
            // entries are copied out as a row of their own (entry k of the row -> piece k % P), so that
            // every unit belongs to exactly one tile and is relabelled by that tile's permutation
            const bool pieces = t.split;
            std::vector<int64_t> pptr;
            const int64_t* u_ptr = ptr;         // row pointer of the units (host) ...
            const int64_t* du_ptr = dev->ptr;   // ... and on the device, with their entries
            const int32_t* du_idx = dev->idx;
            const int2* du_val = dev->val;
            const int64_t n_unit_rows = pieces ? n_vrows : o_n_rows;
            if (pieces) {
                pptr.assign((size_t)n_vrows + 1, 0);
                for (int64_t r = 0; r < o_n_rows; ++r) {
                    const int64_t L = ptr[r + 1] - ptr[r];
                    const int32_t v0 = vptr[(size_t)r], P = vptr[(size_t)r + 1] - v0;
                    for (int32_t q = 0; q < P; ++q) pptr[(size_t)(v0 + q) + 1] = (L - q + P - 1) / P;
                }
                for (int64_t v = 0; v < n_vrows; ++v) pptr[(size_t)v + 1] += pptr[(size_t)v];
                VRX_HIP(d_pptr.upload(pptr.data(), pptr.size(), s));
                VRX_HIP(d_pidx.alloc((size_t)o_nnz));
                VRX_HIP(d_pval.alloc((size_t)o_nnz));
                vrx_build_pieces<<<(unsigned)((o_nnz + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                    o_nnz, o_n_rows, dev->ptr, d_vptr.p, d_pptr.p, dev->idx, dev->val, d_pidx.p, d_pval.p);
                VRX_HIP(hipGetLastError());
                u_ptr = pptr.data();
                du_ptr = d_pptr.p;
                du_idx = d_pidx.p;
                du_val = d_pval.p;
                hc = nullptr;  // (the caller's arrays are the whole rows)
                hw = nullptr;
            }
            const bool pre_done = have_pre && L.greedy_done && !pieces;  // device_build's helper thread already did it
            // the tile position of every unit (what both routes of the greedy start from)
            std::vector<int32_t> tile_of_row_own((size_t)n_unit_rows, -1), tpos((size_t)n_unit_rows, -1);
            for (int64_t pos = 0; pos < (int64_t)t.n_tile * tile_pos; ++pos)
                if (rowmap[(size_t)pos] >= 0) {
                    const int32_t u = pieces ? rowmap[(size_t)pos] : vrow_row[(size_t)rowmap[(size_t)pos]];
                    tile_of_row_own[(size_t)u] = (int32_t)(pos / tile_pos);
                    tpos[(size_t)u] = (int32_t)pos;
                }
            VRX_HIP(d_tile_of_row.upload(tile_of_row_own.data(), tile_of_row_own.size(), s));
            // ---- the greedy on the device (vrx_build.h, vrx_balance_greedy): no entry leaves the GPU ----------
            const int max_block = env_int("VIREO_BALANCE_BLOCK", 64);
            const VrxBalBlocks blocks = vrx_bal_blocks(t.n_slab, max_block);
            const bool check = env_int("VIREO_BALANCE_CHECK", 0) != 0;
            const char* gm = getenv("VIREO_BALANCE_GREEDY");
            const int64_t n_cols_all = (int64_t)t.n_tile * o_n_contract, n_groups = (int64_t)t.n_tile * blocks.nb;
            bool dev_greedy = !(gm && !strcmp(gm, "host")) && !pre_done && blocks.bs <= 64 &&
                              (tile_pos + 1) * 64 + 2 * VRX_BAL_BATCH * 4 <= 160 * 1024 && tile_pos <= 4095 && n_cols_all < (int64_t)INT32_MAX &&
                              n_groups < ((int64_t)1 << 20) && (int64_t)t.n_tile * tile_pos < (int64_t)INT32_MAX;
            // (the sort buffers of the greedy's preparation serve the relabel below as well: multi-GB
            //  allocations and releases are what a large build spends its time on)
            DevBuf<uint64_t> k_in, k_out;
            DevBuf<uint32_t> v_in, v_out;
            VRX_HIP(k_in.alloc((size_t)o_nnz));
            VRX_HIP(k_out.alloc((size_t)o_nnz));
            VRX_HIP(v_in.alloc((size_t)o_nnz));
            VRX_HIP(v_out.alloc((size_t)o_nnz));
            if (dev_greedy) {
                DevBuf<int32_t> d_tpos, d_flags;
                DevBuf<uint64_t> ok_in, ok_out;
                DevBuf<uint32_t> d_cptr, d_deg, d_ostart, d_ovals;
                DevBuf<uint64_t>&bk_in = k_in, &bk_out = k_out;
                DevBuf<uint32_t>&bv_in = v_in, &bv_out = v_out;
                DevBuf<int64_t> d_seg;
                DevBuf<char> tmp;
                int cbits = 1, tbits = 1, gbits = 1;
                while (((int64_t)1 << cbits) < o_n_contract) ++cbits;
                while (((int64_t)1 << tbits) < (int64_t)t.n_tile + 1) ++tbits;
                while (((int64_t)1 << gbits) < n_groups) ++gbits;
                const unsigned nbe = (unsigned)((o_nnz + VRX_BLOCK - 1) / VRX_BLOCK);
                const unsigned nbc = (unsigned)((n_cols_all + 1 + VRX_BLOCK - 1) / VRX_BLOCK);
                VRX_HIP(d_tpos.upload(tpos.data(), tpos.size(), s));
                const int32_t zero2[2] = {0, 0};
                VRX_HIP(d_flags.upload(zero2, 2, s));
                vrx_bal_keys<<<nbe, VRX_BLOCK, 0, s>>>(o_nnz, n_unit_rows, du_ptr, du_idx, du_val, d_tpos.p, (int)tile_pos,
                                                       t.n_tile, cbits, bk_in.p, bv_in.p);
                VRX_HIP(hipGetLastError());
                size_t tmp_bytes = 0, tmp2 = 0, tmp3 = 0;
                VRX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, bk_in.p, bk_out.p, bv_in.p, bv_out.p,
                                                           (size_t)o_nnz, 0, cbits + tbits, s));
                VRX_HIP(ok_in.alloc((size_t)n_cols_all));
                VRX_HIP(ok_out.alloc((size_t)n_cols_all));
                VRX_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp2, ok_in.p, ok_out.p, (size_t)n_cols_all, 0,
                                                          cbits + 12 + gbits, s));
                VRX_HIP(d_deg.alloc((size_t)n_cols_all));
                VRX_HIP(d_ostart.alloc((size_t)n_cols_all));
                VRX_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp3, d_deg.p, d_ostart.p, (size_t)n_cols_all, s));
                VRX_HIP(tmp.alloc(std::max(tmp_bytes, std::max(tmp2, tmp3))));
                VRX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, bk_in.p, bk_out.p, bv_in.p, bv_out.p,
                                                           (size_t)o_nnz, 0, cbits + tbits, s));
                VRX_HIP(d_cptr.alloc((size_t)n_cols_all + 1));
                vrx_bal_cptr<<<nbc, VRX_BLOCK, 0, s>>>(n_cols_all, o_n_contract, o_nnz, bk_out.p, cbits, d_cptr.p);
                VRX_HIP(hipGetLastError());
                vrx_bal_order_keys<<<nbc, VRX_BLOCK, 0, s>>>(n_cols_all, o_n_contract, d_cptr.p, slab_rows, blocks.bs,
                                                             blocks.nb, cbits, ok_in.p, d_flags.p);
                VRX_HIP(hipGetLastError());
                VRX_HIP(hipcub::DeviceRadixSort::SortKeys(tmp.p, tmp2, ok_in.p, ok_out.p, (size_t)n_cols_all, 0,
                                                          cbits + 12 + gbits, s));
                VRX_HIP(d_seg.alloc((size_t)n_groups + 1));
                vrx_bal_groups<<<(unsigned)((n_groups + 1 + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                    n_groups, n_cols_all, ok_out.p, 12 + cbits, d_seg.p);
                VRX_HIP(hipGetLastError());
                vrx_bal_degrees<<<nbc, VRX_BLOCK, 0, s>>>(n_cols_all, ok_out.p, cbits, d_deg.p);
                VRX_HIP(hipGetLastError());
                VRX_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, tmp3, d_deg.p, d_ostart.p, (size_t)n_cols_all, s));
                uint32_t n_stream = 0;  // the entries that count = first entry of the column behind the last one
                VRX_HIP(hipMemcpyAsync(&n_stream, d_cptr.p + n_cols_all, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                VRX_HIP(d_ovals.alloc((size_t)o_nnz));
                vrx_bal_stream<<<nbc, VRX_BLOCK, 0, s>>>(n_cols_all, o_n_contract, ok_out.p, cbits, blocks.nb, d_cptr.p,
                                                         bv_out.p, d_ostart.p, d_ovals.p);
                VRX_HIP(hipGetLastError());
                VRX_HIP(d_posmap.alloc((size_t)n_cols_all));
                VRX_HIP(t.perm.alloc((size_t)(t.n_tile * slots)));
                VRX_HIP(hipMemsetAsync(t.perm.p, 0, (size_t)(t.n_tile * slots) * sizeof(int32_t), s));
                VRX_HIP(hipStreamSynchronize(s));  // (n_stream)
                const size_t lds = (size_t)(tile_pos + 1) * 64 + 2 * VRX_BAL_BATCH * sizeof(uint32_t);
                VRX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(vrx_balance_greedy),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                vrx_balance_greedy<<<(unsigned)n_groups, 64, lds, s>>>(ok_out.p, d_seg.p, d_ostart.p, d_ovals.p,
                                                                      (int64_t)n_stream, o_n_contract, cbits, t.n_slab,
                                                                      slab_rows, blocks.bs, blocks.nb, (int)tile_pos,
                                                                      d_posmap.p, t.perm.p, d_flags.p + 1);
                VRX_HIP(hipGetLastError());
                int32_t flags[2] = {0, 0};
                VRX_HIP(hipMemcpyAsync(flags, d_flags.p, sizeof flags, hipMemcpyDeviceToHost, s));
                VRX_HIP(hipStreamSynchronize(s));
                lap("greedy on the device");
                if (flags[0] || flags[1]) {  // a column deeper than 4095 tile rows / a block without room: cannot be
                    dev_greedy = false;
                    if (timing) fprintf(stderr, "[vrx build] balanced slabs: device greedy declined (%d, %d)\n", flags[0], flags[1]);
                }
            }
            std::vector<int32_t> posmap_own, perm_own, tor_unused;
            if (!dev_greedy || check) {  // ---- the greedy on host threads (the specification) -----------------
                if (!hc && !hw && !pre_done) {
                    DevBuf<uint8_t> d_words;
                    VRX_HIP(d_words.alloc((size_t)o_nnz));
                    vrx_build_words<<<(unsigned)((o_nnz + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(o_nnz, du_val, d_words.p);
                    VRX_HIP(hipGetLastError());
                    h_idx.resize((size_t)o_nnz);
                    h_words.resize((size_t)o_nnz);
                    VRX_HIP(hipMemcpyAsync(h_idx.data(), du_idx, (size_t)o_nnz * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                    VRX_HIP(hipMemcpyAsync(h_words.data(), d_words.p, (size_t)o_nnz, hipMemcpyDeviceToHost, s));
                    VRX_HIP(hipStreamSynchronize(s));
                }
                lap("download rows");
                if (!pre_done) {
                    if (hc) (void)host_words(*hc, o_nnz, o_n_contract, h_words);  // (validated by now)
                    const int32_t* g_idx = hc ? hc->idx : hw ? hw->idx : h_idx.data();
                    const uint8_t* g_words = hw ? hw->words : h_words.data();
                    greedy_tiles(L, pieces, u_ptr, g_idx, g_words, n_unit_rows, posmap_own, perm_own, tor_unused);
                }
                std::vector<int32_t>&posmap = pre_done ? L.posmap : posmap_own, &perm = pre_done ? L.perm : perm_own;
                if (pre_done && timing)
                    fprintf(stderr, "[vrx build] balanced slabs (mode %d): greedy ran beside the upload     %.3f s (hidden)\n", mode, L.greedy_seconds);
                lap("greedy (host threads)");
                std::vector<int32_t>().swap(h_idx);
                std::vector<uint8_t>().swap(h_words);
                if (dev_greedy) {  // VIREO_BALANCE_CHECK=1: the device's result against the specification, bit for bit
                    std::vector<int32_t> dp((size_t)n_cols_all), dq((size_t)(t.n_tile * slots));
                    VRX_HIP(hipMemcpyAsync(dp.data(), d_posmap.p, dp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                    VRX_HIP(hipMemcpyAsync(dq.data(), t.perm.p, dq.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
                    VRX_HIP(hipStreamSynchronize(s));
                    for (int64_t i = 0; i < n_cols_all; ++i)
                        if (dp[(size_t)i] != posmap[(size_t)i]) {
                            vrx_set_error("balanced slabs: the device greedy differs from the host's at tile %lld, contracted "
                                          "row %lld (%d against %d)", (long long)(i / o_n_contract),
                                          (long long)(i % o_n_contract), dp[(size_t)i], posmap[(size_t)i]);
                            return VRX_ERR_UNSUPPORTED;
                        }
                    if (dq != perm) {
                        vrx_set_error("balanced slabs: the device greedy's slab lists differ from the host's");
                        return VRX_ERR_UNSUPPORTED;
                    }
                    if (timing) fprintf(stderr, "[vrx build] balanced slabs (mode %d): device greedy == host greedy (%lld columns)\n", mode, (long long)n_cols_all);
                } else {
                    VRX_HIP(d_posmap.upload(posmap.data(), posmap.size(), s));
                    VRX_HIP(t.perm.upload(perm.data(), perm.size(), s));
                }
            }
            const unsigned nbe = (unsigned)((o_nnz + VRX_BLOCK - 1) / VRX_BLOCK);
            int rbits = 1, pbits = 1;  // key = row << pbits | position: as few radix passes as the sizes need
            while (((int64_t)1 << rbits) < n_unit_rows) ++rbits;
            while (((int64_t)1 << pbits) < std::max<int64_t>(slots, o_n_contract)) ++pbits;
            vrx_build_relabel<<<nbe, VRX_BLOCK, 0, s>>>(o_nnz, n_unit_rows, o_n_contract, du_ptr, du_idx,
                                                        d_tile_of_row.p, d_posmap.p, pbits, k_in.p, v_in.p);
            VRX_HIP(hipGetLastError());
            size_t tmp_bytes = 0;
            VRX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, k_in.p, k_out.p, v_in.p, v_out.p,
                                                       (size_t)o_nnz, 0, pbits + rbits, s));
            DevBuf<char> tmp;
            VRX_HIP(tmp.alloc(tmp_bytes));
            VRX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, k_in.p, k_out.p, v_in.p, v_out.p,
                                                       (size_t)o_nnz, 0, pbits + rbits, s));
            VRX_HIP(d_idx2.alloc((size_t)o_nnz));
            VRX_HIP(d_val2.alloc((size_t)o_nnz));
            vrx_build_relabel_gather<<<nbe, VRX_BLOCK, 0, s>>>(o_nnz, k_out.p, v_out.p, du_val, pbits, d_idx2.p, d_val2.p);
            VRX_HIP(hipGetLastError());
            if (pieces) {  // the stream's rows are the pieces now: one piece per "row", nothing left to cut
                std::vector<int32_t> iota((size_t)n_vrows + 1);
                for (int64_t v = 0; v <= n_vrows; ++v) iota[(size_t)v] = (int32_t)v;
                VRX_HIP(d_iota.upload(iota.data(), iota.size(), s));
                VRX_HIP(hipStreamSynchronize(s));
                A.ptr = d_pptr.p;
                A.vptr = d_iota.p;
                A.vrow_row = d_iota.p;
            }
            VRX_HIP(hipStreamSynchronize(s));
            d_pidx.release();
            d_pval.release();
            lap("upload + relabel + sort");
            t.balance_seconds = now() - tm_begin;
            A.idx = d_idx2.p;
            A.val = d_val2.p;
            t.balanced = true;
        }
        const int64_t n_pos = n_wave * t.n_slab * RW, nsr = (int64_t)t.n_slab * NR * PH;
        VRX_REQUIRE(n_pos < INT32_MAX * (int64_t)VRX_BLOCK, "tiled stream: too many segments");
        VRX_HIP(seg_lo.alloc((size_t)n_pos));
        VRX_HIP(seg_hi.alloc((size_t)n_pos));
        VRX_HIP(rlen.alloc((size_t)(n_wave * nsr)));
        VRX_HIP(d_too.alloc(1));
        VRX_HIP(hipMemsetAsync(d_too.p, 0, sizeof(int32_t), s));
        VRX_HIP(t.bnd.alloc((size_t)(n_wave * per_wave)));
        DevBuf<int64_t> d_wlen;
        VRX_HIP(d_wlen.alloc((size_t)n_wave));
        vrx_build_count<<<(unsigned)((n_pos + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            A, seg_lo.p, seg_hi.p, rlen.p);
        VRX_HIP(hipGetLastError());
        vrx_build_offsets<<<(unsigned)((n_wave + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            A, rlen.p, t.bnd.p, d_wlen.p, d_too.p);
        VRX_HIP(hipGetLastError());
        int32_t h_too = 0;
        VRX_HIP(hipMemcpyAsync(wave_len.data(), d_wlen.p, (size_t)n_wave * sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
        VRX_HIP(hipMemcpyAsync(&h_too, d_too.p, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        VRX_HIP(hipStreamSynchronize(s));
        if (h_too) {
            vrx_set_error("tiled stream: wave stream >= 2^31 words");
            return VRX_ERR_UNSUPPORTED;
        }
        int64_t total = 0, longest_wave = 0;
        for (int64_t w = 0; w < n_wave; ++w) {
            wave_start[(size_t)w] = total;
            total += wave_len[(size_t)w];
            longest_wave = std::max(longest_wave, wave_len[(size_t)w]);
        }
        t.pad_ratio = o_nnz > 0 ? (double)total / (double)o_nnz : 0.0;
        t.imbalance = total > 0 ? (double)longest_wave * (double)n_wave / (double)total : 1.0;
        if (guard && t.pad_ratio * std::max(1.0, t.imbalance / 1.5) > (double)env_int("VIREO_LDS_MAX_PAD", 3)) {
            t.bnd.release();
            t.ready = false;
            return VRX_OK;
        }
        VRX_HIP(t.ent.alloc((size_t)total + 8));
        VRX_HIP(hipMemsetAsync(t.ent.p + total, 0, 8 * sizeof(uint32_t), s));
        VRX_HIP(t.wave_start.upload(wave_start.data(), wave_start.size(), s));
        const int64_t n_fill = n_wave * nsr * (G / 2);
        VRX_REQUIRE(n_fill < INT32_MAX * (int64_t)VRX_BLOCK, "tiled stream: too many rounds");
        vrx_build_fill<<<(unsigned)((n_fill + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            A, seg_lo.p, seg_hi.p, rlen.p, t.bnd.p, t.wave_start.p, t.ent.p);
        VRX_HIP(hipGetLastError());
        VRX_HIP(t.rowmap.upload(rowmap.data(), rowmap.size(), s));
        if (t.split) {
            VRX_HIP(t.vptr.upload(vptr.data(), vptr.size(), s));
            VRX_HIP(t.split_rows.upload(split_rows.data(), split_rows.size(), s));
        }
        VRX_HIP(hipMemcpyAsync(bnd.data(), t.bnd.p, bnd.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        VRX_HIP(hipStreamSynchronize(s));
        const int rc = plan_items(t, bnd.data(), n_wave, NR * PH, n_cu, rowmap.data(), mode, s);
        if (rc) return rc;
        t.ready = true;
        return VRX_OK;
    }
    std::vector<std::vector<uint32_t>> wave_words((size_t)n_wave);
    auto walk = [&](int64_t w) {
        std::vector<uint32_t>& dst = wave_words[(size_t)w];
        dst.reserve((size_t)((double)o_nnz / (double)n_wave * 2.3) + 1024);
        const int32_t* rm = rowmap.data() + w * RW;
        std::vector<int64_t> cursor((size_t)RW);
        std::vector<uint32_t> segw[G], second;
        for (int c = 0; c < RW; ++c) cursor[(size_t)c] = rm[c] >= 0 ? ptr[vrow_row[(size_t)rm[c]]] : 0;
        int32_t* bw = bnd.data() + w * per_wave;
        int64_t rel = 0;
        for (int sl = 0; sl < t.n_slab; ++sl) {
            const int64_t lim = (int64_t)(sl + 1) * slab_rows, base = (int64_t)sl * slab_rows;
            for (int r = 0; r < NR; ++r) {
                if (rel >= INT32_MAX - 4096) {
                    too_long = true;
                    return;
                }
                // the groups' segments of this slab
                int64_t s_lo[G], s_hi[G], s_step[G];
                for (int g = 0; g < G; ++g) {
                    const int32_t v = rm[r * G + g];
                    s_lo[g] = s_hi[g] = 0;
                    s_step[g] = 1;
                    if (v >= 0) {
                        const int32_t row = vrow_row[(size_t)v];
                        int64_t hi = cursor[(size_t)(r * G + g)];
                        const int64_t seg = hi, stop = ptr[row + 1];
                        while (hi < stop && idx[hi] < lim) ++hi;
                        cursor[(size_t)(r * G + g)] = hi;
                        // this piece's share of the row's slab segment [seg, hi)
                        s_step[g] = vptr[(size_t)row + 1] - vptr[(size_t)row];
                        s_lo[g] = seg + ((int64_t)(v - vptr[(size_t)row]) + sl) % s_step[g];
                        s_hi[g] = hi;
                    }
                }
                for (int ph = 0; ph < PH; ++ph) {
                int64_t longest = 0;
                for (int g = 0; g < G; ++g) {
                    std::vector<uint32_t>& sw = segw[g];
                    sw.clear();
                    for (int64_t e = s_lo[g]; e < s_hi[g]; e += s_step[g]) {
                        if (form == 0) {
                            sw.push_back(((uint32_t)(idx[e] - base) << 22) |
                                         ((uint32_t)val[e].x << 11) | (uint32_t)val[e].y);
                        } else if (form == 1) {
                            const uint32_t at = f1_base + (uint32_t)(idx[e] - base) * 256u;
                            push_value(sw, val[e].x, at);
                            push_value(sw, (int64_t)val[e].y - val[e].x, at + 128u);
                        } else {  // form 2: AD entries in phase 0, BD entries in phase 1
                            const uint32_t at = f1_base + (uint32_t)(idx[e] - base) * 128u;
                            push_value(sw, ph == 0 ? (int64_t)val[e].x : (int64_t)val[e].y - val[e].x, at);
                        }
                    }
                    longest = std::max<int64_t>(longest, (int64_t)sw.size());
                }
                // Bank conflicts.  The dense rows are 128 B (ID_prob rows; the AD / BD half
                // rows), so the LDS bank of a slice depends on one address bit of the entry
                // (index parity / half).  ds_read_b128 serves lane groups {0,3,5,6}, {1,2,4,7}
                // (+8) together, and the two groups with the same slice rotation (g & 1) collide
                // whenever that bit agrees.  The order of a segment's entries is free, and the
                // zero words that pad a segment to the round's length can point at either
                // parity: group P walks its bit-0 entries (then bit-0 padding) up to a split
                // position z and its bit-1 entries after it, its partner Q the other way round.
                // z exists whenever the pair's bit-0 entries and its bit-1 entries each fit
                // into the round, i.e. almost always.
                if (parity_order && bit_shift >= 0) {
                    for (int g = 0; g < G; ++g) {
                        // the group that shares g's rotation inside g's service group holds
                        // lanes ^ 24 (ds_read_b128: {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32)
                        const int q = g ^ (24 / VRX_LDS_LPE);
                        if (q < g) continue;
                        std::vector<uint32_t> part[2][2];  // [P / Q][bit]
                        for (int m = 0; m < 2; ++m)
                            for (uint32_t wd : segw[m ? q : g]) part[m][(wd >> bit_shift) & 1u].push_back(wd);
                        const int64_t p0 = (int64_t)part[0][0].size(), p1 = (int64_t)part[0][1].size();
                        const int64_t q0 = (int64_t)part[1][0].size(), q1 = (int64_t)part[1][1].size();
                        const int64_t zlo = std::max(p0, q1), zhi = std::min(longest - p1, longest - q0);
                        if (zlo > zhi) {  // does not fit: opposite orders, as far as that goes
                            segw[g] = part[0][0];
                            segw[g].insert(segw[g].end(), part[0][1].begin(), part[0][1].end());
                            segw[q] = part[1][1];
                            segw[q].insert(segw[q].end(), part[1][0].begin(), part[1][0].end());
                            continue;
                        }
                        const int64_t z = (zlo + zhi) / 2;
                        const uint32_t pad0 = form != 0 ? f1_base : 0u, pad1 = pad0 | 1u << bit_shift;
                        auto lay = [&](std::vector<uint32_t>& out, const std::vector<uint32_t>& first,
                                       uint32_t pad_first, const std::vector<uint32_t>& rest,
                                       uint32_t pad_rest) {
                            out = first;
                            out.resize((size_t)z, pad_first);
                            out.insert(out.end(), rest.begin(), rest.end());
                            out.resize((size_t)longest, pad_rest);
                        };
                        lay(segw[g], part[0][0], pad0, part[0][1], pad1);
                        lay(segw[q], part[1][1], pad1, part[1][0], pad0);
                    }
                }
                // offset | entries in the last trip (0 = full): the kernel skips the padding
                bw[((int64_t)sl * NR + r) * PH + ph] = (int32_t)(rel | (longest % U));
                longest = (longest + U - 1) / U * U;
                dst.resize((size_t)(rel + longest * G));
                for (int64_t j = 0; j < longest; ++j)
                    for (int g = 0; g < G; ++g)  // padding: value 0 (forms 1, 2: at the slab's first row)
                        dst[(size_t)(rel + vrx_trip_slot(j, g, G, U, form))] =
                            j < (int64_t)segw[g].size() ? segw[g][(size_t)j] : pad_word;
                rel += longest * G;  // (a multiple of 64 words: streams stay 16-B aligned)
                }
            }
        }
        bw[(int64_t)t.n_slab * NR * PH] = (int32_t)rel;
        wave_len[(size_t)w] = rel;
    };
    parallel_chunks(n_wave, host_threads(), [&](int64_t b0, int64_t e0, int) {
        for (int64_t w = b0; w < e0; ++w) walk(w);
    });
    if (too_long) {
        vrx_set_error("tiled stream: wave stream >= 2^31 words");
        return VRX_ERR_UNSUPPORTED;
    }
    int64_t total = 0, longest_wave = 0;
    for (int64_t w = 0; w < n_wave; ++w) {
        wave_start[(size_t)w] = total;
        total += wave_len[(size_t)w];
        longest_wave = std::max(longest_wave, wave_len[(size_t)w]);
    }
    // Guard: past VIREO_LDS_MAX_PAD stream words per entry (default 3; padding x imbalance of
    // the slowest wave) the global-gather pass is the faster one, so no stream is kept.
    t.pad_ratio = o_nnz > 0 ? (double)total / (double)o_nnz : 0.0;
    t.imbalance = total > 0 ? (double)longest_wave * (double)n_wave / (double)total : 1.0;
    if (guard && t.pad_ratio * std::max(1.0, t.imbalance / 1.5) > (double)env_int("VIREO_LDS_MAX_PAD", 3)) {
        t.ready = false;
        return VRX_OK;
    }
    // the waves' buffers go to the device back to back (+ slack for the last dwordx4 refill)
    VRX_HIP(t.ent.alloc((size_t)total + 8));
    VRX_HIP(hipMemsetAsync(t.ent.p + total, 0, 8 * sizeof(uint32_t), s));
    {
        std::unique_ptr<uint32_t[]> all(new uint32_t[(size_t)total + 1]);
        parallel_chunks(n_wave, host_threads(), [&](int64_t b0, int64_t e0, int) {
            for (int64_t w = b0; w < e0; ++w) {
                std::vector<uint32_t>& src = wave_words[(size_t)w];
                if (!src.empty())
                    std::memcpy(all.get() + wave_start[(size_t)w], src.data(), src.size() * sizeof(uint32_t));
                std::vector<uint32_t>().swap(src);
            }
        });
        VRX_HIP(hipMemcpyAsync(t.ent.p, all.get(), (size_t)total * sizeof(uint32_t),
                               hipMemcpyHostToDevice, s));
        VRX_HIP(hipStreamSynchronize(s));
    }
    VRX_HIP(t.wave_start.upload(wave_start.data(), wave_start.size(), s));
    VRX_HIP(t.bnd.upload(bnd.data(), bnd.size(), s));
    VRX_HIP(t.rowmap.upload(rowmap.data(), rowmap.size(), s));
    if (t.split) {
        VRX_HIP(t.vptr.upload(vptr.data(), vptr.size(), s));
        VRX_HIP(t.split_rows.upload(split_rows.data(), split_rows.size(), s));
    }
    VRX_HIP(hipStreamSynchronize(s));
    const int rc = plan_items(t, bnd.data(), n_wave, NR * PH, n_cu, rowmap.data(), mode, s);
    if (rc) return rc;
    t.ready = true;
    return VRX_OK;
}

// Rows per wave of the cell pass.  The (tile, slab) visits of a pass are dealt to one workgroup
// per CU (plan_items); a visit is not divisible, so with one or two slabs (few variants: clone
// mode) the busiest CU gets ceil(visits / CUs) of them, and the shorter tile wins when that
// rounds up less (200 k cells, one slab: 261 tiles of 768 rows = 2 visits on the busiest CU,
// 391 tiles of 512 rows = 2 shorter ones).  (build_tiled then spreads the rows over whole
// rounds of CUs, VIREO_LDS_FILL_CUS.)
static int pick_rw_cell(int64_t n_var, int64_t n_cell, int n_cu, int cell_form) {
    const int tall = cell_form == 1 ? VRX_LDS_RW_CELL : VRX_LDS_RW_CELL_PAIR;
    const int slab = VRX_LDS_SLAB_BYTES / 256;
    const int n_slab_c = (int)((n_var + slab - 1) / slab);
    auto cost = [&](int rw) {  // visits of the busiest CU x rows per wave
        const int64_t tiles = (n_cell + VRX_LDS_WAVES * (int64_t)rw - 1) / (VRX_LDS_WAVES * (int64_t)rw);
        const int64_t cus = std::max(1, n_cu);
        return (double)((tiles * n_slab_c + cus - 1) / cus) * rw;
    };
    const int forced = env_int("VIREO_LDS_RW_CELL", 0);
    return forced == VRX_LDS_RW_CELL_SHORT ||
                   (forced == 0 && n_slab_c <= 2 && cost(VRX_LDS_RW_CELL_SHORT) < cost(tall))
               ? VRX_LDS_RW_CELL_SHORT
               : tall;
}

// Which stream words the LDS-resident passes use.  Single-valued AD / BD words (cell form 1,
// variant form 2) cost ~1.56x less per word than (ad, dp) pair words (c3: 0.39 vs 0.51 ms at
// 1.2 words per entry) and hold any count, but a count with more than three significant bits
// takes several words (45 = 40 + 5): deep data -- clone mode, DP ~ Poisson(50), 2.65 words per
// entry -- is better served by one pair word per entry as long as the counts fit its 11 bits.
// The words per entry are estimated from every 61st entry.  VIREO_CELL_FORM / VIREO_VAR_FORM
// force a form.
struct StreamForms {
    int cell, var;   // cell pass: 1 AD/BD, 0 pairs; variant pass: 2 AD/BD phases, 0 pairs
    bool auto_pair;  // pairs chosen by the estimate: fall back to AD/BD when a count is >= 2048
};
static StreamForms pick_forms(int64_t nnz, const int32_t* ad, const int32_t* dp) {
    auto chunks = [](int64_t v) {
        int n = 0;
        while (v > 0) {
            const int len = 64 - __builtin_clzll((uint64_t)v), sh = std::max(0, len - 3);
            v -= (v >> sh) << sh;
            ++n;
        }
        return n;
    };
    int64_t words = 0, seen = 0;
    for (int64_t e = 0; e < nnz; e += 61) {
        const int64_t a = ad[e], d = dp[e];
        if (a < 0 || d < a) continue;  // (rejected by the validation that follows)
        words += chunks(a) + chunks(d - a);
        ++seen;
    }
    const double wpe = seen ? (double)words / (double)seen : 1.0;
    const bool pairs = wpe > (double)env_int("VIREO_PAIR_WORDS_X100", 156) / 100.0;
    StreamForms f;
    f.cell = env_int("VIREO_CELL_FORM", pairs ? 0 : 1);
    f.var = env_int("VIREO_VAR_FORM", pairs ? 0 : 3);
    f.auto_pair = pairs && !getenv("VIREO_CELL_FORM") && !getenv("VIREO_VAR_FORM");
    return f;
}

// Large problems: everything from the merged CSC arrays onwards happens on the device
// (vrx_build.h).  *built = false means "not applicable here" (small problem, counts the pair
// words cannot hold, a stream the padding guard rejects): the caller then runs the host builder.
static int device_build(vrx_problem* p, const int64_t* colptr, const int32_t* rowidx,
                        const int32_t* ad, const int32_t* dp, int rw_cell, int slab_cell, int slab_var,
                        StreamForms forms, bool guard, bool* built) {
    *built = false;
    const int64_t nnz = p->nnz, n_var = p->n_var, n_cell = p->n_cell;
    hipStream_t s = p->stream;
    // Entry ids travel through the transposition as 32-bit sort values and the tiled builder keeps
    // segment bounds as 32-bit entry offsets: up to 2^32 - 1 entries (r6: was 2^31 - 1, and the slow
    // host builder took over without a word; 2.2e9 entries are built here in seconds,
    // profiles/r06_big_probe_*.txt).  Beyond that the host builder below is the path.
    if (nnz <= 0 || nnz >= (int64_t)UINT32_MAX - 4096) return VRX_OK;
    for (int64_t c = 0; c < n_cell; ++c)
        if (colptr[c + 1] < colptr[c]) {
            vrx_set_error("vrx_problem_create: colptr not monotone at column %lld", (long long)c);
            return VRX_ERR_ARG;
        }
    const bool timing = env_int("VIREO_BUILD_TIMING", 0) != 0;
    auto wall = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double lap_t = wall();
    auto lap = [&](const char* what) {  // (VIREO_BUILD_TIMING=1: where a device build spends its wall clock)
        if (!timing) return;
        (void)hipStreamSynchronize(s);
        const double t1 = wall();
        fprintf(stderr, "[vrx build] device_build: %-36s %.3f s\n", what, t1 - lap_t);
        lap_t = t1;
    };
    struct JoinGuard {  // (an early return must not leave a helper thread running on dying buffers)
        std::thread& t;
        ~JoinGuard() {
            if (t.joinable()) t.join();
        }
    };
    // Balanced slabs: the cell orientation's layout and greedy need nothing but the caller's arrays -- they
    // run on a helper thread (and its own pool of host threads) from here on, beside the upload, the
    // validation and the transposition on the device.  Used by build_tiled if the layout it asks for is
    // this one (a count >= 2048 can still change the stream form below; then it is simply recomputed).
    TileLayout cell_layout;
    std::thread early;
    // (only where the greedy runs on host threads, VIREO_BALANCE_GREEDY=host: by default it runs on the device)
    const char* greedy_mode = getenv("VIREO_BALANCE_GREEDY");
    const bool host_greedy = greedy_mode && !strcmp(greedy_mode, "host");
    if (p->want_balance && host_greedy && forms.cell == 1 && env_int("VIREO_BALANCE_EARLY", 1) != 0) {
        const int rw0 = rw_cell, form0 = forms.cell;
        early = std::thread([&, rw0, form0] {
          try {
            const auto t0 = std::chrono::steady_clock::now();
            TileLayout& L = cell_layout;
            if (tile_layout(L, colptr, n_cell, n_var, nnz, rw0, slab_cell, form0, p->n_cu) != VRX_OK) {
                L.ptr = nullptr;  // (matches nothing: build_tiled computes -- and reports -- by itself)
                return;
            }
            if (!host_balance_applies(L)) return;
            std::vector<uint8_t> words;
            if (!host_words(HostCounts{rowidx, ad, dp}, nnz, n_var, words)) return;  // (the validation will say why)
            greedy_tiles(L, false, colptr, rowidx, words.data(), n_cell, L.posmap, L.perm, L.tile_of_row);
            L.greedy_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            L.greedy_done = true;
          } catch (const std::exception&) {  // (out of memory: build_tiled computes -- or fails -- by itself)
            cell_layout.ptr = nullptr;
            cell_layout.greedy_done = false;
          }
        });
    }
    JoinGuard early_guard{early};
    DevBuf<int64_t> d_colptr, d_rptr;
    DevBuf<int32_t> d_row, d_ad, d_dp, d_ecol, d_nvars, d_status, d_ridx;
    DevBuf<int2> d_cval, d_rval;
    DevBuf<uint32_t> keys_in, keys_out, vals_in, vals_out;
    VRX_HIP(d_colptr.upload(colptr, (size_t)n_cell + 1, s));
    VRX_HIP(d_row.upload(rowidx, (size_t)nnz, s));
    VRX_HIP(d_ad.upload(ad, (size_t)nnz, s));
    VRX_HIP(d_dp.upload(dp, (size_t)nnz, s));
    VRX_HIP(d_ecol.alloc((size_t)nnz));
    VRX_HIP(d_nvars.alloc((size_t)n_cell));
    VRX_HIP(hipMemsetAsync(d_nvars.p, 0, (size_t)n_cell * sizeof(int32_t), s));
    const int32_t st0[3] = {INT32_MAX, 0, 0};
    VRX_HIP(d_status.upload(st0, 3, s));
    const unsigned nb = (unsigned)((nnz + VRX_BLOCK - 1) / VRX_BLOCK);
    vrx_build_validate<<<nb, VRX_BLOCK, 0, s>>>(nnz, n_var, n_cell, d_colptr.p, d_row.p, d_ad.p, d_dp.p,
                                                d_ecol.p, d_nvars.p, d_status.p);
    VRX_HIP(hipGetLastError());
    int32_t st[3];
    p->n_vars.assign((size_t)n_cell, 0);
    VRX_HIP(hipMemcpyAsync(st, d_status.p, sizeof st, hipMemcpyDeviceToHost, s));
    VRX_HIP(hipMemcpyAsync(p->n_vars.data(), d_nvars.p, (size_t)n_cell * sizeof(int32_t),
                           hipMemcpyDeviceToHost, s));
    VRX_HIP(hipStreamSynchronize(s));
    if (st[0] != INT32_MAX) {
        if (st[1] == 3)
            vrx_set_error("vrx_problem_create: negative count in column %lld", (long long)st[0]);
        else
            vrx_set_error("vrx_problem_create: row indices of column %lld not strictly "
                          "increasing / out of range", (long long)st[0]);
        return VRX_ERR_ARG;
    }
    const int32_t max_count = st[2];
    if (max_count >= 2048 && forms.auto_pair) {
        // the estimate chose pair words but a count does not fit them: AD/BD words after all, and
        // the cell tile height that belongs to THAT form (the caller picked rw_cell for pairs)
        forms = StreamForms{1, 3, false};
        rw_cell = pick_rw_cell(n_var, n_cell, p->n_cu, forms.cell);
    }
    const int var_form = forms.var, cell_form = forms.cell;
    // (pair words hold 11-bit counts; a forced pair form leaves such data to the host builder)
    if ((var_form < 2 || cell_form != 1) && max_count >= 2048) return VRX_OK;
    lap("upload + validate");
    // ---- transposition: stable sort of (variant, entry) ---------------------------------------
    VRX_HIP(keys_in.alloc((size_t)nnz));
    VRX_HIP(keys_out.alloc((size_t)nnz));
    VRX_HIP(vals_in.alloc((size_t)nnz));
    VRX_HIP(vals_out.alloc((size_t)nnz));
    vrx_build_iota_keys<<<nb, VRX_BLOCK, 0, s>>>(nnz, d_row.p, keys_in.p, vals_in.p);
    VRX_HIP(hipGetLastError());
    int bits = 1;
    while (((int64_t)1 << bits) < n_var) ++bits;
    {
        size_t tmp_bytes = 0;
        VRX_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in.p, keys_out.p, vals_in.p,
                                                   vals_out.p, (size_t)nnz, 0, bits, s));
        DevBuf<char> tmp;
        VRX_HIP(tmp.alloc(tmp_bytes));
        VRX_HIP(hipcub::DeviceRadixSort::SortPairs(tmp.p, tmp_bytes, keys_in.p, keys_out.p, vals_in.p,
                                                   vals_out.p, (size_t)nnz, 0, bits, s));
        VRX_HIP(hipStreamSynchronize(s));
    }
    keys_in.release();
    vals_in.release();
    VRX_HIP(d_rptr.alloc((size_t)n_var + 1));
    vrx_build_rptr<<<(unsigned)((n_var + 1 + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
        n_var, nnz, keys_out.p, d_rptr.p);
    VRX_HIP(hipGetLastError());
    VRX_HIP(d_ridx.alloc((size_t)nnz));
    VRX_HIP(d_rval.alloc((size_t)nnz));
    VRX_HIP(d_cval.alloc((size_t)nnz));
    vrx_build_gather<<<nb, VRX_BLOCK, 0, s>>>(nnz, vals_out.p, d_ecol.p, d_ad.p, d_dp.p, d_ridx.p,
                                              d_rval.p, d_cval.p);
    VRX_HIP(hipGetLastError());
    std::vector<int64_t> rptr((size_t)n_var + 1);
    VRX_HIP(hipMemcpyAsync(rptr.data(), d_rptr.p, rptr.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    VRX_HIP(hipStreamSynchronize(s));
    keys_out.release();
    vals_out.release();
    d_ecol.release();
    d_ad.release();
    d_dp.release();
    // ---- the packed entry arrays of both orientations (no segment tables: the LDS-resident
    //      passes below serve every K) ------------------------------------------------------------
    auto pick_fmt = [&](int64_t n_contract) {
        int f = VRX_FMT_WIDE;
        if (max_count < (1 << 16)) f = VRX_FMT_P64;
        if (max_count < 64 && n_contract <= (1 << 20)) f = VRX_FMT_P32;
        const int forced = env_int("VIREO_ENTRY_FMT", -1);
        return forced >= f && forced <= VRX_FMT_WIDE ? forced : f;
    };
    auto set_orient = [&](Orient& o, int64_t n_rows, int64_t n_contract, const int32_t* d_idx,
                          const int2* d_val) {
        o.n_rows = n_rows;
        o.n_contract = n_contract;
        o.nnz = nnz;
        o.fmt = pick_fmt(n_contract);
        o.n_tiles = 1;
        o.n_seg = 0;
        o.n_multi = o.n_slots = 0;
        VRX_HIP(o.ent.alloc((size_t)nnz * (o.fmt + 1)));
        vrx_build_pack<<<nb, VRX_BLOCK, 0, s>>>(nnz, o.fmt, d_idx, d_val, o.ent.p);
        VRX_HIP(hipGetLastError());
        return VRX_OK;
    };
    int rc;
    if ((rc = set_orient(p->by_cell, n_cell, n_var, d_row.p, d_cval.p))) return rc;
    if ((rc = set_orient(p->by_var, n_var, n_cell, d_ridx.p, d_rval.p))) return rc;
    const DevRows cell_rows{d_colptr.p, d_row.p, d_cval.p}, var_rows{d_rptr.p, d_ridx.p, d_rval.p};
    p->by_cell.tiled.want_balance = p->by_var.tiled.want_balance = p->want_balance;
    lap("transposition");
    // the variant pass on virtual rows (vrx_build.h): derived FIRST, so that -- balanced slabs -- their
    // indices and word counts can travel to the host on a second stream while the host balances the cell
    // orientation (the greedy of vrx_balance_tile reads both orientations' entries on the host)
    DevBuf<int64_t> d_cnt, d_vptr2;
    DevBuf<int32_t> d_vidx;
    DevBuf<int2> d_vval;
    std::vector<int64_t> vptr2;
    int64_t vnnz = 0;
    std::vector<int32_t> hv_idx;
    std::vector<uint8_t> hv_words;
    std::thread helper;
    hipError_t helper_err = hipSuccess;
    if (var_form == 3) {
        VRX_HIP(d_cnt.alloc((size_t)(2 * n_var)));
        const unsigned nbv = (unsigned)((n_var + VRX_BLOCK - 1) / VRX_BLOCK);
        vrx_virt_count<<<nbv, VRX_BLOCK, 0, s>>>(n_var, d_rptr.p, d_ridx.p, d_rval.p, d_cnt.p);
        VRX_HIP(hipGetLastError());
        vptr2.assign((size_t)(2 * n_var + 1), 0);
        VRX_HIP(hipMemcpyAsync(vptr2.data() + 1, d_cnt.p, (size_t)(2 * n_var) * sizeof(int64_t),
                               hipMemcpyDeviceToHost, s));
        VRX_HIP(hipStreamSynchronize(s));
        for (int64_t r = 0; r < 2 * n_var; ++r) vptr2[(size_t)r + 1] += vptr2[(size_t)r];
        vnnz = vptr2[(size_t)(2 * n_var)];
        VRX_HIP(d_vptr2.upload(vptr2.data(), vptr2.size(), s));
        VRX_HIP(d_vidx.alloc((size_t)std::max<int64_t>(vnnz, 1)));
        VRX_HIP(d_vval.alloc((size_t)std::max<int64_t>(vnnz, 1)));
        vrx_virt_fill<<<nbv, VRX_BLOCK, 0, s>>>(n_var, d_rptr.p, d_ridx.p, d_rval.p, d_vptr2.p, d_vidx.p,
                                                d_vval.p);
        VRX_HIP(hipGetLastError());
        if (p->want_balance && host_greedy && vnnz > 0) {
            VRX_HIP(hipStreamSynchronize(s));
            hv_idx.resize((size_t)vnnz);
            hv_words.resize((size_t)vnnz);
            const int dev_id = p->device;
            helper = std::thread([&, dev_id] {
                hipStream_t s2 = nullptr;
                DevBuf<uint8_t> d_words;
                auto ok = [&](hipError_t e) {
                    if (e != hipSuccess && helper_err == hipSuccess) helper_err = e;
                    return e == hipSuccess;
                };
                if (ok(hipSetDevice(dev_id)) && ok(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)) &&
                    ok(d_words.alloc((size_t)vnnz))) {
                    vrx_build_words<<<(unsigned)((vnnz + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s2>>>(vnnz, d_vval.p, d_words.p);
                    ok(hipGetLastError());
                    ok(hipMemcpyAsync(hv_idx.data(), d_vidx.p, (size_t)vnnz * sizeof(int32_t), hipMemcpyDeviceToHost, s2));
                    ok(hipMemcpyAsync(hv_words.data(), d_words.p, (size_t)vnnz, hipMemcpyDeviceToHost, s2));
                    ok(hipStreamSynchronize(s2));
                }
                if (s2) (void)hipStreamDestroy(s2);
            });
        }
    }
    JoinGuard join_guard{helper};
    lap("virtual rows");
    // The two orientations' streams are independent from here on: the variant orientation is built by a second
    // host thread on a stream of its own while this one builds the cell orientation (their host halves --
    // layout, work list -- and their device halves -- the greedy of balanced slabs is a latency-bound kernel on
    // half the CUs -- overlap).  Not for problems whose two sets of transient buffers would not fit together.
    auto build_var = [&](hipStream_t sv) -> int {
        int rcv;
        if (var_form == 3) {
            if (helper.joinable()) helper.join();
            if (helper_err != hipSuccess) {
                vrx_set_error("balanced slabs: download of the variant rows failed: %s", hipGetErrorString(helper_err));
                return VRX_ERR_HIP;
            }
            const HostWords virt_host{hv_idx.data(), hv_words.data()};
            const DevRows virt_rows{d_vptr2.p, d_vidx.p, d_vval.p};
            rcv = build_tiled(p->by_var, vptr2.data(), nullptr, nullptr, VRX_LDS_RW_CELL, VRX_LDS_SLAB_BYTES / 256,
                              guard, 1, 0, sv, p->n_cu, &virt_rows, 2 * n_var, (n_cell + 1) / 2, vnnz, nullptr,
                              hv_idx.empty() ? nullptr : &virt_host);
        } else {
            rcv = build_tiled(p->by_var, rptr.data(), nullptr, nullptr, VRX_LDS_RW_VARIANT, slab_var, guard,
                              var_form == 2 ? 2 : 0, 0, sv, p->n_cu, &var_rows);
        }
        if (rcv) return rcv;
        return hipStreamSynchronize(sv) == hipSuccess ? VRX_OK : VRX_ERR_HIP;
    };
    VRX_HIP(hipStreamSynchronize(s));  // (the rows both builds read are complete)
    const bool concurrent = env_int("VIREO_BUILD_CONCURRENT", 1) != 0 &&
                            nnz < (int64_t)env_int("VIREO_BUILD_CONCURRENT_MAX_MNNZ", 1000) * 1000000;
    std::thread var_thread;
    int rc_var = VRX_OK;
    std::string err_var;
    hipStream_t s2 = nullptr;
    struct StreamGuard {  // (destroyed after the thread that uses it has been joined: declared before its guard)
        hipStream_t& st;
        ~StreamGuard() {
            if (st) (void)hipStreamDestroy(st);
        }
    } s2_guard{s2};
    if (concurrent) {
        VRX_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        const int dev_id = p->device;
        var_thread = std::thread([&, dev_id] {
            if (hipSetDevice(dev_id) != hipSuccess) {
                rc_var = VRX_ERR_HIP;
                err_var = "hipSetDevice failed on the variant orientation's build thread";
                return;
            }
            try {
                rc_var = build_var(s2);
                if (rc_var) err_var = vrx_last_error();
            } catch (const std::exception& e) {  // (a throw must not leave a thread)
                rc_var = VRX_ERR_NOMEM;
                err_var = std::string("variant orientation's build: ") + e.what();
            }
        });
    }
    JoinGuard var_guard{var_thread};
    const HostCounts cell_host{rowidx, ad, dp};
    if (early.joinable()) early.join();
    rc = build_tiled(p->by_cell, colptr, nullptr, nullptr, rw_cell, slab_cell, guard, cell_form, 1, s,
                     p->n_cu, &cell_rows, -1, -1, -1, &cell_host, nullptr, &cell_layout);
    lap("cell stream");
    if (concurrent) {
        var_thread.join();
        lap("wait for the variant stream (built beside it)");
    }
    if (rc) return rc;
    if (!concurrent) {
        rc_var = build_var(s);
        lap("variant stream");
    } else if (rc_var) {
        vrx_set_error("%s", err_var.c_str());
    }
    if (rc_var) return rc_var;
    VRX_HIP(hipStreamSynchronize(s));
    if (!p->by_cell.tiled.ready || !p->by_var.tiled.ready) {  // rejected by the padding guard
        for (Orient* o : {&p->by_cell, &p->by_var}) {
            o->ent.release();
            o->tiled.ent.release();
            o->tiled.wave_start.release();
            o->tiled.bnd.release();
            o->tiled.rowmap.release();
            o->tiled.vptr.release();
            o->tiled.split_rows.release();
            o->tiled.items.release();
            o->tiled.wg_first.release();
            o->tiled.npiece.release();
            o->tiled.ready = false;
        }
        return VRX_OK;
    }
    *built = true;
    return VRX_OK;
}

extern "C" int vrx_problem_create(int device, int64_t n_var, int64_t n_cell, int64_t nnz,
                                  const int64_t* colptr, const int32_t* rowidx, const int32_t* ad,
                                  const int32_t* dp, vrx_problem** out) {
    // (VIREO_BALANCE=1 in the environment: balanced slabs for every problem built through this entry)
    return vrx_problem_create2(device, n_var, n_cell, nnz, colptr, rowidx, ad, dp,
                               env_int("VIREO_BALANCE", 0) != 0 ? VRX_PROBLEM_BALANCED : 0, out);
}

static int problem_create2(int device, int64_t n_var, int64_t n_cell, int64_t nnz, const int64_t* colptr,
                           const int32_t* rowidx, const int32_t* ad, const int32_t* dp, int32_t flags,
                           vrx_problem** out);

extern "C" int vrx_problem_create2(int device, int64_t n_var, int64_t n_cell, int64_t nnz,
                                   const int64_t* colptr, const int32_t* rowidx, const int32_t* ad,
                                   const int32_t* dp, int32_t flags, vrx_problem** out) {
    try {  // (the build allocates host arrays of the problem's size: out of memory is a status, not a throw
           //  across the C boundary)
        return problem_create2(device, n_var, n_cell, nnz, colptr, rowidx, ad, dp, flags, out);
    } catch (const std::bad_alloc&) {
        vrx_set_error("vrx_problem_create: out of host memory");
        return VRX_ERR_NOMEM;
    } catch (const std::exception& e) {
        vrx_set_error("vrx_problem_create: %s", e.what());
        return VRX_ERR_UNSUPPORTED;
    }
}

static int problem_create2(int device, int64_t n_var, int64_t n_cell, int64_t nnz, const int64_t* colptr,
                           const int32_t* rowidx, const int32_t* ad, const int32_t* dp, int32_t flags,
                           vrx_problem** out) {
    VRX_REQUIRE(out, "vrx_problem_create: null output");
    *out = nullptr;
    VRX_REQUIRE(n_var > 0 && n_cell > 0 && nnz >= 0, "vrx_problem_create: bad shape");
    VRX_REQUIRE(n_var < INT32_MAX && n_cell < INT32_MAX, "vrx_problem_create: dimension >= 2^31");
    VRX_REQUIRE(colptr && (nnz == 0 || (rowidx && ad && dp)), "vrx_problem_create: null input");
    VRX_REQUIRE(colptr[0] == 0 && colptr[n_cell] == nnz, "vrx_problem_create: colptr/nnz mismatch");
    int ndev = 0;
    vrx_device_count(&ndev);
    if (device < 0 || device >= ndev) {
        vrx_set_error("vrx_problem_create: device %d not available (%d HIP devices visible)",
                      device, ndev);
        return VRX_ERR_HIP;
    }
    VRX_HIP(hipSetDevice(device));
    std::unique_ptr<vrx_problem> p(new vrx_problem());
    p->device = device;
    p->n_var = n_var;
    p->n_cell = n_cell;
    p->nnz = nnz;
    p->want_balance = (flags & VRX_PROBLEM_BALANCED) != 0;
    hipDeviceProp_t prop;
    VRX_HIP(hipGetDeviceProperties(&prop, device));
    p->n_cu = prop.multiProcessorCount;
    VRX_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));

    // Problems that will run on the LDS-resident passes anyway (both thresholds met) are built
    // on the device; VIREO_BUILD=host keeps the host builder (the specification the device
    // build is tested against), VIREO_BUILD=device takes the device path whenever VIREO_LDS
    // allows the streams.
    StreamForms forms = pick_forms(nnz, ad, dp);
    {
        const int lds0 = env_int("VIREO_LDS", -1);
        const char* bm = getenv("VIREO_BUILD");
        const bool force_dev = bm && !strcmp(bm, "device"), force_host = bm && !strcmp(bm, "host");
        const bool big = nnz >= (int64_t)env_int("VIREO_LDS_MIN_NNZ", 4000000) &&
                         nnz >= (int64_t)env_int("VIREO_LDS_MIN_NNZ_VAR", 32000000);
        if (!force_host && lds0 != 0 && (force_dev || big)) {
            bool built = false;
            int rc = device_build(p.get(), colptr, rowidx, ad, dp, pick_rw_cell(n_var, n_cell, p->n_cu, forms.cell),
                                  std::min(VRX_LDS_SLAB_BYTES / 256, std::max(16, env_int("VIREO_LDS_SLAB_CELL", VRX_LDS_SLAB_BYTES / 256))),
                                  std::min(VRX_LDS_SLAB_BYTES / 128, std::max(16, env_int("VIREO_LDS_SLAB_VAR", VRX_LDS_SLAB_BYTES / 128))), forms,
                                  lds0 != 1, &built);
            if (rc) return rc;
            if (built) {
                p->balance_seconds = p->by_cell.tiled.balance_seconds + p->by_var.tiled.balance_seconds;
                *out = p.release();
                return VRX_OK;
            }
        }
    }

    // validate + interleave (ad, dp); count per-cell and per-variant entries.  Cells are cut
    // into one contiguous chunk per host thread; every thread keeps its own per-variant
    // histogram, which also gives it private write cursors for the transposition below
    // (a parallel counting sort: cells stay increasing inside each variant row).
    RawArray<int2> cval((size_t)nnz);
    std::vector<int64_t> rptr((size_t)n_var + 1, 0);
    p->n_vars.assign((size_t)n_cell, 0);
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(host_threads(), n_cell));
    std::vector<std::vector<int64_t>> hist((size_t)nt, std::vector<int64_t>((size_t)n_var, 0));
    std::vector<int32_t> tmax((size_t)nt, 0);
    std::vector<int64_t> bad_col((size_t)nt, -1), bad_kind((size_t)nt, 0);
    parallel_chunks(n_cell, nt, [&](int64_t c0, int64_t c1, int tid) {
        auto& h = hist[(size_t)tid];
        int32_t mx = 0;
        for (int64_t c = c0; c < c1; ++c) {
            if (colptr[c + 1] < colptr[c]) {
                bad_col[(size_t)tid] = c, bad_kind[(size_t)tid] = 1;
                return;
            }
            int32_t prev = -1, nv = 0;
            for (int64_t e = colptr[c]; e < colptr[c + 1]; ++e) {
                const int32_t r = rowidx[e];
                if (r <= prev || r >= n_var) {
                    bad_col[(size_t)tid] = c, bad_kind[(size_t)tid] = 2;
                    return;
                }
                if (ad[e] < 0 || dp[e] < 0) {
                    bad_col[(size_t)tid] = c, bad_kind[(size_t)tid] = 3;
                    return;
                }
                prev = r;
                cval[(size_t)e] = make_int2(ad[e], dp[e]);
                mx = std::max(mx, std::max(ad[e], dp[e]));
                ++h[(size_t)r];
                nv += dp[e] > 0;
            }
            p->n_vars[(size_t)c] = nv;
        }
        tmax[(size_t)tid] = mx;
    });
    int32_t max_count = 0;
    for (int t = 0; t < nt; ++t) {
        max_count = std::max(max_count, tmax[(size_t)t]);
        if (bad_kind[(size_t)t] == 1) {
            vrx_set_error("vrx_problem_create: colptr not monotone at column %lld",
                          (long long)bad_col[(size_t)t]);
            return VRX_ERR_ARG;
        }
        if (bad_kind[(size_t)t] == 2) {
            vrx_set_error("vrx_problem_create: row indices of column %lld not strictly "
                          "increasing / out of range", (long long)bad_col[(size_t)t]);
            return VRX_ERR_ARG;
        }
        if (bad_kind[(size_t)t] == 3) {
            vrx_set_error("vrx_problem_create: negative count in column %lld",
                          (long long)bad_col[(size_t)t]);
            return VRX_ERR_ARG;
        }
    }
    // rptr = exclusive scan of the per-variant totals; hist[t][r] becomes thread t's first
    // write position inside variant row r
    for (int64_t r = 0; r < n_var; ++r) {
        int64_t at = rptr[(size_t)r];
        for (int t = 0; t < nt; ++t) {
            const int64_t n = hist[(size_t)t][(size_t)r];
            hist[(size_t)t][(size_t)r] = at;
            at += n;
        }
        rptr[(size_t)r + 1] = at;
    }
    RawArray<int32_t> ridx((size_t)nnz);
    RawArray<int2> rval((size_t)nnz);
    VRX_REQUIRE(cval.p && ridx.p && rval.p, "out of host memory");
    parallel_chunks(n_cell, nt, [&](int64_t c0, int64_t c1, int tid) {
        auto& cur = hist[(size_t)tid];
        for (int64_t c = c0; c < c1; ++c)
            for (int64_t e = colptr[c]; e < colptr[c + 1]; ++e) {
                const int64_t q = cur[(size_t)rowidx[e]]++;
                ridx[(size_t)q] = (int32_t)c;
                rval[(size_t)q] = cval[(size_t)e];
            }
    });
    // narrowest entry format that holds the counts and the contracted index
    auto pick_fmt = [&](int64_t n_contract) {
        int f = VRX_FMT_WIDE;
        if (max_count < (1 << 16)) f = VRX_FMT_P64;
        if (max_count < 64 && n_contract <= (1 << 20)) f = VRX_FMT_P32;
        const int forced = env_int("VIREO_ENTRY_FMT", -1);
        return forced >= f && forced <= VRX_FMT_WIDE ? forced : f;  // may only widen
    };
    const int tiles_c = pick_tiles(n_var, 256.0, "VIREO_TILES_CELL");   // W rows: 16 x 16 B
    const int tiles_v = pick_tiles(n_cell, 128.0, "VIREO_TILES_VAR");   // ID rows: 16 x 8 B
    int rc = build_orient(p->by_cell, n_cell, n_var, colptr, rowidx, cval.data(), rptr.data(),
                          tiles_c, pick_fmt(n_var), p->stream);
    if (rc) return rc;
    rc = build_orient(p->by_var, n_var, n_cell, rptr.data(), ridx.data(), rval.data(), colptr,
                      tiles_v, pick_fmt(n_cell), p->stream);
    if (rc) return rc;
    // LDS-resident passes (vrx_spmm_lds) pay off on large problems (two-dimensional tiling,
    // one 160 KiB workgroup per CU); VIREO_LDS=0/1 forces them off/on.  Measured crossovers
    // against the gather kernels (K = 8 and 16): the cell pass wins from ~4 M non-zeros; the
    // variant pass (whose per-range partials also cost the theta kernel a wider read) only
    // ties at 8-16 M and wins clearly at 100 M.
    const int lds = env_int("VIREO_LDS", -1);
    if (max_count >= 2048 && forms.auto_pair) forms = StreamForms{1, 3, false};
    const int cell_form = forms.cell;  // 1: AD/BD stream (any counts)
    if ((max_count < 2048 || cell_form == 1) && lds != 0) {
        // cell pass: slabs of 512 W rows (128 KiB at K = 16); variant pass: 1024 ID rows
        if (lds == 1 || nnz >= (int64_t)env_int("VIREO_LDS_MIN_NNZ", 4000000)) {
            const int rw_cell = pick_rw_cell(n_var, n_cell, p->n_cu, cell_form);
            if (cell_form == 1 || max_count < 2048) {
                rc = build_tiled(p->by_cell, colptr, rowidx, cval.data(), rw_cell,
                                 std::min(VRX_LDS_SLAB_BYTES / 256, std::max(16, env_int("VIREO_LDS_SLAB_CELL", VRX_LDS_SLAB_BYTES / 256))),
                                 lds != 1, cell_form == 1 ? 1 : 0, 1, p->stream, p->n_cu);
                if (rc) return rc;
            }
        }
        const int var_form = forms.var;  // 3: AD / BD virtual rows, 2: AD / BD phases (any counts)
        if (var_form == 3 && (lds == 1 || nnz >= (int64_t)env_int("VIREO_LDS_MIN_NNZ_VAR", 32000000))) {
            std::vector<int64_t> vptr2;
            std::vector<int32_t> vidx;
            std::vector<int2> vval;
            derive_virtual_rows(n_var, rptr.data(), ridx.data(), rval.data(), vptr2, vidx, vval);
            rc = build_tiled(p->by_var, vptr2.data(), vidx.data(), vval.data(), VRX_LDS_RW_CELL,
                             VRX_LDS_SLAB_BYTES / 256, lds != 1, 1, 0, p->stream, p->n_cu, nullptr,
                             2 * n_var, (n_cell + 1) / 2, (int64_t)vidx.size());
            if (rc) return rc;
        } else if ((var_form == 2 || max_count < 2048) &&
            (lds == 1 || nnz >= (int64_t)env_int("VIREO_LDS_MIN_NNZ_VAR", 32000000))) {
            rc = build_tiled(p->by_var, rptr.data(), ridx.data(), rval.data(), VRX_LDS_RW_VARIANT,
                             std::min(VRX_LDS_SLAB_BYTES / 128, std::max(16, env_int("VIREO_LDS_SLAB_VAR", VRX_LDS_SLAB_BYTES / 128))),
                             lds != 1, var_form == 2 ? 2 : 0, 0, p->stream, p->n_cu);
            if (rc) return rc;
        }
    }
    *out = p.release();
    return VRX_OK;
}

extern "C" void vrx_problem_destroy(vrx_problem* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

// The float32 terms are formed on the device in the reference's indexing order (row-major:
// the variant-major orientation's storage order), brought to the host and added there in
// NumPy's float32 pairwise order, so the constant equals the reference's bit for bit (up to a
// term whose float64 value sits within 1e-16 of a float32 rounding boundary: lgamma here,
// log(binom()) there).  Once per problem.
extern "C" int vrx_problem_binom_const(vrx_problem* p, double* sum_out) {
    VRX_REQUIRE(p && sum_out, "vrx_problem_binom_const: null argument");
    if (!p->binom_done) {
        VRX_HIP(hipSetDevice(p->device));
        const Orient& o = p->by_var;
        const int64_t n = o.nnz;
        float total = 0.f;
        if (n > 0) {
            DevBuf<float> terms, sums;
            DevBuf<int32_t> flag;
            VRX_HIP(terms.alloc((size_t)n));
            const unsigned nb = (unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK);
            if (o.fmt == VRX_FMT_P32)
                vrx_binom_terms<VRX_FMT_P32><<<nb, VRX_BLOCK, 0, p->stream>>>(n, o.ent.p, terms.p);
            else if (o.fmt == VRX_FMT_P64)
                vrx_binom_terms<VRX_FMT_P64><<<nb, VRX_BLOCK, 0, p->stream>>>(n, o.ent.p, terms.p);
            else
                vrx_binom_terms<VRX_FMT_WIDE><<<nb, VRX_BLOCK, 0, p->stream>>>(n, o.ent.p, terms.p);
            VRX_HIP(hipGetLastError());
            // NumPy sums 8192 elements at a time: the full buffers are summed on the device in
            // its order, their sums and the last partial buffer are added on the host.  A NaN
            // mark (an entry with dp == 0, which the reference's DP > 0 mask skips) shifts the
            // buffer boundaries: then all terms come back and are compacted first.
            const int64_t n_chunk = n / 8192, tail = n - n_chunk * 8192;
            std::vector<float> hs((size_t)n_chunk), ht((size_t)tail);
            int32_t has_nan = 0;
            VRX_HIP(flag.alloc(1));
            VRX_HIP(hipMemsetAsync(flag.p, 0, sizeof(int32_t), p->stream));
            if (n_chunk > 0) {
                VRX_HIP(sums.alloc((size_t)n_chunk));
                vrx_np_chunk_sums_f32<<<(unsigned)n_chunk, 64, 0, p->stream>>>(n_chunk, terms.p, sums.p, flag.p);
                VRX_HIP(hipGetLastError());
                VRX_HIP(hipMemcpyAsync(hs.data(), sums.p, (size_t)n_chunk * sizeof(float),
                                       hipMemcpyDeviceToHost, p->stream));
            }
            if (tail > 0)
                VRX_HIP(hipMemcpyAsync(ht.data(), terms.p + n_chunk * 8192, (size_t)tail * sizeof(float),
                                       hipMemcpyDeviceToHost, p->stream));
            VRX_HIP(hipMemcpyAsync(&has_nan, flag.p, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
            VRX_HIP(hipStreamSynchronize(p->stream));
            for (float v : ht) has_nan |= v != v;
            if (!has_nan) {
                for (float v : hs) total += v;
                float t = 0.f;
                if (vrx_np_sum_f32(ht.data(), tail, &t)) return VRX_ERR_ARG;
                if (tail > 0) total += t;
            } else {
                std::vector<float> h((size_t)n);
                VRX_HIP(hipMemcpyAsync(h.data(), terms.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost,
                                       p->stream));
                VRX_HIP(hipStreamSynchronize(p->stream));
                int64_t m = 0;
                for (int64_t i = 0; i < n; ++i)
                    if (h[(size_t)i] == h[(size_t)i]) h[(size_t)m++] = h[(size_t)i];  // drop the marks
                if (vrx_np_sum_f32(h.data(), m, &total)) return VRX_ERR_ARG;
            }
        }
        p->binom_sum = (double)total;
        p->binom_done = true;
    }
    *sum_out = p->binom_sum;
    return VRX_OK;
}

// 64-bit FNV-1a of a device array (downloaded): tests compare the streams of the host and the
// device builder with it
template <class T>
static int fnv_of(const DevBuf<T>& b, hipStream_t s, uint64_t* out) {
    std::vector<unsigned char> h(b.n * sizeof(T));
    if (!h.empty()) {
        VRX_HIP(hipMemcpyAsync(h.data(), b.p, h.size(), hipMemcpyDeviceToHost, s));
        VRX_HIP(hipStreamSynchronize(s));
    }
    uint64_t x = 1469598103934665603ull;
    for (unsigned char c : h) x = (x ^ c) * 1099511628211ull;
    *out = x;
    return VRX_OK;
}

extern "C" int vrx_problem_digest(vrx_problem* p, uint64_t* out12) {
    VRX_REQUIRE(p && out12, "vrx_problem_digest: null argument");
    VRX_HIP(hipSetDevice(p->device));
    int rc, k = 0;
    for (Orient* o : {&p->by_var, &p->by_cell}) {
        if ((rc = fnv_of(o->ent, p->stream, out12 + k++))) return rc;
        if ((rc = fnv_of(o->tiled.ent, p->stream, out12 + k++))) return rc;
        if ((rc = fnv_of(o->tiled.bnd, p->stream, out12 + k++))) return rc;
        if ((rc = fnv_of(o->tiled.wave_start, p->stream, out12 + k++))) return rc;
        if ((rc = fnv_of(o->tiled.rowmap, p->stream, out12 + k++))) return rc;
        out12[k++] = (uint64_t)o->n_seg;
    }
    return VRX_OK;
}

extern "C" int vrx_problem_build_info(vrx_problem* p, double* info4) {
    VRX_REQUIRE(p && info4, "vrx_problem_build_info: null argument");
    info4[0] = p->by_var.tiled.ready && p->by_var.tiled.balanced ? 1.0 : 0.0;
    info4[1] = p->by_cell.tiled.ready && p->by_cell.tiled.balanced ? 1.0 : 0.0;
    info4[2] = p->balance_seconds;
    info4[3] = p->by_var.tiled.ready && p->by_cell.tiled.ready && p->by_var.n_seg == 0 ? 1.0 : 0.0;  // built on the device
    return VRX_OK;
}

extern "C" int vrx_problem_n_vars(vrx_problem* p, int32_t* out) {
    VRX_REQUIRE(p && out, "vrx_problem_n_vars: null argument");
    std::memcpy(out, p->n_vars.data(), p->n_vars.size() * sizeof(int32_t));
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// model
// ------------------------------------------------------------------------------------
static constexpr int kTraceInit = 1 << 12;  // ELBO slots per restart a model starts with (ensure_trace grows them)
static constexpr int kEventPairs = 1 << 13;

struct vrx_model {
    vrx_problem* p = nullptr;
    vrx_model_cfg cfg{};
    int K = 0, T = 0, KP = 0;
    int R = 1, Kt = 0;  // restarts in the batch; R * K columns of the dense operands
    int64_t N = 0, M = 0, NK = 0, NKt = 0;
    VrxBatch batch() const { return VrxBatch{R, K, Kt}; }
    int64_t th_rows = 1, th_cols = 0;  // shape of beta_mu / beta_sum
    // variational state
    DevBuf<double> ID, GT, mu, sm;
    // derived tables
    DevBuf<double> psi;  // [3][th_rows][T]   (Vireo)
    DevBuf<double> S;    // [N][K] double2  (sum ad*ID, sum dp*ID)
    DevBuf<double> W;    // [N][K] double2  (W1, W2)
    DevBuf<double> LID;  // [M][K]          logLik_ID
    DevBuf<double> PV, PC;  // split-row partial slots
    DevBuf<double> RV, RC;  // per-range partials of the LDS-resident passes
    // priors
    DevBuf<double> logq_id, logq_gt, prior1, prior2, tmp, tmp2;
    int id_mode = 0, gt_mode = 0;
    int64_t prior_rows = 1;
    // reductions
    int nb_theta = 0, nb_nk = 0, nb_cell = 0, nb_throws = 0, n_th_part = 1;
    int nb_gt = 0;               // blocks (= KL_GT partials) of the grid-stride vrx_gt_update
    int n_cell_part = 0;         // cell partials of the kernel that ran last (softmax or fused cell pass)
    bool theta_pending = false;  // stage-1 partials wait for the finalisation inside vrx_gt_update
    DevBuf<double> part_theta, part_gt, part_cell, part_th;
    // clone mode: vrx_bmm_theta WRITES the KL_theta partials, so the ELBO block that rides in it
    // (VrxElboRide) must read the previous iteration's from another buffer: two halves of part_th,
    // th_cur = the half written last (Vireo: always 0)
    size_t th_cap = 0;
    int th_cur = 0;
    DevBuf<double> d_elbo, d_parts;
    int64_t trace_cap = 0;  // ELBO slots per restart in d_elbo
    DevBuf<int32_t> ctl;  // device-side loop control (VRX_CTL_*)
    DevBuf<double> snapID, snapGT, snapTh;  // vrx_model_snapshot
    bool snap_valid = false;
    // vrx_model_stage_raw: the NEXT restart's raw draws, uploaded on a copy stream while this
    // one fits (two buffers; staged[b] / consumed[b] order the copy against its consumer)
    DevBuf<double> stageID[2], stageGT[2];
    hipStream_t copy_stream = nullptr;
    hipEvent_t staged[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    hipEvent_t polled[2] = {nullptr, nullptr};  // vrx_model_fit's pipelined polls
    double* h_pin = nullptr;  // pinned staging for scalar read-backs
    int wform = 0;            // layout of W: 0 (W1, W2) pairs, 1 planar (Wa | Wb) rows (FORM 1)
    bool w_valid = false;     // W matches (GT, psi) on the device
    bool s_pending = false;   // S still sits in RV as per-range partials (sum fused downstream)
    bool l_pending = false;   // logLik_ID still sits in RC as per-range partials
    // launch-bound problems: the ELBO of the iteration enqueued last has not been finalised yet --
    // it rides as an extra block in the next iteration's vrx_theta_partial (VrxElboRide)
    bool elbo_deferred = false;
    VrxStopRule elbo_rule{};
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> ev_kind;
    int ev_used = 0;
    double prof_ms[VRX_KERN_COUNT] = {0, 0, 0};
    int64_t prof_n[VRX_KERN_COUNT] = {0, 0, 0};
    hipEvent_t t0 = nullptr, t1 = nullptr;

    ~vrx_model() {
        if (h_pin) (void)hipHostFree(h_pin);
        for (auto e : ev) (void)hipEventDestroy(e);
        if (t0) (void)hipEventDestroy(t0);
        if (t1) (void)hipEventDestroy(t1);
        for (int b = 0; b < 2; ++b) {
            if (staged[b]) (void)hipEventDestroy(staged[b]);
            if (consumed[b]) (void)hipEventDestroy(consumed[b]);
            if (polled[b]) (void)hipEventDestroy(polled[b]);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
    }
};

// the device-side ELBO trace holds n entries per restart (grown between fits: its content is
// read back before a fit returns and never carried over)
static int ensure_trace(vrx_model* m, int64_t n) {
    if (n <= m->trace_cap) return VRX_OK;
    VRX_REQUIRE(n < ((int64_t)1 << 31), "max_iter too large");
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    int64_t cap = std::max<int64_t>(m->trace_cap, 1);
    while (cap < n) cap *= 2;
    // into a temporary: a failed allocation (huge max_iter) must leave the old trace and its
    // capacity in place -- the model stays usable and the call returns an error code
    DevBuf<double> grown;
    const hipError_t e = grown.alloc((size_t)m->R * (size_t)cap);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (the runtime keeps a failed call's code for the next hipGetLastError: clear it)
        vrx_set_error("ELBO trace of %lld iterations x %d restarts: %s", (long long)cap, m->R, hipGetErrorString(e));
        return VRX_ERR_HIP;
    }
    m->d_elbo.swap(grown);
    m->trace_cap = cap;
    return VRX_OK;
}

static int pick_kp(int K, int cap = 64) {
    int kp = 1;
    while (kp < K && kp < cap) kp <<= 1;
    return kp;
}

struct ProfScope {  // brackets launches of one kernel class with events when profiling
    vrx_model* m;
    int slot = -1;
    ProfScope(vrx_model* m_, int kind) : m(m_) {
        if (m->prof && m->ev_used + 2 <= (int)m->ev.size()) {
            slot = m->ev_used;
            m->ev_used += 2;
            m->ev_kind[slot / 2] = kind;
            (void)hipEventRecord(m->ev[slot], m->p->stream);
        }
    }
    ~ProfScope() {
        if (slot >= 0) (void)hipEventRecord(m->ev[slot + 1], m->p->stream);
    }
};

static int prof_drain(vrx_model* m) {
    if (m->ev_used == 0) return VRX_OK;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    for (int s = 0; s < m->ev_used; s += 2) {
        float ms = 0.f;
        VRX_HIP(hipEventElapsedTime(&ms, m->ev[s], m->ev[s + 1]));
        m->prof_ms[m->ev_kind[s / 2]] += ms;
        m->prof_n[m->ev_kind[s / 2]] += 1;
    }
    m->ev_used = 0;
    return VRX_OK;
}

template <int MODE>
static bool lds_eligible(const Orient& o, int K);

extern "C" int vrx_model_create(vrx_problem* p, const vrx_model_cfg* cfg, vrx_model** out) {
    VRX_REQUIRE(p && cfg && out, "vrx_model_create: null argument");
    *out = nullptr;
    VRX_REQUIRE(cfg->kind == VRX_KIND_VIREO || cfg->kind == VRX_KIND_BMM,
                "vrx_model_create: unknown model kind %d", cfg->kind);
    VRX_REQUIRE(cfg->n_donor >= 1, "vrx_model_create: n_donor must be >= 1");
    if (cfg->kind == VRX_KIND_VIREO && (cfg->n_gt < 1 || cfg->n_gt > VRX_MAXT)) {
        vrx_set_error("vrx_model_create: n_GT=%d unsupported (1..%d)", cfg->n_gt, VRX_MAXT);
        return VRX_ERR_UNSUPPORTED;
    }
    VRX_HIP(hipSetDevice(p->device));
    std::unique_ptr<vrx_model> m(new vrx_model());
    m->p = p;
    m->cfg = *cfg;
    m->K = cfg->n_donor;
    m->T = cfg->kind == VRX_KIND_VIREO ? cfg->n_gt : 1;
    m->KP = pick_kp(m->K);
    m->N = p->n_var;
    m->M = p->n_cell;
    m->NK = m->N * m->K;
    m->R = std::max(1, (int)cfg->n_batch);
    VRX_REQUIRE(m->R <= 16, "vrx_model_create: at most 16 restarts per batch");
    m->Kt = m->R * m->K;
    m->NKt = m->NK * m->R;
    VRX_REQUIRE(m->NK < INT32_MAX * (int64_t)VRX_BLOCK, "vrx_model_create: n_var*n_donor too large");
    if (cfg->kind == VRX_KIND_VIREO) {
        m->th_rows = cfg->ase_mode ? m->N : 1;
        m->th_cols = m->T;
    } else {
        m->th_rows = m->N;
        m->th_cols = m->K;
    }
    hipStream_t s = p->stream;
    const size_t th = (size_t)(m->R * m->th_rows * m->th_cols);
    VRX_HIP(m->ID.alloc((size_t)((m->M + 1) * m->Kt)));  // (+ a row: an odd M's last DOUBLE row, TiledStream::virt)
    VRX_HIP(hipMemsetAsync(m->ID.p + (size_t)(m->M * m->Kt), 0, (size_t)m->Kt * sizeof(double), s));
    VRX_HIP(m->LID.alloc((size_t)(m->M * m->Kt)));
    VRX_HIP(m->mu.alloc(th));  // (th already counts the R restarts)
    VRX_HIP(m->sm.alloc(th));
    VRX_HIP(m->prior1.alloc(th));
    VRX_HIP(m->prior2.alloc(th));
    VRX_HIP(m->S.alloc((size_t)m->NKt * 2));
    VRX_HIP(m->W.alloc((size_t)m->NKt * 2));
    VRX_HIP(m->PV.alloc((size_t)(p->by_var.n_slots * m->Kt * 2)));
    VRX_HIP(m->PC.alloc((size_t)(p->by_cell.n_slots * m->Kt)));
    {
        const TiledStream &tv = p->by_var.tiled, &tc = p->by_cell.tiled;
        if (lds_eligible<0>(p->by_var, m->Kt) && tv.virt)  // planar sums of the virtual rows
            VRX_HIP(m->RV.alloc((size_t)(tv.n_range * tv.n_vrows * m->Kt)));
        else if (lds_eligible<0>(p->by_var, m->Kt) && (tv.n_range > 1 || tv.split))
            VRX_HIP(m->RV.alloc((size_t)(tv.n_range * tv.n_vrows * m->Kt * 2)));
        if (lds_eligible<1>(p->by_cell, m->Kt) && (tc.n_range > 1 || tc.split))
            VRX_HIP(m->RC.alloc((size_t)(tc.n_range * tc.n_vrows * m->Kt)));
    }
    m->wform = lds_eligible<1>(p->by_cell, m->Kt) && p->by_cell.tiled.form == 1 ? 1 : 0;
    // rows without entries are never written by the passes: zero once
    VRX_HIP(hipMemsetAsync(m->S.p, 0, (size_t)m->NKt * 2 * sizeof(double), s));
    VRX_HIP(hipMemsetAsync(m->LID.p, 0, (size_t)(m->M * m->Kt) * sizeof(double), s));
    VRX_HIP(hipMemsetAsync(m->ID.p, 0, (size_t)(m->M * m->Kt) * sizeof(double), s));
    m->nb_nk = (int)((m->NK + VRX_BLOCK - 1) / VRX_BLOCK);
    m->nb_cell = (int)((m->M * m->KP + VRX_BLOCK - 1) / VRX_BLOCK);
    {   // vrx_cell_softmax strides over the cells: at most VIREO_SOFTMAX_BLOCKS_PER_CU blocks per CU
        // (0 = one block per 256 lanes, as before; c5: dense kernels 40.9 -> 38.0 us at 4, 39.0 at 8,
        //  39.2 at 16; c3: 0.108 -> 0.1055 ms; profiles/r05_ab_softmax_grid.txt)
        const int cap = env_int("VIREO_SOFTMAX_BLOCKS_PER_CU", 4);
        if (cap > 0) m->nb_cell = std::min(m->nb_cell, p->n_cu * cap);
    }
    m->nb_throws = (int)((m->N + VRX_BLOCK - 1) / VRX_BLOCK);
    m->nb_theta = std::min(m->nb_nk, p->n_cu * 4);
    m->nb_gt = std::min(m->nb_nk, p->n_cu * 8);
    m->n_cell_part = m->nb_cell;
    VRX_HIP(m->part_cell.alloc((size_t)m->R * std::max<int64_t>(m->nb_cell, p->by_cell.n_seg / VRX_WAVES + 1) * 2));
    VRX_HIP(m->part_gt.alloc((size_t)m->R * m->nb_nk));
    VRX_HIP(hipMemsetAsync(m->part_gt.p, 0, (size_t)m->R * m->nb_nk * sizeof(double), s));
    if (cfg->kind == VRX_KIND_VIREO) {
        VRX_HIP(m->GT.alloc((size_t)m->NKt * m->T));
        VRX_HIP(m->psi.alloc(3 * th));  // [R][3][rows][T]; th counts the R restarts
        VRX_HIP(m->part_theta.alloc((size_t)m->R * m->nb_theta * 2 * VRX_MAXT));
        m->n_th_part = cfg->ase_mode ? m->nb_throws : 1;
    } else {
        m->n_th_part = m->nb_nk;
    }
    // (clone mode with the range sum fused into vrx_bmm_theta runs 16 lanes per element: 16x the blocks)
    const size_t th_cap = (size_t)m->R * (cfg->kind == VRX_KIND_VIREO ? m->n_th_part : (m->NK * 16 + VRX_BLOCK - 1) / VRX_BLOCK + 1);
    m->th_cap = th_cap;
    VRX_HIP(m->part_th.alloc(th_cap * (cfg->kind == VRX_KIND_BMM ? 2 : 1)));
    VRX_HIP(hipMemsetAsync(m->part_th.p, 0, th_cap * (cfg->kind == VRX_KIND_BMM ? 2 : 1) * sizeof(double), s));
    m->trace_cap = kTraceInit;
    VRX_HIP(m->d_elbo.alloc((size_t)m->R * m->trace_cap));
    VRX_HIP(m->ctl.alloc((size_t)m->R * VRX_CTL_WORDS));
    VRX_HIP(hipMemsetAsync(m->ctl.p, 0, (size_t)m->R * VRX_CTL_WORDS * sizeof(int32_t), s));
    VRX_HIP(m->d_parts.alloc((size_t)m->R * 4));
    VRX_HIP(hipHostMalloc(reinterpret_cast<void**>(&m->h_pin), 64 * sizeof(double), hipHostMallocDefault));
    VRX_HIP(hipEventCreate(&m->t0));
    VRX_HIP(hipEventCreate(&m->t1));
    VRX_HIP(hipStreamSynchronize(s));
    *out = m.release();
    return VRX_OK;
}

extern "C" void vrx_model_destroy(vrx_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->p->device);
    (void)hipStreamSynchronize(m->p->stream);
    delete m;
}

static int h2d(vrx_model* m, DevBuf<double>& dst, const double* src, size_t n) {
    if (!src) return VRX_OK;
    VRX_REQUIRE(dst.n >= n, "internal: upload larger than buffer");
    VRX_HIP(hipMemcpyAsync(dst.p, src, n * sizeof(double), hipMemcpyHostToDevice, m->p->stream));
    return VRX_OK;
}

static int d2h(vrx_model* m, double* dst, const DevBuf<double>& src, size_t n) {
    if (!dst) return VRX_OK;
    VRX_HIP(hipMemcpyAsync(dst, src.p, n * sizeof(double), hipMemcpyDeviceToHost, m->p->stream));
    return VRX_OK;
}

extern "C" int vrx_model_set_state(vrx_model* m, const double* ID_prob, const double* GT_prob,
                                   const double* beta_mu, const double* beta_sum) {
    VRX_REQUIRE(m, "vrx_model_set_state: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    int rc;
    if ((rc = h2d(m, m->ID, ID_prob, (size_t)(m->M * m->Kt)))) return rc;
    if (m->cfg.kind == VRX_KIND_VIREO)
        if ((rc = h2d(m, m->GT, GT_prob, (size_t)m->NKt * m->T))) return rc;
    if ((rc = h2d(m, m->mu, beta_mu, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    if ((rc = h2d(m, m->sm, beta_sum, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    m->w_valid = false;
    return VRX_OK;
}

extern "C" int vrx_model_set_state_raw(vrx_model* m, const double* ID_raw, const double* GT_raw,
                                       const double* beta_mu, const double* beta_sum) {
    VRX_REQUIRE(m, "vrx_model_set_state_raw: null model");
    if (m->K > 128 || m->T > 128) {
        vrx_set_error("vrx_model_set_state_raw: more than 128 columns (normalise on the host)");
        return VRX_ERR_UNSUPPORTED;
    }
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t s = m->p->stream;
    int rc;
    if ((rc = h2d(m, m->ID, ID_raw, (size_t)(m->M * m->Kt)))) return rc;
    if (ID_raw) {
        const int64_t rows = m->M * m->R;  // [M][R][K]: one row per (cell, restart)
        vrx_normalize_rows<<<(unsigned)((rows + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            rows, m->K, m->ID.p);
        VRX_HIP(hipGetLastError());
    }
    if (m->cfg.kind == VRX_KIND_VIREO && GT_raw) {
        if ((rc = h2d(m, m->GT, GT_raw, (size_t)m->NKt * m->T))) return rc;
        vrx_normalize_rows<<<(unsigned)((m->NKt + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            m->NKt, m->T, m->GT.p);
        VRX_HIP(hipGetLastError());
    }
    if ((rc = h2d(m, m->mu, beta_mu, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    if ((rc = h2d(m, m->sm, beta_sum, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    VRX_HIP(hipStreamSynchronize(s));
    m->w_valid = false;
    return VRX_OK;
}

// ---- staged uploads (vireo_wrap.py:64-87: the restarts run one after the other) -------------
// The raw constructor draws of restart i + 1 (45 MB at c3) travel to the device WHILE restart i
// fits: vrx_model_stage_reserve (once, from the thread that owns the model) makes two staging
// buffers, a copy stream and the events; vrx_model_stage_raw may then be called from a SECOND host
// thread concurrently with vrx_model_fit on the same model -- it touches nothing but its staging
// buffer and the copy stream; vrx_model_set_state_staged (owner thread) normalises the buffer into
// the model's state on the compute stream.  Results are those of vrx_model_set_state_raw, bitwise.
extern "C" int vrx_model_stage_reserve(vrx_model* m) {
    VRX_REQUIRE(m, "vrx_model_stage_reserve: null model");
    VRX_REQUIRE(m->R == 1 && m->cfg.kind == VRX_KIND_VIREO, "vrx_model_stage_reserve: single Vireo models only");
    if (m->copy_stream) return VRX_OK;
    VRX_HIP(hipSetDevice(m->p->device));
    for (int b = 0; b < 2; ++b) {
        VRX_HIP(m->stageID[b].alloc((size_t)(m->M * m->K)));
        VRX_HIP(m->stageGT[b].alloc((size_t)m->NK * m->T));
        VRX_HIP(hipEventCreateWithFlags(&m->staged[b], hipEventDisableTiming));
        VRX_HIP(hipEventCreateWithFlags(&m->consumed[b], hipEventDisableTiming));
    }
    VRX_HIP(hipStreamCreateWithFlags(&m->copy_stream, hipStreamNonBlocking));
    return VRX_OK;
}

extern "C" int vrx_model_stage_raw(vrx_model* m, int32_t buf, const double* ID_raw, const double* GT_raw) {
    VRX_REQUIRE(m && ID_raw && GT_raw, "vrx_model_stage_raw: null argument");
    VRX_REQUIRE(buf == 0 || buf == 1, "vrx_model_stage_raw: buffer %d", buf);
    VRX_REQUIRE(m->copy_stream, "vrx_model_stage_raw: call vrx_model_stage_reserve first");
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t c = m->copy_stream;
    // (the buffer's previous content has been normalised into the state: consumed[buf] was
    //  recorded behind that; an event never recorded counts as complete)
    VRX_HIP(hipStreamWaitEvent(c, m->consumed[buf], 0));
    VRX_HIP(hipMemcpyAsync(m->stageID[buf].p, ID_raw, m->stageID[buf].n * sizeof(double), hipMemcpyHostToDevice, c));
    VRX_HIP(hipMemcpyAsync(m->stageGT[buf].p, GT_raw, m->stageGT[buf].n * sizeof(double), hipMemcpyHostToDevice, c));
    VRX_HIP(hipEventRecord(m->staged[buf], c));
    VRX_HIP(hipStreamSynchronize(c));  // (the host arrays may be reused by the caller)
    return VRX_OK;
}

extern "C" int vrx_model_set_state_staged(vrx_model* m, int32_t buf, const double* beta_mu,
                                          const double* beta_sum) {
    VRX_REQUIRE(m, "vrx_model_set_state_staged: null model");
    VRX_REQUIRE(buf == 0 || buf == 1, "vrx_model_set_state_staged: buffer %d", buf);
    VRX_REQUIRE(m->copy_stream, "vrx_model_set_state_staged: nothing was staged");
    if (m->K > 128 || m->T > 128) {
        vrx_set_error("vrx_model_set_state_staged: more than 128 columns (normalise on the host)");
        return VRX_ERR_UNSUPPORTED;
    }
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t s = m->p->stream;
    int rc;
    VRX_HIP(hipStreamWaitEvent(s, m->staged[buf], 0));
    VRX_HIP(hipMemcpyAsync(m->ID.p, m->stageID[buf].p, m->stageID[buf].n * sizeof(double), hipMemcpyDeviceToDevice, s));
    VRX_HIP(hipMemcpyAsync(m->GT.p, m->stageGT[buf].p, m->stageGT[buf].n * sizeof(double), hipMemcpyDeviceToDevice, s));
    VRX_HIP(hipEventRecord(m->consumed[buf], s));
    vrx_normalize_rows<<<(unsigned)((m->M + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(m->M, m->K, m->ID.p);
    VRX_HIP(hipGetLastError());
    vrx_normalize_rows<<<(unsigned)((m->NK + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(m->NK, m->T, m->GT.p);
    VRX_HIP(hipGetLastError());
    if ((rc = h2d(m, m->mu, beta_mu, (size_t)(m->th_rows * m->th_cols)))) return rc;
    if ((rc = h2d(m, m->sm, beta_sum, (size_t)(m->th_rows * m->th_cols)))) return rc;
    VRX_HIP(hipStreamSynchronize(s));
    m->w_valid = false;
    return VRX_OK;
}

// device-side copy of the variational state: the best restart so far is kept in HBM, no
// round trip through the host (vireo_wrap keeps `model_all[argmax]`, vireo_wrap.py:90-91)
extern "C" int vrx_model_snapshot(vrx_model* m, int32_t restore) {
    VRX_REQUIRE(m, "vrx_model_snapshot: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t s = m->p->stream;
    const size_t th = (size_t)(m->R * m->th_rows * m->th_cols);
    if (restore) {
        VRX_REQUIRE(m->snap_valid, "vrx_model_snapshot: nothing saved");
    } else if (!m->snapID.p) {
        VRX_HIP(m->snapID.alloc((size_t)(m->M * m->Kt)));
        if (m->cfg.kind == VRX_KIND_VIREO) VRX_HIP(m->snapGT.alloc((size_t)m->NKt * m->T));
        VRX_HIP(m->snapTh.alloc(2 * th));
    }
    auto cp = [&](double* live, double* saved, size_t n) {
        return restore ? hipMemcpyAsync(live, saved, n * sizeof(double), hipMemcpyDeviceToDevice, s)
                       : hipMemcpyAsync(saved, live, n * sizeof(double), hipMemcpyDeviceToDevice, s);
    };
    VRX_HIP(cp(m->ID.p, m->snapID.p, (size_t)(m->M * m->Kt)));
    if (m->cfg.kind == VRX_KIND_VIREO) VRX_HIP(cp(m->GT.p, m->snapGT.p, (size_t)m->NKt * m->T));
    VRX_HIP(cp(m->mu.p, m->snapTh.p, th));
    VRX_HIP(cp(m->sm.p, m->snapTh.p + th, th));
    VRX_HIP(hipStreamSynchronize(s));
    if (restore)
        m->w_valid = false;
    else
        m->snap_valid = true;
    return VRX_OK;
}

// ---- restart batches -------------------------------------------------------------------
// rows x cols block between two row-major arrays with their own row strides and column offsets
__global__ __launch_bounds__(VRX_BLOCK) void vrx_copy_block(int64_t rows, int cols,
                                                            const double* __restrict__ src,
                                                            int64_t src_ld, int64_t src_off,
                                                            double* __restrict__ dst, int64_t dst_ld,
                                                            int64_t dst_off) {
    const int64_t i = (int64_t)blockIdx.x * VRX_BLOCK + threadIdx.x;
    if (i >= rows * cols) return;
    const int64_t r = i / cols, c = i - r * cols;
    dst[r * dst_ld + dst_off + c] = src[r * src_ld + src_off + c];
}

static int copy_block(hipStream_t s, int64_t rows, int64_t cols, const double* src, int64_t src_ld,
                      int64_t src_off, double* dst, int64_t dst_ld, int64_t dst_off) {
    const int64_t n = rows * cols;
    if (n == 0) return VRX_OK;
    vrx_copy_block<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
        rows, (int)cols, src, src_ld, src_off, dst, dst_ld, dst_off);
    VRX_HIP(hipGetLastError());
    return VRX_OK;
}

// Slot r of a batch model <- one restart's state, given exactly as vrx_model_set_state[_raw]
// takes it for a single model (ID [M][K], GT [N][K][T], beta [rows][cols]).  raw: the draws are
// row-normalised on the device first, in NumPy's order (vireo_model.py:99,104).
extern "C" int vrx_model_set_restart(vrx_model* m, int32_t r, const double* ID, const double* GT,
                                     const double* beta_mu, const double* beta_sum, int32_t raw) {
    VRX_REQUIRE(m, "vrx_model_set_restart: null model");
    VRX_REQUIRE(r >= 0 && r < m->R, "vrx_model_set_restart: slot %d outside the batch of %d", r, m->R);
    if (raw && (m->K > 128 || m->T > 128)) {
        vrx_set_error("vrx_model_set_restart: more than 128 columns (normalise on the host)");
        return VRX_ERR_UNSUPPORTED;
    }
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t s = m->p->stream;
    int rc;
    const size_t n_id = (size_t)(m->M * m->K), n_gt = (size_t)m->NK * m->T;
    const size_t th = (size_t)(m->th_rows * m->th_cols);
    if (ID) {
        if (m->tmp.n != n_id) VRX_HIP(m->tmp.alloc(n_id));
        VRX_HIP(hipMemcpyAsync(m->tmp.p, ID, n_id * sizeof(double), hipMemcpyHostToDevice, s));
        if (raw)
            vrx_normalize_rows<<<(unsigned)((m->M + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                m->M, m->K, m->tmp.p);
        if ((rc = copy_block(s, m->M, m->K, m->tmp.p, m->K, 0, m->ID.p, m->Kt, (int64_t)r * m->K)))
            return rc;
    }
    if (GT && m->cfg.kind == VRX_KIND_VIREO) {
        if (m->tmp2.n != n_gt) VRX_HIP(m->tmp2.alloc(n_gt));
        VRX_HIP(hipMemcpyAsync(m->tmp2.p, GT, n_gt * sizeof(double), hipMemcpyHostToDevice, s));
        if (raw)
            vrx_normalize_rows<<<(unsigned)((m->NK + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                m->NK, m->T, m->tmp2.p);
        const int64_t kt = (int64_t)m->K * m->T;
        if ((rc = copy_block(s, m->N, kt, m->tmp2.p, kt, 0, m->GT.p, (int64_t)m->Kt * m->T, r * kt)))
            return rc;
    }
    if (beta_mu)
        VRX_HIP(hipMemcpyAsync(m->mu.p + r * th, beta_mu, th * sizeof(double), hipMemcpyHostToDevice, s));
    if (beta_sum)
        VRX_HIP(hipMemcpyAsync(m->sm.p + r * th, beta_sum, th * sizeof(double), hipMemcpyHostToDevice, s));
    VRX_HIP(hipStreamSynchronize(s));  // (the host arrays may be reused by the caller)
    m->w_valid = false;
    return VRX_OK;
}

// dst (a single model of the same problem and shape) <- the state of slot r of `src`, on the
// device: the winner of a batch moves on without a host round trip (vireo_wrap.py:90-91)
extern "C" int vrx_model_copy_restart(vrx_model* dst, vrx_model* src, int32_t r) {
    VRX_REQUIRE(dst && src, "vrx_model_copy_restart: null model");
    VRX_REQUIRE(r >= 0 && r < src->R, "vrx_model_copy_restart: slot %d outside the batch of %d", r, src->R);
    VRX_REQUIRE(dst->R == 1 && dst->p == src->p && dst->K == src->K && dst->T == src->T &&
                    dst->cfg.kind == src->cfg.kind && dst->th_rows == src->th_rows,
                "vrx_model_copy_restart: models differ in problem or shape");
    VRX_HIP(hipSetDevice(src->p->device));
    hipStream_t s = src->p->stream;
    int rc;
    if ((rc = copy_block(s, src->M, src->K, src->ID.p, src->Kt, (int64_t)r * src->K, dst->ID.p, src->K, 0)))
        return rc;
    if (src->cfg.kind == VRX_KIND_VIREO) {
        const int64_t kt = (int64_t)src->K * src->T;
        if ((rc = copy_block(s, src->N, kt, src->GT.p, (int64_t)src->Kt * src->T, r * kt, dst->GT.p, kt, 0)))
            return rc;
    }
    const size_t th = (size_t)(src->th_rows * src->th_cols);
    VRX_HIP(hipMemcpyAsync(dst->mu.p, src->mu.p + r * th, th * sizeof(double), hipMemcpyDeviceToDevice, s));
    VRX_HIP(hipMemcpyAsync(dst->sm.p, src->sm.p + r * th, th * sizeof(double), hipMemcpyDeviceToDevice, s));
    VRX_HIP(hipStreamSynchronize(s));
    dst->w_valid = false;
    return VRX_OK;
}

extern "C" int vrx_model_get_state(vrx_model* m, double* ID_prob, double* GT_prob, double* beta_mu,
                                   double* beta_sum) {
    VRX_REQUIRE(m, "vrx_model_get_state: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    int rc;
    if ((rc = d2h(m, ID_prob, m->ID, (size_t)(m->M * m->Kt)))) return rc;
    if (m->cfg.kind == VRX_KIND_VIREO)
        if ((rc = d2h(m, GT_prob, m->GT, (size_t)m->NKt * m->T))) return rc;
    if ((rc = d2h(m, beta_mu, m->mu, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    if ((rc = d2h(m, beta_sum, m->sm, (size_t)(m->R * m->th_rows * m->th_cols)))) return rc;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    return VRX_OK;
}

// (internal: vrx_common.h) the state arrays vrx_comm_bcast_model sends / receives in place
int vrx_model_state_buffers(vrx_model* m, bool will_write, VrxModelBuffers* out) {
    VRX_REQUIRE(m && out, "vrx_model_state_buffers: null argument");
    VRX_HIP(hipSetDevice(m->p->device));
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    const size_t th = (size_t)(m->R * m->th_rows * m->th_cols);
    out->p[0] = m->ID.p;
    out->n[0] = (size_t)(m->M * m->Kt);
    out->p[1] = m->cfg.kind == VRX_KIND_VIREO ? m->GT.p : nullptr;
    out->n[1] = m->cfg.kind == VRX_KIND_VIREO ? (size_t)m->NKt * m->T : 0;
    out->p[2] = m->mu.p;
    out->n[2] = th;
    out->p[3] = m->sm.p;
    out->n[3] = th;
    out->device = m->p->device;
    if (will_write) m->w_valid = false;
    return VRX_OK;
}

extern "C" int vrx_model_get_loglik(vrx_model* m, double* out) {
    VRX_REQUIRE(m && out, "vrx_model_get_loglik: null argument");
    VRX_HIP(hipSetDevice(m->p->device));
    int rc = d2h(m, out, m->LID, (size_t)(m->M * m->Kt));
    if (rc) return rc;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    return VRX_OK;
}

extern "C" int vrx_model_set_loglik(vrx_model* m, const double* in) {
    VRX_REQUIRE(m && in, "vrx_model_set_loglik: null argument");
    VRX_HIP(hipSetDevice(m->p->device));
    int rc = h2d(m, m->LID, in, (size_t)(m->M * m->Kt));
    if (rc) return rc;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    return VRX_OK;
}

extern "C" int vrx_model_get_elbo_parts(vrx_model* m, double* parts4) {
    VRX_REQUIRE(m && parts4, "vrx_model_get_elbo_parts: null argument");
    VRX_HIP(hipSetDevice(m->p->device));
    VRX_HIP(hipMemcpyAsync(parts4, m->d_parts.p, (size_t)m->R * 4 * sizeof(double), hipMemcpyDeviceToHost,
                           m->p->stream));
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    return VRX_OK;
}

// upload a probability table and turn it into row-normalised logs on the device
static int upload_log_rows(vrx_model* m, DevBuf<double>& dst, const double* src, int64_t rows,
                           int cols) {
    const size_t n = (size_t)(rows * cols);
    VRX_HIP(dst.alloc(n));
    VRX_HIP(m->tmp.alloc(n));
    VRX_HIP(hipMemcpyAsync(m->tmp.p, src, n * sizeof(double), hipMemcpyHostToDevice, m->p->stream));
    const int nb = (int)((rows + VRX_BLOCK - 1) / VRX_BLOCK);
    vrx_log_rows<<<nb, VRX_BLOCK, 0, m->p->stream>>>(rows, cols, m->tmp.p, dst.p);
    VRX_HIP(hipGetLastError());
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    m->tmp.release();
    return VRX_OK;
}

extern "C" int vrx_model_set_prior(vrx_model* m, const double* ID_prior, int64_t id_rows,
                                   const double* GT_prior, int64_t gt_rows, const double* s1_prior,
                                   const double* s2_prior, int64_t theta_prior_rows) {
    VRX_REQUIRE(m, "vrx_model_set_prior: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    VRX_REQUIRE(id_rows == 0 || id_rows == 1 || id_rows == m->M,
                "vrx_model_set_prior: ID_prior must have 0, 1 or n_cell rows (got %lld)",
                (long long)id_rows);
    int rc;
    if (id_rows == 0) {
        m->id_mode = 0;
        m->logq_id.release();
    } else {
        VRX_REQUIRE(ID_prior, "vrx_model_set_prior: null ID_prior");
        if ((rc = upload_log_rows(m, m->logq_id, ID_prior, id_rows, m->K))) return rc;
        m->id_mode = id_rows == 1 ? 1 : 2;
    }
    if (m->cfg.kind == VRX_KIND_VIREO) {
        VRX_REQUIRE(gt_rows == 0 || gt_rows == 1 || gt_rows == m->N,
                    "vrx_model_set_prior: GT_prior must have 0, 1 or n_var rows (got %lld)",
                    (long long)gt_rows);
        if (gt_rows == 0) {
            m->gt_mode = 0;
            m->logq_gt.release();
        } else {
            VRX_REQUIRE(GT_prior, "vrx_model_set_prior: null GT_prior");
            if ((rc = upload_log_rows(m, m->logq_gt, GT_prior, gt_rows * m->K, m->T))) return rc;
            m->gt_mode = gt_rows == 1 ? 1 : 2;
        }
    }
    VRX_REQUIRE(s1_prior && s2_prior, "vrx_model_set_prior: null theta prior");
    VRX_REQUIRE(theta_prior_rows == 1 || theta_prior_rows == m->th_rows,
                "vrx_model_set_prior: theta prior rows must be 1 or %lld", (long long)m->th_rows);
    m->prior_rows = theta_prior_rows;
    const size_t n = (size_t)(theta_prior_rows * m->th_cols);
    if ((rc = h2d(m, m->prior1, s1_prior, n))) return rc;
    if ((rc = h2d(m, m->prior2, s2_prior, n))) return rc;
    VRX_HIP(hipStreamSynchronize(m->p->stream));
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------
template <int LPE, int CPL, int MODE>
static void launch_spmm_fmt(const Orient& o, dim3 grid, hipStream_t s, const double* X, int K,
                            double* out, double* partial, const int32_t* ctl, int R,
                            const VrxCellFuse* fuse = nullptr) {
#define VRX_GO(F, FU)                                                                           \
    vrx_spmm<LPE, CPL, MODE, F, FU><<<grid, VRX_BLOCK, 0, s>>>(                                 \
        o.n_seg, o.seg_begin.p, o.seg_len.p, o.seg_dst.p, o.ent.p, X, K, out, partial, ctl, R,  \
        fuse ? *fuse : VrxCellFuse{})
    if constexpr (MODE == 1 && CPL == 1) {
        if (fuse) {  // cell pass + softmax + ELBO partials in one launch (vrx_kernels.h: FUSE = 1)
            if (o.fmt == VRX_FMT_P32)
                VRX_GO(VRX_FMT_P32, 1);
            else if (o.fmt == VRX_FMT_P64)
                VRX_GO(VRX_FMT_P64, 1);
            else
                VRX_GO(VRX_FMT_WIDE, 1);
            return;
        }
    }
    if (o.fmt == VRX_FMT_P32)
        VRX_GO(VRX_FMT_P32, 0);
    else if (o.fmt == VRX_FMT_P64)
        VRX_GO(VRX_FMT_P64, 0);
    else
        VRX_GO(VRX_FMT_WIDE, 0);
#undef VRX_GO
}

// LDS-resident pass: K <= 16 (4 columns per lane, the dense rows zero-padded to a multiple of
// 4 columns in LDS), counts < 2048 (checked when the tiled stream is built)
template <int MODE>
static bool lds_eligible(const Orient& o, int K) {
    static const int mask = env_int("VIREO_LDS_PASS", 3);  // bit 0: variant pass, bit 1: cell pass
    static const int kmin = env_int("VIREO_LDS_MIN_K", 2);
    static const int kmax = env_int("VIREO_LDS_MAX_K", 1 << 20);  // > 16: column blocks of 16
    if (o.tiled.ready && o.n_seg == 0) return true;  // (device-built: no gather tables)
    return o.tiled.ready && (mask >> MODE & 1) && K <= kmax && K >= kmin;
}

// kernel instance for K: zero-padded rows when K % 4, 2 / 4 entries at once when K <= 8 / 4
template <int LPE, int MODE, int RW>
static auto lds_kernel_rw(int K, bool strided) {
    static const int split_on = env_int("VIREO_LDS_SPLIT_K", 1);
    const int split = std::min(LPE, !split_on ? 1 : K <= 4 ? 4 : K <= 8 ? 2 : 1);
    // (the element-wise slab copy handles row strides and rows that do not fill whole lanes)
    const bool pad = K % (16 / LPE) != 0 || strided;
    if constexpr (LPE >= 4)
        if (split == 4)
            return pad ? vrx_spmm_lds<LPE, MODE, RW, 1, 4> : vrx_spmm_lds<LPE, MODE, RW, 0, 4>;
    if constexpr (LPE >= 2)
        if (split == 2)
            return pad ? vrx_spmm_lds<LPE, MODE, RW, 1, 2> : vrx_spmm_lds<LPE, MODE, RW, 0, 2>;
    return pad ? vrx_spmm_lds<LPE, MODE, RW, 1, 1> : vrx_spmm_lds<LPE, MODE, RW, 0, 1>;
}

// the AD/BD form of the cell pass (FORM 1): one instance for K = 16, one that stages
// element-wise and masks its stores for every other K / column block
// (pad: 0 flat rows of 16 columns, 1 element-wise, 2 whole 16-B units -- see the kernel)
template <int RW>
static auto lds_kernel_form1(int pad) {
    return pad == 0   ? vrx_spmm_lds<VRX_LDS_LPE, 1, RW, 0, 1, 1>
           : pad == 1 ? vrx_spmm_lds<VRX_LDS_LPE, 1, RW, 1, 1, 1>
                      : vrx_spmm_lds<VRX_LDS_LPE, 1, RW, 2, 1, 1>;
}

// rows per wave: the pass default, or (cell pass) the shorter tile of short_tile_pays()
template <int LPE, int MODE>
static auto lds_kernel(int K, int ld, bool strided, int rw, int form) {
    // AD/BD forms: flat rows, or whole 16-B units (even K and row stride), or element-wise
    const int pad = K == 16 && !strided ? 0 : ((K | ld) & 1) == 0 ? 2 : 1;
    // (the caller has checked rw against the heights compiled for this mode and form: lds_rw_ok)
    if (MODE == 1 && form == 1) {
        return rw == VRX_LDS_RW_CELL_SHORT ? lds_kernel_form1<VRX_LDS_RW_CELL_SHORT>(pad)
                                           : lds_kernel_form1<VRX_LDS_RW_CELL>(pad);
    }
    if (MODE == 0 && form == 2) {
        return pad == 0   ? vrx_spmm_lds<VRX_LDS_LPE, 0, VRX_LDS_RW_VARIANT, 0, 1, 2>
               : pad == 1 ? vrx_spmm_lds<VRX_LDS_LPE, 0, VRX_LDS_RW_VARIANT, 1, 1, 2>
                          : vrx_spmm_lds<VRX_LDS_LPE, 0, VRX_LDS_RW_VARIANT, 2, 1, 2>;
    }
    if (MODE == 1 && rw == VRX_LDS_RW_CELL_SHORT)
        return lds_kernel_rw<LPE, MODE, MODE == 1 ? VRX_LDS_RW_CELL_SHORT : VRX_LDS_RW_VARIANT>(K, strided);
    return lds_kernel_rw<LPE, MODE, MODE == 1 ? VRX_LDS_RW_CELL_PAIR : VRX_LDS_RW_VARIANT>(K, strided);
}

// the tile heights a kernel instance exists for: a stream built with any other height would be
// walked with the wrong rows-per-wave (bnd / rowmap strides) -- refuse instead of defaulting
template <int MODE>
static bool lds_rw_ok(int rw, int form) {
    if (MODE == 1 && form == 1) return rw == VRX_LDS_RW_CELL || rw == VRX_LDS_RW_CELL_SHORT;
    if (MODE == 0) return rw == VRX_LDS_RW_VARIANT;
    return rw == VRX_LDS_RW_CELL_PAIR || rw == VRX_LDS_RW_CELL_SHORT;
}

template <int LPE, int MODE>
static int launch_lds_one(const Orient& o, hipStream_t s, const double* X, int K, double* dst,
                          const int32_t* ctl, int R) {
    const TiledStream& t = o.tiled;
    VRX_REQUIRE(lds_rw_ok<MODE>(t.rw, t.form),
                "LDS pass: no kernel instance for %d rows per wave (mode %d, form %d)", t.rw, MODE, t.form);
    constexpr int XD = MODE == 1 ? 2 : 1, NV = MODE == 0 ? 2 : 1;
    const unsigned grid = (unsigned)t.n_wg;  // persistent: one workgroup per CU walks its items
    // operands wider than 16 columns go through in blocks of 16 (the stream is re-read per
    // block, like the column chunks of the gather kernels)
    for (int c0 = 0; c0 < K; c0 += 16) {
        const int kb = std::min(16, K - c0);
        const bool f1 = MODE == 1 && t.form == 1;  // planar operand, 256-B LDS rows
        constexpr int CPL = 16 / LPE;  // columns per lane: LDS rows hold whole lanes
        const bool f2 = MODE == 0 && t.form == 2;  // 128-B LDS rows whatever K
        const size_t lds = (size_t)t.slab_rows * (f1 ? 256 : f2 ? 128 : (kb + CPL - 1) / CPL * CPL * (MODE == 1 ? 16 : 8)) +
                           VRX_LDS_WAVES * VRX_RING * 4;
        auto kern = lds_kernel<LPE, MODE>(kb, K, K > 16, t.rw, t.form);
        VRX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        kern<<<grid, VRX_LDS_WAVES * 64, lds, s>>>(t.ent.p, t.wave_start.p, t.bnd.p, t.rowmap.p, t.items.p,
                                     t.wg_first.p, t.n_slab,
                                     t.slab_rows, t.n_contract, t.n_vrows,
                                     X + (size_t)c0 * (f1 ? 1 : XD), kb, K, dst + (size_t)c0 * NV, ctl, R,
                                     t.balanced ? t.perm.p : nullptr);
        VRX_HIP(hipGetLastError());
    }
    return VRX_OK;
}

template <int MODE>
static int launch_spmm_lds(vrx_model* m, const Orient& o, const double* X, int K, double* out,
                           double* range_partial, bool defer_sum) {
    hipStream_t s = m->p->stream;
    const TiledStream& t = o.tiled;
    constexpr int NV = MODE == 0 ? 2 : 1;
    if (MODE == 0 && t.virt) {
        // virtual rows: the cell pass's kernel over (variant, AD) / (variant, BD) rows and the
        // operand as double rows; planar partial sums [slot][virtual piece][K], turned into
        // S = (S1, S1 + S2) by the consumer (vrx_theta_partial) or by vrx_s_from_virtual
        int rc = launch_lds_one<VRX_LDS_LPE, 1>(o, s, X, K, range_partial, m->ctl.p, m->R);
        if (rc) return rc;
        if (!defer_sum) {
            const int64_t n = o.n_rows * K;
            vrx_s_from_virtual<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                o.n_rows, K, t.n_vrows, t.split ? t.vptr.p : nullptr, t.npiece.p, range_partial,
                reinterpret_cast<double2*>(out), m->ctl.p, m->R);
            VRX_HIP(hipGetLastError());
        }
        return VRX_OK;
    }
    double* dst = t.n_range == 1 && !t.split ? out : range_partial;
    int rc;
    rc = launch_lds_one<VRX_LDS_LPE, MODE>(o, s, X, K, dst, m->ctl.p, m->R);  // K < 16 leaves lanes idle
    if (rc) return rc;
    if (t.split && !(MODE == 1 && defer_sum)) {  // rows cut into pieces: sum pieces and ranges in one fixed order
        // (the cell pass's consumer, vrx_cell_softmax, sums the pieces itself when asked to: defer_sum)
        const int64_t n = o.n_rows * K * NV;
        if ((int64_t)t.n_range * t.n_vrows >= 64 * o.n_rows)  // >= 64 terms per row on average
            vrx_sum_pieces_wave<<<(unsigned)((n * 64 + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                o.n_rows, K * NV, t.n_range, t.n_vrows, t.vptr.p, t.npiece.p, range_partial, out, m->ctl.p, m->R);
        else
            vrx_sum_pieces<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
                o.n_rows, K * NV, t.n_range, t.n_vrows, t.vptr.p, t.npiece.p, range_partial, out, m->ctl.p, m->R);
        VRX_HIP(hipGetLastError());
    } else if (t.n_range > 1 && !defer_sum) {
        const int64_t n = o.n_rows * K * NV;
        vrx_sum_ranges<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            n, K * NV, t.npiece.p, range_partial, out, m->ctl.p, m->R);
        VRX_HIP(hipGetLastError());
    }
    return VRX_OK;
}

template <int MODE>
static int launch_spmm(vrx_model* m, const Orient& o, const double* X, int K, double* out,
                       double* partial, const VrxCellFuse* fuse = nullptr) {
    if (o.n_seg == 0) return VRX_OK;
    hipStream_t s = m->p->stream;
    // lanes per entry x columns per lane: 16 B per lane wherever the layout allows
    const int cpl = (MODE == 0 && K % 2 == 0) ? 2 : 1;
    const int lpe = pick_kp((K + cpl - 1) / cpl, 16 / cpl);
    const int cols = lpe * cpl;
    dim3 grid((unsigned)(o.n_seg / VRX_WAVES), (unsigned)((K + cols - 1) / cols));
    if (cpl == 2) {
        if (MODE == 0) {  // (guard keeps the CPL=2 cell-pass templates from being instantiated)
            switch (lpe) {
                case 1: launch_spmm_fmt<1, 2, 0>(o, grid, s, X, K, out, partial, m->ctl.p, m->R); break;
                case 2: launch_spmm_fmt<2, 2, 0>(o, grid, s, X, K, out, partial, m->ctl.p, m->R); break;
                case 4: launch_spmm_fmt<4, 2, 0>(o, grid, s, X, K, out, partial, m->ctl.p, m->R); break;
                default: launch_spmm_fmt<8, 2, 0>(o, grid, s, X, K, out, partial, m->ctl.p, m->R); break;
            }
        }
    } else {
        switch (lpe) {
            case 1: launch_spmm_fmt<1, 1, MODE>(o, grid, s, X, K, out, partial, m->ctl.p, m->R, fuse); break;
            case 2: launch_spmm_fmt<2, 1, MODE>(o, grid, s, X, K, out, partial, m->ctl.p, m->R, fuse); break;
            case 4: launch_spmm_fmt<4, 1, MODE>(o, grid, s, X, K, out, partial, m->ctl.p, m->R, fuse); break;
            case 8: launch_spmm_fmt<8, 1, MODE>(o, grid, s, X, K, out, partial, m->ctl.p, m->R, fuse); break;
            default: launch_spmm_fmt<16, 1, MODE>(o, grid, s, X, K, out, partial, m->ctl.p, m->R, fuse); break;
        }
    }
    VRX_HIP(hipGetLastError());
    if (o.n_multi > 0) {
        constexpr int VPE = MODE == 0 ? 2 : 1;
        const int64_t tot = o.n_multi * K * VPE;
        vrx_sum_slots<VPE><<<(unsigned)((tot + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
            o.n_multi, K, o.multi_row.p, o.multi_ptr.p, partial, out, m->ctl.p, m->R);
        VRX_HIP(hipGetLastError());
    }
    return VRX_OK;
}

// S <- (AD @ ID_prob, DP @ ID_prob)        vireo_model.py:169-170,207-208; bmm_model.py:137-138
static int variant_pass(vrx_model* m, bool defer_sum = false) {
    ProfScope ps(m, VRX_KERN_VARIANT_PASS);
    if (lds_eligible<0>(m->p->by_var, m->Kt)) {
        const TiledStream& tv = m->p->by_var.tiled;
        m->s_pending = defer_sum && (tv.virt || (tv.n_range > 1 && !tv.split));
        return launch_spmm_lds<0>(m, m->p->by_var, m->ID.p, m->Kt, m->S.p, m->RV.p, defer_sum);
    }
    return launch_spmm<0>(m, m->p->by_var, m->ID.p, m->Kt, m->S.p, m->PV.p);
}

// LID <- AD^T W1 + DP^T W2                 vireo_model.py:190-196; bmm_model.py:125-129
static int cell_pass(vrx_model* m, bool defer_sum = false) {
    ProfScope ps(m, VRX_KERN_CELL_PASS);
    if (lds_eligible<1>(m->p->by_cell, m->Kt)) {
        m->l_pending = defer_sum && (m->p->by_cell.tiled.n_range > 1 || m->p->by_cell.tiled.split);
        return launch_spmm_lds<1>(m, m->p->by_cell, m->W.p, m->Kt, m->LID.p, m->RC.p, defer_sum);
    }
    return launch_spmm<1>(m, m->p->by_cell, m->W.p, m->Kt, m->LID.p, m->PC.p);
}

// Small problems (gather kernels, one restart, K <= 16, every cell exactly one segment -- none
// split, none without entries):
// the cell pass's waves hold complete logLik_ID rows, so the softmax and the cells' ELBO terms
// ride in its epilogue -- one launch less per iteration (c2: ~4 us of ~37).
static bool cell_softmax_fusable(const vrx_model* m) {
    const int on = env_int("VIREO_FUSE_SOFTMAX", 1);  // (read per call: the tests switch it)
    const Orient& o = m->p->by_cell;
    return on && m->R == 1 && m->Kt <= 16 && o.n_seg > 0 && o.n_multi == 0 && o.n_empty == 0 &&
           !lds_eligible<1>(o, m->Kt);
}

static int cell_pass_softmax(vrx_model* m) {
    ProfScope ps(m, VRX_KERN_CELL_PASS);
    const Orient& o = m->p->by_cell;
    VrxCellFuse F;
    F.logq = m->logq_id.p;
    F.id_mode = m->id_mode;
    F.logq_uni = -std::log((double)m->K);
    F.ID = m->ID.p;
    F.part = m->part_cell.p;
    m->n_cell_part = (int)(o.n_seg / VRX_WAVES);
    m->l_pending = false;
    return launch_spmm<1>(m, o, m->W.p, m->Kt, m->LID.p, m->PC.p, &F);
}

// the rows a tiled stream cut into pieces: sum of all their terms -> slot 0 of the first piece
// (vrx_fold_split), for the consumers that sum the partial arrays themselves
static int fold_split(vrx_model* m, const TiledStream& t, double* partial) {
    if (t.n_split == 0) return VRX_OK;
    const int64_t lanes = t.n_split * m->Kt * 8;  // eight lanes per (split row, column)
    vrx_fold_split<<<(unsigned)((lanes + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, m->p->stream>>>(
        t.n_split, t.split_rows.p, m->Kt, t.n_vrows, t.vptr.p, t.npiece.p, partial, m->ctl.p, m->R);
    VRX_HIP(hipGetLastError());
    return VRX_OK;
}

// a consumer that cannot fuse the range sum forms S / logLik_ID explicitly
static int resolve_S(vrx_model* m) {
    if (!m->s_pending) return VRX_OK;
    const TiledStream& tv = m->p->by_var.tiled;
    if (tv.virt) {
        const int64_t n = m->NKt;
        vrx_s_from_virtual<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, m->p->stream>>>(
            m->N, m->Kt, tv.n_vrows, tv.split ? tv.vptr.p : nullptr, tv.npiece.p, m->RV.p,
            reinterpret_cast<double2*>(m->S.p), m->ctl.p, m->R);
        VRX_HIP(hipGetLastError());
        m->s_pending = false;
        return VRX_OK;
    }
    const int64_t n = m->NKt * 2;
    vrx_sum_ranges<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, m->p->stream>>>(
        n, m->Kt * 2, m->p->by_var.tiled.npiece.p, m->RV.p, m->S.p, m->ctl.p, m->R);
    VRX_HIP(hipGetLastError());
    m->s_pending = false;
    return VRX_OK;
}

static int resolve_LID(vrx_model* m) {
    if (!m->l_pending) return VRX_OK;
    const int64_t n = m->M * m->Kt;
    const TiledStream& tc = m->p->by_cell.tiled;
    if (tc.split)
        vrx_sum_pieces<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, m->p->stream>>>(
            m->M, m->Kt, tc.n_range, tc.n_vrows, tc.vptr.p, tc.npiece.p, m->RC.p, m->LID.p, m->ctl.p, m->R);
    else
    vrx_sum_ranges<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, m->p->stream>>>(
        n, m->Kt, m->p->by_cell.tiled.npiece.p, m->RC.p, m->LID.p, m->ctl.p, m->R);
    VRX_HIP(hipGetLastError());
    m->l_pending = false;
    return VRX_OK;
}

static VrxElboIn elbo_inputs(vrx_model* m);

// theta update (update=1) or just psi/KL from the current beta (update=0).  defer_final: the
// caller runs gt_step next, whose kernel finalises the shared theta itself (VrxThetaFuse).
static int theta_step(vrx_model* m, int update, bool defer_final = false) {
    ProfScope ps(m, VRX_KERN_DENSE);
    hipStream_t s = m->p->stream;
    const auto& c = m->cfg;
    const TiledStream& tvar = m->p->by_var.tiled;
    const bool bmm_fuse = c.kind == VRX_KIND_BMM && update && m->s_pending && !tvar.virt && !tvar.split;
    if ((c.kind == VRX_KIND_BMM || c.ase_mode) && !bmm_fuse) {
        int rc = resolve_S(m);  // (the ASE kernel does not fuse the range sum)
        if (rc) return rc;
    }
    if (c.kind == VRX_KIND_BMM) {
        // (fused range sum: 16 lanes per element; the KL partials are zero-filled past nb_nk blocks' worth)
        const unsigned nb = bmm_fuse ? (unsigned)((m->NK * 16 + VRX_BLOCK - 1) / VRX_BLOCK) : (unsigned)m->nb_nk;
        VrxElboRide E{};
        if (m->elbo_deferred) {  // the previous iteration's ELBO + stop rule: one extra block, reading
            E.on = 1;            // the KL_theta partials of the half this launch does not write
            E.in = elbo_inputs(m);
            E.rule = m->elbo_rule;
            m->elbo_deferred = false;
        }
        m->th_cur ^= 1;
        m->n_th_part = (int)nb;
        vrx_bmm_theta<<<dim3(nb + E.on, m->R), VRX_BLOCK, 0, s>>>(
            m->NK, update, c.fix_beta_sum, reinterpret_cast<double2*>(m->S.p),
            bmm_fuse ? tvar.npiece.p : nullptr, reinterpret_cast<const double2*>(m->RV.p), m->prior1.p,
            m->prior2.p, m->prior_rows == 1 ? 0 : 1, m->mu.p, m->sm.p, m->W.p, m->K, m->wform,
            m->part_th.p + (size_t)m->th_cur * m->th_cap, m->batch(), m->ctl.p, E);
        if (bmm_fuse) m->s_pending = false;
        m->w_valid = true;
    } else if (c.ase_mode) {
        vrx_theta_ase<<<dim3(m->nb_throws, m->R), VRX_BLOCK, 0, s>>>(
            m->N, m->K, m->T, update, c.fix_beta_sum, reinterpret_cast<const double2*>(m->S.p),
            m->GT.p, m->prior1.p, m->prior2.p, (int)m->prior_rows, m->mu.p, m->sm.p, m->psi.p,
            m->part_th.p, m->batch(), m->ctl.p);
        m->w_valid = false;
    } else {
        if (update) {
            const TiledStream& tv = m->p->by_var.tiled;
            if (m->s_pending && tv.virt && tv.split) {  // rows cut into pieces: their terms first
                int rc = fold_split(m, tv, m->RV.p);
                if (rc) return rc;
            }
            const uint16_t* np = m->s_pending ? tv.npiece.p : nullptr;
            auto* kern = m->T == 3 ? vrx_theta_partial<3> : vrx_theta_partial<VRX_MAXT>;
            VrxElboRide E{};
            if (m->elbo_deferred) {  // the previous iteration's ELBO + stop rule: one extra block
                E.on = 1;
                E.in = elbo_inputs(m);
                E.rule = m->elbo_rule;
                m->elbo_deferred = false;
            }
            kern<<<dim3(m->nb_theta + E.on, m->R), VRX_BLOCK, 0, s>>>(
                m->NK, m->T, reinterpret_cast<double2*>(m->S.p), np,
                reinterpret_cast<const double2*>(m->RV.p), m->s_pending && tv.virt ? tv.n_vrows : 0,
                m->s_pending && tv.virt && tv.split ? tv.vptr.p : nullptr,
                m->GT.p, m->part_theta.p, m->batch(), m->ctl.p, E);
            VRX_HIP(hipGetLastError());
            m->s_pending = false;
        }
        // Worth it only while the partials are few: every block of vrx_gt_update re-reads them
        // (c2, 157 partials: 43.4 -> 41.9 us per iteration; c3, 1024 partials = 128 KB per
        // block: 0.994 -> 1.012 ms, so large problems keep the separate one-block kernel).
        static const int fuse_max = env_int("VIREO_FUSE_THETA_MAX_PARTS", 256);
        if (update && defer_final && m->nb_theta <= fuse_max) {
            m->theta_pending = true;
        } else {
            vrx_theta_final<<<m->R, VRX_BLOCK, 0, s>>>(m->nb_theta, m->T, update, c.fix_beta_sum,
                                                     m->part_theta.p, m->prior1.p, m->prior2.p,
                                                     m->mu.p, m->sm.p, m->psi.p, m->part_th.p, m->ctl.p);
        }
        m->w_valid = false;
    }
    VRX_HIP(hipGetLastError());
    return VRX_OK;
}

// GT softmax (learn=1) or W/KL from the fixed GT (learn=0); always refreshes W
static int gt_step(vrx_model* m, int learn) {
    ProfScope ps(m, VRX_KERN_DENSE);
    if (learn) {
        int rc = resolve_S(m);
        if (rc) return rc;
    }
    VrxThetaFuse F{};
    if (m->theta_pending) {
        F.on = 1;
        F.n_part = m->nb_theta;
        F.fix_sum = m->cfg.fix_beta_sum;
        F.part = m->part_theta.p;
        F.prior1 = m->prior1.p;
        F.prior2 = m->prior2.p;
        F.mu = m->mu.p;
        F.sm = m->sm.p;
        F.psi = m->psi.p;
        F.kl_out = m->part_th.p;
        m->theta_pending = false;
    }
    auto* kern = m->T == 3 ? vrx_gt_update<3> : vrx_gt_update<VRX_MAXT>;
    kern<<<dim3(m->nb_gt, m->R), VRX_BLOCK, 0, m->p->stream>>>(
        m->NK, m->K, m->T, learn, m->cfg.ase_mode, m->N, reinterpret_cast<const double2*>(m->S.p),
        m->psi.p, m->logq_gt.p, m->gt_mode, -std::log((double)m->T), m->GT.p, m->W.p, m->wform,
        m->part_gt.p, F, m->batch(), m->ctl.p);
    VRX_HIP(hipGetLastError());
    m->w_valid = true;
    return VRX_OK;
}

static VrxElboIn elbo_inputs(vrx_model* m) {
    VrxElboIn e;
    e.cell_part = m->part_cell.p;
    e.gt_part = m->part_gt.p;
    e.th_part = m->part_th.p + (size_t)m->th_cur * m->th_cap;
    e.n_cell_part = m->n_cell_part;  // (of the kernel that formed them last)
    e.n_gt_part = m->cfg.kind == VRX_KIND_VIREO ? m->nb_gt : 0;
    e.n_th_part = m->n_th_part;
    e.elbo = m->d_elbo.p;
    e.parts = m->d_parts.p;
    e.trace_stride = m->trace_cap;
    return e;
}

static VrxStopRule no_rule(int slot) {
    VrxStopRule r;
    r.it = slot;
    r.min_iter = r.max_iter = r.active = 0;
    r.eps = 0.0;
    return r;
}

static int softmax_step(vrx_model* m, int update) {
    ProfScope ps(m, VRX_KERN_DENSE);
    hipStream_t s = m->p->stream;
    const double lu = -std::log((double)m->K);
    const TiledStream& tcs = m->p->by_cell.tiled;
    if (m->l_pending && tcs.split) {  // rows cut into pieces: their terms first
        int rc = fold_split(m, tcs, m->RC.p);
        if (rc) return rc;
    }
    const uint16_t* nr = m->l_pending ? tcs.npiece.p : nullptr;  // fused sum of the partials
    const int32_t* vp = m->l_pending && tcs.split ? tcs.vptr.p : nullptr;  // ... of the pieces of long rows too
    const int64_t nvr = m->l_pending ? tcs.n_vrows : m->M;
    m->l_pending = false;
    m->n_cell_part = m->nb_cell;
#define VRX_SM_CASE(KPV)                                                                        \
    case KPV:                                                                                   \
        vrx_cell_softmax<KPV><<<dim3(m->nb_cell, m->R), VRX_BLOCK, 0, s>>>(                     \
            m->M, m->K, update, m->LID.p, nr, m->RC.p, vp, nvr, m->logq_id.p, m->id_mode, lu, m->ID.p, \
            m->part_cell.p, m->batch(), m->ctl.p);                                              \
        break;
    switch (m->KP) {
        VRX_SM_CASE(1)
        VRX_SM_CASE(2)
        VRX_SM_CASE(4)
        VRX_SM_CASE(8)
        VRX_SM_CASE(16)
        VRX_SM_CASE(32)
        VRX_SM_CASE(64)
    }
#undef VRX_SM_CASE
    VRX_HIP(hipGetLastError());
    return VRX_OK;
}

// ELBO of iteration rule.it into the trace; evaluates the stop rule when it is active
static int elbo_step(vrx_model* m, const VrxStopRule& rule) {
    ProfScope ps(m, VRX_KERN_DENSE);
    vrx_elbo_final<<<m->R, VRX_BLOCK, 0, m->p->stream>>>(elbo_inputs(m), rule, m->ctl.p);
    VRX_HIP(hipGetLastError());
    return VRX_OK;
}

// One iteration of _fit_VB (vireo_model.py:257-264) / _fit_BV (bmm_model.py:183-188).
// Enqueue only; no host synchronisation.
// an ELBO that still waits for a ride is finalised by the kernel of its own after all
static int flush_elbo(vrx_model* m) {
    if (!m->elbo_deferred) return VRX_OK;
    m->elbo_deferred = false;
    return elbo_step(m, m->elbo_rule);
}

// shared-theta Vireo updates run vrx_theta_partial, which can carry the previous iteration's ELBO
static bool elbo_can_ride(const vrx_model* m, int min_iter) {
    // Only where an iteration is short against a launch: when the rule fires, the next iteration's
    // variant pass has already run for nothing -- 8 us at c2 (a 29-us iteration minus 3 us, every
    // iteration), 0.3 ms at c3 (where one ELBO kernel per iteration is 1 % of it and a fit would
    // need > 30 iterations to win the wasted pass back).  Measured gain at nnz x columns = 2 / 8 /
    // 16 / 32 M: 10.5 / 6 / 4 / 2.8 % per iteration (profiles/r05_ab_elbo_ride_small_problems.txt);
    // the default stops at 2^25.  VIREO_ELBO_RIDE=0 / 1 forces it off / on (read per call).
    // Clone mode rides in vrx_bmm_theta, for the iterations whose stop rule cannot fire (it <=
    // min_iter, see vrx_model_fit): its fits run min_iter >= 20 iterations by default
    // (bmm_model.py:178), so there the rule is min_iter, not size -- nothing is ever wasted.
    const auto& c = m->cfg;
    if (c.kind == VRX_KIND_VIREO && (c.ase_mode || !c.learn_theta)) return false;
    const bool small = m->p->nnz * (int64_t)m->Kt < ((int64_t)1 << 25);
    const bool dflt = small || (c.kind == VRX_KIND_BMM && min_iter >= 12);
    return env_int("VIREO_ELBO_RIDE", dflt ? 1 : 0) != 0;
}

// defer_elbo: the caller enqueues an iteration WITH the theta update right behind this one
static int enqueue_iteration(vrx_model* m, bool do_theta, const VrxStopRule& rule, bool defer_elbo = false) {
    int rc;
    const auto& c = m->cfg;
    if (m->elbo_deferred && !(c.kind == VRX_KIND_BMM || (do_theta && !c.ase_mode)))
        if ((rc = flush_elbo(m))) return rc;  // (not reached by the callers below: they defer only in front of a ride)
    if (c.kind == VRX_KIND_BMM) {
        if ((rc = variant_pass(m, true))) return rc;
        if ((rc = theta_step(m, 1))) return rc;  // also refreshes W (digamma tables)
    } else {
        bool have_s = false;
        if (do_theta) {
            if ((rc = variant_pass(m, true))) return rc;  // range sum fused into the theta kernel
            have_s = true;
            if ((rc = theta_step(m, 1, true))) return rc;  // (a gt_step always follows)
        }
        if (c.learn_gt) {
            // the reference recomputes AD@ID_prob, DP@ID_prob here (vireo_model.py:207-208);
            // ID_prob has not changed since update_theta_size, so S is reused.
            if (!have_s)
                if ((rc = variant_pass(m, true))) return rc;
            if ((rc = gt_step(m, 1))) return rc;
        } else if (!m->w_valid) {
            if ((rc = gt_step(m, 0))) return rc;
        }
    }
    if (cell_softmax_fusable(m)) {
        if ((rc = cell_pass_softmax(m))) return rc;
    } else {
        if ((rc = cell_pass(m, true))) return rc;  // range sum fused into the softmax kernel
        if ((rc = softmax_step(m, 1))) return rc;
    }
    if (defer_elbo) {  // finalised by the next iteration's vrx_theta_partial (theta_step)
        m->elbo_deferred = true;
        m->elbo_rule = rule;
        return VRX_OK;
    }
    return elbo_step(m, rule);                 // ELBO + the stop rule, on the device
}

static int reset_ctl(vrx_model* m) {  // stop flag, stop iteration, warn flags (tickets stay 0)
    m->elbo_deferred = false;  // (a call that failed half-way may have left one waiting: dropped)
    VRX_HIP(hipMemsetAsync(m->ctl.p, 0, (size_t)m->R * VRX_CTL_WORDS * sizeof(int32_t), m->p->stream));
    return VRX_OK;
}

// psi / KL_theta (and, for fixed GT or BMM, W) consistent with the state just uploaded
static int prepare(vrx_model* m) {
    int rc;
    if ((rc = theta_step(m, 0))) return rc;
    if (m->cfg.kind == VRX_KIND_VIREO && !m->cfg.learn_gt)
        if ((rc = gt_step(m, 0))) return rc;
    return VRX_OK;
}

extern "C" int vrx_model_fit(vrx_model* m, int32_t max_iter, int32_t min_iter, double eps,
                             int32_t delay_fit_theta, double* elbo_trace, int32_t* it_out,
                             int32_t* warn_flags) {
    VRX_REQUIRE(m && elbo_trace && it_out, "vrx_model_fit: null argument");
    VRX_REQUIRE(max_iter >= 1, "vrx_model_fit: max_iter must be >= 1");
    VRX_HIP(hipSetDevice(m->p->device));
    {   // (the reference takes any max_iter, vireo_model.py:251: its trace is np.zeros(max_iter))
        int rc0 = ensure_trace(m, max_iter);
        if (rc0) return rc0;
    }
    hipStream_t s = m->p->stream;
    int rc;
    if ((rc = reset_ctl(m))) return rc;
    if ((rc = prepare(m))) return rc;
    // The stop rule runs on the device (vrx_elbo_final_block); the host enqueues a batch of
    // iterations, then reads three control words.  The first batch reaches the first iteration
    // the rule can fire at (min_iter + 1); a kernel launched after the stop returns at once, so
    // an overshoot costs launches, not work.
    static const int batch = std::max(1, env_int("VIREO_FIT_BATCH", 4));
    // Polls are PIPELINED: batch b + 1 is enqueued before the host waits for the control words
    // batch b left, so the device never idles between batches (a 20-iteration restart used to
    // pay five drained queues).  VIREO_FIT_PIPELINE=0: wait before enqueuing, as before.
    // A batch enqueued behind the one that stopped costs its launches (batch x ~5-10 no-op kernels,
    // drained by the final synchronisation): on launch-bound problems that is more than the drained
    // queues it saves (c2, 32 one-restart fits: 24.8 ms pipelined against 23.1 ms,
    // profiles/r05_ab_pipeline_small_problems.txt), so the default pipelines only where an
    // iteration is long against a launch -- the criterion restarts.restart_batch uses.
    // VIREO_FIT_PIPELINE=1 / 0 forces it on / off (read per call: the tests switch it).
    const int pipeline = env_int("VIREO_FIT_PIPELINE", m->p->nnz * (int64_t)m->Kt >= ((int64_t)1 << 24) ? 1 : 0);
    const bool ride = elbo_can_ride(m, min_iter);
    const int R = m->R;  // elbo_trace [R][max_iter], it_out [R], warn_flags [R]
    // two pinned read-back buffers of R * VRX_CTL_WORDS <= 64 words inside h_pin (64 doubles)
    int32_t* hbuf[2] = {reinterpret_cast<int32_t*>(m->h_pin), reinterpret_cast<int32_t*>(m->h_pin) + 64};
    for (int b = 0; b < 2; ++b)
        if (!m->polled[b]) VRX_HIP(hipEventCreateWithFlags(&m->polled[b], hipEventDisableTiming));
    int it = 0, next = 0, nb = 0;
    auto enqueue_batch = [&]() -> int {  // iterations [next, upto) + the read-back of their control words
        const int upto = std::min(max_iter, next == 0 ? std::max(min_iter + 2, batch) : next + batch);
        for (it = next; it < upto; ++it) {
            VrxStopRule rule;
            rule.it = it;
            rule.min_iter = min_iter;
            rule.max_iter = max_iter;
            rule.active = 1;
            rule.eps = eps;
            const bool do_theta = m->cfg.kind == VRX_KIND_VIREO && m->cfg.learn_theta &&
                                  it >= delay_fit_theta;
            // (the last iteration of a batch finalises its ELBO itself: the poll reads its stop word)
            // Clone mode: vrx_bmm_theta WRITES model state (beta_mu, beta_sum, W), and its ordinary
            // blocks read the stop word before the rider in the same launch has judged the previous
            // iteration -- a stop found there would leave theta one update ahead of what
            // bmm_model.py:190-199 breaks out with.  So an ELBO rides only while its rule cannot
            // fire (it <= min_iter: `judge` of vrx_elbo_final_block is false); from min_iter + 1 on
            // every iteration finalises its own.  (Vireo's vrx_theta_partial writes S and partial
            // sums only; the kernels that write state run behind the rider.)
            const bool defer = ride && it + 1 < upto &&
                               (m->cfg.kind == VRX_KIND_BMM ? it <= min_iter : it + 1 >= delay_fit_theta);
            int rc2;
            if ((rc2 = enqueue_iteration(m, do_theta, rule, defer))) return rc2;
        }
        next = upto;
        VRX_HIP(hipMemcpyAsync(hbuf[nb & 1], m->ctl.p, (size_t)R * VRX_CTL_WORDS * sizeof(int32_t),
                               hipMemcpyDeviceToHost, s));
        VRX_HIP(hipEventRecord(m->polled[nb & 1], s));
        ++nb;
        return VRX_OK;
    };
    if ((rc = enqueue_batch())) return rc;
    int32_t* hctl = hbuf[0];
    for (int b = 0;; ++b) {  // b: the batch whose control words are read next
        if (pipeline && next < max_iter)
            if ((rc = enqueue_batch())) return rc;  // (no-ops if batch b turns out to have stopped)
        VRX_HIP(hipEventSynchronize(m->polled[b & 1]));
        hctl = hbuf[b & 1];
        bool stopped = true;  // the batch runs until its last restart has stopped
        for (int r = 0; r < R; ++r) stopped = stopped && hctl[r * VRX_CTL_WORDS + VRX_CTL_STOP] != 0;
        if (stopped) break;
        if (b + 1 == nb) {  // nothing in flight behind batch b
            if (next >= max_iter) break;
            if ((rc = enqueue_batch())) return rc;
        }
    }
    // (a batch enqueued behind the one that stopped changes nothing: its kernels return at once,
    //  the control words are final; the sync below drains it)
    bool any_stop = false;
    for (int r = 0; r < R; ++r) {
        const int32_t* c = hctl + r * VRX_CTL_WORDS;
        any_stop = any_stop || c[VRX_CTL_STOP] != 0;
        // Python leaves `it` at the last executed index
        it = c[VRX_CTL_STOP] ? c[VRX_CTL_IT] : max_iter - 1;
        it_out[r] = it;
        if (warn_flags) warn_flags[r] = c[VRX_CTL_WARN];
        VRX_HIP(hipMemcpyAsync(elbo_trace + (size_t)r * max_iter, m->d_elbo.p + (size_t)r * m->trace_cap,
                               (size_t)(it + 1) * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    VRX_HIP(hipStreamSynchronize(s));
    if (any_stop) {  // the launches behind a stop did nothing; the next call starts clean
        if ((rc = reset_ctl(m))) return rc;
    }
    return prof_drain(m);
}

extern "C" int vrx_model_run_iters(vrx_model* m, int32_t n_iter, int32_t theta_from_iter,
                                   double* elbo_trace, double* ms_out) {
    VRX_REQUIRE(m && n_iter >= 1, "vrx_model_run_iters: bad argument");
    VRX_HIP(hipSetDevice(m->p->device));
    {
        int rc0 = ensure_trace(m, n_iter);
        if (rc0) return rc0;
    }
    hipStream_t s = m->p->stream;
    int rc;
    if ((rc = reset_ctl(m))) return rc;
    if ((rc = prepare(m))) return rc;
    const bool ride = elbo_can_ride(m, 1 << 20);  // (no stop rule here: nothing is ever wasted)
    VRX_HIP(hipEventRecord(m->t0, s));
    for (int it = 0; it < n_iter; ++it) {
        const bool do_theta = m->cfg.kind == VRX_KIND_VIREO && m->cfg.learn_theta &&
                              it >= theta_from_iter;
        const bool defer = ride && it + 1 < n_iter && (m->cfg.kind == VRX_KIND_BMM || it + 1 >= theta_from_iter);
        if ((rc = enqueue_iteration(m, do_theta, no_rule(it), defer))) return rc;
    }
    VRX_HIP(hipEventRecord(m->t1, s));
    VRX_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    VRX_HIP(hipEventElapsedTime(&ms, m->t0, m->t1));
    if (ms_out) *ms_out = ms;
    if (elbo_trace)  // [R][n_iter]
        for (int r = 0; r < m->R; ++r)
            VRX_HIP(hipMemcpy(elbo_trace + (size_t)r * n_iter, m->d_elbo.p + (size_t)r * m->trace_cap,
                              (size_t)n_iter * sizeof(double), hipMemcpyDeviceToHost));
    return prof_drain(m);
}

extern "C" int vrx_model_step(vrx_model* m, int32_t which, double* elbo_out) {
    VRX_REQUIRE(m, "vrx_model_step: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    hipStream_t s = m->p->stream;
    const auto& c = m->cfg;
    int rc;
    if ((rc = reset_ctl(m))) return rc;
    switch (which) {
        case VRX_STEP_THETA:
            if ((rc = variant_pass(m))) return rc;
            if ((rc = theta_step(m, 1))) return rc;
            break;
        case VRX_STEP_GT:
            VRX_REQUIRE(c.kind == VRX_KIND_VIREO, "vrx_model_step: GT step needs a Vireo model");
            if ((rc = theta_step(m, 0))) return rc;
            if ((rc = variant_pass(m))) return rc;
            if ((rc = gt_step(m, 1))) return rc;
            break;
        case VRX_STEP_ID:
        case VRX_STEP_LOGLIK:
            if ((rc = theta_step(m, 0))) return rc;  // BMM: refreshes W as well
            if (c.kind == VRX_KIND_VIREO)
                if ((rc = gt_step(m, 0))) return rc;
            if ((rc = cell_pass(m))) return rc;
            if (which == VRX_STEP_ID)
                if ((rc = softmax_step(m, 1))) return rc;
            break;
        case VRX_STEP_SOFTMAX:
            if ((rc = softmax_step(m, 1))) return rc;
            break;
        case VRX_STEP_ELBO:
            VRX_REQUIRE(elbo_out, "vrx_model_step: null elbo_out");
            if ((rc = theta_step(m, 0))) return rc;
            if (c.kind == VRX_KIND_VIREO)
                if ((rc = gt_step(m, 0))) return rc;
            if ((rc = softmax_step(m, 0))) return rc;
            if ((rc = elbo_step(m, no_rule(0)))) return rc;
            for (int r = 0; r < m->R; ++r)  // elbo_out [R]
                VRX_HIP(hipMemcpyAsync(m->h_pin + r, m->d_elbo.p + (size_t)r * m->trace_cap, sizeof(double),
                                       hipMemcpyDeviceToHost, s));
            VRX_HIP(hipStreamSynchronize(s));
            for (int r = 0; r < m->R; ++r) elbo_out[r] = m->h_pin[r];
            break;
        default:
            vrx_set_error("vrx_model_step: unknown step %d", which);
            return VRX_ERR_ARG;
    }
    VRX_HIP(hipStreamSynchronize(s));
    return prof_drain(m);
}

#ifdef VRX_PROBE_BUILD
// scratch builds only (scratch/vrx_probe.h): read and clear the per-wave records of vrx_spmm_lds
extern "C" int vrx_debug_probe_visits(unsigned long long* out) {
    VRX_HIP(hipDeviceSynchronize());
    VRX_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(vrx_probe_visit), sizeof(vrx_probe_visit)));
    return VRX_OK;
}
extern "C" int vrx_debug_probe(unsigned long long* rec) {
    VRX_HIP(hipDeviceSynchronize());
    VRX_HIP(hipMemcpyFromSymbol(rec, HIP_SYMBOL(vrx_probe_rec), sizeof(vrx_probe_rec)));
    void* dev = nullptr;
    VRX_HIP(hipGetSymbolAddress(&dev, HIP_SYMBOL(vrx_probe_rec)));
    VRX_HIP(hipMemset(dev, 0, sizeof(vrx_probe_rec)));
    return VRX_OK;
}
#endif

extern "C" int vrx_model_info(vrx_model* m, int32_t* info) {
    VRX_REQUIRE(m && info, "vrx_model_info: null argument");
    const Orient &v = m->p->by_var, &c = m->p->by_cell;
    info[0] = lds_eligible<0>(v, m->Kt);
    info[1] = lds_eligible<1>(c, m->Kt);
    info[2] = v.fmt;
    info[3] = c.fmt;
    info[4] = v.n_tiles;
    info[5] = c.n_tiles;
    info[6] = v.tiled.ready ? v.tiled.n_range : 0;
    info[7] = c.tiled.ready ? c.tiled.n_range : 0;
    info[8] = v.tiled.ready ? (int32_t)(v.tiled.pad_ratio * 1000.0 + 0.5) : 0;
    info[9] = c.tiled.ready ? (int32_t)(c.tiled.pad_ratio * 1000.0 + 0.5) : 0;
    info[10] = v.tiled.ready ? (int32_t)(v.tiled.n_vrows - (v.tiled.virt ? 2 : 1) * v.n_rows) : 0;
    info[11] = c.tiled.ready ? (int32_t)(c.tiled.n_vrows - c.n_rows) : 0;
    info[12] = c.tiled.ready ? c.tiled.form : 0;
    info[13] = v.tiled.ready ? (v.tiled.virt ? 3 : v.tiled.form) : 0;  // 3: AD/BD virtual rows
    info[14] = m->R;
    // longest wave stream / mean wave stream of each pass, x 1000: variant in the low, cell in the high half
    const int32_t iv = v.tiled.ready ? (int32_t)std::min(65535.0, v.tiled.imbalance * 1000.0 + 0.5) : 0;
    const int32_t ic = c.tiled.ready ? (int32_t)std::min(32767.0, c.tiled.imbalance * 1000.0 + 0.5) : 0;
    info[15] = iv | (ic << 16);
    return VRX_OK;
}

extern "C" int vrx_model_profile(vrx_model* m, int32_t enable) {
    VRX_REQUIRE(m, "vrx_model_profile: null model");
    VRX_HIP(hipSetDevice(m->p->device));
    if (enable && m->ev.empty()) {
        m->ev.resize(2 * kEventPairs);
        m->ev_kind.resize(kEventPairs);
        for (auto& e : m->ev) VRX_HIP(hipEventCreate(&e));
    }
    m->prof = enable != 0;
    m->ev_used = 0;
    for (int i = 0; i < VRX_KERN_COUNT; ++i) {
        m->prof_ms[i] = 0.0;
        m->prof_n[i] = 0;
    }
    return VRX_OK;
}

extern "C" int vrx_model_profile_read(vrx_model* m, double* ms_total, int64_t* launches) {
    VRX_REQUIRE(m && ms_total && launches, "vrx_model_profile_read: null argument");
    VRX_HIP(hipSetDevice(m->p->device));
    int rc = prof_drain(m);
    if (rc) return rc;
    for (int i = 0; i < VRX_KERN_COUNT; ++i) {
        ms_total[i] = m->prof_ms[i];
        launches[i] = m->prof_n[i];
    }
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// one-shot cell log-likelihood against caller-supplied tables (doublet step)
// ------------------------------------------------------------------------------------
extern "C" int vrx_problem_doublet(vrx_problem* p, int64_t n_donor, int64_t n_gt,
                                   const double* GT_prob, const double* psi1, const double* psi2,
                                   const double* psis, int64_t psi_rows, const double* ID_prior,
                                   int64_t id_rows, double* logLik, double* prob_out) {
    VRX_REQUIRE(p && GT_prob && psi1 && psi2 && psis && logLik, "vrx_problem_doublet: null argument");
    VRX_REQUIRE(n_donor >= 2 && n_gt >= 1 && n_gt <= 3,
                "vrx_problem_doublet: needs n_donor >= 2 and n_GT <= 3");
    VRX_REQUIRE(psi_rows == 1 || psi_rows == p->n_var, "vrx_problem_doublet: psi rows must be 1 or n_var");
    const int64_t C = n_donor + n_donor * (n_donor - 1) / 2;
    const int G = (int)(n_gt + n_gt * (n_gt - 1) / 2);
    VRX_REQUIRE(id_rows == 0 || id_rows == 1 || id_rows == p->n_cell,
                "vrx_problem_doublet: ID_prior must have 0, 1 or n_cell rows");
    VRX_REQUIRE(id_rows == 0 || ID_prior, "vrx_problem_doublet: null ID_prior");
    vrx_model_cfg cfg{};
    cfg.kind = VRX_KIND_VIREO;
    cfg.n_donor = (int32_t)C;
    cfg.n_gt = 1;  // the pair classes live in registers; no N x C x G tensor
    vrx_model* m = nullptr;
    int rc = vrx_model_create(p, &cfg, &m);
    if (rc) return rc;
    std::unique_ptr<vrx_model> guard(m);
    hipStream_t s = p->stream;
    DevBuf<double> gt, psi;
    DevBuf<int2> pairs;
    std::vector<int2> hp;
    for (int a = 0; a < n_donor; ++a)
        for (int b = a + 1; b < n_donor; ++b) hp.push_back(make_int2(a, b));  // combinations order
    const size_t n_gt_el = (size_t)(p->n_var * n_donor * n_gt), th = (size_t)(psi_rows * G);
    VRX_HIP(gt.upload(GT_prob, n_gt_el, s));
    VRX_HIP(pairs.upload(hp.data(), hp.size(), s));
    VRX_HIP(psi.alloc(3 * th));
    VRX_HIP(hipMemcpyAsync(psi.p, psi1, th * sizeof(double), hipMemcpyHostToDevice, s));
    VRX_HIP(hipMemcpyAsync(psi.p + th, psi2, th * sizeof(double), hipMemcpyHostToDevice, s));
    VRX_HIP(hipMemcpyAsync(psi.p + 2 * th, psis, th * sizeof(double), hipMemcpyHostToDevice, s));
    const int64_t n = p->n_var * C;
    vrx_doublet_w<<<(unsigned)((n + VRX_BLOCK - 1) / VRX_BLOCK), VRX_BLOCK, 0, s>>>(
        p->n_var, (int)n_donor, (int)n_gt, (int)C, psi_rows == 1 ? 0 : 1, gt.p, pairs.p, psi.p,
        psi.p + th, psi.p + 2 * th, m->W.p, m->wform);
    VRX_HIP(hipGetLastError());
    if ((rc = cell_pass(m))) return rc;
    if ((rc = d2h(m, logLik, m->LID, (size_t)(m->M * m->K)))) return rc;
    if (prob_out) {
        if (id_rows > 0) {
            if ((rc = upload_log_rows(m, m->logq_id, ID_prior, id_rows, m->K))) return rc;
            m->id_mode = id_rows == 1 ? 1 : 2;
        }
        if ((rc = softmax_step(m, 1))) return rc;
        if ((rc = d2h(m, prob_out, m->ID, (size_t)(m->M * m->K)))) return rc;
    }
    VRX_HIP(hipStreamSynchronize(s));
    return VRX_OK;
}

extern "C" int vrx_problem_donor_reads(vrx_problem* p, int64_t n_col, const double* ID_prob,
                                       double* AD_reads, double* DP_reads) {
    VRX_REQUIRE(p && ID_prob && AD_reads && DP_reads, "vrx_problem_donor_reads: null argument");
    VRX_REQUIRE(n_col >= 1, "vrx_problem_donor_reads: bad shape");
    vrx_model_cfg cfg{};
    cfg.kind = VRX_KIND_BMM;  // no genotype layer needed: only ID_prob and S
    cfg.n_donor = (int32_t)n_col;
    vrx_model* m = nullptr;
    int rc = vrx_model_create(p, &cfg, &m);
    if (rc) return rc;
    std::unique_ptr<vrx_model> guard(m);
    if ((rc = h2d(m, m->ID, ID_prob, (size_t)(m->M * m->K)))) return rc;
    if ((rc = variant_pass(m))) return rc;
    std::vector<double> S((size_t)m->NK * 2);
    if ((rc = d2h(m, S.data(), m->S, S.size()))) return rc;
    VRX_HIP(hipStreamSynchronize(p->stream));
    for (int64_t i = 0; i < m->NK; ++i) {
        AD_reads[i] = S[(size_t)i * 2];
        DP_reads[i] = S[(size_t)i * 2 + 1];
    }
    return VRX_OK;
}

extern "C" int vrx_problem_cell_loglik(vrx_problem* p, int64_t n_col, int64_t n_class,
                                       const double* GT, const double* psi1, const double* psi2,
                                       const double* psis, int64_t psi_rows,
                                       const double* ID_prior, int64_t id_rows, double* logLik,
                                       double* prob_out) {
    VRX_REQUIRE(p && GT && psi1 && psi2 && psis && logLik, "vrx_problem_cell_loglik: null argument");
    VRX_REQUIRE(id_rows == 0 || id_rows == 1 || id_rows == p->n_cell,
                "vrx_problem_cell_loglik: ID_prior must have 0, 1 or n_cell rows");
    VRX_REQUIRE(id_rows == 0 || ID_prior, "vrx_problem_cell_loglik: null ID_prior");
    VRX_REQUIRE(n_col >= 1 && n_class >= 1, "vrx_problem_cell_loglik: bad shape");
    if (n_class > VRX_MAXT) {
        vrx_set_error("vrx_problem_cell_loglik: %lld genotype classes unsupported (max %d)",
                      (long long)n_class, VRX_MAXT);
        return VRX_ERR_UNSUPPORTED;
    }
    VRX_REQUIRE(psi_rows == 1 || psi_rows == p->n_var, "vrx_problem_cell_loglik: psi rows must be 1 or n_var");
    vrx_model_cfg cfg{};
    cfg.kind = VRX_KIND_VIREO;
    cfg.n_donor = (int32_t)n_col;
    cfg.n_gt = (int32_t)n_class;
    cfg.learn_gt = 0;
    cfg.learn_theta = 0;
    cfg.ase_mode = psi_rows == 1 ? 0 : 1;
    vrx_model* m = nullptr;
    int rc = vrx_model_create(p, &cfg, &m);
    if (rc) return rc;
    std::unique_ptr<vrx_model> guard(m);
    hipStream_t s = p->stream;
    const size_t th = (size_t)(psi_rows * n_class);
    if ((rc = h2d(m, m->GT, GT, (size_t)m->NK * m->T))) return rc;
    VRX_HIP(hipMemcpyAsync(m->psi.p, psi1, th * sizeof(double), hipMemcpyHostToDevice, s));
    VRX_HIP(hipMemcpyAsync(m->psi.p + th, psi2, th * sizeof(double), hipMemcpyHostToDevice, s));
    VRX_HIP(hipMemcpyAsync(m->psi.p + 2 * th, psis, th * sizeof(double), hipMemcpyHostToDevice, s));
    if ((rc = gt_step(m, 0))) return rc;
    if ((rc = cell_pass(m))) return rc;
    if ((rc = d2h(m, logLik, m->LID, (size_t)(m->M * m->K)))) return rc;
    if (prob_out) {
        if (id_rows > 0) {
            if ((rc = upload_log_rows(m, m->logq_id, ID_prior, id_rows, m->K))) return rc;
            m->id_mode = id_rows == 1 ? 1 : 2;
        }
        if ((rc = softmax_step(m, 1))) return rc;
        if ((rc = d2h(m, prob_out, m->ID, (size_t)(m->M * m->K)))) return rc;
    }
    VRX_HIP(hipStreamSynchronize(s));
    return VRX_OK;
}
