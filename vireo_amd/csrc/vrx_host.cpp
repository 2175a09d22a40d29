// Host-only pieces of libvireo_hip.so (no device code).
//
// vrx_mt19937_random_sample: NumPy's legacy global stream, continued in C.
//   The reference draws every restart's initial state with np.random.rand from the global
//   RandomState (vireoSNP/utils/vireo_model.py:98,103 via vireo_wrap.py:66-71), so restart i's
//   state depends on how many doubles restarts 0..i-1 consumed.  When the restarts are sharded
//   over GPUs every rank still has to walk the whole stream; NumPy produces ~7 ns per double,
//   which at N=100k x M=50k x K=16 (5.6 M doubles per restart) costs more than the fits
//   themselves.  Here the stream is advanced by regenerating the 624-word Mersenne-Twister
//   state in its three data-parallel stages (auto-vectorised: ~0.2 ns per skipped double) and
//   doubles are only formed for the restarts a rank owns.
//   Algorithm restated from NumPy's public implementation (numpy/random/src/mt19937):
//   MT19937 (Matsumoto & Nishimura 1998) with the genrand_res53 conversion
//   (a >> 5, b >> 6) -> (a * 2^26 + b) / 2^53.
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/vireo_hip.h"
#include "vrx_balance.h"

// ------------------------------------------------------------------------------------
// errors: a thread-local message behind every non-zero status (include/vireo_hip.h)
// ------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

void vrx_set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

extern "C" const char* vrx_last_error(void) { return g_last_error.c_str(); }

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_A = 0x9908b0dfu, MT_UP = 0x80000000u, MT_LO = 0x7fffffffu;

inline uint32_t mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & MT_UP) | (v & MT_LO);
    return (y >> 1) ^ ((0u - (y & 1u)) & MT_A);
}

// one regeneration of the state, in place; every loop's reads are either of words the loop
// does not write or lie >= 227 words behind its writes, so the loops vectorise
// (compiled for AVX-512 / AVX2 / baseline x86-64 and picked at load time: the build flags
//  stay generic)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void mt_regen(uint32_t* __restrict__ k) {
    uint32_t nxt[MT_N + 1];
    std::memcpy(nxt, k, sizeof(uint32_t) * MT_N);
    for (int i = 0; i < MT_N - MT_M; ++i) k[i] = nxt[i + MT_M] ^ mt_twist(nxt[i], nxt[i + 1]);
    // k[i - 227] was written by the first loop / by this loop >= 227 iterations ago
    for (int i = MT_N - MT_M; i < 2 * (MT_N - MT_M); ++i)
        k[i] = k[i - (MT_N - MT_M)] ^ mt_twist(nxt[i], nxt[i + 1]);
    for (int i = 2 * (MT_N - MT_M); i < MT_N - 1; ++i)
        k[i] = k[i - (MT_N - MT_M)] ^ mt_twist(nxt[i], nxt[i + 1]);
    k[MT_N - 1] = k[MT_M - 1] ^ mt_twist(nxt[MT_N - 1], k[0]);
}

inline uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

inline double mt_double(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}
}  // namespace

// ------------------------------------------------------------------------------------
// Jump-ahead.  A rank of a restart shard has to step the generator past every other rank's
// restarts (c4 at 8 ranks: 28 of 32 restarts, ~0.3 G outputs).  The state transition is linear
// over GF(2): 624 * R steps are the polynomial g(x) = x^(624 R) mod phi(x) of the transition
// matrix T applied to the state, phi = T's characteristic polynomial (degree 19937; Haramoto,
// Matsumoto, Nishimura, Panneton, L'Ecuyer 2008).  phi comes from Berlekamp-Massey on one output
// bit of 2 x 19937 steps (once per process), g from square-and-multiply (cached per R), and
// g(T) s from Horner's rule on a sliding 624-word window: 19937 single steps + ~10^4 XORs of
// 624 words, ~2 ms whatever R is.
// ------------------------------------------------------------------------------------
namespace {
constexpr int MT_DEG = 19937, MT_PW = (MT_DEG + 64) / 64;  // words of a polynomial of degree <= 19937

struct MtPoly {
    uint64_t w[MT_PW];
    bool bit(int i) const { return (w[i >> 6] >> (i & 63)) & 1u; }
};

// one step of the recurrence on a circular window: the word at `head` leaves, a new one enters
inline void mt_step(uint32_t* win, int& head) {
    const int i1 = head + 1 == MT_N ? 0 : head + 1;
    int im = head + MT_M;
    if (im >= MT_N) im -= MT_N;
    win[head] = win[im] ^ mt_twist(win[head], win[i1]);
    head = i1;
}

// phi(x) = x^19937 + ... : minimal polynomial of the sequence (lowest bit of every raw word) of
// an arbitrary non-zero state, Berlekamp-Massey over GF(2)
const MtPoly& mt_char_poly() {
    static MtPoly phi;
    static std::once_flag once;
    std::call_once(once, [] {
        constexpr int NS = 2 * MT_DEG + 64, SW = (NS + 63) / 64;
        std::vector<uint64_t> seq((size_t)SW, 0);
        uint32_t win[MT_N];
        uint32_t x = 19650218u;
        for (int i = 0; i < MT_N; ++i) {
            win[i] = x;
            x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i + 1u;
        }
        int head = 0;
        for (int t = 0; t < NS; ++t) {
            mt_step(win, head);
            const int newest = head == 0 ? MT_N - 1 : head - 1;
            if (win[newest] & 1u) seq[(size_t)(t >> 6)] |= 1ull << (t & 63);
        }
        // C(x) connection polynomial, B(x) its previous version; bit i = coefficient of x^i
        constexpr int CW = (MT_DEG + 2 + 63) / 64 + 1;
        std::vector<uint64_t> Cp((size_t)CW, 0), Bp((size_t)CW, 0), Tp((size_t)CW, 0);
        // reversed sequence window so that the discrepancy is a word-wise AND + parity:
        // rev[j] holds seq bit (t - j) at bit j  ->  kept by shifting one bit per step
        std::vector<uint64_t> rev((size_t)CW, 0);
        Cp[0] = Bp[0] = 1;
        int L = 0, m = 1;
        for (int t = 0; t < NS; ++t) {
            // shift the reversed window left by one and insert seq[t] at bit 0
            uint64_t carry = (seq[(size_t)(t >> 6)] >> (t & 63)) & 1u;
            for (int j = 0; j < CW; ++j) {
                const uint64_t nc = rev[(size_t)j] >> 63;
                rev[(size_t)j] = (rev[(size_t)j] << 1) | carry;
                carry = nc;
            }
            uint64_t acc = 0;
            const int lw = (L >> 6) + 1;
            for (int j = 0; j < lw && j < CW; ++j) acc ^= Cp[(size_t)j] & rev[(size_t)j];
            // (bits above L of C are zero, so whole words may be used)
            if (__builtin_parityll(acc)) {
                Tp = Cp;
                const int ws = m >> 6, bs = m & 63;  // C ^= B << m
                for (int j = CW - 1; j >= ws; --j) {
                    uint64_t v = Bp[(size_t)(j - ws)] << bs;
                    if (bs && j - ws - 1 >= 0) v |= Bp[(size_t)(j - ws - 1)] >> (64 - bs);
                    Cp[(size_t)j] ^= v;
                }
                if (2 * L <= t) {
                    L = t + 1 - L;
                    Bp = Tp;
                    m = 1;
                } else {
                    ++m;
                }
            } else {
                ++m;
            }
        }
        // phi(x) = x^L C(1/x): coefficient of x^(L - i) = c_i   (L == 19937 for this generator)
        std::memset(phi.w, 0, sizeof phi.w);
        if (L == MT_DEG)
            for (int i = 0; i <= L; ++i)
                if ((Cp[(size_t)(i >> 6)] >> (i & 63)) & 1u) phi.w[(L - i) >> 6] |= 1ull << ((L - i) & 63);
    });
    return phi;
}

// r(x) (degree < 2 * 19937 + 64, PW2 words) reduced mod phi, result in r[0 .. MT_PW)
void mt_poly_reduce(uint64_t* r, int top_bit, const MtPoly& phi) {
    // phi shifted left by 0..63 bits, so that a set bit is cleared with word-aligned XORs
    static uint64_t sh[64][MT_PW + 1];
    static std::once_flag once;
    std::call_once(once, [&] {
        for (int s = 0; s < 64; ++s) {
            for (int j = 0; j <= MT_PW; ++j) sh[s][j] = 0;
            for (int j = 0; j < MT_PW; ++j) {
                sh[s][j] |= s ? phi.w[j] << s : phi.w[j];
                if (s) sh[s][j + 1] |= phi.w[j] >> (64 - s);
            }
        }
    });
    for (int b = top_bit; b >= MT_DEG; --b) {
        if (!((r[b >> 6] >> (b & 63)) & 1u)) continue;
        const int d = b - MT_DEG, ws = d >> 6, bs = d & 63;  // r ^= phi << d
        for (int j = 0; j <= MT_PW; ++j) r[ws + j] ^= sh[bs][j];
    }
}

// g(x) = x^(624 * regens) mod phi(x), cached per `regens` (a few entries: a job skips by one or
// two distinct lengths; the result is handed out by value so that trimming cannot dangle)
bool mt_jump_poly(uint64_t regens, MtPoly* res) {
    static std::mutex mu;
    static std::map<uint64_t, MtPoly> cache;
    constexpr size_t kCacheCap = 16;  // 2.5 KB each
    {
        std::lock_guard<std::mutex> lk(mu);
        auto it = cache.find(regens);
        if (it != cache.end()) {
            *res = it->second;
            return true;
        }
    }
    const MtPoly& phi = mt_char_poly();
    if (!phi.bit(MT_DEG)) return false;  // (Berlekamp-Massey did not find degree 19937)
    // E = 624 * regens, square-and-multiply from the top bit: g <- g^2 (bit spreading), g <- g x
    unsigned __int128 E = (unsigned __int128)regens * 624u;
    int top = 0;
    for (int b = 0; b < 100; ++b)
        if ((E >> b) & 1u) top = b;
    constexpr int PW2 = 2 * MT_PW + 2;
    std::vector<uint64_t> r((size_t)PW2 + MT_PW + 2, 0), g((size_t)MT_PW, 0);
    g[0] = 1;  // x^0
    auto spread = [](uint32_t v) {  // bits of v to the even bit positions of a 64-bit word
        uint64_t x = v;
        x = (x | (x << 16)) & 0x0000ffff0000ffffull;
        x = (x | (x << 8)) & 0x00ff00ff00ff00ffull;
        x = (x | (x << 4)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x << 2)) & 0x3333333333333333ull;
        x = (x | (x << 1)) & 0x5555555555555555ull;
        return x;
    };
    for (int b = top; b >= 0; --b) {
        std::fill(r.begin(), r.end(), 0);
        for (int j = 0; j < MT_PW; ++j) {
            r[(size_t)(2 * j)] = spread((uint32_t)g[(size_t)j]);
            r[(size_t)(2 * j + 1)] = spread((uint32_t)(g[(size_t)j] >> 32));
        }
        mt_poly_reduce(r.data(), 2 * MT_DEG, phi);
        if ((E >> b) & 1u) {  // times x
            uint64_t carry = 0;
            for (int j = 0; j <= MT_PW; ++j) {
                const uint64_t nc = r[(size_t)j] >> 63;
                r[(size_t)j] = (r[(size_t)j] << 1) | carry;
                carry = nc;
            }
            mt_poly_reduce(r.data(), MT_DEG, phi);
        }
        for (int j = 0; j < MT_PW; ++j) g[(size_t)j] = r[(size_t)j];
    }
    MtPoly out;
    for (int j = 0; j < MT_PW; ++j) out.w[j] = g[(size_t)j];
    std::lock_guard<std::mutex> lk(mu);
    if (cache.size() >= kCacheCap) cache.erase(cache.begin());
    cache.emplace(regens, out);
    *res = out;
    return true;
}

// key <- g(T) key  (the window 624 * regens steps further on; the low 31 bits of key[0] are
// not part of the state and come out undefined: the caller regenerates once more)
#if !defined(__HIP_DEVICE_COMPILE__) && defined(__x86_64__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void mt_apply_poly(const MtPoly& g, uint32_t* __restrict__ key) {
    uint32_t src[MT_N], win[MT_N];
    std::memcpy(src, key, sizeof src);
    int deg = MT_DEG - 1;
    while (deg >= 0 && !g.bit(deg)) --deg;
    if (deg < 0) {
        std::memset(key, 0, sizeof src);
        return;
    }
    std::memcpy(win, src, sizeof src);  // h = s   (the leading coefficient)
    int head = 0;
    for (int i = deg - 1; i >= 0; --i) {
        mt_step(win, head);  // h = T h
        if (g.bit(i)) {      // h ^= s, word j of s onto window position head + j
            const int n1 = MT_N - head;
            for (int j = 0; j < n1; ++j) win[head + j] ^= src[j];
            for (int j = n1; j < MT_N; ++j) win[j - n1] ^= src[j];
        }
    }
    for (int j = 0; j < MT_N; ++j) key[j] = win[head + j < MT_N ? head + j : head + j - MT_N];
}
}  // namespace

// 0 = off (always step), otherwise the smallest number of regenerations a skip jumps over
static int64_t mt_jump_threshold() {
    static const int64_t t = [] {
        const char* e = getenv("VIREO_MT_JUMP_MIN_REGENS");
        return e ? (int64_t)atoll(e) : (int64_t)20000;  // ~2.5 ms of stepping = one Horner pass
    }();
    return t;
}

// skip n doubles; thr = the fewest regenerations worth a jump (0: always step)
static int mt_skip(uint32_t* key624, int32_t* pos, int64_t n, int64_t thr) {
    int p = *pos;
    int64_t words = 2 * n;  // 32-bit outputs to consume
    {
        // far skips jump: all but the last two regenerations as ONE polynomial of the transition
        // matrix (the regenerations that follow rebuild the bits a jump leaves undefined)
        if (thr > 0 && words > 0) {
            // (the skip performs (p + words - 1) / 624 regenerations; the jump takes a count that
            //  depends on `words` alone, so that equal skips share one cached polynomial)
            const int64_t jump = words / MT_N - 2;
            if (jump >= thr) {
                MtPoly g;
                if (mt_jump_poly((uint64_t)jump, &g)) {
                    mt_apply_poly(g, key624);
                    words -= jump * (int64_t)MT_N;
                }
            }
        }
        while (words > 0) {
            if (p == MT_N) {
                mt_regen(key624);
                p = 0;
            }
            const int64_t take = words < MT_N - p ? words : MT_N - p;
            p += (int)take;
            words -= take;
        }
        *pos = p;
        return VRX_OK;
    }
}

extern "C" int vrx_mt19937_skip(uint32_t* key624, int32_t* pos, int64_t n, int32_t jump) {
    if (!key624 || !pos || n < 0 || *pos < 0 || *pos > MT_N) {
        vrx_set_error("vrx_mt19937_skip: bad argument");
        return VRX_ERR_ARG;
    }
    return mt_skip(key624, pos, n, jump < 0 ? mt_jump_threshold() : jump ? 1 : 0);
}

extern "C" int vrx_mt19937_random_sample(uint32_t* key624, int32_t* pos, double* out, int64_t n) {
    if (!key624 || !pos || n < 0 || *pos < 0 || *pos > MT_N) {
        vrx_set_error("vrx_mt19937_random_sample: bad argument");
        return VRX_ERR_ARG;
    }
    if (!out) return mt_skip(key624, pos, n, mt_jump_threshold());  // skip: only the state moves
    int p = *pos;
    int64_t done = 0;
    uint32_t t[MT_N];
    bool have_a = false;
    uint32_t a = 0;
    while (done < n) {
        if (p == MT_N) {
            mt_regen(key624);
            p = 0;
        }
        const int avail = MT_N - p;
        for (int i = 0; i < avail; ++i) t[i] = mt_temper(key624[p + i]);
        int i = 0;
        if (have_a && avail > 0) {  // a double straddling two regenerations
            out[done++] = mt_double(a, t[0]);
            have_a = false;
            i = 1;
        }
        const int64_t room = n - done;
        int64_t pairs = (avail - i) / 2;
        if (pairs > room) pairs = room;
        for (int64_t j = 0; j < pairs; ++j) out[done + j] = mt_double(t[i + 2 * j], t[i + 2 * j + 1]);
        done += pairs;
        i += (int)(2 * pairs);
        if (done < n && i < avail) {  // one word left in this state: first half of the next double
            a = t[i];
            have_a = true;
            ++i;
        }
        p += i;
    }
    *pos = p;
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// np.sum of a contiguous float32 array, bit for bit
// ------------------------------------------------------------------------------------
// NumPy's pairwise summation of one iterator chunk (blocks of <= 128 terms on eight running
// sums, halves cut at multiples of 8), in float32 -- how the reference's
// `np.sum(get_binom_coeff(AD, DP))` (vireo_model.py:313, bmm_model.py:239) adds its terms
// (restated from numpy/_core/src/umath/loops_utils.h.src; pinned by tests/test_host_cpu.py
// against np.sum itself and by the golden constants)
static float np_pairwise_sum_f32(const float* a, int64_t n) {
    if (n < 8) {
        float r = 0.f;
        for (int64_t i = 0; i < n; ++i) r += a[i];
        return r;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int64_t i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}


// np.sum walks a 1-D array through its iterator buffer, 8192 elements at a time
// (np.getbufsize()): every chunk is summed pairwise, the chunk sums are added in order.
extern "C" int vrx_np_sum_f32(const float* a, int64_t n, float* out) {
    if ((!a && n > 0) || n < 0 || !out) {
        vrx_set_error("vrx_np_sum_f32: bad argument");
        return VRX_ERR_ARG;
    }
    float total = 0.f;
    for (int64_t at = 0; at < n; at += 8192)
        total += np_pairwise_sum_f32(a + at, n - at < 8192 ? n - at : 8192);
    *out = total;
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// MatrixMarket coordinate files (cellSNP.tag.AD.mtx / .DP.mtx, vartrix alt / ref matrices)
// ------------------------------------------------------------------------------------
// The reference loads them with scipy.io.mmread (vireoSNP/utils/io_utils.py:57,72-73), a
// single-threaded text parser that takes minutes at 1e8 entries.  Here the file is mapped, cut
// into one piece per thread at line boundaries, the lines of every piece are counted, and then
// every thread parses its piece straight into its slice of the caller's COO arrays.  Format
// (NIST Matrix Market): a `%%MatrixMarket matrix coordinate <field> <symmetry>` banner, `%`
// comment lines, one `rows cols entries` line, then `row col [value]` lines, 1-based.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

namespace {
struct MtxMap {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~MtxMap() {
        if (p) munmap(const_cast<char*>(p), n);
        if (fd >= 0) close(fd);
    }
    bool open_file(const char* path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) return true;
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = static_cast<const char*>(m);
        madvise(m, n, MADV_SEQUENTIAL);
        return true;
    }
};

struct MtxHead {
    int64_t rows = 0, cols = 0, nnz = 0;
    int field = 0;      // 0 integer, 1 real, 2 pattern
    size_t body = 0;    // offset of the first entry line
};

// banner, comments and the size line; false on anything this reader does not handle
bool mtx_head(const MtxMap& f, MtxHead& h, std::string& why) {
    size_t at = 0;
    auto line_end = [&](size_t from) {
        const void* e = memchr(f.p + from, '\n', f.n - from);
        return e ? (size_t)(static_cast<const char*>(e) - f.p) : f.n;
    };
    if (f.n < 15 || std::memcmp(f.p, "%%MatrixMarket", 14) != 0) {
        why = "no %%MatrixMarket banner";
        return false;
    }
    size_t e = line_end(0);
    std::string banner(f.p, e);
    for (auto& c : banner) c = (char)tolower((unsigned char)c);
    if (banner.find("coordinate") == std::string::npos) {
        why = "not a coordinate (sparse) matrix";
        return false;
    }
    if (banner.find("general") == std::string::npos) {
        why = "symmetric / skew / hermitian storage is not supported";
        return false;
    }
    if (banner.find("integer") != std::string::npos)
        h.field = 0;
    else if (banner.find("real") != std::string::npos)
        h.field = 1;
    else if (banner.find("pattern") != std::string::npos)
        h.field = 2;
    else {
        why = "unsupported field (complex)";
        return false;
    }
    at = e + 1;
    while (at < f.n && (f.p[at] == '%' || f.p[at] == '\n' || f.p[at] == '\r')) at = line_end(at) + 1;
    if (at >= f.n) {
        why = "no size line";
        return false;
    }
    e = line_end(at);
    std::string size_line(f.p + at, e - at);
    long long r = 0, c = 0, z = 0;
    if (sscanf(size_line.c_str(), "%lld %lld %lld", &r, &c, &z) != 3 || r < 0 || c < 0 || z < 0) {
        why = "bad size line";
        return false;
    }
    h.rows = r;
    h.cols = c;
    h.nnz = z;
    h.body = e + 1 <= f.n ? e + 1 : f.n;
    return true;
}

inline const char* skip_ws(const char* p, const char* e) {
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}

inline const char* parse_i64(const char* p, const char* e, int64_t& out, bool& ok) {
    p = skip_ws(p, e);
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) neg = *p++ == '-';
    if (p >= e || *p < '0' || *p > '9') {
        ok = false;
        return p;
    }
    // (more digits than an int64 holds: not a count / an index of this format -- reject instead
    //  of overflowing; found by the UBSan run, tests/test_host_sanitizers_cpu.py)
    uint64_t v = 0;
    while (p < e && *p >= '0' && *p <= '9') {
        const uint64_t d = (uint64_t)(*p++ - '0');
        if (v > ((uint64_t)INT64_MAX - d) / 10) {
            ok = false;
            return p;
        }
        v = v * 10 + d;
    }
    out = neg ? -(int64_t)v : (int64_t)v;
    return p;
}
}  // namespace

extern "C" int vrx_mtx_header(const char* path, int64_t* n_rows, int64_t* n_cols, int64_t* nnz) {
    if (!path || !n_rows || !n_cols || !nnz) {
        vrx_set_error("vrx_mtx_header: null argument");
        return VRX_ERR_ARG;
    }
    MtxMap f;
    MtxHead h;
    std::string why;
    if (!f.open_file(path)) {
        vrx_set_error("vrx_mtx_header: cannot read %s", path);
        return VRX_ERR_ARG;
    }
    if (!mtx_head(f, h, why)) {
        vrx_set_error("vrx_mtx_header: %s: %s", path, why.c_str());
        return VRX_ERR_UNSUPPORTED;
    }
    *n_rows = h.rows;
    *n_cols = h.cols;
    *nnz = h.nnz;
    return VRX_OK;
}

extern "C" int vrx_mtx_read(const char* path, int64_t nnz, int32_t* row, int32_t* col, int32_t* val,
                            int n_threads) {
    if (!path || nnz < 0 || (nnz > 0 && (!row || !col || !val))) {
        vrx_set_error("vrx_mtx_read: bad argument");
        return VRX_ERR_ARG;
    }
    MtxMap f;
    MtxHead h;
    std::string why;
    if (!f.open_file(path)) {
        vrx_set_error("vrx_mtx_read: cannot read %s", path);
        return VRX_ERR_ARG;
    }
    if (!mtx_head(f, h, why)) {
        vrx_set_error("vrx_mtx_read: %s: %s", path, why.c_str());
        return VRX_ERR_UNSUPPORTED;
    }
    if (h.nnz != nnz) {
        vrx_set_error("vrx_mtx_read: %s holds %lld entries, caller expects %lld", path,
                      (long long)h.nnz, (long long)nnz);
        return VRX_ERR_ARG;
    }
    if (h.rows >= INT32_MAX || h.cols >= INT32_MAX) {
        vrx_set_error("vrx_mtx_read: dimension >= 2^31");
        return VRX_ERR_UNSUPPORTED;
    }
    const char* const body = f.p + h.body;
    const char* const end = f.p + f.n;
    int T = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min(T, 64), (end - body) / (1 << 16) + 1));
    // piece boundaries at line starts
    std::vector<const char*> cut((size_t)T + 1, end);
    cut[0] = body;
    for (int t = 1; t < T; ++t) {
        const char* c = body + (size_t)(end - body) * t / T;
        const void* nl = c < end ? memchr(c, '\n', (size_t)(end - c)) : nullptr;
        cut[(size_t)t] = nl ? static_cast<const char*>(nl) + 1 : end;
        if (cut[(size_t)t] < cut[(size_t)t - 1]) cut[(size_t)t] = cut[(size_t)t - 1];
    }
    auto is_entry_line = [](const char* p, const char* e) {
        p = skip_ws(p, e);
        return p < e && *p != '%';
    };
    // pass 1: entry lines per piece
    std::vector<int64_t> count((size_t)T, 0);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t] {
                const char* p = cut[(size_t)t];
                const char* e = cut[(size_t)t + 1];
                int64_t c = 0;
                while (p < e) {
                    const void* nl = memchr(p, '\n', (size_t)(e - p));
                    const char* le = nl ? static_cast<const char*>(nl) : e;
                    if (is_entry_line(p, le)) ++c;
                    p = le + 1;
                }
                count[(size_t)t] = c;
            });
        for (auto& th : pool) th.join();
    }
    std::vector<int64_t> first((size_t)T + 1, 0);
    for (int t = 0; t < T; ++t) first[(size_t)t + 1] = first[(size_t)t] + count[(size_t)t];
    if (first[(size_t)T] != nnz) {
        vrx_set_error("vrx_mtx_read: %s declares %lld entries but holds %lld entry lines", path,
                      (long long)nnz, (long long)first[(size_t)T]);
        return VRX_ERR_ARG;
    }
    // pass 2: parse
    std::vector<int64_t> bad((size_t)T, -1);
    {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t)
            pool.emplace_back([&, t] {
                const char* p = cut[(size_t)t];
                const char* e = cut[(size_t)t + 1];
                int64_t at = first[(size_t)t];
                while (p < e) {
                    const void* nl = memchr(p, '\n', (size_t)(e - p));
                    const char* le = nl ? static_cast<const char*>(nl) : e;
                    if (is_entry_line(p, le)) {
                        bool ok = true;
                        int64_t r = 0, c = 0, v = 1;
                        const char* q = parse_i64(p, le, r, ok);
                        q = parse_i64(q, le, c, ok);
                        if (ok && h.field == 0) {
                            q = parse_i64(q, le, v, ok);
                        } else if (ok && h.field == 1) {  // real: integral counts written as floats
                            q = skip_ws(q, le);
                            char buf[64];
                            const size_t len = std::min<size_t>(sizeof buf - 1, (size_t)(le - q));
                            std::memcpy(buf, q, len);
                            buf[len] = 0;
                            char* stop = nullptr;
                            const double d = strtod(buf, &stop);
                            // (range first: converting a double outside int64 is undefined)
                            ok = stop != buf && d >= -2147483648.0 && d <= 2147483647.0;
                            v = ok ? (int64_t)d : 0;
                            ok = ok && (double)v == d;
                        }
                        if (!ok || r < 1 || r > h.rows || c < 1 || c > h.cols || v < INT32_MIN ||
                            v > INT32_MAX) {
                            bad[(size_t)t] = at;
                            return;
                        }
                        row[at] = (int32_t)(r - 1);
                        col[at] = (int32_t)(c - 1);
                        val[at] = (int32_t)v;
                        ++at;
                    }
                    p = le + 1;
                }
            });
        for (auto& th : pool) th.join();
    }
    for (int t = 0; t < T; ++t)
        if (bad[(size_t)t] >= 0) {
            vrx_set_error("vrx_mtx_read: %s: malformed, out-of-range or non-integral entry (entry %lld)",
                          path, (long long)bad[(size_t)t]);
            return VRX_ERR_ARG;
        }
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// (AD, DP) -> one CSC pattern with an (ad, dp) pair per entry
// ------------------------------------------------------------------------------------
// The union of the two patterns, column by column (both inputs in canonical CSC form: row indices
// strictly increasing inside a column), entries where both counts are zero dropped.  Replaces a
// SciPy sparse addition on packed values (1.0 s at 1e8 entries) by a two-pointer merge per
// column over all cores.  Index arrays int32 or int64, count arrays int32 / int64 / float64
// (the reference's loaders produce int64 from MatrixMarket and float64 from VCF input:
// io_utils.py:57, vcf_utils.py:204).
namespace {
struct CscView {
    const void *ptr, *idx, *dat;
    int ptr64, idx64, dkind;  // dkind: 0 int32, 1 int64, 2 float64
    int64_t p(int64_t c) const {
        return ptr64 ? static_cast<const int64_t*>(ptr)[c] : static_cast<const int32_t*>(ptr)[c];
    }
    int64_t i(int64_t e) const {
        return idx64 ? static_cast<const int64_t*>(idx)[e] : static_cast<const int32_t*>(idx)[e];
    }
    // the count of entry e; ok = false if it is negative, not integral or >= 2^31
    int32_t v(int64_t e, bool& ok) const {
        if (dkind == 0) {
            const int32_t x = static_cast<const int32_t*>(dat)[e];
            ok &= x >= 0;
            return x;
        }
        if (dkind == 1) {
            const int64_t x = static_cast<const int64_t*>(dat)[e];
            ok &= x >= 0 && x <= INT32_MAX;
            return (int32_t)x;
        }
        const double x = static_cast<const double*>(dat)[e];
        ok &= x >= 0.0 && x <= (double)INT32_MAX && x == (double)(int64_t)x;
        return (int32_t)x;
    }
};

template <class F>
void over_columns(int64_t n_col, int n_threads, F&& f) {
    int T = n_threads > 0 ? n_threads : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, n_col));
    std::vector<std::thread> pool;
    for (int t = 0; t < T; ++t)
        pool.emplace_back([&, t] { f(n_col * t / T, n_col * (t + 1) / T, t); });
    for (auto& th : pool) th.join();
}
}  // namespace

// pass 1 (rowidx == NULL): colptr[c + 1] = entries of column c of the union (caller zeroes
// colptr[0] and turns the counts into offsets); pass 2: fills rowidx / ad / dp at colptr.
// Returns VRX_ERR_ARG for unsorted / duplicate row indices or counts that are not
// non-negative integers below 2^31 (thread-local message names the column).
extern "C" int vrx_merge_counts(int64_t n_var, int64_t n_cell, const void* ad_ptr, const void* ad_idx,
                                const void* ad_dat, int ad_ptr64, int ad_idx64, int ad_kind,
                                const void* dp_ptr, const void* dp_idx, const void* dp_dat,
                                int dp_ptr64, int dp_idx64, int dp_kind, int64_t* colptr,
                                int32_t* rowidx, int32_t* ad, int32_t* dp, int n_threads) {
    if (n_var < 0 || n_cell < 0 || !ad_ptr || !dp_ptr || !colptr || ad_kind < 0 || ad_kind > 2 ||
        dp_kind < 0 || dp_kind > 2) {
        vrx_set_error("vrx_merge_counts: bad argument");
        return VRX_ERR_ARG;
    }
    const CscView A{ad_ptr, ad_idx, ad_dat, ad_ptr64, ad_idx64, ad_kind};
    const CscView D{dp_ptr, dp_idx, dp_dat, dp_ptr64, dp_idx64, dp_kind};
    const bool fill = rowidx != nullptr;
    std::vector<int64_t> bad(256, -1);
    over_columns(n_cell, n_threads, [&](int64_t c0, int64_t c1, int tid) {
        for (int64_t c = c0; c < c1; ++c) {
            int64_t a = A.p(c), ae = A.p(c + 1), d = D.p(c), de = D.p(c + 1);
            int64_t out = fill ? colptr[c] : 0, n = 0;
            int64_t prev = -1;
            bool ok = ae >= a && de >= d;
            while (ok && (a < ae || d < de)) {
                const int64_t ra = a < ae ? A.i(a) : INT64_MAX, rd = d < de ? D.i(d) : INT64_MAX;
                const int64_t r = ra < rd ? ra : rd;
                ok &= r > prev && r < n_var;
                prev = r;
                int32_t va = 0, vd = 0;
                if (ra == r) va = A.v(a++, ok);
                if (rd == r) vd = D.v(d++, ok);
                if (va != 0 || vd != 0) {
                    if (fill) {
                        rowidx[out] = (int32_t)r;
                        ad[out] = va;
                        dp[out] = vd;
                        ++out;
                    }
                    ++n;
                }
            }
            if (!ok) {
                if (tid < 256) bad[(size_t)tid] = c;
                return;
            }
            if (!fill) colptr[c + 1] = n;
        }
    });
    for (int64_t c : bad)
        if (c >= 0) {
            vrx_set_error("vrx_merge_counts: column %lld: row indices not strictly increasing / out of "
                          "range, or a count that is not a non-negative integer below 2^31",
                          (long long)c);
            return VRX_ERR_ARG;
        }
    return VRX_OK;
}

// COO (file order of a MatrixMarket file) -> CSC, what `mmread(...).tocsc()` does in read_cellSNP /
// read_vartrix (vireoSNP/utils/io_utils.py:57,72-73) with SciPy's single-threaded coo_tocsr: at
// 1e8 entries that conversion, twice, was the longest phase of the whole command.  A stable
// counting sort by column on all cores: per-thread column histograms over contiguous chunks of
// the entries, one pass over the columns to turn them into write offsets, an ordered scatter.
// Entries keep their file order inside a column; *canonical = 1 when the rows of every column
// come out strictly increasing (cellSNP writes variant-major files: always) -- otherwise the
// caller lets SciPy sort and sum the duplicates, as the reference would.  data is int64 like
// mmread's `integer` matrices.
extern "C" int vrx_coo_to_csc(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t* row,
                              const int32_t* col, const int32_t* val, int64_t* colptr,
                              int32_t* rowidx, int64_t* data, int32_t* canonical, int n_threads) {
    if (n_rows < 0 || n_cols < 0 || nnz < 0 || !colptr || !canonical ||
        (nnz > 0 && (!row || !col || !val || !rowidx || !data))) {
        vrx_set_error("vrx_coo_to_csc: bad argument");
        return VRX_ERR_ARG;
    }
    int T = n_threads > 0 ? n_threads : (int)std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    T = (int)std::max<int64_t>(1, std::min<int64_t>(T, nnz / 65536 + 1));
    std::vector<std::vector<int64_t>> hist((size_t)T, std::vector<int64_t>((size_t)n_cols, 0));
    std::vector<char> bad((size_t)T, 0);
    auto over_entries = [&](auto&& f) {
        std::vector<std::thread> pool;
        for (int t = 0; t < T; ++t) pool.emplace_back([&, t] { f(nnz * t / T, nnz * (t + 1) / T, t); });
        for (auto& th : pool) th.join();
    };
    over_entries([&](int64_t e0, int64_t e1, int t) {
        int64_t* h = hist[(size_t)t].data();
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t c = col[e], r = row[e];
            if (c < 0 || c >= n_cols || r < 0 || r >= n_rows) {
                bad[(size_t)t] = 1;
                return;
            }
            ++h[c];
        }
    });
    for (char b : bad)
        if (b) {
            vrx_set_error("vrx_coo_to_csc: an index outside the %lld x %lld matrix", (long long)n_rows,
                          (long long)n_cols);
            return VRX_ERR_ARG;
        }
    int64_t base = 0;
    colptr[0] = 0;
    for (int64_t c = 0; c < n_cols; ++c) {  // thread t's entries of column c start at hist[t][c]
        for (int t = 0; t < T; ++t) {
            const int64_t n = hist[(size_t)t][(size_t)c];
            hist[(size_t)t][(size_t)c] = base;
            base += n;
        }
        colptr[c + 1] = base;
    }
    over_entries([&](int64_t e0, int64_t e1, int t) {
        int64_t* h = hist[(size_t)t].data();
        for (int64_t e = e0; e < e1; ++e) {
            const int64_t at = h[col[e]]++;
            rowidx[at] = row[e];
            data[at] = val[e];
        }
    });
    std::vector<char> loose(256, 0);
    over_columns(n_cols, n_threads, [&](int64_t c0, int64_t c1, int tid) {
        for (int64_t c = c0; c < c1; ++c)
            for (int64_t e = colptr[c] + 1; e < colptr[c + 1]; ++e)
                if (rowidx[e] <= rowidx[e - 1]) {
                    loose[(size_t)(tid & 255)] = 1;
                    return;
                }
    });
    *canonical = 1;
    for (char b : loose)
        if (b) *canonical = 0;
    return VRX_OK;
}

// ------------------------------------------------------------------------------------
// Text writers of the command (vireoSNP/utils/io_utils.py:147-170 prob_singlet.tsv /
// prob_doublet.tsv; vcf_utils.py:234-296 the donor genotype VCF).  The reference formats every
// number with a Python "%" expression inside nested loops; at 50 k cells x 120 donor pairs or
// 100 k variants x 16 donors that is where a run spends its time once the fit takes a fraction
// of a second.  Here rows are cut into chunks, every chunk is formatted (and deflated, each
// chunk one gzip member: concatenated members are one valid gzip file) on its own thread, and
// the chunks are written in order.  Same bytes as the reference's text, number for number:
// "%.2e" of glibc and of CPython are both correctly rounded.
// ------------------------------------------------------------------------------------
#include <zlib.h>

#include <functional>

namespace {
int writer_threads() {
    const char* v = getenv("VIREO_HOST_THREADS");
    int n = v && *v ? atoi(v) : (int)std::thread::hardware_concurrency();
    return std::max(1, std::min(n, 32));
}

bool gzip_member(const std::string& in, std::string& out) {
    z_stream zs;
    std::memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK) return false;
    out.resize(deflateBound(&zs, (uLong)in.size()));
    zs.next_in = reinterpret_cast<Bytef*>(const_cast<char*>(in.data()));
    zs.avail_in = (uInt)in.size();
    zs.next_out = reinterpret_cast<Bytef*>(&out[0]);
    zs.avail_out = (uInt)out.size();
    const int rc = deflate(&zs, Z_FINISH);
    const uLong n = zs.total_out;
    deflateEnd(&zs);
    if (rc != Z_STREAM_END) return false;
    out.resize(n);
    return true;
}

// rows [0, n_rows) formatted by `line(row, text)` (appends one or more complete lines) in
// chunks of at most ~64 MB of text; `head` goes in front of the first chunk
int write_rows(const char* path, const std::string& head, int64_t n_rows, int64_t bytes_per_row,
               bool gz, const std::function<void(int64_t, std::string&)>& line) {
    const char* cb = getenv("VIREO_WRITER_CHUNK_BYTES");  // (tests shrink the chunks)
    const int64_t chunk_bytes = cb && *cb ? std::max<int64_t>(1, atoll(cb)) : (64ll << 20);
    // chunks of at most chunk_bytes of text, and enough of them to keep every thread busy
    // (but not below ~256 KB of text each)
    int64_t per_chunk = std::max<int64_t>(1, chunk_bytes / std::max<int64_t>(bytes_per_row, 1));
    const int64_t spread = (n_rows + 4 * writer_threads() - 1) / (4 * writer_threads());
    const int64_t floor_rows = std::max<int64_t>(1, std::min<int64_t>(chunk_bytes, 256 << 10) /
                                                        std::max<int64_t>(bytes_per_row, 1));
    per_chunk = std::max<int64_t>(1, std::min(per_chunk, std::max(spread, floor_rows)));
    const int64_t n_chunk = std::max<int64_t>(1, (n_rows + per_chunk - 1) / per_chunk);
    FILE* f = std::fopen(path, "wb");
    if (!f) {
        vrx_set_error("cannot open %s for writing", path);
        return VRX_ERR_ARG;
    }
    const int nt = (int)std::min<int64_t>(writer_threads(), n_chunk);
    bool ok = true;
    for (int64_t c0 = 0; c0 < n_chunk && ok; c0 += nt) {  // one wave of chunks at a time, in order
        const int64_t nc = std::min<int64_t>(nt, n_chunk - c0);
        std::vector<std::string> out((size_t)nc);
        std::vector<char> good((size_t)nc, 1);
        std::vector<std::thread> th;
        for (int64_t j = 0; j < nc; ++j)
            th.emplace_back([&, j] {
                const int64_t c = c0 + j, lo = c * per_chunk, hi = std::min(n_rows, lo + per_chunk);
                std::string text;
                text.reserve((size_t)((hi - lo) * bytes_per_row + head.size() + 64));
                if (c == 0) text = head;
                for (int64_t r = lo; r < hi; ++r) line(r, text);
                if (gz)
                    good[(size_t)j] = gzip_member(text, out[(size_t)j]);
                else
                    out[(size_t)j].swap(text);
            });
        for (auto& t : th) t.join();
        for (int64_t j = 0; j < nc && ok; ++j)
            ok = good[(size_t)j] && std::fwrite(out[(size_t)j].data(), 1, out[(size_t)j].size(), f) ==
                                        out[(size_t)j].size();
    }
    if (std::fclose(f) != 0) ok = false;
    if (!ok) {
        vrx_set_error("writing %s failed", path);
        return VRX_ERR_ARG;
    }
    return VRX_OK;
}

inline void put_int(std::string& s, int64_t v) {
    char b[24];
    const int n = std::snprintf(b, sizeof b, "%lld", (long long)v);
    s.append(b, (size_t)n);
}
}  // namespace

// header + one line per row: names[row] and the row's `cols` numbers in `fmt` (a printf format of
// one double, e.g. "%.2e"), tab separated.  names: the concatenated row labels, name_off[row]
// .. name_off[row + 1] each.  gz != 0: `path` is written as gzip.
extern "C" int vrx_write_table(const char* path, const char* header, const char* names,
                               const int64_t* name_off, const double* table, int64_t rows,
                               int64_t cols, const char* fmt, int32_t gz) {
    if (!path || !header || !names || !name_off || (!table && rows * cols > 0) || rows < 0 ||
        cols < 0 || !fmt) {
        vrx_set_error("vrx_write_table: bad argument");
        return VRX_ERR_ARG;
    }
    const std::string f(fmt);
    return write_rows(path, header, rows, 24 + 10 * cols, gz != 0, [&](int64_t r, std::string& s) {
        s.append(names + name_off[r], (size_t)(name_off[r + 1] - name_off[r]));
        char b[64];
        for (int64_t c = 0; c < cols; ++c) {
            const int n = std::snprintf(b, sizeof b, f.c_str(), table[r * cols + c]);
            s.push_back('\t');
            s.append(b, (size_t)std::min<int>(n, (int)sizeof b - 1));
        }
        s.push_back('\n');
    });
}

// The records of the donor genotype VCF (vcf_utils.py:283-290 with the tags GT:AD:DP:PL):
// head (comment and #CHROM lines), then per variant its prefix (the eight fixed columns and the
// FORMAT column, prefix_off[i] .. prefix_off[i + 1]) and per sample
// <0/0|1/0|1/1>:<AD>:<DP>:<PL,PL,PL>.  call [n_var][n_sample] in 0..2, ad / dp [n_var][n_sample],
// pl [n_var][n_sample][3], already rounded to integers.
extern "C" int vrx_write_vcf_records(const char* path, const char* head, const char* prefix,
                                     const int64_t* prefix_off, const int8_t* call,
                                     const int64_t* ad, const int64_t* dp, const int64_t* pl,
                                     int64_t n_var, int64_t n_sample, int32_t gz) {
    if (!path || !head || !prefix || !prefix_off || n_var < 0 || n_sample < 0 ||
        (n_var * n_sample > 0 && (!call || !ad || !dp || !pl))) {
        vrx_set_error("vrx_write_vcf_records: bad argument");
        return VRX_ERR_ARG;
    }
    static const char* const gt_name[3] = {"0/0", "1/0", "1/1"};
    for (int64_t i = 0; i < n_var * n_sample; ++i)
        if (call[i] < 0 || call[i] > 2) {
            vrx_set_error("vrx_write_vcf_records: genotype call outside 0..2");
            return VRX_ERR_ARG;
        }
    return write_rows(path, head, n_var, 64 + 24 * n_sample, gz != 0, [&](int64_t i, std::string& s) {
        s.append(prefix + prefix_off[i], (size_t)(prefix_off[i + 1] - prefix_off[i]));
        for (int64_t d = 0; d < n_sample; ++d) {
            const int64_t j = i * n_sample + d;
            s.push_back('\t');
            s.append(gt_name[call[j]], 3);
            s.push_back(':');
            put_int(s, ad[j]);
            s.push_back(':');
            put_int(s, dp[j]);
            s.push_back(':');
            put_int(s, pl[j * 3]);
            s.push_back(',');
            put_int(s, pl[j * 3 + 1]);
            s.push_back(',');
            put_int(s, pl[j * 3 + 2]);
        }
        s.push_back('\n');
    });
}

// A `coordinate integer general` MatrixMarket file from 0-based COO arrays, in the given order:
// the inverse of vrx_mtx_read (cellSNP writes such files; the reference only reads them,
// io_utils.py:57).  Used by bench.py's end-to-end leg (1e8 entries: NumPy's savetxt needs minutes)
// and by the parser's round-trip test.
extern "C" int vrx_mtx_write(const char* path, int64_t n_rows, int64_t n_cols, int64_t nnz,
                             const int32_t* row, const int32_t* col, const int32_t* val) {
    if (!path || n_rows < 0 || n_cols < 0 || nnz < 0 || (nnz > 0 && (!row || !col || !val))) {
        vrx_set_error("vrx_mtx_write: bad argument");
        return VRX_ERR_ARG;
    }
    std::string head = "%%MatrixMarket matrix coordinate integer general\n%\n";
    put_int(head, n_rows);
    head.push_back(' ');
    put_int(head, n_cols);
    head.push_back(' ');
    put_int(head, nnz);
    head.push_back('\n');
    return write_rows(path, head.c_str(), nnz, 24, false, [&](int64_t e, std::string& s) {
        put_int(s, (int64_t)row[e] + 1);
        s.push_back(' ');
        put_int(s, (int64_t)col[e] + 1);
        s.push_back(' ');
        put_int(s, (int64_t)val[e]);
        s.push_back('\n');
    });
}


// ------------------------------------------------------------------------------------
// Balanced slabs (r6; TiledStream::perm, vrx_problem_create2 + VRX_PROBLEM_BALANCED): which contracted
// rows share a slab of an LDS-resident pass, for ONE row tile.
// The padding of a round is the maximum over its 16 rows of their words in ONE slab; which contracted
// rows share a slab is free per tile.  Greedy: the contracted rows ("columns" of the tile's sub-matrix)
// most-covered first, each to the slab -- among those with room -- where the sum of the present loads of
// the tile rows it touches is smallest (ties: the lowest slab); columns without an entry in the tile fill
// what is left.  Counted on the c3 matrix: 1.62 -> 1.17 executed slots per word.  Deterministic (the result
// is part of the stream): the int16 and int32 score loops and their AVX2 clones add the same integers.
// The reference has no counterpart (it multiplies SciPy's CSC / CSR as they are, vireo_model.py:167-196).
// ------------------------------------------------------------------------------------
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(__HIP_DEVICE_COMPILE__)  // (host code only)
#define VRX_SIMD_CLONES __attribute__((target_clones("avx2", "default")))
#else
#define VRX_SIMD_CLONES
#endif

// score[sl] = sum over the column's rows of load[row][w0 + sl], then the first slab with room and the smallest
// score (-1: none has room).  The sums are formed a block of slabs at a time with the block in registers
// (the rows of `load` are read once per block: the matrix is L2-resident, the accumulators never leave the
// registers); `full` holds all-ones for a slab without room, OR-ed in before the minimum.
// An entry of a column is one record: row inside the tile in the low RB bits, its word count above.
template <class ACC, int B, class E, int RB>
static inline __attribute__((always_inline)) void score_block(const int16_t* load, int nsp, const E* ent,
                                                              uint32_t n, int at, ACC* sc) {
    ACC acc[B];
    for (int k = 0; k < B; ++k) acc[k] = 0;
    for (uint32_t e = 0; e < n; ++e) {
        const int16_t* L = load + (size_t)(ent[e] & ((1u << RB) - 1)) * nsp + at;
        for (int k = 0; k < B; ++k) acc[k] = (ACC)(acc[k] + L[k]);
    }
    for (int k = 0; k < B; ++k) sc[k] = acc[k];
}

template <class ACC, class E, int RB>
static inline __attribute__((always_inline)) int best_slab(const int16_t* load, int nsp, const E* ent,
                                                           uint32_t n, int w0, int wn, const ACC* full, ACC* sc) {
    int sl = 0;
    for (; sl + 128 <= wn; sl += 128) score_block<ACC, 128, E, RB>(load, nsp, ent, n, w0 + sl, sc + sl);
    for (; sl + 64 <= wn; sl += 64) score_block<ACC, 64, E, RB>(load, nsp, ent, n, w0 + sl, sc + sl);
    for (; sl + 32 <= wn; sl += 32) score_block<ACC, 32, E, RB>(load, nsp, ent, n, w0 + sl, sc + sl);
    for (; sl + 8 <= wn; sl += 8) score_block<ACC, 8, E, RB>(load, nsp, ent, n, w0 + sl, sc + sl);
    for (; sl < wn; ++sl) {
        ACC a = 0;
        for (uint32_t e = 0; e < n; ++e) a = (ACC)(a + load[(size_t)(ent[e] & ((1u << RB) - 1)) * nsp + w0 + sl]);
        sc[sl] = a;
    }
    const ACC none = (ACC)(sizeof(ACC) == 2 ? 0x7fff : 0x7fffffff);
    ACC m = none;
    for (int k = 0; k < wn; ++k) {
        sc[k] = (ACC)(sc[k] | full[w0 + k]);
        m = sc[k] < m ? sc[k] : m;
    }
    if (m == none) return -1;
    for (int k = 0; k < wn; ++k)
        if (sc[k] == m) return w0 + k;
    return -1;
}

#define VRX_BEST_SLAB(NAME, ACC, E, RB)                                                                        \
    VRX_SIMD_CLONES static int NAME(const int16_t* load, int nsp, const E* ent, uint32_t n, int w0, int wn,        \
                                    const ACC* full, ACC* sc) {                                                    \
        return best_slab<ACC, E, RB>(load, nsp, ent, n, w0, wn, full, sc);                                         \
    }
VRX_BEST_SLAB(best_slab_i16_e16, int16_t, uint16_t, 11)
VRX_BEST_SLAB(best_slab_i32_e16, int32_t, uint16_t, 11)
VRX_BEST_SLAB(best_slab_i16_e32, int16_t, uint32_t, 16)
VRX_BEST_SLAB(best_slab_i32_e32, int32_t, uint32_t, 16)
#undef VRX_BEST_SLAB
static int best_slab_of(const int16_t* load, int nsp, const uint16_t* ent, uint32_t n, int w0, int wn,
                        const int16_t* full, int16_t* sc) { return best_slab_i16_e16(load, nsp, ent, n, w0, wn, full, sc); }
static int best_slab_of(const int16_t* load, int nsp, const uint16_t* ent, uint32_t n, int w0, int wn,
                        const int32_t* full, int32_t* sc) { return best_slab_i32_e16(load, nsp, ent, n, w0, wn, full, sc); }
static int best_slab_of(const int16_t* load, int nsp, const uint32_t* ent, uint32_t n, int w0, int wn,
                        const int16_t* full, int16_t* sc) { return best_slab_i16_e32(load, nsp, ent, n, w0, wn, full, sc); }
static int best_slab_of(const int16_t* load, int nsp, const uint32_t* ent, uint32_t n, int w0, int wn,
                        const int32_t* full, int32_t* sc) { return best_slab_i32_e32(load, nsp, ent, n, w0, wn, full, sc); }

template <class E, int RB>
static void balance_tile(const int32_t* rows, int64_t n_rows_tile, const int64_t* ptr, const int32_t* idx,
                         const uint8_t* words, int64_t n_contract, int n_slab, int slab_rows, int max_block,
                         int32_t* posmap, int32_t* perm, std::vector<uint32_t>& cptr) {
    const int64_t NC = n_contract;
    const size_t ne = cptr[(size_t)NC];
    std::vector<E> ent(ne);  // the tile's entries, column by column
    {
        std::vector<uint32_t> cur(cptr.begin(), cptr.end() - 1);
        for (int64_t i = 0; i < n_rows_tile; ++i)
            for (int64_t e = ptr[rows[i]]; e < ptr[rows[i] + 1]; ++e)
                if (words[e] != 0) ent[cur[(size_t)idx[e]]++] = (E)((uint32_t)i | (uint32_t)words[e] << RB);
    }
    // columns by degree, descending (counting sort, stable in the column index)
    uint32_t maxdeg = 0;
    for (int64_t c = 0; c < NC; ++c) maxdeg = std::max(maxdeg, cptr[(size_t)c + 1] - cptr[(size_t)c]);
    std::vector<uint32_t> dstart((size_t)maxdeg + 2, 0);
    for (int64_t c = 0; c < NC; ++c) ++dstart[(size_t)(maxdeg - (cptr[(size_t)c + 1] - cptr[(size_t)c])) + 1];
    for (size_t d = 0; d <= maxdeg; ++d) dstart[d + 1] += dstart[d];
    std::vector<int32_t> order((size_t)NC);
    for (int64_t c = 0; c < NC; ++c) order[dstart[(size_t)(maxdeg - (cptr[(size_t)c + 1] - cptr[(size_t)c]))]++] = (int32_t)c;
    const int nsp = (n_slab + 31) / 32 * 32;
    const VrxBalBlocks blocks = vrx_bal_blocks(n_slab, max_block);
    std::vector<int16_t> load((size_t)n_rows_tile * (size_t)nsp, 0), score16((size_t)nsp), full16((size_t)nsp, 0);
    std::vector<int32_t> score((size_t)nsp), full32((size_t)nsp, 0), cap((size_t)n_slab, slab_rows), fill((size_t)n_slab, 0);
    for (int sl = n_slab; sl < nsp; ++sl) {  // (padding slabs never have room)
        full16[(size_t)sl] = 0x7fff;
        full32[(size_t)sl] = 0x7fffffff;
    }
    for (int64_t p = 0; p < (int64_t)n_slab * slab_rows; ++p) perm[p] = 0;
    auto place = [&](int32_t c, int sl) {
        const int local = fill[(size_t)sl]++;
        if (--cap[(size_t)sl] == 0) {
            full16[(size_t)sl] = 0x7fff;
            full32[(size_t)sl] = 0x7fffffff;
        }
        posmap[c] = sl * slab_rows + local;
        perm[(int64_t)sl * slab_rows + local] = c;
    };
    std::vector<int> next_free((size_t)blocks.nb);  // per block: the first slab that may still have room
    for (int bi = 0; bi < blocks.nb; ++bi) next_free[(size_t)bi] = bi * blocks.bs;
    int32_t max_load = 0;  // the largest entry of `load` so far: decides whether 16-bit scores cannot overflow
    for (int64_t k = 0; k < NC; ++k) {
        const int32_t c = order[(size_t)k];
        const uint32_t a = cptr[(size_t)c], b = cptr[(size_t)c + 1];
        const int bi = (int)(c / slab_rows) / blocks.bs, w0 = bi * blocks.bs, wn = std::min(blocks.bs, n_slab - w0);
        int best = -1;
        if (a != b)
            best = (int64_t)(b - a) * max_load < 32000
                       ? best_slab_of(load.data(), nsp, ent.data() + a, b - a, w0, wn, full16.data(), score16.data())
                       : best_slab_of(load.data(), nsp, ent.data() + a, b - a, w0, wn, full32.data(), score.data());
        if (best < 0) {  // no entry in this tile (they come last): the first slab of the block with room
            int& nf = next_free[(size_t)bi];
            while (nf < w0 + wn && cap[(size_t)nf] == 0) ++nf;
            best = nf;  // (a block holds at least as many positions as columns: there is one)
        }
        place(c, best);
        for (uint32_t e = a; e < b; ++e) {
            int16_t& L = load[(size_t)(ent[e] & ((1u << RB) - 1)) * nsp + best];
            L = (int16_t)std::min<int32_t>(L + (int32_t)(ent[e] >> RB), VRX_BAL_LOAD_MAX);
            max_load = std::max<int32_t>(max_load, L);
        }
    }
}

// The per-tile greedy of the balanced-slab build, host version = the specification (vrx_balance.h).
// rows: the tile's rows (ids into ptr); idx / words: contracted index and FORM 1 word count of every entry;
// max_block: the largest block of slabs a contracted row may move in (<= 0: all of them);
// posmap[c] <- slab * slab_rows + slab-local position of contracted row c; perm = its inverse (unused
// positions name row 0: the pass stages some valid row there and no word refers to it)
void vrx_balance_tile(const int32_t* rows, int64_t n_rows_tile, const int64_t* ptr, const int32_t* idx,
                      const uint8_t* words, int64_t n_contract, int n_slab, int slab_rows, int max_block,
                      int32_t* posmap, int32_t* perm) {
    std::vector<uint32_t> cptr((size_t)n_contract + 1, 0);
    uint8_t wmax = 0;
    for (int64_t i = 0; i < n_rows_tile; ++i)
        for (int64_t e = ptr[rows[i]]; e < ptr[rows[i] + 1]; ++e) {
            cptr[(size_t)idx[e] + 1] += words[e] != 0;
            wmax = std::max(wmax, words[e]);
        }
    for (int64_t c = 0; c < n_contract; ++c) cptr[(size_t)c + 1] += cptr[(size_t)c];
    if (n_rows_tile <= 2048 && wmax < 32)  // (a 96-row tile and counts below 2^30: always)
        balance_tile<uint16_t, 11>(rows, n_rows_tile, ptr, idx, words, n_contract, n_slab, slab_rows, max_block, posmap, perm, cptr);
    else
        balance_tile<uint32_t, 16>(rows, n_rows_tile, ptr, idx, words, n_contract, n_slab, slab_rows, max_block, posmap, perm, cptr);
}
