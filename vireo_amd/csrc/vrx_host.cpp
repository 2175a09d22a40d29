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
#include <cstdint>
#include <cstring>

#include "../../include/vireo_hip.h"

void vrx_set_error(const char* fmt, ...);

namespace {
constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_A = 0x9908b0dfu, MT_UP = 0x80000000u, MT_LO = 0x7fffffffu;

inline uint32_t mt_twist(uint32_t u, uint32_t v) {
    const uint32_t y = (u & MT_UP) | (v & MT_LO);
    return (y >> 1) ^ ((0u - (y & 1u)) & MT_A);
}

// one regeneration of the state, in place; every loop's reads are either of words the loop
// does not write or lie >= 227 words behind its writes, so the loops vectorise
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

extern "C" int vrx_mt19937_random_sample(uint32_t* key624, int32_t* pos, double* out, int64_t n) {
    if (!key624 || !pos || n < 0 || *pos < 0 || *pos > MT_N) {
        vrx_set_error("vrx_mt19937_random_sample: bad argument");
        return VRX_ERR_ARG;
    }
    int p = *pos;
    int64_t words = 2 * n;  // 32-bit outputs to consume
    if (!out) {             // skip: only the state moves
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
