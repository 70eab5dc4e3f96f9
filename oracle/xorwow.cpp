// =============================================================================
//  ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle.cpp header).
//
//  XORWOW priorities for the JPL coloring: restates
//  GenerateRandomColor (src/color_impl.cu:225-237): curandCreateGenerator(
//  CURAND_RNG_PSEUDO_DEFAULT) = XORWOW, seed 1234, curandGenerate of T u32.
//  cuRAND is an un-vendored third-party dependency (version unpinned,
//  config/config.mk:16-20).  Its documented LEGACY ordering for XORWOW is
//      value n  <-  subsequence (n mod 4096) [2^67 apart], position floor(n/4096)
//  (SURVEY.md Q2).  Two independent routes are provided so tests can cross-check:
//    orc_xorwow_legacy      own xorshift/Weyl step + rocRAND's host jump-ahead
//                           (rocrand_xorwow.h, seed scramble identical to cuRAND's)
//    orc_xorwow_sequential  own seed scramble + own step, subsequence 0 only
//  PARITY UNPINNED: no cuRAND output is available in this environment.
// =============================================================================
#include <cstdint>
#include <vector>
#include <rocrand/rocrand_xorwow.h>

namespace {
struct State {
    uint32_t x[5];
    uint32_t d;
};

// Marsaglia xorwow step as cuRAND/rocRAND implement it
inline uint32_t step(State& s) {
    const uint32_t t = s.x[0] ^ (s.x[0] >> 2);
    s.x[0] = s.x[1];
    s.x[1] = s.x[2];
    s.x[2] = s.x[3];
    s.x[3] = s.x[4];
    s.x[4] = (s.x[4] ^ (s.x[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437U;
    return s.d + s.x[4];
}

inline State seeded(unsigned long long seed) {
    State s;
    s.x[0] = 123456789U; s.x[1] = 362436069U; s.x[2] = 521288629U; s.x[3] = 88675123U; s.x[4] = 5783321U;
    s.d = 6615241U;
    const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fU;
    const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbU;
    const uint32_t t0 = 1228688033U * s0;
    const uint32_t t1 = 2073658381U * s1;
    s.x[0] += t0; s.x[1] ^= t0; s.x[2] += t1; s.x[3] ^= t1; s.x[4] += t0;
    s.d += t1 + t0;
    return s;
}

struct Engine : public rocrand_device::xorwow_engine {
    Engine(unsigned long long seed, unsigned long long subseq) : rocrand_device::xorwow_engine(seed, subseq, 0) {}
    State state() const {
        State s;
        for (int i = 0; i < 5; ++i) s.x[i] = m_state.x[i];
        s.d = m_state.d;
        return s;
    }
};
}  // namespace

extern "C" {

void orc_xorwow_legacy(unsigned long long seed, int n, uint32_t* out) {
    const int NSUB = 4096;
    int nsub = n < NSUB ? n : NSUB;
    for (int s = 0; s < nsub; ++s) {
        State st = Engine(seed, (unsigned long long)s).state();
        for (long long i = s; i < n; i += NSUB) out[i] = step(st);
    }
}

void orc_xorwow_sequential(unsigned long long seed, int n, uint32_t* out) {
    State st = seeded(seed);
    for (int i = 0; i < n; ++i) out[i] = step(st);
}

// the 4096 subsequence start states (6 words each: x0..x4, d) -- handed to the
// product path's tests to check its device-side jump-ahead
void orc_xorwow_substates(unsigned long long seed, int nsub, uint32_t* out) {
    for (int s = 0; s < nsub; ++s) {
        State st = Engine(seed, (unsigned long long)s).state();
        for (int i = 0; i < 5; ++i) out[s * 6 + i] = st.x[i];
        out[s * 6 + 5] = st.d;
    }
}
}
