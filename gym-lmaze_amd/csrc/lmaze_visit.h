// The visit map of v4-v6 in its clock-relative frame (include/lmaze.h "The visit map"): bit-pattern arithmetic shared by
// the kernels (lmaze_foveal.hip) and by a host-compiled property test (tests/csrc/visit_clock_host.c, CPU suite), which
// runs it against the reference's own recurrence  state[2] = (state[2] + window) / 2  (lmaze_env_v4.py:211-214).
//
// A cell holds s = v * 2^(clock - LMAZE_VISIT_BIAS), v = the reference's float32 value.  Integer arithmetic on the bit
// patterns throughout, so the result does not depend on the device's denormal mode.
#ifndef LMAZE_VISIT_H_
#define LMAZE_VISIT_H_

#include <stdint.h>

#ifdef __HIPCC__
#define LMAZE_HD __host__ __device__ __forceinline__
#else
#define LMAZE_HD static inline
#endif

#define LMAZE_VISIT_BIAS 126    /* the frame in which stored == true value                                        */
#define LMAZE_VISIT_RENORM 250  /* a map whose clock got here is rewritten in true values first (clock := BIAS)  */

LMAZE_HD float lmaze_bits_float(uint32_t u) { union { uint32_t u; float f; } c; c.u = u; return c.f; }
LMAZE_HD uint32_t lmaze_float_bits(float f) { union { uint32_t u; float f; } c; c.f = f; return c.u; }

// n further halvings of a value whose exponent cannot absorb them.  The reference halves the float32 plane once per
// update (float64 temporary, stored back as float32): exact until the value drops below 2^-126, rounded to nearest-even
// on EVERY step from there -- replayed here one step at a time (dropping one bit is either exact or a tie), at most 25
// steps before any value is 0.  n < 0 (a clock below the frame a subnormal was stored in) doubles, always exact.
LMAZE_HD uint32_t lmaze_visit_true_slow(uint32_t bits, int n) {
    if (n < -300) n = -300;                   // a corrupted clock must not turn into a long loop (the result is garbage anyway)
    for (; n < 0; ++n) bits = (bits >> 23) ? bits + (1u << 23) : bits << 1;
    if (n == 0) return bits;
    const int f = (int)(bits >> 23);
    if (f > 1) {                              // the exact part: down to exponent field 1
        const int exact = n < f - 1 ? n : f - 1;
        bits -= (uint32_t)exact << 23;
        n -= exact;
    }
    if (n > 25) return 0u;
    uint32_t m = bits;                        // field <= 1: the pattern IS the value in units of 2^-149
    for (; n > 0; --n) {
        const uint32_t q = m >> 1;
        m = q + (m & q & 1u);                 // round half to even
    }
    return m;
}

// bit pattern of the reference's value of a cell stored as `bits` under clock E
LMAZE_HD uint32_t lmaze_visit_true(uint32_t bits, int E) {
    const int n = E - LMAZE_VISIT_BIAS, f = (int)(bits >> 23);
    if (f >= 1 && f - n >= 1) return (uint32_t)((int)bits - n * (1 << 23));
    if (bits == 0u) return 0u;
    return lmaze_visit_true_slow(bits, n);
}

// stored form, under clock E, of a value w in [0.5, 1] (what an update leaves in a window cell)
LMAZE_HD uint32_t lmaze_visit_store(float w, int E) {
    return (uint32_t)((int)lmaze_float_bits(w) + (E - LMAZE_VISIT_BIAS) * (1 << 23));
}

// a window cell at clock E: v' = fl32((v + 1) / 2) (v4:214: one float32 add, an exact halving), stored under E + 1.
// v below 2^-126 adds nothing whether the device flushes it or not: 1 + v rounds to 1.
LMAZE_HD uint32_t lmaze_visit_add(uint32_t bits, int E) {
    const float v = lmaze_bits_float(lmaze_visit_true(bits, E));
    return lmaze_visit_store((v + 1.0f) * 0.5f, E + 1);
}

#endif
