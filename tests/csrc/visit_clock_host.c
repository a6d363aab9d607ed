/* TEST INFRASTRUCTURE (CPU suite): the product's clock-relative visit-map arithmetic (gym-lmaze_amd/csrc/lmaze_visit.h,
 * the very header the kernels include) compiled for the host and run against the reference's eager recurrence
 *     state[2] = (state[2] + window) / 2          lmaze_env_v4.py:211-214 (float64 temporary, float32 store)
 * on random visit histories of one cell: bursts of visits, gaps of up to several hundred steps (through the subnormal
 * range to zero), renormalisation at LMAZE_VISIT_RENORM.  Returns the number of mismatching comparisons. */
#include <stdint.h>
#include "../../gym-lmaze_amd/csrc/lmaze_visit.h"

static uint64_t rng(uint64_t* s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return *s >> 33; }

int64_t visit_clock_fuzz(uint64_t seed, int64_t ticks, int64_t* compared, int64_t* subnormal_seen, int64_t* renorms) {
    uint64_t s = seed;
    float eager = 0.0f;              /* the reference's cell                                   */
    uint32_t stored = 0u;            /* the product's cell                                     */
    int clock = 0;
    int64_t bad = 0;
    int mode = 0, left = 0;
    for (int64_t t = 0; t < ticks; ++t) {
        if (left == 0) {             /* a new regime: mostly-in / mostly-out / long absence     */
            mode = (int)(rng(&s) % 4);
            left = mode == 3 ? (int)(100 + rng(&s) % 200) : (int)(1 + rng(&s) % 40);
        }
        --left;
        const int in = mode == 0 ? (rng(&s) % 8 != 0) : (mode == 1 ? (rng(&s) % 2) : (mode == 2 ? (rng(&s) % 8 == 0) : 0));
        if (rng(&s) % 997 == 0) { eager = 0.0f; stored = 0u; clock = 0; }        /* reset(): zeros, clock 0 */
        if (clock >= LMAZE_VISIT_RENORM) {                                        /* as the kernel does on entry */
            stored = lmaze_visit_true(stored, clock);
            clock = LMAZE_VISIT_BIAS;
            ++*renorms;
        }
        /* the reference: float64 temporary, stored as float32 */
        eager = (float)(((double)eager + (in ? 1.0 : 0.0)) / 2.0);
        /* the product: the whole-plane halving is the clock; only a window cell is touched */
        if (in) stored = lmaze_visit_add(stored, clock);
        ++clock;
        const uint32_t want = lmaze_float_bits(eager), got = lmaze_visit_true(stored, clock);
        if (want != 0u && want < 0x00800000u) ++*subnormal_seen;
        ++*compared;
        if (want != got) ++bad;
    }
    return bad;
}

uint32_t vc_true(uint32_t bits, int E) { return lmaze_visit_true(bits, E); }
uint32_t vc_add(uint32_t bits, int E) { return lmaze_visit_add(bits, E); }
