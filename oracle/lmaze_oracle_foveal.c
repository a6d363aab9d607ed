/*
 * lmaze_oracle_foveal.c -- TEST INFRASTRUCTURE ONLY.  Plain-C CPU restatement of the
 * reference's foveal variants: v1 = gym_lmaze/envs/lmaze_env_v1.py, v2 = lmaze_env_v2.py,
 * v4 = lmaze_env_v4.py.  Same role and rules as lmaze_oracle.c (checker only; never linked,
 * loaded or called by the product).
 *
 * Parity status: PINNED by tests/test_oracle_golden.py against tests/golden/v1_*, v2_*, v4_*
 * (rollouts of the reference's own step()/reset()/setFovealGoal(), oracle/gen_golden.py).
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lmaze.h"

void lmaze_oracle_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

#define F LMAZE_FOVEA
#define W25 (F * F)

static int channels_of(int variant) {
    return variant == LMAZE_VARIANT_V1 ? 4 : (variant == LMAZE_VARIANT_V2 ? 5 : 7);
}

static uint8_t cell_at(const uint8_t* lay, int G, int x, int y) {
    /* outside the array the reference would fail or wrap; the padded layouts never get there */
    return (x < 0 || y < 0 || x >= G || y >= G) ? 'W' : lay[x * G + y];
}

static float free_plane(uint8_t c) { return (c == 'B' || c == 'S' || c == 'X') ? 1.0f : 0.0f; } /* v1:79, v2:94 */

/* ---- observation builders: float[C,5,5] = the reference's retState before its xE loop ---- */

/* v1 getGlobalView (v1:204-238, goal plane = 'X' cells) / getLocalView (v1:242-279, goal plane =
 * one-hot at flat index f_goal_x*G + f_goal_y with numpy's negative-index wrap, v1:244-245) */
static void view_v1(const uint8_t* lay, int G, int bx, int by, int local, int fgx, int fgy, float* o) {
    long flat = (long)fgx * G + fgy;
    if (flat < 0) flat += (long)G * G;
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < F; ++j) {
            const int x = bx - 2 + i, y = by - 2 + j;
            const int in = x >= 0 && y >= 0 && x < G && y < G;
            const uint8_t c = cell_at(lay, G, x, y);
            o[0 * W25 + i * F + j] = (x == bx && y == by) ? 1.0f : 0.0f;           /* ball plane   */
            o[1 * W25 + i * F + j] = (in && c == 'W') ? 1.0f : 0.0f;               /* wall  v1:70  */
            o[2 * W25 + i * F + j] = local ? ((in && (long)x * G + y == flat) ? 1.0f : 0.0f)
                                           : ((in && c == 'X') ? 1.0f : 0.0f);     /* v1:74 / 245  */
            o[3 * W25 + i * F + j] = in ? free_plane(c) : 0.0f;                    /* free  v1:78  */
        }
}

/* v2: [free, goal](cur window) + action plane + [free, goal](previous window)  (v2:185-193);
 * v4: three planes each side, the third the visit map (v4:231-239), "previous" sampled from the
 * CURRENT visit map because retStatelast is a view of self.state. */
static void view_v24(int variant, const uint8_t* lay, int G, int cx, int cy, int px, int py, int gx, int gy,
                     int action, const float* visit, float* o) {
    const int per = variant == LMAZE_VARIANT_V4 ? 3 : 2;
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < F; ++j) {
            for (int side = 0; side < 2; ++side) {
                const int x = (side ? px : cx) - 2 + i, y = (side ? py : cy) - 2 + j;
                const int in = x >= 0 && y >= 0 && x < G && y < G;
                float* dst = o + (side ? (per + 1) * W25 : 0) + i * F + j;
                dst[0 * W25] = in ? free_plane(lay[x * G + y]) : 0.0f;
                dst[1 * W25] = (x == gx && y == gy) ? 1.0f : 0.0f;
                if (per == 3) dst[2 * W25] = in ? visit[x * G + y] : 0.0f;
            }
            o[per * W25 + i * F + j] = (action >= 0 && action == i * F + j) ? 1.0f : 0.0f; /* v2:135-136 */
        }
}

/* v4:116-119 / v4:211-214: state[2] = (state[2] + onehot_window(ball)) / 2 over the whole plane.
 * The reference adds in float64 and stores float32; one float32 add and an exact halving give the
 * same bits. */
static void visit_update(float* visit, int G, int bx, int by) {
    for (int x = 0; x < G; ++x)
        for (int y = 0; y < G; ++y) {
            const float w = (x >= bx - 2 && x <= bx + 2 && y >= by - 2 && y <= by + 2) ? 1.0f : 0.0f;
            visit[x * G + y] = (visit[x * G + y] + w) * 0.5f;
        }
}

static int check(const LmazeFovealParams* p, const uint8_t* layouts, const LmazeFovealBuffers* b, int64_t n) {
    if (!p || !layouts || !b) return LMAZE_E_NULL;
    if (p->variant != LMAZE_VARIANT_V1 && p->variant != LMAZE_VARIANT_V2 && p->variant != LMAZE_VARIANT_V4)
        return LMAZE_E_VARIANT;
    if (p->grid < F || p->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (p->n_layouts < 1 || p->n_layouts > LMAZE_MAX_LAYOUTS) return LMAZE_E_LAYOUT;
    if (n < 0) return LMAZE_E_COUNT;
    if (!b->ball_xy || !b->step_count || !b->reward || !b->done || !b->obs) return LMAZE_E_NULL;
    if (p->variant == LMAZE_VARIANT_V1 && (!b->fgoal_xy || !b->foveal_step_count || !b->foveal_reward || !b->foveal_done))
        return LMAZE_E_NULL;
    if (p->variant != LMAZE_VARIANT_V1 && (!b->goal_xy || !b->layout_id)) return LMAZE_E_NULL;
    if (p->variant == LMAZE_VARIANT_V4 && !b->visit) return LMAZE_E_NULL;
    return 0;
}

/* ---------------------------------------------------------------------------------- */
int lmaze_oracle_foveal_step(const LmazeFovealParams* p, const uint8_t* layouts, const int32_t* action,
                             const LmazeFovealBuffers* b, int64_t n) {
    int rc = check(p, layouts, b, n);
    if (rc) return rc;
    if (!action) return LMAZE_E_NULL;
    const int G = p->grid, C = channels_of(p->variant);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        float* obs = b->obs + (size_t)e * C * W25;
        int bx = b->ball_xy[2 * e], by = b->ball_xy[2 * e + 1];
        const int a = action[e];
        if (p->variant == LMAZE_VARIANT_V1) {
            const uint8_t* lay = layouts;
            const int fgx = b->fgoal_xy[2 * e], fgy = b->fgoal_xy[2 * e + 1];
            const int sc = b->step_count[e] + 1;                    /* v1:117 */
            const int fsc = b->foveal_step_count[e] + 1;            /* v1:118 */
            float fr = -0.0f, r = -0.0f;                            /* v1:120-121 */
            int local_done = 0;                                     /* v1:123 */
            int ox = 0, oy = 0;                                     /* v1:125-133 */
            if (a == 0) ox = -1; else if (a == 1) ox = 1; else if (a == 2) oy = -1; else if (a == 3) oy = 1;
            const uint8_t c = cell_at(lay, G, bx + ox, by + oy);
            if (c == 'W') {                                         /* v1:135-138 */
                r = p->reward_wall; fr = p->reward_wall;
            } else if (c == 'B') {                                  /* v1:140-163 */
                bx += ox; by += oy;
                r = p->reward_move; fr = p->reward_move;
                if (bx < fgx - 1 || bx > fgx + 2 || by < fgy - 1 || by > fgy + 2) {
                    local_done = 1; fr = p->reward_wall;
                } else if (by == fgy && bx == fgx) {
                    local_done = 1; fr = p->reward_goal;
                }
            } else if (c == 'X') {                                  /* v1:165-183 */
                bx += ox; by += oy;
                r = p->reward_goal;
                if (by == fgy && bx == fgx) { local_done = 1; fr = p->reward_goal; }
                else fr = p->reward_move;
            }
            const int done = (r == p->reward_goal || sc == p->step_limit);                     /* v1:294-304 */
            const int fdone = local_done || fr == p->reward_goal || fsc == p->foveal_step_limit || done; /* v1:308-324 */
            b->ball_xy[2 * e] = bx; b->ball_xy[2 * e + 1] = by;
            b->step_count[e] = sc; b->foveal_step_count[e] = fsc;
            b->reward[e] = r; b->foveal_reward[e] = fr;
            b->done[e] = (uint8_t)done; b->foveal_done[e] = (uint8_t)fdone;
            view_v1(lay, G, bx, by, 1, fgx, fgy, obs);              /* v1:200 getLocalView */
        } else {
            if (a < 0 || a >= W25) continue;                        /* the reference raises before touching anything */
            const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
            const int gx = b->goal_xy[2 * e], gy = b->goal_xy[2 * e + 1];
            const int px = bx, py = by;                             /* window of the previous observation */
            float r = -0.0f;                                        /* v2:146 */
            const int sc = b->step_count[e] + 1;                    /* v2:147 */
            const int fx = bx + a / F - 2, fy = by + a % F - 2;     /* v2:151-152 */
            if (fx < G - 2 && fx > 1 && fy < G - 2 && fy > 1) {     /* v2:157-159 */
                bx = fx; by = fy;
            } else {                                                /* v2:160-169: clamp the offending axis only */
                if (fx >= G - 2) bx = G - 3;
                if (fx <= 1) bx = 2;
                if (fy >= G - 2) by = G - 3;
                if (fy <= 1) by = 2;
            }
            float* visit = p->variant == LMAZE_VARIANT_V4 ? b->visit + (size_t)e * G * G : NULL;
            if (visit) visit_update(visit, G, bx, by);              /* v4:211-214 */
            const uint8_t c = cell_at(lay, G, fx, fy);
            if (fx == gx && fy == gy) r = p->reward_goal;           /* v2:175-180, judged at f_goal */
            else if (c == 'W') r = p->reward_wall;
            else if (c == 'B' || c == 'S') r = p->reward_move;
            b->ball_xy[2 * e] = bx; b->ball_xy[2 * e + 1] = by;
            b->step_count[e] = sc;
            b->reward[e] = r;
            b->done[e] = (r == p->reward_goal || sc > p->step_limit) ? 1 : 0;   /* v2:222 */
            view_v24(p->variant, lay, G, bx, by, px, py, gx, gy, a, visit, obs);
        }
    }
    return 0;
}

/* k-th (row-major) interior cell accepted by the reference's rejection loops; -1 if none.
 * kind 0: goal (v2:279: not 'W', not 'S');  kind 1: ball (v2:292: not 'W', not 'X', != goal) */
static int count_or_kth(const uint8_t* lay, int G, int kind, int goal_cell, int k) {
    int cnt = 0;
    for (int x = 1; x <= G - 2; ++x)
        for (int y = 1; y <= G - 2; ++y) {
            const uint8_t c = lay[x * G + y];
            const int ok = kind == 0 ? (c != 'W' && c != 'S') : (c != 'W' && c != 'X' && x * G + y != goal_cell);
            if (ok) { if (cnt == k) return x * G + y; ++cnt; }
        }
    return k < 0 ? cnt : -1;
}

int lmaze_oracle_foveal_reset(const LmazeFovealParams* p, const uint8_t* layouts, const uint8_t* mask, int32_t place,
                              uint64_t seed, uint64_t epoch, int64_t env_base, const LmazeFovealBuffers* b, int64_t n) {
    int rc = check(p, layouts, b, n);
    if (rc) return rc;
    const int G = p->grid, C = channels_of(p->variant), L = p->n_layouts;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        if (mask && !mask[e]) continue;
        const uint64_t ge = (uint64_t)(env_base + e);
        uint32_t ctr[4] = {(uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32)};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t r[4];
        lmaze_oracle_philox4x32_10(ctr, key, r);
        float* obs = b->obs + (size_t)e * C * W25;
        if (p->variant == LMAZE_VARIANT_V1) {
            if (place) {                                            /* v1:82-84 ball = first 'S' */
                for (int cidx = 0; cidx < G * G; ++cidx)
                    if (layouts[cidx] == 'S') { b->ball_xy[2 * e] = cidx / G; b->ball_xy[2 * e + 1] = cidx % G; break; }
            }
            b->reward[e] = -0.0f; b->foveal_reward[e] = -0.0f;      /* v1:90-91 */
            b->step_count[e] = 0;                                   /* v1:93 (fovealStepCount kept, v1:94) */
            b->done[e] = 0; b->foveal_done[e] = 0;
            view_v1(layouts, G, b->ball_xy[2 * e], b->ball_xy[2 * e + 1], 0, 0, 0, obs);   /* v1:100 getGlobalView */
            continue;
        }
        if (place) {
            int lid = b->layout_id[e];
            const int lid_new = (int)(((uint64_t)r[2] * (uint32_t)L) >> 32);
            if (p->variant == LMAZE_VARIANT_V4) lid = lid_new;      /* v4:97: setGrid first */
            if (lid < 0 || lid >= L) lid = 0;
            const uint8_t* lay = layouts + (size_t)lid * G * G;
            const int cg = count_or_kth(lay, G, 0, -1, -1);
            int goal_cell = -1;
            if (cg > 0) {
                goal_cell = count_or_kth(lay, G, 0, -1, (int)(((uint64_t)r[0] * (uint32_t)cg) >> 32));
                b->goal_xy[2 * e] = goal_cell / G; b->goal_xy[2 * e + 1] = goal_cell % G;
            }
            const int cb = count_or_kth(lay, G, 1, goal_cell, -1);
            if (cb > 0) {
                const int cell = count_or_kth(lay, G, 1, goal_cell, (int)(((uint64_t)r[1] * (uint32_t)cb) >> 32));
                b->ball_xy[2 * e] = cell / G; b->ball_xy[2 * e + 1] = cell % G;
            }
            b->layout_id[e] = lid_new;                              /* v2:92: setGrid last */
        }
        const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
        const int bx = b->ball_xy[2 * e], by = b->ball_xy[2 * e + 1];
        b->reward[e] = -0.0f;                                       /* v2:84 */
        b->step_count[e] = 0;                                       /* v2:86 */
        b->done[e] = 0;
        float* visit = NULL;
        if (p->variant == LMAZE_VARIANT_V4) {
            visit = b->visit + (size_t)e * G * G;
            memset(visit, 0, sizeof(float) * (size_t)G * G);        /* v4:112 */
            visit_update(visit, G, bx, by);                         /* v4:116-119 */
        }
        view_v24(p->variant, lay, G, bx, by, bx, by, b->goal_xy[2 * e], b->goal_xy[2 * e + 1], -1, visit, obs);
    }
    return 0;
}

/* v1:104-110 */
int lmaze_oracle_v1_set_foveal_goal(const LmazeFovealParams* p, const uint8_t* layouts, const int32_t* ij,
                                    const uint8_t* mask, const LmazeFovealBuffers* b, int64_t n) {
    int rc = check(p, layouts, b, n);
    if (rc) return rc;
    if (p->variant != LMAZE_VARIANT_V1) return LMAZE_E_VARIANT;
    if (!ij) return LMAZE_E_NULL;
    const int G = p->grid;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        if (mask && !mask[e]) continue;
        const int bx = b->ball_xy[2 * e], by = b->ball_xy[2 * e + 1];
        b->fgoal_xy[2 * e] = bx + ij[2 * e] - 2;
        b->fgoal_xy[2 * e + 1] = by + ij[2 * e + 1] - 2;
        b->foveal_step_count[e] = 0;
        view_v1(layouts, G, bx, by, 1, b->fgoal_xy[2 * e], b->fgoal_xy[2 * e + 1], b->obs + (size_t)e * 4 * W25);
    }
    return 0;
}

/* the xE loop on float planes, v2:197-203 */
int lmaze_oracle_expand_planes(const float* planes, int32_t channels, int32_t g, int32_t E, float* out, int64_t n) {
    if (!planes || !out) return LMAZE_E_NULL;
    if (g < 1 || g > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (E < 1 || E > 16 || channels < 1 || channels > 16) return LMAZE_E_EXPANSION;
    if (n < 0) return LMAZE_E_COUNT;
    const int S = g * E;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n * channels; ++i) {
        const float* src = planes + (size_t)i * g * g;
        float* dst = out + (size_t)i * S * S;
        for (int x = 0; x < g; ++x)
            for (int xx = 0; xx < E; ++xx)
                for (int y = 0; y < g; ++y)
                    for (int yy = 0; yy < E; ++yy) dst[(size_t)(x * E + xx) * S + y * E + yy] = src[x * g + y];
    }
    return 0;
}

/* ====================================================================================== */
/* v5 / v6: gym_lmaze/envs/lmaze_env_v5.py (v6 = v5 + safeFovealGoal, lmaze_env_v6.py:505-523) */
/* ====================================================================================== */
static int check_v56(const LmazeFovealParams* p, const uint8_t* layouts, const LmazeFovealBuffers* b, int64_t n) {
    if (!p || !layouts || !b) return LMAZE_E_NULL;
    if (p->variant != LMAZE_VARIANT_V5 && p->variant != LMAZE_VARIANT_V6) return LMAZE_E_VARIANT;
    if (p->grid < F || p->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (p->n_layouts < 1 || p->n_layouts > LMAZE_MAX_LAYOUTS) return LMAZE_E_LAYOUT;
    if (n < 0) return LMAZE_E_COUNT;
    if (!b->ball_xy || !b->goal_xy || !b->fgoal_xy || !b->layout_id || !b->step_count || !b->foveal_step_count ||
        !b->reward || !b->foveal_reward || !b->done || !b->foveal_done || !b->visit || !b->obs || !b->ball1_xy ||
        !b->fovea_xy || !b->last_xy || !b->foveal_goal || !b->obs_local)
        return LMAZE_E_NULL;
    return 0;
}

/* numpy index semantics on an axis of 5: -5..-1 wrap, anything else outside 0..4 raises (-> -1) */
static int wrap5(int i) {
    if (i >= 0 && i < F) return i;
    if (i < 0 && i >= -F) return i + F;
    return -1;
}

/* v5:306-348.  fov = [window3(fovea0), fovealGoal plane, window3(last)]; visit sampled live. */
static void fov_obs_v5(const uint8_t* lay, int G, int f0x, int f0y, int lx, int ly, int gx, int gy, int fg,
                       const float* visit, float* o) {
    view_v24(LMAZE_VARIANT_V4, lay, G, f0x, f0y, lx, ly, gx, gy, -1, visit, o);
    for (int c = 0; c < W25; ++c) o[3 * W25 + c] = (c == fg) ? 1.0f : 0.0f;   /* v5:166-169 one-hot */
}

/* v5:356-380 */
static void loc_obs_v5(const uint8_t* lay, int G, int f0x, int f0y, int f1x, int f1y, int b0x, int b0y, int b1x,
                       int b1y, int fg, float* o) {
    for (int i = 0; i < F; ++i)
        for (int j = 0; j < F; ++j) {
            const int x = f0x - 2 + i, y = f0y - 2 + j;
            const int in = x >= 0 && y >= 0 && x < G && y < G;
            o[0 * W25 + i * F + j] = in ? free_plane(lay[x * G + y]) : 0.0f;   /* v5:361 */
            o[1 * W25 + i * F + j] = 0.0f;
            o[2 * W25 + i * F + j] = 0.0f;
            o[3 * W25 + i * F + j] = (i * F + j == fg) ? 1.0f : 0.0f;           /* v5:368 */
        }
    int i = wrap5(b0x - f1x + 2), j = wrap5(b0y - f1y + 2);                     /* v5:364 */
    if (i >= 0 && j >= 0) o[1 * W25 + i * F + j] = 1.0f;
    i = wrap5(b1x - f1x + 2); j = wrap5(b1y - f1y + 2);                         /* v5:365 */
    if (i >= 0 && j >= 0) o[2 * W25 + i * F + j] = 1.0f;
}

int lmaze_oracle_v5_step(const LmazeFovealParams* p, const uint8_t* layouts, const int32_t* action,
                         const LmazeFovealBuffers* b, int64_t n) {
    int rc = check_v56(p, layouts, b, n);
    if (rc) return rc;
    if (!action) return LMAZE_E_NULL;
    const int G = p->grid;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
        float* visit = b->visit + (size_t)e * G * G;
        const int a = action[e];
        int b0x = b->ball_xy[2 * e], b0y = b->ball_xy[2 * e + 1];
        const int b1x = b0x, b1y = b0y;                            /* v5:193-194 */
        const int gx = b->goal_xy[2 * e], gy = b->goal_xy[2 * e + 1];
        const int fgx = b->fgoal_xy[2 * e], fgy = b->fgoal_xy[2 * e + 1];
        const int f1x = b->fovea_xy[4 * e + 2], f1y = b->fovea_xy[4 * e + 3];
        const int fsc = b->foveal_step_count[e];
        int ld = b->foveal_done[e], gd = b->done[e];               /* both persist across step() calls */
        float lr = -0.0f, gr;                                      /* v5:196 */
        const int sc = b->step_count[e] + 1;                       /* v5:197 */
        int dx = 0, dy = 0;                                        /* v5:205-217 */
        if (a == 0) dx = 1; else if (a == 1) dx = -1; else if (a == 2) dy = 1; else if (a == 3) dy = -1;
        const int nx = b0x + dx, ny = b0y + dy;
        const uint8_t c = cell_at(lay, G, nx, ny);
        if (c == 'W') {                                            /* v5:232-233 */
            lr = p->reward_wall;
        } else if (nx == fgx && ny == fgy) {                       /* v5:235-239 */
            lr = p->reward_goal; b0x = nx; b0y = ny; ld = 1;
        } else if (c == 'B' || c == 'S' || c == 'X') {             /* v5:241-248 */
            if (nx < f1x - 3 || nx > f1x + 2 || ny < f1y - 3 || ny > f1y + 2) ld = 1;
            lr = p->reward_move; b0x = nx; b0y = ny;
        }
        if (nx == gx && ny == gy) { gr = p->reward_goal; gd = 1; } /* v5:254-262, judged on the projected cell */
        else if (nx == fgx && ny == fgy) gr = p->reward_move;
        else gr = p->reward_wall;
        const int f0x = b0x, f0y = b0y;                            /* v5:264-265 */
        if (sc >= p->step_limit) ld = 1;                           /* v5:267 */
        if (fsc >= p->foveal_step_limit) { gd = 1; ld = 1; }       /* v5:269-271 */
        int lx = b->last_xy[2 * e], ly = b->last_xy[2 * e + 1];
        if (ld) visit_update(visit, G, f0x, f0y);                  /* v5:313-318 */
        if (fsc == 0) { lx = f0x; ly = f0y; }                      /* v5:322-323 */
        fov_obs_v5(lay, G, f0x, f0y, lx, ly, gx, gy, b->foveal_goal[e], visit, b->obs + (size_t)e * 7 * W25);
        if (ld) { lx = f0x; ly = f0y; }                            /* v5:344-346 */
        loc_obs_v5(lay, G, f0x, f0y, f1x, f1y, b0x, b0y, b1x, b1y, b->foveal_goal[e], b->obs_local + (size_t)e * 4 * W25);
        b->ball_xy[2 * e] = b0x; b->ball_xy[2 * e + 1] = b0y;
        b->ball1_xy[2 * e] = b1x; b->ball1_xy[2 * e + 1] = b1y;
        b->fovea_xy[4 * e] = f0x; b->fovea_xy[4 * e + 1] = f0y;
        b->last_xy[2 * e] = lx; b->last_xy[2 * e + 1] = ly;
        b->step_count[e] = sc;
        b->reward[e] = gr; b->foveal_reward[e] = lr;
        b->done[e] = (uint8_t)gd; b->foveal_done[e] = (uint8_t)ld;
    }
    return 0;
}

int lmaze_oracle_v5_planner_step(const LmazeFovealParams* p, const uint8_t* layouts, const int32_t* goal,
                                 const uint8_t* mask, const LmazeFovealBuffers* b, int64_t n) {
    int rc = check_v56(p, layouts, b, n);
    if (rc) return rc;
    if (!goal) return LMAZE_E_NULL;
    const int G = p->grid;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        if (mask && !mask[e]) continue;
        const int g = goal[e];
        if (g < 0 || g >= W25) continue;                           /* the reference raises half-way (v5:169) */
        const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
        const int b0x = b->ball_xy[2 * e], b0y = b->ball_xy[2 * e + 1];
        b->step_count[e] = 0;                                      /* v5:160 */
        b->reward[e] = -0.0f;                                      /* v5:161 globalReward */
        b->foveal_done[e] = 0;                                     /* v5:162 localDone */
        b->foveal_goal[e] = g;                                     /* v5:166-169 */
        b->fgoal_xy[2 * e] = b0x + g / F - 2;                      /* v5:172-173 */
        b->fgoal_xy[2 * e + 1] = b0y + g % F - 2;
        if (b->foveal_step_count[e] > 0) {                         /* v5:175-177 */
            b->fovea_xy[4 * e + 2] = b->fovea_xy[4 * e];
            b->fovea_xy[4 * e + 3] = b->fovea_xy[4 * e + 1];
        }
        b->foveal_step_count[e] += 1;                              /* v5:179 */
        loc_obs_v5(lay, G, b->fovea_xy[4 * e], b->fovea_xy[4 * e + 1], b->fovea_xy[4 * e + 2], b->fovea_xy[4 * e + 3],
                   b0x, b0y, b->ball1_xy[2 * e], b->ball1_xy[2 * e + 1], g, b->obs_local + (size_t)e * 4 * W25);
    }
    return 0;
}

/* reset(), v5:104-150; placement as v4 (setGrid first, v5:105) when place != 0 */
int lmaze_oracle_v5_reset(const LmazeFovealParams* p, const uint8_t* layouts, const uint8_t* mask, int32_t place,
                          uint64_t seed, uint64_t epoch, int64_t env_base, const LmazeFovealBuffers* b, int64_t n) {
    int rc = check_v56(p, layouts, b, n);
    if (rc) return rc;
    const int G = p->grid, L = p->n_layouts;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        if (mask && !mask[e]) continue;
        if (place) {
            const uint64_t ge = (uint64_t)(env_base + e);
            uint32_t ctr[4] = {(uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32)};
            uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
            uint32_t r[4];
            lmaze_oracle_philox4x32_10(ctr, key, r);
            const int lid = (int)(((uint64_t)r[2] * (uint32_t)L) >> 32);
            const uint8_t* lay = layouts + (size_t)lid * G * G;
            b->layout_id[e] = lid;
            const int cg = count_or_kth(lay, G, 0, -1, -1);
            int goal_cell = -1;
            if (cg > 0) {
                goal_cell = count_or_kth(lay, G, 0, -1, (int)(((uint64_t)r[0] * (uint32_t)cg) >> 32));
                b->goal_xy[2 * e] = goal_cell / G; b->goal_xy[2 * e + 1] = goal_cell % G;
            }
            const int cb = count_or_kth(lay, G, 1, goal_cell, -1);
            if (cb > 0) {
                const int cell = count_or_kth(lay, G, 1, goal_cell, (int)(((uint64_t)r[1] * (uint32_t)cb) >> 32));
                b->ball_xy[2 * e] = cell / G; b->ball_xy[2 * e + 1] = cell % G;
            }
        }
        const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
        const int bx = b->ball_xy[2 * e], by = b->ball_xy[2 * e + 1];
        b->foveal_reward[e] = -0.0f; b->reward[e] = -0.0f;         /* v5:107-108 */
        b->foveal_step_count[e] = 0; b->step_count[e] = 0;         /* v5:109-110 */
        b->done[e] = 0; b->foveal_done[e] = 0;                     /* v5:111-112 */
        b->foveal_goal[e] = 12;                                    /* v5:127-128 one-hot at the centre */
        float* visit = b->visit + (size_t)e * G * G;
        memset(visit, 0, sizeof(float) * (size_t)G * G);           /* v5:130 */
        b->fovea_xy[4 * e] = bx; b->fovea_xy[4 * e + 1] = by;      /* v5:136-143 */
        b->fovea_xy[4 * e + 2] = bx; b->fovea_xy[4 * e + 3] = by;
        b->fgoal_xy[2 * e] = bx; b->fgoal_xy[2 * e + 1] = by;
        b->ball1_xy[2 * e] = bx; b->ball1_xy[2 * e + 1] = by;
        b->last_xy[2 * e] = bx; b->last_xy[2 * e + 1] = by;        /* fovealStepCount == 0, v5:322 */
        fov_obs_v5(lay, G, bx, by, bx, by, b->goal_xy[2 * e], b->goal_xy[2 * e + 1], 12, visit, b->obs + (size_t)e * 7 * W25);
    }
    return 0;
}

/* The two-level loop around step() (include/lmaze.h lmaze_v5_hier_step), by composition of the three functions
 * above, each pinned on its own against the reference's recorded rollouts: reset() where globalDone is set on
 * entry (v5:104-150), plannerStep(goal) where localDone is set or the env was just reset (v5:158-182), step(action)
 * for everybody (v5:187-292). */
int lmaze_oracle_v5_hier_step(const LmazeFovealParams* p, const uint8_t* layouts, const int32_t* action,
                              const int32_t* planner_goal, const LmazeFovealBuffers* b, int64_t n, uint64_t seed,
                              uint64_t epoch, int64_t env_base) {
    int rc = check_v56(p, layouts, b, n);
    if (rc) return rc;
    if (!action || !planner_goal) return LMAZE_E_NULL;
    if (n == 0) return 0;
    uint8_t* m = (uint8_t*)malloc((size_t)2 * (size_t)n);
    if (!m) return LMAZE_E_COUNT;
    uint8_t *m_reset = m, *m_plan = m + n;
    for (int64_t e = 0; e < n; ++e) {
        m_reset[e] = b->done[e] != 0;
        m_plan[e] = m_reset[e] || b->foveal_done[e] != 0;
    }
    rc = lmaze_oracle_v5_reset(p, layouts, m_reset, 1, seed, epoch, env_base, b, n);
    if (!rc) rc = lmaze_oracle_v5_planner_step(p, layouts, planner_goal, m_plan, b, n);
    if (!rc) rc = lmaze_oracle_v5_step(p, layouts, action, b, n);
    free(m);
    return rc;
}

/* v6:505-523 */
int lmaze_oracle_v6_safe_foveal_goal(const LmazeFovealParams* p, const uint8_t* layouts, uint64_t seed, uint64_t epoch,
                                     int64_t env_base, const LmazeFovealBuffers* b, int32_t* out_goal, int64_t n) {
    int rc = check_v56(p, layouts, b, n);
    if (rc) return rc;
    if (!out_goal) return LMAZE_E_NULL;
    const int G = p->grid;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const uint8_t* lay = layouts + (size_t)b->layout_id[e] * G * G;
        const int bx = b->ball_xy[2 * e], by = b->ball_xy[2 * e + 1];
        const uint64_t ge = (uint64_t)(env_base + e);
        uint32_t ctr[4] = {(uint32_t)ge, (uint32_t)(ge >> 32), (uint32_t)epoch, (uint32_t)(epoch >> 32)};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
        uint32_t r[4];
        lmaze_oracle_philox4x32_10(ctr, key, r);
        int cnt = 0;
        for (int c = 0; c < W25; ++c) cnt += cell_at(lay, G, bx - 2 + c / F, by - 2 + c % F) != 'W';
        int pick = 12;
        if (cnt > 0) {
            int k = (int)(((uint64_t)r[0] * (uint32_t)cnt) >> 32);
            for (int c = 0; c < W25; ++c)
                if (cell_at(lay, G, bx - 2 + c / F, by - 2 + c % F) != 'W') { if (k == 0) { pick = c; break; } --k; }
        }
        out_goal[e] = pick;
    }
    return 0;
}
