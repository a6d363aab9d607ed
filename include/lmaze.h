/*
 * lmaze.h -- C ABI of the MI355X-native batched L-maze step path (liblmaze_hip.so).
 *
 * The reference (gkm2708/gym-lmaze) has no FFI: its hot path is the body of each env
 * class's step()/reset() in pure Python.  This header is the boundary a maintainer of
 * the reference would bind instead of those bodies (ctypes stub in INTEGRATION.md).
 * Every entry point names the reference lines it replaces; "v0" below means
 * gym_lmaze/envs/lmaze_env.py, "vK" means gym_lmaze/envs/lmaze_env_vK.py.
 *
 * Conventions
 *   - plain pointers and sizes only; all pointers are DEVICE pointers (HBM) unless
 *     the name ends in _host; the caller owns every buffer; nothing is allocated,
 *     freed or synchronised inside; work is queued on `stream` (a hipStream_t passed
 *     as void*, NULL = the null stream).
 *   - N independent mazes ("envs"), struct-of-arrays, one element per env.
 *   - a maze layout is G*G bytes, row-major, holding the reference's own cell
 *     characters: 'W' wall, 'B' blank, 'S' start, 'X' goal marker (v0:37-48).
 *   - coordinates: x = row (first array axis), y = column, exactly as in the reference.
 *   - return value: 0 on success, a negative LMAZE_E_* for a rejected argument, or a
 *     positive hipError_t from the launch.
 */
#ifndef LMAZE_H_
#define LMAZE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMAZE_ABI_VERSION 4

/* which reference class the transition rules come from */
enum {
    LMAZE_VARIANT_V0 = 0, /* LmazeEnv      (v0:146-237) 4-neighbour, sticky reward           */
    LMAZE_VARIANT_V3 = 3  /* LmazeEnv_v3   (v3:220-402) 4-neighbour, look-ahead goal test    */
};

/* how `layout` is addressed */
enum {
    LMAZE_LAYOUT_SHARED = 0, /* one layout  uint8[G*G]    for every env (staged in LDS)      */
    LMAZE_LAYOUT_PER_ENV = 1 /* own layout  uint8[N*G*G]  per env (LDS tile per workgroup)   */
};

/* compact observation: one int32 per cell; every reference plane is a bit test */
enum {
    LMAZE_OBS_BALL = 1, /* v0 plane 0 (v0:80,178-183)  | v3 plane 1 (v3:167,256-261)         */
    LMAZE_OBS_WALL = 2, /* v0 plane 1 (v0:92-94): cell == 'W'                                */
    LMAZE_OBS_GOAL = 4, /* v0 plane 2 (v0:96-98): cell == 'X' | v3 plane 2: one-hot goal_xy  */
    LMAZE_OBS_FREE = 8  /* v0 plane 3 (v0:105-107): cell == 'B' | v3 plane 0: cell != 'W'    */
};

enum {
    LMAZE_E_NULL = -1,      /* a required pointer is NULL                                     */
    LMAZE_E_GRID = -2,      /* grid outside [3, LMAZE_MAX_GRID]                               */
    LMAZE_E_VARIANT = -3,   /* params.variant does not match the entry point                  */
    LMAZE_E_LAYOUT = -4,    /* unknown layout_mode                                            */
    LMAZE_E_COUNT = -5,     /* n < 0 or n too large for one launch                            */
    LMAZE_E_ALIGN = -6,     /* obs/ball_xy not aligned as documented                          */
    LMAZE_E_EXPANSION = -7, /* expansion ratio / channel count out of range                   */
    LMAZE_E_NODEVICE = -8   /* no HIP device / wrong architecture                             */
};

#define LMAZE_MAX_GRID 64
/* most envs one call accepts (LMAZE_E_COUNT beyond).  A launch is further limited to 2^24 - 1 workgroups of 256
 * threads (HIP: grid * block < 2^32); a count whose launch would need more -- the per-env-layout kernels take 4
 * envs per workgroup, i.e. 2^26 envs -- is refused with hipErrorInvalidConfiguration, never truncated. */
#define LMAZE_MAX_ENVS ((int64_t)1 << 30)
#define LMAZE_MAX_CHANNELS 8

/* Constants the reference hard-codes in __init__ (v0:17-23, v3:76-99). */
typedef struct LmazeParams {
    int32_t variant;     /* LMAZE_VARIANT_*                                                  */
    int32_t grid;        /* G = realgrid, side of the square layout incl. border (v0:17)     */
    int32_t layout_mode; /* LMAZE_LAYOUT_*                                                   */
    int32_t step_limit;  /* v0: done when stepCount == limit (v0:247); v3: > limit (v3:398)  */
    float reward_wall;   /* negativeNominal  -1.0   (v0:21)                                  */
    float reward_move;   /* positiveNominal  -0.01  (v0:22)                                  */
    float reward_goal;   /* positiveFull    100.0   (v0:23)                                  */
    int32_t launch_hint; /* 0 = library default launch policy.  Performance only, never results
                            (lmaze_step.hip launch_shared / launch_one); bits not listed are 0.
                              bits 0-3   workgroups resident per CU (1..8; 0 = default)
                              bits 4-7   chunks of envs a workgroup takes one after the other, loading
                                         the next chunk's inputs while it stores the current one
                                         (1..15; 0 = default)
                              bit  8     keep the workgroup/LDS kernel where the library would pick the
                                         wave-autonomous one (8x8 shared layouts whose planes stay
                                         on-die); for the wave-autonomous kernel bits 4-7 = envs per
                                         wave (1: 64, 2: 32, 3: 16), bits 0-3 = waves per workgroup
                                         (1, 2, 4)
                              bit  9     streaming regime without the fused reset: drop the early wait on the
                                         per-env loads that staggers the workgroups of a CU (a measured +8-15 %
                                         at 1M x 11x11; the launch-policy guard test times both)
                              bits 10-11 envs per workgroup (0 = default):
                                           11x11, 12x12        1: 64   2: 32   3: 16
                                           14x14, 18x18        1: 32   2: 16
                                           8x8 (large batch)   1: 128  2: 64
                                           32x32               1: 8    2: 4
                                           any other G         1: 256  2: 64   3: 16                     */
} LmazeParams;

int lmaze_abi_version(void);

/* Human-readable text for a code returned by any entry point (static storage). */
const char* lmaze_strerror(int code);

/* Number of visible HIP devices and, for `device`, CU count / arch name ("gfx950").
 * Returns 0 or LMAZE_E_NODEVICE.  name_host may be NULL. */
int lmaze_device_info(int device, int32_t* cu_count_host, char* name_host, int32_t name_len);

/*
 * Which kernel, grid and launch policy lmaze_step_v0 / _v3 (auto_reset != 0: the *_autoreset forms) would queue for n
 * envs with these params -- decided by the very code that launches, nothing is queued or dereferenced.  text_host
 * receives one line, e.g. "step_shared_kernel<11, v0, step, 32, nt> grid=32768 block=256 lds=20480
 * envs_per_workgroup=32 workgroups_per_cu=0 chunks=1" (workgroups_per_cu 0 = no cap; with_obs: 0 transition only, 1 the
 * int32 planes, 2 the narrow planes of lmaze_step_u8).  For bench.py's
 * roofline.kernel and the launch-policy guard test; no reference counterpart.
 */
int lmaze_describe_step(const LmazeParams* params, int64_t n, int32_t auto_reset, int32_t with_obs, char* text_host,
                        int32_t len);

/*
 * One step() of N v0 mazes: replaces v0:146-237 (action decode 153-170, collision and
 * position update 172-195, reward 174/184/194, done 246-249, plane build 208-215).
 *   action      int32[N]    0:(-1,0) 1:(+1,0) 2:(0,-1) 3:(0,+1), anything else (0,0)
 *   ball_xy     int32[N,2]  (ball_x0, ball_y0), read and updated; 8-byte aligned
 *   step_count  int32[N]    stepCount, incremented first (v0:151)
 *   reward      float[N]    read AND written: v0 keeps the previous reward when the
 *                           target cell is neither 'W','B' nor 'X' (no else, v0:172-195)
 *   done        uint8[N]    reward == reward_goal || step_count == step_limit (v0:246-249)
 *   goal_count  int32[N]    goalCount (v0:195); may be NULL
 *   obs         int32[N,G,G] compact planes after the move, fully rewritten; may be NULL
 *                           (transition only); 16-byte aligned
 * Envs never auto-reset (the reference does not); stepping after done follows v0 exactly.
 */
int lmaze_step_v0(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                  int32_t* ball_xy, int32_t* step_count, float* reward, uint8_t* done,
                  int32_t* goal_count, int32_t* obs, int64_t n, void* stream);

/*
 * One step() of N v3 mazes: replaces v3:220-402 (decode 234-247, collision/move 251-262,
 * look-ahead goal test 264-265, done 398).  The caller maps the reference's string
 * actions to ids ("left"/"0"->0, "right"/"1"->1, "up"/"2"->2, "down"/"3"->3, anything
 * else, including a Python int, -> a no-op id such as -1).
 *   goal_xy     int32[N,2]  (goal_x, goal_y), read only (set by reset, v3:147-152)
 *   reward      float[N]    written only (re-zeroed to -0.0 every step, v3:224)
 *   done        uint8[N]    reward == reward_goal || step_count > step_limit
 */
int lmaze_step_v3(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                  int32_t* ball_xy, const int32_t* goal_xy, int32_t* step_count, float* reward,
                  uint8_t* done, int32_t* obs, int64_t n, void* stream);

/*
 * Compact planes of the CURRENT state without stepping: what reset() returns after
 * placement (v0:92-120, v3:166-196).  goal_xy is read for LMAZE_VARIANT_V3 only (NULL
 * otherwise).
 */
int lmaze_observe(const LmazeParams* params, const uint8_t* layout, const int32_t* ball_xy,
                  const int32_t* goal_xy, int32_t* obs, int64_t n, void* stream);

/*
 * Masked on-device reset: replaces the placement + bookkeeping part of reset()
 * (v0:67-110; v3:142-167).  For every env with mask[i] != 0 (mask NULL = all):
 * step_count = 0, reward = -0.0, done = 0, and a new ball cell (v3: first a new goal
 * cell) drawn uniformly from the cells the reference's rejection loop accepts
 * (v0:70-78: interior, not 'W', not 'X'; v3:147-161: goal interior not 'W', ball interior
 * not 'W' and != goal).  Draws come from Philox4x32-10 keyed by (seed, env_base + i,
 * epoch): env_base is the global index of this shard's env 0, so a batch sharded over
 * several GPUs draws exactly what one GPU holding the whole batch would.  The reference
 * draws from Python's global Mersenne Twister, so placement parity with it is
 * distributional, not bitwise.  obs (nullable): the planes of the envs that were reset are
 * re-rendered (with a mask, envs outside it keep their current planes untouched).
 */
int lmaze_reset(const LmazeParams* params, const uint8_t* layout, const uint8_t* mask,
                uint64_t seed, uint64_t epoch, int64_t env_base, int32_t* ball_xy, int32_t* goal_xy,
                int32_t* step_count, float* reward, uint8_t* done, int32_t* obs, int64_t n,
                void* stream);

/*
 * step() with the reset fused in front of it, for rollouts longer than one episode: an env
 * whose done[i] is set ON ENTRY (by the previous step) is first reset exactly as
 * lmaze_reset(mask = done, seed, epoch, env_base) would -- new placement, step_count = 0,
 * reward = -0.0 (v0:64-110, v3:134-167) -- and then takes this step's action, i.e. the
 * user loop `if done: env.reset()` followed by `env.step(a)`, in one launch and with no
 * extra HBM traffic.  Results are bit-identical to calling lmaze_reset then lmaze_step_*.
 * The caller advances `epoch` every call so successive episodes draw fresh placements.
 * goal_xy (v3) is read and, for reset envs, rewritten.
 *
 * Device-resident epoch (both nullable; for launches captured in a hipGraph, whose host
 * arguments are frozen): with epoch_in_dev != NULL the launch draws with epoch + *epoch_in_dev,
 * and with epoch_out_dev != NULL too its first workgroup stores *epoch_in_dev + 1 there.  The two
 * must be DIFFERENT 8-byte-aligned device words (the rest of the grid still reads the first);
 * a captured rollout alternates them launch by launch, so each replay continues the count and
 * draws fresh placements.  LMAZE_E_ALIGN if they alias, are misaligned, or only _out is given.
 */
int lmaze_step_v0_autoreset(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                            int32_t* ball_xy, int32_t* step_count, float* reward, uint8_t* done,
                            int32_t* goal_count, int32_t* obs, int64_t n, uint64_t seed,
                            uint64_t epoch, int64_t env_base, const uint64_t* epoch_in_dev,
                            uint64_t* epoch_out_dev, void* stream);

int lmaze_step_v3_autoreset(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                            int32_t* ball_xy, int32_t* goal_xy, int32_t* step_count, float* reward,
                            uint8_t* done, int32_t* obs, int64_t n, uint64_t seed, uint64_t epoch,
                            int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev,
                            void* stream);

/*
 * The step with a NARROW observation: the same LMAZE_OBS_* bit mask in one byte per cell, obs8 uint8[N,G,G] (16-byte
 * aligned, nullable), 37 + G*G bytes per env-step instead of 37 + 4 G*G.  Shared layouts only (LMAZE_E_LAYOUT
 * otherwise); params->variant selects the rules, goal_xy is v3's (read; rewritten for reset envs), goal_count v0's (nullable).
 * auto_reset != 0 fuses the reset in exactly as lmaze_step_*_autoreset (seed, epoch, env_base, device-resident epoch words).
 * State and obs8 are what lmaze_step_v0 / _v3 leave, each plane dword narrowed to a byte.  The int32 planes remain the mode
 * BASELINE's metric is quoted on; this one has its own algorithmic bytes (bench.py --obs-dtype u8).
 * lmaze_observe_u8: the planes of the current state without stepping (mask != NULL: only the envs with mask[i] != 0).
 */
int lmaze_step_u8(const LmazeParams* params, const uint8_t* layout, const int32_t* action, int32_t* ball_xy, int32_t* goal_xy,
                  int32_t* step_count, float* reward, uint8_t* done, int32_t* goal_count, uint8_t* obs8, int64_t n,
                  int32_t auto_reset, uint64_t seed, uint64_t epoch, int64_t env_base, const uint64_t* epoch_in_dev,
                  uint64_t* epoch_out_dev, void* stream);
int lmaze_observe_u8(const LmazeParams* params, const uint8_t* layout, const int32_t* ball_xy, const int32_t* goal_xy,
                     const uint8_t* mask, uint8_t* obs8, int64_t n, void* stream);

/*
 * T steps of N v0 / v3 mazes over a pre-generated action tensor int32[T,N] (row t = step t): exactly T calls of
 * lmaze_step_v0 / _v3 -- with auto_reset != 0 of the *_autoreset forms, step t drawing with epoch + t -- with
 * bit-identical state and planes at the end (params->variant selects the rules; goal_xy for v3 only, goal_count for v0
 * only, both nullable as in the step calls).  reward_t float[T,N] / done_t uint8[T,N] (nullable) receive every step's
 * reward and done row.  The whole rollout is ONE launch: the lane that owns an env keeps its state in registers across the
 * T steps (on-die shared 8x8: a wave per 64 envs; otherwise a workgroup per 4-64 envs with the layout -- or its envs' own
 * layouts -- in LDS, read once per rollout), the planes are rewritten every step as T launches would, the per-env state
 * goes back once at the end (65 536 x 8x8: a step costs 6 us as a launch of its own, a third of it launch gap, 2.3-2.5 us
 * here; 1M x 32x32 per-env layouts 838 -> 722 us per step; lmaze_describe_step names the step kernel, this call its
 * rollout form; params->launch_hint bits 12-14 = k > 0: 4 << (k - 1) envs per workgroup of the shared-layout form
 * instead of the size the library picks -- for batches beyond the L2s the largest that keeps the resident workgroups'
 * planes inside them).  The caller advances its epoch by T.
 */
int lmaze_rollout(const LmazeParams* params, const uint8_t* layout, const int32_t* actions, int32_t T, int32_t* ball_xy,
                  int32_t* goal_xy, int32_t* step_count, float* reward, uint8_t* done, int32_t* goal_count, int32_t* obs,
                  float* reward_t, uint8_t* done_t, int64_t n, int32_t auto_reset, uint64_t seed, uint64_t epoch,
                  int64_t env_base, void* stream);

/*
 * Reference-layout observation: replaces the 5-deep upsample loop (v0:217-234,
 * v3:295-301).  out[i, c, x*E+xx, y*E+yy] = float((obs[i,x,y] & channel_mask[c]) != 0).
 *   obs           int32[N,G,G]        compact planes
 *   channel_mask  int32[channels]     HOST array, one LMAZE_OBS_* bit per output plane,
 *                                     e.g. v0 {1,2,4,8}, v3 {8,1,4}
 *   out           float[N,channels,G*E,G*E], 16-byte aligned
 */
int lmaze_render_expanded(const int32_t* obs, int32_t grid, int32_t expansion,
                          const int32_t* channel_mask_host, int32_t channels, float* out,
                          int64_t n, void* stream);

/*
 * Episode statistics of a batch, off the step path (the reference only keeps goalCount, v0:24,195):
 * out4 int64[4] (device; zeroed by this call) =
 *   { #envs with done set, #envs with reward == reward_goal, sum of step_count over the done envs,
 *     sum of goal_count (0 when goal_count is NULL) }.
 * Integer sums, so the result does not depend on the order of the reduction.  A multi-GPU run sums
 * the four numbers over ranks with one all_reduce (the only collective this library ever needs).
 */
int lmaze_episode_stats(const uint8_t* done, const float* reward, const int32_t* step_count,
                        const int32_t* goal_count, float reward_goal, int64_t n, int64_t* out4, void* stream);

/*
 * Measured ceiling of the device this library runs on (SURVEY 8(d) asks for a measured fill / copy
 * ceiling beside the 8 TB/s figure): src == NULL fills `bytes` of dst with one 16-byte store per
 * thread in launch order; otherwise copies src -> dst the same way.  bytes % 16 == 0, both pointers
 * 16-byte aligned.  bench.py times it with events on the launch stream and reports
 * roofline.measured_ceiling; nothing on the step path calls it.  No reference counterpart.
 */
int lmaze_bandwidth_probe(const void* src, void* dst, int64_t bytes, void* stream);

/* ====================================================================================== */
/* Foveal variants: the agent sees a 5x5 window (a8 crop, a9 frame history of SURVEY 8a).  */
/*   v1 = gym_lmaze/envs/lmaze_env_v1.py   v2 = lmaze_env_v2.py   v4 = lmaze_env_v4.py       */
/* The observation is float[N,C,5,5]: exactly the reference's `retState` before its xE loop   */
/* (v1 C=4: v1:242-256; v2 C=5: v2:185-193; v4 C=7: v4:231-239); lmaze_expand_planes makes    */
/* the (C,35,35) reference layout from it.                                                    */
/* ====================================================================================== */
enum {
    LMAZE_VARIANT_V1 = 1, /* LmazeEnv_v1 (v1:114-200): 4-neighbour move, two reward streams, foveal goal */
    LMAZE_VARIANT_V2 = 2, /* LmazeEnv_v2 (v2:127-225): 25-way teleport inside the fovea, 5 layouts       */
    LMAZE_VARIANT_V4 = 4, /* LmazeEnv_v4 (v4:167-272): v2 + float visit-map plane                        */
    LMAZE_VARIANT_V5 = 5, /* LmazeEnv_v5 (v5:187-292): two-level planner / local loop, 8-tuple return    */
    LMAZE_VARIANT_V6 = 6  /* LmazeEnv_v6: v5 + safeFovealGoal (v6:505-523); same step rules             */
};

#define LMAZE_FOVEA 5
#define LMAZE_MAX_LAYOUTS 16

typedef struct LmazeFovealParams {
    int32_t variant;            /* LMAZE_VARIANT_V1 / V2 / V4                                          */
    int32_t grid;               /* G (v1: 14, v1:22; v2/v4: 18)                                        */
    int32_t n_layouts;          /* rows of the layout table uint8[L,G,G] (v1: 1; v2/v4: 5, v2:309-405) */
    int32_t step_limit;         /* v1: done at stepCount == 200 (v1:295); v2/v4: > 50 (v2:222);
                                   v5/v6: localDone at stepCount >= 10 (v5:45,267)                     */
    int32_t foveal_step_limit;  /* v1: fovealStepCount == 10 (v1:309); v5/v6: >= 50 ends the episode
                                   (v5:46,269-271); unused by v2/v4                                    */
    float reward_wall;          /* negativeNominal -1.0                                                 */
    float reward_move;          /* positiveNominal v1 +0.01 (v1:28); v2/v4 -0.01 (v2:47)               */
    float reward_goal;          /* positiveFull    v1 1.0 (v1:29);   v2/v4 100.0                       */
    int32_t launch_hint;        /* 0 = library default launch policy; else bits 0-3 = workgroups per CU
                                   (1..8, 0 = no cap), bits 4-7 = envs per workgroup, 2: 32, 3: 64, 4: 128,
                                   5: 256 (anything else = default), bits 8-9 = chunks of that many envs a
                                   workgroup takes, minus one.  Performance only, never results
                                   (lmaze_foveal.hip launch_foveal_mode); other bits 0.                  */
} LmazeFovealParams;

/* Device pointers, one element per env; entries a variant does not use may be NULL. */
typedef struct LmazeFovealBuffers {
    int32_t* ball_xy;           /* [N,2] ball_x0, ball_y0 (window centre)                               */
    int32_t* goal_xy;           /* [N,2] goal_x, goal_y             v2, v4 (v1 reads 'X' off the layout) */
    int32_t* fgoal_xy;          /* [N,2] f_goal_x, f_goal_y         v1 (v1:104-110)                     */
    int32_t* layout_id;         /* [N]   row of the layout table    v2, v4 (v2:306)                     */
    int32_t* step_count;        /* [N]   stepCount                                                      */
    int32_t* foveal_step_count; /* [N]   fovealStepCount            v1 (not reset by reset(), v1:94)    */
    float* reward;              /* [N]   originalReward                                                 */
    float* foveal_reward;       /* [N]   fovealReward               v1                                  */
    uint8_t* done;              /* [N]                                                                  */
    uint8_t* foveal_done;       /* [N]   isFovealEpisodeFinished()  v1 (v1:308-324)                     */
    float* visit;               /* visit map state[2] of v4, v5, v6 (v4:116-119, v5:313-318) in the library's own
                                   CLOCK-RELATIVE, TILED form: lmaze_foveal_visit_bytes(G, N) bytes, 64-byte
                                   aligned, opaque to the caller -- see "The visit map" below; the reference's
                                   float[N,G,G] comes out of lmaze_foveal_materialise_visit                     */
    float* obs;                 /* [N,C,5,5], 16-byte aligned (v5/v6: the foveal observation, C = 7)    */
    /* v5 / v6 only (v5:62-78).  For them: reward = globalReward, foveal_reward = originalReward (the
     * local stream), done = globalDone, foveal_done = localDone, fgoal_xy = f_goal_x0/y0.               */
    int32_t* ball1_xy;          /* [N,2] ball_x1, ball_y1 (previous ball, v5:193-194)                   */
    int32_t* fovea_xy;          /* [N,4] fovea_x0, fovea_y0, fovea_x1, fovea_y1                         */
    int32_t* last_xy;           /* [N,2] window centre `retStatelast` is a view of (v5:322-323,344-346) */
    int32_t* foveal_goal;       /* [N]   index 0..24 of the one-hot fovealGoal plane (v5:166-169)        */
    float* obs_local;           /* [N,4,5,5] buildLocalObservation (v5:356-380), 16-byte aligned        */
    int32_t* visit_clock;       /* [N]   v4, v5, v6: bits 0-7 the whole-plane halvings the map has taken in its current
                                   frame; the library keeps a tag in the upper bits (v5/v6: which window centre the
                                   env's "previous window" record behind the tiles belongs to).  Opaque, as `visit`. */
} LmazeFovealBuffers;

/*
 * The visit map (v4:116-119,211-214; v5:313-318).  The reference keeps float32[G,G] per env and, on every update,
 * halves the WHOLE plane after adding 1 to the 5x5 window: state[2] = (state[2] + window) / 2.  Only window cells
 * are ever observable, so this library stores each cell relative to a per-env clock instead:
 *     stored s = v * 2^(clock - 126)      v = the reference's float32 value, clock = visit_clock[i]
 * "halve the whole plane" is clock += 1 and touches no cell; a window cell takes v' = fl32((v + 1) / 2) -- one
 * float32 add and an exact halving, which is the reference's float64 round trip rounded once -- and is stored
 * under the new clock.  Reading a cell back is an exponent subtraction while v stays in the normal range; below
 * 2^-126 the reference's own sequence of round-to-nearest-even halvings is replayed on the bit pattern (at most 25
 * steps to 0), so cells that were last seen hundreds of steps ago still come out bit-identical.  When a clock
 * reaches 250 the env's map is rewritten once in true values (clock := 126); reset() writes zeros (clock := 0).
 * Layout: tiles of 4x4 cells (64 bytes, one memory sector), ceil(G/4)^2 tiles per env, row-major tiles, row-major
 * cells inside a tile; a 5x5 window is always exactly 2x2 tiles (3 memory lines of 128 bytes on average).  A step
 * reads ten and writes back at most ten 16-byte tile rows of an env's map instead of streaming all 4*G*G bytes
 * twice.  Behind the tiles of the whole batch sit N records of 28 words: the true values of the window the
 * observation shows as "previous" (v5/v6 show it unchanged for up to ten steps: 112 contiguous bytes instead of a
 * second gather).  lmaze_foveal_visit_bytes() covers both.
 */
int64_t lmaze_foveal_visit_bytes(int32_t grid, int64_t n);

/* As lmaze_describe_step, for lmaze_foveal_step (auto_reset != 0: lmaze_foveal_step_autoreset; v5/v6:
 * lmaze_v5_hier_step). */
int lmaze_describe_foveal_step(const LmazeFovealParams* params, int64_t n, int32_t auto_reset, char* text_host, int32_t len);

/* out float[N,G,G] = the reference's state[2] of every env (true values, row-major), from the clock-relative
 * tiles.  Off the step path (tests, LmazeEnv_v4.state, checkpoints). */
int lmaze_foveal_materialise_visit(const LmazeFovealParams* params, const LmazeFovealBuffers* bufs, float* out,
                                   int64_t n, void* stream);

/* The inverse: take float[N,G,G] true values (e.g. the reference's own state[2]) into the tiled form;
 * visit_clock[i] := 126, the frame in which stored == true value. */
int lmaze_foveal_load_visit(const LmazeFovealParams* params, const LmazeFovealBuffers* bufs, const float* in,
                            int64_t n, void* stream);

/*
 * One step() of N foveal envs (v1:114-200 | v2:127-225 | v4:167-272).
 *   layouts  uint8[L,G,G] device; action int32[N] (v1: 0..3 else no move; v2/v4: 0..24 =
 *   5*row+col of the target cell inside the window; the reference raises IndexError outside
 *   that range before it changes anything, here such an id leaves the env -- state and obs --
 *   untouched).
 * Layouts need the reference's 2-cell (v1) / 4-cell (v2, v4) 'W' padding so the window never
 * leaves the array (v1:40-53, v2:309-326).
 */
int lmaze_foveal_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                      const LmazeFovealBuffers* bufs, int64_t n, void* stream);

/*
 * step() with the reset fused in front (v1, v2, v4): an env whose done[i] is set ON ENTRY is first reset
 * exactly as lmaze_foveal_reset(mask = done, place = 1, seed, epoch, env_base) would, then takes this step's
 * action -- bit-identical to the two calls, one launch.  (A masked reset launch per step costs +78 % on v2
 * and +39 % on v4 at 1M envs; fused it is free.)  v5/v6 episodes restart through plannerStep and are not
 * covered.  epoch_in_dev / epoch_out_dev: the device-resident epoch, as for lmaze_step_v0_autoreset.
 */
int lmaze_foveal_step_autoreset(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                                const LmazeFovealBuffers* bufs, int64_t n, uint64_t seed, uint64_t epoch,
                                int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev,
                                void* stream);

/*
 * reset() of the envs with mask[i] != 0 (NULL = all): step_count = 0, rewards = -0.0, done
 * flags cleared, visit map zeroed and visit_clock = 0 (v4: then (0 + window)/2, v4:116-119), and the reset
 * observation written (v1: global view v1:204-238; v2/v4: [window, zero action plane, window],
 * v2:109-110).  place != 0 also draws the placement on the device with Philox4x32-10 keyed by
 * (seed, env_base + i, epoch): v1 ball = the 'S' cell (v1:82-84); v2 goal, ball on the CURRENT
 * layout and only then a new layout_id (the reset-order quirk of v2:90-92); v4 layout_id first
 * (v4:97-104).  place == 0 keeps the caller's ball/goal/layout_id (e.g. the reference's own
 * draws).  Unmasked envs are untouched, their obs included.
 */
int lmaze_foveal_reset(const LmazeFovealParams* params, const uint8_t* layouts, const uint8_t* mask,
                       int32_t place, uint64_t seed, uint64_t epoch, int64_t env_base,
                       const LmazeFovealBuffers* bufs, int64_t n, void* stream);

/*
 * v1 setFovealGoal(i, j) (v1:104-110) for the envs with mask[i] != 0 (NULL = all):
 * f_goal = ball + (i, j) - 2, fovealStepCount = 0, obs = local view (v1:242-279).
 *   ij int32[N,2]
 */
int lmaze_v1_set_foveal_goal(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* ij,
                             const uint8_t* mask, const LmazeFovealBuffers* bufs, int64_t n, void* stream);

/*
 * v5/v6 plannerStep(goal) (v5:158-182) for the envs with mask[i] != 0 (NULL = all):
 * stepCount = 0, globalReward = -0.0, localDone = False, foveal goal = ball + (goal/5, goal%5) - 2,
 * fovea history shifted once fovealStepCount > 0, fovealStepCount += 1, obs_local written.
 * goal int32[N] in 0..24; the reference raises IndexError half-way through for any other value,
 * here such an env is left untouched.  lmaze_foveal_step / lmaze_foveal_reset take these variants
 * too: step() = v5:187-292 with actions 0:(+1,0) 1:(-1,0) 2:(0,+1) 3:(0,-1) (v5:205-217; the
 * opposite sign convention to v0), writing obs (foveal, v5:306-348) and obs_local.  Where the
 * reference's buildLocalObservation indexes outside its 5x5 frame (ball more than 2 cells right /
 * below fovea_1) it raises IndexError AFTER the state was updated; the kernel leaves that one-hot
 * plane empty and the state is the reference's.
 */
int lmaze_v5_planner_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* goal,
                          const uint8_t* mask, const LmazeFovealBuffers* bufs, int64_t n, void* stream);

/*
 * The two-level loop of v5/v6 as ONE launch per env-step: what a caller of the reference does around step() --
 *     if globalDone: reset()                       (v5:104-150)
 *     if localDone or it was just reset: plannerStep(goal)   (v5:158-182)
 *     step(action)                                 (v5:187-292)
 * -- for every env, keyed on the done[i] / foveal_done[i] flags ON ENTRY.  Bit-identical to
 * lmaze_foveal_reset(mask = done, place = 1, seed, epoch, env_base), then lmaze_v5_planner_step(planner_goal,
 * mask = done | foveal_done), then lmaze_foveal_step(action): state, visit map, obs and obs_local.  planner_goal
 * int32[N] is read for every env and used by those that take the plannerStep (a value outside 0..24 skips that
 * env's plannerStep, as lmaze_v5_planner_step does).  epoch_in_dev / epoch_out_dev: the device-resident epoch, as
 * for lmaze_step_v0_autoreset.
 */
int lmaze_v5_hier_step(const LmazeFovealParams* params, const uint8_t* layouts, const int32_t* action,
                       const int32_t* planner_goal, const LmazeFovealBuffers* bufs, int64_t n, uint64_t seed,
                       uint64_t epoch, int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev,
                       void* stream);

/*
 * v6 safeFovealGoal() (v6:505-523): for every env one window cell index 0..24 drawn uniformly from the
 * cells of the 5x5 window around the ball that are not 'W' (the reference rejects on np.random; here
 * Philox keyed by (seed, env_base + i, epoch), index (r*count)>>32 among the accepted cells in
 * row-major order).  out_goal int32[N].
 */
int lmaze_v6_safe_foveal_goal(const LmazeFovealParams* params, const uint8_t* layouts, uint64_t seed,
                              uint64_t epoch, int64_t env_base, const LmazeFovealBuffers* bufs,
                              int32_t* out_goal, int64_t n, void* stream);

/*
 * The xE nearest-neighbour loop on float planes (v1:258-277, v2:197-203, v4:243-249):
 * out[i, c, x*E+xx, y*E+yy] = planes[i, c, x, y].   planes float[N,C,g,g]; out 16-byte aligned.
 */
int lmaze_expand_planes(const float* planes, int32_t channels, int32_t g, int32_t expansion, float* out,
                        int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LMAZE_H_ */
