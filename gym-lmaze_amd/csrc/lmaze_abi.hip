// lmaze_abi.hip -- the extern "C" surface declared in include/lmaze.h: argument checks and
// launches only.  No allocation, no synchronisation, no host-side compute fallback: if the
// launch fails the hipError_t goes back to the caller.
#include <stdio.h>
#include <string.h>

#include "lmaze_common.h"


using namespace lmaze;

static int check_params(const LmazeParams* p, int64_t n) {
    if (!p) return LMAZE_E_NULL;
    if (p->grid < 3 || p->grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (p->layout_mode != LMAZE_LAYOUT_SHARED && p->layout_mode != LMAZE_LAYOUT_PER_ENV) return LMAZE_E_LAYOUT;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    return 0;
}

static bool misaligned(const void* p, uintptr_t a) { return ((uintptr_t)p & (a - 1)) != 0; }

static StepArgs make_args(const LmazeParams* p, const uint8_t* layout, const int32_t* action, int32_t* ball_xy,
                          const int32_t* goal_xy, int32_t* step_count, float* reward, uint8_t* done,
                          int32_t* goal_count, int32_t* obs, int64_t n) {
    StepArgs a;
    a.layout = layout;
    a.action = action;
    a.ball = reinterpret_cast<int2*>(ball_xy);
    a.goal = reinterpret_cast<const int2*>(goal_xy);
    a.step_count = step_count;
    a.reward = reward;
    a.done = done;
    a.goal_count = goal_count;
    a.obs = obs;
    a.obs8 = nullptr;
    a.n = n;
    a.grid = p->grid;
    a.step_limit = p->step_limit;
    a.reward_wall = p->reward_wall;
    a.reward_move = p->reward_move;
    a.reward_goal = p->reward_goal;
    a.envs_per_block = 0;
    a.auto_reset = 0;
    a.seed = 0;
    a.epoch = 0;
    a.env_base = 0;
    a.epoch_in = nullptr;
    a.epoch_out = nullptr;
    a.goal_rw = nullptr;
    a.mask = nullptr;
    a.launch_hint = p->launch_hint;
    a.info = nullptr;
    return a;
}

namespace lmaze {
void describe_launch(LaunchInfo* info, const char* kernel, int epb, int per_cu, int chunks, bool nt, int64_t grid, int block, size_t lds) {
    snprintf(info->kernel, sizeof(info->kernel), "%s", kernel);
    info->envs_per_workgroup = epb;
    info->workgroups_per_cu = per_cu;
    info->chunks = chunks;
    info->non_temporal = nt ? 1 : 0;
    info->grid = grid;
    info->block = block;
    info->lds = (int64_t)lds;
}
}  // namespace lmaze

static int format_launch(const LaunchInfo& i, char* text, int32_t len) {
    if (!text || len < 1) return LMAZE_E_NULL;
    snprintf(text, (size_t)len, "%s grid=%lld block=%d lds=%lld envs_per_workgroup=%d workgroups_per_cu=%d chunks=%d", i.kernel,
             (long long)i.grid, i.block, (long long)i.lds, i.envs_per_workgroup, i.workgroups_per_cu, i.chunks);
    return 0;
}

extern "C" {

int lmaze_abi_version(void) { return LMAZE_ABI_VERSION; }

const char* lmaze_strerror(int code) {
    switch (code) {
        case 0: return "ok";
        case LMAZE_E_NULL: return "a required pointer is NULL";
        case LMAZE_E_GRID: return "grid outside [3, 64]";
        case LMAZE_E_VARIANT: return "params.variant does not match the entry point";
        case LMAZE_E_LAYOUT: return "unknown layout_mode";
        case LMAZE_E_COUNT: return "env count out of range";
        case LMAZE_E_ALIGN: return "buffer not aligned as documented";
        case LMAZE_E_EXPANSION: return "expansion ratio / channel count out of range";
        case LMAZE_E_NODEVICE: return "no usable HIP device";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown lmaze error";
    }
}

int lmaze_device_info(int device, int32_t* cu_count_host, char* name_host, int32_t name_len) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) {
        (void)hipGetLastError();
        return LMAZE_E_NODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return LMAZE_E_NODEVICE;
    if (cu_count_host) *cu_count_host = prop.multiProcessorCount;
    if (name_host && name_len > 0) {
        strncpy(name_host, prop.gcnArchName, (size_t)name_len - 1);
        name_host[name_len - 1] = 0;
    }
    return 0;
}

int lmaze_describe_step(const LmazeParams* params, int64_t n, int32_t auto_reset, int32_t with_obs, char* text_host,
                        int32_t len) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V0 && params->variant != LMAZE_VARIANT_V3) return LMAZE_E_VARIANT;
    if (!text_host || len < 1) return LMAZE_E_NULL;
    text_host[0] = 0;
    if (n == 0) return 0;
    LaunchInfo info;
    memset(&info, 0, sizeof(info));
    // nothing is dereferenced: the launcher fills `info` where it would have queued the kernel
    StepArgs a = make_args(params, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                           with_obs ? reinterpret_cast<int32_t*>(16) : nullptr, n);
    a.auto_reset = auto_reset ? 1 : 0;
    a.info = &info;
    if (with_obs == 2) {             // the narrow-observation step (lmaze_step_u8)
        if (params->layout_mode != LMAZE_LAYOUT_SHARED) return LMAZE_E_LAYOUT;
        if (params->grid < 4) return LMAZE_E_GRID;
        a.obs = nullptr;
        a.obs8 = reinterpret_cast<uint8_t*>(16);
        rc = (int)launch_step_u8(params->variant, true, a, nullptr);
        if (rc) return rc;
        return format_launch(info, text_host, len);
    }
    rc = (int)launch_step(params->variant, true, a, params->layout_mode, nullptr);
    if (rc) return rc;
    return format_launch(info, text_host, len);
}

int lmaze_step_v0(const LmazeParams* params, const uint8_t* layout, const int32_t* action, int32_t* ball_xy,
                  int32_t* step_count, float* reward, uint8_t* done, int32_t* goal_count, int32_t* obs,
                  int64_t n, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V0) return LMAZE_E_VARIANT;
    if (!layout || !action || !ball_xy || !step_count || !reward || !done) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16)) return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, action, ball_xy, nullptr, step_count, reward, done, goal_count, obs, n);
    return (int)launch_step(LMAZE_VARIANT_V0, true, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_step_v3(const LmazeParams* params, const uint8_t* layout, const int32_t* action, int32_t* ball_xy,
                  const int32_t* goal_xy, int32_t* step_count, float* reward, uint8_t* done, int32_t* obs,
                  int64_t n, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V3) return LMAZE_E_VARIANT;
    if (!layout || !action || !ball_xy || !goal_xy || !step_count || !reward || !done) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16))
        return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, action, ball_xy, goal_xy, step_count, reward, done, nullptr, obs, n);
    return (int)launch_step(LMAZE_VARIANT_V3, true, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_step_v0_autoreset(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                            int32_t* ball_xy, int32_t* step_count, float* reward, uint8_t* done,
                            int32_t* goal_count, int32_t* obs, int64_t n, uint64_t seed, uint64_t epoch,
                            int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V0) return LMAZE_E_VARIANT;
    if (!layout || !action || !ball_xy || !step_count || !reward || !done) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16)) return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, action, ball_xy, nullptr, step_count, reward, done, goal_count, obs, n);
    if (bad_epoch_words(epoch_in_dev, epoch_out_dev)) return LMAZE_E_ALIGN;
    a.auto_reset = 1;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.epoch_in = epoch_in_dev;
    a.epoch_out = epoch_out_dev;
    return (int)launch_step(LMAZE_VARIANT_V0, true, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_step_u8(const LmazeParams* params, const uint8_t* layout, const int32_t* action, int32_t* ball_xy, int32_t* goal_xy,
                  int32_t* step_count, float* reward, uint8_t* done, int32_t* goal_count, uint8_t* obs8, int64_t n,
                  int32_t auto_reset, uint64_t seed, uint64_t epoch, int64_t env_base, const uint64_t* epoch_in_dev,
                  uint64_t* epoch_out_dev, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    const bool v3 = params->variant == LMAZE_VARIANT_V3;
    if (params->variant != LMAZE_VARIANT_V0 && !v3) return LMAZE_E_VARIANT;
    if (params->layout_mode != LMAZE_LAYOUT_SHARED) return LMAZE_E_LAYOUT;
    if (params->grid < 4) return LMAZE_E_GRID;        // a 16-byte store must not span more than two envs
    if (!layout || !action || !ball_xy || !step_count || !reward || !done || (v3 && !goal_xy)) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs8, 16)) return LMAZE_E_ALIGN;
    if (bad_epoch_words(epoch_in_dev, epoch_out_dev)) return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, action, ball_xy, v3 ? goal_xy : nullptr, step_count, reward, done, v3 ? nullptr : goal_count, nullptr, n);
    a.obs8 = obs8;
    a.auto_reset = auto_reset ? 1 : 0;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.epoch_in = epoch_in_dev;
    a.epoch_out = epoch_out_dev;
    a.goal_rw = v3 ? reinterpret_cast<int2*>(goal_xy) : nullptr;
    return (int)launch_step_u8(params->variant, true, a, (hipStream_t)stream);
}

int lmaze_observe_u8(const LmazeParams* params, const uint8_t* layout, const int32_t* ball_xy, const int32_t* goal_xy,
                     const uint8_t* mask, uint8_t* obs8, int64_t n, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    const bool v3 = params->variant == LMAZE_VARIANT_V3;
    if (params->variant != LMAZE_VARIANT_V0 && !v3) return LMAZE_E_VARIANT;
    if (params->layout_mode != LMAZE_LAYOUT_SHARED) return LMAZE_E_LAYOUT;
    if (params->grid < 4) return LMAZE_E_GRID;
    if (!layout || !ball_xy || !obs8 || (v3 && !goal_xy)) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs8, 16)) return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, nullptr, const_cast<int32_t*>(ball_xy), v3 ? goal_xy : nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, n);
    a.obs8 = obs8;
    a.mask = mask;
    return (int)launch_step_u8(params->variant, false, a, (hipStream_t)stream);
}

int lmaze_rollout(const LmazeParams* params, const uint8_t* layout, const int32_t* actions, int32_t T, int32_t* ball_xy,
                  int32_t* goal_xy, int32_t* step_count, float* reward, uint8_t* done, int32_t* goal_count, int32_t* obs,
                  float* reward_t, uint8_t* done_t, int64_t n, int32_t auto_reset, uint64_t seed, uint64_t epoch,
                  int64_t env_base, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    const bool v3 = params->variant == LMAZE_VARIANT_V3;
    if (params->variant != LMAZE_VARIANT_V0 && !v3) return LMAZE_E_VARIANT;
    if (T < 0) return LMAZE_E_COUNT;
    if (!layout || !actions || !ball_xy || !step_count || !reward || !done || (v3 && !goal_xy)) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16)) return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, actions, ball_xy, v3 ? goal_xy : nullptr, step_count, reward, done, v3 ? nullptr : goal_count, obs, n);
    a.auto_reset = auto_reset ? 1 : 0;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.goal_rw = v3 ? reinterpret_cast<int2*>(goal_xy) : nullptr;
    return (int)launch_rollout(params->variant, a, params->layout_mode, actions, T, reward_t, done_t, (hipStream_t)stream);
}

int lmaze_step_v3_autoreset(const LmazeParams* params, const uint8_t* layout, const int32_t* action,
                            int32_t* ball_xy, int32_t* goal_xy, int32_t* step_count, float* reward,
                            uint8_t* done, int32_t* obs, int64_t n, uint64_t seed, uint64_t epoch,
                            int64_t env_base, const uint64_t* epoch_in_dev, uint64_t* epoch_out_dev, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V3) return LMAZE_E_VARIANT;
    if (!layout || !action || !ball_xy || !goal_xy || !step_count || !reward || !done) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16))
        return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, action, ball_xy, goal_xy, step_count, reward, done, nullptr, obs, n);
    if (bad_epoch_words(epoch_in_dev, epoch_out_dev)) return LMAZE_E_ALIGN;
    a.auto_reset = 1;
    a.seed = seed;
    a.epoch = epoch;
    a.env_base = env_base;
    a.epoch_in = epoch_in_dev;
    a.epoch_out = epoch_out_dev;
    a.goal_rw = reinterpret_cast<int2*>(goal_xy);
    return (int)launch_step(LMAZE_VARIANT_V3, true, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_observe(const LmazeParams* params, const uint8_t* layout, const int32_t* ball_xy,
                  const int32_t* goal_xy, int32_t* obs, int64_t n, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V0 && params->variant != LMAZE_VARIANT_V3) return LMAZE_E_VARIANT;
    if (!layout || !ball_xy || !obs) return LMAZE_E_NULL;
    if (params->variant == LMAZE_VARIANT_V3 && !goal_xy) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16))
        return LMAZE_E_ALIGN;
    StepArgs a = make_args(params, layout, nullptr, const_cast<int32_t*>(ball_xy), goal_xy, nullptr, nullptr,
                           nullptr, nullptr, obs, n);
    return (int)launch_step(params->variant, false, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_reset(const LmazeParams* params, const uint8_t* layout, const uint8_t* mask, uint64_t seed,
                uint64_t epoch, int64_t env_base, int32_t* ball_xy, int32_t* goal_xy, int32_t* step_count, float* reward,
                uint8_t* done, int32_t* obs, int64_t n, void* stream) {
    int rc = check_params(params, n);
    if (rc) return rc;
    if (params->variant != LMAZE_VARIANT_V0 && params->variant != LMAZE_VARIANT_V3) return LMAZE_E_VARIANT;
    if (!layout || !ball_xy || !step_count || !reward || !done) return LMAZE_E_NULL;
    if (params->variant == LMAZE_VARIANT_V3 && !goal_xy) return LMAZE_E_NULL;
    if (misaligned(ball_xy, 8) || misaligned(goal_xy, 8) || misaligned(obs, 16) || misaligned(layout, 16))
        return LMAZE_E_ALIGN;
    ResetArgs r;
    r.layout = layout;
    r.mask = mask;
    r.ball = reinterpret_cast<int2*>(ball_xy);
    r.goal = reinterpret_cast<int2*>(goal_xy);
    r.step_count = step_count;
    r.reward = reward;
    r.done = done;
    r.n = n;
    r.seed = seed;
    r.epoch = epoch;
    r.env_base = env_base;
    r.grid = params->grid;
    hipError_t e = launch_reset(params->variant, r, params->layout_mode, (hipStream_t)stream);
    if (e != hipSuccess || !obs) return (int)e;
    StepArgs a = make_args(params, layout, nullptr, ball_xy, goal_xy, nullptr, nullptr, nullptr, nullptr, obs, n);
    a.mask = mask;  // only the envs that were reset are re-rendered
    return (int)launch_step(params->variant, false, a, params->layout_mode, (hipStream_t)stream);
}

int lmaze_episode_stats(const uint8_t* done, const float* reward, const int32_t* step_count,
                        const int32_t* goal_count, float reward_goal, int64_t n, int64_t* out4, void* stream) {
    if (!done || !reward || !step_count || !out4) return LMAZE_E_NULL;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    if (misaligned(out4, 8)) return LMAZE_E_ALIGN;
    return (int)launch_episode_stats(done, reward, step_count, goal_count, reward_goal, n, out4, (hipStream_t)stream);
}

int lmaze_bandwidth_probe(const void* src, void* dst, int64_t bytes, void* stream) {
    if (!dst) return LMAZE_E_NULL;
    if (bytes < 0 || bytes > ((int64_t)1 << 36) || (bytes & 15)) return LMAZE_E_COUNT;
    if (misaligned(dst, 16) || (src && misaligned(src, 16))) return LMAZE_E_ALIGN;
    return (int)launch_probe(src, dst, bytes, (hipStream_t)stream);
}

int lmaze_render_expanded(const int32_t* obs, int32_t grid, int32_t expansion, const int32_t* channel_mask_host,
                          int32_t channels, float* out, int64_t n, void* stream) {
    if (!obs || !channel_mask_host || !out) return LMAZE_E_NULL;
    if (grid < 1 || grid > LMAZE_MAX_GRID) return LMAZE_E_GRID;
    if (expansion < 1 || expansion > 16 || channels < 1 || channels > LMAZE_MAX_CHANNELS) return LMAZE_E_EXPANSION;
    if (n < 0 || n > LMAZE_MAX_ENVS) return LMAZE_E_COUNT;
    if (misaligned(out, 16) || misaligned(obs, 4)) return LMAZE_E_ALIGN;
    ExpandArgs a;
    a.obs = obs;
    a.out = out;
    a.n = n;
    a.grid = grid;
    a.expansion = expansion;
    a.channels = channels;
    a.chunk_floats = 0;   // chosen by the launcher, like the reciprocals
    a.inv_l = a.inv_cells = 0;
    for (int c = 0; c < LMAZE_MAX_CHANNELS; ++c) a.mask[c] = c < channels ? channel_mask_host[c] : 0;
    return (int)launch_expand(a, (hipStream_t)stream);
}

}  // extern "C"
