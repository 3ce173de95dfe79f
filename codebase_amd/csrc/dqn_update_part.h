// One slice of the learner-kernel instantiations (hipcc time is dominated by these templates, so the shape list is spread over
// several translation units that build in parallel).  The including .hip defines MARL_PART_NAME and MARL_PART_SHAPES(X).
#include "dqn_update_kernels.h"

#define MARL_PART_CAT2(a, b) a##b
#define MARL_PART_CAT(a, b) MARL_PART_CAT2(a, b)

namespace marl {

int MARL_PART_NAME(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                   const ReplaySrc* rsrc, float gamma, int32_t double_q, int32_t mode, void* workspace, int64_t workspace_bytes,
                   float* grad, float* loss, hipStream_t stream, const QmixCtx* qx, const RetStats* rst, bool* found) {
    *found = true;
#define X(d, h, a)                                                                                                    \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a)                                                       \
        return launch_lossgrad<MlpShape<d, h, a>>(s, params, target_params, bt, rsrc, gamma, double_q, mode, workspace, \
                                                  workspace_bytes, grad, loss, stream, qx, rst);
    MARL_PART_SHAPES(X)
#undef X
    *found = false;
    return 0;
}

// marlhip_idqn_update_n's step with the fused epilogue (reduce + clip-norm partials -> clip + Adam + target + next packs), for the
// shapes whose learner kernel keeps its packs in the workspace; *found = false: not such a shape, take the two-call path
int MARL_PART_CAT(MARL_PART_NAME, _fused)(const marlhip_net_shape* s, const float* params, const float* target_params, const marlhip_batch* bt,
                                         const ReplaySrc* rsrc, float gamma, int32_t double_q, int32_t mode, void* workspace,
                                         int64_t workspace_bytes, float* grad, float* loss, hipStream_t stream, UpdFuse* fuse, bool* found) {
    *found = true;
#define X(d, h, a)                                                                                                        \
    if (s->obs_dim == d && s->hidden == h && s->n_actions == a) {                                                         \
        if constexpr (fused_epilogue_ok<MlpShape<d, h, a>>())                                                             \
            return launch_lossgrad<MlpShape<d, h, a>>(s, params, target_params, bt, rsrc, gamma, double_q, mode, workspace, \
                                                      workspace_bytes, grad, loss, stream, nullptr, nullptr, fuse);       \
    }
    MARL_PART_SHAPES(X)
#undef X
    *found = false;
    return 0;
}

// the QMIX mixer stage alone (phase 0: mixers forward + backward from given chosen / bootstrap values; phase 1: the mixer-gradient
// reduce), for callers that compute the agent networks themselves (the recurrent path, gru.hip); keyed on the observation width
int MARL_PART_CAT(MARL_PART_NAME, _mix)(const marlhip_net_shape* s, const QmixCtx* qx, const marlhip_batch* bt, const QmixIo* io, float gamma,
                                       int phase, const float* loss, hipStream_t stream, bool* found) {
    *found = true;
    const ReplaySrc none = {};
#define X(d, h, a)                                                                                                       \
    if (s->obs_dim == d)                                                                                                 \
        return phase == 0 ? qmix_dispatch_mix<d, false>(s->n_agents, *qx, bt, none, *io, gamma, stream)                   \
                          : qmix_dispatch_reduce<d>(s->n_agents, *qx, bt->max_len, bt->batch, loss, stream);
    MARL_PART_SHAPES(X)
#undef X
    *found = false;
    return 0;
}

}  // namespace marl
