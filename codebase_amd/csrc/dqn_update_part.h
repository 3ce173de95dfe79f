// One slice of the learner-kernel instantiations (hipcc time is dominated by these templates, so the shape list is spread over
// several translation units that build in parallel).  The including .hip defines MARL_PART_NAME and MARL_PART_SHAPES(X).
#include "dqn_update_kernels.h"

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

}  // namespace marl
