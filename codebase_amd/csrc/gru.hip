// extern "C" entry points of the recurrent (use_rnn) Q-network path; kernels in gru.h / gru_bwd.h
#include "gru.h"
#include "collect_common.h"

using namespace marl;

// (obs dim, actions) with compiled recurrent kernels, hidden 64: the LBF widths and the warehouse
#define MARL_GRU_SHAPES(X) X(12, 6) X(15, 6) X(18, 6) X(21, 6) X(24, 6) X(27, 6) X(39, 6) X(71, 5)

static int gru_check(const marlhip_net_shape* s) {
    MARL_REQUIRE(s != nullptr, "net shape is NULL");
    if (agent_map_validate(s) != 0) return -1;
    MARL_REQUIRE(s->hidden == 64, "recurrent networks are compiled for hidden 64 (layers: [64, 64]), got %d", s->hidden);
#define X(d, a) if (s->obs_dim == d && s->n_actions == a) return 0;
    MARL_GRU_SHAPES(X)
#undef X
    set_error("no recurrent kernels for obs_dim %d, %d actions (MARL_GRU_SHAPES)", s->obs_dim, s->n_actions);
    return -1;
}

extern "C" int marlhip_gru_nparams(const marlhip_net_shape* s) {
    if (gru_check(s) != 0) return -1;
#define X(d, a) if (s->obs_dim == d && s->n_actions == a) return GruShape<d, 64, a>::NPARAM;
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}

extern "C" int64_t marlhip_gru_record_floats(const marlhip_net_shape* s, int32_t steps, int32_t batch) {
    if (gru_check(s) != 0) return -1;
    return (int64_t)s->n_agents * steps * ((batch + 15) / 16) * GruShape<15, 64, 6>::REC;  // REC depends on H only
}

template <class S>
static int gru_forward(const marlhip_net_shape* s, const float* params, const float* obs, int steps, int B, const float* h_in, float* h_out,
                       float* q_out, float* rec, hipStream_t st) {
    const int P = s->n_agents;
    float* packs = collect_pack_scratch((size_t)P * S::NFWD * sizeof(float), st);
    MARL_REQUIRE(packs != nullptr, "gru_forward: cannot allocate the pack scratch");
    hipLaunchKernelGGL((gru_pack_kernel<S>), dim3((S::NFWD + 255) / 256, P), dim3(256), 0, st, params, agent_map(s), packs);
    MARL_CHECK_LAUNCH("gru_pack_kernel");
    const size_t lds = (size_t)S::NFWD * sizeof(float);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_fwd_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr = true;
    }
    hipLaunchKernelGGL((gru_seq_fwd_kernel<S>), dim3((B + 63) / 64, P), dim3(256), lds, st, (const float*)packs, obs, steps, B, h_in, h_out, q_out,
                       rec);
    MARL_CHECK_LAUNCH("gru_seq_fwd_kernel");
    return 0;
}

extern "C" int marlhip_gru_forward(const marlhip_net_shape* s, const float* params, const float* obs, int32_t steps, int32_t batch,
                                   const float* h_in, float* h_out, float* q_out, float* record, void* stream) {
    if (gru_check(s) != 0) return -1;
    MARL_REQUIRE(params && obs && q_out && steps > 0 && batch > 0, "gru_forward: bad argument");
#define X(d, a) \
    if (s->obs_dim == d && s->n_actions == a) return gru_forward<GruShape<d, 64, a>>(s, params, obs, steps, batch, h_in, h_out, q_out, record, (hipStream_t)stream);
    MARL_GRU_SHAPES(X)
#undef X
    return -1;
}
