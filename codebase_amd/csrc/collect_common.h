// helpers shared by the fused collectors (collect.hip, ac_collect.hip)
#pragma once
#include "common.h"
#include "mlp.h"

namespace marl {

__device__ __forceinline__ uint32_t act_noise_word(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t, int widx) {
    U4 c;
    c.x = env; c.y = episode; c.z = t | ((uint32_t)(widx >> 2) << 16); c.w = STREAM_ACT;
    const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int s = widx & 3;
    return s == 0 ? o.x : (s == 1 ? o.y : (s == 2 ? o.z : o.w));
}

// obs element selection: lane g of an env takes elements 4ks+g of the observation vector
template <int P, int F, int KS1>
__device__ __forceinline__ void pick_obs(const LbfObs<P, F>& o, int g, float (&x)[KS1]) {
    constexpr int D = 3 * (P + F);
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const float e0 = (4 * ks + 0 < D) ? o.v[(4 * ks + 0 < D) ? 4 * ks + 0 : 0] : 0.f;
        const float e1 = (4 * ks + 1 < D) ? o.v[(4 * ks + 1 < D) ? 4 * ks + 1 : 0] : 0.f;
        const float e2 = (4 * ks + 2 < D) ? o.v[(4 * ks + 2 < D) ? 4 * ks + 2 : 0] : 0.f;
        const float e3 = (4 * ks + 3 < D) ? o.v[(4 * ks + 3 < D) ? 4 * ks + 3 : 0] : 0.f;
        x[ks] = g == 0 ? e0 : (g == 1 ? e1 : (g == 2 ? e2 : e3));
    }
}

}  // namespace marl
