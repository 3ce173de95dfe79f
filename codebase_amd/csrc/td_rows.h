// The TD stage between a forward pass that left the values of every (agent, step, episode) in HBM and a backward pass that takes
// dL/dvalues: Double-Q / max bootstrap with action masks, IDQN rows (QNetwork._compute_loss, marlbase/dqn/model.py:118-163) or the
// VDN sum (model.py:224-266).  Used by the recurrent learner (gru.hip) and by the GEMM path of the wide networks (wide.hip).
#pragma once
#include "common.h"

namespace marl {

// q, tq: [P][T+1][B][A]; dq: [P][T+1][B][A] (row T stays zero); lrow: [T][B]
static __global__ __launch_bounds__(256) void gru_td_kernel(int P, int T, int B, int A, const float* __restrict__ q, const float* __restrict__ tq,
                                                     marlhip_batch bt, float gamma, int double_q, int vdn, float* __restrict__ dq,
                                                     float* __restrict__ lrow) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= T * B) return;
    const int t = i / B, b = i - t * B;
    const float fl = bt.filled[i], dn = bt.dones[(size_t)(t + 1) * B + b];
    float tot_ch = 0.f, tot_tq = 0.f, loss = 0.f;
    for (int p = 0; p < P; ++p) {
        const float* qn = q + (((size_t)p * (T + 1) + t + 1) * B + b) * A;
        const float* tn = tq + (((size_t)p * (T + 1) + t + 1) * B + b) * A;
        const float* mk = bt.action_mask ? bt.action_mask + (((size_t)p * (T + 1) + t + 1) * B + b) * A : nullptr;
        int best = 0;
        float bv = -__builtin_huge_valf();
        for (int a = 0; a < A; ++a) {  // first index of the maximum (torch.argmax), over the online (Double-Q) or target values
            float v = double_q ? qn[a] : tn[a];
            if (mk != nullptr && mk[a] == 0.f) v = -1e8f;
            if (v > bv) { bv = v; best = a; }
        }
        float boot = tn[best];
        if (mk != nullptr && mk[best] == 0.f) boot = -1e8f;
        const int act = (int)bt.actions[((size_t)p * T + t) * B + b];
        const float ch = q[(((size_t)p * (T + 1) + t) * B + b) * A + act];
        if (vdn) {
            tot_ch += ch;
            tot_tq += boot;
        } else {
            const float y = bt.rewards[((size_t)p * T + t) * B + b] + gamma * boot * (1.f - dn);
            const float delta = ch - y;
            loss += delta * delta;
            for (int a = 0; a < A; ++a) dq[(((size_t)p * (T + 1) + t) * B + b) * A + a] = a == act ? 2.f * fl * delta : 0.f;
        }
    }
    if (vdn) {
        const float y = bt.rewards[(size_t)t * B + b] + gamma * tot_tq * (1.f - dn);  // batch.rewards[0] (model.py:228)
        const float delta = tot_ch - y;
        loss = delta * delta;
        for (int p = 0; p < P; ++p) {
            const int act = (int)bt.actions[((size_t)p * T + t) * B + b];
            for (int a = 0; a < A; ++a) dq[(((size_t)p * (T + 1) + t) * B + b) * A + a] = a == act ? 2.f * fl * delta : 0.f;
        }
    }
    lrow[i] = fl * loss;
}


// chosen_p = Q_p(o_t)[a_t], bootstrap_p = target Q_p(o_{t+1})[argmax] (QMixNetwork._compute_loss, dqn/model.py:384-410), and the
// transition's scalars in the [R] = [T * B] layout the mixer kernels read
static __global__ __launch_bounds__(256) void gru_qsel_kernel(int P, int T, int B, int A, const float* __restrict__ q, const float* __restrict__ tq,
                                                       marlhip_batch bt, int double_q, float* __restrict__ chosen, float* __restrict__ tqsel,
                                                       float* __restrict__ r0, float* __restrict__ dn, float* __restrict__ fl) {
    const int i = blockIdx.x * 256 + threadIdx.x, R = T * B;
    if (i >= R) return;
    const int t = i / B, b = i - t * B;
    for (int p = 0; p < P; ++p) {
        const float* qn = q + (((size_t)p * (T + 1) + t + 1) * B + b) * A;
        const float* tn = tq + (((size_t)p * (T + 1) + t + 1) * B + b) * A;
        const float* mk = bt.action_mask ? bt.action_mask + (((size_t)p * (T + 1) + t + 1) * B + b) * A : nullptr;
        int best = 0;
        float bv = -__builtin_huge_valf();
        for (int a = 0; a < A; ++a) {
            float v = double_q ? qn[a] : tn[a];
            if (mk != nullptr && mk[a] == 0.f) v = -1e8f;
            if (v > bv) { bv = v; best = a; }
        }
        float boot = tn[best];
        if (mk != nullptr && mk[best] == 0.f) boot = -1e8f;
        tqsel[(size_t)p * R + i] = boot;
        chosen[(size_t)p * R + i] = q[(((size_t)p * (T + 1) + t) * B + b) * A + (int)bt.actions[((size_t)p * T + t) * B + b]];
    }
    r0[i] = bt.rewards[i];  // batch.rewards[0] (model.py:379)
    dn[i] = bt.dones[(size_t)(t + 1) * B + b];
    fl[i] = bt.filled[i];
}

// dL/dchosen_p [P][R] from the mixer -> dense dL/dq rows [P][T+1][B][A] (row T was zeroed)
static __global__ __launch_bounds__(256) void gru_expand_dq_kernel(int P, int T, int B, int A, const float* __restrict__ dqm, marlhip_batch bt,
                                                            float* __restrict__ dq) {
    const int i = blockIdx.x * 256 + threadIdx.x, R = T * B;
    if (i >= R) return;
    const int t = i / B, b = i - t * B;
    for (int p = 0; p < P; ++p) {
        const int act = (int)bt.actions[((size_t)p * T + t) * B + b];
        const float v = dqm[(size_t)p * R + i];
        for (int a = 0; a < A; ++a) dq[(((size_t)p * (T + 1) + t) * B + b) * A + a] = a == act ? v : 0.f;
    }
}

}  // namespace marl
