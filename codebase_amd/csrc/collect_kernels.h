// K2 (batched QNetwork.act) and the fused IDQN collector for gfx950.
//
// marlhip_dqn_act   - marlbase/dqn/model.py:94-116 over N envs (modular entry point).
// marlhip_idqn_collect - marlbase/dqn/train.py:202-237 (_collect_trajectory) for N envs in ONE
//   launch per round: reset -> T x (act -> env.step -> ReplayBuffer.add).  A wave owns 16 envs
//   (the N dimension of the 16x16x4 MFMA); the 4 lanes {j, j+16, j+32, j+48} carry the SAME env
//   redundantly so that every lane can feed its own B-operand rows (obs element 4ks+g) without
//   any cross-lane traffic.  Env state lives in registers for the whole episode; the critics sit
//   in LDS as MFMA A-operand packs; the only HBM traffic is the replay write
//   (4*P*D + P + 4*P + 2 bytes per env-step).
#pragma once
#include "common.h"
#include "mlp.h"
#include "collect_common.h"
#include "env_traits.h"

namespace marl {

constexpr int COL_BLOCK = 256;

// part selector: the env.observe_id instantiations build in their own translation unit (collect_oid.hip)
int idqn_collect_dispatch_oid(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const LbfParams& q, const float* params, float epsilon,
                              uint32_t round, const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, int slot_base, int write_replay,
                              int clear_stale, int use_proper_termination, float* fin_return, int32_t* fin_length, hipStream_t stream);

template <class S>
__global__ __launch_bounds__(COL_BLOCK) void dqn_act_kernel(int P, int N, AgentMap am, const float* __restrict__ params,
                                                            const float* __restrict__ obs, float eps,
                                                            const float* __restrict__ u_in, const int32_t* __restrict__ rand_in,
                                                            uint64_t seed, const uint32_t* __restrict__ episode,
                                                            const int32_t* __restrict__ ep_length, int32_t* __restrict__ actions,
                                                            float* __restrict__ q_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int n = (blockIdx.x * 4 + wave) * 16 + j;
    const bool valid = n < N;
    const int nn = valid ? n : N - 1;
    float u;
    uint32_t epi = 0, t = 0;
    if (u_in != nullptr) {
        u = u_in[nn];
    } else {
        epi = episode[nn];
        t = (uint32_t)ep_length[nn];
        u = u01_f32(act_noise_word(seed, (uint32_t)nn, epi, t, 0));
    }
    const bool explore = eps > u;
    for (int p = 0; p < P; ++p) {
        __syncthreads();
        mlp_stage_fwd<S>(params + (size_t)am.net[p] * S::NPARAM, lds, tid, COL_BLOCK);
        __syncthreads();
        float x[S::KS1];
        const float* xrow = obs + ((size_t)p * N + nn) * S::D;
#pragma unroll
        for (int ks = 0; ks < S::KS1; ++ks) {
            const int d = 4 * ks + g;
            x[ks] = (d < S::D && valid) ? xrow[d] : 0.f;
        }
        f4 h1[S::MT], h2[S::MT], q;
        mlp_forward<S>(lds, lane, x, h1, h2, q);
        const int greedy = argmax_rows<S::A>(q, lane);
        int ra;
        if (rand_in != nullptr) ra = rand_in[(size_t)p * N + nn];
        else ra = (int)bounded_nr(act_noise_word(seed, (uint32_t)nn, epi, t, 1 + p), (uint32_t)S::A);
        if (valid) {
            if (g == 0) actions[(size_t)p * N + n] = explore ? ra : greedy;
            if (q_out != nullptr) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < S::A) q_out[((size_t)p * N + n) * S::A + 4 * g + r] = q[r];
            }
        }
    }
}

// NW: waves per block of 16 envs, each running the Q-networks of the agents p = w mod NW on its own copy of the env state (see
// ac_collect_kernel: the joint action is swapped through LDS once per step, every copy steps with it)
// HS = 2 (round 4): every agent's forward pass on TWO waves (mlp_forward_h2), i.e. NW * HS waves per env block, for launches that would
// otherwise leave half of the SIMDs idle (4096 envs x 2 agents); the half-1 wave ends up with the Q values and publishes the action, the
// half-0 waves of the block write its records.  One env block per workgroup (the forward pass has workgroup barriers inside).
template <class ENV, int H, bool OID, int NW, int HS = 1>
__global__ __launch_bounds__(NW * HS > 4 ? 64 * NW * HS : COL_BLOCK) void idqn_collect_kernel(typename ENV::Params q, const float* __restrict__ packs, float eps,
                                                                 uint32_t round, marlhip_replay_shape rs, marlhip_replay_buffers rb,
                                                                 int slot_base, int write_replay, int clear_stale, int proper_term,
                                                                 float* __restrict__ fin_return, int32_t* __restrict__ fin_length) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0), A = ENV::A;
    using S = MlpShape<D, H, A>;
    using PP = PackPlan<S, P, ENV::LDS_MAX>;
    constexpr bool RESIDENT = PP::RESIDENT || PP::A3REG;  // no per-step staging
    constexpr bool FROM_GLOBAL = NW > 1 && !RESIDENT;      // packs too large for the LDS: each wave reads its agents' from L2
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_act[NW > 1 ? 2 * 4 * P * 16 : 1];     // [step parity][env block of the workgroup][agent][env]
    // (the wave index through readfirstlane: everything derived from it - the agent, its pack's address - is then known to be wave-uniform
    // and lives in scalar registers; as a per-lane value the 8-agent hidden-128 kernels kept one 64-bit address per pack load in vector
    // registers, spilled them, and waited out every reload: 85 k cycles per step for two forward passes)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
    constexpr int WPB = NW * HS;  // waves per env block
    const int blk = wave / WPB, aw = (wave % WPB) % NW, half = (wave % WPB) / NW;
    const int bpw = (int)blockDim.x / (64 * WPB);  // env blocks per workgroup: 4 / WPB, or ONE when the launch has fewer waves than the chip has SIMDs
    const int n = (blockIdx.x * bpw + blk) * 16 + j;
    const int N = q.n_envs, T = rs.max_len;
    typename ENV::Ctx ctx;
    ctx.init(q, reinterpret_cast<uint8_t*>(lds) + (FROM_GLOBAL ? 0 : PP::LDS_BYTES), wave, j);
    const bool lead = g == 0 && aw == 0 && half == 0;  // the lane that writes an env's per-env records
    constexpr int XR = PP::A3REG ? 2 : 1, XT = S::MT / 2 / XR;  // exchange rounds of mlp_forward_h2 and tiles per round (hidden 128: 10 KB of LDS left next to the packs)
    __shared__ f4 s_xh[HS > 1 ? NW * 2 * XT * 64 : 1], s_xq[HS > 1 ? NW * 64 : 1];  // mlp_forward_h2's exchange (one env block per workgroup)
    const bool valid = n < N;
    const uint32_t env_id = (uint32_t)(valid ? n : N - 1);

    constexpr int K = P / NW;  // a wave's own agents: p = aw + k * NW, k < K (NW = 1: every agent, p = k)
    f4 a3[PP::A3REG ? K : 1][S::MT];  // output-layer operands of the wave's agents, when the full packs do not fit the LDS
    if (RESIDENT) {  // requested here, waited for in front of the first forward pass: the reset below runs under the copy
        for (int p = 0; p < P; ++p)
            stage_packed_async(packs + (size_t)p * S::NFWD, lds + (size_t)p * PP::STRIDE, PP::STRIDE, wave, (int)blockDim.x >> 6, lane);
        if (PP::A3REG) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int mt = 0; mt < S::MT; ++mt)
                    a3[k][mt] = reinterpret_cast<const f4*>(packs + (size_t)(aw + k * NW) * S::NFWD + S::pA3)[((HS == 2 && mt < S::MT / 2 ? half * (S::MT / 2) : 0) + mt) * 64 + lane];  // HS = 2: entries [0, MT/2) hold the half's own tiles
        }
    }

    typename ENV::State s;
    ENV::reset(q, s, ctx, env_id, round);
    const int slot = (int)(((int64_t)slot_base + (int64_t)env_id) % rs.capacity);
    float* ro = rb.obs + (size_t)slot * P * (T + 1) * D;
    uint8_t* ra_ = rb.act + (size_t)slot * P * T;
    float* rr = rb.rew + (size_t)slot * P * T;
    uint8_t* rd = rb.done + (size_t)slot * (T + 1);
    uint8_t* rf = rb.filled + (size_t)slot * T;
    const bool wr = valid && write_replay && half == 0;  // (HS = 2: the half-0 wave of an agent writes its rows)

    float x[K][S::KS1];  // the wave observes, forwards and stores for its own agents
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = aw + k * NW;
        ENV::template observe<S::KS1, OID>(q, s, ctx, p, g, x[k]);
        if (wr) {  // ReplayBuffer.init_episode (train.py:65-68)
#pragma unroll
            for (int ks = 0; ks < S::KS1; ++ks)
                if (4 * ks + g < D) ro[((size_t)p * (T + 1) + 0) * D + 4 * ks + g] = x[k][ks];
        }
    }

    bool alive = valid;
    float ep_ret[P];
#pragma unroll
    for (int p = 0; p < P; ++p) ep_ret[p] = 0.f;
    int len = 0;
    if (RESIDENT) stage_async_wait();  // the packs requested at the top have landed, for every wave

    for (int t = 0; t < T; ++t) {
        const bool any_alive = __any(alive);
        if (RESIDENT && WPB == 1) {
            if (!any_alive) break;  // wave-uniform: all 16 envs of this wave are finished (NW > 1: the barrier below keeps every wave looping)
        }
        int act[P], own[K];
        float u;
        int rnd[P];
        act_noise<P>(q.seed, env_id, round, (uint32_t)t, (uint32_t)A, u, rnd);
        const bool explore = eps > u;
#pragma unroll
        for (int p = 0; p < P; ++p) act[p] = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int p = aw + k * NW;
            own[k] = 0;
            if (WPB > 1 && !any_alive) continue;  // (uniform over the env block; HS = 2: over the workgroup, which is one block)
            const float* pack;
            if (RESIDENT) {
                pack = lds + (size_t)p * PP::STRIDE;
            } else if (FROM_GLOBAL) {
                pack = packs + (size_t)p * S::NFWD;
            } else {
                __syncthreads();
                stage_packed<S>(packs + (size_t)p * S::NFWD, lds, tid, (int)blockDim.x);
                __syncthreads();
                pack = lds;
            }
            f4 h1[S::MT], h2[S::MT], qv, unused;
            if constexpr (HS == 2) mlp_forward_h2<S, XR>(pack, lane, x[k], half, s_xh + aw * (2 * XT * 64), s_xq + aw * 64, qv, PP::A3REG ? a3[PP::A3REG ? k : 0] : nullptr);  // (qv: half 1 only)
            else if constexpr (FROM_GLOBAL) mlp_forward_g<S>(pack, lane, x[k], qv);
            else mlp_forward_p<S, false>(pack, pack, lane, x[k], h1, h2, qv, unused, PP::A3REG ? a3[PP::A3REG ? k : 0] : nullptr);
            const int greedy = argmax_rows<A>(qv, lane);
            own[k] = explore ? pick_agent<P>(rnd, p) : greedy;
            if (WPB == 1) act[k] = own[k];
        }
        if (WPB > 1) {  // swap the chosen actions among the waves of the env block (double-buffered: one barrier per step)
            int* sa = s_act + (((t & 1) * 4 + blk) * P) * 16;
#pragma unroll
            for (int k = 0; k < K; ++k)
                if (g == 0 && half == HS - 1) sa[(aw + k * NW) * 16 + j] = own[k];
            __syncthreads();
#pragma unroll
            for (int p = 0; p < P; ++p) act[p] = sa[p * 16 + j];
            if (HS > 1) {
#pragma unroll
                for (int k = 0; k < K; ++k) own[k] = pick_agent<P>(act, aw + k * NW);  // the half-0 wave stores it
            }
        }
        if (alive) {
            double raw[P];
            float rw[P];
            bool done;
            ENV::step(q, s, ctx, env_id, round, act, raw, done);
            const bool trunc = q.time_limit > 0 && ENV::elapsed(s) >= q.time_limit;
            const bool stored_done = proper_term ? done : (done || trunc);  // train.py:219-225
            lbf_wrap_rewards<P>(q, env_id, raw, rw, lead);
            ++len;
#pragma unroll
            for (int p = 0; p < P; ++p) ep_ret[p] += (float)raw[p];  // RecordEpisodeStatistics (wrappers.py:33)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
                ENV::template observe<S::KS1, OID>(q, s, ctx, p, g, x[k]);
                if (wr) {  // ReplayBuffer.add (train.py:73-84)
#pragma unroll
                    for (int ks = 0; ks < S::KS1; ++ks)
                        if (4 * ks + g < D) ro[((size_t)p * (T + 1) + t + 1) * D + 4 * ks + g] = x[k][ks];
                    if (g == 0) {
                        ra_[p * T + t] = (uint8_t)own[k];
                        rr[p * T + t] = pick_agent<P>(rw, p);
                    }
                }
            }
            if (wr && lead) {
                rd[t + 1] = stored_done ? 1 : 0;
                rf[t] = 1;
            }
            if (done || trunc) {
                alive = false;
                if (lead) {
#pragma unroll
                    for (int p = 0; p < P; ++p) fin_return[(size_t)p * N + n] = ep_ret[p];
                    fin_length[n] = len;
                }
            }
        }
    }
    if (valid && lead) {
        if (alive) {  // rs.max_len shorter than the env's own limits: report what was collected
#pragma unroll
            for (int p = 0; p < P; ++p) fin_return[(size_t)p * N + n] = ep_ret[p];
            fin_length[n] = len;
        }
        if (wr && clear_stale)
            for (int t = len; t < T; ++t) rf[t] = 0;
    }
}

template <class ENV, int H, bool OID, int NW, int HS = 1>
int launch_collect_nw(const typename ENV::Params& q, const float* packs, float eps, uint32_t round, const marlhip_replay_shape* rs,
                      const marlhip_replay_buffers* rb, int slot_base, int write_replay, int clear_stale, int proper_term, float* fin_return,
                      int32_t* fin_length, hipStream_t st) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0);
    using S = MlpShape<D, H, ENV::A>;
    using PP = PackPlan<S, P, ENV::LDS_MAX>;
    const size_t lds_bytes = ((NW > 1 && !(PP::RESIDENT || PP::A3REG)) ? 0 : PP::LDS_BYTES) + ENV::lds_bytes(q);
    static LdsAttr attr_set;
    if (attr_set.need(lds_bytes)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&idqn_collect_kernel<ENV, H, OID, NW, HS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set.done(lds_bytes);
    }
    // one env block per workgroup while the launch leaves SIMDs empty (see launch_ac_collect_nw): the action swap's barrier then only
    // joins the NW waves that need each other
    const bool one_block = HS > 1 || (NW > 1 && (int64_t)((q.n_envs + 15) / 16) * NW <= 1024);
    const int threads = (one_block || NW * HS > 4) ? 64 * NW * HS : COL_BLOCK, per_wg = 16 * (threads / (64 * NW * HS));
    timing_begin(TIMER_COLLECT, st);
    hipLaunchKernelGGL((idqn_collect_kernel<ENV, H, OID, NW, HS>), dim3((q.n_envs + per_wg - 1) / per_wg), dim3(threads), lds_bytes, st, q, packs, eps, round,
                       *rs, *rb, slot_base, write_replay, clear_stale, proper_term, fin_return, fin_length);
    timing_end(TIMER_COLLECT, st);
    MARL_CHECK_LAUNCH("idqn_collect_kernel");
    return 0;
}

// (Round 4, measured and dropped: 8 waves per env block for 8 agents in a 512-thread workgroup - the kernels take NW > 4 - is SLOWER, 1.46 ->
// 1.90 ms per 15x15-8p rollout at hidden 128: eight waves per workgroup halve the register budget per wave and the spills double.)
template <class ENV>
constexpr int col_max_nw() { return ENV::P % 4 == 0 ? 4 : (ENV::P % 2 == 0 ? 2 : 1); }

template <class ENV, int H, bool OID>
int launch_collect(const typename ENV::Params& q, const AgentMap& am, const float* params, float eps, uint32_t round, const marlhip_replay_shape* rs,
                   const marlhip_replay_buffers* rb, int slot_base, int write_replay, int clear_stale, int proper_term,
                   float* fin_return, int32_t* fin_length, hipStream_t st) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0), NWMAX = col_max_nw<ENV>();
    using S = MlpShape<D, H, ENV::A>;
    MARL_REQUIRE(ENV::lds_bytes(q) <= ENV::LDS_MAX, "collector: the env needs %zu bytes of LDS per workgroup, compiled for %zu", ENV::lds_bytes(q),
                 (size_t)ENV::LDS_MAX);
    float* packs = nullptr;
    if (launch_fwd_pack<S>(P, am, params, &packs, st) != 0) return -1;
    // agent-per-wave copies while the launch leaves SIMDs empty (N / 16 waves on 1024 SIMDs); never with env.standardise_rewards (its
    // per-env running records are read and committed by one wave in lockstep); MARLHIP_COL_NW=1 keeps one wave per env block
    static const int forced = getenv("MARLHIP_COL_NW") ? atoi(getenv("MARLHIP_COL_NW")) : 0;
    const bool split = NWMAX > 1 && q.reward_stats == nullptr && (forced ? forced > 1 : q.n_envs <= 8192);  // measured: ahead up to 8192 envs (2 and 4 agents), behind from 16384
#define MARL_COL_LAUNCH_ARGS q, (const float*)packs, eps, round, rs, rb, slot_base, write_replay, clear_stale, proper_term, fin_return, fin_length, st
    if constexpr (NWMAX > 1) {
        // two waves per agent while even agent-per-wave leaves half of the SIMDs idle (2 agents with LDS-resident packs, <= 4096 envs).
        // Measured and NOT extended (scripts/gpu_runs/r4W.sh, r4X.sh): 4 agents x 2 waves in a 512-thread workgroup fit 256 registers but
        // are 8 - 15 % SLOWER (warehouse rollout 6.9 -> 7.5 ms at hidden 128) - the unit's four matrix pipes are already busy with four
        // one-agent waves, a second wave per SIMD only adds the exchange; 2 agents with packs read from L2 (hidden 128): +-0, and so is a
        // fourth / fifth operand group in flight in mlp_forward_g - that pass is bound by the L2 bandwidth of 256 units each re-reading
        // its agents' packs every step, not by its MFMAs or its load latency (spreading the units over 1 / 4 / 8 identical pack sets: +-0
        // too, r4Y.sh).  2 agents at hidden 128 (the output layer's operands in registers, 146 KB of packs in LDS) do take the form, with the
        // exchange in two rounds of 8 KB: rollout 226 -> 185 us;
        // MARLHIP_COL_HS=1 keeps one wave per agent
        if constexpr (P == 2 && (PackPlan<S, P, ENV::LDS_MAX>::RESIDENT || PackPlan<S, P, ENV::LDS_MAX>::A3REG) && S::MT % 4 == 0) {
            static const bool hs_off = getenv("MARLHIP_COL_HS") != nullptr && atoi(getenv("MARLHIP_COL_HS")) == 1;
            if (split && !hs_off && (int64_t)((q.n_envs + 15) / 16) * NWMAX * 2 <= 1024) return launch_collect_nw<ENV, H, OID, NWMAX, 2>(MARL_COL_LAUNCH_ARGS);
        }
        if (split) return launch_collect_nw<ENV, H, OID, NWMAX>(MARL_COL_LAUNCH_ARGS);
    }
    return launch_collect_nw<ENV, H, OID, 1>(MARL_COL_LAUNCH_ARGS);
#undef MARL_COL_LAUNCH_ARGS
}

}  // namespace marl
