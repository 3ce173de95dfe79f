// Fused actor-critic rollout collector for gfx950 - replaces _collect_trajectories
// (marlbase/ac/train.py:24-119) over N vector envs in ONE launch:
//   envs.reset() -> while running.any(): model.act (actor MLP -> Categorical sample) -> envs.step
//   (auto-reset vector env) -> masked writes of the still-running envs into the time-major batch.
// Same wave organisation as the IDQN collector (16 envs per wave, env state in registers, actor packs in
// LDS, f32 MFMA).  Reference semantics kept on purpose:
//   * an env whose episode ended is ignored until every env has finished (train.py:71,110);
//   * gymnasium(<1.0) AsyncVectorEnv auto-reset: the observation stored at t+1 of the FINAL transition is
//     the first observation of the env's NEXT episode, not the terminal one (train.py:79,90);
//   * done = done | truncated unless use_proper_termination (train.py:85-88).
// Sampling: Categorical(logits=actor_i(o_i)) by inverse CDF on the fp32 softmax with one Philox uniform per
// (env, step, agent) - torch.multinomial's stream is not reproducible on the device.
// Reset streams: the collection call with index `round` resets from Philox episode 2*round; the auto-reset
// observation comes from episode 2*round+1.
#pragma once
#include "collect_common.h"
#include "env_traits.h"
#include "mlp_keep.h"

namespace marl {

// part selector: the env.observe_id instantiations build in their own translation unit (ac_collect_oid.hip)
int ac_collect_dispatch_oid(const marlhip_lbf_config* cfg, const marlhip_net_shape* s, const LbfParams& q, const float* actor_params,
                            uint32_t round, int max_len, int use_proper_termination, float* batch_obs, int64_t* batch_act, float* batch_rew,
                            uint8_t* batch_done, float* batch_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                            hipStream_t stream);

constexpr int ACOL_BLOCK = 256;

// MARL_ACOL_PROF=1 (profiling builds only, scripts/build_variants.py): s_memtime at the region boundaries of a rollout step, summed per
// region over the steps and added by lane 0 of every wave to acol_prof[]: 0 actor forward, 1 Philox + Categorical sample, 2 action swap
// (barrier), 3 env step, 4 rewards / bookkeeping / auto-reset, 5 observation, 6 batch stores, 7 the whole kernel, 8 waves,
// 9 pack staging, 10 first reset + observation + row 0.
#ifndef MARL_ACOL_PROF
#define MARL_ACOL_PROF 0
#endif
#if MARL_ACOL_PROF
static __device__ unsigned long long acol_prof[16];
#define ACOL_TS_BEGIN unsigned long long ts_ = __builtin_readcyclecounter(), sp_[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; const unsigned long long ts0_ = ts_;
#define ACOL_TS(k)                                                      \
    {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        sp_[k] += now_ - ts_;                                           \
        ts_ = now_;                                                     \
    }
#define ACOL_TS_END                                                                          \
    if ((threadIdx.x & 63) == 0) {                                                           \
        sp_[7] = __builtin_readcyclecounter() - ts0_;                                        \
        for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&acol_prof[k_], sp_[k_]);                   \
        atomicAdd(&acol_prof[8], 1ull);                                                      \
        atomicAdd(&acol_prof[9], sp_[9]);                                                    \
        atomicAdd(&acol_prof[10], sp_[10]);                                                  \
    }
#else
#define ACOL_TS_BEGIN
#define ACOL_TS(k)
#define ACOL_TS_END
#endif


// softmax inverse-CDF sample for the env of batch row j; logits in C layout (lane (g,j) holds a = 4g+r)
template <int A>
__device__ __forceinline__ int sample_rows(const f4& logits, int lane, float u) {
    const int j = lane & 15;
    float l[A];
#pragma unroll
    for (int a = 0; a < A; ++a) l[a] = __shfl(logits[a & 3], (a >> 2) * 16 + j);
    float m = l[0];
#pragma unroll
    for (int a = 1; a < A; ++a) m = fmaxf(m, l[a]);
    float e[A], sum = 0.f;
#pragma unroll
    for (int a = 0; a < A; ++a) {
        e[a] = expf(l[a] - m);
        sum += e[a];
    }
    const float thr = u * sum;
    int act = A - 1;
#pragma unroll
    for (int a = A - 1; a >= 0; --a) {  // first a with cumsum(e)[a] > thr
        float ca = 0.f;
#pragma unroll
        for (int b = 0; b <= a; ++b) ca += e[b];
        if (ca > thr) act = a;
    }
    return act;
}

// NW = waves that share one block of 16 envs, each running the actors p = w mod NW: every one of them keeps its own copy of the
// env state (registers + its own LDS byte columns), computes only its agents' forward passes and samples, and the NW waves swap the
// sampled actions through LDS once per step; then each steps its copy with the full joint action - the same deterministic integer
// work, so the copies never diverge.  It buys latency when the launch cannot fill the chip anyway (N / 16 < number of SIMDs): the
// per-step chain of P forward passes becomes P / NW.  Actor packs that do not fit the LDS are then read straight from global memory
// (L2) by the wave that needs them - no per-step staging of every agent's pack by the whole workgroup.
// (8 waves per env block: measured and dropped, see col_max_nw in collect_kernels.h)
template <class ENV>
constexpr int acol_max_nw() { return ENV::P % 4 == 0 ? 4 : (ENV::P % 2 == 0 ? 2 : 1); }
constexpr int acol_tiles(int NW) { return NW > 4 ? NW : 4; }  // store tiles of a workgroup: one per wave

// Transposed batch stores (round 4).  A lane holds elements 4 ks + g of ITS env's observation, so a direct store instruction is 64
// separate 4-byte writes into 16 rows 4 * P * D bytes apart - on the warehouse (D = 71: 18 such instructions per step and wave) the
// stores were 20 % of a rollout step.  With a [16 envs][D] tile per wave in LDS (the exact image of the 16 consecutive agent-p rows
// of the batch), lane l then stores elements l, l + 64, ... of the tile: 64 consecutive floats of one or two rows per instruction.
// The (row, element) of every such store is the same at every step: byte offsets precomputed once, tile reads and writes on
// immediate offsets.  Used where the rows are wide enough to pay (D >= 32) and the LDS has the room next to resident packs.
template <int D>
struct ObsTile {
    static constexpr int ELEMS = 16 * D, NI = (ELEMS + 63) / 64, BYTES = ELEMS * 4;
};

template <class ENV, int H, bool OID, int NW>
constexpr size_t acol_lds_fixed() {  // packs (when they live in LDS) + the env's own bytes: what the kernel used before the tiles
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0);
    using PP = PackPlan<MlpShape<D, H, ENV::A>, P, ENV::LDS_MAX>;
    return ((NW > 1 && !(PP::RESIDENT || PP::A3REG)) ? 0 : PP::LDS_BYTES) + ENV::LDS_MAX;
}

template <class ENV, int H, bool OID, int NW>
constexpr bool acol_tstore() {
    constexpr int D = ENV::D0 + (OID ? ENV::P : 0);
    return D >= 32 && acol_lds_fixed<ENV, H, OID, NW>() + acol_tiles(NW) * (size_t)ObsTile<D>::BYTES + 4608 <= 160u * 1024u;  // 4608: the static action-swap buffer
}

__device__ __forceinline__ void wave_lds_fence_acol() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// HS = 2: two waves per agent (mlp_forward_h2; see idqn_collect_kernel) - NW * HS waves per env block, one block per workgroup
template <class ENV, int H, bool OID, int NW, int HS = 1>
__global__ __launch_bounds__(NW * HS > 4 ? 64 * NW * HS : ACOL_BLOCK) void ac_collect_kernel(typename ENV::Params q, const float* __restrict__ actor /* pre-packed [P][NFWD] */, uint32_t round, int T,
                                                                int proper_term, float* __restrict__ b_obs,
                                                                int64_t* __restrict__ b_act, float* __restrict__ b_rew,
                                                                uint8_t* __restrict__ b_done, float* __restrict__ b_filled,
                                                                float* __restrict__ fin_return, int32_t* __restrict__ fin_length,
                                                                int32_t* __restrict__ t_max, AcGhost gh, AcKeep kp) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0), A = ENV::A;
    using S = MlpShape<D, H, A>;
    using PP = PackPlan<S, P, ENV::LDS_MAX>;
    constexpr bool RESIDENT = PP::RESIDENT || PP::A3REG;  // no per-step staging
    constexpr bool FROM_GLOBAL = NW > 1 && !RESIDENT;      // packs too large for the LDS: each wave reads its agents' from L2
    const bool ghost = gh.env_ids != nullptr;              // second pass (AcGhost): no batch writes, episode records instead
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_act[NW > 1 ? 2 * 4 * P * 16 : 1];     // [step parity][env block of the workgroup][agent][env]
    // (the wave index through readfirstlane: everything derived from it - the agent, its pack's address - is then known to be wave-uniform
    // and lives in scalar registers; as a per-lane value the 8-agent hidden-128 kernels kept one 64-bit address per pack load in vector
    // registers, spilled them, and waited out every reload: 85 k cycles per step for two forward passes)
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, g = lane >> 4, j = lane & 15;
    constexpr int WPB = NW * HS;  // waves per env block
    const int blk = wave / WPB, aw = (wave % WPB) % NW, half = (wave % WPB) / NW;
    const bool storer = half == 0;  // HS = 2: the half-0 wave of an agent writes its rows (the half-1 wave samples)
    constexpr int XR = PP::A3REG ? 2 : 1, XT = S::MT / 2 / XR;  // exchange rounds of mlp_forward_h2 and tiles per round
    __shared__ f4 s_xh[HS > 1 ? NW * 2 * XT * 64 : 1], s_xq[HS > 1 ? NW * 64 : 1];
    const int bpw = (int)blockDim.x / (64 * WPB);  // env blocks per workgroup: 4 / NW, or ONE when the launch has fewer waves than the chip has SIMDs             // env block inside the workgroup, this wave's agent residue
    ACOL_TS_BEGIN
    const int n = (blockIdx.x * bpw + blk) * 16 + j;
    const int N = q.n_envs;
    typename ENV::Ctx ctx;
    ctx.init(q, reinterpret_cast<uint8_t*>(lds) + (FROM_GLOBAL ? 0 : PP::LDS_BYTES), wave, j);
    const bool valid = n < N;
    const int nn = valid ? n : N - 1;
    const uint32_t env_id = (uint32_t)(ghost ? gh.env_ids[nn] : nn);
    const int t_first = ghost ? gh.t_start[nn] : 0;
    uint32_t gen = ghost ? 1u : 0u;  // episode generation of the env inside this rollout: reset stream 2 * round + gen
    constexpr int K = P / NW;  // a wave's own agents: p = aw + k * NW, k < K (NW = 1: every agent, p = k)
    using OT = ObsTile<D>;
    constexpr bool TSTORE = acol_tstore<ENV, H, OID, NW>();
    // the wave's tile behind the packs and the env's bytes; fast path only for blocks of 16 envs inside the batch (wave-uniform)
    float* tile = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(lds) + (FROM_GLOBAL ? 0 : PP::LDS_BYTES) + ENV::LDS_MAX) + (size_t)wave * OT::ELEMS;
    const int n0 = (blockIdx.x * bpw + blk) * 16;
    const bool tstore = TSTORE && !ghost && n0 + 16 <= N && storer;
    // AcKeep (common.h): the wave leaves the logits and both hidden layers of its agents' rows where the learner step reads them (whole env
    // blocks of the batch only - the launcher checked B == n_envs and B % 16 == 0); wave-uniform
    const bool keep = kp.hid != nullptr && !ghost && n0 + 16 <= N;
    f4* const kp_hid = reinterpret_cast<f4*>(kp.hid);
    auto keep_slot = [&](int t, int p) { return ((((size_t)p * kp.T + t) * kp.bpt + (n0 >> 4)) * kp.stride) * 64 + lane; };
    int t_off[TSTORE ? OT::NI : 1];  // float offset of tile element 64 i + lane inside the block's 16 batch rows (agent 0)
    if constexpr (TSTORE) {
#pragma unroll
        for (int i = 0; i < OT::NI; ++i) {
            const int idx = 64 * i + lane, e = idx / D;
            t_off[i] = e * (P * D) + (idx - e * D);
        }
    }
    f4 a3[PP::A3REG ? K : 1][S::MT];  // output-layer operands of the wave's agents, when the full packs do not fit the LDS
    if (RESIDENT) {  // requested here, waited for in front of the first forward pass: the reset below runs under the copy
        for (int p = 0; p < P; ++p)
            stage_packed_async(actor + (size_t)p * S::NFWD, lds + (size_t)p * PP::STRIDE, PP::STRIDE, wave, (int)blockDim.x >> 6, lane);
        if (PP::A3REG) {
#pragma unroll
            for (int k = 0; k < K; ++k)
#pragma unroll
                for (int mt = 0; mt < S::MT; ++mt)
                    a3[k][mt] = reinterpret_cast<const f4*>(actor + (size_t)(aw + k * NW) * S::NFWD + S::pA3)[((HS == 2 && mt < S::MT / 2 ? half * (S::MT / 2) : 0) + mt) * 64 + lane];  // HS = 2: entries [0, MT/2) hold the half's own tiles
        }
    }
    ACOL_TS(9)
    typename ENV::State s;
    ENV::reset(q, s, ctx, env_id, 2u * round + gen);
    // batch_obs[t][n][p*D + d]
    auto obs_row = [&](int t) { return b_obs + ((size_t)t * N + env_id) * (P * D); };
    float x[K][S::KS1];  // the wave observes, forwards, samples and stores for its own agents
    // rows t of the block's 16 envs for agent p through the wave's tile: `keep` = this env's row is the observation (else zeros)
    auto store_rows_t = [&](int t, int p, const float (&xv)[S::KS1], bool keep) {
        wave_lds_fence_acol();  // the previous use of the tile has been read out
#pragma unroll
        for (int ks = 0; ks < S::KS1; ++ks)
            if (4 * ks < D) {
                if (4 * ks + 3 < D || 4 * ks + g < D) tile[j * D + 4 * ks + g] = keep ? xv[ks] : 0.f;
            }
        wave_lds_fence_acol();
        float* dst = b_obs + ((size_t)t * N + n0) * (P * D) + p * D;
#pragma unroll
        for (int i = 0; i < OT::NI; ++i) {
            if (64 * i + 63 < OT::ELEMS || 64 * i + lane < OT::ELEMS) dst[t_off[TSTORE ? i : 0]] = tile[64 * i + lane];
        }
    };
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int p = aw + k * NW;
        ENV::template observe<S::KS1, OID>(q, s, ctx, p, g, x[k]);
        if (tstore) {
            store_rows_t(0, p, x[k], true);
        } else if (valid && !ghost && storer) {
#pragma unroll
            for (int ks = 0; ks < S::KS1; ++ks)
                if (4 * ks + g < D) obs_row(0)[p * D + 4 * ks + g] = x[k][ks];
        }
    }
    const bool lead = g == 0 && aw == 0 && half == 0;  // the lane that writes an env's per-env records
    if (valid && lead && !ghost) b_done[env_id] = 0;
    bool running = valid && !ghost;
    int n_rec = 0;  // second pass: episodes recorded for this env
    float ep_ret[P];
#pragma unroll
    for (int p = 0; p < P; ++p) ep_ret[p] = 0.f;
    int len = 0;
    if (RESIDENT) stage_async_wait();  // the packs requested at the top have landed, for every wave
    ACOL_TS(10)
    for (int t = 0; t < T; ++t) {
        ACOL_TS(6)
        int act[P], own[K];
#pragma unroll
        for (int p = 0; p < P; ++p) act[p] = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) own[k] = 0;
        if (ghost) running = valid && t >= t_first && t < gh.t_stop;
        const bool any_running = __any(running);
        if ((RESIDENT || FROM_GLOBAL) ? any_running : true) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
                const float* pack;
                if (RESIDENT) {
                    pack = lds + (size_t)p * PP::STRIDE;
                } else if (FROM_GLOBAL) {
                    pack = actor + (size_t)p * S::NFWD;
                } else {
                    __syncthreads();
                    stage_packed<S>(actor + (size_t)p * S::NFWD, lds, tid, (int)blockDim.x);
                    __syncthreads();
                    pack = lds;
                }
                f4 h1[S::MT], h2[S::MT], logits, unused;
                f4* const d1 = keep ? kp_hid + kp.off_h1 + keep_slot(t, p) : nullptr;
                f4* const d2 = keep ? kp_hid + kp.off_h2 + keep_slot(t, p) : nullptr;
                if constexpr (HS == 2) {
                    mlp_forward_h2_keep<S, XR>(pack, lane, x[k], half, s_xh + aw * (2 * XT * 64), s_xq + aw * 64, logits, PP::A3REG ? a3[PP::A3REG ? k : 0] : nullptr, d1, d2);  // (logits: half 1 only)
                } else if constexpr (FROM_GLOBAL) {
                    mlp_forward_g_keep<S>(pack, lane, x[k], logits, d1, d2);
                } else {
                    mlp_forward_p<S, false>(pack, pack, lane, x[k], h1, h2, logits, unused, PP::A3REG ? a3[PP::A3REG ? k : 0] : nullptr);
                    if (keep) {
#pragma unroll
                        for (int mt = 0; mt < S::MT; ++mt) {
                            d1[mt * 64] = h1[mt];
                            d2[mt * 64] = h2[mt];
                        }
                    }
                }
                if (keep && half == HS - 1) {  // mlp_rows_fwd_kernel's `out`: lane (g, j) holds outputs 4 g + r of env j
                    float* lo = kp.logits + (((size_t)p * kp.T + t) * kp.B + n) * A + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < A) lo[r] = logits[r];
                }
                ACOL_TS(0)
                const float u = u01_f32(act_noise_word(q.seed, env_id, 2u * round, (uint32_t)t, 1 + p));
                own[k] = sample_rows<A>(logits, lane, u);
                if (WPB == 1) act[k] = own[k];
                ACOL_TS(1)
            }
        } else if (keep) {  // no env of the block is running any more: the rows stay in the batch (filled = 0) and their gradients are
            // masked, but 0 x (whatever the record held) must be 0
            const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
                f4* const d1 = kp_hid + kp.off_h1 + keep_slot(t, p);
                f4* const d2 = kp_hid + kp.off_h2 + keep_slot(t, p);
#pragma unroll
                for (int m = 0; m < S::MT / HS; ++m) {
                    d1[(half * (S::MT / HS) + m) * 64] = zero4;
                    d2[(half * (S::MT / HS) + m) * 64] = zero4;
                }
                if (half == HS - 1) {
                    float* lo = kp.logits + (((size_t)p * kp.T + t) * kp.B + n) * A + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * g + r < A) lo[r] = 0.f;
                }
            }
        }
        if (WPB > 1) {  // swap the sampled actions among the waves of the env block (double-buffered: one barrier per step)
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
        ACOL_TS(2)
        const bool ran = running;
        float rw_own[K];
#pragma unroll
        for (int k = 0; k < K; ++k) rw_own[k] = 0.f;
        if (running) {
            double raw[P];
            float rw[P];
            bool done;
            ENV::step(q, s, ctx, env_id, 2u * round + gen, act, raw, done);
            ACOL_TS(3)
            const bool trunc = q.time_limit > 0 && ENV::elapsed(s) >= q.time_limit;
            const bool fin = done || trunc;
            const bool stored_done = proper_term ? done : fin;
            lbf_wrap_rewards<P>(q, env_id, raw, rw, lead);
            ++len;
#pragma unroll
            for (int p = 0; p < P; ++p) ep_ret[p] += (float)raw[p];
            if (fin && ghost) {  // a further episode of an env that is no longer part of the batch: its statistics, then the next one
                if (lead && n_rec < gh.cap) {
                    const size_t at = (size_t)n * gh.cap + n_rec;
#pragma unroll
                    for (int p = 0; p < P; ++p) gh.ret[at * P + p] = ep_ret[p];
                    gh.meta[2 * at] = len;
                    gh.meta[2 * at + 1] = t + 1;
                }
                if (lead) gh.cnt[n] = n_rec + 1;  // the TRUE count: a value above `cap` tells the caller that records were dropped
                ++n_rec;
                len = 0;
#pragma unroll
                for (int p = 0; p < P; ++p) ep_ret[p] = 0.f;
            }
            if (fin) {  // vector-env auto-reset: the observation returned for this step is the next episode's first
                ++gen;
                ENV::reset(q, s, ctx, env_id, 2u * round + gen);
            }
            ACOL_TS(4)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
                ENV::template observe<S::KS1, OID>(q, s, ctx, p, g, x[k]);
                ACOL_TS(5)
                rw_own[k] = pick_agent<P>(rw, p);
                if (ghost || !storer) continue;
                if (tstore) continue;  // the rows of all 16 envs of the block go out together below
#pragma unroll
                for (int ks = 0; ks < S::KS1; ++ks)
                    if (4 * ks + g < D) obs_row(t + 1)[p * D + 4 * ks + g] = x[k][ks];
                if (g == 0) {
                    b_act[((size_t)t * N + n) * P + p] = own[k];
                    b_rew[((size_t)t * N + n) * P + p] = pick_agent<P>(rw, p);
                }
            }
            if (lead && !ghost) {
                b_done[(size_t)(t + 1) * N + n] = stored_done ? 1 : 0;
                b_filled[(size_t)t * N + n] = 1.f;
            }
            if (fin && !ghost) {
                running = false;
                if (lead) {
#pragma unroll
                    for (int p = 0; p < P; ++p) fin_return[(size_t)p * N + n] = ep_ret[p];
                    fin_length[n] = len;
                    atomicMax(t_max, len);
                }
            }
        } else if (valid && !ghost && !tstore && storer) {
            // rows of an env that is no longer running stay zero, as in the reference's freshly allocated batch
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
#pragma unroll
                for (int ks = 0; ks < S::KS1; ++ks)
                    if (4 * ks + g < D) obs_row(t + 1)[p * D + 4 * ks + g] = 0.f;
                if (g == 0) {
                    b_act[((size_t)t * N + n) * P + p] = 0;
                    b_rew[((size_t)t * N + n) * P + p] = 0.f;
                }
            }
            if (lead) {
                b_done[(size_t)(t + 1) * N + n] = 0;
                b_filled[(size_t)t * N + n] = 0.f;
            }
        }
        if (tstore) {  // wave-uniform: rows t + 1 of the block's 16 envs (zeros for the envs that were not running), actions, rewards
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int p = aw + k * NW;
                store_rows_t(t + 1, p, x[k], ran);
                if (g == 0) {
                    b_act[((size_t)t * N + n) * P + p] = ran ? own[k] : 0;
                    b_rew[((size_t)t * N + n) * P + p] = ran ? rw_own[k] : 0.f;
                }
            }
            if (lead && !ran) {
                b_done[(size_t)(t + 1) * N + n] = 0;
                b_filled[(size_t)t * N + n] = 0.f;
            }
        }
    }
    ACOL_TS_END
    if (valid && running && lead && !ghost) {  // T shorter than the env's limits
#pragma unroll
        for (int p = 0; p < P; ++p) fin_return[(size_t)p * N + n] = ep_ret[p];
        fin_length[n] = len;
        atomicMax(t_max, len);
    }
}

template <class ENV, int H, bool OID, int NW, int HS = 1>
int launch_ac_collect_nw(const typename ENV::Params& q, const float* packs, uint32_t round, int T, int proper_term, float* b_obs, int64_t* b_act,
                         float* b_rew, uint8_t* b_done, float* b_filled, float* fin_return, int32_t* fin_length, int32_t* t_max, hipStream_t st) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0);
    using S = MlpShape<D, H, ENV::A>;
    using PP = PackPlan<S, P, ENV::LDS_MAX>;
    // packs read from global memory (NW > 1, not resident): the LDS holds the env's own bytes only; then the waves' store tiles
    const size_t lds_bytes = acol_tstore<ENV, H, OID, NW>() ? acol_lds_fixed<ENV, H, OID, NW>() + acol_tiles(NW) * (size_t)ObsTile<D>::BYTES
                                                            : ((NW > 1 && !(PP::RESIDENT || PP::A3REG)) ? 0 : PP::LDS_BYTES) + ENV::lds_bytes(q);
    static LdsAttr attr_set;
    if (attr_set.need(lds_bytes)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ac_collect_kernel<ENV, H, OID, NW, HS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set.done(lds_bytes);
    }
    // one env block (16 envs, NW waves) per workgroup while the launch has fewer waves than the chip has SIMDs: the per-step action
    // swap is a workgroup barrier, and with several blocks per workgroup every block waits for the slowest one's step
    const bool one_block = HS > 1 || (NW > 1 && (int64_t)((q.n_envs + 15) / 16) * NW <= 1024);
    const int threads = (one_block || NW * HS > 4) ? 64 * NW * HS : ACOL_BLOCK, per_wg = 16 * (threads / (64 * NW * HS));  // envs per workgroup
    // the actors' forward pass kept for the learner step (AcKeep): first passes over the whole batch only, and the record is laid out for
    // this rollout's (T, B)
    AcKeep keep_here = ac_keep_current();
    if (keep_here.hid != nullptr) {
        MARL_REQUIRE(ac_ghost_current().env_ids == nullptr, "ac_collect: the second pass of a rollout keeps no forward pass");
        MARL_REQUIRE(keep_here.T == T && keep_here.B == q.n_envs && q.n_envs % 16 == 0,
                     "ac_collect: the kept forward pass is laid out for %d x %d rows, the rollout has %d x %d (envs in whole blocks of 16)",
                     keep_here.T, keep_here.B, T, q.n_envs);
    }
    timing_begin(TIMER_COLLECT, st);
    hipLaunchKernelGGL((ac_collect_kernel<ENV, H, OID, NW, HS>), dim3((q.n_envs + per_wg - 1) / per_wg), dim3(threads), lds_bytes, st, q, packs, round, T,
                       proper_term, b_obs, b_act, b_rew, b_done, b_filled, fin_return, fin_length, t_max, ac_ghost_current(), keep_here);
    timing_end(TIMER_COLLECT, st);
    MARL_CHECK_LAUNCH("ac_collect_kernel");
    return 0;
}

template <class ENV, int H, bool OID>
int launch_ac_collect(const typename ENV::Params& q, const AgentMap& am, const float* actor, uint32_t round, int T, int proper_term, float* b_obs, int64_t* b_act,
                      float* b_rew, uint8_t* b_done, float* b_filled, float* fin_return, int32_t* fin_length, int32_t* t_max,
                      hipStream_t st) {
    constexpr int P = ENV::P, D = ENV::D0 + (OID ? P : 0), NWMAX = acol_max_nw<ENV>();
    using S = MlpShape<D, H, ENV::A>;
    MARL_REQUIRE(ENV::lds_bytes(q) <= ENV::LDS_MAX, "collector: the env needs %zu bytes of LDS per workgroup, compiled for %zu", ENV::lds_bytes(q),
                 (size_t)ENV::LDS_MAX);
    (void)hipMemsetAsync(t_max, 0, sizeof(int32_t), st);
    float* packs = nullptr;
    if (launch_fwd_pack<S>(P, am, actor, &packs, st) != 0) return -1;
    // agent-per-wave copies when the launch would leave most SIMDs empty anyway (N / 16 waves on 1024 SIMDs); MARLHIP_ACOL_NW=1 keeps one wave per env block
    static const int forced = getenv("MARLHIP_ACOL_NW") ? atoi(getenv("MARLHIP_ACOL_NW")) : 0;
    // (env.standardise_rewards keeps per-env running records in memory that one wave per env must read and commit in lockstep)
    const bool split = NWMAX > 1 && q.reward_stats == nullptr && (forced ? forced > 1 : q.n_envs <= 8192);  // measured: ahead up to 8192 envs (2 and 4 agents), behind from 16384
#define MARL_ACOL_LAUNCH_ARGS q, (const float*)packs, round, T, proper_term, b_obs, b_act, b_rew, b_done, b_filled, fin_return, fin_length, t_max, st
    if constexpr (NWMAX > 1) {
        // two waves per agent while even agent-per-wave leaves half of the SIMDs idle (see launch_collect); MARLHIP_ACOL_HS=1 turns it off
        if constexpr (P == 2 && (PackPlan<S, P, ENV::LDS_MAX>::RESIDENT || PackPlan<S, P, ENV::LDS_MAX>::A3REG) && S::MT % 4 == 0 && !acol_tstore<ENV, H, OID, NWMAX>()) {
            static const bool hs_off = getenv("MARLHIP_ACOL_HS") != nullptr && atoi(getenv("MARLHIP_ACOL_HS")) == 1;
            if (split && !hs_off && (int64_t)((q.n_envs + 15) / 16) * NWMAX * 2 <= 1024) return launch_ac_collect_nw<ENV, H, OID, NWMAX, 2>(MARL_ACOL_LAUNCH_ARGS);
        }
        if (split) return launch_ac_collect_nw<ENV, H, OID, NWMAX>(MARL_ACOL_LAUNCH_ARGS);
    }
    return launch_ac_collect_nw<ENV, H, OID, 1>(MARL_ACOL_LAUNCH_ARGS);
#undef MARL_ACOL_LAUNCH_ARGS
}

}  // namespace marl
