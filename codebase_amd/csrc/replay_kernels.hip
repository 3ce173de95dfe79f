// Episode replay (K3 write, K4 sample-gather) - replaces marlbase/dqn/train.py:19-124.
// Storage is episode-major (one episode = one contiguous record per array); sampling transposes
// B whole episodes into the reference's time-major Batch layout.  HBM-bound byte movement:
// per sampled episode read 4*P*D*(T+1) + P*T*(1+4) + (T+1) + T bytes, write
// 4*P*D*(T+1) + 8*P*T + 4*P*T + 4*(T+1) + 4*T bytes (3,421 B + 3,924 B for 8x8-2p-3f, T=25).
#include <stdlib.h>

#include "common.h"

namespace marl {

// grid (ceil(N * D / (256 * ADD_ITEMS)), P): a thread takes ADD_ITEMS observation elements of one agent, one per block-wide stride (a wave's
// 64 lanes stay on 64 consecutive elements = 4.3 rows: coalesced reads, whole-row stores), 32-bit index arithmetic, every request of the
// thread in front of its first store.  The writes are the replay's D-float rows ((T + 1) * D floats apart per agent): partial lines by
// LAYOUT - the replay is episode-major for the sampler's and the learner's sake, so one time step of N envs is 2 N scattered 60-byte rows
// plus their 1- to 4-byte side records.  Round 5, under the memory-side counters (profiles/r05_hbm_ubench_pmc.md): 390 MB of write requests
// per launch for 138 MB of algorithmic writes = ~6 M partial-sector requests in ~300 us - 20 G requests/s, each to a DRAM page of its own.
// That request rate, not bytes (2 TB/s at the memory side) and not bytes in flight (four elements per thread instead of one: 312 -> 306 us),
// is the ceiling of this kernel: 0.95 TB/s of algorithmic bytes.  The fused collectors, which carry the env-steps/s metric, write whole
// episodes from registers and never come here.
constexpr int ADD_ITEMS = 4;
__global__ __launch_bounds__(256) void replay_add_kernel(marlhip_replay_shape rs, marlhip_replay_buffers rb,
                                                         const int32_t* __restrict__ slot, const int32_t* __restrict__ tt,
                                                         const uint8_t* __restrict__ active, const float* __restrict__ obs,
                                                         const int32_t* __restrict__ actions, const float* __restrict__ rewards,
                                                         const uint8_t* __restrict__ done, int N, int init_only) {
    const int P = rs.n_agents, D = rs.obs_dim, T = rs.max_len, p = blockIdx.y;
    const uint32_t total = (uint32_t)N * (uint32_t)D, stride = gridDim.x * 256u;
    uint32_t nd[ADD_ITEMS];
    int n[ADD_ITEMS], sl[ADD_ITEMS], t[ADD_ITEMS];
    float v[ADD_ITEMS];
    bool ok[ADD_ITEMS];
#pragma unroll
    for (int k = 0; k < ADD_ITEMS; ++k) {  // every request of the thread first (clamped addresses, masks applied at the stores)
        nd[k] = blockIdx.x * 256u + threadIdx.x + (uint32_t)k * stride;
        const uint32_t c = nd[k] < total ? nd[k] : total - 1;
        n[k] = (int)(c / (uint32_t)D);
        sl[k] = slot[n[k]];
        t[k] = init_only ? -1 : tt[n[k]];
        ok[k] = nd[k] < total && (active == nullptr || active[n[k]] != 0);
        v[k] = obs[(size_t)p * N * D + c];
    }
#pragma unroll
    for (int k = 0; k < ADD_ITEMS; ++k) {
        const int row = t[k] + 1;
        if (!ok[k] || row > T) continue;  // reference asserts t < max_episode_length (train.py:74)
        const int d = (int)(nd[k] - (uint32_t)n[k] * (uint32_t)D);
        const size_t sp = (size_t)sl[k] * P + p;
        rb.obs[(sp * (T + 1) + row) * D + d] = v[k];
        if (d == 0 && !init_only) {
            rb.act[sp * T + t[k]] = (uint8_t)actions[(size_t)p * N + n[k]];
            rb.rew[sp * T + t[k]] = rewards[(size_t)p * N + n[k]];
            if (p == 0) {
                rb.done[(size_t)sl[k] * (T + 1) + t[k] + 1] = done[n[k]] ? 1 : 0;
                rb.filled[(size_t)sl[k] * T + t[k]] = 1;
            }
        }
    }
}

__global__ __launch_bounds__(256) void replay_draw_kernel(int batch, int length, uint64_t seed, uint32_t counter,
                                                          int32_t* __restrict__ idx_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    U4 c;
    c.x = (uint32_t)(b >> 2); c.y = counter; c.z = 0; c.w = STREAM_SAMPLE;
    const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int s = b & 3;
    const uint32_t w = s == 0 ? o.x : (s == 1 ? o.y : (s == 2 ? o.z : o.w));
    idx_out[b] = (int32_t)bounded_nr(w, (uint32_t)length);
}

// one thread per OUTPUT element: stores are fully coalesced ([p][t][b][d] is contiguous in (b,d));
// reads are D-float runs of the sampled episodes (L2 / Infinity-Cache resident replay).
__global__ __launch_bounds__(256) void replay_sample_kernel(marlhip_replay_shape rs, marlhip_replay_buffers rb,
                                                            const int32_t* __restrict__ idx, int B, float* __restrict__ obss,
                                                            int64_t* __restrict__ actions, float* __restrict__ rewards,
                                                            float* __restrict__ dones, float* __restrict__ filled) {
    const int P = rs.n_agents, D = rs.obs_dim, T = rs.max_len;
    const int64_t n_obs = (int64_t)P * (T + 1) * B * D;
    const int64_t n_pt = (int64_t)P * T * B;
    const int64_t n_d = (int64_t)(T + 1) * B;
    const int64_t n_f = (int64_t)T * B;
    const int64_t total = n_obs + n_pt + n_d + n_f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_obs) {
            const int d = (int)(i % D);
            const int b = (int)((i / D) % B);
            const int t = (int)((i / ((int64_t)D * B)) % (T + 1));
            const int p = (int)(i / ((int64_t)D * B * (T + 1)));
            obss[i] = rb.obs[(((size_t)idx[b] * P + p) * (T + 1) + t) * D + d];
        } else if (i < n_obs + n_pt) {
            const int64_t k = i - n_obs;
            const int b = (int)(k % B);
            const int t = (int)((k / B) % T);
            const int p = (int)(k / ((int64_t)B * T));
            const size_t src = ((size_t)idx[b] * P + p) * T + t;
            actions[k] = (int64_t)rb.act[src];
            rewards[k] = rb.rew[src];
        } else if (i < n_obs + n_pt + n_d) {
            const int64_t k = i - n_obs - n_pt;
            const int b = (int)(k % B), t = (int)(k / B);
            dones[k] = rb.done[(size_t)idx[b] * (T + 1) + t] ? 1.f : 0.f;
        } else {
            const int64_t k = i - n_obs - n_pt - n_d;
            const int b = (int)(k % B), t = (int)(k / B);
            filled[k] = rb.filled[(size_t)idx[b] * T + t] ? 1.f : 0.f;
        }
    }
}

// LDS-staged transposing gather of the observation block (>= 85 % of the sampled bytes).
// A workgroup owns EB consecutive batch rows: it streams their EB episode records - each one
// contiguous P*(T+1)*D floats in the episode-major replay - into LDS with fully coalesced loads,
// then writes, for every (p,t), the EB*D-float run obss[p][t][b0..b0+EB-1][:] with coalesced stores.
// Both HBM sides move whole lines; the (episode-major -> time-major) transpose happens in LDS.
__global__ __launch_bounds__(256) void replay_sample_obs_kernel(marlhip_replay_shape rs, const float* __restrict__ robs,
                                                                const int32_t* __restrict__ idx, int B, int EB,
                                                                float* __restrict__ obss) {
    extern __shared__ __attribute__((aligned(16))) float lds_ep[];
    const int P = rs.n_agents, D = rs.obs_dim, T = rs.max_len;
    const int E = P * (T + 1) * D;  // floats per episode record
    const int b0 = blockIdx.x * EB;
    const int nb = min(EB, B - b0);
    __shared__ int s_idx[16];
    if (threadIdx.x < nb) s_idx[threadIdx.x] = idx[b0 + threadIdx.x];
    __syncthreads();
    // one flat loop over all EB records (not one loop per record): every thread keeps 8 independent
    // loads in flight, enough outstanding bytes per CU to cover HBM latency
    const int all = nb * E;
#pragma unroll 8
    for (int i = threadIdx.x; i < all; i += 256) {
        const int e = i / E, k = i - e * E;
        lds_ep[i] = robs[(size_t)s_idx[e] * E + k];
    }
    __syncthreads();
    const int run = nb * D;              // contiguous floats per (p,t) in the output
    const int total = P * (T + 1) * run;
    for (int o = threadIdx.x; o < total; o += 256) {
        const int pt = o / run, rem = o - pt * run;
        const int e = rem / D, d = rem - e * D;
        obss[((size_t)pt * B + b0) * D + rem] = lds_ep[e * E + pt * D + d];
    }
}

// 16-byte variant: every episode record is a whole number of float4 and every output run starts 16-byte aligned
// (E % 4 == 0, (B * D) % 4 == 0, (EB * D) % 4 == 0): 4x the bytes in flight per load/store instruction.
// The same workgroup then transposes the episodes' small per-transition records (actions u8 -> i64, rewards, dones,
// filled) through the LDS it just drained: coalesced record reads, EB-element runs on the output side.
__global__ __launch_bounds__(256) void replay_sample_obs4_kernel(marlhip_replay_shape rs, marlhip_replay_buffers rb,
                                                                 const int32_t* __restrict__ idx, int B, int EB,
                                                                 float* __restrict__ obss, int64_t* __restrict__ actions,
                                                                 float* __restrict__ rewards, float* __restrict__ dones,
                                                                 float* __restrict__ filled) {
    const float* __restrict__ robs = rb.obs;
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float lds_ep[];
    const int P = rs.n_agents, D = rs.obs_dim, T = rs.max_len;
    const int E = P * (T + 1) * D, E4 = E >> 2;
    const int b0 = blockIdx.x * EB;
    const int nb = min(EB, B - b0);
    __shared__ int s_idx[16];
    if (threadIdx.x < nb) s_idx[threadIdx.x] = idx[b0 + threadIdx.x];
    __syncthreads();
    // small records ([e][PT act | PT rew | T+1 done | T filled], 2 PT + 2 T + 1 <= E values per episode): requested first,
    // parked in registers until the observation block has left the LDS, so their latency hides under the obs phases
    constexpr int SMAX = 12;  // register slots per thread; further values (large P*T) are loaded late
    const int PT = P * T, REC = 2 * PT + 2 * T + 1;
    auto small_load = [&](int i) -> float {
        const int e = i / REC, k = i - e * REC;
        const size_t ep = (size_t)s_idx[e];
        if (k < PT) return (float)rb.act[ep * PT + k];
        if (k < 2 * PT) return rb.rew[ep * PT + (k - PT)];
        if (k < 2 * PT + T + 1) return rb.done[ep * (T + 1) + (k - 2 * PT)] ? 1.f : 0.f;
        return rb.filled[ep * T + (k - 2 * PT - T - 1)] ? 1.f : 0.f;
    };
    float sm[SMAX];
#pragma unroll
    for (int c = 0; c < SMAX; ++c) {
        const int i = threadIdx.x + 256 * c;
        sm[c] = i < nb * REC ? small_load(i) : 0.f;
    }
    const int all4 = nb * E4;
    f4* l4 = reinterpret_cast<f4*>(lds_ep);
#pragma unroll 4
    for (int i = threadIdx.x; i < all4; i += 256) {
        const int e = i / E4, k = i - e * E4;
        l4[i] = reinterpret_cast<const f4*>(robs + (size_t)s_idx[e] * E)[k];
    }
    __syncthreads();
    const int run = nb * D;  // contiguous floats per (p,t) in the output
    if ((run & 3) == 0) {
        const int run4 = run >> 2, total4 = P * (T + 1) * run4;
        for (int o = threadIdx.x; o < total4; o += 256) {
            const int pt = o / run4, rem = (o - pt * run4) << 2;
            f4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int e = (rem + c) / D, d = (rem + c) - e * D;
                v[c] = lds_ep[e * E + pt * D + d];
            }
            *reinterpret_cast<f4*>(obss + ((size_t)pt * B + b0) * D + rem) = v;
        }
    } else {  // ragged last workgroup
        const int total = P * (T + 1) * run;
        for (int o = threadIdx.x; o < total; o += 256) {
            const int pt = o / run, rem = o - pt * run;
            const int e = rem / D, d = rem - e * D;
            obss[((size_t)pt * B + b0) * D + rem] = lds_ep[e * E + pt * D + d];
        }
    }
    // ---- small records through the drained LDS
    __syncthreads();
#pragma unroll
    for (int c = 0; c < SMAX; ++c) {
        const int i = threadIdx.x + 256 * c;
        if (i < nb * REC) lds_ep[i] = sm[c];
    }
    for (int i = threadIdx.x + 256 * SMAX; i < nb * REC; i += 256) lds_ep[i] = small_load(i);
    __syncthreads();
    for (int o = threadIdx.x; o < nb * REC; o += 256) {
        const int k = o / nb, e = o - k * nb;  // consecutive threads -> consecutive batch rows of one (field, p, t)
        const float v = lds_ep[e * REC + k];
        if (k < PT) actions[(size_t)k * B + b0 + e] = (int64_t)v;
        else if (k < 2 * PT) rewards[(size_t)(k - PT) * B + b0 + e] = v;
        else if (k < 2 * PT + T + 1) dones[(size_t)(k - 2 * PT) * B + b0 + e] = v;
        else filled[(size_t)(k - 2 * PT - T - 1) * B + b0 + e] = v;
    }
}

// the small per-transition arrays of the Batch (actions i64, rewards, dones, filled)
__global__ __launch_bounds__(256) void replay_sample_small_kernel(marlhip_replay_shape rs, marlhip_replay_buffers rb,
                                                                  const int32_t* __restrict__ idx, int B,
                                                                  int64_t* __restrict__ actions, float* __restrict__ rewards,
                                                                  float* __restrict__ dones, float* __restrict__ filled) {
    const int P = rs.n_agents, T = rs.max_len;
    const int64_t n_pt = (int64_t)P * T * B, n_d = (int64_t)(T + 1) * B, n_f = (int64_t)T * B;
    const int64_t total = n_pt + n_d + n_f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_pt) {
            const int b = (int)(i % B), t = (int)((i / B) % T), p = (int)(i / ((int64_t)B * T));
            const size_t src = ((size_t)idx[b] * P + p) * T + t;
            actions[i] = (int64_t)rb.act[src];
            rewards[i] = rb.rew[src];
        } else if (i < n_pt + n_d) {
            const int64_t k = i - n_pt;
            const int b = (int)(k % B), t = (int)(k / B);
            dones[k] = rb.done[(size_t)idx[b] * (T + 1) + t] ? 1.f : 0.f;
        } else {
            const int64_t k = i - n_pt - n_d;
            const int b = (int)(k % B), t = (int)(k / B);
            filled[k] = rb.filled[(size_t)idx[b] * T + t] ? 1.f : 0.f;
        }
    }
}

inline int check_replay(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb) {
    MARL_REQUIRE(rs && rb, "replay: NULL shape/buffers");
    MARL_REQUIRE(rs->capacity > 0 && rs->n_agents > 0 && rs->obs_dim > 0 && rs->max_len > 0, "replay: bad shape");
    MARL_REQUIRE(rb->obs && rb->act && rb->rew && rb->done && rb->filled, "replay: NULL buffer");
    return 0;
}

inline int grid_for(int64_t total) {
    int64_t g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace marl

using namespace marl;

extern "C" int marlhip_replay_init_episode(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* slot,
                                           const uint8_t* active, const float* obs, int32_t n_envs, void* stream) {
    if (check_replay(rs, rb) != 0) return -1;
    MARL_REQUIRE(slot && obs && n_envs > 0, "replay_init_episode: NULL pointer");
    MARL_REQUIRE((int64_t)n_envs * rs->obs_dim < ((int64_t)1 << 31), "replay_init_episode: n_envs * obs_dim must stay below 2^31");
    hipLaunchKernelGGL(replay_add_kernel, dim3((unsigned)(((int64_t)n_envs * rs->obs_dim + 256 * ADD_ITEMS - 1) / (256 * ADD_ITEMS)), rs->n_agents), dim3(256), 0, (hipStream_t)stream,
                       *rs, *rb, slot, (const int32_t*)nullptr, active, obs, (const int32_t*)nullptr, (const float*)nullptr,
                       (const uint8_t*)nullptr, n_envs, 1);
    MARL_CHECK_LAUNCH("replay_init_episode");
    return 0;
}

extern "C" int marlhip_replay_add(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* slot,
                                  const int32_t* t, const uint8_t* active, const float* obs, const int32_t* actions,
                                  const float* rewards, const uint8_t* done, int32_t n_envs, void* stream) {
    if (check_replay(rs, rb) != 0) return -1;
    MARL_REQUIRE(slot && t && obs && actions && rewards && done && n_envs > 0, "replay_add: NULL pointer");
    MARL_REQUIRE((int64_t)n_envs * rs->obs_dim < ((int64_t)1 << 31), "replay_add: n_envs * obs_dim must stay below 2^31");
    hipLaunchKernelGGL(replay_add_kernel, dim3((unsigned)(((int64_t)n_envs * rs->obs_dim + 256 * ADD_ITEMS - 1) / (256 * ADD_ITEMS)), rs->n_agents), dim3(256), 0, (hipStream_t)stream,
                       *rs, *rb, slot, t, active, obs, actions, rewards, done, n_envs, 0);
    MARL_CHECK_LAUNCH("replay_add");
    return 0;
}

// ---- one step of a modular collection round (recurrent Q-networks, Q-networks on the GEMM path): ReplayBuffer.add for the envs that are
// still alive at step t (dqn/train.py:73-89, 219-225) with the env-step call's outputs as they are, then alive &= ~(done | truncated) -
// what the vectorised driver otherwise does with half a dozen tensor operations around marlhip_replay_add
namespace marl {
__global__ __launch_bounds__(256) void replay_add_step_kernel(marlhip_replay_shape rs, marlhip_replay_buffers rb, const int32_t* __restrict__ slot, int t,
                                                              int proper_term, const uint8_t* __restrict__ alive, const float* __restrict__ obs,
                                                              const int32_t* __restrict__ actions, const float* __restrict__ rewards,
                                                              const uint8_t* __restrict__ done, const uint8_t* __restrict__ trunc, int N) {
    const int P = rs.n_agents, D = rs.obs_dim, T = rs.max_len, p = blockIdx.y;
    if (t + 1 > T) return;
    const uint32_t total = (uint32_t)N * (uint32_t)D, stride = gridDim.x * 256u;
    uint32_t nd[ADD_ITEMS];
    int n[ADD_ITEMS], sl[ADD_ITEMS];
    float v[ADD_ITEMS];
    bool ok[ADD_ITEMS];
#pragma unroll
    for (int k = 0; k < ADD_ITEMS; ++k) {  // (as replay_add_kernel: all requests before the first store)
        nd[k] = blockIdx.x * 256u + threadIdx.x + (uint32_t)k * stride;
        const uint32_t c = nd[k] < total ? nd[k] : total - 1;
        n[k] = (int)(c / (uint32_t)D);
        sl[k] = slot[n[k]];
        ok[k] = nd[k] < total && alive[n[k]] != 0;
        v[k] = obs[(size_t)p * N * D + c];
    }
#pragma unroll
    for (int k = 0; k < ADD_ITEMS; ++k) {
        if (!ok[k]) continue;
        const int d = (int)(nd[k] - (uint32_t)n[k] * (uint32_t)D);
        const size_t sp = (size_t)sl[k] * P + p;
        rb.obs[(sp * (T + 1) + t + 1) * D + d] = v[k];
        if (d == 0) {
            rb.act[sp * T + t] = (uint8_t)actions[(size_t)p * N + n[k]];
            rb.rew[sp * T + t] = rewards[(size_t)p * N + n[k]];
            if (p == 0) {
                rb.done[(size_t)sl[k] * (T + 1) + t + 1] = (proper_term ? done[n[k]] != 0 : (done[n[k]] | trunc[n[k]]) != 0) ? 1 : 0;
                rb.filled[(size_t)sl[k] * T + t] = 1;
            }
        }
    }
}
__global__ __launch_bounds__(256) void clear_alive_kernel(int N, uint8_t* __restrict__ alive, const uint8_t* __restrict__ done, const uint8_t* __restrict__ trunc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N && (done[n] | trunc[n]) != 0) alive[n] = 0;
}
}  // namespace marl

extern "C" int marlhip_replay_add_step(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* slot, int32_t t,
                                       int32_t use_proper_termination, uint8_t* alive, const float* obs, const int32_t* actions, const float* rewards,
                                       const uint8_t* done, const uint8_t* truncated, int32_t n_envs, void* stream) {
    if (check_replay(rs, rb) != 0) return -1;
    MARL_REQUIRE(slot && alive && obs && actions && rewards && done && truncated && n_envs > 0 && t >= 0, "replay_add_step: bad argument");
    MARL_REQUIRE((int64_t)n_envs * rs->obs_dim < ((int64_t)1 << 31), "replay_add_step: n_envs * obs_dim must stay below 2^31");
    hipLaunchKernelGGL(replay_add_step_kernel, dim3((unsigned)(((int64_t)n_envs * rs->obs_dim + 256 * ADD_ITEMS - 1) / (256 * ADD_ITEMS)), rs->n_agents), dim3(256), 0,
                       (hipStream_t)stream, *rs, *rb, slot, t, use_proper_termination, (const uint8_t*)alive, obs, actions, rewards, done, truncated, n_envs);
    hipLaunchKernelGGL(clear_alive_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_envs, alive, done, truncated);
    MARL_CHECK_LAUNCH("replay_add_step");
    return 0;
}

extern "C" int marlhip_replay_sample(const marlhip_replay_shape* rs, const marlhip_replay_buffers* rb, const int32_t* idx,
                                     int32_t batch, int32_t length, uint64_t seed, uint32_t counter, int32_t* idx_out, float* obss,
                                     int64_t* actions, float* rewards, float* dones, float* filled, void* stream) {
    if (check_replay(rs, rb) != 0) return -1;
    MARL_REQUIRE(batch > 0 && obss && actions && rewards && dones && filled, "replay_sample: NULL pointer");
    if (idx == nullptr) {
        MARL_REQUIRE(idx_out != nullptr, "replay_sample: idx_out scratch needed when idx is NULL");
        MARL_REQUIRE(length > 0 && length <= rs->capacity, "replay_sample: length %d out of range", length);
        hipLaunchKernelGGL(replay_draw_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, batch, length, seed,
                           counter, idx_out);
        MARL_CHECK_LAUNCH("replay_draw");
        idx = idx_out;
    }
    const int P = rs->n_agents, D = rs->obs_dim, T = rs->max_len;
    // episodes per workgroup: as many as fit 52 KB of LDS, at most 8 (more, smaller workgroups per CU hide the gather latency better
    // than longer output runs help)
    const int ep_bytes = P * (T + 1) * D * (int)sizeof(float);
    static const int lds_kb = getenv("MARLHIP_SAMPLE_LDS_KB") ? atoi(getenv("MARLHIP_SAMPLE_LDS_KB")) : 52;  // 3 x 52 KB <= 160 KB
    static const int eb_max = getenv("MARLHIP_SAMPLE_EB_MAX") ? atoi(getenv("MARLHIP_SAMPLE_EB_MAX")) : 8;  // measured at B = 65536 from a 4.5 GB replay: 8 -> 3.0 TB/s, 16 -> 2.8, 4 -> 2.5
    int EB = (lds_kb * 1024) / ep_bytes;
    if (EB > eb_max) EB = eb_max;
    if (EB >= 4) EB &= ~3;  // whole float4 runs in the output (16-byte stores)
    if (getenv("MARLHIP_SAMPLE_SIMPLE") != nullptr || EB < 1) {  // one-thread-per-element gather (reference variant)
        const int64_t total = (int64_t)P * (T + 1) * batch * D + (int64_t)P * T * batch + (int64_t)(2 * T + 1) * batch;
        timing_begin(TIMER_SAMPLE, (hipStream_t)stream);
        hipLaunchKernelGGL(replay_sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, *rs, *rb, idx, batch,
                           obss, actions, rewards, dones, filled);
        timing_end(TIMER_SAMPLE, (hipStream_t)stream);
        MARL_CHECK_LAUNCH("replay_sample");
        return 0;
    }
    const bool vec4 = ((P * (T + 1) * D) % 4 == 0) && (((int64_t)batch * D) % 4 == 0) && ((EB * D) % 4 == 0) &&
                      getenv("MARLHIP_SAMPLE_SCALAR") == nullptr;
    timing_begin(TIMER_SAMPLE, (hipStream_t)stream);
    if (vec4) {  // one kernel: observations + the small records
        static LdsAttr attr;
        if (attr.need((size_t)EB * ep_bytes)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&replay_sample_obs4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)((size_t)EB * ep_bytes));
            attr.done((size_t)EB * ep_bytes);
        }
        hipLaunchKernelGGL(replay_sample_obs4_kernel, dim3((batch + EB - 1) / EB), dim3(256), (size_t)EB * ep_bytes,
                           (hipStream_t)stream, *rs, *rb, idx, batch, EB, obss, actions, rewards, dones, filled);
        timing_end(TIMER_SAMPLE, (hipStream_t)stream);
        MARL_CHECK_LAUNCH("replay_sample_obs4");
        return 0;
    } else
        hipLaunchKernelGGL(replay_sample_obs_kernel, dim3((batch + EB - 1) / EB), dim3(256), (size_t)EB * ep_bytes,
                           (hipStream_t)stream, *rs, (const float*)rb->obs, idx, batch, EB, obss);
    timing_end(TIMER_SAMPLE, (hipStream_t)stream);
    MARL_CHECK_LAUNCH("replay_sample_obs");
    const int64_t small = (int64_t)P * T * batch + (int64_t)(2 * T + 1) * batch;
    hipLaunchKernelGGL(replay_sample_small_kernel, dim3(grid_for(small)), dim3(256), 0, (hipStream_t)stream, *rs, *rb, idx, batch,
                       actions, rewards, dones, filled);
    MARL_CHECK_LAUNCH("replay_sample_small");
    return 0;
}
