// Philox4x32-10 counter-based RNG for the batched env / collector kernels.
// One stream per (env, episode, stream-id); see oracle/philox.py for the layout.
// The reference has no reproducible stream on this path (python `random` is never
// seeded: marlbase/dqn/model.py:105,109,113), so this is the path's own definition.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define MARL_HD __host__ __device__ __forceinline__
#else
#define MARL_HD inline
#endif

namespace marl {

enum : uint32_t { STREAM_ACT = 0u, STREAM_RESET = 1u, STREAM_SAMPLE = 2u };

struct U4 {
    uint32_t x, y, z, w;
};

MARL_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

MARL_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // both halves of a product from ONE 64-bit multiply (v_mad_u64_u32): the 32-bit integer multiplies are quarter-rate
        // instructions, and a separate mul_hi + mul_lo pair per product made Philox the largest part of an env reset
        const uint64_t p0 = (uint64_t)0xD2511F53u * (uint64_t)c.x, p1 = (uint64_t)0xCD9E8D57u * (uint64_t)c.z;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        U4 n;
        n.x = hi1 ^ c.y ^ k0;
        n.y = lo1;
        n.z = hi0 ^ c.w ^ k1;
        n.w = lo0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// float32 uniform in [0,1): top 24 bits.
MARL_HD float u01_f32(uint32_t w) { return (float)(w >> 8) * (1.0f / 16777216.0f); }

// multiply-shift map onto [0,n) without rejection (per-step random actions).
MARL_HD uint32_t bounded_nr(uint32_t w, uint32_t n) { return mulhi32(w, n); }

// Sequential words of one (env, episode, stream) stream + unbiased bounded draw
// (Lemire multiply-shift with rejection) - the reset kernel's draw source.
struct DrawStream {
    uint32_t k0, k1, env, episode, stream, idx;
    U4 blk;
    uint32_t blk_no;

    MARL_HD void init(uint64_t seed, uint32_t env_id, uint32_t episode_idx, uint32_t stream_id) {
        k0 = (uint32_t)seed;
        k1 = (uint32_t)(seed >> 32);
        env = env_id;
        episode = episode_idx;
        stream = stream_id;
        idx = 0;
        blk_no = 0xFFFFFFFFu;
        blk.x = blk.y = blk.z = blk.w = 0;
    }
    // (Measured and dropped, round 4: generating the first 6 blocks of a reset's stream up front, in lockstep, instead of on demand
    // inside the lanes' out-of-step rejection loops - env-only 759 -> 754 M env-steps/s: the select chain that picks a word out of 24
    // registers costs what the on-demand Philox calls do.)
    MARL_HD uint32_t next_u32() {
        const uint32_t b = idx >> 2;
        if (b != blk_no) {
            U4 c;
            c.x = env; c.y = episode; c.z = b; c.w = stream;
            blk = philox4x32_10(c, k0, k1);
            blk_no = b;
        }
        const uint32_t s = idx & 3u;
        ++idx;
        return s == 0 ? blk.x : (s == 1 ? blk.y : (s == 2 ? blk.z : blk.w));
    }
    // uniform integer in [lo, hi)
    MARL_HD int integers(int lo, int hi) {
        const uint32_t n = (uint32_t)(hi - lo);
        uint64_t m = (uint64_t)next_u32() * n;
        uint32_t l = (uint32_t)m;
        if (l < n) {
            const uint32_t t = (0u - n) % n;
            while (l < t) {
                m = (uint64_t)next_u32() * n;
                l = (uint32_t)m;
            }
        }
        return lo + (int)(m >> 32);
    }
};


// Per-step action noise of env `env` at step `t` of episode `episode` (stream STREAM_ACT):
// word 0 -> u (ONE uniform decides explore-vs-greedy for the whole joint action,
// marlbase/dqn/model.py:105), word 1+p -> random action of agent p.  Words beyond the first
// Philox block come from blocks t | (k << 16).
template <int P>
MARL_HD void act_noise(uint64_t seed, uint32_t env, uint32_t episode, uint32_t t, uint32_t n_actions, float& u, int (&ra)[P]) {
    constexpr int NBLK = (1 + P + 3) / 4;
    uint32_t w[4 * NBLK];
#pragma unroll
    for (int k = 0; k < NBLK; ++k) {
        U4 c;
        c.x = env; c.y = episode; c.z = t | ((uint32_t)k << 16); c.w = STREAM_ACT;
        const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        w[4 * k] = o.x; w[4 * k + 1] = o.y; w[4 * k + 2] = o.z; w[4 * k + 3] = o.w;
    }
    u = u01_f32(w[0]);
#pragma unroll
    for (int p = 0; p < P; ++p) ra[p] = (int)bounded_nr(w[1 + p], n_actions);
}

}  // namespace marl
