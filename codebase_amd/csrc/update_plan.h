// Filled-aware task plan of the LDS-resident learner kernel (dqn_lossgrad_kernel, replay form, MODE 0).
//
// The reference's loss is a filled-weighted sum (marlbase/dqn/model.py:160-163): rows with filled = 0 - the padding behind an episode's
// last transition - contribute exactly zero to the loss and to every gradient entry.  The learner kernel nevertheless walked all T steps
// of every sampled episode: with a trained policy (mean episode length 17.5 of 25 on Foraging-8x8-2p-3f) 30 % of its MFMA time went into
// products with zero.  This header plans the walk instead - ONE launch for all U updates of a round, in front of the update loop:
//   1. workgroup u draws update u's B episode indices (the same Philox stream the kernel would draw from: replay_draw) and reads each
//      episode's stored length (the `filled` prefix of the episode-major replay);
//   2. a STABLE counting sort orders the draws by length, longest first - any permutation of the batch is the same update (a sum over
//      rows), and a stable sort of equal lengths is the identity: a batch of full-length episodes keeps its draw order;
//   3. a 16-episode tile then walks only t < L_tile (= its first episode's length), split into ceil(L_tile / c) chunks with the smallest
//      c for which all chunks fit the launch's waves - short tiles become one short task, long tiles several, every wave gets about
//      the same number of steps;
//   4. the tasks are written as a [slot][wave] table the kernel reads instead of computing (grp, t0, t1) from its indices.
// When every tile is full length the table IS the static plan (same chunks, same wave), so such an update has the bits of the unplanned
// kernel.  Otherwise the sums differ from it in summation order only (exact zeros dropped, chunk boundaries moved) - deterministic,
// inside the bounds of tests/test_gpu_bench_path_vs_oracle.run_case.
#pragma once
#include "common.h"

namespace marl {

struct PlanDims {
    int B, T, ngroups;
    int waves;      // waves per agent of the learner launch (gridDim.x * WAVES)
    int nc_static;  // chunks per tile of the static plan (upd_plan)
    int cap_slots;  // rows of the [slot][wave] table
    int stride;     // int32 per update: [4 header][B sorted episode indices][cap_slots * waves tasks]
    int planned;    // 0: shape outside the planner's limits, the kernel runs its static plan
};

constexpr int PLAN_HDR = 4;          // nslots, chunk length c, longest episode, stored transitions of the batch
constexpr int PLAN_THREADS = 1024;
constexpr int PLAN_MAX_B = 8192;
constexpr int PLAN_MAX_SEGBINS = 12288;  // (T + 1) * ceil(B / 64) LDS counters

inline PlanDims plan_dims(int P, int T, int B, int nwg, int waves_per_wg, int nc_static) {
    PlanDims d = {};
    d.B = B; d.T = T; d.ngroups = (B + 15) / 16;
    d.waves = nwg * waves_per_wg;
    d.nc_static = nc_static;
    const int static_slots = (d.ngroups * nc_static + d.waves - 1) / d.waves;
    const int bal_slots = (d.ngroups + d.waves - 1) / d.waves;
    d.cap_slots = static_slots > bal_slots ? static_slots : bal_slots;
    d.stride = PLAN_HDR + B + d.cap_slots * d.waves;
    const int segs = (B + 63) / 64;
    d.planned = (B >= 256 && B <= PLAN_MAX_B && T <= 255 && d.ngroups <= 65535 && (T + 1) * segs <= PLAN_MAX_SEGBINS) ? 1 : 0;
    (void)P;
    return d;
}

__device__ __forceinline__ int plan_task(int grp, int t0, int t1) { return (grp << 16) | (t0 << 8) | t1; }

// one workgroup per update
static __global__ __launch_bounds__(PLAN_THREADS) void update_plan_kernel(marlhip_replay_buffers rb, uint64_t seed, uint32_t counter0, int length, PlanDims d,
                                                                          int32_t* __restrict__ plan, int32_t* __restrict__ idx_out, int idx_out_update) {
    extern __shared__ int32_t pl_lds[];
    const int B = d.B, T = d.T, nb = T + 1, segs = (B + 63) / 64;
    int32_t* s_idx = pl_lds;                  // [B] drawn episode
    int32_t* s_sorted = s_idx + B;            // [B] episode at sorted position
    int32_t* s_cnt = s_sorted + B;            // [segs][nb] per-segment bin counts -> exclusive prefix inside the bin
    int32_t* s_base = s_cnt + segs * nb;      // [nb] first sorted position of a bin
    int32_t* s_S = s_base + nb;               // [T + 1] tasks needed with chunk length c
    int32_t* s_tile = s_S + nb;               // [ngroups] tile length, then the tile's first task
    uint8_t* s_len = reinterpret_cast<uint8_t*>(s_tile + d.ngroups);   // [B] length in draw order
    uint8_t* s_slen = s_len + ((B + 3) & ~3);                          // [B] length in sorted order
    __shared__ int s_misc[4];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    int32_t* out = plan + (size_t)u * d.stride;
    for (int i = tid; i < segs * nb + 2 * nb; i += PLAN_THREADS) s_cnt[i] = 0;  // (s_cnt, s_base, s_S are contiguous)
    if (tid == 0) { s_misc[0] = T; s_misc[3] = 0; }
    __syncthreads();
    int my_min = T, my_total = 0;
    // ---- 1. draws + lengths; per 64-element segment: rank among the equal lengths to the left, count per length
    for (int i0 = 0; i0 < B; i0 += PLAN_THREADS) {
        const int i = i0 + tid;
        const bool in = i < B;
        int e = 0, len = 0;
        if (in) {
            U4 c;
            c.x = (uint32_t)(i >> 2); c.y = counter0 + (uint32_t)u; c.z = 0; c.w = STREAM_SAMPLE;
            const U4 o = philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
            const int s = i & 3;
            e = (int)bounded_nr(s == 0 ? o.x : (s == 1 ? o.y : (s == 2 ? o.z : o.w)), (uint32_t)length);
            // the episode's `filled` bytes are 0 / 1 (a prefix mask: their count is the stored length): eight at a time through unaligned
            // 64-bit loads (an episode's row starts at e * T bytes: any alignment) + the tail bytes - 4 loads instead of 25 at T = 25
            const uint8_t* f = rb.filled + (size_t)e * T;
            typedef unsigned long long __attribute__((aligned(1))) u64_unaligned;
            int t = 0;
            for (; t + 8 <= T; t += 8) {  // non-zero bytes of the word (any non-zero value counts, as in the learner kernel's own test)
                const unsigned long long w = *reinterpret_cast<const u64_unaligned*>(f + t);
                len += __popcll((((w & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | w) & 0x8080808080808080ull);
            }
            for (; t < T; ++t) len += f[t] ? 1 : 0;
            s_idx[i] = e;
            s_len[i] = (uint8_t)len;
            my_min = len < my_min ? len : my_min;
            my_total += len;
            if (idx_out != nullptr && u == idx_out_update) idx_out[i] = e;
        }
        const int bin = in ? T - len : -1;  // longest first
        // groups of equal bins inside the wave (at most T + 1 rounds; a trained policy has a handful of distinct lengths per 64)
        unsigned long long todo = __ballot(in);
        int rank = 0, cnt = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int v = __shfl(bin, leader);
            const unsigned long long m = __ballot(in && bin == v);
            if (bin == v) {
                rank = __popcll(m & ((1ull << lane) - 1ull));
                cnt = __popcll(m);
                if (lane == leader) s_cnt[(i >> 6) * nb + v] = cnt;
            }
            todo &= ~m;
        }
        if (in) s_sorted[i] = rank;  // (parked until the bases exist)
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const int o = __shfl_xor(my_min, off);
        my_min = o < my_min ? o : my_min;
        my_total += __shfl_xor(my_total, off);
    }
    if (lane == 0) {
        atomicMin(&s_misc[0], my_min);
        atomicAdd(&s_misc[3], my_total);
    }
    __syncthreads();
    const int W = d.waves;
    if (s_misc[0] >= T) {
        // every drawn episode runs to the time limit (a fresh run's regime, the headline's): nothing to order, nothing to skip - the draws
        // in draw order and the static plan, entry for entry (what the sort and the balanced table below would produce as well)
        for (int i = tid; i < B; i += PLAN_THREADS) out[PLAN_HDR + i] = s_idx[i];
        const int nc = d.nc_static, ntasks = d.ngroups * nc;
        for (int i = tid; i < d.cap_slots * W; i += PLAN_THREADS) {
            int e = 0;
            if (i < ntasks) {
                const int grp = i / nc, ch = i - grp * nc;
                e = plan_task(grp, (ch * T) / nc, ((ch + 1) * T) / nc);  // slot = task / W, wave = task % W: the static loop's own assignment
            }
            out[PLAN_HDR + B + i] = e;
        }
        if (tid == 0) {
            out[0] = (ntasks + W - 1) / W;
            out[1] = T;
            out[2] = T;
            out[3] = s_misc[3];
        }
        return;
    }
    // ---- 2. exclusive prefix over the segments inside each bin, then over the bins
    if (tid < nb) {
        int run = 0;
        for (int s = 0; s < segs; ++s) {
            const int c = s_cnt[s * nb + tid];
            s_cnt[s * nb + tid] = run;
            run += c;
        }
        s_base[tid] = run;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int b = 0; b < nb; ++b) {
            const int c = s_base[b];
            s_base[b] = run;
            run += c;
        }
    }
    __syncthreads();
    // ---- 3. scatter (stable: bin base + segments to the left + equal lengths to the left inside the segment)
    int32_t pos_keep[(PLAN_MAX_B + PLAN_THREADS - 1) / PLAN_THREADS];
#pragma unroll
    for (int k = 0; k < (PLAN_MAX_B + PLAN_THREADS - 1) / PLAN_THREADS; ++k) {
        const int i = k * PLAN_THREADS + tid;
        pos_keep[k] = -1;
        if (i < B) {
            const int bin = T - (int)s_len[i];
            pos_keep[k] = s_base[bin] + s_cnt[(i >> 6) * nb + bin] + s_sorted[i];
        }
    }
    __syncthreads();  // (s_sorted held the ranks: every position is formed before the first one is overwritten)
#pragma unroll
    for (int k = 0; k < (PLAN_MAX_B + PLAN_THREADS - 1) / PLAN_THREADS; ++k) {
        const int i = k * PLAN_THREADS + tid;
        if (i < B) {
            s_sorted[pos_keep[k]] = s_idx[i];
            s_slen[pos_keep[k]] = s_len[i];
        }
    }
    __syncthreads();
    for (int i = tid; i < B; i += PLAN_THREADS) out[PLAN_HDR + i] = s_sorted[i];
    // ---- 4. tile lengths (sorted longest first: a tile's first episode is its longest) and the chunk length
    for (int k0 = 0; k0 < d.ngroups; k0 += PLAN_THREADS) {
        const int k = k0 + tid;
        const int L = k < d.ngroups ? (int)s_slen[16 * k] : 0;
        if (k < d.ngroups) s_tile[k] = L;
        if (__ballot(k < d.ngroups) == 0ull) continue;
        for (int c = 1; c <= T; ++c) {  // tasks needed with chunk length c: summed per wave, one atomic per wave and c
            int v = (L + c - 1) / c;
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0) atomicAdd(&s_S[c], v);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const int Lmax = s_tile[0], Lmin = s_tile[d.ngroups - 1];
        const bool full = Lmin >= T;  // every tile walks all T steps: the static plan, bit for bit
        int c = T, rounds = 1;
        if (!full) {
            rounds = (d.ngroups + W - 1) / W;  // a tile is at least one task
            c = T;
            for (int cc = 1; cc <= T; ++cc)
                if (s_S[cc] <= W * rounds) { c = cc; break; }
        }
        s_misc[0] = full ? 1 : 0;
        s_misc[1] = c;
        s_misc[2] = Lmax;
    }
    __syncthreads();
    const bool full = s_misc[0] != 0;
    const int c = s_misc[1];
    // the table starts empty (t0 == t1: the kernel skips the entry)
    for (int i = tid; i < d.cap_slots * W; i += PLAN_THREADS) out[PLAN_HDR + B + i] = 0;
    __syncthreads();
    if (full) {
        const int nc = d.nc_static, ntasks = d.ngroups * nc;
        for (int task = tid; task < ntasks; task += PLAN_THREADS) {
            const int grp = task / nc, ch = task - grp * nc;
            const int t0 = (ch * T) / nc, t1 = ((ch + 1) * T) / nc;
            out[PLAN_HDR + B + task] = plan_task(grp, t0, t1);  // slot = task / W, wave = task % W: the static loop's own assignment
        }
        if (tid == 0) out[0] = (ntasks + W - 1) / W;
    } else {
        // first task of every tile: exclusive prefix of ceil(L / c) (one thread; ngroups is a few hundred)
        if (tid == 0) {
            int run = 0;
            for (int k = 0; k < d.ngroups; ++k) {
                const int m = (s_tile[k] + c - 1) / c;
                s_tile[k] = (s_tile[k] << 16) | run;  // (length | first task)
                run += m;
            }
            s_misc[1] = run;
            out[0] = (run + W - 1) / W;
        }
        __syncthreads();
        for (int k = tid; k < d.ngroups; k += PLAN_THREADS) {
            const int L = s_tile[k] >> 16, first = s_tile[k] & 0xFFFF, m = (L + c - 1) / c;
            for (int jj = 0; jj < m; ++jj) {
                const int task = first + jj, slot = task / W, q = task - slot * W;
                const int wave = (slot & 1) ? W - 1 - q : q;  // snake: a wave's second task comes from the other end of the size order
                out[PLAN_HDR + B + slot * W + wave] = plan_task(k, (jj * L) / m, ((jj + 1) * L) / m);
            }
        }
    }
    if (tid == 0) {
        out[1] = c;
        out[2] = s_misc[2];
        out[3] = s_misc[3];
    }
}

inline size_t plan_lds_bytes(const PlanDims& d) {
    const int nb = d.T + 1, segs = (d.B + 63) / 64;
    return (size_t)(2 * d.B + segs * nb + 2 * nb + d.ngroups) * 4 + 2 * (size_t)((d.B + 3) & ~3);
}

}  // namespace marl
