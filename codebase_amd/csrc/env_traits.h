// What the fused collectors need from an env: per-lane state, reset / step / observation pick, and the per-workgroup LDS
// the env keeps for itself.  Two models: Level-Based Foraging (entity lists in registers, no LDS) and the warehouse
// (agents in registers, the shelf layer of each of the workgroup's 64 envs as bytes in LDS).
#pragma once
#include "collect_common.h"
#include "rware_core.h"

namespace marl {

template <int P_, int F_>
struct LbfEnvT {
    static constexpr int P = P_, D0 = 3 * (P_ + F_), A = 6;
    static constexpr size_t LDS_MAX = 0;  // static bound of lds_bytes(), reserved by the pack plan
    using Params = LbfParams;
    using State = LbfState<P_, F_>;
    struct Ctx {
        __device__ __forceinline__ void init(const Params&, uint8_t*, int, int) {}
    };
    static size_t lds_bytes(const Params&) { return 0; }
    static __device__ __forceinline__ void reset(const Params& q, State& s, Ctx&, uint32_t env, uint32_t episode) {
        DrawStream rng;
        rng.init(q.seed, env, episode, STREAM_RESET);
        lbf_reset(q, s, rng);
    }
    static __device__ __forceinline__ void step(const Params& q, State& s, Ctx&, uint32_t, uint32_t, const int* act, double* raw, bool& done) {
        lbf_step(q, s, act, raw, done);
    }
    static __device__ __forceinline__ int elapsed(const State& s) { return s.step; }
    template <int KS1, bool OID>
    static __device__ __forceinline__ void observe(const Params& q, const State& s, Ctx&, int p, int g, float (&x)[KS1]) {
        LbfObs<P_, F_> o;
        lbf_observe(q, s, p, o);
        pick_obs_id<P_, F_, KS1, OID>(o, p, g, x);
    }
};

template <int P_, int MAXCELLS>
struct RwEnvT {
    static constexpr int P = P_, D0 = RW_OBS_DIM, A = 5;
    static constexpr size_t LDS_MAX = (size_t)MAXCELLS * 64;
    using Params = RwParams;
    using State = RwState<P_>;
    struct Ctx {
        RwGrid grid;
        RwRequested<P_> rq;  // the queue as a bit set, rebuilt whenever the state changes (reset / step)
        // env (wave, j) of the workgroup owns byte column wave*16+j of every cell row; the 4 lanes that carry one env
        // redundantly read and write the same bytes with the same values
        __device__ __forceinline__ void init(const Params&, uint8_t* lds, int wave, int j) { grid.g = lds + wave * 16 + j; grid.stride = 64; }
    };
    static size_t lds_bytes(const Params& q) { return (size_t)q.rows * q.cols * 64; }
    static __device__ __forceinline__ void reset(const Params& q, State& s, Ctx& c, uint32_t env, uint32_t episode) {
        DrawStream rng;
        rng.init(q.seed, env, episode, STREAM_RESET);
        rw_reset(q, s, c.grid, rng);
        c.rq.build(q, s);
    }
    static __device__ __forceinline__ void step(const Params& q, State& s, Ctx& c, uint32_t env, uint32_t episode, const int* act, double* raw,
                                                bool& done) {
        DrawStream req;
        req.init(q.seed, env, episode, STREAM_REQUEST);
        if (rw_step(q, s, c.grid, act, raw, done, req)) c.rq.build(q, s);  // the queue only changes on a delivery
    }
    static __device__ __forceinline__ int elapsed(const State& s) { return s.steps; }
    template <int KS1, bool OID>
    static __device__ __forceinline__ void observe(const Params& q, const State& s, Ctx& c, int p, int g, float (&x)[KS1]) {
        if constexpr (OID) {  // with the ObserveID prefix every index shifts by P: the general per-element route
            const uint64_t word = rw_window_word(q, s, c.grid, c.rq, p);
            constexpr int IDW = P_, D = D0 + IDW;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const int i = 4 * ks + g;  // the lane's own element of this k-step
                float v = 0.f;
                if (i < IDW) v = i == p ? 1.f : 0.f;
                else if (i < D) v = rw_obs_elem_word(q, s, p, word, i - IDW);
                x[ks] = v;
            }
        } else {
            // lane g of an env takes elements 4 ks + g: the 8 own features are two select chains, every window feature one bit of the mask
            const uint64_t bits = rw_obs_bits(q, s, c.grid, c.rq, p);
            const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
            const int ax = rw_pick<P_>(s.ax, p), ay = rw_pick<P_>(s.ay, p), ad = rw_pick<P_>(s.ad, p);
            const int carrying = rw_pick<P_>(s.ac, p) != 0 ? 1 : 0, hw = rw_is_highway(q, ax, ay) ? 1 : 0;
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                int v = 0;
                if (ks == 0) v = g == 0 ? ax : (g == 1 ? ay : (g == 2 ? carrying : (ad == 0 ? 1 : 0)));
                else if (ks == 1) v = g == 3 ? hw : (ad == g + 1 ? 1 : 0);
                else if (4 * ks - 8 < 63) v = (int)((((4 * ks - 8) < 32 ? lo : hi) >> (((4 * ks - 8) & 31) + g)) & 1u);  // 4 ks - 8 is a multiple of 4: + g stays inside the word
                x[ks] = (float)v;
            }
        }
    }
};

}  // namespace marl
