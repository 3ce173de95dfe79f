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
        // env (wave, j) of the workgroup owns byte column wave*16+j of every cell row; the 4 lanes that carry one env
        // redundantly read and write the same bytes with the same values
        __device__ __forceinline__ void init(const Params&, uint8_t* lds, int wave, int j) { grid.g = lds + wave * 16 + j; grid.stride = 64; }
    };
    static size_t lds_bytes(const Params& q) { return (size_t)q.rows * q.cols * 64; }
    static __device__ __forceinline__ void reset(const Params& q, State& s, Ctx& c, uint32_t env, uint32_t episode) {
        DrawStream rng;
        rng.init(q.seed, env, episode, STREAM_RESET);
        rw_reset(q, s, c.grid, rng);
    }
    static __device__ __forceinline__ void step(const Params& q, State& s, Ctx& c, uint32_t env, uint32_t episode, const int* act, double* raw,
                                                bool& done) {
        DrawStream req;
        req.init(q.seed, env, episode, STREAM_REQUEST);
        rw_step(q, s, c.grid, act, raw, done, req);
    }
    static __device__ __forceinline__ int elapsed(const State& s) { return s.steps; }
    template <int KS1, bool OID>
    static __device__ __forceinline__ void observe(const Params& q, const State& s, Ctx& c, int p, int g, float (&x)[KS1]) {
        int code[9];
        rw_window(q, s, c.grid, p, code);
        constexpr int IDW = OID ? P_ : 0, D = D0 + IDW;
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            float e[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = 4 * ks + k;  // compile-time element index
                e[k] = i < IDW ? (i == p ? 1.f : 0.f) : (i < D ? rw_obs_elem(q, s, p, code, i - IDW < 0 ? 0 : i - IDW) : 0.f);
            }
            x[ks] = g == 0 ? e[0] : (g == 1 ? e[1] : (g == 2 ? e[2] : e[3]));
        }
    }
};

}  // namespace marl
