// IDQN learner step on gfx950 (K5-K8): fused critic/target forward + Double-Q TD target + masked
// MSE + backward on f32 MFMA, deterministic partial-gradient reduction, global-norm clip + Adam +
// target update.  Replaces QNetwork._compute_loss / update (marlbase/dqn/model.py:118-196).
//
// Work decomposition (oracle/mfma_emul.py is the lane-level statement of this file):
//   wave task = (agent p, group of 16 episodes, chunk [t0,t1) of transitions); a workgroup (4
//   waves) serves ONE agent whose critic / target / transposed weights sit in LDS as MFMA
//   A-operand packs.  A task walks time BACKWARDS: at step t it forwards the critic on the 16
//   rows obs[p][t][b0..b0+15], turns the bootstrap value carried from step t+1 into the TD error
//   of transition t, back-propagates that row block immediately (activations never leave
//   registers except for the two LDS transposes the weight-gradient GEMMs need), then forwards
//   the target net on the same rows to produce the bootstrap value for transition t-1.
//   Weight gradients accumulate in MFMA accumulators across all tasks of the wave; the 4 waves
//   fold through LDS and the workgroup writes ONE partial record; a second kernel sums records
//   in fixed order (bitwise reproducible, no float atomics) and applies 1/sum(filled).
// MFMA-bound: 360 v_mfma_f32_16x16x4_f32 per 16-row block at H=64 (96 critic fwd, 96 target
// fwd, 168 backward) = 46.1 kFLOP/row issued vs 43.5 kFLOP/row algorithmic.
#pragma once
#include <stdlib.h>

#include "common.h"
#include "mlp.h"
#include "p2p.h"

namespace marl {

bool p2p_is_builtin(marlhip_exchange_fn fn);  // p2p.hip
}
#include "update_plan.h"
namespace marl {


__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// C-layout registers regs[MT] -> LDS tile[h][TS] (row j of hidden unit h at h * TS + j).  TS = 24 keeps the ds_read_b128 of
// tile_read conflict-free (lane (g, i) reads 4 rows of unit 16mt+i: with TS = 16 the 16 lanes of a read group land 2-way on the
// same banks, with 24 they spread over all 64) and the ds_write_b32 of tile_write at 2-way, which costs nothing extra.
template <int TS, int MT>
__device__ __forceinline__ void tile_write_s(float* tile, const f4 (&regs)[MT], int g, int j) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(16 * mt + 4 * g + r) * TS + j] = regs[mt][r];
}

// lane (g,i) reads tile[16mt+i][4g..4g+3]: operand registers of k-steps ks=0..3 (row 4g+ks)
template <int TS>
__device__ __forceinline__ f4 tile_read_s(const float* tile, int mt, int g, int i) {
    return *reinterpret_cast<const f4*>(tile + (16 * mt + i) * TS + 4 * g);
}

// stride-16 forms (the register-resident kernels of dqn_update_tp.h, gru_bwd.h, qmix.h keep unpadded tiles)
template <int MT>
__device__ __forceinline__ void tile_write(float* tile, const f4 (&regs)[MT], int g, int j) { tile_write_s<16, MT>(tile, regs, g, j); }
__device__ __forceinline__ f4 tile_read(const float* tile, int mt, int g, int i) { return tile_read_s<16>(tile, mt, g, i); }

__device__ __forceinline__ float sum16(float v) {  // over the 16 lanes j of one g
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    return v;
}

template <class S>
struct UpdLds {
    static constexpr int oC = 0, oT = S::NFWD, oB = 2 * S::NFWD, oTiles = oB + S::NBWD;
    static constexpr int REC = S::NPARAM + 2;                 // partial record: grads, loss, n_filled
    static constexpr int STRIP_OFF = (S::NPARAM + 3) / 4 * 4;     // strips start 16-byte aligned behind the weights
    static constexpr int FOLD = STRIP_OFF + (2 * S::H + 18) * 16;  // per-wave fold region (weights + bias/loss strips)
    // per wave: [h2 tile, later the dH1 tile][h1 tile][dH2 tile][dQ tile]; the dH1 tile re-uses the h2 tile, whose only reader
    // (the dW3 operands) has retired long before dH1 exists
    static constexpr int per_wave(int ts) { return 3 * ts * S::H + 16 * ts; }
    // fold regions live at once: one per wave, or two when four copies of the gradient exceed the LDS (the 71-wide warehouse rows: 36 KB
    // each) - the waves then fold in two rounds (the second pair adds onto the first pair's regions)
    static constexpr int fold_waves(int waves) { return waves * FOLD * 4 <= 160 * 1024 ? waves : (waves + 1) / 2; }
    static constexpr int total_ts(int waves, int ts) {
        return (oTiles + waves * per_wave(ts)) > fold_waves(waves) * FOLD ? (oTiles + waves * per_wave(ts)) : fold_waves(waves) * FOLD;
    }
    static constexpr int TS = total_ts(4, 24) * 4 <= 160 * 1024 ? 24 : 16;  // padded tiles when they fit next to the packs
    static constexpr int TILE = TS * S::H;
    static constexpr int PER_WAVE = per_wave(TS);
    // floats: packs + tiles during the walk, the per-wave fold regions (overlaying them) in the epilogue
    static constexpr int total(int waves) { return total_ts(waves, TS); }
    static constexpr bool FITS = total(4) * 4 <= 160 * 1024;  // else the shape runs on the register-resident kernels below
};

// which learner kernels a shape uses: LDS-resident packs (hidden 64, narrow inputs) or weights in registers split over the
// waves (hidden 128; hidden 64 when the packs of a wide first layer do not fit the LDS, e.g. the 71-wide warehouse rows)
template <class S>
constexpr bool use_tp() { return S::H > 64 || !UpdLds<S>::FITS; }

// packs of one agent in the workspace: [critic fwd NFWD][target fwd NFWD][critic bwd NBWD]
template <class S>
__global__ __launch_bounds__(256) void dqn_pack_kernel(const float* __restrict__ params, const float* __restrict__ tparams,
                                                       AgentMap am, float* __restrict__ packs) {
    constexpr int TOT = 2 * S::NFWD + S::NBWD;
    const int p = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= TOT) return;
    const float* w = params + (size_t)am.net[p] * S::NPARAM;
    float v;
    if (idx < S::NFWD) v = mlp_fwd_pack_elem<S>(w, idx);
    else if (idx < 2 * S::NFWD) v = mlp_fwd_pack_elem<S>(tparams + (size_t)am.net[p] * S::NPARAM, idx - S::NFWD);
    else v = mlp_bwd_pack_elem<S>(w, idx - 2 * S::NFWD);
    packs[(size_t)p * TOT + idx] = v;
}

// where the rows come from: a materialised Batch in the reference layout (marlhip_dqn_loss_grad), or the
// episode-major replay itself, gathered in-kernel through sampled episode indices
// (marlhip_dqn_loss_grad_replay: no sample kernel, no Batch round trip through HBM)
struct ReplaySrc {
    marlhip_replay_buffers rb;
    const int32_t* idx;  // [B] or nullptr: Philox(seed; stream 2, counter) draw in [0, length)
    int32_t* idx_out;    // optional record of the indices used
    uint64_t seed;
    uint32_t counter;
    int length;
    int capacity;  // episodes in the replay: when every array is < 2 GB the kernel addresses it through buffer descriptors (32-bit offsets)
    // filled-aware plan of this update (update_plan.h; nullptr: the static plan): [header][B episode indices, longest first][slot][wave] tasks
    const int32_t* plan = nullptr;
};

__device__ __forceinline__ int replay_draw(const ReplaySrc& r, int b) {
    U4 c;
    c.x = (uint32_t)(b >> 2); c.y = r.counter; c.z = 0; c.w = STREAM_SAMPLE;
    const U4 o = philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    const int s = b & 3;
    return (int)bounded_nr(s == 0 ? o.x : (s == 1 ? o.y : (s == 2 ? o.z : o.w)), (uint32_t)r.length);
}

// value-decomposition learners (VDN now, QMIX next) split the step around a MIXER:
//   MODE 1 "qsel": agent networks forward only -> chosen_p[t][b] = Q_p(o_t)[a_t], tqsel_p[t][b] = bootstrap value
//   mixer kernel : (chosen, tqsel, r, done, filled) -> dq_p[t][b] = dL/dchosen_p (unnormalised) and per-row loss
//   MODE 2 "bwd" : agent networks forward (critic only) + backward with the external dq
// MODE 0 is the fused independent-learner step (IDQN), where the "mixer" is the identity per agent.
struct MixBufs {
    float* chosen;   // [P][T][B]
    float* tqsel;    // [P][T][B]
    float* r0;       // [T][B] reward of agent 0 (VDNetwork uses batch.rewards[0], dqn/model.py:228)
    float* dn;       // [T][B] done(t+1)
    float* fl;       // [T][B] filled(t)
    float* dq;       // [T][B] (agent stride 0) or [P][T][B]
    float* lrow;     // [T][B] filled * delta^2
    int dq_agent_stride;
    float* rew_all;     // optional [P][T][B]: every agent's reward (MODE 1 publishes it for the standardising mixer)
    const float* dout;  // MODE 4: [P][T][B][A] external gradient w.r.t. EVERY network output (actor-critic learners)
};

template <int I>
struct IntC { static constexpr int value = I; };

// MARL_STEP_PROF=1 (profiling builds only, scripts/build_variants.py): s_memtime at the region boundaries of a step, summed per
// region and added to prof[0..5] by every wave: 0 row loads + masks, 1 critic forward, 2 target forward (+ TD, tile writes),
// 3 dH2 + dW3, 4 dH1 (+ bootstrap), 5 dW2 + dW1.  Reading the counter drains the LDS queue, so the instrumented kernel runs a
// few percent slower than the product build.
#ifndef MARL_STEP_PROF
#define MARL_STEP_PROF 0
#endif
#if MARL_STEP_PROF
#define MARL_TS_BEGIN unsigned long long ts_ = __builtin_readcyclecounter();
#define MARL_TS(k)                                                      \
    {                                                                   \
        const unsigned long long now_ = __builtin_readcyclecounter();   \
        sp[k] += now_ - ts_;                                            \
        ts_ = now_;                                                     \
    }
#else
#define MARL_TS_BEGIN
#define MARL_TS(k)
#endif

// Schedule of one time step (what runs in the shadow of which MFMAs; one wave per SIMD, so everything that is not an MFMA has
// to hide behind one):
//   A  critic forward(t)         96 MFMA   | row loads of step t-1 in flight
//   B  target forward(t)         96 MFMA   | TD error of transition t -> dQ, tile writes (dQ, h2, h1), first backward operands
//   C  backward(t)              176 MFMA   | bootstrap value for transition t-1 (argmax / gather on permlane swaps), relu masks
//                                          | and tile writes of dH2 / dH1 next to dW3 / the first half of dW2
// The first step of a chunk (t = t1) has no transition (A, B, bootstrap), the last (t = t0) needs no target (A, TD, C).
// STORED (the two-pass form, MODE 1 then MODE 2): the qsel pass leaves the critic's two hidden layers of every transition row (post-relu,
// MFMA C layout: hst[(((p T + t) ngroups + group) 2 MT + layer MT + tile) 64 + lane], 512 bytes per row at hidden 64) and the bwd pass
// reads them back one step ahead next to the rows instead of running the critic forward again - 96 of its 272 MFMAs per row block.
template <class S, int WAVES, bool REPLAY, int MODE, bool STORED = false>
__global__ __launch_bounds__(64 * WAVES, WAVES / 4) void dqn_lossgrad_kernel(const float* __restrict__ packs, marlhip_batch bt, ReplaySrc rs,
                                                                 MixBufs mix, float gamma, int double_q, int n_chunks,
                                                                 float* __restrict__ partials, unsigned long long* prof,
                                                                 f4* __restrict__ hst = nullptr) {
    static_assert(!STORED || MODE == 1 || MODE == 2 || MODE == 4, "stored activations belong to the two-pass forms");
    using L = UpdLds<S>;
    constexpr int MT = S::MT, NT1 = S::DP / 16, D = S::D, H = S::H, A = S::A, TS = L::TS, N1 = S::KS1 / 4;
    constexpr int UPD_BLOCK = 64 * WAVES;
    constexpr bool HAS_TGT = (MODE == 0 || MODE == 1);  // target forward + bootstrap value in this pass
    constexpr bool HAS_BWD = (MODE != 1);               // backward of the row block in this pass
    constexpr bool DB1_FREE = (D % 16) != 0;            // a padding column exists in the dW1 operand
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int p = blockIdx.y;
    const int T = bt.max_len, B = bt.batch;
    const unsigned long long t_begin = prof ? __builtin_readcyclecounter() : 0;

    {   // the three packs were laid out by dqn_pack_kernel exactly as LDS wants them: 16-byte linear copy
        constexpr int TOT4 = (2 * S::NFWD + S::NBWD) / 4;
        const f4* src = reinterpret_cast<const f4*>(packs + (size_t)p * (2 * S::NFWD + S::NBWD));
        f4* dst = reinterpret_cast<f4*>(lds);
        copy_f4_to_lds(src, dst, TOT4, tid, UPD_BLOCK);
    }
    __syncthreads();

    const float* cpk = lds + L::oC;
    const float* tpk = lds + L::oT;
    FwdHead<S> chead, thead;
    if (FwdHead<S>::RESIDENT) {
        chead.load(cpk, lane);
        if (HAS_TGT) thead.load(tpk, lane);
    }
    const f4* T3 = reinterpret_cast<const f4*>(lds + L::oB + S::pT3);
    const f4* T2 = reinterpret_cast<const f4*>(lds + L::oB + S::pT2);
    float* TH2 = lds + L::oTiles + wave * L::PER_WAVE;
    float* TH1 = TH2 + L::TILE;
    float* TG2 = TH1 + L::TILE;
    float* TQ = TG2 + L::TILE;
    float* TG1 = TH2;  // alias (see UpdLds)

    const int P = gridDim.y;
    // row (t, b) of agent p = obss + p * agent stride + (t * B + b) * row stride (defaults: the dqn/train.py Batch)
    const size_t obs_as = bt.obs_agent_stride > 0 ? (size_t)bt.obs_agent_stride : (bt.obs_agent_stride < 0 ? 0 : (size_t)(T + 1) * B * D);
    const size_t obs_rs = bt.obs_row_stride ? (size_t)bt.obs_row_stride : (size_t)D;
    const float* obs_p = REPLAY ? nullptr : bt.obss + (size_t)p * obs_as;
    const int64_t* act_p = (REPLAY || MODE == 4) ? nullptr : bt.actions + (size_t)p * T * B;
    const float* rew_p = (REPLAY || MODE == 4) ? nullptr : bt.rewards + (size_t)p * T * B;

    // buffer descriptors of the replay arrays (raw buffers, 32-bit offsets): usable while the largest array stays below 2 GB
    const long long rp_eps = REPLAY ? rs.capacity : 0, rp_obs_bytes = rp_eps * P * (T + 1) * D * 4;
    const bool buf32 = REPLAY && rp_eps > 0 && rp_obs_bytes < (1ll << 31);
    constexpr int kRaw = 0x00020000;  // raw buffer, dword data format (gfx90a / gfx94x / gfx950)
    const int rp_n = buf32 ? (int)rp_eps : 0;  // (descriptors of an unused path cover nothing)
    const __amdgpu_buffer_rsrc_t r_obs = __builtin_amdgcn_make_buffer_rsrc((void*)rs.rb.obs, 0, rp_n * P * (T + 1) * D * 4, kRaw);
    const __amdgpu_buffer_rsrc_t r_act = __builtin_amdgcn_make_buffer_rsrc((void*)rs.rb.act, 0, rp_n * P * T, kRaw);
    const __amdgpu_buffer_rsrc_t r_rew = __builtin_amdgcn_make_buffer_rsrc((void*)rs.rb.rew, 0, rp_n * P * T * 4, kRaw);
    const __amdgpu_buffer_rsrc_t r_done = __builtin_amdgcn_make_buffer_rsrc((void*)rs.rb.done, 0, rp_n * (T + 1), kRaw);
    const __amdgpu_buffer_rsrc_t r_fill = __builtin_amdgcn_make_buffer_rsrc((void*)rs.rb.filled, 0, rp_n * T, kRaw);

    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 dW1[MT][NT1], dW2[MT][MT], dW3[MT], db1[MT], db2[MT], db3 = zero4;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        dW3[a] = zero4; db1[a] = zero4; db2[a] = zero4;
#pragma unroll
        for (int b = 0; b < MT; ++b) dW2[a][b] = zero4;
#pragma unroll
        for (int b = 0; b < NT1; ++b) dW1[a][b] = zero4;
    }
    float loss_acc = 0.f, nfill_acc = 0.f;
#if MARL_STEP_PROF
    unsigned long long sp[6] = {0, 0, 0, 0, 0, 0};
#endif

    const unsigned long long t_loop_begin = prof ? __builtin_readcyclecounter() : 0;
    const int ngroups = (B + 15) >> 4;
    const int ntasks = ngroups * n_chunks;
    // tasks of this wave: computed (the static plan: every tile walks all T steps in n_chunks chunks) or read from the update's
    // filled-aware table (update_plan.h: a tile walks only the steps its longest episode has, in chunks sized to balance the waves)
    const int waves_all = gridDim.x * WAVES, wave_id = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave);
    const int32_t* ptab = (REPLAY && MODE == 0) ? rs.plan : nullptr;
    const int nslots = ptab != nullptr ? ptab[0] : (ntasks + waves_all - 1) / waves_all;
    for (int slot = 0; slot < nslots; ++slot) {
        int grp, c = 1, t0, t1;
        if (ptab != nullptr) {
            const int e = ptab[PLAN_HDR + B + slot * waves_all + wave_id];
            grp = e >> 16; t0 = (e >> 8) & 255; t1 = e & 255;
        } else {
            const int task = slot * waves_all + wave_id;
            if (task >= ntasks) break;
            grp = task / n_chunks;
            c = task - grp * n_chunks;
            t0 = (c * T) / n_chunks; t1 = ((c + 1) * T) / n_chunks;
        }
        if (t1 <= t0) continue;
        const int b0 = grp * 16;
        const bool rowok = (b0 + j) < B;
        const int bj = rowok ? b0 + j : B - 1;
        // replay form: episode of batch row j (forward operand / scalars) and of rows 4g..4g+3 (dW1 operand)
        int ej = 0, eg[4] = {0, 0, 0, 0};
        if (REPLAY) {
            ej = rs.idx ? rs.idx[bj] : replay_draw(rs, bj);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int row = b0 + 4 * g + ks, rc = row < B ? row : B - 1;
                eg[ks] = rs.idx ? rs.idx[rc] : replay_draw(rs, rc);
            }
            if (rs.idx_out != nullptr && p == 0 && c == 0 && g == 0 && rowok) rs.idx_out[bj] = ej;
        }
        float tq_next = 0.f;
        // rows of time step t in the two operand shapes the step needs + the transition's scalars;
        // issued one step ahead so the loads fly under the previous step's MFMAs
        struct Rows {
            float x[S::KS1];   // forward B operand: X[row j][4ks+g]
            float bx[NT1][4];  // dW1 B operand:     X[row 4g+ks][16nt+j]
            int a_sel;
            float rw, dn, fl;
            int dn_raw, fl_raw;  // replay form: the done / filled BYTES as loaded; turned into 0.f / 1.f by mask_rows one step later -
                                 // converting them in load_rows would make every step wait for loads it has only just issued
            float dq, lr;      // MODE 2: external dL/dchosen and per-row loss
            float dqv[4];      // MODE 4: external dL/d(output 4g+r)
            float mk[4];       // batch.action_mask of outputs 4g+r at this observation (1 = allowed); all ones without masks
            f4 hs[(STORED && (MODE == 2 || MODE == 4)) ? 2 * S::MT : 1];  // STORED bwd pass: the network's h1 | h2 tiles of this row block and time step
        };
        // branch-free: every address is clamped in-bounds and loaded unconditionally (a guarded load is
        // an exec-masked branch + a conservative vmcnt(0) at the join); masks are applied at the point of use
        // Replay rows through buffer descriptors when the arrays allow 32-bit offsets: address = descriptor base + per-lane
        // offset (fixed for the whole task, one VGPR) + scalar offset (the time step, SALU) + immediate - no vector address
        // arithmetic per step.  Arrays of 2 GB and more take the 64-bit global path below.
        int vx = 0, vxl = 0, vbx0[4] = {0, 0, 0, 0}, vbxl[4] = {0, 0, 0, 0}, vsc = 0, vdn = 0, vfl = 0;
        if (REPLAY && buf32) {
            const int ebase = (ej * P + p) * (T + 1) * D * 4;
            vx = ebase + 4 * g;
            vxl = ebase + 4 * ((4 * (S::KS1 - 1) + g) < D ? (4 * (S::KS1 - 1) + g) : D - 1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int gb = (eg[ks] * P + p) * (T + 1) * D * 4;
                vbx0[ks] = gb + 4 * j;
                vbxl[ks] = gb + 4 * ((16 * (NT1 - 1) + j) < D ? (16 * (NT1 - 1) + j) : D - 1);
            }
            vsc = (ej * P + p) * T;
            vdn = ej * (T + 1) + 1;
            vfl = ej * T;
        }
        auto load_rows = [&](int t, Rows& R) {
            const int tt = t < T ? t : T - 1;
            if (REPLAY && buf32) {
                const int so = t * D * 4;
#pragma unroll
                for (int ks = 0; ks + 1 < S::KS1; ++ks)
                    R.x[ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, vx + 16 * ks, so, 0));
                R.x[S::KS1 - 1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, vxl, so, 0));
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int nt = 0; nt + 1 < NT1; ++nt)
                        R.bx[nt][ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, vbx0[ks] + 64 * nt, so, 0));
                    R.bx[NT1 - 1][ks] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_obs, vbxl[ks], so, 0));
                }
                R.a_sel = (int)__builtin_amdgcn_raw_buffer_load_b8(r_act, vsc, tt, 0);
                R.rw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_rew, 4 * vsc, 4 * tt, 0));
                R.dn_raw = (int)__builtin_amdgcn_raw_buffer_load_b8(r_done, vdn, tt, 0);
                R.fl_raw = (int)__builtin_amdgcn_raw_buffer_load_b8(r_fill, vfl, tt, 0);
            } else if (REPLAY) {
                const float* xrow = rs.rb.obs + (((size_t)ej * P + p) * (T + 1) + t) * D;
#pragma unroll
                for (int ks = 0; ks < S::KS1; ++ks) {
                    const int d = 4 * ks + g;
                    R.x[ks] = xrow[d < D ? d : D - 1];
                }
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int d = 16 * nt + j;
                        R.bx[nt][ks] = rs.rb.obs[(((size_t)eg[ks] * P + p) * (T + 1) + t) * D + (d < D ? d : D - 1)];
                    }
                R.a_sel = (int)rs.rb.act[((size_t)ej * P + p) * T + tt];
                R.rw = rs.rb.rew[((size_t)ej * P + p) * T + tt];
                R.dn_raw = rs.rb.done[(size_t)ej * (T + 1) + tt + 1];
                R.fl_raw = rs.rb.filled[(size_t)ej * T + tt];
            } else {
                const float* xrow = obs_p + ((size_t)t * B + bj) * obs_rs;
#pragma unroll
                for (int ks = 0; ks < S::KS1; ++ks) {
                    const int d = 4 * ks + g;
                    R.x[ks] = xrow[d < D ? d : D - 1];
                }
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int row = b0 + 4 * g + ks, d = 16 * nt + j;
                        R.bx[nt][ks] = obs_p[((size_t)t * B + (row < B ? row : B - 1)) * obs_rs + (d < D ? d : D - 1)];
                    }
                if (MODE != 4) {
                    R.a_sel = (int)act_p[(size_t)tt * B + bj];
                    R.rw = rew_p[(size_t)tt * B + bj];
                    R.dn = bt.dones[(size_t)(tt + 1) * B + bj];
                }
                R.fl = bt.filled[(size_t)tt * B + bj];
            }
            if (!REPLAY && (MODE == 0 || MODE == 1) && bt.action_mask != nullptr) {
                const float* mrow = bt.action_mask + (((size_t)p * (T + 1) + t) * B + bj) * A;
#pragma unroll
                for (int r = 0; r < 4; ++r) R.mk[r] = mrow[4 * g + r < A ? 4 * g + r : A - 1];
            }
            if (MODE == 2) {
                R.dq = mix.dq[(size_t)p * mix.dq_agent_stride + (size_t)tt * B + bj];
                R.lr = mix.lrow[(size_t)tt * B + bj];
                if constexpr (STORED) {
                    const f4* hp = hst + ((((size_t)p * T + tt) * ngroups + grp) * (2 * MT)) * 64 + lane;
#pragma unroll
                    for (int k = 0; k < 2 * MT; ++k) R.hs[k] = hp[k * 64];
                }
            }
            if (MODE == 4) {
                if constexpr (STORED) {  // (left by the actor-critic step's forward-rows pass, a2c_core.h)
                    const f4* hp = hst + ((((size_t)p * T + tt) * ngroups + grp) * (2 * MT)) * 64 + lane;
#pragma unroll
                    for (int k = 0; k < 2 * MT; ++k) R.hs[k] = hp[k * 64];
                }
                const float* drow = mix.dout + (((size_t)p * T + tt) * B + bj) * A;
#pragma unroll
                for (int r = 0; r < 4; ++r) R.dqv[r] = drow[4 * g + r < A ? 4 * g + r : A - 1];
                R.lr = mix.lrow[(size_t)tt * B + bj];
            }
        };
        // Padding needs no zeroing: observation columns >= D meet zero weights in the forward pack and land in dW1 columns the fold
        // drops; batch rows >= B carry filled = 0, hence dQ = 0 and zero gradient rows whatever their (finite, clamped-address)
        // observations are.  The first padding column of the dW1 operand is set to ONE instead: dW1[h][D] then accumulates
        // sum_rows dH1[row][h] = db1 inside the MFMAs (DB1_FREE; shapes without padding keep the VALU adds).
        auto mask_rows = [&](Rows& R) {
            if (DB1_FREE) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) R.bx[D / 16][ks] = (j == D % 16) ? 1.f : R.bx[D / 16][ks];
            }
            if (REPLAY) {
                R.dn = R.dn_raw ? 1.f : 0.f;
                R.fl = R.fl_raw ? 1.f : 0.f;
            }
            R.fl = rowok ? R.fl : 0.f;
        };

        // FIRST: t == t1, no transition to learn from (forwards + bootstrap only); LAST: t == t0, nobody needs its bootstrap value.
        // cur: rows of step t (loaded one step ago); nxt: receives the rows of step t-1.  The two buffers alternate between
        // steps (the walk is unrolled by two), so no register copies - and no waits for loads that were only just issued -
        // sit between steps.
        auto step = [&](auto first_c, auto last_c, const int t, Rows& cur, Rows& nxt) {
            constexpr bool FIRST = decltype(first_c)::value != 0, LAST = decltype(last_c)::value != 0;
            constexpr bool DO_TGT = HAS_TGT && !LAST, DO_BWD = HAS_BWD && !FIRST, DO_PUB = MODE == 1 && !FIRST;
            MARL_TS_BEGIN
            if (!LAST) load_rows(t - 1, nxt);
            mask_rows(cur);
            MARL_TS(0)
            f4 h1[MT], h2[MT], q, qb, tq = zero4, tqb = zero4;
            // ---- A: critic forward (q arrives as two partial chains, added where it is first used)
            if constexpr (STORED && (MODE == 2 || MODE == 4)) {  // (an earlier pass ran the forward on these rows; Q is not needed here)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    h1[mt] = cur.hs[mt];
                    h2[mt] = cur.hs[MT + mt];
                }
                q = zero4;
                qb = zero4;
            } else {
                mlp_forward_f<S>(cpk, chead, lane, cur.x, h1, h2, q, qb, [](int, int) {});
            }
            if constexpr (STORED && MODE == 1) {
                if (!FIRST) {  // a transition row (t < T): leave its hidden layers for the bwd pass
                    f4* hp = hst + ((((size_t)p * T + t) * ngroups + grp) * (2 * MT)) * 64 + lane;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        hp[mt * 64] = h1[mt];
                        hp[(MT + mt) * 64] = h2[mt];
                    }
                }
            }
            MARL_TS(1)
            // ---- the critic's epilogue, emitted as fillers of the target forward's MFMA groups (or on its own without one)
            f4 dQ[1] = {zero4};
            f4 t3[MT], t2[2][MT], aQ = zero4, bH2[MT];
            // MODE 0 / 2: dQ has ONE non-zero per row (the chosen action), so dH2 = W3^T dQ^T is row a_sel of W3 times that
            // value: 4 LDS reads + 16 multiplies instead of 16 MFMAs whose operand is 15/16 zeros (and no accumulator read-out)
            constexpr bool ONE_HOT = (MODE == 0 || MODE == 2);
            f4 w3sel[MT];
            float dqs_row = 0.f;
            // (k, e): MFMA group k of the target forward, e = -1 the VALU burst in front of it, e = 0..3 LDS traffic in front of its quarters
            auto epilogue = [&](int k, int e) {
                if (k == 0 && e == -1) {
                    q += qb;
                    if (DO_PUB) {
                        // qsel pass: publish Q_p(o_t)[a_t] and (agent 0) the transition's scalars for the mixer
                        const float ch = gather_rows_pl<A>(q, lane, cur.a_sel);
                        if (g == 0 && rowok) {
                            mix.chosen[((size_t)p * T + t) * B + bj] = ch;
                            if (mix.rew_all != nullptr) mix.rew_all[((size_t)p * T + t) * B + bj] = cur.rw;
                            if (p == 0) {
                                mix.r0[(size_t)t * B + bj] = cur.rw;
                                mix.dn[(size_t)t * B + bj] = cur.dn;
                                mix.fl[(size_t)t * B + bj] = cur.fl;
                            }
                        }
                    }
                    if (DO_BWD) {
                        // ---- TD error of transition t (model.py:129,152,160-163)
                        const int a_sel = cur.a_sel;
                        const float fl = cur.fl;
                        float dqs = 0.f;
                        if (ONE_HOT) mlp_w3_row<S>(cpk, lane, a_sel < A ? (a_sel < 0 ? 0 : a_sel) : A - 1, w3sel);
                        if (MODE == 0) {
                            const float y = cur.rw + gamma * tq_next * (1.f - cur.dn);
                            const float delta = gather_rows_pl<A>(q, lane, a_sel) - y;
                            loss_acc += fl * delta * delta;  // every g lane of row j carries the same sums; the fold reads g == 0
                            nfill_acc += fl;
                            dqs = 2.f * fl * delta;
                        } else {  // the mixer already formed dL/dchosen; agent 0's rows carry the loss bookkeeping
                            dqs = rowok ? cur.dq : 0.f;
                            const bool book = p == 0 && rowok;
                            loss_acc += book ? cur.lr : 0.f;
                            nfill_acc += book ? fl : 0.f;
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if (MODE == 4) dQ[0][r] = (rowok && 4 * g + r < A) ? cur.dqv[r] : 0.f;
                            else dQ[0][r] = (4 * g + r == a_sel) ? dqs : 0.f;
                        }
                        db3 += dQ[0];
                        dqs_row = dqs;
                    }
                }
                if (DO_BWD && e >= 0) {  // one hidden tile per quarter
                    const int mt = e;
                    if (k == 1) {
                        if (e == 0) {
                            wave_lds_fence();  // the previous step's tile reads precede these writes in program order
                            tile_write_s<TS, 1>(TQ, dQ, g, j);
                        }
                        if (mt < MT) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) TH2[(16 * mt + 4 * g + r) * TS + j] = h2[mt][r];
                        }
                    }
                    if (k == 2 && mt < MT) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) TH1[(16 * mt + 4 * g + r) * TS + j] = h1[mt][r];
                    }
                    if (k == 3) {
                        if (e == 0) {
                            wave_lds_fence();
                            aQ = tile_read_s<TS>(TQ, 0, g, j);
                        }
                        if (mt < MT) {
                            if (!ONE_HOT) t3[mt] = T3[mt * 64 + lane];
                            t2[0][mt] = T2[(mt * MT + 0) * 64 + lane];
                            bH2[mt] = tile_read_s<TS>(TH2, mt, g, j);
                        }
                    }
                }
            };
            // ---- B: target forward (its hidden activations are scratch).  Under Double-Q only ONE target output per row is
            // needed (the action the online network picks): the output layer is then a dot product in the bootstrap stage
            f4 g2[MT];
            if (DO_TGT) {
                f4 g1[MT];
                static_assert(N1 + MT + 1 >= 4, "epilogue stages need four MFMA groups");
                if (double_q) mlp_forward_f<S, false>(tpk, thead, lane, cur.x, g1, g2, tq, tqb, epilogue);
                else mlp_forward_f<S, true>(tpk, thead, lane, cur.x, g1, g2, tq, tqb, epilogue);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    epilogue(k, -1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) epilogue(k, e);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            MARL_TS(2)
            // ---- bootstrap value for transition t-1 (model.py:132-145): a filler of the backward's MFMA groups
            auto bootstrap = [&]() {
                const bool masked = !REPLAY && bt.action_mask != nullptr;
                if (double_q) {  // the online network picks the action (model.py:138-145), the target network values it
                    if (masked) {  // model.py:136-142: disallowed actions of o_t read as -1e8
#pragma unroll
                        for (int r = 0; r < 4; ++r) q[r] = cur.mk[r] == 0.f ? -1e8f : q[r];
                    }
                    const int a_p = argmax_rows_pl<A>(q, lane);
                    tq_next = mlp_output_at<S>(tpk, lane, a_p, g2);  // the same a_p in all four lanes of a row
                    if (masked) {  // every action disallowed: the reference's target reads -1e8 there too
                        f4 mk4 = {cur.mk[0], cur.mk[1], cur.mk[2], cur.mk[3]};
                        tq_next = gather_rows_pl<A>(mk4, lane, a_p) == 0.f ? -1e8f : tq_next;
                    }
                } else {
                    tq += tqb;
                    if (masked) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) tq[r] = cur.mk[r] == 0.f ? -1e8f : tq[r];
                    }
                    const int a_p = argmax_rows_pl<A>(tq, lane);
                    tq_next = gather_rows_pl<A>(tq, lane, a_p);
                }
                if (MODE == 1 && g == 0 && rowok) mix.tqsel[((size_t)p * T + (t - 1)) * B + bj] = tq_next;
            };
            if (DO_BWD) {
                // ---- C: backward of row block t.  Regions are fenced with sched_barrier; inside a region the listed VALU / LDS
                // work has no dependence on the region's MFMAs, so it issues in their shadow.
                // C0: dH2^T = W3^T dQ^T
                f4 dH2[MT];
                if (ONE_HOT) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dH2[mt] = w3sel[mt] * dqs_row;
                } else {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) dH2[mt] = zero4;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) dH2[mt] = MARL_MFMA(t3[mt][r], dQ[0][r], dH2[mt]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // C1: relu mask, db2, publish the dH2 tile | dW3[a][h2] += dQ^T H2; request the second W2^T step and the h1 tile
                f4 bH1[MT], aG2[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dH2[mt][r] = h2[mt][r] > 0.f ? dH2[mt][r] : 0.f;
                    db2[mt] += dH2[mt];
                }
                tile_write_s<TS, MT>(TG2, dH2, g, j);
#pragma unroll
                for (int m1 = 0; m1 < MT; ++m1) t2[1][m1] = T2[(m1 * MT + 1) * 64 + lane];
#pragma unroll
                for (int nt = 0; nt < MT; ++nt) bH1[nt] = tile_read_s<TS>(TH1, nt, g, j);
                MARL_VB()
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int nt = 0; nt < MT; ++nt) dW3[nt] = MARL_MFMA(aQ[ks], bH2[nt][ks], dW3[nt]);
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_fence();
                MARL_TS(3)
                // C2..: dH1^T = W2^T dH2^T in MT steps (m2); the bootstrap value rides in the first
                f4 dH1[MT];
#pragma unroll
                for (int m1 = 0; m1 < MT; ++m1) dH1[m1] = zero4;
#pragma unroll
                for (int m2 = 0; m2 < MT; ++m2) {
                    const int cb = m2 & 1;
                    if (m2 == 0) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) aG2[mt] = tile_read_s<TS>(TG2, mt, g, j);
                        if (DO_TGT) bootstrap();
                        MARL_VB()
                    }
                    if constexpr (MARL_MFMA_VGPR && MT == 4) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            mfma4_v(dH1, t2[cb][0][r], t2[cb][1][r], t2[cb][2][r], t2[cb][3][r], dH2[m2][r]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if (m2 == MT - 1) mfma_settle(dH1);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
#pragma unroll
                            for (int m1 = 0; m1 < MT; ++m1) dH1[m1] = MARL_MFMA(t2[cb][m1][r], dH2[m2][r], dH1[m1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (m2 + 2 < MT) {
#pragma unroll
                        for (int m1 = 0; m1 < MT; ++m1) t2[cb][m1] = T2[(m1 * MT + m2 + 2) * 64 + lane];
                    }
                }
                MARL_TS(4)
                // C6: relu mask, db1, publish the dH1 tile (over the retired h2 tile) | first half of dW2[h2][h1] += dH2^T H1
#pragma unroll
                for (int m1 = 0; m1 < MT; ++m1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dH1[m1][r] = h1[m1][r] > 0.f ? dH1[m1][r] : 0.f;
                    if (!DB1_FREE) db1[m1] += dH1[m1];
                }
                tile_write_s<TS, MT>(TG1, dH1, g, j);
                MARL_VB()
#pragma unroll
                for (int mt = 0; mt < MT / 2; ++mt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int nt = 0; nt < MT; ++nt) dW2[mt][nt] = MARL_MFMA(aG2[mt][ks], bH1[nt][ks], dW2[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
                wave_lds_fence();
                // C7: request the dH1 tile | second half of dW2
                f4 aG1[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) aG1[mt] = tile_read_s<TS>(TG1, mt, g, j);
#pragma unroll
                for (int mt = MT / 2; mt < MT; ++mt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int nt = 0; nt < MT; ++nt) dW2[mt][nt] = MARL_MFMA(aG2[mt][ks], bH1[nt][ks], dW2[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
                // C8: dW1[h1][d] += dH1^T X   (B operand prefetched from global one step ahead)
#pragma unroll
                for (int nt = 0; nt < NT1; ++nt)
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) dW1[mt][nt] = MARL_MFMA(aG1[mt][ks], cur.bx[nt][ks], dW1[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
                MARL_TS(5)
            } else if (DO_TGT) {
                bootstrap();
            }
        };

        Rows ra, rb;
        if (HAS_TGT) {
            load_rows(t1, ra);
            step(IntC<1>{}, IntC<0>{}, t1, ra, rb);
            int t = t1 - 1;
            for (; t - 1 > t0; t -= 2) {
                step(IntC<0>{}, IntC<0>{}, t, rb, ra);
                step(IntC<0>{}, IntC<0>{}, t - 1, ra, rb);
            }
            if (t > t0) {
                step(IntC<0>{}, IntC<0>{}, t, rb, ra);
                step(IntC<0>{}, IntC<1>{}, t0, ra, rb);
            } else {
                step(IntC<0>{}, IntC<1>{}, t0, rb, ra);
            }
        } else {  // no target in this pass: every step is critic forward + backward; rows t1-1 .. t0
            load_rows(t1 - 1, ra);
            for (int t = t1 - 1; t >= t0; --t) {
                load_rows(t > t0 ? t - 1 : t0, rb);
                step(IntC<0>{}, IntC<1>{}, t, ra, rb);
                ra = rb;
            }
        }
    }

    const unsigned long long t_loop_end = prof ? __builtin_readcyclecounter() : 0;
    if (MODE == 1) return;  // forward-only pass: nothing to fold
    // ---- fold the waves through LDS and write ONE partial record per workgroup.  Every wave stores its
    // accumulators into its own LDS region in parallel (weights at their canonical index, bias / loss
    // partials as [value][16 lanes] strips), one barrier, then all threads sum the regions in a fixed
    // order - no cross-lane shuffles, no serialisation between waves, bitwise reproducible.
    constexpr int FW = L::fold_waves(WAVES);  // fold regions (WAVES, or WAVES / 2 in two rounds)
    __syncthreads();
#pragma unroll
    for (int round = 0; round < WAVES / FW; ++round) {
        if (wave / FW == round) {
            float* mine = lds + (size_t)(wave % FW) * L::FOLD;
            float* strips = mine + L::STRIP_OFF;  // [(2H + 16) bias rows + 2 loss rows][16]
            auto put = [&](float& slot, float v) { slot = round == 0 ? v : slot + v; };  // (second round: this lane's own slot of the first round's copy)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = 16 * mt + 4 * g + r;
#pragma unroll
                    for (int nt = 0; nt < NT1; ++nt) {
                        const int d = 16 * nt + j;
                        if (d < D) put(mine[S::oW1 + o * D + d], dW1[mt][nt][r]);
                    }
#pragma unroll
                    for (int nt = 0; nt < MT; ++nt) put(mine[S::oW2 + o * H + 16 * nt + j], dW2[mt][nt][r]);
                    put(strips[o * 16 + j], DB1_FREE ? (j == D % 16 ? dW1[mt][D / 16][r] : 0.f) : db1[mt][r]);
                    put(strips[(H + o) * 16 + j], db2[mt][r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int a = 4 * g + r;
#pragma unroll
                for (int nt = 0; nt < MT; ++nt)
                    if (a < A) put(mine[S::oW3 + a * H + 16 * nt + j], dW3[nt][r]);
                put(strips[(2 * H + a) * 16 + j], db3[r]);
            }
            if (g == 0) {
                put(strips[(2 * H + 16) * 16 + j], loss_acc);
                put(strips[(2 * H + 17) * 16 + j], nfill_acc);
            }
        }
        __syncthreads();
    }
    float* rec = partials + ((size_t)p * gridDim.x + blockIdx.x) * L::REC;
    // plain elements: sum of the four regions in wave order, 16 bytes per thread and step (the bias / loss slots are written by
    // the strip pass below; what this pass leaves there is overwritten)
    if constexpr (L::REC % 4 == 0 && L::FOLD % 4 == 0) {
        for (int i4 = tid; i4 < L::REC / 4; i4 += UPD_BLOCK) {
            f4 acc = reinterpret_cast<const f4*>(lds)[i4];
#pragma unroll
            for (int w = 1; w < FW; ++w) acc += reinterpret_cast<const f4*>(lds + (size_t)w * L::FOLD)[i4];
            reinterpret_cast<f4*>(rec)[i4] = acc;
        }
    } else {
        for (int i = tid; i < L::REC; i += UPD_BLOCK) {
            float acc = lds[i];
#pragma unroll
            for (int w = 1; w < FW; ++w) acc += lds[(size_t)w * L::FOLD + i];
            rec[i] = acc;
        }
    }
    __syncthreads();  // the strip pass overwrites slots the plain pass has just stored (same workgroup, global memory)
    // strips: b1 (H), b2 (H), b3 (16 slots, A used), loss, n_filled - one thread per strip, 16 lane partials per wave region
    for (int sidx = tid; sidx < 2 * H + 18; sidx += UPD_BLOCK) {
        int i = -1;
        if (sidx < H) i = S::ob1 + sidx;
        else if (sidx < 2 * H) i = S::ob2 + (sidx - H);
        else if (sidx < 2 * H + 16) i = (sidx - 2 * H) < A ? S::ob3 + (sidx - 2 * H) : -1;
        else i = S::NPARAM + (sidx - 2 * H - 16);
        if (i < 0) continue;
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < FW; ++w) {
            const f4* sp = reinterpret_cast<const f4*>(lds + (size_t)w * L::FOLD + L::STRIP_OFF + sidx * 16);
            float t = 0.f;
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4) {
                const f4 v = sp[k4];
                t += v.x; t += v.y; t += v.z; t += v.w;
            }
            acc += t;
        }
        rec[i] = acc;
    }
    if (prof != nullptr && lane == 0) {
        const unsigned long long t_end = __builtin_readcyclecounter();
#if MARL_STEP_PROF
#pragma unroll
        for (int k = 0; k < 6; ++k) atomicAdd(&prof[k], sp[k]);
#endif
        atomicAdd(&prof[8], t_loop_begin - t_begin);    // pack staging
        atomicAdd(&prof[9], t_loop_end - t_loop_begin);  // whole task loop
        atomicAdd(&prof[10], t_end - t_loop_end);        // fold + record write
        atomicAdd(&prof[11], t_end - t_begin);
    }
}

// grad[p][i] = (sum over the agent's records) / n_filled ; loss = sum of all loss fields / n_filled.
// n_filled comes from agent 0's records only (every agent sees the same filled mask).
// sumsq_out (may be null): per-block sum of squares of the block's 64 gradient values, for the clip norm of adam_pack_kernel
struct ReduceP2p {  // the in-library exchange folded into the reduce launch (marlhip_idqn_update_n_dist with marlhip_p2p_allreduce)
    P2pPeers peers;
    int rank, world, max_chunks;
    int64_t slot_floats;
    uint32_t epoch;
    long long timeout_ticks;
};

constexpr int REDUCE_BATCH = 32;
__device__ __forceinline__ void dqn_reduce_body(const float* __restrict__ partials, int P, int nwg, int nparam, const AgentMap& am,
                                                float* __restrict__ grad, float* __restrict__ loss, float* __restrict__ sumsq_out,
                                                const ReduceP2p* x = nullptr) {
    __shared__ float s_red[8];
    __shared__ float s_part[4][64];
    const int rec = nparam + 2;
    // 64 parameters per block, the record range split over the 4 waves (4x the loads in flight and 4x
    // the blocks of a thread-per-parameter loop); fixed summation order: slice-local in w order, then
    // (s0 + s1) + (s2 + s3).  Requested FIRST: these loads do not depend on n_filled, and at the reference's cadence (a dozen records per
    // agent) the launch is two dependent round trips to L2 if the n_filled reduction and its barrier sit in front of them (round 5)
    const int l64 = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l64;
    float acc = 0.f;
    // grad is [am.nblk][nparam]: a shared network's gradient is the sum over its agents, in agent order
    if (i < am.nblk * nparam) {
        const int blk = i / nparam, k = i - blk * nparam;
        for (int p = 0; p < P; ++p) {
            if (am.net[p] != blk) continue;
            const float* src = partials + (size_t)p * nwg * rec + k;
            // 32 record loads in flight per thread, summed in w order (explicit batches: `#pragma unroll 16 / 32` on the plain loop made the
            // compiler wait for every load before the next - 5.7 -> 17.7 us; its 8-wide form ran 5.6 us, this one 4.5 us: gpurun r6AI)
            int w = slice;
            for (; w + 4 * (REDUCE_BATCH - 1) < nwg; w += 4 * REDUCE_BATCH) {
                float v[REDUCE_BATCH];
#pragma unroll
                for (int k = 0; k < REDUCE_BATCH; ++k) v[k] = src[(size_t)(w + 4 * k) * rec];
#pragma unroll
                for (int k = 0; k < REDUCE_BATCH; ++k) acc += v[k];
            }
#pragma unroll 8
            for (; w < nwg; w += 4) acc += src[(size_t)w * rec];
        }
    }
    // n_filled (agent 0's records) and the loss sum (all records): strided loads + fixed-order tree
    float nf = 0.f, ls = 0.f;
    for (int w = threadIdx.x; w < P * nwg; w += 256) {
        ls += partials[(size_t)w * rec + nparam];
        if (w < nwg) nf += partials[(size_t)w * rec + nparam + 1];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        nf += __shfl_xor(nf, off);
        ls += __shfl_xor(ls, off);
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[threadIdx.x >> 6] = nf;
        s_red[4 + (threadIdx.x >> 6)] = ls;
    }
    s_part[slice][l64] = acc;
    __syncthreads();
    nf = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    ls = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
    float gv = 0.f;
    if (slice == 0) {
        const bool in = i < am.nblk * nparam;
        if (in) gv = ((s_part[0][l64] + s_part[1][l64]) + (s_part[2][l64] + s_part[3][l64])) / nf;
        if (x != nullptr) {  // wave 0 holds the block's 64 values: publish, wait for the peers' block, sum over the ranks in rank order
            bool late;
            gv = p2p_wave_sum(x->peers, x->rank, x->world, x->slot_floats, x->max_chunks, x->epoch, (int)blockIdx.x, (int64_t)i, in, gv,
                              x->timeout_ticks, late);
        }
        if (in) grad[i] = gv;
    }
    if (sumsq_out != nullptr && slice == 0) {  // wave 0 holds the block's 64 values: fixed-order butterfly
        float sq = gv * gv;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq += __shfl_xor(sq, off);
        if (l64 == 0) sumsq_out[blockIdx.x] = sq;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        loss[0] = ls / nf;
        loss[1] = nf;
    }
}

static __global__ __launch_bounds__(256) void dqn_reduce_kernel(const float* __restrict__ partials, int P, int nwg, int nparam,
                                                         AgentMap am, float* __restrict__ grad, float* __restrict__ loss) {
    dqn_reduce_body(partials, P, nwg, nparam, am, grad, loss, nullptr);
}

// reduce + the ranks' exchange in one launch: grad = SUM over the ranks of the mean gradients (clip + Adam then apply 1 / world)
static __global__ __launch_bounds__(256) void dqn_reduce_p2p_kernel(const float* __restrict__ partials, int P, int nwg, int nparam,
                                                             AgentMap am, float* __restrict__ grad, float* __restrict__ loss, ReduceP2p x) {
    dqn_reduce_body(partials, P, nwg, nparam, am, grad, loss, nullptr, &x);
}

static __global__ __launch_bounds__(256) void dqn_reduce_sq_kernel(const float* __restrict__ partials, int P, int nwg, int nparam,
                                                            AgentMap am, float* __restrict__ grad, float* __restrict__ loss,
                                                            float* __restrict__ sumsq_out) {
    dqn_reduce_body(partials, P, nwg, nparam, am, grad, loss, sumsq_out);
}

// ---- clip_grad_norm_ + Adam + target update (model.py:169-196) -----------------------------
static __global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ grad, int64_t n, float scale,
                                                    float* __restrict__ scratch) {
    __shared__ float red[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    float v = 0.f;
    if (i < n) {
        const float gv = grad[i] * scale;
        v = gv * gv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct AdamArgs {
    float lr_step;    // fp32(lr / (1 - beta1^step))
    float bc2_sqrt;   // fp32(sqrt(1 - beta2^step))
    float w1;         // fp32(1 - beta1): lerp weight
    float beta2, w2;  // beta2, fp32(1 - beta2)
    float eps, max_norm, grad_scale, tau;
    int hard_update;
    // getattr(optim, cfg.optimizer)(params, lr=cfg.lr) (dqn/model.py:66-71, ac/model.py:103-105) with torch's default hyper-parameters:
    // 0 Adam, 1 SGD (p += -lr g), 2 RMSprop (alpha 0.99, not centred, no momentum), 3 AdamW (decoupled weight decay 1e-2, then Adam)
    int opt = 0;
    float neg_lr = 0.f;   // fp32(-lr)                     (SGD, RMSprop)
    float alpha = 0.f;    // RMSprop smoothing, w2 = fp32(1 - alpha)
    float decay = 1.f;    // AdamW: fp32(1 - lr * weight_decay)
};

// one parameter's step; m / v are the optimiser's two state slots (Adam: exp_avg / exp_avg_sq; RMSprop: v = square_avg)
__device__ __forceinline__ void opt_step(const AdamArgs& a, float gv, float& mi, float& vi, float& pi) {
    if (a.opt == 1) {  // torch.optim.SGD: param.add_(grad, alpha=-lr)
        pi = pi + a.neg_lr * gv;
        return;
    }
    if (a.opt == 2) {  // torch.optim.RMSprop: square_avg.mul_(alpha).addcmul_(g, g, value=1 - alpha); avg = sqrt + eps; addcdiv_(g, avg, -lr)
        vi = vi * a.alpha + (a.w2 * gv) * gv;
        const float avg = sqrtf(vi) + a.eps;
        pi = pi + a.neg_lr * (gv / avg);
        return;
    }
    if (a.opt == 3) pi = pi * a.decay;  // torch.optim.AdamW: param.mul_(1 - lr * weight_decay) first
    // torch.optim.Adam (single-tensor): lerp_, mul_/addcmul_, sqrt/div/add_, addcdiv_
    mi = mi + a.w1 * (gv - mi);
    vi = vi * a.beta2 + a.w2 * gv * gv;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    pi = pi + (-a.lr_step) * (mi / denom);
}

static __global__ __launch_bounds__(256) void adam_kernel(int64_t n, int nblocks, float* __restrict__ params,
                                                   const float* __restrict__ grad, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ target, AdamArgs a,
                                                   const float* __restrict__ scratch, float* __restrict__ gnorm_out) {
    __shared__ float s_coef;
    __shared__ float s_red[4];
    {   // every block re-sums the per-block partials of sumsq_kernel: strided loads + fixed-order tree (reproducible)
        float ss = 0.f;
        for (int b = threadIdx.x; b < nblocks; b += 256) ss += scratch[b];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
        // clip_coef = max_norm / (total_norm + 1e-6), clamped to 1 (torch.nn.utils.clip_grad_norm_)
        float coef = 1.f;
        if (a.max_norm > 0.f) coef = fminf(a.max_norm / (total + 1e-6f), 1.f);
        s_coef = coef;
        if (gnorm_out != nullptr && blockIdx.x == 0) gnorm_out[0] = total;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gv = (grad[i] * a.grad_scale) * s_coef;
    float mi = m[i], vi = v[i], pi = params[i];
    opt_step(a, gv, mi, vi, pi);
    m[i] = mi;
    v[i] = vi;
    params[i] = pi;
    if (target != nullptr) {
        if (a.hard_update) target[i] = pi;
        else if (a.tau > 0.f) target[i] = (1.f - a.tau) * target[i] + a.tau * pi;
    }
}

}  // namespace marl

#include "ret_stats.h"
#include "dqn_update_tp.h"
#include "qmix_gen.h"
#include "qmix.h"

namespace marl {

// small blocks (marlhip_dqn_clip_adam: n <= 32k): norm + clip + Adam + target in ONE workgroup (one launch instead of two)
static __global__ __launch_bounds__(1024) void adam_fused_kernel(int64_t n, float* __restrict__ params, const float* __restrict__ grad,
                                                          float* __restrict__ m, float* __restrict__ v, float* __restrict__ target,
                                                          AdamArgs a, float* __restrict__ gnorm_out) {
    __shared__ float red[16];
    __shared__ float s_coef;
    float ss = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float gv = grad[i] * a.grad_scale;
        ss += gv * gv;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k];
        const float total = sqrtf(t);
        s_coef = a.max_norm > 0.f ? fminf(a.max_norm / (total + 1e-6f), 1.f) : 1.f;
        if (gnorm_out != nullptr) gnorm_out[0] = total;
    }
    __syncthreads();
    const float coef = s_coef;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float gv = (grad[i] * a.grad_scale) * coef;
        float mi = m[i], vi = v[i], pi = params[i];
        opt_step(a, gv, mi, vi, pi);
        m[i] = mi;
        v[i] = vi;
        params[i] = pi;
        if (target != nullptr) {
            if (a.hard_update) target[i] = pi;
            else if (a.tau > 0.f) target[i] = (1.f - a.tau) * target[i] + a.tau * pi;
        }
    }
}

// clip + Adam + target update as adam_kernel, and the step's parameter values written straight into the MFMA packs the next
// loss/grad launch stages (marlhip_idqn_update_n keeps the packs alive between updates, so dqn_pack_kernel runs once per call
// instead of once per update).  sumsq: per-block partials of dqn_reduce_sq_kernel (nsq of them).
template <class S>
__device__ __forceinline__ void adam_pack_body(int i, int n, int nsq, float* __restrict__ params, const float* grad,
                                               float* __restrict__ m, float* __restrict__ v, float* __restrict__ target,
                                               const AdamArgs& a, const float* sumsq, float* __restrict__ gnorm_out,
                                               const AgentMap& am, int P, float* __restrict__ packs) {
    constexpr int TOT = 2 * S::NFWD + S::NBWD;
    __shared__ float s_coef;
    __shared__ float s_red[4];
    // this thread's operands, requested BEFORE the clip-norm reduction and its barriers: they do not depend on the coefficient (round 5:
    // one round trip to L2 per launch instead of two at the reference's cadence, where the launch is nothing but latency)
    const bool mine = i >= 0 && i < n;
    const float g_raw = mine ? grad[i] : 0.f, m_old = mine ? m[i] : 0.f, v_old = mine ? v[i] : 0.f, p_old = mine ? params[i] : 0.f;
    const bool tgt = a.hard_update || a.tau > 0.f;
    const float t_old = (mine && tgt && !a.hard_update) ? target[i] : 0.f;
    {
        float ss = 0.f;
        if (nsq > 0) {
            for (int b = threadIdx.x; b < nsq; b += 256) ss += sumsq[b];
        } else {
            // data-parallel form (marlhip_idqn_update_n_dist): the gradient was summed over the ranks AFTER the reduce launch, so no
            // partial sums exist - every block takes the norm of the whole exchanged, scaled gradient itself (n floats from L2; fixed
            // order, so every block and every rank forms the same clip coefficient)
            // (16-byte loads, all of a thread's requests independent: the 13 us of the scalar loop - 87 dependent-looking round trips
            // per thread, measured on the forced-dist profile - become ~2)
            const int n4 = n >> 2;
            const f4* g4 = reinterpret_cast<const f4*>(grad);
            f4 part = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int j = threadIdx.x; j < n4; j += 256) {
                const f4 gj = g4[j] * a.grad_scale;
                part += gj * gj;
            }
            ss = (part[0] + part[1]) + (part[2] + part[3]);
            for (int j = 4 * n4 + threadIdx.x; j < n; j += 256) {
                const float gj = grad[j] * a.grad_scale;
                ss += gj * gj;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = ss;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = sqrtf((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
        s_coef = a.max_norm > 0.f ? fminf(a.max_norm / (total + 1e-6f), 1.f) : 1.f;
        if (gnorm_out != nullptr && blockIdx.x == 0) gnorm_out[0] = total;
    }
    __syncthreads();
    if (!mine) return;
    const float gv = (g_raw * a.grad_scale) * s_coef;
    float mi = m_old, vi = v_old;
    mi = mi + a.w1 * (gv - mi);
    vi = vi * a.beta2 + a.w2 * gv * gv;
    const float denom = sqrtf(vi) / a.bc2_sqrt + a.eps;
    float pi = p_old;
    pi = pi + (-a.lr_step) * (mi / denom);
    m[i] = mi;
    v[i] = vi;
    params[i] = pi;
    float ti = 0.f;
    if (tgt) {
        ti = a.hard_update ? pi : (1.f - a.tau) * t_old + a.tau * pi;
        target[i] = ti;
    }
    const int blk = i / S::NPARAM, k = i - blk * S::NPARAM;
    int fwd, bwd;
    mlp_param_to_pack<S>(k, fwd, bwd);
    for (int p = 0; p < P; ++p) {
        if (am.net[p] != blk) continue;
        float* pk = packs + (size_t)p * TOT;
        pk[fwd] = pi;
        if (bwd >= 0) pk[2 * S::NFWD + bwd] = pi;
        if (tgt) pk[S::NFWD + fwd] = ti;
    }
}

template <class S>
static __global__ __launch_bounds__(256) void adam_pack_kernel(int n, int nsq, float* __restrict__ params, const float* __restrict__ grad,
                                                        float* __restrict__ m, float* __restrict__ v, float* __restrict__ target,
                                                        AdamArgs a, const float* __restrict__ sumsq, float* __restrict__ gnorm_out,
                                                        AgentMap am, int P, float* __restrict__ packs) {
    adam_pack_body<S>(blockIdx.x * 256 + threadIdx.x, n, nsq, params, grad, m, v, target, a, sumsq, gnorm_out, am, P, packs);
}

// (Measured and dropped, r02: the same epilogue as ONE launch - reduce, a grid barrier, then Adam + packs in the same workgroups.
// Through hipLaunchCooperativeKernel the launch itself costs ~30 us (bench 27.1 -> 21.8 M env-steps/s); with a hand-rolled barrier
// on a plain launch (agent-scope release / acquire around an atomic arrival counter) the L2 write-back + invalidate of the fences
// costs more than the kernel boundary it replaces (27.1 -> 26.3 M; 0.92 -> 0.75 M at the reference cadence).  Two launches it is.
// r03: for the small updates of the reference cadence (a dozen records per agent) the whole epilogue in ONE 1024-thread workgroup - no grid
// barrier needed - was tried as well: one CU's worth of loads in flight makes it ~60 us against 9 us for the two launches that spread
// the same bytes over 130 workgroups (0.92 -> 0.31 M env-steps/s, scripts/gpu_runs/r3U.sh).  Dropped.
// r05: the step inside the NEXT learner launch's prologue - every workgroup steps its agent's whole block itself (ping-pong state, the
// new values scattered straight into its LDS packs: no Adam launch, no pack staging; bitwise the three-launch results, commit 6f18cb7,
// scripts/gpu_runs/r5C.sh).  22 parameters per thread x ~150 VALU instructions (three IEEE divisions / roots and the pack index math each)
// at one wave per SIMD is 8.3 us of serial instruction issue against the 4.9 us launch + 1.7 us staging it removes: the learner launch
// went 19.7 -> 26.3 us at the reference cadence (0.864 -> 0.782 M env-steps/s) and 98.3 -> 106.0 us at the headline (27.8 -> 26.8 M).
// Dropped: an optimiser step spread over 88 workgroups beats the same step repeated in each of 26.)

struct UpdPlan {
    int nwg, n_chunks;
};

// 4 waves per workgroup (1 per SIMD): 8 waves x 5 transpose tiles do not fit the 160 KiB LDS next to the packs and spill
constexpr int UPD_WAVES = 4;

inline UpdPlan upd_plan(int P, int T, int B) {
    const int ngroups = (B + 15) / 16;
    constexpr int W = UPD_WAVES;
    const int want_waves = 256 * W / (P > 0 ? P : 1) > W ? 256 * W / P : W;  // fill every CU with one workgroup
    int nc = (want_waves + ngroups - 1) / ngroups;
    if (nc < 1) nc = 1;
    if (nc > T) nc = T;
    const int tasks = ngroups * nc;
    int nwg = (tasks + W - 1) / W;
    const int cap = 256 / P > 1 ? 256 / P : 1;
    if (nwg > cap) nwg = cap;
    UpdPlan pl = {nwg, nc};
    return pl;
}

// VDN mixer (VDNetwork._compute_loss, dqn/model.py:237,254-269): chosen_tot = sum_p chosen_p, target_tot = sum_p tq_p,
// y = r_0 + gamma * target_tot * (1 - done), delta = chosen_tot - y; dL/dchosen_p = 2 * filled * delta for every p.
// ret (optional): the standardised returns of colstd_returns_kernel instead of r + gamma * target * (1 - done)
static __global__ __launch_bounds__(256) void vdn_mix_kernel(MixBufs mix, int P, int T, int B, float gamma, const float* __restrict__ ret) {
    const int n = T * B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float ch = 0.f, tq = 0.f;
        for (int p = 0; p < P; ++p) {
            ch += mix.chosen[(size_t)p * n + i];
            tq += mix.tqsel[(size_t)p * n + i];
        }
        const float y = ret != nullptr ? ret[i] : mix.r0[i] + gamma * tq * (1.f - mix.dn[i]);
        const float delta = ch - y;
        const float fl = mix.fl[i];
        mix.dq[i] = 2.f * fl * delta;
        mix.lrow[i] = fl * delta * delta;
    }
}

// the independent learners' "mixer" for the two-pass form (wide first layers, where the single-pass kernel would spill): per agent
// delta_p = chosen_p - (r_p + gamma tq_p (1 - done)), dq_p = 2 filled delta_p, row loss = filled sum_p delta_p^2 (dqn/model.py:152,160-163)
static __global__ __launch_bounds__(256) void idqn_mix_kernel(MixBufs mix, int P, int T, int B, float gamma) {
    const int n = T * B;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float fl = mix.fl[i], nd = 1.f - mix.dn[i];
        float l = 0.f;
        for (int p = 0; p < P; ++p) {
            const float delta = mix.chosen[(size_t)p * n + i] - (mix.rew_all[(size_t)p * n + i] + gamma * mix.tqsel[(size_t)p * n + i] * nd);
            mix.dq[(size_t)p * n + i] = 2.f * fl * delta;
            l += delta * delta;
        }
        mix.lrow[i] = fl * l;
    }
}

// workspace layout (floats unless noted): [partial records][pad16][packs][mixer buffers (4P+5) T B][pad8][128 B phase counters]
struct WsLayout {
    int64_t rec_bytes, pack_off, mix_off, total;
};

// tensor-parallel learner (hidden 128, wide rows): pass F leaves the critic's second hidden layer of every transition row for pass B
// (dqn_update_tp.h, h2_out): P * T * [row blocks of 16, counted in pairs] * 16 * H floats behind everything else
#ifndef MARL_TP_NB1S_D
#define MARL_TP_NB1S_D 80  // rows wider than this walk one row block per step in the pass that reads both hidden layers back
#endif
inline int64_t tp_h2_floats(int P, int T, int B, int H) { return (int64_t)P * T * (((B + 31) / 32) * 2) * 16 * H; }  // == tp_h2_blocks(B) row blocks

inline WsLayout ws_layout(int P, int nwg, int rec, int pack, int T, int B) {
    WsLayout w;
    w.rec_bytes = (int64_t)P * nwg * rec * sizeof(float);
    w.pack_off = (w.rec_bytes + 15) & ~(int64_t)15;
    w.mix_off = w.pack_off + (int64_t)P * pack * sizeof(float);
    // mixer buffers of either path ((4P+5) planes of T*B) + the standardising mixer's block partials (2 P per 256 rows)
    const int64_t tb = (int64_t)T * B, mixf = (4 * P + 5) * tb + 2 * P * (tb / 256 + 1);
    w.total = ((w.mix_off + mixf * (int64_t)sizeof(float) + 7) & ~(int64_t)7) + 128;
    return w;
}

// LDS-resident learner, two-pass form (VDN / QMIX / standardised IDQN): the qsel pass leaves the critic's two hidden layers of every transition
// row for the bwd pass - P * T * [row groups of 16] * 2 layers * [H / 16 tiles] * 256 floats behind everything else (+ 128 bytes of slack:
// the MARLHIP_PROF counters sit at the very end of the caller's workspace)
inline int64_t lds_h_floats(int P, int T, int B, int H) { return (int64_t)P * T * ((B + 15) / 16) * 2 * (H / 16) * 256 + 32; }
#ifndef MARL_LDS_STORED
#define MARL_LDS_STORED 1
#endif

#ifndef MARL_TP_W
// Waves per workgroup of the tensor-parallel kernels (each owns H / (16 W) hidden tiles).  8 (one tile per wave, two waves per SIMD, <= 256
// registers) was measured against 4 (scripts/gpu_runs/r3AC.sh): IDQN 128-128 8.21 -> 7.95 M, VDN 15x15-4p 3.79 -> 3.26 M (27 / 39-wide first
// layers spill 0.1 - 0.4 KB per lane), IA2C 128-128 104.9 -> 107.2 M: twice the barrier participants and exchange traffic cost what the
// second wave per SIMD hides.  4 stays.
#define MARL_TP_W 4
#endif
#ifndef MARL_TP_NBF
#define MARL_TP_NBF 2  // row blocks per step of the forward pass
#endif
// hidden 128: tensor-parallel passes (dqn_update_tp.h) - pass F, mixer, pass B, reduce
inline UpdPlan upd_plan_tp(int P, int T, int B, int NB, int occ = 1) {  // `occ` workgroups per CU, tasks = (block set, time chunk)
    const int nsets = (B + 16 * NB - 1) / (16 * NB);
    const int want = occ * 256 / P > 1 ? occ * 256 / P : 1;
    int nc = (want + nsets - 1) / nsets;
    if (nc < 1) nc = 1;
    if (nc > T) nc = T;
    int nwg = nsets * nc;
    if (nwg > want) nwg = want;
    UpdPlan pl = {nwg, nc};
    return pl;
}

// QMIX mixer stage (qmix.h), dispatched on the (agents, obs dim) pair
template <int D, bool REPLAY>
int qmix_dispatch_mix(int P, const QmixCtx& qx, const marlhip_batch* bt, const ReplaySrc& src, const QmixIo& io, float gamma,
                      hipStream_t st) {
    if (qx.generic) return qmix_gen_mix(qx, qx.gen, bt, REPLAY ? &src : nullptr, io, gamma, st);
#define X(p, d) \
    if constexpr (D == d) { if (P == p) return qmix_launch_mix<QmixShape<p, d>, REPLAY>(qx, bt, src, io, gamma, st); }
    MARL_QMIX_SHAPES(X)
#undef X
    set_error("no QMIX mixer kernel for %d agents x %d observations (add it to MARL_QMIX_SHAPES)", P, D);
    return -1;
}

template <int D>
int qmix_dispatch_reduce(int P, const QmixCtx& qx, int T, int B, const float* loss, hipStream_t st) {
    if (qx.generic) return qmix_gen_reduce(qx, qx.gen, T, B, loss, st);
#define X(p, d) \
    if constexpr (D == d) { if (P == p) return qmix_launch_reduce<QmixShape<p, d>>(qx, T, B, loss, st); }
    MARL_QMIX_SHAPES(X)
#undef X
    set_error("no QMIX mixer kernel for %d agents x %d observations", P, D);
    return -1;
}

template <class S, bool REPLAY>
int launch_lossgrad_tp(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt,
                       const ReplaySrc& src, float gamma, int double_q, int mode, void* ws, int64_t ws_bytes, float* grad,
                       float* loss, hipStream_t st, const QmixCtx* qx, const RetStats* rst) {
    // pass B reads BOTH hidden layers back from pass F (round 4: STORED1 - no layer-1 weights / row copies in registers), so it walks two row
    // blocks per step up to 80-wide rows (the recomputing form: 48)
    constexpr int W = MARL_TP_W, TPW = S::H / (16 * W), NB = S::D > MARL_TP_NB1S_D ? 1 : 2, NBF = MARL_TP_NBF, NT = W * TPW, REC = S::NPARAM + 2;
    const int P = s->n_agents, T = bt->max_len, B = bt->batch;
    const AgentMap am = agent_map(s);
    static_assert(NB <= 2 && NBF <= 2, "tp_h2_blocks counts row blocks in pairs");
    const UpdPlan pl = upd_plan_tp(P, T, B, NB, MARL_TP_BWD_OCC), plF = upd_plan_tp(P, T, B, NBF);
    const WsLayout wl = ws_layout(P, pl.nwg, REC, 0, T, B);
    const int64_t h2_off = (wl.total + 15) & ~(int64_t)15, need = h2_off + 2 * tp_h2_floats(P, T, B, S::H) * (int64_t)sizeof(float);  // h2 | h1 records
    MARL_REQUIRE(ws_bytes >= need, "dqn_loss_grad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)need);
    f4* h2buf = reinterpret_cast<f4*>(static_cast<char*>(ws) + h2_off);
    float* mixf = reinterpret_cast<float*>(static_cast<char*>(ws) + wl.mix_off);
    const size_t tb = (size_t)T * B;
    TpMix mix;
    mix.chosen = mixf; mix.tqsel = mixf + P * tb; mix.rew = mixf + 2 * P * tb; mix.dq = mixf + 3 * P * tb;
    mix.dn = mixf + 4 * P * tb; mix.fl = mix.dn + tb; mix.lrow = mix.fl + tb;
    const size_t ldsF = 2 * (size_t)(2 * NBF * NT + 2 * NBF * W) * 256 * sizeof(float);  // two alternating sets (dqn_update_tp.h)
    const size_t ldsB = (size_t)tp_bwd_lds_floats<S, W, TPW, NB, true>() * sizeof(float);
    static_assert(2 * (2 * NBF * NT + 2 * NBF * W) * 256 * 4 <= 160 * 1024 && tp_bwd_lds_floats<S, W, TPW, NB, true>() * 4 <= 160 * 1024, "tensor-parallel passes: LDS");
    static LdsAttr attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_fwd_kernel<S, W, TPW, REPLAY, NBF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsF);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&tp_bwd_kernel<S, W, TPW, REPLAY, NB, false, true, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB);
        attr_set.done();
    }
    const dim3 grid(pl.nwg, P), block(64 * W);
    timing_begin(TIMER_LOSSGRAD, st);
    hipLaunchKernelGGL((tp_fwd_kernel<S, W, TPW, REPLAY, NBF>), dim3(plF.nwg, P), block, ldsF, st, params, tparams, am, *bt, src, mix,
                       double_q, plF.n_chunks, h2buf);
    if (mode == 2) {
        QmixIo io = {mix.chosen, mix.tqsel, mix.rew, mix.dn, mix.fl, mix.dq, mix.lrow, nullptr};
        const int rc = qmix_dispatch_mix<S::D, REPLAY>(P, *qx, bt, src, io, gamma, st);
        if (rc != 0) return rc;
    } else if (mode == 3) {  // IDQN, standardised returns; block partials live behind lrow (2 P ceil(tb/256) <= tb floats)
        const int rc = launch_std_mixer(P, (int)tb, gamma, *rst, mix.chosen, mix.tqsel, mix.rew, mix.dn, mix.fl, mix.dq, mix.lrow,
                                        mixf + (size_t)(4 * P + 5) * tb, st);
        if (rc != 0) return rc;
    } else {
        const float* ret = nullptr;
        if (mode == 1 && rst != nullptr) {  // VDN with standardise_returns (the returns take a spare plane behind lrow)
            float* rbuf = mixf + (size_t)(4 * P + 3) * tb;
            const int rc = launch_colstd(T, B, gamma, *rst, mix.tqsel, P, tb, mix.rew, mix.dn, rbuf, st);
            if (rc != 0) return rc;
            ret = rbuf;
        }
        hipLaunchKernelGGL(tp_mix_kernel, dim3((unsigned)((tb + 255) / 256 > 1024 ? 1024 : (tb + 255) / 256)), dim3(256), 0, st, mix, P,
                           T, B, gamma, mode == 1 ? 1 : 0, ret);
    }
    hipLaunchKernelGGL((tp_bwd_kernel<S, W, TPW, REPLAY, NB, false, true, true>), grid, block, ldsB, st, params, am, *bt, src, mix, pl.n_chunks,
                       (float*)ws, (const f4*)h2buf, (const f4*)h2buf + tp_h2_floats(P, T, B, S::H) / 4);
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("tp_lossgrad");
    const int n = am.nblk * S::NPARAM;
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    if (mode == 2) return qmix_dispatch_reduce<S::D>(P, *qx, T, B, loss, st);
    return 0;
}

// marlhip_idqn_update_n's in-call state: the packs in the workspace stay valid from one update to the next because the Adam launch
// of update u writes them for update u+1 (adam_pack_kernel); the reduce launch leaves the clip norm's partial sums
struct UpdFuse {
    int packs_valid;  // in: skip dqn_pack_kernel; out: 1 after a step that kept them current
    AdamArgs adam;
    float *params_rw, *target_rw, *exp_avg, *exp_avg_sq, *gnorm;
    float* sumsq;     // >= ceil(n / 64) floats
    // data-parallel training (marlhip_idqn_update_n_dist): called between the gradient reduce and the Adam launch; leaves the SUM over
    // the ranks in `grad`, ordered on the stream; adam.grad_scale = 1 / world.  nullptr: single GPU.
    marlhip_exchange_fn exchange = nullptr;
    void* exchange_ctx = nullptr;
    // filled-aware task plans (update_plan.h): one planning launch covers as many of the call's remaining updates as the idle mixer planes
    // of the workspace hold; the caller sets plan_enabled and, before every update, updates_left (this one included)
    int plan_enabled = 0, updates_left = 0, plan_left = 0, plan_pos = 0;
};

template <class S, bool REPLAY>
int launch_lossgrad_src(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt,
                        const ReplaySrc& src, float gamma, int double_q, int mode, void* ws, int64_t ws_bytes, float* grad,
                        float* loss, hipStream_t st, const QmixCtx* qx, const RetStats* rst, UpdFuse* fuse = nullptr) {
    if constexpr (use_tp<S>()) {
        if (fuse != nullptr) { set_error("fused update epilogue: not a fused-kernel shape"); return -1; }
        return launch_lossgrad_tp<S, REPLAY>(s, params, tparams, bt, src, gamma, double_q, mode, ws, ws_bytes, grad, loss, st, qx, rst);
    } else {
    using L = UpdLds<S>;
    static_assert(L::FITS, "packs + tiles / fold regions exceed the 160 KiB LDS");
    const int P = s->n_agents, T = bt->max_len, B = bt->batch;
    const AgentMap am = agent_map(s);
    const UpdPlan pl = upd_plan(P, T, B);
    constexpr int PACK = 2 * S::NFWD + S::NBWD;
    static_assert(PACK % 4 == 0 && L::oT == S::NFWD && L::oB == 2 * S::NFWD, "pack layout == LDS layout");
    const WsLayout wl = ws_layout(P, pl.nwg, L::REC, PACK, T, B);
    MARL_REQUIRE(ws_bytes >= wl.total, "dqn_loss_grad: workspace %lld < %lld bytes", (long long)ws_bytes, (long long)wl.total);
    float* packs = reinterpret_cast<float*>(static_cast<char*>(ws) + wl.pack_off);
    float* mixf = reinterpret_cast<float*>(static_cast<char*>(ws) + wl.mix_off);
    const size_t lds_bytes = (size_t)L::total(4) * sizeof(float);
    static LdsAttr attr_set;
    if (attr_set.need()) {
        if constexpr (S::D <= 48)  // (wider first layers never launch the single-pass kernel: IDQN_TWO_PASS below)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, REPLAY, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, REPLAY, 1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, REPLAY, 2>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, REPLAY, 1, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dqn_lossgrad_kernel<S, 4, REPLAY, 2, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set.done();
    }
    // two-pass form: room behind the layout for the critic's hidden layers of every transition row (marlhip_dqn_workspace_bytes reserves
    // it; a caller-sized workspace without it - the actor-critic step's - keeps the recomputing bwd pass)
    const int64_t hs_off = (wl.total + 15) & ~(int64_t)15;
    // independent learners on wide first layers (the 71-wide warehouse rows: the single-pass kernel spills 0.5 KB per lane there - 80 dW1
    // accumulator registers next to both forwards) take the two-pass form too, with idqn_mix_kernel between the passes
    constexpr bool IDQN_TWO_PASS = S::D > 48;
    const bool two_pass = mode != 0 || IDQN_TWO_PASS;
    const bool stored = MARL_LDS_STORED && two_pass && ws_bytes >= hs_off + lds_h_floats(P, T, B, S::H) * (int64_t)sizeof(float) + 128;
    f4* hst = stored ? reinterpret_cast<f4*>(static_cast<char*>(ws) + hs_off) : nullptr;
    if (fuse == nullptr || !fuse->packs_valid) {
        hipLaunchKernelGGL((dqn_pack_kernel<S>), dim3((PACK + 255) / 256, P), dim3(256), 0, st, params, tparams, am, packs);
        MARL_CHECK_LAUNCH("dqn_pack_kernel");
    }
    unsigned long long* prof =
        getenv("MARLHIP_PROF") ? reinterpret_cast<unsigned long long*>(static_cast<char*>(ws) + ws_bytes - 128) : nullptr;
    const size_t tb = (size_t)T * B;
    MixBufs mix;
    mix.chosen = mixf; mix.tqsel = mixf + P * tb; mix.r0 = mixf + 2 * P * tb; mix.dn = mix.r0 + tb; mix.fl = mix.dn + tb;
    mix.dq = mix.fl + tb; mix.lrow = mix.dq + tb; mix.dq_agent_stride = 0;
    mix.rew_all = nullptr;
    mix.dout = nullptr;
    if (mode == 2 || mode == 3 || (mode == 0 && IDQN_TWO_PASS)) {  // QMIX / standardised IDQN / two-pass IDQN: one dq plane per agent
        mix.lrow = mix.dq + P * tb;
        mix.dq_agent_stride = (int)tb;
    }
    if (mode == 3 || (mode == 0 && IDQN_TWO_PASS)) mix.rew_all = mix.lrow + tb;  // [P][tb]; the statistics' block partials follow it
    const dim3 grid(pl.nwg, P), block(256);
    ReplaySrc psrc = src;
    if constexpr (REPLAY && !IDQN_TWO_PASS) {
        // the single-pass learner leaves the mixer planes of the workspace idle: they hold the filled-aware plans of the next K updates
        if (fuse != nullptr && fuse->plan_enabled && !two_pass && src.idx == nullptr) {
            const PlanDims pd = plan_dims(P, T, B, pl.nwg, UPD_WAVES, pl.n_chunks);
            const int64_t region = (int64_t)(4 * P + 5) * T * B * (int64_t)sizeof(float);
            const int K = pd.planned ? (int)(region / ((int64_t)pd.stride * 4) > 4096 ? 4096 : region / ((int64_t)pd.stride * 4)) : 0;
            if (K >= 1) {
                int32_t* pbase = reinterpret_cast<int32_t*>(mixf);
                if (fuse->plan_left == 0) {
                    const int n = fuse->updates_left < K ? (fuse->updates_left > 0 ? fuse->updates_left : 1) : K;
                    const size_t plds = plan_lds_bytes(pd);
                    static LdsAttr plan_attr;
                    if (plan_attr.need()) {
                        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&update_plan_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
                        plan_attr.done();
                    }
                    hipLaunchKernelGGL(update_plan_kernel, dim3(n), dim3(PLAN_THREADS), plds, st, src.rb, src.seed, src.counter, src.length, pd, pbase,
                                       src.idx_out, fuse->updates_left <= n ? fuse->updates_left - 1 : -1);
                    MARL_CHECK_LAUNCH("update_plan_kernel");
                    fuse->plan_left = n;
                    fuse->plan_pos = 0;
                }
                psrc.plan = pbase + (size_t)fuse->plan_pos * pd.stride;
                psrc.idx = psrc.plan + PLAN_HDR;
                psrc.idx_out = nullptr;  // (the planning launch recorded the draws, in draw order)
                fuse->plan_pos += 1;
                fuse->plan_left -= 1;
            }
        }
    }
    timing_begin(TIMER_LOSSGRAD, st);
    if (!two_pass) {
        if constexpr (!IDQN_TWO_PASS)
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, REPLAY, 0>), grid, block, lds_bytes, st, (const float*)packs, *bt, psrc, mix,
                               gamma, double_q, pl.n_chunks, (float*)ws, prof);
    } else {
        if (stored)
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, REPLAY, 1, true>), grid, block, lds_bytes, st, (const float*)packs, *bt, src, mix,
                               gamma, double_q, pl.n_chunks, (float*)ws, prof, hst);
        else
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, REPLAY, 1>), grid, block, lds_bytes, st, (const float*)packs, *bt, src, mix,
                               gamma, double_q, pl.n_chunks, (float*)ws, prof);
        if (mode == 0) {
            hipLaunchKernelGGL(idqn_mix_kernel, dim3((unsigned)((tb + 255) / 256 > 1024 ? 1024 : (tb + 255) / 256)), dim3(256), 0, st, mix, P, T, B, gamma);
        } else if (mode == 2) {
            QmixIo io = {mix.chosen, mix.tqsel, mix.r0, mix.dn, mix.fl, mix.dq, mix.lrow, nullptr};
            const int rc = qmix_dispatch_mix<S::D, REPLAY>(P, *qx, bt, src, io, gamma, st);
            if (rc != 0) return rc;
        } else if (mode == 3) {
            const int rc = launch_std_mixer(P, (int)tb, gamma, *rst, mix.chosen, mix.tqsel, mix.rew_all, mix.dn, mix.fl, mix.dq,
                                            mix.lrow, mixf + (size_t)(4 * P + 5) * tb, st);
            if (rc != 0) return rc;
        } else {
            const float* ret = nullptr;
            if (rst != nullptr) {  // VDN with standardise_returns: per-batch-column statistics; the returns take a spare mixer plane
                float* rbuf = mixf + (size_t)(2 * P + 5) * tb;
                const int rc = launch_colstd(T, B, gamma, *rst, mix.tqsel, P, tb, mix.r0, mix.dn, rbuf, st);
                if (rc != 0) return rc;
                ret = rbuf;
            }
            hipLaunchKernelGGL(vdn_mix_kernel, dim3((unsigned)((tb + 255) / 256 > 1024 ? 1024 : (tb + 255) / 256)), dim3(256), 0, st,
                               mix, P, T, B, gamma, ret);
        }
        if (stored)
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, REPLAY, 2, true>), grid, block, lds_bytes, st, (const float*)packs, *bt, src, mix,
                               gamma, double_q, pl.n_chunks, (float*)ws, prof, hst);
        else
            hipLaunchKernelGGL((dqn_lossgrad_kernel<S, 4, REPLAY, 2>), grid, block, lds_bytes, st, (const float*)packs, *bt, src, mix,
                               gamma, double_q, pl.n_chunks, (float*)ws, prof);
    }
    timing_end(TIMER_LOSSGRAD, st);
    MARL_CHECK_LAUNCH("dqn_lossgrad_kernel");
    const int n = am.nblk * S::NPARAM;
    if (fuse != nullptr) {  // reduce (+ clip-norm partials) -> clip + Adam + target + next update's packs: two launches
        MARL_REQUIRE(mode == 0 || mode == 1, "fused update epilogue: IDQN / VDN only");
        int nsq = (n + 63) / 64;
        if (fuse->exchange != nullptr) {
            // N > 1: reduce -> all-reduce(SUM) of the flat gradient (the caller's RCCL hop) -> Adam, the clip norm taken from the
            // exchanged gradient inside the Adam launch (the clip must use the global post-reduce norm, dqn/model.py:170)
            P2pState* ps = p2p_is_builtin(fuse->exchange) ? static_cast<P2pState*>(fuse->exchange_ctx) : nullptr;
            timing_begin(TIMER_EXCHANGE, st);
            if (ps != nullptr && ps->connected && n <= ps->max_floats) {
                // the library's own exchange: folded into the reduce launch (each workgroup publishes its 64 values, waits for the same
                // workgroup of the peers, sums in rank order) - one launch less per update than reduce -> exchange kernel
                ps->epoch += 1;
                ReduceP2p x = {ps->peers, ps->rank, ps->world, ps->max_chunks, ps->max_floats, ps->epoch, p2p_timeout_ticks()};
                hipLaunchKernelGGL(dqn_reduce_p2p_kernel, dim3(nsq), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss, x);
                MARL_CHECK_LAUNCH("dqn_reduce_p2p_kernel");
            } else {
                hipLaunchKernelGGL(dqn_reduce_kernel, dim3(nsq), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss);
                MARL_CHECK_LAUNCH("dqn_reduce_kernel");
                const int rc = fuse->exchange(fuse->exchange_ctx, grad, (int64_t)n, (void*)st);
                MARL_REQUIRE(rc == 0, "idqn_update_n_dist: the gradient exchange callback failed (%d)", rc);
            }
            timing_end(TIMER_EXCHANGE, st);
            nsq = 0;  // the clip norm of the EXCHANGED gradient: taken inside the Adam launch (the same arithmetic for every exchange)
        } else {
            hipLaunchKernelGGL(dqn_reduce_sq_kernel, dim3(nsq), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss,
                               fuse->sumsq);
            MARL_CHECK_LAUNCH("dqn_reduce_sq_kernel");
        }
        hipLaunchKernelGGL((adam_pack_kernel<S>), dim3((n + 255) / 256), dim3(256), 0, st, n, nsq, fuse->params_rw, (const float*)grad,
                           fuse->exp_avg, fuse->exp_avg_sq, fuse->target_rw, fuse->adam, (const float*)fuse->sumsq, fuse->gnorm, am, P,
                           packs);
        MARL_CHECK_LAUNCH("adam_pack_kernel");
        fuse->packs_valid = 1;
        return 0;
    }
    hipLaunchKernelGGL(dqn_reduce_kernel, dim3((n + 63) / 64), dim3(256), 0, st, (const float*)ws, P, pl.nwg, S::NPARAM, am, grad, loss);
    MARL_CHECK_LAUNCH("dqn_reduce_kernel");
    if (mode == 2) return qmix_dispatch_reduce<S::D>(P, *qx, T, B, loss, st);
    return 0;
    }
}

template <class S>
int launch_lossgrad(const marlhip_net_shape* s, const float* params, const float* tparams, const marlhip_batch* bt,
                    const ReplaySrc* rsrc, float gamma, int double_q, int mode, void* ws, int64_t ws_bytes, float* grad, float* loss,
                    hipStream_t st, const QmixCtx* qx, const RetStats* rst, UpdFuse* fuse = nullptr) {
    if (rsrc != nullptr)
        return launch_lossgrad_src<S, true>(s, params, tparams, bt, *rsrc, gamma, double_q, mode, ws, ws_bytes, grad, loss, st, qx, rst, fuse);
    ReplaySrc none = {};
    return launch_lossgrad_src<S, false>(s, params, tparams, bt, none, gamma, double_q, mode, ws, ws_bytes, grad, loss, st, qx, rst, fuse);
}

// can this shape take marlhip_idqn_update_n's fused epilogue (LDS-resident-pack learner kernel)?
template <class S>
constexpr bool fused_epilogue_ok() { return !use_tp<S>(); }

}  // namespace marl

