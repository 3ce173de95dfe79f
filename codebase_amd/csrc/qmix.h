// QMIX monotonic mixer (QMixer, marlbase/dqn/model.py:272-331) and its part of QMixNetwork._compute_loss
// (model.py:374-427) on gfx950.  Included by dqn_update.hip after its helpers; the agent networks run through the
// same forward-only / backward-with-external-dq passes VDN uses, this file is the stage between them:
//
//   chosen_p[t][b], tqsel_p[t][b]  ->  y = mixer(chosen, s_t), y' = target_mixer(tqsel, s_{t+1})
//   delta = y - (r_0 + gamma y' (1 - done)),  dq_p = dL/dchosen_p,  gradient of the mixer parameters
//
// with s = the concatenation of all agents' observations (model.py:389,412; state_dim = P*D, model.py:360).
// One row = one (t, b) pair, R = T*B rows.  Three launches, every GEMM on v_mfma_f32_16x16x4_f32 (exact f32):
//   qmix_net_kernel<target>  per 16-row block and wave: Y1 = act(S W1cat^T + bias), the four state-fed first layers in ONE GEMM
//                     (features [0,32) hyper_w_1.0 +ReLU | [32,64) hyper_w_final.0 +ReLU | [64,128) hyper_b_1 | [128,192) V.0 +ReLU;
//                     weights stream through an LDS window in 64-column K chunks), kept in registers and fed straight into the mixing
//                     network: w1 = |hyper_w_1.2 h1 + c|, z = sum_p q_p w1_p + b1, hidden = elu(z), wf = |hyper_w_final.2 hf + c|,
//                     y = hidden.wf + V.2 hv + c.  The [T*B][192] first-layer activations never exist in memory.
//   qmix_net_kernel<online>  the same, then the TD error and the backward down to the first-layer pre-activations; it writes dq_p and
//                     the operands of the weight-gradient GEMMs.
//   qmix_wgrad_kernel split-K (over rows) weight-gradient GEMMs, both groups (first layers | mixing network) side by side in one
//                     launch; 8 waves own disjoint accumulator tiles, no LDS, next block's operands prefetched.
//   qmix_reduce_kernel sums the per-workgroup records in fixed order and applies 1/sum(filled).
// hypernet_layers == 2, embed_dim 64, hypernet_embed 32 (configs/algorithm/qmix.yaml:14-17) are compiled in.
// d|x|/dx at exactly 0 is taken as -1 (torch: 0): a pre-activation that is exactly 0.0f does not occur with
// non-degenerate parameters.
#pragma once

#ifndef MARL_QMIX_WG
#define MARL_QMIX_WG 1  // weight gradients inside the online instance's kernel where the operand tiles fit the LDS (qmix_wg_round)
#endif
#ifndef MARL_QMIX_L1_NB
#define MARL_QMIX_L1_NB 1  // row blocks per wave and step in qmix_l1_kernel (register blocking vs resident workgroups per CU)
#endif
#ifndef MARL_QMIX_FUSE_NCH
// Shapes with at most this many 64-column K chunks of first-layer weights take the fused per-instance kernel, the rest the split form.
// Measured (profiles/r03_qmix_mixer_ab.md): fused wins with 1-3 chunks (2p 8x8: 110 vs 162 us for both instances, 4p 15x15: 465 vs 535,
// rware 2ag: 1009 vs 1237); with 5 chunks and 8 agents the fused online instance spills (150 KB of mixing pack leave it one wave per SIMD).
#define MARL_QMIX_FUSE_NCH 3
#endif

namespace marl {

template <int P_, int D_>
struct QmixShape {
    static constexpr int P = P_, D = D_, SD = P_ * D_, E = 64, HE = 32;
    static constexpr int NF1 = 2 * HE + 2 * E;  // 192 first-layer features
    static constexpr int MT1 = NF1 / 16;        // 12 feature tiles
    static constexpr int KS4 = (SD + 15) / 16;  // state columns in groups of 16 (4 k-steps)
    static constexpr int NCH = (KS4 + 3) / 4;   // LDS chunks of 64 state columns
    static constexpr int W1T = 4 * P;           // feature tiles of w1 (E*P / 16)
    // canonical parameter block = mixer.parameters() order
    static constexpr int oA1 = 0, oa1 = oA1 + HE * SD, oB1 = oa1 + HE, oc1 = oB1 + E * P * HE, oAf = oc1 + E * P,
                         oaf = oAf + HE * SD, oBf = oaf + HE, ocf = oBf + E * HE, oBb = ocf + E, ocb = oBb + E * SD,
                         oAv = ocb + E, oav = oAv + E * SD, obv = oav + E, ocv = obv + E;
    static constexpr int NPARAM = ocv + 1;
    // first-layer pack: A[ks4][mt][lane][e] = W1cat[16mt+i][16ks4+4e+g], then bias[192]
    static constexpr int pL1b = KS4 * MT1 * 256, NL1 = pL1b + NF1;
    // mixing pack: P1[W1T][2][64][4] c1[E*P] PF[4][2][64][4] cf[64] bv[64] cv[4] | T1[2][W1T][64][4] TF[2][4][64][4]
    static constexpr int mP1 = 0, mc1 = mP1 + W1T * 512, mPF = mc1 + E * P, mcf = mPF + 2048, mbv = mcf + 64, mcv = mbv + 64,
                         mT1 = mcv + 4, mTF = mT1 + W1T * 512, NMIX = mTF + 2048, NMIX_FWD = mT1;
    static constexpr int NPACK = 2 * NL1 + NMIX + NMIX_FWD;  // online L1 | target L1 | online mix | target mix (forward part)
    // opt-in fp16 first layers (marlhip_qmix_mixer.l1_fp16): the same A operands as four halfs (8 bytes) per (k-group, tile, lane),
    // behind the fp32 packs: [online | target], NL1H float slots each
    static constexpr int NL1H = KS4 * MT1 * 64 * 2, NPACK_ALL = NPACK + 2 * NL1H;
};

template <class Q>
__host__ __device__ __forceinline__ int qmix_l1_row(int f) {  // canonical offset of row f of the combined first layer
    return f < 32 ? Q::oA1 + f * Q::SD : f < 64 ? Q::oAf + (f - 32) * Q::SD : f < 128 ? Q::oBb + (f - 64) * Q::SD : Q::oAv + (f - 128) * Q::SD;
}
template <class Q>
__host__ __device__ __forceinline__ int qmix_l1_bias(int f) {
    return f < 32 ? Q::oa1 + f : f < 64 ? Q::oaf + f - 32 : f < 128 ? Q::ocb + f - 64 : Q::oav + f - 128;
}

template <class Q>
__device__ __forceinline__ float qmix_l1_pack_elem(const float* __restrict__ w, int idx) {
    if (idx >= Q::pL1b) return w[qmix_l1_bias<Q>(idx - Q::pL1b)];
    const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
    const int mt = rest % Q::MT1, ks4 = rest / Q::MT1;
    const int f = 16 * mt + (lane & 15), k = 16 * ks4 + 4 * e + (lane >> 4);
    return k < Q::SD ? w[qmix_l1_row<Q>(f) + k] : 0.f;
}

template <class Q>
__device__ __forceinline__ float qmix_mix_pack_elem(const float* __restrict__ w, int idx) {
    constexpr int HE = Q::HE;
    if (idx < Q::mc1) {  // P1[mt2][k1][lane][r] = B1[16mt2+i][16k1+4g+r]
        const int r = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8, k1 = rest & 1, mt2 = rest >> 1;
        return w[Q::oB1 + (16 * mt2 + (lane & 15)) * HE + 16 * k1 + 4 * (lane >> 4) + r];
    }
    if (idx < Q::mPF) return w[Q::oc1 + idx - Q::mc1];
    if (idx < Q::mcf) {  // PF[mt][k1][lane][r] = Bf[16mt+i][16k1+4g+r]
        const int x = idx - Q::mPF, r = x & 3, lane = (x >> 2) & 63, rest = x >> 8, k1 = rest & 1, mt = rest >> 1;
        return w[Q::oBf + (16 * mt + (lane & 15)) * HE + 16 * k1 + 4 * (lane >> 4) + r];
    }
    if (idx < Q::mbv) return w[Q::ocf + idx - Q::mcf];
    if (idx < Q::mcv) return w[Q::obv + idx - Q::mbv];
    if (idx < Q::mT1) return idx == Q::mcv ? w[Q::ocv] : 0.f;
    if (idx < Q::mTF) {  // T1[m][kt][lane][r]: A[i = h1 feature 16m+i][k = w1 feature 16kt+4g+r] = B1[16kt+4g+r][16m+i]
        const int x = idx - Q::mT1, r = x & 3, lane = (x >> 2) & 63, rest = x >> 8, kt = rest % Q::W1T, m = rest / Q::W1T;
        return w[Q::oB1 + (16 * kt + 4 * (lane >> 4) + r) * HE + 16 * m + (lane & 15)];
    }
    const int x = idx - Q::mTF, r = x & 3, lane = (x >> 2) & 63, rest = x >> 8, kt = rest & 3, m = rest >> 2;
    return w[Q::oBf + (16 * kt + 4 * (lane >> 4) + r) * HE + 16 * m + (lane & 15)];
}

// draw_out (replay form, else null): the blocks behind the pack range draw the episode index of every batch row once per update (the mixer
// kernels then read it instead of re-running Philox) - in the same launch, the two jobs do not depend on each other
template <class Q>
__global__ __launch_bounds__(256) void qmix_pack_kernel(const float* __restrict__ mixer, const float* __restrict__ tmixer,
                                                        float* __restrict__ packs, ReplaySrc rs, int B, int32_t* __restrict__ draw_out) {
    constexpr int PACK_BLOCKS = (Q::NPACK + 255) / 256;
    if ((int)blockIdx.x >= PACK_BLOCKS) {
        const int b = (blockIdx.x - PACK_BLOCKS) * 256 + threadIdx.x;
        if (b < B) draw_out[b] = replay_draw(rs, b);
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Q::NPACK) return;
    float v;
    if (idx < Q::NL1) v = qmix_l1_pack_elem<Q>(mixer, idx);
    else if (idx < 2 * Q::NL1) v = qmix_l1_pack_elem<Q>(tmixer, idx - Q::NL1);
    else if (idx < 2 * Q::NL1 + Q::NMIX) v = qmix_mix_pack_elem<Q>(mixer, idx - 2 * Q::NL1);
    else v = qmix_mix_pack_elem<Q>(tmixer, idx - 2 * Q::NL1 - Q::NMIX);
    packs[idx] = v;
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
typedef short sh4 __attribute__((ext_vector_type(4)));
// weight-gradient operands of the opt-in low-precision mixer: bf16, not fp16 - per-row gradients have no bound an fp16 exponent could hold
// (a synthetic batch with raw integer observations reaches 1e5 per row: inf in fp16), and bf16 keeps small-integer states exact
__device__ __forceinline__ sh4 to_bf16x4(float a, float b, float c, float d) {
    bf4 v;
    v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    return __builtin_bit_cast(sh4, v);
}
#define MARL_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k((a), (b), (c), 0, 0, 0)

// the first-layer A operands rounded to fp16 (round-to-nearest-even), [online | target] behind the fp32 packs
template <class Q>
__global__ __launch_bounds__(256) void qmix_pack_half_kernel(const float* __restrict__ mixer, const float* __restrict__ tmixer, float* __restrict__ packs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;  // one (net, k-group, tile, lane) entry = 4 halfs
    constexpr int N = Q::KS4 * Q::MT1 * 64;
    if (idx >= 2 * N) return;
    const int net = idx / N, ent = idx - net * N;
    const float* w = net == 0 ? mixer : tmixer;
    h4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (_Float16)qmix_l1_pack_elem<Q>(w, 4 * ent + e);
    reinterpret_cast<h4*>(packs + Q::NPACK)[idx] = v;
}

// where the state rows come from: the reference-layout Batch obss[P][T+1][B][D] or the episode-major replay
template <class Q, bool REPLAY>
struct QmixRows {
    const float* obs;
    ReplaySrc rs;
    int T, B;
    // element (p, d) of the state of (row, time offset) = base(row, toff)[p * pstride() + d]
    __device__ __forceinline__ const float* base(int row, int toff) const {
        const int t0 = row / B, b = row - t0 * B, t = t0 + toff;
        if (REPLAY) {
            const int e = rs.idx ? rs.idx[b] : replay_draw(rs, b);
            return rs.rb.obs + ((size_t)e * Q::P * (T + 1) + t) * Q::D;
        }
        return obs + ((size_t)t * B + b) * Q::D;
    }
    __device__ __forceinline__ size_t pstride() const { return REPLAY ? (size_t)(T + 1) * Q::D : (size_t)(T + 1) * B * Q::D; }
};

template <class Q>
__device__ __forceinline__ size_t qmix_state_off(int k, size_t ps) {  // k < SD
    const int p = k / Q::D, d = k - p * Q::D;
    return (size_t)p * ps + d;
}

// ---------------------------------------------------------------------------------------------------------
// split form (shapes whose first-layer weights stream through LDS in many K chunks, MARL_QMIX_FUSE_NCH): first layers Y1[row][192]
// ---------------------------------------------------------------------------------------------------------
// HALF: the A operands come as fp16 (packh: h4 per entry) and the states are rounded to fp16 on the way in (LBF / warehouse observations
// are small integers: exact); one v_mfma_f32_16x16x16_f16 per (k-group, tile) instead of four f32 MFMAs, fp32 accumulation, fp32 bias.
// Round 5: the B operand (state rows) goes through LDS with the weights.  Until then every wave fetched it with 80 dword loads per block, each
// touching 16 rows x 16 bytes - 47 % of the kernel's wave cycles waiting on memory, the L2 missing on 46 % of its requests (rows re-fetched
// sector by sector, profiles/r05_qmix8p_sq_counters.md).  Now the chunk staging that brings 64 weight columns into LDS also brings the
// workgroup's 64 rows x the same 64 state columns: every load instruction is 64 consecutive columns of one row, every row sector is fetched
// once, and a lane's four k values of a group are one ds_read_b128 out of a [k-group][g][row][e] tile (g-stride padded to 68 floats: the
// transposing writes spread over the banks).  A operand, pack layout and the order of products per accumulator are unchanged: bitwise the
// results of the direct-load form.
template <class Q>
struct QmixL1Lds {
    static constexpr int WCH = 4 * Q::MT1 * 256;          // floats of one weight chunk (64 columns x 192 features)
    static constexpr int GS = 68;                         // floats per (k-group, g) slab of a wave's state tile: 16 rows x 4 e + 4 pad
    static constexpr int STILE = 4 * 4 * GS;              // one wave's tile: 4 k-groups x 4 g
    static constexpr int FLOATS = WCH + 4 * STILE;
};

template <class Q, bool REPLAY, bool HALF = false>
__global__ __launch_bounds__(256, 2) void qmix_l1_kernel(const float* __restrict__ pack, QmixRows<Q, REPLAY> src, int toff, int R,
                                                      float* __restrict__ Y1, const h4* __restrict__ packh = nullptr) {
    using LL = QmixL1Lds<Q>;
    constexpr int MT1 = Q::MT1, KS4 = Q::KS4, NCH = Q::NCH, SD = Q::SD;
    static_assert(MARL_QMIX_L1_NB == 1, "the LDS-staged state tile is one row block per wave");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f4* lds4 = reinterpret_cast<f4*>(lds);
    h4* ldsh = reinterpret_cast<h4*>(lds);
    float* stile = lds + LL::WCH;  // (the fp16 form's weight chunk takes half of the weight region; the state tiles stay where they are)
    const f4* pack4 = reinterpret_cast<const f4*>(pack);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int ngroups = (R + 63) / 64;
    const size_t ps = src.pstride();
    // staging role: column (tid & 63) of the chunk, rows (tid >> 6) + 4 i of the workgroup's 64; where the element lands: its row's wave tile,
    // slab (k-group, g) = (column / 16, column % 4), slot (row % 16) * 4 + (column % 16) / 4
    const int scol = tid & 63, srow0 = tid >> 6;
    const int sdst = ((scol >> 4) * 4 + (scol & 3)) * LL::GS + ((scol & 15) >> 2);
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int row = (grp * 4 + wave) * 16 + j;
        const float* rbs[16];  // the 16 rows this thread stages, for all chunks of the group
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = grp * 64 + srow0 + 4 * i;
            rbs[i] = src.base(r < R ? r : R - 1, toff);
        }
        f4 acc[MT1];
#pragma unroll
        for (int mt = 0; mt < MT1; ++mt) acc[mt] = *reinterpret_cast<const f4*>(pack + Q::pL1b + 16 * mt + 4 * g);
        // chunk c + 1's state column rides in registers while chunk c is multiplied (requested right behind the barrier that publishes chunk c;
        // the weight share as well was measured to spill: 48 more registers next to 48 accumulators and the 16 row pointers)
        float sv[16];
        bool kin = false;
        auto request = [&](int c) {
            const int k = 64 * c + scol;  // this thread's state column of the chunk (clamped: columns past SD are staged as zeros)
            kin = k < SD;
            const size_t off = qmix_state_off<Q>(kin ? k : SD - 1, ps);
#pragma unroll
            for (int i = 0; i < 16; ++i) sv[i] = rbs[i][off];
        };
        request(0);
#pragma unroll 1
        for (int c = 0; c < NCH; ++c) {  // (not unrolled: five copies of the prefetch with their own operand sets spill 750 B per lane)
            const int n4 = (KS4 - 4 * c) < 4 ? (KS4 - 4 * c) : 4;
            __syncthreads();  // the previous chunk's readers are done
            if (HALF) copy_f4_to_lds(reinterpret_cast<const f4*>(packh + c * 4 * MT1 * 64), lds4, n4 * MT1 * 64 / 2, tid, 256);
            else copy_f4_to_lds(pack4 + c * 4 * MT1 * 64, lds4, n4 * MT1 * 64, tid, 256);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int rl = srow0 + 4 * i;  // row inside the workgroup's 64: wave rl / 16, row rl % 16 of its tile
                stile[(rl >> 4) * LL::STILE + sdst + (rl & 15) * 4] = kin ? sv[i] : 0.f;
            }
            __syncthreads();
            if (c + 1 < NCH) request(c + 1);
            const float* mine = stile + wave * LL::STILE + g * LL::GS + j * 4;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                if (q4 < n4) {
                    const f4 x = *reinterpret_cast<const f4*>(mine + q4 * 4 * LL::GS);  // x[e] = state[row j][16 (4 c + q4) + 4 e + g]
                    if constexpr (HALF) {
                        h4 xh;
#pragma unroll
                        for (int e = 0; e < 4; ++e) xh[e] = (_Float16)x[e];
#pragma unroll
                        for (int mt = 0; mt < MT1; ++mt) {
                            const h4 a = ldsh[(q4 * MT1 + mt) * 64 + lane];
                            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(a, xh, acc[mt], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int mt = 0; mt < MT1; ++mt) {
                            const f4 a = lds4[(q4 * MT1 + mt) * 64 + lane];
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mt] = MARL_MFMA(a[e], x[e], acc[mt]);
                        }
                    }
                }
            }
        }
        if (row < R) {
            float* y = Y1 + (size_t)row * Q::NF1 + 4 * g;
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt)
                *reinterpret_cast<f4*>(y + 16 * mt) = (mt < 4 || mt >= 8) ? relu4(acc[mt]) : acc[mt];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// one mixer instance on one 16-row block per wave: first layers -> mixing network (-> TD error and backward)
// ---------------------------------------------------------------------------------------------------------
struct QmixIo {
    const float* chosen;  // [P][R]
    const float* tqsel;   // [P][R]
    const float* r0;      // [R] reward of agent 0 (QMixNetwork uses batch.rewards[0], model.py:379)
    const float* dn;      // [R]
    const float* fl;      // [R]
    float* dq;            // [P][R]
    float* lrow;          // [R]
    float* ytgt;          // [R] target-mixer output
    int ytgt_is_return;   // standardise_returns: ytgt already holds the standardised return (colstd_returns_kernel ran on it in place)
};

// backward operands, written by the online instance (Rp = rows padded to whole blocks, nblk = Rp/16):
//   G1T [nblk][192][16]  d(pre-activation) of the first layers, feature-major inside a block (an MFMA A tile)
//   DW1T[nblk][E*P][16]  d(pre-abs w1),  DWFT[nblk][64][16]  d(pre-abs w_final),  DY[Rp]  dL/dy
//   HB  [Rp][192]        the online first-layer activations the weight gradients multiply with: columns [0,32) h1, [32,64) hf,
//                        [128,192) hv (columns [64,128), hyper_b_1's output, are not needed again and not written)
struct QmixBwd {
    float* G1T;
    float* DW1T;
    float* DWFT;
    float* DY;
    float* HB;
};

template <int N>
__device__ __forceinline__ void qmix_store_tiles(float* dst, const f4 (&v)[N], int g, int j) {  // dst[(16mt+4g+r)*16 + j]
#pragma unroll
    for (int mt = 0; mt < N; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * mt + 4 * g + r) * 16 + j] = v[mt][r];
}

// The backward transposes of hyper_w_1.2 / hyper_w_final.2 (T1 | TF, online instance) stay in L2 instead of LDS when the forward pack, they
// and the first-layer window do not fit the 160 KB together (8 agents: 150 KB of mixing pack alone).
template <class Q>
constexpr int qmix_l1_window_floats(bool half) { return (Q::NCH == 1 ? Q::KS4 : 4) * Q::MT1 * 256 / (half ? 2 : 1); }
template <class Q>
constexpr bool qmix_t1_global(bool half) { return (Q::NMIX + qmix_l1_window_floats<Q>(half)) * 4 > 160 * 1024; }
template <class Q, bool ONLINE>
constexpr int qmix_net_lds_floats(bool half) {
    return ((ONLINE && !qmix_t1_global<Q>(half)) ? Q::NMIX : Q::NMIX_FWD) + qmix_l1_window_floats<Q>(half);
}

// The mixing network (and, ONLINE, the TD error and the backward) of one 16-row block on one wave, from the first-layer activations in the
// MFMA's C layout (feature 4g+r of tile k, row j).  lds: the staged mixing pack; TG: T1 | TF are read from the pack in global memory.
// OUT (online): where the operands of the weight-gradient GEMMs go - 0: G1T / DW1T / DWFT / DY in memory (split form: Y1 is there already),
// 1: the same plus HB, the activations they multiply with (fused form), 2: nowhere, they stay in registers (*out) for the in-kernel
// weight gradients.
template <class Q>
struct QmixTiles {
    f4 g1[12];        // d(pre-activation) of the first layers: [dh1 (2) | dhf (2) | dz (4) | dhv (4)]
    f4 dw1[Q::W1T];   // d(pre-abs w1)
    f4 dwf[4];        // d(pre-abs w_final)
    float dy;         // dL/dy of row j
};
template <class Q, bool ONLINE, bool TG, int OUT>
__device__ __forceinline__ void qmix_mix_block(const float* lds, const float* __restrict__ packMix, const f4 (&h1)[2], const f4 (&hf)[2], f4 (&z)[4],
                                               const f4 (&hv)[4], const QmixIo& io, int R, float gamma, const QmixBwd& bw, int blk, int lane,
                                               QmixTiles<Q>* out = nullptr) {
    constexpr int P = Q::P, W1T = Q::W1T;
    const int g = lane >> 4, j = lane & 15;
    const int row = blk * 16 + j;
    const bool ok = row < R;
    const int rc = ok ? row : R - 1;
    const f4* P1 = reinterpret_cast<const f4*>(lds + Q::mP1);
    const f4* PF = reinterpret_cast<const f4*>(lds + Q::mPF);
    auto t1_at = [&](int idx) -> f4 {  // T1 | TF entry (TF starts at (mTF - mT1) / 4)
        if constexpr (TG) return reinterpret_cast<const f4*>(packMix + Q::mT1)[idx];
        else return reinterpret_cast<const f4*>(lds + Q::mT1)[idx];
    };
    float q[P];
#pragma unroll
    for (int p = 0; p < P; ++p) q[p] = (ONLINE ? io.chosen : io.tqsel)[(size_t)p * R + rc];
    // w1 pre-abs = hyper_w_1.2 h1 + c1 ; z += q_p |w1_p|, agent by agent (the target instance keeps no w1 tiles)
    f4 w1[ONLINE ? W1T : 4];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            const int mt2 = 4 * p + et;
            f4 a2 = *reinterpret_cast<const f4*>(lds + Q::mc1 + 16 * mt2 + 4 * g);
#pragma unroll
            for (int k1 = 0; k1 < 2; ++k1) {
                const f4 a = P1[(mt2 * 2 + k1) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) a2 = MARL_MFMA(a[r], h1[k1][r], a2);
            }
            w1[ONLINE ? mt2 : et] = a2;
        }
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) z[et][r] = fmaf(q[p], fabsf(w1[ONLINE ? 4 * p + et : et][r]), z[et][r]);
        if (!ONLINE) __builtin_amdgcn_sched_barrier(0);  // keep the agents' tile groups from being interleaved (register pressure)
    }
    f4 ez[4], hid[4], wfp[4];
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ez[et][r] = expf(fminf(z[et][r], 0.f));
            hid[et][r] = z[et][r] > 0.f ? z[et][r] : ez[et][r] - 1.f;
        }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f4 a2 = *reinterpret_cast<const f4*>(lds + Q::mcf + 16 * mt + 4 * g);
#pragma unroll
        for (int k1 = 0; k1 < 2; ++k1) {
            const f4 a = PF[(mt * 2 + k1) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) a2 = MARL_MFMA(a[r], hf[k1][r], a2);
        }
        wfp[mt] = a2;
    }
    f4 bv[4];
    float yp = 0.f;
#pragma unroll
    for (int et = 0; et < 4; ++et) {
        bv[et] = *reinterpret_cast<const f4*>(lds + Q::mbv + 16 * et + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            yp = fmaf(hid[et][r], fabsf(wfp[et][r]), yp);
            yp = fmaf(hv[et][r], bv[et][r], yp);
        }
    }
    yp += __shfl_xor(yp, 16);
    yp += __shfl_xor(yp, 32);
    const float y = yp + lds[Q::mcv];
    if constexpr (!ONLINE) {
        if (g == 0 && ok) io.ytgt[row] = y;
    } else {
    // ---- TD error and backward (model.py:419-427)
    const float fl = ok ? io.fl[rc] : 0.f;
    const float delta = y - (io.ytgt_is_return ? io.ytgt[rc] : io.r0[rc] + gamma * io.ytgt[rc] * (1.f - io.dn[rc]));
    const float dy = 2.f * fl * delta;
    if (g == 0) {
        if constexpr (OUT != 2) bw.DY[blk * 16 + j] = dy;
        if (ok) io.lrow[row] = fl * delta * delta;
    }
    if constexpr (OUT == 1) {  // the activations the weight-gradient GEMMs multiply with (rows past R: finite values next to zero gradients)
        float* hb = bw.HB + (size_t)(blk * 16 + j) * Q::NF1 + 4 * g;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            *reinterpret_cast<f4*>(hb + 16 * k) = h1[k];
            *reinterpret_cast<f4*>(hb + 32 + 16 * k) = hf[k];
        }
#pragma unroll
        for (int et = 0; et < 4; ++et) *reinterpret_cast<f4*>(hb + 128 + 16 * et) = hv[et];
    }
    f4 dz[4], dwf[4], dhv[4];
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dhid = dy * fabsf(wfp[et][r]);
            dwf[et][r] = wfp[et][r] > 0.f ? dy * hid[et][r] : -(dy * hid[et][r]);
            dz[et][r] = z[et][r] > 0.f ? dhid : dhid * ez[et][r];
            dhv[et][r] = hv[et][r] > 0.f ? dy * bv[et][r] : 0.f;
        }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        float s = 0.f;
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float w = w1[4 * p + et][r];
                s = fmaf(dz[et][r], fabsf(w), s);
                const float dwp = dz[et][r] * q[p];
                w1[4 * p + et][r] = w > 0.f ? dwp : -dwp;  // w1 now holds d(pre-abs w1)
            }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0 && ok) io.dq[(size_t)p * R + row] = s;
    }
    if constexpr (OUT != 2) {
        qmix_store_tiles<W1T>(bw.DW1T + (size_t)blk * (Q::E * P * 16), w1, g, j);
        qmix_store_tiles<4>(bw.DWFT + (size_t)blk * (Q::E * 16), dwf, g, j);
    }
    f4 dh[4];  // [dh1 (2 tiles) | dhf (2 tiles)]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f4 a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < W1T; ++kt) {
            const f4 a = t1_at((m * W1T + kt) * 64 + lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) a1 = MARL_MFMA(a[r], w1[kt][r], a1);
        }
        f4 af = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            const f4 a = t1_at((Q::mTF - Q::mT1) / 4 + (m * 4 + kt) * 64 + lane);
#pragma unroll
            for (int r = 0; r < 4; ++r) af = MARL_MFMA(a[r], dwf[kt][r], af);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dh[m][r] = h1[m][r] > 0.f ? a1[r] : 0.f;
            dh[2 + m][r] = hf[m][r] > 0.f ? af[r] : 0.f;
        }
    }
    if constexpr (OUT == 2) {
        out->dy = dy;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out->g1[k] = dh[k];
            out->g1[4 + k] = dz[k];
            out->g1[8 + k] = dhv[k];
            out->dwf[k] = dwf[k];
        }
#pragma unroll
        for (int k = 0; k < W1T; ++k) out->dw1[k] = w1[k];
    } else {
        float* g1 = bw.G1T + (size_t)blk * (Q::NF1 * 16);
        qmix_store_tiles<4>(g1, dh, g, j);
        qmix_store_tiles<4>(g1 + 64 * 16, dz, g, j);
        qmix_store_tiles<4>(g1 + 128 * 16, dhv, g, j);
    }
    }
}

// In-kernel weight gradients (WG, online instance): per row block the operands of the three weight-gradient GEMMs are
// OPT = 20 + W1T tiles of 16 x 16 floats: [G1 (12) | dw1 (W1T) | dwf (4) | h1 (2) | hf (2)], feature-major (an MFMA A operand; the
// activations are read the same way as B operands).  They go through an LDS region of RB block slots behind the packs, RB = as many as
// fit (4, 2 or 1; 0: the shape keeps qmix_wgrad_kernel): the four waves of a workgroup step publish their blocks in 4 / RB rounds and, per
// round, every wave accumulates ITS accumulator tiles (a quarter of dW1cat, dB1, dBf) over the published blocks - G1T / DW1T / DWFT / HB
// (34-38 KB per row block, written and read back) never exist in memory.
template <class Q>
constexpr int qmix_op_tiles() { return 20 + Q::W1T; }
template <class Q>
constexpr int qmix_wg_round(bool half) {
    // (one K chunk only: with the chunk prefetch registers on top, the 3-agent 24/27-wide and the warehouse shapes spill 0.1-0.6 KB per lane)
    if (Q::NCH > 1 || qmix_t1_global<Q>(half) || MARL_QMIX_WG == 0) return 0;
    for (int rb = 4; rb >= 1; rb >>= 1)
        if ((Q::NMIX + qmix_l1_window_floats<Q>(half) + rb * qmix_op_tiles<Q>() * 256) * 4 <= 160 * 1024) return rb;
    return 0;
}

// HALF: the first-layer A operands come as fp16 (packh: h4 per entry) and the states are rounded to fp16 on the way in (LBF / warehouse
// observations are small integers: exact); one v_mfma_f32_16x16x16_f16 per (k-group, tile) instead of four f32 MFMAs, fp32 accumulation,
// fp32 bias.
//
// Per workgroup step the four waves take four consecutive row blocks.  Phase 1 (all waves in step: the weights stream through one LDS
// window in 64-column K chunks when they do not fit at once): acc[12] = the 192 first-layer features of the wave's 16 rows, in the MFMA's C
// layout - exactly the operand layout of phase 2, so Y1 never exists in memory.  Phase 2 (wave-independent): w1 = |hyper_w_1.2 h1 + c|,
// z = sum_p q_p w1_p + b1, hidden = elu(z), wf = |hyper_w_final.2 hf + c|, y = hidden.wf + V.2 hv + c; the online instance goes on with the
// TD error and the backward down to the first-layer pre-activations and writes dq_p and the operands of the weight-gradient GEMMs.
template <class Q, bool REPLAY, bool ONLINE, bool HALF = false, bool WG = false>
__global__ __launch_bounds__(256, (ONLINE || qmix_net_lds_floats<Q, false>(HALF) * 8 > 160 * 1024) ? 1 : 2) void qmix_net_kernel(const float* __restrict__ packL1, const h4* __restrict__ packh,
                                                                       const float* __restrict__ packMix, QmixRows<Q, REPLAY> src, QmixIo io,
                                                                       int R, float gamma, QmixBwd bw, float* __restrict__ partials = nullptr) {
    static_assert(!WG || ONLINE, "weight gradients belong to the online instance");
    constexpr int MT1 = Q::MT1, KS4 = Q::KS4, NCH = Q::NCH, SD = Q::SD;
    constexpr bool TG = ONLINE && qmix_t1_global<Q>(HALF);
    constexpr int NPK = (ONLINE && !TG) ? Q::NMIX : Q::NMIX_FWD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* win = lds + NPK;  // first-layer window
    f4* win4 = reinterpret_cast<f4*>(win);
    h4* winh = reinterpret_cast<h4*>(win);
    const f4* pack4 = reinterpret_cast<const f4*>(packL1);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    // window refill: chunk c = n4(c) * MT1 * 64 entries of 16 bytes (8 with HALF: two per f4).  With several chunks the NEXT chunk's loads are
    // issued before the current chunk's MFMAs and land in registers; they go to the window behind a barrier after the MFMAs (the window
    // always holds the chunk the next MFMA group needs: chunk 0 again at the end of a step, phase 2 does not touch it).
    constexpr int WIN4 = qmix_l1_window_floats<Q>(HALF) / 4, PRE = (WIN4 + 255) / 256;
    auto chunk_n4 = [](int c) { return ((KS4 - 4 * c) < 4 ? (KS4 - 4 * c) : 4) * MT1 * 64 / (HALF ? 2 : 1); };
    auto chunk_src = [&](int c) -> const f4* {
        return HALF ? reinterpret_cast<const f4*>(packh + c * 4 * MT1 * 64) : pack4 + c * 4 * MT1 * 64;
    };
    copy_f4_to_lds(reinterpret_cast<const f4*>(packMix), reinterpret_cast<f4*>(lds), NPK / 4, tid, 256);
    copy_f4_to_lds(chunk_src(0), win4, chunk_n4(0), tid, 256);
    __syncthreads();
    const size_t ps = src.pstride();
    const int nblk = (R + 15) / 16, ngroups = (nblk + 3) / 4;
    // ---- in-kernel weight gradients: this wave's accumulator tiles
    constexpr int W1T = Q::W1T, OPT = qmix_op_tiles<Q>(), RB = WG ? qmix_wg_round<Q>(HALF) : 1, NR = 4 / (RB > 0 ? RB : 1);
    constexpr int NG = KS4 >= 4 ? 4 : (KS4 >= 2 ? 2 : 1), MG = 4 / NG, MPW = MT1 / MG, NPW = (KS4 + NG - 1) / NG, M2W = W1T / 4;
    constexpr int tG1 = 0, tW1 = 12, tWF = 12 + W1T, tH1 = 16 + W1T, tHF = 18 + W1T;  // tile offsets inside a block slot
    float* OP = win + qmix_l1_window_floats<Q>(HALF);
    const int mg = wave % MG, ng = wave / MG;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 accW[WG ? MPW : 1][WG ? NPW : 1], accB1[WG ? M2W : 1][2], accBf[2], dbv_acc[4];
    float cs1[WG ? MPW : 1], csc1[WG ? M2W : 1], cscf = 0.f, dcv_acc = 0.f;
    unsigned soff[WG ? NPW : 1];
    bool sval[WG ? NPW : 1];
    if constexpr (WG) {
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
            cs1[m] = 0.f;
#pragma unroll
            for (int n = 0; n < NPW; ++n) accW[m][n] = zero4;
        }
#pragma unroll
        for (int m = 0; m < M2W; ++m) { csc1[m] = 0.f; accB1[m][0] = zero4; accB1[m][1] = zero4; }
        accBf[0] = zero4; accBf[1] = zero4;
#pragma unroll
        for (int et = 0; et < 4; ++et) dbv_acc[et] = zero4;
#pragma unroll
        for (int n = 0; n < NPW; ++n) {  // state column of this lane in each owned N tile of dW1cat
            const int k = 16 * (ng + n * NG) + j;
            sval[n] = (ng + n * NG) < KS4 && k < SD;
            soff[n] = (unsigned)qmix_state_off<Q>(k < SD ? k : SD - 1, ps);
        }
    }
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int blk = grp * 4 + wave;
        const int row = blk * 16 + j;
        const bool ok = row < R;
        const int rc = ok ? row : R - 1;
        // ---- phase 1: first layers
        f4 acc[MT1];
        {
            const float* rb = src.base(rc, ONLINE ? 0 : 1);
#pragma unroll
            for (int mt = 0; mt < MT1; ++mt) acc[mt] = *reinterpret_cast<const f4*>(packL1 + Q::pL1b + 16 * mt + 4 * g);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                constexpr int full = 4;
                const int n4 = (KS4 - 4 * c) < full ? (KS4 - 4 * c) : full;
                f4 pre[NCH > 1 ? PRE : 1];
                const int cn = c + 1 < NCH ? c + 1 : 0;
                if (NCH > 1) {
                    const f4* nsrc = chunk_src(cn);
#pragma unroll
                    for (int u = 0; u < PRE; ++u)
                        if (tid + 256 * u < chunk_n4(cn)) pre[u] = nsrc[tid + 256 * u];
                }
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    if (q4 < n4) {
                        float x[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int k = 16 * (4 * c + q4) + 4 * e + g;
                            const float v = rb[qmix_state_off<Q>(k < SD ? k : SD - 1, ps)];
                            x[e] = k < SD ? v : 0.f;
                        }
                        if constexpr (HALF) {
                            h4 xh;
#pragma unroll
                            for (int e = 0; e < 4; ++e) xh[e] = (_Float16)x[e];
#pragma unroll
                            for (int mt = 0; mt < MT1; ++mt)
                                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x16f16(winh[(q4 * MT1 + mt) * 64 + lane], xh, acc[mt], 0, 0, 0);
                        } else {
#pragma unroll
                            for (int mt = 0; mt < MT1; ++mt) {
                                const f4 a = win4[(q4 * MT1 + mt) * 64 + lane];
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[mt] = MARL_MFMA(a[e], x[e], acc[mt]);
                            }
                        }
                    }
                }
                if (NCH > 1) {
                    __syncthreads();  // every wave is done with the window
#pragma unroll
                    for (int u = 0; u < PRE; ++u)
                        if (tid + 256 * u < chunk_n4(cn)) win4[tid + 256 * u] = pre[u];
                    __syncthreads();
                }
            }
        }
        const bool active = blk < nblk;  // (wave-uniform)
        if (!WG && !active) continue;     // (all barriers of this form are in phase 1)
        // ---- phase 2: mixing network.  acc tiles: [0,2) hyper_w_1.0 | [2,4) hyper_w_final.0 | [4,8) hyper_b_1 | [8,12) V.0
        f4 h1[2], hf[2], z[4], hv[4];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            h1[k] = relu4(acc[k]);
            hf[k] = relu4(acc[2 + k]);
        }
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            z[et] = acc[4 + et];
            hv[et] = relu4(acc[8 + et]);
        }
        if constexpr (!WG) {
            qmix_mix_block<Q, ONLINE, TG, 1>(lds, packMix, h1, hf, z, hv, io, R, gamma, bw, blk, lane);
        } else {
            QmixTiles<Q> tl;
            if (active) {
                qmix_mix_block<Q, true, TG, 2>(lds, packMix, h1, hf, z, hv, io, R, gamma, bw, blk, lane, &tl);
                if (g == 0) dcv_acc += tl.dy;
#pragma unroll
                for (int et = 0; et < 4; ++et)
#pragma unroll
                    for (int r = 0; r < 4; ++r) dbv_acc[et][r] = fmaf(tl.dy, hv[et][r], dbv_acc[et][r]);  // dV.2: per-lane partial, rows folded at the end
            }
#pragma unroll
            for (int rd = 0; rd < NR; ++rd) {
                if (active && wave / RB == rd) {  // publish my block
                    float* T = OP + (wave % RB) * (OPT * 256);
                    tile_write<12>(T + tG1 * 256, tl.g1, g, j);
                    tile_write<W1T>(T + tW1 * 256, tl.dw1, g, j);
                    tile_write<4>(T + tWF * 256, tl.dwf, g, j);
                    tile_write<2>(T + tH1 * 256, h1, g, j);
                    tile_write<2>(T + tHF * 256, hf, g, j);
                }
                // the state rows of the round's blocks as B operands (row 4g+e at k-step e), requested before the barrier
                float bS[RB][NPW][4];
#pragma unroll
                for (int sl = 0; sl < RB; ++sl) {
                    const int b = grp * 4 + rd * RB + sl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int rw = b * 16 + 4 * g + e;
                        const float* rbs = src.base(rw < R ? rw : R - 1, 0);
#pragma unroll
                        for (int n = 0; n < NPW; ++n) bS[sl][n][e] = rbs[soff[n]];
                    }
                }
                __syncthreads();
#pragma unroll
                for (int sl = 0; sl < RB; ++sl) {
                    if (grp * 4 + rd * RB + sl < nblk) {
                        const float* T = OP + sl * (OPT * 256);
                        f4 a[MPW];
#pragma unroll
                        for (int m = 0; m < MPW; ++m) {
                            a[m] = tile_read(T + tG1 * 256, mg * MPW + m, g, j);
                            if (ng == 0) cs1[m] += (a[m][0] + a[m][1]) + (a[m][2] + a[m][3]);
                        }
#pragma unroll
                        for (int n = 0; n < NPW; ++n)
#pragma unroll
                            for (int m = 0; m < MPW; ++m)
#pragma unroll
                                for (int e = 0; e < 4; ++e) accW[m][n] = MARL_MFMA(a[m][e], sval[n] ? bS[sl][n][e] : 0.f, accW[m][n]);
                        const f4 b1[2] = {tile_read(T + tH1 * 256, 0, g, j), tile_read(T + tH1 * 256, 1, g, j)};
#pragma unroll
                        for (int m = 0; m < M2W; ++m) {
                            const f4 a2 = tile_read(T + tW1 * 256, wave + 4 * m, g, j);
                            csc1[m] += (a2[0] + a2[1]) + (a2[2] + a2[3]);
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                                for (int e = 0; e < 4; ++e) accB1[m][nt] = MARL_MFMA(a2[e], b1[nt][e], accB1[m][nt]);
                        }
                        const f4 a3 = tile_read(T + tWF * 256, wave, g, j);
                        cscf += (a3[0] + a3[1]) + (a3[2] + a3[3]);
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const f4 bf = tile_read(T + tHF * 256, nt, g, j);
#pragma unroll
                            for (int e = 0; e < 4; ++e) accBf[nt] = MARL_MFMA(a3[e], bf[e], accBf[nt]);
                        }
                    }
                }
                __syncthreads();  // the slots are rewritten by the next round / step
            }
        }
    }
    if constexpr (WG) {
        // ---- every wave owns disjoint entries of the workgroup's record (canonical mixer.parameters() offsets)
        float* rec = partials + (size_t)blockIdx.x * Q::NPARAM;
#pragma unroll
        for (int m = 0; m < MPW; ++m) {
            const int mt = mg * MPW + m;
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
                const int k = 16 * (ng + n * NG) + j;
                if ((ng + n * NG) < KS4 && k < SD) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) rec[qmix_l1_row<Q>(16 * mt + 4 * g + r) + k] = accW[m][n][r];
                }
            }
            if (ng == 0) {
                float sm = cs1[m];
                sm += __shfl_xor(sm, 16);
                sm += __shfl_xor(sm, 32);
                if (g == 0) rec[qmix_l1_bias<Q>(16 * mt + j)] = sm;
            }
        }
#pragma unroll
        for (int m = 0; m < M2W; ++m) {
            const int mt2 = wave + 4 * m;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rec[Q::oB1 + (16 * mt2 + 4 * g + r) * Q::HE + 16 * nt + j] = accB1[m][nt][r];
            float sm = csc1[m];
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            if (g == 0) rec[Q::oc1 + 16 * mt2 + j] = sm;
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) rec[Q::oBf + (16 * wave + 4 * g + r) * Q::HE + 16 * nt + j] = accBf[nt][r];
        {
            float sm = cscf;
            sm += __shfl_xor(sm, 16);
            sm += __shfl_xor(sm, 32);
            if (g == 0) rec[Q::ocf + 16 * wave + j] = sm;
        }
        // dV.2 weight and bias: rows (lanes j) folded per wave, the four waves' sums through the (now idle) operand region in fixed order
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = sum16(dbv_acc[et][r]);
                if (j == 0) OP[wave * 80 + 16 * et + 4 * g + r] = v;
            }
        {
            const float v = sum16(dcv_acc);  // (only the lanes of g == 0 carry it)
            if (lane == 0) OP[wave * 80 + 64] = v;
        }
        __syncthreads();
        if (tid < 65) rec[(tid < 64 ? Q::obv : Q::ocv - 64) + tid] = (OP[tid] + OP[80 + tid]) + (OP[160 + tid] + OP[240 + tid]);
    }
}

// split form: the mixing network of one 16-row block per wave from Y1 in memory
template <class Q, bool ONLINE>
__global__ __launch_bounds__(256, 1) void qmix_mix_kernel(const float* __restrict__ pack, const float* __restrict__ Y1, QmixIo io, int R,
                                                          float gamma, QmixBwd bw) {
    constexpr int NPK = ONLINE ? Q::NMIX : Q::NMIX_FWD;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    copy_f4_to_lds(reinterpret_cast<const f4*>(pack), reinterpret_cast<f4*>(lds), NPK / 4, tid, 256);
    __syncthreads();
    const int nblk = (R + 15) / 16;
    // the NEXT block's first-layer rows are requested before the current block's mixing network runs (round 5: one wave per SIMD here - the
    // 150 KB of mixing pack - so nothing else covers the 12 row loads at the top of a block)
    struct Rows { f4 h1[2], hf[2], z[4], hv[4]; };
    auto request = [&](int blk, Rows& r) {
        const int row = blk * 16 + j;
        const float* y1 = Y1 + (size_t)(row < R ? row : R - 1) * Q::NF1 + 4 * g;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            r.h1[k] = *reinterpret_cast<const f4*>(y1 + 16 * k);
            r.hf[k] = *reinterpret_cast<const f4*>(y1 + 32 + 16 * k);
        }
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            r.z[et] = *reinterpret_cast<const f4*>(y1 + 64 + 16 * et);
            r.hv[et] = *reinterpret_cast<const f4*>(y1 + 128 + 16 * et);
        }
    };
    const int stride = gridDim.x * 4;
    int blk = blockIdx.x * 4 + wave;
    Rows cur, nxt;
    if (blk < nblk) request(blk, cur);
    for (; blk < nblk; blk += stride) {
        request(blk + stride < nblk ? blk + stride : blk, nxt);
        qmix_mix_block<Q, ONLINE, false, 0>(lds, pack, cur.h1, cur.hf, cur.z, cur.hv, io, R, gamma, bw, blk, lane);
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradients: dW1cat[192][SD] = G1^T S (kernel 1), dB1[E*P][32] = DW1^T h1, dBf[64][32] = DWF^T hf, dV.2 =
// sum_rows dy hv (kernel 2); bias gradients = column sums of the same A operands.  Workgroup = 8 waves over a
// contiguous range of row blocks; every wave accumulates its own tiles in registers (K order inside a block: row
// 4g+e at k-step e, the same on both operands).  No LDS: operands come straight from L2 / HBM and the NEXT block's
// operands are requested before the current block's MFMAs (the loop is latency-bound otherwise: 24-150 MFMAs per wave
// and block).  Both kernels write disjoint entries of the same per-workgroup record.
// ---------------------------------------------------------------------------------------------------------
// H16 (the opt-in marlhip_qmix_mixer.l1_fp16, BASELINE config 5's "fp16 mixer on MFMA"): both operands of a block rounded to bf16 (see
// to_bf16x4) and ONE v_mfma_f32_16x16x16_bf16 per tile pair instead of four f32 MFMAs (the lane's four k values of a block are exactly the
// 4-element operand), fp32 accumulation; the bias gradients (column sums) stay fp32.
// Round 5: the first group's operands go through LDS.  Until then every wave fetched its own B operand - 20 dword loads per block, each
// touching 16 rows x 16 bytes, behind a per-lane episode-index load, with ONE block of prefetch because accumulators (120) and two operand
// sets (88) left no registers for a second - and ran at mfma_busy 0.26 (profiles/r05_qmix8p_sq_counters.md).  Now the WORKGROUP stages a
// block once: the state tile column-major (TS[col][16 rows]: a lane's four k values of a column are one ds_read_b128, and the two m-groups
// that read the same states from L2 separately share it) and the G1T tile as it lies in memory; a thread carries 10 + 8 staged values of the
// NEXT block in registers while the current one is multiplied out of the other LDS buffer - one barrier per block.  Same products in the same
// order per accumulator: results are bitwise those of the register form.
template <class Q>
struct QmixWg1Lds {
    static constexpr int SDP = Q::KS4 * 16;                  // state columns padded to whole tiles
    static constexpr int TS = SDP * 16, TA = Q::NF1 * 16;    // floats per buffer
    static constexpr int FLOATS = 2 * (TS + TA);
    static constexpr int NST = (SDP + 31) / 32;              // state elements a thread stages per block (512 threads: 16 rows x 32 columns a pass)
    static constexpr int NA4 = (TA / 4 + 511) / 512;         // float4s of the G1T tile per thread
};

template <class Q, bool REPLAY, bool H16 = false>
__device__ __forceinline__ void qmix_wgrad1_body(const QmixRows<Q, REPLAY>& src, const QmixBwd& bw, int R, float* __restrict__ partials, int bid, int nwg,
                                                 float* lds) {
    using LW = QmixWg1Lds<Q>;
    constexpr int SD = Q::SD, NTS = Q::KS4, NF1 = Q::NF1;
    constexpr int MG = NTS >= 4 ? 2 : 4, NG = 8 / MG, MPW = Q::MT1 / MG, NPW = (NTS + NG - 1) / NG;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const int mg = w % MG, ng = w / MG;
    const size_t ps = src.pstride();
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 accW[MPW][NPW];
    float cs1[MPW];
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
        cs1[m] = 0.f;
#pragma unroll
        for (int n = 0; n < NPW; ++n) accW[m][n] = zero4;
    }
    // staging role of this thread: row sr of the block, columns sc + 32 i
    const int sr = tid & 15, sc = tid >> 4;
    unsigned soff[LW::NST];
#pragma unroll
    for (int i = 0; i < LW::NST; ++i) {
        const int c = sc + 32 * i;
        soff[i] = (unsigned)qmix_state_off<Q>(c < SD ? c : SD - 1, ps);  // (columns past SD: a valid element; they only reach accumulator columns that are never written)
    }
    const int nblk = (R + 15) / 16;
    const int per = (nblk + nwg - 1) / nwg;
    const int bA = bid * per, bB = (bA + per) < nblk ? bA + per : nblk;
    float sv[LW::NST];
    f4 av[LW::NA4];
    auto request = [&](int blk) {  // this thread's share of block blk into registers
        const int row = blk * 16 + sr;
        const float* rb = src.base(row < R ? row : R - 1, 0);
#pragma unroll
        for (int i = 0; i < LW::NST; ++i) sv[i] = rb[soff[i]];
        const f4* ga = reinterpret_cast<const f4*>(bw.G1T + (size_t)blk * (NF1 * 16));
#pragma unroll
        for (int i = 0; i < LW::NA4; ++i) av[i] = (tid + 512 * i) < LW::TA / 4 ? ga[tid + 512 * i] : zero4;
    };
    auto publish = [&](int buf) {  // registers -> LDS buffer `buf`
        float* ts = lds + buf * (LW::TS + LW::TA);
        f4* ta = reinterpret_cast<f4*>(ts + LW::TS);
#pragma unroll
        for (int i = 0; i < LW::NST; ++i)
            if (sc + 32 * i < LW::SDP) ts[(sc + 32 * i) * 16 + sr] = sv[i];
#pragma unroll
        for (int i = 0; i < LW::NA4; ++i)
            if ((tid + 512 * i) < LW::TA / 4) ta[tid + 512 * i] = av[i];
    };
    auto mac = [&](int buf) {
        const float* ts = lds + buf * (LW::TS + LW::TA);
        const float* ta = ts + LW::TS;
        f4 a[MPW], b[NPW];
#pragma unroll
        for (int m = 0; m < MPW; ++m) a[m] = *reinterpret_cast<const f4*>(ta + (16 * (mg * MPW + m) + j) * 16 + 4 * g);
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int tile = ng + n * NG;
            b[n] = *reinterpret_cast<const f4*>(ts + (16 * (tile < NTS ? tile : NTS - 1) + j) * 16 + 4 * g);
        }
#pragma unroll
        for (int m = 0; m < MPW; ++m)
            if (ng == 0) cs1[m] += (a[m][0] + a[m][1]) + (a[m][2] + a[m][3]);
        if constexpr (H16) {
            sh4 ah[MPW], bh[NPW];
#pragma unroll
            for (int m = 0; m < MPW; ++m) ah[m] = to_bf16x4(a[m][0], a[m][1], a[m][2], a[m][3]);
#pragma unroll
            for (int n = 0; n < NPW; ++n) bh[n] = to_bf16x4(b[n][0], b[n][1], b[n][2], b[n][3]);
#pragma unroll
            for (int n = 0; n < NPW; ++n)
#pragma unroll
                for (int m = 0; m < MPW; ++m) accW[m][n] = MARL_MFMA_BF16(ah[m], bh[n], accW[m][n]);
        } else {
#pragma unroll
            for (int n = 0; n < NPW; ++n)
#pragma unroll
                for (int m = 0; m < MPW; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) accW[m][n] = MARL_MFMA(a[m][e], b[n][e], accW[m][n]);
        }
    };
    if (bA < bB) {
        request(bA);
        publish(0);
        if (bA + 1 < bB) request(bA + 1);
        __syncthreads();
        for (int blk = bA; blk < bB; ++blk) {
            const int buf = (blk - bA) & 1;
            if (blk + 1 < bB) publish(buf ^ 1);      // block blk + 1 (requested one block ago); its buffer was last read two blocks ago, behind a barrier
            if (blk + 2 < bB) request(blk + 2);
            mac(buf);
            __syncthreads();
        }
    }
    float* rec = partials + (size_t)bid * Q::NPARAM;
#pragma unroll
    for (int m = 0; m < MPW; ++m) {
        const int mt = mg * MPW + m;
#pragma unroll
        for (int n = 0; n < NPW; ++n) {
            const int k = 16 * (ng + n * NG) + j;
            if ((ng + n * NG) < NTS && k < SD) {
#pragma unroll
                for (int r = 0; r < 4; ++r) rec[qmix_l1_row<Q>(16 * mt + 4 * g + r) + k] = accW[m][n][r];
            }
        }
        if (ng == 0) {
            float s = cs1[m];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) rec[qmix_l1_bias<Q>(16 * mt + j)] = s;
        }
    }
}

template <class Q, bool H16 = false>
__device__ __forceinline__ void qmix_wgrad2_body(const QmixBwd& bw, int R, float* __restrict__ partials, int bid, int nwg) {
    const float* __restrict__ Y1 = bw.HB;
    constexpr int P = Q::P, W1T = Q::W1T, NF1 = Q::NF1, M2W = (W1T + 7) / 8;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, g = lane >> 4, j = lane & 15;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 accB1[M2W][2], accBf = zero4, dbv = zero4;
    float csc1[M2W], cscf = 0.f, dcv = 0.f;
#pragma unroll
    for (int m = 0; m < M2W; ++m) {
        csc1[m] = 0.f;
        accB1[m][0] = zero4;
        accB1[m][1] = zero4;
    }
    struct Ops {
        f4 a2[M2W], a3;
        float hb[2][4], hf[4], dy[4], hv[4][4];
    };
    auto load = [&](int blk, Ops& o) {
#pragma unroll
        for (int m = 0; m < M2W; ++m) {
            const int mt2 = w + 8 * m;
            o.a2[m] = mt2 < W1T ? *reinterpret_cast<const f4*>(bw.DW1T + (size_t)blk * (Q::E * P * 16) + (16 * mt2 + j) * 16 + 4 * g) : zero4;
        }
        o.a3 = *reinterpret_cast<const f4*>(bw.DWFT + (size_t)blk * (Q::E * 16) + (16 * (w >> 1) + j) * 16 + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = blk * 16 + 4 * g + e;
            const float* y = Y1 + (size_t)(row < R ? row : R - 1) * NF1;
            o.hb[0][e] = y[j];
            o.hb[1][e] = y[16 + j];
            o.hf[e] = y[32 + 16 * (w & 1) + j];
            if (w == 7) {
                o.dy[e] = bw.DY[blk * 16 + 4 * g + e];
#pragma unroll
                for (int et = 0; et < 4; ++et) o.hv[et][e] = y[128 + 16 * et + j];
            }
        }
    };
    const int nblk = (R + 15) / 16;
    const int per = (nblk + nwg - 1) / nwg;
    const int bA = bid * per, bB = (bA + per) < nblk ? bA + per : nblk;
    Ops cur, nxt;
    if (bA < bB) load(bA, cur);
    for (int blk = bA; blk < bB; ++blk) {
        load(blk + 1 < bB ? blk + 1 : blk, nxt);
        sh4 hbh[2], hfh;
        if constexpr (H16) {
            hbh[0] = to_bf16x4(cur.hb[0][0], cur.hb[0][1], cur.hb[0][2], cur.hb[0][3]);
            hbh[1] = to_bf16x4(cur.hb[1][0], cur.hb[1][1], cur.hb[1][2], cur.hb[1][3]);
            hfh = to_bf16x4(cur.hf[0], cur.hf[1], cur.hf[2], cur.hf[3]);
        }
#pragma unroll
        for (int m = 0; m < M2W; ++m) {
            csc1[m] += (cur.a2[m][0] + cur.a2[m][1]) + (cur.a2[m][2] + cur.a2[m][3]);
            if constexpr (H16) {
                const sh4 a2h = to_bf16x4(cur.a2[m][0], cur.a2[m][1], cur.a2[m][2], cur.a2[m][3]);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) accB1[m][nt] = MARL_MFMA_BF16(a2h, hbh[nt], accB1[m][nt]);
            } else {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) accB1[m][nt] = MARL_MFMA(cur.a2[m][e], cur.hb[nt][e], accB1[m][nt]);
            }
        }
        if ((w & 1) == 0) cscf += (cur.a3[0] + cur.a3[1]) + (cur.a3[2] + cur.a3[3]);
        if constexpr (H16) {
            accBf = MARL_MFMA_BF16(to_bf16x4(cur.a3[0], cur.a3[1], cur.a3[2], cur.a3[3]), hfh, accBf);
        } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) accBf = MARL_MFMA(cur.a3[e], cur.hf[e], accBf);
        }
        if (w == 7) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dcv += cur.dy[e];
#pragma unroll
                for (int et = 0; et < 4; ++et) dbv[et] = fmaf(cur.dy[e], cur.hv[et][e], dbv[et]);
            }
        }
        cur = nxt;
    }
    float* rec = partials + (size_t)bid * Q::NPARAM;
#pragma unroll
    for (int m = 0; m < M2W; ++m) {
        const int mt2 = w + 8 * m;
        if (mt2 < W1T) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) rec[Q::oB1 + (16 * mt2 + 4 * g + r) * Q::HE + 16 * nt + j] = accB1[m][nt][r];
            float s = csc1[m];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) rec[Q::oc1 + 16 * mt2 + j] = s;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) rec[Q::oBf + (16 * (w >> 1) + 4 * g + r) * Q::HE + 16 * (w & 1) + j] = accBf[r];
    if ((w & 1) == 0) {
        float s = cscf;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (g == 0) rec[Q::ocf + 16 * (w >> 1) + j] = s;
    }
    if (w == 7) {
#pragma unroll
        for (int et = 0; et < 4; ++et) {
            float s = dbv[et];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (g == 0) rec[Q::obv + 16 * et + j] = s;
        }
        float s = j == 0 ? dcv : 0.f;
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        if (lane == 0) rec[Q::ocv] = s;
    }
}

// both weight-gradient GEMM groups in one launch: workgroups [0, nwg) take the first-layer gradients, [nwg, 2 nwg) the mixing-network ones;
// workgroup b of either group writes its (disjoint) entries of record b.  Each group alone is latency-bound on its operand stream and leaves
// most of the chip idle; side by side they overlap.
template <class Q, bool REPLAY, bool H16 = false>
__global__ __launch_bounds__(512, 1) void qmix_wgrad_kernel(QmixRows<Q, REPLAY> src, QmixBwd bw, int R, float* __restrict__ partials, int only = 0) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // QmixWg1Lds<Q>::FLOATS (the first group's operand tiles)
    const int nwg = gridDim.x >> 1;
    if (only == 1 && (int)blockIdx.x >= nwg) return;  // (diagnostics, MARLHIP_QMIX_WG_ONLY: time one group alone; the gradient is then incomplete)
    if (only == 2 && (int)blockIdx.x < nwg) return;
    if ((int)blockIdx.x < nwg) qmix_wgrad1_body<Q, REPLAY, H16>(src, bw, R, partials, blockIdx.x, nwg, lds);
    else qmix_wgrad2_body<Q, H16>(bw, R, partials, blockIdx.x - nwg, nwg);
}

// mixer_grad[i] = (sum over records, fixed order) / n_filled ; n_filled = nf[1] as written by dqn_reduce_kernel
static __global__ __launch_bounds__(256) void qmix_reduce_kernel(const float* __restrict__ partials, int nwg, int nparam,
                                                          const float* __restrict__ loss_nf, float* __restrict__ grad) {
    __shared__ float s_part[4][64];
    const int l64 = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + l64;
    float acc = 0.f;
    if (i < nparam) {
#pragma unroll 8
        for (int wg = slice; wg < nwg; wg += 4) acc += partials[(size_t)wg * nparam + i];
    }
    s_part[slice][l64] = acc;
    __syncthreads();
    if (slice == 0 && i < nparam) grad[i] = ((s_part[0][l64] + s_part[1][l64]) + (s_part[2][l64] + s_part[3][l64])) / loss_nf[1];
}

// ---- host side ---------------------------------------------------------------------------------------------
struct QmixCtx {  // what marlhip_qmix_loss_grad adds to the agent-network call
    const float* mixer;
    const float* tmixer;
    float* mgrad;
    void* ws;  // qmix part of the workspace
    int64_t ws_bytes;
    const RetStats* rst = nullptr;  // standardise_returns: per-batch-column statistics (ret_stats.h), or null
    bool l1_fp16 = false;           // marlhip_qmix_mixer.l1_fp16: first layers of both mixers on the fp16 MFMA
    bool generic = false;           // the mixer of qmix_gen.h (hypernet_layers 1, widths beyond 64 / 32, (agents, obs) pairs without a compiled kernel)
    QmixGenDims gen = {};
};

struct QmixWs {
    int64_t packs, hb, g1t, dw1, dwf, dy, ytgt, idx, partials, total;  // byte offsets
    int nwg3;  // workgroups per group of qmix_wgrad_kernel
    int nrec;  // per-workgroup gradient records qmix_reduce_kernel sums (the online instance's grid with in-kernel weight gradients)
};

template <class Q>
inline QmixWs qmix_ws_layout(int T, int B) {
    const int64_t R = (int64_t)T * B, nblk = (R + 15) / 16, Rp = nblk * 16;
    QmixWs w;
    auto al = [](int64_t x) { return (x + 255) & ~(int64_t)255; };
    w.packs = 0;
    w.hb = al(w.packs + (int64_t)Q::NPACK_ALL * 4);  // QmixBwd::HB
    w.g1t = al(w.hb + Rp * Q::NF1 * 4);              // QmixBwd::G1T
    w.dw1 = al(w.g1t + Rp * Q::NF1 * 4);
    w.dwf = al(w.dw1 + Rp * Q::E * Q::P * 4);
    w.dy = al(w.dwf + Rp * Q::E * 4);
    w.ytgt = al(w.dy + Rp * 4);
    w.idx = al(w.ytgt + Rp * 4);
    w.partials = al(w.idx + (int64_t)B * 4);
    // weight-gradient workgroups: the loop over row blocks is latency-bound when a block carries few MFMAs (small P), so the
    // light shapes get 3 workgroups per CU; the heavy ones (register-bound, 1 workgroup per CU) one per CU
    const int64_t cap = Q::P <= 4 ? 768 : 256;
    w.nwg3 = (int)(nblk < cap ? nblk : cap);
    const int64_t ngroups = (nblk + 3) / 4;
    w.nrec = qmix_wg_round<Q>(false) > 0 ? (int)(ngroups < 256 ? ngroups : 256) : w.nwg3;
    w.total = al(w.partials + (int64_t)w.nwg3 * Q::NPARAM * 4);
    return w;
}

// the mixer stage between the agent forward and backward passes
template <class Q, bool REPLAY>
int qmix_launch_mix(const QmixCtx& qx, const marlhip_batch* bt, const ReplaySrc& rsrc, const QmixIo& io, float gamma, hipStream_t st) {
    const int T = bt->max_len, B = bt->batch, R = T * B;
    const QmixWs wl = qmix_ws_layout<Q>(T, B);
    MARL_REQUIRE(qx.ws_bytes >= wl.total, "qmix_loss_grad: mixer workspace %lld < %lld bytes", (long long)qx.ws_bytes, (long long)wl.total);
    char* base = static_cast<char*>(qx.ws);
    float* packs = reinterpret_cast<float*>(base + wl.packs);
    QmixBwd bw;
    bw.G1T = reinterpret_cast<float*>(base + wl.g1t);
    bw.DW1T = reinterpret_cast<float*>(base + wl.dw1);
    bw.DWFT = reinterpret_cast<float*>(base + wl.dwf);
    bw.DY = reinterpret_cast<float*>(base + wl.dy);
    QmixIo io2 = io;
    io2.ytgt = reinterpret_cast<float*>(base + wl.ytgt);
    QmixRows<Q, REPLAY> src;
    src.obs = bt->obss; src.rs = rsrc; src.T = T; src.B = B;
    int32_t* idxbuf = nullptr;
    if (REPLAY && rsrc.idx == nullptr) {  // drawn by the pack launch below
        idxbuf = reinterpret_cast<int32_t*>(base + wl.idx);
        src.rs.idx = idxbuf;
    }
    bw.HB = reinterpret_cast<float*>(base + wl.hb);
    hipLaunchKernelGGL((qmix_pack_kernel<Q>), dim3((Q::NPACK + 255) / 256 + (idxbuf != nullptr ? (B + 255) / 256 : 0)), dim3(256), 0, st, qx.mixer,
                       qx.tmixer, packs, rsrc, B, idxbuf);
    const h4* packh = reinterpret_cast<const h4*>(packs + Q::NPACK);  // [online | target], Q::KS4 * Q::MT1 * 64 entries each
    if (qx.l1_fp16)
        hipLaunchKernelGGL((qmix_pack_half_kernel<Q>), dim3((2 * Q::KS4 * Q::MT1 * 64 + 255) / 256), dim3(256), 0, st, qx.mixer, qx.tmixer, packs);
    const int nblk = (R + 15) / 16;
    const float* l1o = packs, *l1t = packs + Q::NL1, *mxo = packs + 2 * Q::NL1, *mxt = packs + 2 * Q::NL1 + Q::NMIX;
    const h4* packh_t = packh + Q::KS4 * Q::MT1 * 64;
    auto standardise = [&]() -> int {  // QMixNetwork with standardise_returns: the target mixer's output becomes the standardised return
        if (qx.rst == nullptr) return 0;
        const int rc = launch_colstd(T, B, gamma, *qx.rst, io2.ytgt, 1, 0, io2.r0, io2.dn, io2.ytgt, st);
        io2.ytgt_is_return = 1;
        return rc;
    };
    timing_begin(TIMER_QMIX, st);
    if constexpr (Q::NCH <= MARL_QMIX_FUSE_NCH) {
        // fused form: one launch per mixer instance, the first-layer activations stay in registers
        constexpr bool WG = qmix_wg_round<Q>(false) > 0;  // weight gradients inside the online instance (then also with the smaller fp16 window)
        constexpr int OPF = qmix_op_tiles<Q>() * 256;
        constexpr int L_ON = (qmix_net_lds_floats<Q, true>(false) + (WG ? qmix_wg_round<Q>(false) * OPF : 0)) * 4;
        constexpr int L_ONH = (qmix_net_lds_floats<Q, true>(true) + (WG ? qmix_wg_round<Q>(true) * OPF : 0)) * 4;
        constexpr int L_TG = qmix_net_lds_floats<Q, false>(false) * 4, L_TGH = qmix_net_lds_floats<Q, false>(true) * 4;
        static_assert(L_ON <= 160 * 1024 && L_ONH <= 160 * 1024 && L_TG <= 160 * 1024, "qmix_net_kernel: LDS");
        static LdsAttr attr_set;
        if (attr_set.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_net_kernel<Q, REPLAY, true, false, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, L_ON);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_net_kernel<Q, REPLAY, false>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TG);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_net_kernel<Q, REPLAY, true, true, WG>), hipFuncAttributeMaxDynamicSharedMemorySize, L_ONH);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_net_kernel<Q, REPLAY, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, L_TGH);
            attr_set.done();
        }
        // resident workgroups: one per CU for the online instance (registers), as many as the LDS allows (at most 2) for the target's
        const int ngroups = (nblk + 3) / 4, per_cu_t = 2 * L_TG <= 160 * 1024 ? 2 : 1;
        const int g_t = ngroups < 256 * per_cu_t ? ngroups : 256 * per_cu_t, g_o = ngroups < 256 ? ngroups : 256;
        if (qx.l1_fp16)
            hipLaunchKernelGGL((qmix_net_kernel<Q, REPLAY, false, true>), dim3(g_t), dim3(256), L_TGH, st, l1t, packh_t, mxt, src, io2, R, gamma, bw);
        else
            hipLaunchKernelGGL((qmix_net_kernel<Q, REPLAY, false>), dim3(g_t), dim3(256), L_TG, st, l1t, (const h4*)nullptr, mxt, src, io2, R, gamma, bw);
        if (standardise() != 0) return -1;
        float* recs = reinterpret_cast<float*>(base + wl.partials);
        if (qx.l1_fp16)
            hipLaunchKernelGGL((qmix_net_kernel<Q, REPLAY, true, true, WG>), dim3(g_o), dim3(256), L_ONH, st, l1o, packh, mxo, src, io2, R, gamma, bw, recs);
        else
            hipLaunchKernelGGL((qmix_net_kernel<Q, REPLAY, true, false, WG>), dim3(g_o), dim3(256), L_ON, st, l1o, (const h4*)nullptr, mxo, src, io2, R, gamma,
                               bw, recs);
        if (WG) {  // the records are complete: g_o == wl.nrec of them
            timing_end(TIMER_QMIX, st);
            MARL_CHECK_LAUNCH("qmix mixer stage");
            return 0;
        }
    } else {
        // split form: Y1 of either instance through memory (target: the G1T buffer, dead until the online backward; online: HB, all 192 columns)
        constexpr int CH = QmixL1Lds<Q>::FLOATS * (int)sizeof(float);  // weight chunk + the four waves' state tiles
        constexpr int LM_ON = Q::NMIX * (int)sizeof(float), LM_TG = Q::NMIX_FWD * (int)sizeof(float);
        static LdsAttr attr_set;
        if (attr_set.need()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_l1_kernel<Q, REPLAY>), hipFuncAttributeMaxDynamicSharedMemorySize, CH);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_l1_kernel<Q, REPLAY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, CH);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_mix_kernel<Q, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LM_ON);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_mix_kernel<Q, false>), hipFuncAttributeMaxDynamicSharedMemorySize, LM_TG);
            attr_set.done();
        }
        const int ngroups = (R + 64 * MARL_QMIX_L1_NB - 1) / (64 * MARL_QMIX_L1_NB);
        // two workgroups of the first-layer kernel fit a CU (65 KB of LDS, <= 256 registers): 512 resident ones walk the groups - a third per CU
        // would only start when the first have finished (768 of them left half the chip idle for a third of the launch)
        const int g1 = ngroups < 512 ? ngroups : 512;
        const int g2 = (nblk + 3) / 4 < 256 ? (nblk + 3) / 4 : 256;
        float* y1t = bw.G1T, *y1o = bw.HB;
        if (qx.l1_fp16)
            hipLaunchKernelGGL((qmix_l1_kernel<Q, REPLAY, true>), dim3(g1), dim3(256), CH, st, l1t, src, 1, R, y1t, packh_t);
        else
            hipLaunchKernelGGL((qmix_l1_kernel<Q, REPLAY>), dim3(g1), dim3(256), CH, st, l1t, src, 1, R, y1t, (const h4*)nullptr);
        hipLaunchKernelGGL((qmix_mix_kernel<Q, false>), dim3(g2), dim3(256), LM_TG, st, mxt, (const float*)y1t, io2, R, gamma, bw);
        if (standardise() != 0) return -1;
        if (qx.l1_fp16)
            hipLaunchKernelGGL((qmix_l1_kernel<Q, REPLAY, true>), dim3(g1), dim3(256), CH, st, l1o, src, 0, R, y1o, packh);
        else
            hipLaunchKernelGGL((qmix_l1_kernel<Q, REPLAY>), dim3(g1), dim3(256), CH, st, l1o, src, 0, R, y1o, (const h4*)nullptr);
        hipLaunchKernelGGL((qmix_mix_kernel<Q, true>), dim3(g2), dim3(256), LM_ON, st, mxo, (const float*)y1o, io2, R, gamma, bw);
    }
    constexpr int LW_BYTES = QmixWg1Lds<Q>::FLOATS * (int)sizeof(float);
    static_assert(LW_BYTES <= 160 * 1024, "qmix_wgrad_kernel: LDS");
    static LdsAttr attr_wg;
    if (attr_wg.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_wgrad_kernel<Q, REPLAY, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LW_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qmix_wgrad_kernel<Q, REPLAY>), hipFuncAttributeMaxDynamicSharedMemorySize, LW_BYTES);
        attr_wg.done();
    }
    if (qx.l1_fp16)  // the opt-in covers the mixer's weight-gradient products too (round 4)
        hipLaunchKernelGGL((qmix_wgrad_kernel<Q, REPLAY, true>), dim3(2 * wl.nwg3), dim3(512), LW_BYTES, st, src, bw, R, reinterpret_cast<float*>(base + wl.partials));
    else
        hipLaunchKernelGGL((qmix_wgrad_kernel<Q, REPLAY>), dim3(2 * wl.nwg3), dim3(512), LW_BYTES, st, src, bw, R, reinterpret_cast<float*>(base + wl.partials),
                           getenv("MARLHIP_QMIX_WG_ONLY") ? atoi(getenv("MARLHIP_QMIX_WG_ONLY")) : 0);
    timing_end(TIMER_QMIX, st);
    MARL_CHECK_LAUNCH("qmix mixer stage");
    return 0;
}

// after dqn_reduce_kernel has written loss[1] = sum(filled)
template <class Q>
int qmix_launch_reduce(const QmixCtx& qx, int T, int B, const float* loss, hipStream_t st) {
    const QmixWs wl = qmix_ws_layout<Q>(T, B);
    hipLaunchKernelGGL(qmix_reduce_kernel, dim3((Q::NPARAM + 63) / 64), dim3(256), 0, st,
                       (const float*)(static_cast<char*>(qx.ws) + wl.partials), wl.nrec, Q::NPARAM, loss, qx.mgrad);
    MARL_CHECK_LAUNCH("qmix_reduce_kernel");
    return 0;
}

// (agents, obs dim) pairs with a compiled mixer = the LBF and warehouse shapes of common.h
#define MARL_QMIX_SHAPES(X) \
    X(2, 12) X(2, 15) X(3, 18) X(3, 24) X(4, 21) X(4, 27) X(8, 39) /* env.observe_id: */ X(2, 14) X(2, 17) X(3, 21) X(3, 27) X(4, 25) X(4, 31) X(8, 47) /* rware: */ X(2, 71) X(4, 71)

}  // namespace marl
