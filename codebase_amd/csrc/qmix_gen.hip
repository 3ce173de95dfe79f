// The generic QMIX mixer stage (qmix_gen.h): QMixer.forward / backward (marlbase/dqn/model.py:272-331) and its part of
// QMixNetwork._compute_loss (model.py:374-427) for `hypernet_layers` 1 or 2 and any widths, on the GEMM path of wide_mlp.h.
//
//   S        [(T+1) B][SD']   state rows = all agents' observations concatenated (model.py:389,412), materialised once from the Batch or
//                             gathered from the replay; the online mixer reads rows [0, T B), the target mixer rows [B, (T+1) B)
//   L == 2:  H1 = relu(S A1^T + a1), HF = relu(S Af^T + af), W1 = H1 B1^T + c1, WF = HF Bf^T + cf          (model.py:287-296)
//   L == 1:  W1 = S W1^T + b,  WF = S Wf^T + b                                                              (model.py:283-285)
//   both  :  B1 = S Bb^T + cb,  HV = relu(S Av^T + av)                                                      (model.py:303-311)
//   row   :  z_e = sum_p q_p |W1[p][e]| + B1_e, hid = elu(z), y = sum_e hid_e |WF_e| + bv . HV + cv         (model.py:313-331)
//   online:  delta = y - (r_0 + gamma y' (1 - done)), dL/dy = 2 filled delta (normalised by sum(filled) in the folds), gradients
//            w.r.t. every hypernet output row by row, then dX = (dY W) * relu' and dW = dY^T [X | 1] as GEMMs.
// fp32 throughout; every dot product has a fixed summation order (GEMM k order; split-K partials folded in split order).
#include "wide_mlp.h"
#include "dqn_update_kernels.h"  // QmixCtx, QmixIo, ReplaySrc, launch_colstd (templates only: no learner kernel is instantiated here)
#include "qmix_gen.h"

namespace marl {

namespace {

struct GenLayout {  // canonical parameter offsets (floats)
    int64_t w1a, w1a_b, w1b, w1b_b, wfa, wfa_b, wfb, wfb_b, bb, bb_b, av, av_b, bv, cv, n;
};

GenLayout gen_layout(const QmixGenDims& d) {
    const int64_t SD = (int64_t)d.P * d.D, E = d.E, EP = (int64_t)d.E * d.P, HE = d.HE;
    GenLayout g = {};
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += n; return at; };
    if (d.L == 2) {
        g.w1a = take(HE * SD); g.w1a_b = take(HE);
        g.w1b = take(EP * HE); g.w1b_b = take(EP);
        g.wfa = take(HE * SD); g.wfa_b = take(HE);
        g.wfb = take(E * HE); g.wfb_b = take(E);
    } else {  // one Linear per hypernet: the "b" (output) layer reads the state itself
        g.w1a = g.w1a_b = g.wfa = g.wfa_b = -1;
        g.w1b = take(EP * SD); g.w1b_b = take(EP);
        g.wfb = take(E * SD); g.wfb_b = take(E);
    }
    g.bb = take(E * SD); g.bb_b = take(E);
    g.av = take(E * SD); g.av_b = take(E);
    g.bv = take(E); g.cv = take(1);
    g.n = o;
    return g;
}

struct GenWs {  // byte offsets
    int64_t S, H1, HF, W1, WF, B1, HV, GW1, GWF, GB1, GHV, GH1, GHF, DY, ytgt, idx, part, total;
    int SDp, splits, chunk;
};

GenWs gen_ws(const QmixGenDims& d, int T, int B) {
    const int64_t R = (int64_t)T * B, RA = (int64_t)(T + 1) * B, SD = (int64_t)d.P * d.D, E = d.E, EP = (int64_t)d.E * d.P, HE = d.HE;
    GenWs w = {};
    w.SDp = (int)((SD + 3) & ~(int64_t)3);
    int64_t o = 0;
    auto take = [&](int64_t floats) { const int64_t at = o; o = (o + floats * 4 + 255) & ~(int64_t)255; return at; };
    w.S = take(RA * w.SDp);
    w.H1 = take(d.L == 2 ? R * HE : 0); w.HF = take(d.L == 2 ? R * HE : 0);
    w.W1 = take(R * EP); w.WF = take(R * E); w.B1 = take(R * E); w.HV = take(R * E);
    w.GW1 = take(R * EP); w.GWF = take(R * E); w.GB1 = take(R * E); w.GHV = take(R * E);
    w.GH1 = take(d.L == 2 ? R * HE : 0); w.GHF = take(d.L == 2 ? R * HE : 0);
    w.DY = take(R); w.ytgt = take(R); w.idx = take(B);
    w.splits = wide_splits((int)R);
    w.chunk = (int)(((R + w.splits - 1) / w.splits + 15) & ~(int64_t)15);
    const int64_t in_max = (d.L == 2 ? (SD > HE ? SD : HE) : SD) + 1, out_max = d.L == 2 ? (EP > HE ? EP : HE) : EP;
    w.part = take((int64_t)w.splits * out_max * (in_max > E + 1 ? in_max : E + 1));
    w.total = o;
    return w;
}

// S[(t, b)][p D + d] = obs_p(t, b)[d]; columns [SD, SD') are zero.  Batch: obss[P][T+1][B][D]; replay: episode idx[b] of obs[cap][P][T+1][D]
__global__ __launch_bounds__(256) void qg_gather_kernel(const float* __restrict__ obss, const float* __restrict__ rb_obs, const int32_t* __restrict__ idx,
                                                         int P, int D, int T, int B, int SDp, float* __restrict__ S) {
    const int64_t n = (int64_t)(T + 1) * B * SDp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / SDp;
        const int k = (int)(i - row * SDp), t = (int)(row / B), b = (int)(row - (int64_t)t * B);
        float v = 0.f;
        if (k < P * D) {
            const int p = k / D, dd = k - p * D;
            v = rb_obs != nullptr ? rb_obs[(((int64_t)idx[b] * P + p) * (T + 1) + t) * D + dd] : obss[(((int64_t)p * (T + 1) + t) * B + b) * D + dd];
        }
        S[i] = v;
    }
}

__global__ __launch_bounds__(256) void qg_draw_kernel(ReplaySrc rs, int B, int32_t* __restrict__ out) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b < B) out[b] = rs.idx ? rs.idx[b] : replay_draw(rs, b);
}

struct QgMix {
    const float* q;  // [P][R]: chosen (online) / bootstrap values (target)
    const float *W1, *WF, *B1, *HV, *bv, *cv;
    const float *r0, *dn, *fl;
    float* ytgt;
    int ytgt_is_return;
    float *GW1, *GWF, *GB1, *GHV, *DY, *dq, *lrow;
    int P, E, R;
    float gamma;
};

__device__ __forceinline__ float qg_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

__device__ __forceinline__ float qg_sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }  // d|x|/dx as torch takes it (0 at 0)

// one wave per row; lane e, e + 64, ... of the embedding
template <bool ONLINE>
__global__ __launch_bounds__(256) void qg_mix_kernel(QgMix a) {
    constexpr int MAXP = 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int P = a.P, E = a.E, EP = a.E * a.P;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < a.R; row += (int64_t)gridDim.x * 4) {
        float q[MAXP];
#pragma unroll
        for (int p = 0; p < MAXP; ++p) q[p] = p < P ? a.q[(int64_t)p * a.R + row] : 0.f;
        const float* w1 = a.W1 + row * EP;
        const float* wf = a.WF + row * E;
        const float* b1 = a.B1 + row * E;
        const float* hv = a.HV + row * E;
        float y = 0.f;
        for (int e = lane; e < E; e += 64) {
            float z = b1[e];
#pragma unroll
            for (int p = 0; p < MAXP; ++p)
                if (p < P) z += q[p] * fabsf(w1[p * E + e]);
            const float hid = z > 0.f ? z : expm1f(z);
            y += hid * fabsf(wf[e]) + a.bv[e] * hv[e];
        }
        y = qg_wave_sum(y) + a.cv[0];
        if (!ONLINE) {
            if (lane == 0) a.ytgt[row] = y;
            continue;
        }
        const float ret = a.ytgt_is_return ? a.ytgt[row] : a.r0[row] + a.gamma * a.ytgt[row] * (1.f - a.dn[row]);
        const float f = a.fl[row], delta = y - ret, dy = 2.f * f * delta;
        if (lane == 0) {
            a.lrow[row] = f * delta * delta;
            a.DY[row] = dy;
        }
        float dq[MAXP];
#pragma unroll
        for (int p = 0; p < MAXP; ++p) dq[p] = 0.f;
        for (int e = lane; e < E; e += 64) {
            float z = b1[e];
#pragma unroll
            for (int p = 0; p < MAXP; ++p)
                if (p < P) z += q[p] * fabsf(w1[p * E + e]);
            const float hid = z > 0.f ? z : expm1f(z);
            const float wfe = wf[e];
            a.GWF[row * E + e] = dy * hid * qg_sign(wfe);
            const float dz = dy * fabsf(wfe) * (z > 0.f ? 1.f : hid + 1.f);  // elu'(z) = exp(z) = elu(z) + 1 for z <= 0
            a.GB1[row * E + e] = dz;
            a.GHV[row * E + e] = hv[e] > 0.f ? dy * a.bv[e] : 0.f;
#pragma unroll
            for (int p = 0; p < MAXP; ++p)
                if (p < P) {
                    const float w = w1[p * E + e];
                    a.GW1[row * EP + p * E + e] = dz * q[p] * qg_sign(w);
                    dq[p] += dz * fabsf(w);
                }
        }
#pragma unroll
        for (int p = 0; p < MAXP; ++p)
            if (p < P) {
                const float s = qg_wave_sum(dq[p]);
                if (lane == 0) a.dq[(int64_t)p * a.R + row] = s;
            }
    }
}

// Y[rows][n_out] = act(X W^T + bias): X row r at x + r * xs, W [n_out][n_in] row-major
void gen_linear(const float* x, int64_t xs, int rows, const float* W, const float* bias, int n_in, int n_out, float* y, bool relu, hipStream_t st) {
    GemmOp g = {};
    g.M = rows; g.N = n_out; g.K = n_in; g.k_chunk = 1 << 30; g.b_ones = -1; g.epi = relu ? 2 : 1;
    g.A = x; g.a_m = xs; g.a_k = 1; g.B = W; g.b_k = 1; g.b_n = n_in; g.C = y; g.c_m = n_out; g.bias = bias;
    wide_gemm<true, true>(g, 1, st);
}

// all hypernet / bias-net outputs of one mixer instance for the R rows starting at S (row stride SDp)
void gen_forward(const QmixGenDims& d, const GenLayout& L, const float* w, const float* S, int SDp, int R, char* base, const GenWs& ws, hipStream_t st) {
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    const int SD = d.P * d.D, E = d.E, EP = d.E * d.P, HE = d.HE;
    if (d.L == 2) {
        gen_linear(S, SDp, R, w + L.w1a, w + L.w1a_b, SD, HE, f(ws.H1), true, st);
        gen_linear(S, SDp, R, w + L.wfa, w + L.wfa_b, SD, HE, f(ws.HF), true, st);
        gen_linear(f(ws.H1), HE, R, w + L.w1b, w + L.w1b_b, HE, EP, f(ws.W1), false, st);
        gen_linear(f(ws.HF), HE, R, w + L.wfb, w + L.wfb_b, HE, E, f(ws.WF), false, st);
    } else {
        gen_linear(S, SDp, R, w + L.w1b, w + L.w1b_b, SD, EP, f(ws.W1), false, st);
        gen_linear(S, SDp, R, w + L.wfb, w + L.wfb_b, SD, E, f(ws.WF), false, st);
    }
    gen_linear(S, SDp, R, w + L.bb, w + L.bb_b, SD, E, f(ws.B1), false, st);
    gen_linear(S, SDp, R, w + L.av, w + L.av_b, SD, E, f(ws.HV), true, st);
}

}  // namespace

int qmix_gen_check(const QmixGenDims& d) {
    MARL_REQUIRE(d.L == 1 || d.L == 2, "QMixer: hypernet_layers must be 1 or 2 (marlbase/dqn/model.py:283-301), got %d", d.L);
    MARL_REQUIRE(d.P >= 1 && d.P <= 16 && d.D >= 1 && d.E >= 1 && d.E <= 1024 && (d.L == 1 || (d.HE >= 1 && d.HE <= 1024)),
                 "QMixer: agents %d / obs %d / embed_dim %d / hypernet_embed %d out of range (agents <= 16, widths <= 1024)", d.P, d.D, d.E, d.HE);
    return 0;
}

int64_t qmix_gen_nparams(const QmixGenDims& d) { return gen_layout(d).n; }

int64_t qmix_gen_ws_bytes(const QmixGenDims& d, int T, int B) { return gen_ws(d, T, B).total; }

int qmix_gen_mix(const QmixCtx& qx, const QmixGenDims& d, const marlhip_batch* bt, const ReplaySrc* rs, const QmixIo& io, float gamma, hipStream_t st) {
    if (qmix_gen_check(d) != 0) return -1;
    MARL_REQUIRE(!qx.l1_fp16, "QMixer: the opt-in fp16 first layers exist for mixing = {64, 2, 32} on the compiled shapes only");
    const int T = bt->max_len, B = bt->batch, R = T * B;
    const GenWs ws = gen_ws(d, T, B);
    MARL_REQUIRE(qx.ws_bytes >= ws.total, "qmix_loss_grad: mixer workspace %lld < %lld bytes", (long long)qx.ws_bytes, (long long)ws.total);
    const GenLayout L = gen_layout(d);
    char* base = static_cast<char*>(qx.ws);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    timing_begin(TIMER_QMIX, st);
    const int32_t* idx = nullptr;
    if (rs != nullptr) {
        int32_t* ib = reinterpret_cast<int32_t*>(base + ws.idx);
        hipLaunchKernelGGL(qg_draw_kernel, dim3((B + 255) / 256), dim3(256), 0, st, *rs, B, ib);
        idx = ib;
    }
    {
        const int64_t n = (int64_t)(T + 1) * B * ws.SDp;
        const int64_t want = (n + 255) / 256;
        hipLaunchKernelGGL(qg_gather_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, st, bt->obss, rs != nullptr ? rs->rb.obs : nullptr, idx,
                           d.P, d.D, T, B, ws.SDp, f(ws.S));
    }
    QgMix a = {};
    a.W1 = f(ws.W1); a.WF = f(ws.WF); a.B1 = f(ws.B1); a.HV = f(ws.HV);
    a.r0 = io.r0; a.dn = io.dn; a.fl = io.fl; a.ytgt = f(ws.ytgt); a.ytgt_is_return = 0;
    a.GW1 = f(ws.GW1); a.GWF = f(ws.GWF); a.GB1 = f(ws.GB1); a.GHV = f(ws.GHV); a.DY = f(ws.DY); a.dq = io.dq; a.lrow = io.lrow;
    a.P = d.P; a.E = d.E; a.R = R; a.gamma = gamma;
    const int grid = (R + 3) / 4 < 4096 ? (R + 3) / 4 : 4096;
    // target mixer on obs[1:] (model.py:404-413)
    gen_forward(d, L, qx.tmixer, f(ws.S) + (int64_t)B * ws.SDp, ws.SDp, R, base, ws, st);
    a.q = io.tqsel; a.bv = qx.tmixer + L.bv; a.cv = qx.tmixer + L.cv;
    hipLaunchKernelGGL((qg_mix_kernel<false>), dim3(grid), dim3(256), 0, st, a);
    if (qx.rst != nullptr) {  // standardise_returns: the target mixer's output becomes the standardised return (model.py:415-422)
        if (launch_colstd(T, B, gamma, *qx.rst, a.ytgt, 1, 0, io.r0, io.dn, a.ytgt, st) != 0) return -1;
        a.ytgt_is_return = 1;
    }
    // online mixer on obs[:-1], TD error, backward
    gen_forward(d, L, qx.mixer, f(ws.S), ws.SDp, R, base, ws, st);
    a.q = io.chosen; a.bv = qx.mixer + L.bv; a.cv = qx.mixer + L.cv;
    hipLaunchKernelGGL((qg_mix_kernel<true>), dim3(grid), dim3(256), 0, st, a);
    if (d.L == 2) {  // dH = (dW B) * [H > 0] for both hypernets
        auto back = [&](const float* G, int n_out, const float* Wb, const float* H, float* GH) {
            GemmOp g = {};
            g.M = R; g.N = d.HE; g.K = n_out; g.k_chunk = 1 << 30; g.b_ones = -1; g.epi = 3;
            g.A = G; g.a_m = n_out; g.a_k = 1; g.B = Wb; g.b_k = d.HE; g.b_n = 1; g.C = GH; g.c_m = d.HE; g.gate = H; g.gate_m = d.HE;
            wide_gemm<true, false>(g, 1, st);
        };
        back(f(ws.GW1), d.E * d.P, qx.mixer + L.w1b, f(ws.H1), f(ws.GH1));
        back(f(ws.GWF), d.E, qx.mixer + L.wfb, f(ws.HF), f(ws.GHF));
    }
    timing_end(TIMER_QMIX, st);
    MARL_CHECK_LAUNCH("generic qmix mixer stage");
    return 0;
}

int qmix_gen_reduce(const QmixCtx& qx, const QmixGenDims& d, int T, int B, const float* loss, hipStream_t st) {
    if (qmix_gen_check(d) != 0) return -1;
    const int R = T * B;
    const GenWs ws = gen_ws(d, T, B);
    const GenLayout L = gen_layout(d);
    char* base = static_cast<char*>(qx.ws);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    float* part = f(ws.part);
    float* mg = qx.mgrad;
    const int SD = d.P * d.D, E = d.E, EP = d.E * d.P, HE = d.HE;
    // dW[out][in] (+ bias column) = dY^T [X | 1] over row slices, then the fold in slice order with 1 / sum(filled) (loss[1])
    auto wgrad = [&](const float* dy, int n_out, const float* x, int64_t xs, int n_in, float* dW, float* db) {
        GemmOp g = {};
        g.M = n_out; g.N = n_in + 1; g.K = R; g.k_chunk = ws.chunk; g.b_ones = n_in; g.epi = 0;
        g.A = dy; g.a_m = 1; g.a_k = n_out; g.B = x; g.b_k = xs; g.b_n = 1;
        g.C = part; g.c_m = n_in + 1; g.c_split = (int64_t)n_out * (n_in + 1);
        wide_gemm<false, false>(g, ws.splits, st);
        const int n = n_out * (n_in + 1);
        hipLaunchKernelGGL(wide_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)part, ws.splits, g.c_split, n_out, n_in + 1, loss + 1,
                           dW, db);
    };
    const float* S = f(ws.S);
    if (d.L == 2) {
        wgrad(f(ws.GH1), HE, S, ws.SDp, SD, mg + L.w1a, mg + L.w1a_b);
        wgrad(f(ws.GW1), EP, f(ws.H1), HE, HE, mg + L.w1b, mg + L.w1b_b);
        wgrad(f(ws.GHF), HE, S, ws.SDp, SD, mg + L.wfa, mg + L.wfa_b);
        wgrad(f(ws.GWF), E, f(ws.HF), HE, HE, mg + L.wfb, mg + L.wfb_b);
    } else {
        wgrad(f(ws.GW1), EP, S, ws.SDp, SD, mg + L.w1b, mg + L.w1b_b);
        wgrad(f(ws.GWF), E, S, ws.SDp, SD, mg + L.wfb, mg + L.wfb_b);
    }
    wgrad(f(ws.GB1), E, S, ws.SDp, SD, mg + L.bb, mg + L.bb_b);
    wgrad(f(ws.GHV), E, S, ws.SDp, SD, mg + L.av, mg + L.av_b);
    wgrad(f(ws.DY), 1, f(ws.HV), E, E, mg + L.bv, mg + L.cv);
    MARL_CHECK_LAUNCH("generic qmix mixer gradients");
    return 0;
}

}  // namespace marl
