// Centralised critics whose input is too wide for the register-resident kernels (critic.centralised, marlbase/ac/model.py:62-66, 155-157:
// every critic reads the concatenation of all agents' observations - 312 floats for 8 LBF agents, 284 for the 4-agent warehouse), hidden
// width 64 or 128, ONE output.  Round 4 takes them off the layer-by-layer GEMM path of wide_mlp.h:
//
//   wc_fwd_kernel   a 64-row tile per workgroup, all three layers in one launch: layer 1 K-STREAMED (16-deep slices of the observation rows
//                   and of W1 through LDS, as wide_gemm128_kernel does), its relu output left in LDS as the A operand of layer 2 (W2 slices
//                   streamed the same way), layer 3 a dot product per row.  The target critic writes nothing but the values; the online
//                   critic also leaves both hidden layers ([P][rows][H], written with full-row stores from the LDS tile) for the backward pass.
//   wc_bwd_kernel   persistent workgroups over 64-row tiles: dY2 = dv W3 * (Y2 > 0) elementwise, dY1 = (dY2 W2) * (Y1 > 0) with dY2 as the
//                   LDS-resident A operand, and dW2 += dY2^T Y1 right there (both tiles are in LDS, k-major for that product; the wave's
//                   64 x 64 quadrant of dW2 stays in registers across the tiles).  Writes dY1 for the first layer's weight-gradient
//                   product and column sums (dW3, db3, db2, db1) - no ones column, no one-row GEMM for the output layer, no recompute.
//   wc_wgrad_kernel dW1 = dY1^T X over row ranges: both operands are contiguous ACROSS the reduction index, so the LDS tiles are k-major
//                   ([16][tile + 16]): 16-byte stores straight from the 16-byte loads, conflict-free 4-byte operand reads (the generic
//                   kernel transposes with 8-way conflicting scalar stores); every agent in one launch (grid.y).
// Sums over rows are folded in a fixed order (per-range partials, then ranges in order): bitwise reproducible, no atomics.
#pragma once
#include "wide_mlp.h"

namespace marl {

// 16-byte-aligned, zero-padded image of one network: W1[H][DP] (DP = D up to a multiple of 16: no k bounds in the loop) | b1 | W2[H][H] |
// W2^T[H][H] (B operand of the backward data product, k-contiguous like the others) | b2 | W3[H] | b3
struct WcPack {
    int D, DP, H;
    __host__ __device__ WcPack(int d, int h) : D(d), DP((d + 15) & ~15), H(h) {}
    __host__ __device__ int oB1() const { return H * DP; }
    __host__ __device__ int oW2() const { return oB1() + H; }
    __host__ __device__ int oW2T() const { return oW2() + H * H; }
    __host__ __device__ int oB2() const { return oW2T() + H * H; }
    __host__ __device__ int oW3() const { return oB2() + H; }
    __host__ __device__ int oB3() const { return oW3() + H; }
    __host__ __device__ int total() const { return oB3() + 4; }
};

static __global__ __launch_bounds__(256) void wc_pack_kernel(const float* __restrict__ params, int64_t nparam, WcPack pk, float* __restrict__ packs) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= pk.total()) return;
    const float* w = params + (int64_t)blockIdx.y * nparam;
    const int D = pk.D, H = pk.H;
    const int cb1 = H * D, cW2 = cb1 + H, cb2 = cW2 + H * H, cW3 = cb2 + H, cb3 = cW3 + H;  // FCNetwork's parameters() order
    float v = 0.f;
    if (idx < pk.oB1()) {
        const int n = idx / pk.DP, k = idx - n * pk.DP;
        v = k < D ? w[n * D + k] : 0.f;
    } else if (idx < pk.oW2()) {
        v = w[cb1 + idx - pk.oB1()];
    } else if (idx < pk.oW2T()) {
        v = w[cW2 + idx - pk.oW2()];
    } else if (idx < pk.oB2()) {
        const int j = idx - pk.oW2T(), k = j / H, n = j - k * H;
        v = w[cW2 + n * H + k];
    } else if (idx < pk.oW3()) {
        v = w[cb2 + idx - pk.oB2()];
    } else if (idx < pk.oB3()) {
        v = w[cW3 + idx - pk.oW3()];
    } else if (idx == pk.oB3()) {
        v = w[cb3];
    }
    packs[(int64_t)blockIdx.y * pk.total() + idx] = v;
}

constexpr int WC_ROWS = 64;  // rows of a tile

struct WcFwdArgs {
    const float* packs;  // [nblk][pk.total()]
    AgentMap am;
    const float* x; int64_t x_as, x_rs;  // observation rows: agent stride (0: every critic reads the same rows), row stride
    int rows, D;
    bool x_vec;          // 16-byte loads of the rows are legal
    float* out;          // [P][rows]
    float* y1; float* y2;  // [P][rows][H] (STORE)
};

// thread map of the row-wise passes: thread -> (row = tid / 4, part = tid % 4), part covers columns [part H / 4, (part + 1) H / 4)
template <int H, bool STORE>
__global__ __launch_bounds__(256) void wc_fwd_kernel(const WcFwdArgs g) {
    constexpr int LD = 20, LH = H + 4, NU = H / 32, NH = H / 64, C4 = H / 16;
    __shared__ __attribute__((aligned(16))) float As[WC_ROWS * LD], Bs[H * LD], Hs[WC_ROWS * LH];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int p = blockIdx.y, m0 = blockIdx.x * WC_ROWS;
    const WcPack pk(g.D, H);
    const float* w = g.packs + (int64_t)g.am.net[p] * pk.total();
    const float* x = g.x + (int64_t)p * g.x_as;
    const int row = tid >> 2, part = tid & 3;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 acc[2][NU];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = zero4;
    // ---- layer 1: rows x W1^T, k in slices of 16
    f4 ra, rb[NH];
    auto load1 = [&](int k0) {
        ra = gemm_load4(x, m0 + row, g.x_rs, k0 + 4 * part, g.D, m0 + row < g.rows, -1, g.x_vec);
#pragma unroll
        for (int h = 0; h < NH; ++h) rb[h] = *reinterpret_cast<const f4*>(w + (64 * h + row) * pk.DP + k0 + 4 * part);
    };
    auto mfma_slice = [&](const float* Arow, int lda, int acol) {
        f4 a[2], b[NU];
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f4*>(Arow + (32 * wm + 16 * t + i) * lda + acol + 4 * q);
#pragma unroll
        for (int u = 0; u < NU; ++u) b[u] = *reinterpret_cast<const f4*>(Bs + ((H / 2) * wn + 16 * u + i) * LD + 4 * q);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < NU; ++u) acc[t][u] = MARL_MFMA(a[t][s], b[u][s], acc[t][u]);
    };
    load1(0);
    for (int k0 = 0; k0 < pk.DP; k0 += 16) {
        __syncthreads();  // the previous slice has been multiplied
        *reinterpret_cast<f4*>(As + row * LD + 4 * part) = ra;
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<f4*>(Bs + (64 * h + row) * LD + 4 * part) = rb[h];
        __syncthreads();
        if (k0 + 16 < pk.DP) load1(k0 + 16);
        mfma_slice(As, LD, 0);
    }
    // ---- h1 = relu(. + b1) into the LDS tile [row][H + 4]: layer 2's A operand (and the source of the full-row stores)
    f4 rw[NH];
    auto load2 = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) rw[h] = *reinterpret_cast<const f4*>(w + pk.oW2() + (64 * h + row) * H + k0 + 4 * part);
    };
    auto dump = [&](const float* bias) {  // acc (C layout: row 4q + r, column i of each tile) + bias, relu -> Hs
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int n = (H / 2) * wn + 16 * u + i;
            const float b = bias[n];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) Hs[(32 * wm + 16 * t + 4 * q + r) * LH + n] = fmaxf(acc[t][u][r] + b, 0.f);
        }
    };
    auto copy_out = [&](float* y) {  // the tile's rows as whole 16-byte pieces
        if (m0 + row < g.rows) {
            float* dst = y + ((int64_t)p * g.rows + m0 + row) * H + part * (H / 4);
#pragma unroll
            for (int c = 0; c < C4; ++c) *reinterpret_cast<f4*>(dst + 4 * c) = *reinterpret_cast<const f4*>(Hs + row * LH + part * (H / 4) + 4 * c);
        }
    };
    load2(0);
    dump(w + pk.oB1());
    // ---- layer 2: h1 (LDS) x W2^T
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) acc[t][u] = zero4;
    for (int k0 = 0; k0 < H; k0 += 16) {
        __syncthreads();  // slice k0 - 16 has been multiplied (first pass: layer 1's last slice; h1 is complete)
#pragma unroll
        for (int h = 0; h < NH; ++h) *reinterpret_cast<f4*>(Bs + (64 * h + row) * LD + 4 * part) = rw[h];
        __syncthreads();
        if (k0 + 16 < H) load2(k0 + 16);
        if (STORE && k0 == 0) copy_out(g.y1);
        mfma_slice(Hs, LH, k0);
    }
    __syncthreads();  // every wave has read h1
    dump(w + pk.oB2());
    __syncthreads();
    if (STORE) copy_out(g.y2);
    // ---- layer 3: one output; the row's four parts are neighbouring lanes
    {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            const f4 hv = *reinterpret_cast<const f4*>(Hs + row * LH + part * (H / 4) + 4 * c);
            const f4 wv = *reinterpret_cast<const f4*>(w + pk.oW3() + part * (H / 4) + 4 * c);
            s += hv[0] * wv[0] + hv[1] * wv[1] + hv[2] * wv[2] + hv[3] * wv[3];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        if (part == 0 && m0 + row < g.rows) g.out[(int64_t)p * g.rows + m0 + row] = s + w[pk.oB3()];
    }
}

// column sums of the backward pass, one record per (workgroup, row group): [dW3 (H) | db3 | db2 (H) | db1 (H)]
__host__ __device__ inline int wc_partn(int H) { return 3 * H + 1; }

struct WcBwdArgs {
    const float* packs;
    AgentMap am;
    int rows, D, tiles;
    const float* dout;   // [P][rows]: dL/dvalue, already masked by filled
    const float* y1; const float* y2;  // [P][rows][H]
    float* d1;           // [P][rows][H]: dL/d(layer-1 pre-activation), the A operand of the first layer's weight-gradient product
    float* part;         // [P][gridDim.x * (256 / H)][wc_partn(H)]
    float* w2part;       // [P][gridDim.x][H][H]: the workgroup's share of dW2
};

template <int H>
constexpr int wc_bwd_lds_floats() { return H * 20 + 2 * WC_ROWS * (H + 4) + WC_ROWS; }

// Persistent: a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... of agent blockIdx.y and keeps dW2 (its wave's 64 x 64
// quadrant of the H x H matrix; both operands are already in LDS, k-major) and the column sums in registers across them.
// (Measured and dropped, scripts/gpu_runs/r4AB.sh: W2^T RESIDENT in LDS - one workgroup per unit, 7 barriers per tile instead of 21, the next
// tile's rows requested into registers under the current tile's products - 774 us against this form's 760 us on the 15x15-8p batch.)
template <int H>
__global__ __launch_bounds__(256, 2) void wc_bwd_kernel(const WcBwdArgs g) {  // two workgroups per compute unit: <= 256 registers
    constexpr int LD = 20, LH = H + 4, NU = H / 32, NH = H / 64, C4 = H / 16, NG = 256 / H, RPG = WC_ROWS / NG;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Bs = lds;                    // [H][LD]   W2^T slice
    float* R1 = Bs + H * LD;            // [64][LH]  Y2 tile, then Y1 (B operand of dW2)
    float* R2 = R1 + WC_ROWS * LH;      // [64][LH]  dY2 tile (A operand of both products), then dY1
    float* dvs = R2 + WC_ROWS * LH;     // [64]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int p = blockIdx.y;
    const WcPack pk(g.D, H);
    const float* w = g.packs + (int64_t)g.am.net[p] * pk.total();
    const int row = tid >> 2, part = tid & 3, c0 = part * (H / 4);
    const int cn = tid % H, cg = tid / H;
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 dW2[NU][NU];  // rows: layer-2 units (H / 2) wm + 16 t + 4 q + r; columns: layer-1 units (H / 2) wn + 16 u + i
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u) dW2[t][u] = zero4;
    float s3 = 0.f, s2 = 0.f, s1 = 0.f, sb = 0.f;
    f4 rw[NH];
    auto load2 = [&](int k0) {
#pragma unroll
        for (int h = 0; h < NH; ++h) rw[h] = *reinterpret_cast<const f4*>(w + pk.oW2T() + (64 * h + row) * H + k0 + 4 * part);
    };
    for (int tile = blockIdx.x; tile < g.tiles; tile += gridDim.x) {
        const int m0 = tile * WC_ROWS;
        const bool ok = m0 + row < g.rows;
        const int64_t grow = ((int64_t)p * g.rows + (ok ? m0 + row : g.rows - 1)) * H + c0;
        // ---- a. dY2 = dv W3 * (Y2 > 0), row-wise; Y2 and dY2 to LDS
        f4 y1v[C4];
        {
            const float dv = ok ? g.dout[(int64_t)p * g.rows + m0 + row] : 0.f;
            if (part == 0) dvs[row] = dv;
            f4 y2v[C4];
#pragma unroll
            for (int c = 0; c < C4; ++c) y2v[c] = *reinterpret_cast<const f4*>(g.y2 + grow + 4 * c);
#pragma unroll
            for (int c = 0; c < C4; ++c) y1v[c] = *reinterpret_cast<const f4*>(g.y1 + grow + 4 * c);
            load2(0);
#pragma unroll
            for (int c = 0; c < C4; ++c) {
                const f4 w3 = *reinterpret_cast<const f4*>(w + pk.oW3() + c0 + 4 * c);
                f4 y = ok ? y2v[c] : zero4, d;
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = y[e] > 0.f ? dv * w3[e] : 0.f;
                *reinterpret_cast<f4*>(R1 + row * LH + c0 + 4 * c) = y;
                *reinterpret_cast<f4*>(R2 + row * LH + c0 + 4 * c) = d;
                if (!ok) y1v[c] = zero4;
            }
        }
        __syncthreads();
        // ---- b. column sums over this group's rows, in row order: dW3 = sum dv Y2, db2 = sum dY2, db3 = sum dv
        for (int m = cg * RPG; m < (cg + 1) * RPG; ++m) {
            const float dv = dvs[m];
            s3 += dv * R1[m * LH + cn];
            s2 += R2[m * LH + cn];
            sb += dv;
        }
        // ---- c. dY1 (before the gate) = dY2 (LDS) x W2: B operand W2^T[k_out][n], slices of 16 n
        f4 acc[2][NU];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[t][u] = zero4;
        for (int k0 = 0; k0 < H; k0 += 16) {
            if (k0 > 0) __syncthreads();  // slice k0 - 16 has been multiplied (and, at k0 = 16, every wave is past step b: Y2 is no longer needed)
#pragma unroll
            for (int h = 0; h < NH; ++h) *reinterpret_cast<f4*>(Bs + (64 * h + row) * LD + 4 * part) = rw[h];
            if (k0 == 16) {  // Y1 takes Y2's place: the B operand of dW2 below (rows past the batch are zeros)
#pragma unroll
                for (int c = 0; c < C4; ++c) *reinterpret_cast<f4*>(R1 + row * LH + c0 + 4 * c) = y1v[c];
            }
            __syncthreads();
            if (k0 + 16 < H) load2(k0 + 16);
            f4 a[2], b[NU];
#pragma unroll
            for (int t = 0; t < 2; ++t) a[t] = *reinterpret_cast<const f4*>(R2 + (32 * wm + 16 * t + i) * LH + k0 + 4 * q);
#pragma unroll
            for (int u = 0; u < NU; ++u) b[u] = *reinterpret_cast<const f4*>(Bs + ((H / 2) * wn + 16 * u + i) * LD + 4 * q);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < NU; ++u) acc[t][u] = MARL_MFMA(a[t][s], b[u][s], acc[t][u]);
        }
        // ---- dW2 += dY2^T Y1 over the tile's 64 rows: both tiles are k-major for this product (row = reduction index)
#pragma unroll 2
        for (int s = 0; s < WC_ROWS / 4; ++s) {
            float a2[NU], b2[NU];
#pragma unroll
            for (int t = 0; t < NU; ++t) a2[t] = R2[(4 * s + q) * LH + (H / 2) * wm + 16 * t + i];
#pragma unroll
            for (int u = 0; u < NU; ++u) b2[u] = R1[(4 * s + q) * LH + (H / 2) * wn + 16 * u + i];
#pragma unroll
            for (int t = 0; t < NU; ++t)
#pragma unroll
                for (int u = 0; u < NU; ++u) dW2[t][u] = MARL_MFMA(a2[t], b2[u], dW2[t][u]);
        }
        __syncthreads();  // every wave has read dY2
        // ---- d. through the LDS tile back to rows: gate with Y1 (its LDS tile), store, column sums for db1
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) R2[(32 * wm + 16 * t + 4 * q + r) * LH + (H / 2) * wn + 16 * u + i] = acc[t][u][r];
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C4; ++c) {
            f4 d = *reinterpret_cast<const f4*>(R2 + row * LH + c0 + 4 * c);
            const f4 y1g = *reinterpret_cast<const f4*>(R1 + row * LH + c0 + 4 * c);  // Y1 is still in its tile (rows past the batch: zeros)
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = y1g[e] > 0.f ? d[e] : 0.f;
            *reinterpret_cast<f4*>(R2 + row * LH + c0 + 4 * c) = d;
            if (ok) *reinterpret_cast<f4*>(g.d1 + grow + 4 * c) = d;
        }
        __syncthreads();
        for (int m = cg * RPG; m < (cg + 1) * RPG; ++m) s1 += R2[m * LH + cn];
        __syncthreads();  // the next tile rewrites both LDS tiles
    }
    float* rec = g.part + ((int64_t)p * gridDim.x * NG + (int64_t)blockIdx.x * NG + cg) * wc_partn(H);
    rec[cn] = s3;
    rec[H + 1 + cn] = s2;
    rec[2 * H + 1 + cn] = s1;
    if (cn == 0) rec[H] = sb;
    float* C = g.w2part + ((int64_t)p * gridDim.x + blockIdx.x) * H * H;
#pragma unroll
    for (int t = 0; t < NU; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) C[((H / 2) * wm + 16 * t + 4 * q + r) * H + (H / 2) * wn + 16 * u + i] = dW2[t][u][r];
}

// column partials -> gradient entries: stage A adds groups of 64 records, stage B the groups, both in index order; / nf
constexpr int WC_FOLD_GROUP = 64;
static __global__ __launch_bounds__(256) void wc_fold_cols_a_kernel(const float* __restrict__ part, int nslots, int partn, float* __restrict__ g1) {
    const int j = blockIdx.x * 256 + threadIdx.x, grp = blockIdx.y, p = blockIdx.z, ngroups = gridDim.y;
    if (j >= partn) return;
    const int s0 = grp * WC_FOLD_GROUP, s1 = min(nslots, s0 + WC_FOLD_GROUP);
    float acc = 0.f;
    for (int s = s0; s < s1; ++s) acc += part[((int64_t)p * nslots + s) * partn + j];
    g1[((int64_t)p * ngroups + grp) * partn + j] = acc;
}
static __global__ __launch_bounds__(256) void wc_fold_cols_b_kernel(const float* __restrict__ g1, int ngroups, int H, const float* __restrict__ nf,
                                                                    float* __restrict__ gp, int64_t nparam, int ow3, int ob3, int ob2, int ob1) {
    const int partn = wc_partn(H), j = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    if (j >= partn) return;
    float acc = 0.f;
    for (int grp = 0; grp < ngroups; ++grp) acc += g1[((int64_t)p * ngroups + grp) * partn + j];
    acc /= nf[0];
    float* dst = gp + (int64_t)p * nparam;
    if (j < H) dst[ow3 + j] = acc;
    else if (j == H) dst[ob3] = acc;
    else if (j < 2 * H + 1) dst[ob2 + j - H - 1] = acc;
    else dst[ob1 + j - 2 * H - 1] = acc;
}

struct WcWgradArgs {
    const float* dy; int64_t dy_as;       // [P][rows][H]
    const float* yp; int64_t yp_as, yp_rs;  // previous layer's rows (hidden layer or observations)
    int rows, N, k_chunk;
    bool vec;
    float* part;  // [P][splits][H][N]
};

// C[m][n] = sum_k dy[k][m] yp[k][n] over rows k of split z; grid (N blocks of 128, P, splits)
#ifndef MARL_WC_WGRAD_KS
#define MARL_WC_WGRAD_KS 16  // rows per LDS slice of wc_wgrad_kernel (two barriers per slice; 32 measured: +-0, scripts/gpu_runs/r4AC.sh)
#endif
template <int H>
__global__ __launch_bounds__(256) void wc_wgrad_kernel(const WcWgradArgs g) {
    constexpr int LDA = H + 16, LDB = 128 + 16, TM = H / 32, NH = H / 64, KS = MARL_WC_WGRAD_KS, NS = KS / 16;
    __shared__ __attribute__((aligned(16))) float As[KS * LDA], Bs[KS * LDB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4, wm = wave >> 1, wn = wave & 1;
    const int p = blockIdx.y, z = blockIdx.z, n0 = blockIdx.x * 128;
    const int kbeg = z * g.k_chunk, kend = min(g.rows, kbeg + g.k_chunk);
    const float* dy = g.dy + (int64_t)p * g.dy_as;
    const float* yp = g.yp + (int64_t)p * g.yp_as;
    const int ka = tid >> 4, c = 4 * (tid & 15);
    const f4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f4 acc[TM][4];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[t][u] = zero4;
    f4 ra[NS][NH], rb[NS][2];
    auto load = [&](int k0) {
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            const int k = k0 + 16 * ss + ka;
            const bool ok = k < kend;
#pragma unroll
            for (int h = 0; h < NH; ++h) ra[ss][h] = ok ? *reinterpret_cast<const f4*>(dy + (int64_t)k * H + 64 * h + c) : zero4;
#pragma unroll
            for (int h = 0; h < 2; ++h) rb[ss][h] = gemm_load4(yp, k, g.yp_rs, n0 + 64 * h + c, g.N, ok, -1, g.vec);
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += KS) {
        __syncthreads();
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
#pragma unroll
            for (int h = 0; h < NH; ++h) *reinterpret_cast<f4*>(As + (16 * ss + ka) * LDA + 64 * h + c) = ra[ss][h];
#pragma unroll
            for (int h = 0; h < 2; ++h) *reinterpret_cast<f4*>(Bs + (16 * ss + ka) * LDB + 64 * h + c) = rb[ss][h];
        }
        __syncthreads();
        if (k0 + KS < kend) load(k0 + KS);
#pragma unroll
        for (int s = 0; s < KS / 4; ++s) {
            float a[TM], b[4];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = As[(4 * s + q) * LDA + (H / 2) * wm + 16 * t + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) b[u] = Bs[(4 * s + q) * LDB + 64 * wn + 16 * u + i];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[t][u] = MARL_MFMA(a[t], b[u], acc[t][u]);
        }
    }
    float* C = g.part + ((int64_t)p * gridDim.z + z) * H * g.N;
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = n0 + 64 * wn + 16 * u + i;
            if (n < g.N) {
#pragma unroll
                for (int r = 0; r < 4; ++r) C[(int64_t)((H / 2) * wm + 16 * t + 4 * q + r) * g.N + n] = acc[t][u][r];
            }
        }
}

// dW[p][idx] = (sum over the splits, in split order) / nf
static __global__ __launch_bounds__(256) void wc_fold_w_kernel(const float* __restrict__ part, int splits, int n, const float* __restrict__ nf,
                                                               float* __restrict__ gp, int64_t nparam, int ow) {
    const int idx = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y;
    if (idx >= n) return;
    float acc = 0.f;
    for (int z = 0; z < splits; ++z) acc += part[((int64_t)p * splits + z) * n + idx];
    gp[(int64_t)p * nparam + ow + idx] = acc / nf[0];
}

// row ranges of wc_wgrad_kernel: about 2048 workgroups in all (two full rounds of the ~1024 that are resident at once: a launch of 1.2
// rounds runs as long as one of 2), every range at least 256 rows, at most 256 ranges
inline int wc_wgrad_splits(int P, int rows, int D) {
    const int per = ((D + 127) / 128) * (P > 0 ? P : 1);
    int s = (2048 + per - 1) / per;
    if (s > rows / 256) s = rows / 256;
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}
inline int64_t wc_rec_floats(int P, int rows, int H) { return 2 * (int64_t)P * rows * H; }
inline int wc_tiles(int rows) { return (rows + WC_ROWS - 1) / WC_ROWS; }
inline int64_t wc_pack_bytes(int nblk, int D, int H) { return ((int64_t)nblk * WcPack(D, H).total() * 4 + 255) & ~(int64_t)255; }

struct WcWs {
    int64_t packs, d1, part, g1, w2part, wpart, gp, nf, total;
};
// workgroups of wc_bwd_kernel per agent: two per compute unit over all agents (its LDS lets two be resident)
inline int wc_bwd_wgs(int P, int rows) {
    const int g = 512 / (P > 0 ? P : 1), t = wc_tiles(rows);
    return g < 1 ? 1 : (g > t ? t : g);
}
inline WcWs wc_ws(int P, int rows, int D, int H) {
    WcWs w = {};
    int64_t o = 0;
    auto take = [&](int64_t nfloat) { const int64_t at = o; o = (o + nfloat * 4 + 255) & ~(int64_t)255; return at; };
    const int nwg = wc_bwd_wgs(P, rows), nslots = nwg * (256 / H), ngroups = (nslots + WC_FOLD_GROUP - 1) / WC_FOLD_GROUP;
    w.packs = take((int64_t)P * WcPack(D, H).total());
    w.d1 = take((int64_t)P * rows * H);
    w.part = take((int64_t)P * nslots * wc_partn(H));
    w.g1 = take((int64_t)P * ngroups * wc_partn(H));
    w.w2part = take((int64_t)P * nwg * H * H);
    w.wpart = take((int64_t)P * wc_wgrad_splits(P, rows, D) * H * D);
    w.gp = take((int64_t)P * WideNet{D, H, 1, 2}.nparam());
    w.nf = take(4 + 2 * WIDE_COUNT_MAX_WG);
    w.total = o;
    return w;
}

inline bool wc_rows_vec(const float* x, int64_t as, int64_t rs) {
    return (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (as & 3) == 0 && (rs & 3) == 0;
}

// out[p][row] = critic_p(x row); rec (may be null): both hidden layers of every row for wc_backward_rows ([2][P][rows][H])
template <int H>
int wc_forward_rows(int P, const AgentMap& am, const float* params, int D, const float* obs, int64_t as, int64_t rs, int rows, float* out, float* rec,
                    hipStream_t st) {
    const WcPack pk(D, H);
    float* packs = collect_pack_scratch((size_t)wc_pack_bytes(am.nblk, D, H), st);
    if (packs == nullptr) return -1;
    hipLaunchKernelGGL(wc_pack_kernel, dim3((pk.total() + 255) / 256, am.nblk), dim3(256), 0, st, params, (int64_t)WideNet{D, H, 1, 2}.nparam(), pk, packs);
    WcFwdArgs g = {};
    g.packs = packs; g.am = am; g.x = obs; g.x_as = as; g.x_rs = rs; g.rows = rows; g.D = D; g.x_vec = wc_rows_vec(obs, as, rs);
    g.out = out;
    if (rec != nullptr) {
        g.y1 = rec; g.y2 = rec + (int64_t)P * rows * H;
        hipLaunchKernelGGL((wc_fwd_kernel<H, true>), dim3(wc_tiles(rows), P), dim3(256), 0, st, g);
    } else {
        hipLaunchKernelGGL((wc_fwd_kernel<H, false>), dim3(wc_tiles(rows), P), dim3(256), 0, st, g);
    }
    MARL_CHECK_LAUNCH("wc_fwd_kernel");
    return 0;
}

// grad[blk][nparam] = d(sum_rows dout[row] critic(x row))/dparams / sum(filled) from the hidden layers wc_forward_rows left in rec;
// loss[0] = sum(lrow) / sum(filled), loss[1] = sum(filled).  ws: wc_ws(..).total bytes.
template <int H>
int wc_backward_rows(int P, const AgentMap& am, const float* params, int D, const float* obs, int64_t as, int64_t rs, int rows, const float* filled,
                     const float* dout, const float* lrow, void* ws, float* grad, float* loss, hipStream_t st, const float* rec) {
    const WcWs w = wc_ws(P, rows, D, H);
    const WideNet net{D, H, 1, 2};
    const WcPack pk(D, H);
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    float *packs = f(w.packs), *nf = f(w.nf), *gp = f(w.gp);
    const int64_t nparam = net.nparam();
    wide_count(filled, lrow, rows, nf + 4, nf, st);
    hipLaunchKernelGGL(wc_pack_kernel, dim3((pk.total() + 255) / 256, am.nblk), dim3(256), 0, st, params, nparam, pk, packs);
    const int nwg = wc_bwd_wgs(P, rows), nslots = nwg * (256 / H), ngroups = (nslots + WC_FOLD_GROUP - 1) / WC_FOLD_GROUP, partn = wc_partn(H);
    WcBwdArgs b = {};
    b.packs = packs; b.am = am; b.rows = rows; b.D = D; b.tiles = wc_tiles(rows); b.dout = dout; b.y1 = rec; b.y2 = rec + (int64_t)P * rows * H;
    b.d1 = f(w.d1); b.part = f(w.part); b.w2part = f(w.w2part);
    constexpr int LDSB = wc_bwd_lds_floats<H>() * (int)sizeof(float);
    static LdsAttr attr_set;
    if (attr_set.need()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wc_bwd_kernel<H>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
        attr_set.done();
    }
    hipLaunchKernelGGL((wc_bwd_kernel<H>), dim3(nwg, P), dim3(256), LDSB, st, b);
    hipLaunchKernelGGL(wc_fold_w_kernel, dim3((H * H + 255) / 256, P), dim3(256), 0, st, (const float*)b.w2part, nwg, H * H, (const float*)nf + 1, gp, nparam,
                       (int)net.oW(2));
    hipLaunchKernelGGL(wc_fold_cols_a_kernel, dim3((partn + 255) / 256, ngroups, P), dim3(256), 0, st, (const float*)b.part, nslots, partn, f(w.g1));
    hipLaunchKernelGGL(wc_fold_cols_b_kernel, dim3((partn + 255) / 256, P), dim3(256), 0, st, (const float*)f(w.g1), ngroups, H, (const float*)nf + 1, gp,
                       nparam, (int)net.oW(3), (int)net.ob(3), (int)net.ob(2), (int)net.ob(1));
    const int splits = wc_wgrad_splits(P, rows, D), chunk = (((rows + splits - 1) / splits) + 15) & ~15;
    auto wgrad = [&](const float* dy, const float* yp, int64_t yp_as, int64_t yp_rs, int N, int ow, bool vec) {
        WcWgradArgs a = {};
        a.dy = dy; a.dy_as = (int64_t)rows * H; a.yp = yp; a.yp_as = yp_as; a.yp_rs = yp_rs; a.rows = rows; a.N = N; a.k_chunk = chunk; a.vec = vec;
        a.part = f(w.wpart);
        hipLaunchKernelGGL((wc_wgrad_kernel<H>), dim3((N + 127) / 128, P, splits), dim3(256), 0, st, a);
        hipLaunchKernelGGL(wc_fold_w_kernel, dim3((H * N + 255) / 256, P), dim3(256), 0, st, (const float*)a.part, splits, H * N, (const float*)nf + 1, gp,
                           nparam, ow);
    };
    wgrad(b.d1, obs, as, rs, D, (int)net.oW(1), wc_rows_vec(obs, as, rs));
    const int n = am.nblk * (int)nparam;
    hipLaunchKernelGGL(wide_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)gp, P, (int)nparam, am, grad);
    wide_count(filled, lrow, rows, nf + 4, loss, st);
    MARL_CHECK_LAUNCH("wc_bwd_kernel");
    return 0;
}

}  // namespace marl
