// Two-hidden-layer MLPs whose first layer is too wide for the fused kernels: centralised critics (critic.centralised, marlbase/ac/model.py:62-66,
// 155-157: every critic reads the concatenation of all agents' observations) of more than 4 agents or of the warehouse's 71-wide
// observations.  The fused kernels keep one agent's weights in LDS / registers and the dW1 accumulators in registers, which caps the
// input width; here the three layers are plain f32 MFMA GEMMs over all rows with the activations in HBM, sized at RUN time
// (input width, hidden width, outputs): any width works, at the price of writing and re-reading [rows][H] activations.
//
//   forward   Y1 = relu(X W1^T + b1), Y2 = relu(Y1 W2^T + b2), out = Y2 W3^T + b3      (FCNetwork, utils/models.py:34-48; 1 .. 4 hidden layers)
//   backward  dY2 = (dout W3) * (Y2 > 0), dY1 = (dY2 W2) * (Y1 > 0), dWk = dYk^T [Y(k-1) | 1]  (the ones column yields dbk)
//
// One kernel, wide_gemm_kernel: C[m][n] = sum_k A(m, k) B(k, n) on v_mfma_f32_16x16x4_f32, a 64 x 64 tile per workgroup (wave w: rows
// 16w..16w+15, four 16 x 16 column tiles), k in slices of 16 staged through LDS ([k][64 + 16] floats: the four k-quarters of an MFMA
// operand read land in different banks), the next slice prefetched into registers while the current one multiplies.  Operands are
// addressed by (row, column) strides, so X^T / W^T never exist in memory; weight gradients split the row dimension over grid.z and a
// second kernel adds the partial tiles in a fixed order (bitwise reproducible, no atomics) and applies 1 / sum(filled).
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "mlp.h"

namespace marl {

// (Round 3: a form with UNCONDITIONAL operand loads from clamped addresses - the `in range ? A[...] : 0` fetches compile to a branch around
// every load, 24 - 102 loads in one dependent chain per k-slice (profiles/r02_isa_scan.txt) - was measured and taken out again: it is 8 - 11 %
// SLOWER on every GEMM-path row (MAA2C 15x15-8p 6.27 -> 5.73 M env-steps/s, MAPPO rware 4.83 -> 4.31 M, IDQN 256-256 0.725 -> 0.681 M;
// profiles/r03_flatload_ab.md): the clamped form issues every load of the padded tile, the branchy one skips the out-of-range ones, and
// these GEMMs are not bound by load latency.)

// (Round 4: the product WITHOUT LDS and barriers - every wave fetching its 64 x 64 quadrant's operands from global memory straight into
// MFMA layout (a 16-byte load per tile and slice along k-contiguous operands, four dword loads along m / n-contiguous ones), two or
// three slices deep - was built as a drop-in for wide_gemm128_kernel and measured 20 - 30 % SLOWER on every GEMM-path row (MAA2C
// 15x15-8p 7.47 -> 6.31 / 5.79 M env-steps/s at depth 2 / 3, MAPPO rware-tiny-4ag 6.29 -> 4.95 / 4.34 M, IDQN 256-256 0.89 -> 0.71 /
// 0.61 M; scripts/gpu_runs/r4J.sh): an operand fetch in MFMA layout touches 16 rows x 64 bytes per instruction and every wave repeats
// its workgroup neighbours' loads, so the texture path, not the matrix pipe, sets the pace.  Staging through LDS stays.)

#ifndef MARL_WIDE_PF
// slices of register prefetch in wide_gemm128_kernel.  Measured (scripts/gpu_runs/r4M.sh): 2 (a second register set, the slice stored at the
// top of iteration k requested two iterations earlier) is within 1 % of 1 on every GEMM-path row - resident workgroups hide the latency
#define MARL_WIDE_PF 1
#endif
#ifndef MARL_WIDE_KM
#define MARL_WIDE_KM 1  // k-major LDS tiles for the weight-gradient products of wide_gemm128_kernel (0: the transposing row-major form, for A/B runs)
#endif
#ifndef MARL_WIDE_KB
// k depth of one LDS slice of wide_gemm128_kernel.  Measured (scripts/gpu_runs/r3Q.sh): 32 (128 MFMAs per wave between barrier pairs, 37 KB of
// LDS, twice the prefetch registers) is 3 - 13 % SLOWER than 16 on every GEMM-path row (MAPPO rware 6.20 -> 5.38 M): the kernels live on
// resident workgroups per CU, not on barrier count.
#define MARL_WIDE_KB 16
#endif

struct GemmOp {
    const float* A; int64_t a_m, a_k;      // A(m, k) = A[m * a_m + k * a_k]
    const float* B; int64_t b_k, b_n;      // B(k, n) = B[k * b_k + n * b_n]; column n == b_ones reads as 1 (bias-gradient column)
    float* C; int64_t c_m;                 // C[z * c_split + m * c_m + n]
    int M, N, K;
    int k_chunk;                           // grid.z = z covers k in [z * k_chunk, min(K, (z + 1) * k_chunk))
    int64_t c_split;
    const float* bias;                     // epi 1, 2: + bias[n]
    const float* gate; int64_t gate_m;     // epi 3: * (gate[m * gate_m + n] > 0)   (relu mask of the activation the gradient flows into)
    int b_ones;                            // -1: none
    int epi;                               // 0 store, 1 + bias, 2 relu(+ bias), 3 gate
    bool a_vec, b_vec;                     // 16-byte operand loads are legal (set by wide_gemm from the pointers and strides)
};

// Four consecutive elements c .. c + 3 (c a multiple of 4) along an operand's CONTIGUOUS dimension on line `line` (element (line, c) at
// base[line * ls + c]); elements at or past `lim`, or on a line that is out of range, read as 0; element `ones` (the bias-gradient
// column, -1: none) reads as 1.  vec (wave-uniform, set by wide_gemm when base, line stride and slice origin are 16-byte aligned): ONE
// 16-byte load when the four are in range - a wave's 64 lanes then fetch whole 64-byte runs instead of 64 x 4 bytes from 64 different
// rows per instruction (the scalar form of round 2 was bound by the texture addresser, not by its MFMAs: 0.21 - 0.27 of the f32 peak).
__device__ __forceinline__ f4 gemm_load4(const float* __restrict__ base, int64_t line, int64_t ls, int c, int lim, bool line_ok, int ones, bool vec) {
    f4 v = {0.f, 0.f, 0.f, 0.f};
    if (!line_ok) return v;
    const float* p = base + line * ls + c;
    if (vec && c + 3 < lim) {
        v = *reinterpret_cast<const f4*>(p);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e < lim) v[e] = p[e];
    }
    if (ones >= 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (c + e == ones) v[e] = 1.f;
    }
    return v;
}

// A_KC / B_KC: the operand is contiguous along k (true) or along m / n (false) - picks the global-load pattern that coalesces.
// Thread map of a 16-deep slice: k-contiguous operands - thread t takes row t / 4 and k = 4 (t % 4) .. + 3 (16 lanes cover four rows' 64 bytes);
// m / n-contiguous ones - thread t takes k = t / 16 and rows 4 (t % 16) .. + 3.
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void wide_gemm_kernel(const GemmOp g) {
    constexpr int LD = 80;
    __shared__ float As[16 * LD], Bs[16 * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int kbeg = blockIdx.z * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    f4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f4{0.f, 0.f, 0.f, 0.f};
    f4 ra, rb;
    auto load = [&](int k0) {
        if (A_KC) ra = gemm_load4(g.A, m0 + (tid >> 2), g.a_m, k0 + 4 * (tid & 3), kend, m0 + (tid >> 2) < g.M, -1, g.a_vec);
        else ra = gemm_load4(g.A, k0 + (tid >> 4), g.a_k, m0 + 4 * (tid & 15), g.M, k0 + (tid >> 4) < kend, -1, g.a_vec);
        if (B_KC) {
            const int n = n0 + (tid >> 2);
            rb = gemm_load4(g.B, n, g.b_n, k0 + 4 * (tid & 3), kend, n < g.N && n != g.b_ones, -1, g.b_vec);
            if (n == g.b_ones) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rb[e] = k0 + 4 * (tid & 3) + e < kend ? 1.f : 0.f;
            }
        } else {
            rb = gemm_load4(g.B, k0 + (tid >> 4), g.b_k, n0 + 4 * (tid & 15), g.b_ones >= 0 ? g.N - 1 : g.N, k0 + (tid >> 4) < kend, g.b_ones, g.b_vec);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (A_KC) As[(4 * (tid & 3) + e) * LD + (tid >> 2)] = ra[e];
            else As[(tid >> 4) * LD + 4 * (tid & 15) + e] = ra[e];
            if (B_KC) Bs[(4 * (tid & 3) + e) * LD + (tid >> 2)] = rb[e];
            else Bs[(tid >> 4) * LD + 4 * (tid & 15) + e] = rb[e];
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        __syncthreads();  // the previous slice has been multiplied
        store();
        __syncthreads();
        if (k0 + 16 < kend) load(k0 + 16);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float a = As[(4 * s + q) * LD + 16 * wave + i];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MARL_MFMA(a, Bs[(4 * s + q) * LD + 16 * nt + i], acc[nt]);
        }
    }
    float* C = g.C + (int64_t)blockIdx.z * g.c_split;
    // epilogue operands requested together from clamped addresses (see wide_gemm128_kernel): no dependent round trip per element
    f4 bias4 = {0.f, 0.f, 0.f, 0.f}, gate[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + 16 * nt + i, nc = n < g.N ? n : g.N - 1;
        if (g.epi == 1 || g.epi == 2) bias4[nt] = g.bias[nc];
        if (g.epi == 3) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 16 * wave + 4 * q + r;
                gate[nt][r] = g.gate[(int64_t)(m < g.M ? m : g.M - 1) * g.gate_m + nc];
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + 16 * nt + i;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + 16 * wave + 4 * q + r;
            if (m < g.M && n < g.N) {
                float v = acc[nt][r];
                if (g.epi == 1 || g.epi == 2) v += bias4[nt];
                if (g.epi == 2) v = fmaxf(v, 0.f);
                if (g.epi == 3) v = gate[nt][r] > 0.f ? v : 0.f;
                C[(int64_t)m * g.c_m + n] = v;
            }
        }
    }
}

// The same product on a 128 x 128 tile per workgroup for the large GEMMs (both output dimensions >= 128: hidden layers over all rows,
// weight gradients of wide layers): wave w owns a 64 x 64 quadrant = 4 x 4 MFMA tiles, so a 16-deep k-slice costs 8 ds_read_b128
// for 64 MFMAs (the 64 x 64 kernel above: 20 ds_read_b32 for 16).  LDS layout [row][16 + 4], k in its natural order: lane quarter q
// takes k = 4 q + s at MFMA step s (the same assignment on both operands), so its four steps' operands are one 16-byte read and the
// k-contiguous global loads land as 16-byte stores.  Two passes of the thread map above per slice (128 rows).
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void wide_gemm128_kernel(const GemmOp g) {
    constexpr int KB = MARL_WIDE_KB, NS = KB / 16, LD = KB + 4;  // slice depth: NS sub-slices of 16 between two barriers
    __shared__ __attribute__((aligned(16))) float As[128 * LD], Bs[128 * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, q = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;  // the wave's quadrant
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const int kbeg = blockIdx.z * g.k_chunk, kend = min(g.K, kbeg + g.k_chunk);
    f4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f4{0.f, 0.f, 0.f, 0.f};
    // PF register sets: the slice stored to LDS at the top of iteration k was requested PF iterations earlier (MARL_WIDE_PF above)
    constexpr int PF = MARL_WIDE_PF;  // slices of prefetch: 2 = two register sets, 1 = one
    f4 rset[PF][2][NS][2];  // [set][A | B][sub-slice][row half]
    auto load = [&](int kbase, int set) {
        f4 (&ra)[NS][2] = rset[set][0];
        f4 (&rb)[NS][2] = rset[set][1];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            const int k0 = kbase + 16 * ss;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (A_KC) {
                    const int m = m0 + 64 * h + (tid >> 2);
                    ra[ss][h] = gemm_load4(g.A, m, g.a_m, k0 + 4 * (tid & 3), kend, m < g.M, -1, g.a_vec);
                } else {
                    ra[ss][h] = gemm_load4(g.A, k0 + (tid >> 4), g.a_k, m0 + 64 * h + 4 * (tid & 15), g.M, k0 + (tid >> 4) < kend, -1, g.a_vec);
                }
                if (B_KC) {
                    const int n = n0 + 64 * h + (tid >> 2);
                    rb[ss][h] = gemm_load4(g.B, n, g.b_n, k0 + 4 * (tid & 3), kend, n < g.N && n != g.b_ones, -1, g.b_vec);
                    if (n == g.b_ones) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) rb[ss][h][e] = k0 + 4 * (tid & 3) + e < kend ? 1.f : 0.f;
                    }
                } else {
                    rb[ss][h] = gemm_load4(g.B, k0 + (tid >> 4), g.b_k, n0 + 64 * h + 4 * (tid & 15), g.b_ones >= 0 ? g.N - 1 : g.N,
                                           k0 + (tid >> 4) < kend, g.b_ones, g.b_vec);
                }
            }
        }
    };
    // weight-gradient products (neither operand k-contiguous: both are contiguous ACROSS the reduction index) keep their LDS tiles K-MAJOR
    // (round 4, from wide_critic.h's wc_wgrad_kernel): [k][128 + 16] - the 16-byte global loads land as 16-byte LDS stores and an MFMA
    // operand is a conflict-free 4-byte read (lane quarter q takes k = 4 s + q at step s: rows 144 floats apart fall 16 banks apart).
    // The row-major form transposed with four scalar stores per load, 8 lanes to a bank.
    constexpr bool KM = MARL_WIDE_KM && !A_KC && !B_KC;
    constexpr int LDK = 128 + 16;
    static_assert(!KM || KB * LDK <= 128 * LD, "k-major tiles fit the row-major arrays");
    auto store = [&](int set) {
        f4 (&ra)[NS][2] = rset[set][0];
        f4 (&rb)[NS][2] = rset[set][1];
#pragma unroll
        for (int ss = 0; ss < NS; ++ss)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if constexpr (KM) {
                    *reinterpret_cast<f4*>(As + (16 * ss + (tid >> 4)) * LDK + 64 * h + 4 * (tid & 15)) = ra[ss][h];
                    *reinterpret_cast<f4*>(Bs + (16 * ss + (tid >> 4)) * LDK + 64 * h + 4 * (tid & 15)) = rb[ss][h];
                    continue;
                }
                if (A_KC) {
                    *reinterpret_cast<f4*>(As + (64 * h + (tid >> 2)) * LD + 16 * ss + 4 * (tid & 3)) = ra[ss][h];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) As[(64 * h + 4 * (tid & 15) + e) * LD + 16 * ss + (tid >> 4)] = ra[ss][h][e];
                }
                if (B_KC) {
                    *reinterpret_cast<f4*>(Bs + (64 * h + (tid >> 2)) * LD + 16 * ss + 4 * (tid & 3)) = rb[ss][h];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) Bs[(64 * h + 4 * (tid & 15) + e) * LD + 16 * ss + (tid >> 4)] = rb[ss][h][e];
                }
            }
    };
    auto slice = [&](int k0, int set) {  // one k-slice: its operands to LDS, the slice two ahead requested, then the MFMAs
        __syncthreads();
        store(set);
        __syncthreads();
        if (k0 + PF * KB < kend) load(k0 + PF * KB, set);
#pragma unroll
        for (int ss = 0; ss < NS; ++ss) {
            if constexpr (KM) {
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) {
                    float ak[4], bk[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        ak[t] = As[(16 * ss + 4 * s2 + q) * LDK + 64 * wm + 16 * t + i];
                        bk[t] = Bs[(16 * ss + 4 * s2 + q) * LDK + 64 * wn + 16 * t + i];
                    }
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MARL_MFMA(ak[mt], bk[nt], acc[mt][nt]);
                }
                continue;
            }
            f4 a[4], b[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[t] = *reinterpret_cast<const f4*>(As + (64 * wm + 16 * t + i) * LD + 16 * ss + 4 * q);
                b[t] = *reinterpret_cast<const f4*>(Bs + (64 * wn + 16 * t + i) * LD + 16 * ss + 4 * q);
            }
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = MARL_MFMA(a[mt][s2], b[nt][s2], acc[mt][nt]);
        }
    };
    if (kbeg < kend) load(kbeg, 0);
    if constexpr (PF == 2) {
        if (kbeg + KB < kend) load(kbeg + KB, 1);
        for (int k0 = kbeg; k0 < kend; k0 += 2 * KB) {
            slice(k0, 0);
            if (k0 + KB < kend) slice(k0 + KB, 1);
        }
    } else {
        for (int k0 = kbeg; k0 < kend; k0 += KB) slice(k0, 0);
    }
    float* C = g.C + (int64_t)blockIdx.z * g.c_split;
    // epilogue operands up front (round 4): a bias or gate value fetched under the `in range` branch of ITS element is a dependent round
    // trip per element - 64 per lane, first-touch HBM accesses for the gate - and made the layer-to-layer gradient product an
    // epilogue-latency kernel (the static scan lists it as a chain of 64 loads).  The lane's 4 bias values and its 64 gate values are
    // requested in one go from clamped addresses; out-of-range elements are still not stored.
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.epi == 1 || g.epi == 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + 64 * wn + 16 * nt + i;
            bias4[nt] = g.bias[n < g.N ? n : g.N - 1];
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        f4 gate[4];
        if (g.epi == 3) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = n0 + 64 * wn + 16 * nt + i;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + 64 * wm + 16 * mt + 4 * q + r;
                    gate[nt][r] = g.gate[(int64_t)(m < g.M ? m : g.M - 1) * g.gate_m + (n < g.N ? n : g.N - 1)];
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + 64 * wn + 16 * nt + i;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 64 * wm + 16 * mt + 4 * q + r;
                if (m < g.M && n < g.N) {
                    float v = acc[mt][nt][r];
                    if (g.epi == 1 || g.epi == 2) v += bias4[nt];
                    if (g.epi == 2) v = fmaxf(v, 0.f);
                    if (g.epi == 3) v = gate[nt][r] > 0.f ? v : 0.f;
                    C[(int64_t)m * g.c_m + n] = v;
                }
            }
        }
    }
}

// dW[m][n < N - 1] and db[m] (column N - 1) = (sum over the splits, in split order) * inv, inv = 1 / nf[0]
static __global__ __launch_bounds__(256) void wide_fold_kernel(const float* __restrict__ partial, int splits, int64_t split_stride, int M, int N,
                                                               const float* __restrict__ nf, float* __restrict__ dW, float* __restrict__ db) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= M * N) return;
    float acc = 0.f;
    for (int z = 0; z < splits; ++z) acc += partial[(int64_t)z * split_stride + idx];
    acc /= nf[0];
    const int m = idx / N, n = idx - m * N;
    if (n < N - 1) dW[(int64_t)m * (N - 1) + n] = acc;
    else db[m] = acc;
}

// out[0] = sum(lrow) / nf, out[1] = nf = sum(filled) in two fixed-order stages (the launch_backward_rows contract): per-workgroup sums of a
// contiguous slice each, then one workgroup adds them in slice order
constexpr int WIDE_COUNT_MAX_WG = 256;
static __global__ __launch_bounds__(256) void wide_count_kernel(const float* __restrict__ filled, const float* __restrict__ lrow, int n,
                                                                float* __restrict__ part /* [gridDim.x][2] */) {
    __shared__ float sh[2][4];
    const int per = (n + gridDim.x - 1) / gridDim.x, beg = blockIdx.x * per, end = min(n, beg + per);
    float nf = 0.f, ls = 0.f;
    for (int k = beg + threadIdx.x; k < end; k += 256) {
        nf += filled[k];
        if (lrow != nullptr) ls += lrow[k];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        nf += __shfl_xor(nf, off);
        ls += __shfl_xor(ls, off);
    }
    if ((threadIdx.x & 63) == 0) {
        sh[0][threadIdx.x >> 6] = nf;
        sh[1][threadIdx.x >> 6] = ls;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
        part[2 * blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
    }
}
static __global__ __launch_bounds__(64) void wide_count_final_kernel(const float* __restrict__ part, int nwg, float* __restrict__ out) {
    if (threadIdx.x != 0) return;
    float nf = 0.f, ls = 0.f;
    for (int w = 0; w < nwg; ++w) {
        nf += part[2 * w];
        ls += part[2 * w + 1];
    }
    out[0] = ls / nf;
    out[1] = nf;
}
inline void wide_count(const float* filled, const float* lrow, int n, float* part, float* out, hipStream_t st) {
    int nwg = (n + 4095) / 4096;
    nwg = nwg < 1 ? 1 : (nwg > WIDE_COUNT_MAX_WG ? WIDE_COUNT_MAX_WG : nwg);
    hipLaunchKernelGGL(wide_count_kernel, dim3(nwg), dim3(256), 0, st, filled, lrow, n, part);
    hipLaunchKernelGGL(wide_count_final_kernel, dim3(1), dim3(64), 0, st, (const float*)part, nwg, out);
}

// grad[blk][i] = sum over the agents of block blk, in agent order, of gp[p][i]
static __global__ __launch_bounds__(256) void wide_gather_kernel(const float* __restrict__ gp, int P, int nparam, AgentMap am,
                                                                 float* __restrict__ grad) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= am.nblk * nparam) return;
    const int blk = idx / nparam, k = idx - blk * nparam;
    float acc = 0.f;
    for (int p = 0; p < P; ++p)
        if (am.net[p] == blk) acc += gp[(int64_t)p * nparam + k];
    grad[idx] = acc;
}

// run-time shape of one network: D inputs, L hidden layers of H units (FCNetwork builds any list, marlbase/utils/models.py:34-48;
// unequal widths are zero-padded to H by the caller), A outputs; parameters in FCNetwork's parameters() order.
// Layers are numbered 1 .. L (hidden) and L + 1 (output).
struct WideNet {
    static constexpr int MAXL = 16;  // == marlhip_net_shape.n_hidden's upper bound (common.h: net_shape_validate)
    int D, H, A;
    int L = 2;
    int n_in(int k) const { return k == 1 ? D : H; }
    int n_out(int k) const { return k <= L ? H : A; }
    int64_t oW(int k) const {
        int64_t off = 0;
        for (int j = 1; j < k; ++j) off += (int64_t)n_out(j) * n_in(j) + n_out(j);
        return off;
    }
    int64_t ob(int k) const { return oW(k) + (int64_t)n_out(k) * n_in(k); }
    int64_t nparam() const { return oW(L + 2); }
    int widest() const { return D > H ? D : H; }
};

inline int wide_splits(int rows) {  // row slices of the weight-gradient GEMMs: >= 2048 rows each, at most 256 slices
    int s = rows / 2048;
    return s < 1 ? 1 : (s > 256 ? 256 : s);
}

struct WideWs {
    int64_t y[WideNet::MAXL], d[2], partial, gp, nf, total;  // byte offsets
};

// backward = true: the whole workspace of wide_backward_rows; false: the activation buffers of wide_forward_rows
inline WideWs wide_ws(const WideNet& s, int P, int rows, bool backward) {
    WideWs w = {};
    int64_t o = 0;
    auto take = [&](int64_t nfloat) { const int64_t at = o; o = (o + nfloat * 4 + 255) & ~(int64_t)255; return at; };
    for (int k = 0; k < s.L; ++k) w.y[k] = take((int64_t)rows * s.H);
    if (backward) {
        w.d[0] = take((int64_t)rows * s.H);
        w.d[1] = take((int64_t)rows * s.H);
        w.partial = take((int64_t)wide_splits(rows) * s.H * (s.widest() + 1));
        w.gp = take((int64_t)P * s.nparam());
        w.nf = take(4 + 2 * WIDE_COUNT_MAX_WG);  // [loss, nf | slack | per-workgroup partials]
    }
    w.total = o;
    return w;
}

template <bool A_KC, bool B_KC>
inline void wide_gemm(const GemmOp& g0, int splits, hipStream_t st) {
    GemmOp g = g0;
    if ((A_KC ? g.a_k : g.a_m) != 1 || (B_KC ? g.b_k : g.b_n) != 1) {  // (a programming error, not a run-time condition)
        fprintf(stderr, "wide_gemm: operand not unit-stride along its contiguous dimension\n");
        abort();
    }
    {   // 16-byte loads along the contiguous dimension: unit stride there, line stride and base a multiple of 4 floats / 16 bytes, and the
        // slice origins multiples of 4 (k-contiguous operands start at z * k_chunk; m / n-contiguous ones at multiples of 64 / 128)
        auto al = [](const float* p, int64_t unit, int64_t line) { return unit == 1 && (line & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
        const bool kc_ok = (g.k_chunk & 3) == 0 || splits == 1;
        g.a_vec = A_KC ? al(g.A, g.a_k, g.a_m) && kc_ok : al(g.A, g.a_m, g.a_k);
        g.b_vec = B_KC ? al(g.B, g.b_k, g.b_n) && kc_ok : al(g.B, g.b_n, g.b_k);
        if (getenv("MARLHIP_WIDE_SCALAR_LOADS") != nullptr) g.a_vec = g.b_vec = false;  // diagnostics: the element-wise loads
    }
    static const bool small_only = getenv("MARLHIP_WIDE_GEMM64") != nullptr;  // diagnostics: the 64 x 64 kernel for everything
    const int64_t wgs128 = (int64_t)((g.N + 127) / 128) * ((g.M + 127) / 128) * splits;
    if (g.M >= 128 && g.N >= 128 && wgs128 >= 512 && !small_only)  // (a smaller launch fills the chip better with 64 x 64 tiles)
        hipLaunchKernelGGL((wide_gemm128_kernel<A_KC, B_KC>), dim3((g.N + 127) / 128, (g.M + 127) / 128, splits), dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((wide_gemm_kernel<A_KC, B_KC>), dim3((g.N + 63) / 64, (g.M + 63) / 64, splits), dim3(256), 0, st, g);
}

// hidden activations Y_1 .. Y_L of `rows` rows x (row r at x + r * row_stride) for one network: y[k - 1] = Y_k [rows][H]
inline void wide_hidden(const WideNet& s, const float* prm, const float* x, int64_t row_stride, int rows, float* const* y, hipStream_t st) {
    for (int k = 1; k <= s.L; ++k) {
        GemmOp g = {};
        g.M = rows; g.N = s.H; g.K = s.n_in(k); g.k_chunk = 1 << 30; g.b_ones = -1; g.epi = 2;
        g.A = k == 1 ? x : y[k - 2]; g.a_m = k == 1 ? row_stride : s.H; g.a_k = 1;
        g.B = prm + s.oW(k); g.b_k = 1; g.b_n = s.n_in(k);
        g.C = y[k - 1]; g.c_m = s.H; g.bias = prm + s.ob(k);
        wide_gemm<true, true>(g, 1, st);
    }
}

// out[p][row][A] = MLP_p(x row); x row r of agent p at obs + p * agent_stride + r * row_stride; scratch: wide_ws(.., false).total bytes
inline int wide_forward_rows(const WideNet& s, int P, const AgentMap& am, const float* params, const float* obs, int64_t agent_stride,
                             int64_t row_stride, int rows, float* out, void* scratch, hipStream_t st) {
    const WideWs w = wide_ws(s, P, rows, false);
    float* y[WideNet::MAXL];
    for (int k = 0; k < s.L; ++k) y[k] = reinterpret_cast<float*>(static_cast<char*>(scratch) + w.y[k]);
    for (int p = 0; p < P; ++p) {
        const float* prm = params + (int64_t)am.net[p] * s.nparam();
        wide_hidden(s, prm, obs + (int64_t)p * agent_stride, row_stride, rows, y, st);
        GemmOp g = {};
        g.M = rows; g.N = s.A; g.K = s.H; g.k_chunk = 1 << 30; g.b_ones = -1; g.epi = 1;
        g.A = y[s.L - 1]; g.a_m = s.H; g.a_k = 1; g.B = prm + s.oW(s.L + 1); g.b_k = 1; g.b_n = s.H; g.C = out + (int64_t)p * rows * s.A; g.c_m = s.A;
        g.bias = prm + s.ob(s.L + 1);
        wide_gemm<true, true>(g, 1, st);
    }
    MARL_CHECK_LAUNCH("wide_gemm_kernel (forward)");
    return 0;
}

// grad[blk][nparam] = d(sum_rows <dout row, MLP(x row)>)/dparams / sum(filled); dout: agent p's [rows][A] at dout + p * dout_agent_stride
// (already masked by filled);
// loss[0] = sum(lrow) / sum(filled), loss[1] = sum(filled).  ws: wide_ws(.., true).total bytes.
inline int wide_backward_rows(const WideNet& s, int P, const AgentMap& am, const float* params, const float* obs, int64_t agent_stride,
                              int64_t row_stride, int rows, const float* filled, const float* dout, int64_t dout_agent_stride,
                              const float* lrow, void* ws, float* grad, float* loss, hipStream_t st) {
    const WideWs w = wide_ws(s, P, rows, true);
    char* base = static_cast<char*>(ws);
    auto f = [&](int64_t off) { return reinterpret_cast<float*>(base + off); };
    float* y[WideNet::MAXL];
    for (int k = 0; k < s.L; ++k) y[k] = f(w.y[k]);
    float *dbuf[2] = {f(w.d[0]), f(w.d[1])}, *part = f(w.partial), *gp = f(w.gp), *nf = f(w.nf);
    wide_count(filled, lrow, rows, nf + 4, nf, st);
    const int splits = wide_splits(rows), chunk = (((rows + splits - 1) / splits) + 15) & ~15;
    const int H = s.H, A = s.A, L = s.L;
    for (int p = 0; p < P; ++p) {
        const float* prm = params + (int64_t)am.net[p] * s.nparam();
        const float* x = obs + (int64_t)p * agent_stride;
        const float* dq = dout + (int64_t)p * dout_agent_stride;
        float* gpp = gp + (int64_t)p * s.nparam();
        wide_hidden(s, prm, x, row_stride, rows, y, st);
        // weight gradient of a layer: dW[out][in] (+ bias column) = dY^T [Yprev | 1], rows sliced over grid.z, then the fold
        auto wgrad = [&](const float* dy, int n_out, const float* yprev, int64_t yprev_stride, int n_in, float* dW, float* db) {
            GemmOp g = {};
            g.M = n_out; g.N = n_in + 1; g.K = rows; g.k_chunk = chunk; g.b_ones = n_in; g.epi = 0;
            g.A = dy; g.a_m = 1; g.a_k = n_out; g.B = yprev; g.b_k = yprev_stride; g.b_n = 1;
            g.C = part; g.c_m = n_in + 1; g.c_split = (int64_t)n_out * (n_in + 1);
            wide_gemm<false, false>(g, splits, st);
            const int n = n_out * (n_in + 1);
            hipLaunchKernelGGL(wide_fold_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)part, splits, g.c_split, n_out, n_in + 1,
                               (const float*)nf + 1, dW, db);
        };
        // output layer, then the hidden layers from the last to the first: dY_k = (dY_{k+1} W_{k+1}) * (Y_k > 0), dW_k = dY_k^T [Y_{k-1} | 1]
        wgrad(dq, A, y[L - 1], H, H, gpp + s.oW(L + 1), gpp + s.ob(L + 1));
        const float* dnext = dq;
        int n_next = A;
        for (int k = L; k >= 1; --k) {
            float* dk = dbuf[k & 1];
            GemmOp g = {};
            g.M = rows; g.N = H; g.K = n_next; g.k_chunk = 1 << 30; g.b_ones = -1; g.epi = 3;
            g.A = dnext; g.a_m = n_next; g.a_k = 1; g.B = prm + s.oW(k + 1); g.b_k = H; g.b_n = 1; g.C = dk; g.c_m = H; g.gate = y[k - 1]; g.gate_m = H;
            wide_gemm<true, false>(g, 1, st);
            wgrad(dk, H, k == 1 ? x : y[k - 2], k == 1 ? row_stride : H, s.n_in(k), gpp + s.oW(k), gpp + s.ob(k));
            dnext = dk;
            n_next = H;
        }
    }
    const int n = am.nblk * (int)s.nparam();
    hipLaunchKernelGGL(wide_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)gp, P, (int)s.nparam(), am, grad);
    wide_count(filled, lrow, rows, nf + 4, loss, st);
    MARL_CHECK_LAUNCH("wide_gemm_kernel (backward)");
    return 0;
}

}  // namespace marl
