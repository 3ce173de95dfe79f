// The collectors' forward passes that also LEAVE both hidden layers behind (AcKeep, common.h): mlp_forward_g and mlp_forward_h2 of mlp.h
// with two destination pointers.  Group order, operand order and every accumulator's k order are those of mlp.h's functions (and so of
// mlp_forward_p / mlp_forward_p2, the learner step's forward-rows pass): the stored tiles and the returned logits have their bits.  The
// stores go out where the tiles are complete - layer 1 behind its last k-group, layer 2 in front of the output layer - so that no tile
// lives longer than it does in the plain functions.
// d1 / d2: this lane's entry of tile 0 of the layer inside the row block's slot (tile m at + 64 m entries); nullptr: keep nothing.
#pragma once
#include "mlp.h"

namespace marl {

// mlp_forward_g (packs in global memory, three operand groups in flight)
template <class S>
__device__ __forceinline__ void mlp_forward_g_keep(const float* __restrict__ gpack, int lane, const float (&x)[S::KS1], f4& q, f4* __restrict__ d1,
                                                   f4* __restrict__ d2) {
    constexpr int MT = S::MT, N1 = S::KS1 / 4, G = N1 + MT + 1;
    typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gpack), 0, S::NFWD * 4, 0x00020000);
    const int lo16 = lane * 16, g16 = (lane >> 4) * 16;
    auto ld16 = [&](int float_off, int voff) -> f4 {
        const u4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, float_off * 4, 0);
        return __builtin_bit_cast(f4, v);
    };
    f4 op[3][MT], acc[MT], nb[MT], h1[MT], h2[MT], o3;
    auto fetch = [&](auto gi_c) {
        constexpr int gi = decltype(gi_c)::value;
        if constexpr (gi < G) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                op[gi % 3][mt] = gi < N1 ? ld16(S::pA1 + (mt * N1 + gi) * 256, lo16)
                                         : (gi < N1 + MT ? ld16(S::pA2 + (mt * MT + (gi - N1)) * 256, lo16) : ld16(S::pA3 + mt * 256, lo16));
        }
    };
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        acc[mt] = ld16(S::pb1 + 16 * mt, g16);
        nb[mt] = ld16(S::pb2 + 16 * mt, g16);
    }
    o3 = ld16(S::pb3, g16);
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    auto step = [&](auto gi_c) {
        constexpr int gi = decltype(gi_c)::value;
        fetch(std::integral_constant<int, gi + 2>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (gi < N1) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[gi % 3][mt][e], x[4 * gi + e], acc[mt]);
            if constexpr (gi == N1 - 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    h1[mt] = relu4(acc[mt]);
                    acc[mt] = nb[mt];
                }
                if (d1 != nullptr) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) d1[mt * 64] = h1[mt];
                }
            }
        } else if constexpr (gi < N1 + MT) {
            constexpr int k1 = gi - N1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = MARL_MFMA(op[gi % 3][mt][r], h1[k1][r], acc[mt]);
            if constexpr (k1 == MT - 1) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) h2[mt] = relu4(acc[mt]);
                if (d2 != nullptr) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) d2[mt * 64] = h2[mt];
                }
            }
        } else {
#pragma unroll
            for (int k1 = 0; k1 < MT; ++k1)
#pragma unroll
                for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op[gi % 3][k1][r], h2[k1][r], o3);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    marl_static_for<0, G>(step);
    q = o3;
}

// mlp_forward_h2 (one network on two waves): wave `half` leaves ITS tiles [half MT/2, (half + 1) MT/2) of both layers
template <class S, int XR = 1>
__device__ __forceinline__ void mlp_forward_h2_keep(const float* lds, int lane, const float (&x)[S::KS1], int half, f4* xh, f4* xq, f4& q, const f4* a3r,
                                                    f4* __restrict__ d1, f4* __restrict__ d2) {
    constexpr int MT = S::MT, MH = MT / 2, N1 = S::KS1 / 4, MX = MH / XR;
    static_assert(MT % 2 == 0 && MH % XR == 0, "two waves split the hidden tiles evenly");
    const int g = lane >> 4, t0 = half * MH;
    const f4* A1 = reinterpret_cast<const f4*>(lds + S::pA1);
    const f4* A2 = reinterpret_cast<const f4*>(lds + S::pA2);
    const f4* A3 = reinterpret_cast<const f4*>(lds + S::pA3);
    f4 acc[MH], h1[MT], h2[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = *reinterpret_cast<const f4*>(lds + S::pb1 + 16 * (t0 + m) + 4 * g);
#pragma unroll
    for (int s = 0; s < N1; ++s) {
        f4 op[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) op[m] = A1[((t0 + m) * N1 + s) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[m] = MARL_MFMA(op[m][e], x[4 * s + e], acc[m]);
    }
    f4 other[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = relu4(acc[m]);
    if (d1 != nullptr) {
#pragma unroll
        for (int m = 0; m < MH; ++m) d1[(t0 + m) * 64] = acc[m];
    }
#pragma unroll
    for (int xr = 0; xr < XR; ++xr) {
        if (xr > 0) __syncthreads();  // the previous round has been read
#pragma unroll
        for (int m = 0; m < MX; ++m) xh[(half * MX + m) * 64 + lane] = acc[xr * MX + m];
        __syncthreads();
#pragma unroll
        for (int m = 0; m < MX; ++m) other[xr * MX + m] = xh[((1 - half) * MX + m) * 64 + lane];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) h1[mt] = (mt / MH == half) ? acc[mt % MH] : other[mt % MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) acc[m] = *reinterpret_cast<const f4*>(lds + S::pb2 + 16 * (t0 + m) + 4 * g);
#pragma unroll
    for (int k1 = 0; k1 < MT; ++k1) {
        f4 op[MH];
#pragma unroll
        for (int m = 0; m < MH; ++m) op[m] = A2[((t0 + m) * MT + k1) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < MH; ++m) acc[m] = MARL_MFMA(op[m][r], h1[k1][r], acc[m]);
    }
#pragma unroll
    for (int m = 0; m < MH; ++m) h2[m] = relu4(acc[m]);
    if (d2 != nullptr) {
#pragma unroll
        for (int m = 0; m < MH; ++m) d2[(t0 + m) * 64] = h2[m];
    }
    f4 op3[MH];
#pragma unroll
    for (int m = 0; m < MH; ++m) op3[m] = a3r != nullptr ? a3r[m] : A3[(t0 + m) * 64 + lane];
    f4 o3 = *reinterpret_cast<const f4*>(lds + S::pb3 + 4 * g);
    if (half == 0) {
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op3[m][r], h2[m][r], o3);
        xq[lane] = o3;
    }
    __syncthreads();
    if (half == 1) {
        o3 = xq[lane];
#pragma unroll
        for (int m = 0; m < MH; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) o3 = MARL_MFMA(op3[m][r], h2[m][r], o3);
    }
    q = o3;
}

}  // namespace marl
