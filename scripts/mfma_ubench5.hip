// bf16 MFMA (v_mfma_f32_16x16x32_bf16) next to VALU work: what a split-bf16 ("bf16 x 3", fp32 accumulate) form of the learner could
// run at.  One f32 product group of K = 32 (8 x v_mfma_f32_16x16x4_f32 = 256 cycles) becomes 6 bf16 MFMAs (hi*hi, hi*mid, mid*hi,
// hi*lo, lo*hi, mid*mid) plus the on-the-fly split of the activation operand (per f32 value: 3 conversions, 2 subtractions).
//   A: bf16 MFMAs alone, 1 / 4 accumulator chains
//   B: 6 bf16 MFMAs + n independent VALU instructions per group (does the vector ALU overlap with the bf16 matrix pipe?)
//   C: the split itself: 8 f32 values -> three packed bf16x8 operands, then the 6 MFMAs (B operand split per group, A pre-split)
//   D: f32 reference: 8 x v_mfma_f32_16x16x4_f32 per group
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define MFMA_BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {  // v_cvt_pk_bf16_f32 (round to nearest even)
    unsigned r;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

// 8 f32 values -> (hi, mid, lo) packed bf16x8; x = hi + mid + lo up to 2^-24 |x|
__device__ __forceinline__ void split8(const float (&x)[8], bf8& hi, bf8& mid, bf8& lo) {
    u4 h, m, l;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float a = x[2 * k], b = x[2 * k + 1];
        const unsigned ph = pk_bf16(a, b);
        const float ra = a - __uint_as_float(ph << 16), rb = b - __uint_as_float(ph & 0xFFFF0000u);
        const unsigned pm = pk_bf16(ra, rb);
        const float sa = ra - __uint_as_float(pm << 16), sb = rb - __uint_as_float(pm & 0xFFFF0000u);
        h[k] = ph; m[k] = pm; l[k] = pk_bf16(sa, sb);
    }
    hi = __builtin_bit_cast(bf8, h); mid = __builtin_bit_cast(bf8, m); lo = __builtin_bit_cast(bf8, l);
}

// MODE 0: CH chains of bf16 MFMAs only (6 per "group"); 1: + NV v_max per group; 2: + the split of one B operand per group;
// 3: f32 reference (8 f32 MFMAs per group per chain)
template <int CH, int NV, int MODE>
__global__ __launch_bounds__(256, 1) void kb(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    float x[8], v[16];
    for (int i = 0; i < 8; ++i) x[i] = in[lane + i];
    for (int i = 0; i < 16; ++i) v[i] = in[lane + 8 + i];
    bf8 ah, am, al, bh, bm, bl;
    split8(x, ah, am, al);
    split8(x, bh, bm, bl);
    const float fa = in[lane], fb = in[lane + 64];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
#pragma unroll
                for (int c = 0; c < CH; ++c) acc[c] = MFMA_F32(fa, fb, acc[c]);
        } else {
            if (MODE == 2) {
                x[0] += 1e-30f * acc[0][0];  // a data dependence on the previous group, as a layer chain has
                split8(x, bh, bm, bl);
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                acc[c] = MFMA_BF(ah, bh, acc[c]);
                acc[c] = MFMA_BF(ah, bm, acc[c]);
                acc[c] = MFMA_BF(am, bh, acc[c]);
                acc[c] = MFMA_BF(ah, bl, acc[c]);
                acc[c] = MFMA_BF(al, bh, acc[c]);
                acc[c] = MFMA_BF(am, bm, acc[c]);
            }
            if (MODE == 1) {
#pragma unroll
                for (int i = 0; i < NV; ++i) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i & 15]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 5) & 15]));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s + x[0];
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <class K>
double run(K kern, int groups_per_iter) {
    float *in, *out;
    unsigned long long* cyc;
    const int grid = 256, iters = 2000;
    (void)hipMalloc(&in, 4096 * sizeof(float));
    (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (float)((i * 37) % 101) - 0.04f;
    (void)hipMemcpy(in, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(in); (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / (grid * 4.0) / iters / groups_per_iter;
}

// accuracy of the 6-product split against an fp64 dot product, next to the f32 fmaf chain (host side, same rounding as the device cvt)
static float bf16_rne(float x) {
    unsigned u; __builtin_memcpy(&u, &x, 4);
    const unsigned r = u + 0x7FFFu + ((u >> 16) & 1u);
    const unsigned o = r & 0xFFFF0000u;
    float y; __builtin_memcpy(&y, &o, 4);
    return y;
}

int main() {
    printf("cycles per K=32 product group of ONE accumulator tile (16 x 16 outputs), one wave per SIMD, 256 workgroups\n");
    printf("f32: 8 x v_mfma_f32_16x16x4_f32, 4 chains          : %7.1f\n", run(kb<4, 0, 3>, 4));
    printf("bf16x3: 6 x v_mfma_f32_16x16x32_bf16, 1 chain       : %7.1f\n", run(kb<1, 0, 0>, 1));
    printf("bf16x3: 6 x v_mfma_f32_16x16x32_bf16, 4 chains      : %7.1f\n", run(kb<4, 0, 0>, 4));
    printf("  + 4 / 16 / 64 v_max per 4 groups                  : %7.1f %7.1f %7.1f\n", run(kb<4, 4, 1>, 4), run(kb<4, 16, 1>, 4), run(kb<4, 64, 1>, 4));
    printf("  + split of one 8-value B operand per 4 groups     : %7.1f\n", run(kb<4, 0, 2>, 4));
    printf("  + split of one 8-value B operand per group (1 ch) : %7.1f\n", run(kb<1, 0, 2>, 1));
    // accuracy
    double worst3 = 0, worstf = 0;
    unsigned s = 12345;
    for (int trial = 0; trial < 2000; ++trial) {
        double ref = 0; float f = 0.f, acc3 = 0.f;
        for (int k = 0; k < 64; ++k) {
            s = s * 1664525u + 1013904223u; const float a = ((int)(s >> 8) % 20001 - 10000) * 1e-4f;
            s = s * 1664525u + 1013904223u; const float b = ((int)(s >> 8) % 20001 - 10000) * 1e-4f;
            ref += (double)a * (double)b;
            f = fmaf(a, b, f);
            const float ah = bf16_rne(a), am = bf16_rne(a - ah), al = bf16_rne(a - ah - am);
            const float bh = bf16_rne(b), bm = bf16_rne(b - bh), bl = bf16_rne(b - bh - bm);
            acc3 += ah * bh; acc3 += ah * bm; acc3 += am * bh; acc3 += ah * bl; acc3 += al * bh; acc3 += am * bm;
        }
        worst3 = fmax(worst3, fabs(acc3 - ref)); worstf = fmax(worstf, fabs(f - ref));
    }
    printf("64-term dot products of values in [-1, 1]: max |error| f32 fmaf chain %.3g, bf16x3 six-product form (f32 accumulate) %.3g\n", worstf, worst3);
    return 0;
}
