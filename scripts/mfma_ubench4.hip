// Does a SECOND wave on the same SIMD hide the VALU cost that one wave pays next to f32 MFMAs (mfma_ubench2: +9..13 cycles per VALU
// instruction)?  256 workgroups of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD, every wave running the same
// 16-MFMA + FILL x 16 x 4 VALU loop; and a role split: even waves only MFMAs, odd waves only VALU.
// Reported: SIMD cycles per MFMA (wave cycles / MFMAs per wave / waves per SIMD) and TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench4.hip -o scripts/_bin/mfma_ubench4
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ROLE 0: every wave MFMA + FILL VALU groups; ROLE 1: waves with (wave / 4) even run MFMAs only, odd run the VALU part only
template <int FILL, int ROLE>
__global__ __launch_bounds__(1024) void k16(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f4 acc[4], a[4];
    for (int c = 0; c < 4; ++c) {
        acc[c] = f4{0.f, 0.f, 0.f, 0.f};
        a[c] = f4{in[lane + c], in[lane + c + 1], 0.5f, 0.25f};
    }
    float b = in[lane], v0 = in[lane + 64], v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    const bool do_mfma = ROLE == 0 || ((wave >> 2) & 1) == 0, do_valu = ROLE == 0 || ((wave >> 2) & 1) == 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[c] = MFMA(a[c][e], b, acc[c]);
#pragma unroll
                    for (int f = 0; f < FILL; ++f) {
                        v0 = v0 * 1.0001f + v1;
                        v1 = v1 * 0.9999f + v2;
                        v2 = v2 * 1.0002f + v3;
                        v3 = v3 * 0.9998f + v0;
                    }
                }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[c] = MFMA(a[c][e], b, acc[c]);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int f = 0; f < FILL; ++f) {
                    v0 = v0 * 1.0001f + v1;
                    v1 = v1 * 0.9999f + v2;
                    v2 = v2 * 1.0002f + v3;
                    v3 = v3 * 0.9998f + v0;
                }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2 + v3;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 1024 + threadIdx.x] = s;
    if (lane == 0 && do_mfma) atomicAdd(cyc, t1 - t0);
    if (lane == 0 && !do_mfma) atomicAdd(cyc + 1, t1 - t0);
}

template <class K>
void run(const char* name, K kern, int block, int role) {
    const int grid = 256;
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096 * sizeof(float));
    hipMalloc(&out, (size_t)grid * 1024 * sizeof(float));
    hipMalloc(&cyc, 16);
    std::vector<float> h(4096, 0.001f);
    hipMemcpy(in, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 16);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(block), 0, 0, in, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2] = {0, 0};
    hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const int wps = block / 256;                              // waves per SIMD
    const double mw = role ? grid * (block / 64) / 2.0 : grid * (block / 64.0);  // waves issuing MFMAs
    const double mw_per_simd = role ? wps / 2.0 : wps;
    const double n = (double)iters * 16;
    printf("%-52s %d waves/SIMD: %7.2f SIMD cycles/MFMA  %8.1f us  %6.1f TFLOP/s", name, wps, c[0] / mw / n / mw_per_simd, ms * 1e3,
           mw * n * 2048 / (ms * 1e-3) / 1e12);
    if (role) printf("   (VALU-only waves: %.2f cycles per VALU op)", c[1] / mw / ((double)iters * 16 * 4 * (name[0] - '0')));
    printf("\n");
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
    for (int block : {256, 512, 1024}) {
        run("0x4 VALU per MFMA, all waves alike", k16<0, 0>, block, 0);
        run("1x4 VALU per MFMA, all waves alike", k16<1, 0>, block, 0);
        run("2x4 VALU per MFMA, all waves alike", k16<2, 0>, block, 0);
    }
    for (int block : {512, 1024}) {
        run("1x4 VALU per MFMA, MFMA waves / VALU waves split", k16<1, 1>, block, 1);
        run("2x4 VALU per MFMA, MFMA waves / VALU waves split", k16<2, 1>, block, 1);
    }
    return 0;
}
