// Calibration of the f32 MFMA issue rate the learner kernels can expect with ONE wave per SIMD (256 workgroups x 256 threads, every
// CU busy as in the real kernel): cycles per v_mfma_f32_16x16x4_f32 for 1 / 2 / 4 / 8 interleaved accumulator chains, with the A
// operand held in registers or re-read from LDS every 4 MFMAs (ds_read_b128), and the 32x32x2 form.  s_memtime around the loop.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench.hip -o scripts/_bin/mfma_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int CH, bool LDSOP, int FILL>
__global__ __launch_bounds__(256, 1) void k16(const float* in, float* out, unsigned long long* cyc, int iters) {
    __shared__ f4 lds[64 * 16];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 64 * 16; i += 256) lds[i] = f4{in[i & 63], in[(i + 1) & 63], 0.5f, 0.25f};
    __syncthreads();
    f4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    f4 a[CH];
    for (int c = 0; c < CH; ++c) a[c] = lds[c * 64 + lane];
    float b = in[lane], v0 = in[lane + 64], v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f4 an[CH];
        if (LDSOP) {
#pragma unroll
            for (int c = 0; c < CH; ++c) an[c] = lds[((it + c) & 15) * 64 + lane];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                acc[c] = MFMA(a[c][e], b, acc[c]);
#pragma unroll
                for (int f = 0; f < FILL; ++f) {  // independent VALU fillers
                    v0 = v0 * 1.0001f + v1;
                    v1 = v1 * 0.9999f + v2;
                    v2 = v2 * 1.0002f + v3;
                    v3 = v3 * 0.9998f + v0;
                }
            }
        }
        if (LDSOP) {
#pragma unroll
            for (int c = 0; c < CH; ++c) a[c] = an[c];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1 + v2 + v3;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <int CH>
__global__ __launch_bounds__(256, 1) void k32(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f16v acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const float a = in[lane], b = in[lane + 64];
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int c = 0; c < CH; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <class K>
void run(const char* name, K kern, int mfma_per_iter, double flop_per_mfma, int grid) {
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096 * sizeof(float));
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    std::vector<float> h(4096, 0.001f);
    hipMemcpy(in, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
    }
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double waves = grid * 4.0, n = (double)iters * mfma_per_iter;
    printf("%-44s grid %4d: %7.2f cycles/MFMA  %8.1f us  %7.1f TFLOP/s  (clock %.2f GHz)\n", name, grid, c / waves / n, ms * 1e3,
           waves * n * flop_per_mfma / (ms * 1e-3) / 1e12, c / waves / (ms * 1e-3) / 1e9);
    hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
    for (int grid : {256, 1}) {
        run("16x16x4 1 chain, regs", k16<1, false, 0>, 4, 2048, grid);
        run("16x16x4 2 chains, regs", k16<2, false, 0>, 8, 2048, grid);
        run("16x16x4 4 chains, regs", k16<4, false, 0>, 16, 2048, grid);
        run("16x16x4 8 chains, regs", k16<8, false, 0>, 32, 2048, grid);
        run("16x16x4 4 chains, A from LDS each group", k16<4, true, 0>, 16, 2048, grid);
        run("16x16x4 4 chains, regs, 1x4 VALU/MFMA", k16<4, false, 1>, 16, 2048, grid);
        run("16x16x4 4 chains, LDS, 1x4 VALU/MFMA", k16<4, true, 1>, 16, 2048, grid);
        run("16x16x4 4 chains, regs, 2x4 VALU/MFMA", k16<4, false, 2>, 16, 2048, grid);
        run("32x32x2 1 chain", k32<1>, 4, 4096, grid);
        run("32x32x2 2 chains", k32<2>, 8, 4096, grid);
        run("32x32x2 4 chains", k32<4>, 16, 4096, grid);
    }
    return 0;
}
