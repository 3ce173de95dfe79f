// Bursts: [16 MFMAs on 4 chains][n VALU in a row], n = 4 / 16 / 64 - cycles per VALU when they are NOT interleaved with MFMAs;
// and the same burst reading the accumulators the MFMAs just wrote (drain included).  Also v_pk_mul_f32 / v_pk_add_f32 bursts.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// MODE 0: independent v_max burst; 1: burst = relu of the accumulators (reads MFMA results); 2: v_pk_mul_f32 burst; 3: no burst
template <int N, int MODE>
__global__ __launch_bounds__(256, 1) void kb(const float* in, float* out, unsigned long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = f4{0.f, 0.f, 0.f, 0.f};
    float a = in[lane], b = in[lane + 64];
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = in[lane + i];
    f4 h[4];
    for (int c = 0; c < 4; ++c) h[c] = f4{0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = MFMA(a, b, acc[c]);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i) asm volatile("v_max_f32 %0, %1, %2" : "=v"(v[i & 15]) : "v"(v[(i + 1) & 15]), "v"(v[(i + 5) & 15]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < N; ++i) h[(i >> 2) & 3][i & 3] = __builtin_amdgcn_fmed3f(acc[(i >> 2) & 3][i & 3], 0.f, __builtin_huge_valf());
            b = h[0][0] * 1e-30f + b;  // keep the relu results alive without changing the MFMA inputs much
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < N; ++i) {
                f2 x = {v[(2 * i) & 15], v[(2 * i + 1) & 15]}, y = {v[(2 * i + 3) & 15], v[(2 * i + 7) & 15]};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(y));
                v[(2 * i) & 15] = x[0];
                v[(2 * i + 1) & 15] = x[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3] + h[c][0] + h[c][1] + h[c][2] + h[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <class K>
double run(K kern) {
    float *in, *out;
    unsigned long long* cyc;
    const int grid = 256, iters = 1000;
    (void)hipMalloc(&in, 4096 * sizeof(float));
    (void)hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    (void)hipMalloc(&cyc, 8);
    std::vector<float> h(4096, 0.001f);
    (void)hipMemcpy(in, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
        (void)hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    (void)hipFree(in); (void)hipFree(out); (void)hipFree(cyc);
    return (double)c / (grid * 4.0) / iters;
}

int main() {
    const double base = run(kb<0, 3>);
    printf("16 MFMAs alone: %.1f cycles\n", base);
    printf("independent v_max burst:   n=4 %+.1f/op   n=16 %+.1f/op   n=64 %+.1f/op\n", (run(kb<4, 0>) - base) / 4, (run(kb<16, 0>) - base) / 16,
           (run(kb<64, 0>) - base) / 64);
    printf("relu of the accumulators:  n=4 %+.1f/op   n=16 %+.1f/op\n", (run(kb<4, 1>) - base) / 4, (run(kb<16, 1>) - base) / 16);
    printf("v_pk_mul_f32 burst:        n=4 %+.1f/op   n=16 %+.1f/op   n=64 %+.1f/op  (2 elements per op)\n", (run(kb<4, 2>) - base) / 4,
           (run(kb<16, 2>) - base) / 16, (run(kb<64, 2>) - base) / 64);
    return 0;
}
