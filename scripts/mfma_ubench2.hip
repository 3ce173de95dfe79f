// What does ONE extra instruction cost next to back-to-back v_mfma_f32_16x16x4_f32 (one wave per SIMD, all CUs busy)?
// Loop body = 16 MFMAs on 4 accumulator chains with K copies of instruction X placed one behind each of the first K MFMAs
// (inline asm, so the placement is exactly this).  cost(X) = (cycles per iteration - 512) / K.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/mfma_ubench2.hip -o scripts/_bin/mfma_ubench2
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

#define M(c) "v_mfma_f32_16x16x4_f32 %[c" #c "], %[a], %[b], %[c" #c "]\n"

#define KERNEL(NAME, X)                                                                                                   \
    template <int K>                                                                                                      \
    __global__ __launch_bounds__(256, 1) void NAME(const float* in, float* out, unsigned long long* cyc, int iters) {     \
        __shared__ f4 lds[1024];                                                                                          \
        const int lane = threadIdx.x & 63;                                                                                \
        for (int i = threadIdx.x; i < 1024; i += 256) lds[i] = f4{in[i & 63], 1.f, 0.5f, 0.25f};                          \
        __syncthreads();                                                                                                  \
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, t0 = c0, t1 = c0;                                                \
        float a = in[lane], b = in[lane + 64], v0 = a, v1 = b, v2 = a + b, v3 = a - b;                                    \
        unsigned addr = (unsigned)(size_t)(&lds[lane]) & 0xffff;                                                         \
        const float* gp = in + lane;                                                                                      \
        const unsigned long long s0 = __builtin_readcyclecounter();                                                       \
        for (int it = 0; it < iters; ++it) {                                                                              \
            _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                               \
                asm volatile(M(0) : [c0] "+a"(c0) : [a] "v"(a), [b] "v"(b));                                              \
                if (4 * g + 0 < K) { X }                                                                                  \
                asm volatile(M(1) : [c1] "+a"(c1) : [a] "v"(a), [b] "v"(b));                                              \
                if (4 * g + 1 < K) { X }                                                                                  \
                asm volatile(M(2) : [c2] "+a"(c2) : [a] "v"(a), [b] "v"(b));                                              \
                if (4 * g + 2 < K) { X }                                                                                  \
                asm volatile(M(3) : [c3] "+a"(c3) : [a] "v"(a), [b] "v"(b));                                              \
                if (4 * g + 3 < K) { X }                                                                                  \
            }                                                                                                             \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                   \
        }                                                                                                                 \
        const unsigned long long s1 = __builtin_readcyclecounter();                                                       \
        out[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2 + v3 + c0[0] + c1[1] + c2[2] + c3[3] + t0[0] + t1[1];          \
        if (lane == 0) atomicAdd(cyc, s1 - s0);                                                                           \
    }

KERNEL(k_none, )
KERNEL(k_fma, asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(v0) : "v"(v1), "v"(v2));)
KERNEL(k_fma_indep, asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(v0) : "v"(v1), "v"(v2), "v"(v3));)
KERNEL(k_max, asm volatile("v_max_f32 %0, %1, %2" : "=v"(v0) : "v"(v1), "v"(v2));)
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(v1));)
KERNEL(k_accread, asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v0) : "a"(t0[0]));)
KERNEL(k_accwrite, asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(t0[0]) : "v"(v1));)
KERNEL(k_cmp, asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(v1), "v"(v2) : "vcc");)
KERNEL(k_salu, asm volatile("s_add_u32 s40, s40, 1" ::: "s40");)
KERNEL(k_nop, asm volatile("s_nop 0");)
KERNEL(k_dsr128, asm volatile("ds_read_b128 %0, %1" : "=v"(t0) : "v"(addr));)
KERNEL(k_dsr128a, asm volatile("ds_read_b128 %0, %1" : "=a"(t1) : "v"(addr));)
KERNEL(k_dsr32, asm volatile("ds_read_b32 %0, %1" : "=v"(v0) : "v"(addr));)
KERNEL(k_dsw32, asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v1) : "memory");)
KERNEL(k_dsw128, asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(t0) : "memory");)
KERNEL(k_gld, asm volatile("global_load_dword %0, %1, off" : "=v"(v0) : "v"(gp));)
KERNEL(k_perm, asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v0), "+v"(v1));)

template <class K>
double run(K kern, int grid) {
    float *in, *out;
    unsigned long long* cyc;
    hipMalloc(&in, 4096 * sizeof(float));
    hipMalloc(&out, (size_t)grid * 256 * sizeof(float));
    hipMalloc(&cyc, 8);
    std::vector<float> h(4096, 0.001f);
    hipMemcpy(in, h.data(), 4096 * sizeof(float), hipMemcpyHostToDevice);
    const int iters = 1000;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, in, out, cyc, iters);
        hipDeviceSynchronize();
    }
    unsigned long long c = 0;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    hipFree(in); hipFree(out); hipFree(cyc);
    return (double)c / (grid * 4.0) / iters;
}

#define ROW(name, kern)                                                                                            \
    {                                                                                                              \
        const double c4 = run(kern<4>, 256), c8 = run(kern<8>, 256), c16 = run(kern<16>, 256);                      \
        printf("%-28s +%6.1f (4 per 16 MFMA)  +%6.1f (8)  +%6.1f (16)   cycles each\n", name, (c4 - base) / 4, (c8 - base) / 8, \
               (c16 - base) / 16);                                                                                 \
    }

int main() {
    const double base = run(k_none<0>, 256);
    printf("16 MFMAs alone: %.1f cycles per iteration (%.2f per MFMA)\n", base, base / 16);
    ROW("v_fma_f32 (dependent)", k_fma)
    ROW("v_fma_f32 (independent)", k_fma_indep)
    ROW("v_max_f32", k_max)
    ROW("v_mov_b32", k_mov)
    ROW("v_accvgpr_read_b32", k_accread)
    ROW("v_accvgpr_write_b32", k_accwrite)
    ROW("v_cmp_gt_f32", k_cmp)
    ROW("s_add_u32", k_salu)
    ROW("s_nop 0", k_nop)
    ROW("ds_read_b128 -> VGPR", k_dsr128)
    ROW("ds_read_b128 -> AGPR", k_dsr128a)
    ROW("ds_read_b32", k_dsr32)
    ROW("ds_write_b32", k_dsw32)
    ROW("ds_write_b128", k_dsw128)
    ROW("global_load_dword", k_gld)
    ROW("v_permlane32_swap", k_perm)
    return 0;
}
