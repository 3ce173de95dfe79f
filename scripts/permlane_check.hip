// Standalone check of the gfx950 permlane-swap helpers mlp.h builds on (xor-16 / xor-32 exchange, 4-lane sum, argmax, gather)
// against __shfl_xor.  Build: hipcc --offload-arch=gfx950 -O2 -I codebase_amd/csrc scripts/permlane_check.hip -o scripts/_bin/permlane_check
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "mlp.h"

using namespace marl;

__global__ void k(const float* in, float* out, int* iout) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[lane] = xor16_other(v, lane);
    out[64 + lane] = xor32_other(v, lane);
    out[128 + lane] = sum_g(v);
    out[192 + lane] = __shfl_xor(v, 16);
    out[256 + lane] = __shfl_xor(v, 32);
    f4 q;
    for (int r = 0; r < 4; ++r) q[r] = in[64 + 4 * lane + r];
    iout[lane] = argmax_rows_pl<6>(q, lane);
    iout[64 + lane] = argmax_rows<6>(q, lane);
    const int a = (lane * 7) % 6;
    out[320 + lane] = gather_rows_pl<6>(q, lane, a);
    out[384 + lane] = gather_rows(q, lane, a);
}

int main() {
    float h[64 + 256], *d, *o;
    int* io;
    srand(3);
    for (int i = 0; i < 320; ++i) h[i] = (float)(rand() % 7) - 3.f;  // small integers: plenty of ties for the argmax rule
    hipMalloc(&d, sizeof h);
    hipMalloc(&o, 448 * sizeof(float));
    hipMalloc(&io, 128 * sizeof(int));
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, io);
    float r[448];
    int ir[128];
    if (hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(ir, io, sizeof ir, hipMemcpyDeviceToHost) != hipSuccess) {
        printf("PERMLANE_CHECK hip error\n");
        return 2;
    }
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        bad += r[l] != r[192 + l];
        bad += r[64 + l] != r[256 + l];
        const int j = l & 15;
        bad += r[128 + l] != (h[j] + h[j + 16]) + (h[j + 32] + h[j + 48]) && r[128 + l] != h[j] + h[j + 16] + h[j + 32] + h[j + 48];
        bad += ir[l] != ir[64 + l];
        bad += r[320 + l] != r[384 + l];
    }
    printf(bad ? "PERMLANE_CHECK FAILED (%d mismatches)\n" : "PERMLANE_CHECK OK\n", bad);
    return bad != 0;
}
