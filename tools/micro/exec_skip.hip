// Does a wave64 VALU instruction on gfx950 take fewer cycles when whole 16-lane quarters of EXEC are off?  (It does not on GCN; measured here because the
// contact sweeps keep one lane in sixteen busy.)  Build: hipcc -O3 --offload-arch=gfx950 -o exec_skip exec_skip.hip.  One wave per launch, a chain of dependent v_fma_f32, shader-clock ticks per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void chain(float* out, unsigned long long* ticks, unsigned long long mask, int iters) {
    const int lane = threadIdx.x & 63;
    float x = out[lane], y = 1.0001f, z = 0.5f;
    unsigned long long t0 = 0, t1 = 0;
    if ((mask >> lane) & 1ull) {
        t0 = __builtin_readcyclecounter();
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 64; ++k) x = fmaf(x, y, z);
        }
        t1 = __builtin_readcyclecounter();
    }
    out[lane] = x;
    if (lane == __ffsll((long long)mask) - 1) ticks[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 64 * sizeof(float)); (void)hipMemset(out, 0, 64 * sizeof(float));
    (void)hipMalloc(&ticks, 8);
    const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0x1ull, 0x0001000100010001ull, 0xffull};
    const char* names[] = {"all 64 lanes", "lanes 0-31", "lanes 0-15 (one quarter)", "lane 0", "lanes 0,16,32,48", "lanes 0-7"};
    for (int m = 0; m < 6; ++m) {
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, ticks, masks[m], 1000);
        (void)hipDeviceSynchronize();
        unsigned long long t; (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        printf("%-28s %.3f ticks per dependent v_fma_f32\n", names[m], (double)t / 64000.0);
    }
    return 0;
}
