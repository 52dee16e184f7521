// Probe: a kernel that holds `blocks` workgroups (256 threads, a few registers) busy for `usec` microseconds -- a stand-in for the
// resident workgroups of a collective running next to the step kernel (tools/gpu_interference.py).
// hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probe/libspin.so tools/probe/spin.hip
#include <hip/hip_runtime.h>
__global__ void spin_kernel(long long ticks, int* sink) {
    // claim 264 registers per lane (256 VGPRs + 8 AGPRs): the footprint of RCCL's rcclGenericKernel on gfx950 (261-280, read from the
    // code object metadata of the librccl.so that ships with PyTorch)
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a7, v255" ::: "v255", "a7");
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < ticks) acc += 1;
    if (acc == -1) *sink = acc;
}
extern "C" int spin_launch(int blocks, int usec, void* stream, int* sink_dev) {
    // wall_clock64 counts at 100 MHz
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long long)usec * 100, sink_dev);
    return (int)hipGetLastError();
}
