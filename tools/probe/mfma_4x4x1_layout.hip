// Probe: operand/result lane layout of v_mfma_f32_4x4x1_16b_f32 on gfx950 (used by the lattice solve of usim_kernels.hip).
// hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_probe tools/probe/mfma_4x4x1_layout.hip && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(float* outA, float* outB) {
    const int l = threadIdx.x;
    v4f z = {0.f, 0.f, 0.f, 0.f};
    v4f da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), 1.0f, z, 0, 0, 0);     // D = index of the A lane used
    v4f db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)(l + 1), z, 0, 0, 0);     // D = index of the B lane used
    for (int r = 0; r < 4; ++r) { outA[l * 4 + r] = da[r]; outB[l * 4 + r] = db[r]; }
}
int main() {
    float *a, *b; hipMalloc(&a, 256 * 4); hipMalloc(&b, 256 * 4);
    probe<<<1, 64>>>(a, b);
    float ha[256], hb[256]; hipMemcpy(ha, a, 1024, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        int la = (int)ha[l * 4 + r] - 1, lb = (int)hb[l * 4 + r] - 1;
        // expectation: block = l / 4; D register r of lane l = A[block][i = r] * B[block][j = l % 4]
        if (la != (l / 4) * 4 + r || lb != l) ok = 0;
        if (l < 8) printf("lane %2d reg %d: A from lane %2d, B from lane %2d\n", l, r, la, lb);
    }
    printf("layout D[lane l][reg r] = A[lane 4*(l/4)+r] * B[lane l]: %s\n", ok ? "CONFIRMED" : "DIFFERENT");
    return ok ? 0 : 1;
}
