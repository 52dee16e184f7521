/* usim_client.c -- the C ABI of include/usim.h from plain C: no Python, no PyTorch; device memory from the HIP runtime.
 * What a reference maintainer's native binding does for SubprocVecEnv (src/rl.py:130): create, reset, T x (actions -> step), read back.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/usim_client.c -o usim_client \
 *       -L robotic-ultrasound-imaging_amd/lib -lusim -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,robotic-ultrasound-imaging_amd/lib -Wl,-rpath,/opt/rocm/lib
 *   ./usim_client [n_envs] [steps] [seed]
 * Prints one line: n, steps, mean reward per step, episodes ended, and an FNV-1a hash of the last observation block (tests/test_gpu_cabi.py compares the
 * hash with the Python host class driven the same way). */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "usim.h"

#define CHECK_USIM(x) do { int rc_ = (x); if (rc_ != USIM_OK) { fprintf(stderr, "%s: %s\n", #x, usim_strerror(rc_)); return 1; } } while (0)
#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 256, steps = argc > 2 ? atoi(argv[2]) : 100;
    usim_config cfg = { sizeof cfg };
    CHECK_USIM(usim_default_config(&cfg));
    if (argc > 3) cfg.seed = (uint64_t)strtoull(argv[3], NULL, 10);
    usim_handle* h = NULL;
    CHECK_USIM(usim_create(&cfg, n, 0, &h));
    const int A = usim_action_dim(h);

    float *act, *obs, *rew; uint8_t* done;
    CHECK_HIP(hipMalloc((void**)&act, sizeof(float) * n * A));
    CHECK_HIP(hipMalloc((void**)&obs, sizeof(float) * n * USIM_OBS_DIM));
    CHECK_HIP(hipMalloc((void**)&rew, sizeof(float) * n));
    CHECK_HIP(hipMalloc((void**)&done, n));
    float* h_act = (float*)malloc(sizeof(float) * n * A);
    float* h_obs = (float*)malloc(sizeof(float) * n * USIM_OBS_DIM);
    float* h_rew = (float*)malloc(sizeof(float) * n);
    uint8_t* h_done = (uint8_t*)malloc(n);

    usim_step_io io = {0};
    io.act_dev = act; io.obs_dev = obs; io.rew_dev = rew; io.done_dev = done;
    CHECK_USIM(usim_reset(h, NULL, obs, NULL));
    double total = 0.0; long ended = 0;
    for (int t = 0; t < steps; ++t) {
        /* a deterministic action pattern in [-0.5, 0.5] (any policy would go here) */
        for (int i = 0; i < n * A; ++i) h_act[i] = (float)((i * 7 + t * 13) % 21 - 10) * 0.05f;
        CHECK_HIP(hipMemcpy(act, h_act, sizeof(float) * n * A, hipMemcpyHostToDevice));
        CHECK_USIM(usim_step(h, &io, 1, NULL));
        CHECK_HIP(hipMemcpy(h_rew, rew, sizeof(float) * n, hipMemcpyDeviceToHost));
        CHECK_HIP(hipMemcpy(h_done, done, n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) { total += h_rew[i]; ended += h_done[i]; }
    }
    CHECK_HIP(hipMemcpy(h_obs, obs, sizeof(float) * n * USIM_OBS_DIM, hipMemcpyDeviceToHost));
    uint64_t hash = 1469598103934665603ull;
    const unsigned char* b = (const unsigned char*)h_obs;
    for (size_t i = 0; i < sizeof(float) * (size_t)n * USIM_OBS_DIM; ++i) { hash ^= b[i]; hash *= 1099511628211ull; }
    printf("n %d steps %d mean_reward %.6f ended %ld obs_hash %016llx version %s\n", n, steps, total / ((double)n * steps), ended, (unsigned long long)hash, usim_version());
    usim_destroy(h);
    hipFree(act); hipFree(obs); hipFree(rew); hipFree(done);
    free(h_act); free(h_obs); free(h_rew); free(h_done);
    return 0;
}
