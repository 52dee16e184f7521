/* Stand-alone driver for sanitizer runs of the oracle (make -C oracle sanitize): 64 soft-torso environments, 300 steps of
 * the seeded synthetic actions with auto-reset, state round trip, explicit reset.  Prints a checksum; exits non-zero on NaN. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define REAL double
#include "usim_oracle.h"

int uso_last_info(void* h, double* out);

int main(void) {
    uso_config c;
    uso_default_config(&c);
    c.horizon = 120;                         /* several auto-resets inside the run */
    const int n = 64;
    void* h = uso_create(&c, n);
    const int A = uso_action_dim(h), E = uso_num_elements(h);
    double *act = malloc(sizeof(double) * n * A), *obs = malloc(sizeof(double) * n * USO_OBS_DIM), *rew = malloc(sizeof(double) * n);
    double *term = malloc(sizeof(double) * n * USO_OBS_DIM), *sc = malloc(sizeof(double) * n * USO_NSCALAR), *lat = malloc(sizeof(double) * n * E * 2);
    uint8_t* done = malloc(n);
    int32_t* con = malloc(sizeof(int32_t) * n * (1 + USO_MAXC));
    double info[64 * 8];
    uso_reset(h, NULL, obs);
    double sum = 0; long ndone = 0;
    for (int k = 0; k < 300; k++) {
        uso_random_actions(h, k, act);
        uso_step(h, act, obs, rew, done, term, con, 1);
        uso_last_info(h, info);
        for (int i = 0; i < n; i++) { sum += rew[i]; ndone += done[i]; }
        for (int i = 0; i < n * USO_OBS_DIM; i++) if (!isfinite(obs[i])) { fprintf(stderr, "non-finite observation at step %d\n", k); return 2; }
        if (k == 150) { uso_get_state(h, sc, lat); uso_set_state(h, sc, lat); }
    }
    double params[64 * 13];
    for (int i = 0; i < n; i++) { double p[13] = {0.05, 0.02, 0.8962, -0.05, -0.03, 0.8962, 0.3, 0, 0, 0.002 * (i - 32), 1400, 25, 0.01}; memcpy(params + 13 * i, p, sizeof p); }
    uso_reset_explicit(h, NULL, params, obs);
    printf("selftest ok: reward sum %.6f, episode ends %ld\n", sum, ndone);
    uso_destroy(h);
    free(act); free(obs); free(rew); free(term); free(sc); free(lat); free(done); free(con);
    return 0;
}
