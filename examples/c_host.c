/*
 * c_host.c - a plain C99 host (no HIP headers, no Python) driving the engine through its C ABI the way libsafecrypto would:
 * the reference-shaped vtable of include/sc_hip_vtable.h for single polynomials, and the batched host-buffer entries of
 * include/sc_hip.h.  It checks the probe vectors SURVEY.md section 8(a) lists for the reference (input a[i] = i):
 *   q = 7681, n = 256:  fwd[0..7] = 3472 1413 1460 7105 2628 1995 3472 7092, fwd[255] = 5877, (a*a)[0..5] = 7405 5745 4343 3201 2321 1705
 *   q = 12289, n = 512: fwd[0..7] = 3563 6129 5350 5063 2188 5583 2185 11079, fwd[511] = 0,   (a*a)[0..5] = 8733 1319 6708 324 6747 1401
 * Build:  gcc -std=c99 -O2 -Iinclude examples/c_host.c -Llibsafecrypto_amd/lib -lsc_hip -Wl,-rpath,$PWD/libsafecrypto_amd/lib -o c_host
 * Exit code 0 = all vectors reproduced on the GPU.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sc_hip.h"
#include "sc_hip_vtable.h"

static int check(const char *what, const int32_t *got, const int32_t *want, int n)
{
    for (int i = 0; i < n; i++)
        if (got[i] != want[i]) { fprintf(stderr, "%s: mismatch at %d: %d != %d\n", what, i, got[i], want[i]); return 1; }
    printf("%s ok\n", what);
    return 0;
}

static int run(int32_t q, int n, const int32_t *fwd8, int32_t fwd_last, const int32_t *sq6)
{
    const sc_hip_ntt32_vtable_t *vt = sc_hip_utils_arith_ntt32();
    if (!vt) { fprintf(stderr, "no engine: %s\n", sc_hip_last_error()); return 1; }
    sc_hip_ntt_params_t p;
    memset(&p, 0, sizeof p);
    p.n = (size_t)n; p.u.ntt32.q = q; p.q_dbl = q; p.inv_q_dbl = 1.0 / q;
    int32_t *a = malloc(sizeof(int32_t) * n), *f = malloc(sizeof(int32_t) * n), *c = malloc(sizeof(int32_t) * n);
    char name[64];
    int rc = 0;
    for (int i = 0; i < n; i++) a[i] = i;
    /* exactly the call sequence of a scheme: sc_ntt->fwd_ntt_32_16(v, &ntt, t, w); normalize_32; mul_32_pointwise; inv_ntt_32_16 */
    vt->fwd_ntt_32_16(f, &p, a, NULL);
    vt->normalize_32(f, (size_t)n, &p);
    snprintf(name, sizeof name, "q=%d n=%d fwd[0..7]", q, n); rc |= check(name, f, fwd8, 8);
    snprintf(name, sizeof name, "q=%d n=%d fwd[n-1]", q, n);  rc |= check(name, f + n - 1, &fwd_last, 1);
    vt->mul_32_pointwise(c, &p, f, f);
    vt->inv_ntt_32_16(c, &p, c, NULL, NULL);
    snprintf(name, sizeof name, "q=%d n=%d (a*a)[0..5]", q, n); rc |= check(name, c, sq6, 6);
    /* the same product through the batched host entry, 64 polynomials at once */
    sc_hip_ctx_t *ctx = NULL; sc_hip_plan_t *plan = NULL;
    if (sc_hip_ctx_create(0, &ctx) || sc_hip_plan_create(ctx, q, n, &plan)) { fprintf(stderr, "%s\n", sc_hip_last_error()); return 1; }
    const size_t batch = 64;
    int32_t *A = malloc(sizeof(int32_t) * n * batch), *Fh = malloc(sizeof(int32_t) * n * batch), *C = malloc(sizeof(int32_t) * n * batch);
    for (size_t b = 0; b < batch; b++) memcpy(A + b * n, a, sizeof(int32_t) * n);
    if (sc_hip_ntt_fwd_host(plan, batch, A, Fh) || sc_hip_polymul_ntt_host(plan, batch, A, Fh, C)) { fprintf(stderr, "%s\n", sc_hip_last_error()); return 1; }
    snprintf(name, sizeof name, "q=%d n=%d batched (a*a)[0..5], polynomial 63", q, n); rc |= check(name, C + 63 * n, sq6, 6);
    rc |= sc_hip_vtable_errors() != 0;
    sc_hip_plan_destroy(plan); sc_hip_ctx_destroy(ctx);
    free(a); free(f); free(c); free(A); free(Fh); free(C);
    return rc;
}

int main(void)
{
    static const int32_t f7681[8] = { 3472, 1413, 1460, 7105, 2628, 1995, 3472, 7092 }, s7681[6] = { 7405, 5745, 4343, 3201, 2321, 1705 };
    static const int32_t f12289[8] = { 3563, 6129, 5350, 5063, 2188, 5583, 2185, 11079 }, s12289[6] = { 8733, 1319, 6708, 324, 6747, 1401 };
    int rc = run(7681, 256, f7681, 5877, s7681);
    rc |= run(12289, 512, f12289, 0, s12289);
    printf(rc ? "FAILED\n" : "all reference probe vectors reproduced through the C ABI\n");
    return rc;
}
