/*
 * edt_mt.c — multi-threaded exact separable EDT with closest-obstacle tracking: the timed CPU
 * baseline at full size (SURVEY.md §8(d) "CPU baseline" (ii)).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY, like everything under oracle/: bench.py's `cpu_baseline` leg
 * and tests/ load it; the product path never does.  The reference has no CPU implementation of
 * this path, so this is not a restatement of reference code but the textbook algorithm the
 * reference's batch EDT (src/kernel/edt/local_edt_core.h:14-193) also implements on the GPU:
 * three 1-D passes, the 2nd and 3rd as Meijster lower-envelope scans with integer Sep().  Pass
 * order x → y → z (rows contiguous in memory first); the reference goes y → x → z — the result
 * (exact squared distance to the nearest occupied voxel) does not depend on the order, only the
 * choice among equidistant obstacles does, and this baseline is validated on distances
 * (tests/test_oracle_edt.py::test_edt_mt_equals_brute_force) plus the witness property of its
 * closest obstacle.
 *
 *   int go_edt_mt(const int8_t *type, int X, int Y, int Z, int nthreads, int32_t *dist_sq, int32_t *coc_packed)
 *
 * type: N = X*Y*Z voxel types, x fastest, 2 = OCCUPIED.  dist_sq[N]: squared distance in voxels,
 * GO_EDT_MT_NONE when the volume holds no obstacle.  coc_packed[N] (may be NULL): closest obstacle
 * x | y << 10 | z << 20 (sides <= 1024), -1 when none.  Work is split over `nthreads` POSIX
 * threads: z-planes for passes x and y, y-rows for pass z.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define GO_EDT_MT_NONE 0x3fffffff
#define OCCUPIED 2

typedef struct {
    const int8_t *type;
    int X, Y, Z;
    int16_t *cx;        /* pass x: nearest occupied x' in the row, -1 none */
    uint32_t *cxy;      /* pass y: cx | cy << 16, 0xffffffff none */
    int32_t *dist;
    int32_t *coc;
    int pass, tid, nth;
} job_t;

/* lower envelope over the compacted sites (pos[k], val[k]), k < K: for every u in [0,n) the site
 * minimising (u-pos)² + val; writes the winning site index into win[u].  Meijster et al. 2000,
 * with sites that carry no obstacle left out up front. */
static void envelope(const int *pos, const int *val, int K, int n, int *s, int *t, int *win)
{
    int q = 0;
    s[0] = 0; t[0] = 0;
    for (int k = 1; k < K; k++) {
        while (q >= 0) {
            const long long a = (long long)(t[q] - pos[s[q]]) * (t[q] - pos[s[q]]) + val[s[q]];
            const long long b = (long long)(t[q] - pos[k]) * (t[q] - pos[k]) + val[k];
            if (a > b) q--; else break;
        }
        if (q < 0) { q = 0; s[0] = k; t[0] = 0; }
        else {
            const long long i = pos[s[q]], u = pos[k];
            const long long w = 1 + (u * u - i * i + (long long)val[k] - (long long)val[s[q]]) / (2 * (u - i));
            if (w < n) { q++; s[q] = k; t[q] = (int)(w < 0 ? 0 : w); }
        }
    }
    for (int u = n - 1; u >= 0; u--) {
        win[u] = s[q];
        if (u == t[q] && q > 0) q--;
    }
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    const int X = j->X, Y = j->Y, Z = j->Z;
    const size_t plane = (size_t)X * Y;
    const int L = (X > Y ? (X > Z ? X : Z) : (Y > Z ? Y : Z)) + 1;
    int *pos = (int *)malloc(sizeof(int) * L * 5), *val = pos + L, *s = val + L, *t = s + L, *win = t + L;
    if (j->pass == 0) {              /* along x: nearest occupied voxel of the row (two sweeps) */
        for (int z = j->tid; z < Z; z += j->nth) for (int y = 0; y < Y; y++) {
            const int8_t *ty = j->type + z * plane + (size_t)y * X;
            int16_t *o = j->cx + z * plane + (size_t)y * X;
            int last = -1;
            for (int x = 0; x < X; x++) { if (ty[x] == OCCUPIED) last = x; o[x] = (int16_t)last; }
            last = -1;
            for (int x = X - 1; x >= 0; x--) {
                if (ty[x] == OCCUPIED) last = x;
                if (last >= 0 && (o[x] < 0 || last - x < x - o[x])) o[x] = (int16_t)last;
            }
        }
    } else if (j->pass == 1) {       /* along y inside a plane */
        for (int z = j->tid; z < Z; z += j->nth) for (int x = 0; x < X; x++) {
            const int16_t *c = j->cx + z * plane + x;
            uint32_t *o = j->cxy + z * plane + x;
            int K = 0;
            for (int y = 0; y < Y; y++) { const int v = c[(size_t)y * X]; if (v >= 0) { pos[K] = y; val[K] = (x - v) * (x - v); K++; } }
            if (K == 0) { for (int y = 0; y < Y; y++) o[(size_t)y * X] = 0xffffffffu; continue; }
            envelope(pos, val, K, Y, s, t, win);
            for (int y = 0; y < Y; y++) { const int sy = pos[win[y]]; o[(size_t)y * X] = (uint32_t)(uint16_t)c[(size_t)sy * X] | ((uint32_t)sy << 16); }
        }
    } else {                         /* along z: one y-row of columns at a time */
        for (int y = j->tid; y < Y; y += j->nth) for (int x = 0; x < X; x++) {
            const uint32_t *c = j->cxy + (size_t)y * X + x;
            int K = 0;
            for (int z = 0; z < Z; z++) {
                const uint32_t v = c[z * plane];
                if (v != 0xffffffffu) { const int dx = x - (int)(v & 0xffffu), dy = y - (int)(v >> 16); pos[K] = z; val[K] = dx * dx + dy * dy; K++; }
            }
            int32_t *d = j->dist + (size_t)y * X + x;
            int32_t *co = j->coc ? j->coc + (size_t)y * X + x : NULL;
            if (K == 0) { for (int z = 0; z < Z; z++) { d[z * plane] = GO_EDT_MT_NONE; if (co) co[z * plane] = -1; } continue; }
            envelope(pos, val, K, Z, s, t, win);
            for (int z = 0; z < Z; z++) {
                const int k = win[z], sz = pos[k];
                d[z * plane] = (z - sz) * (z - sz) + val[k];
                if (co) { const uint32_t v = c[sz * plane]; co[z * plane] = (int32_t)((v & 0xffffu) | ((v >> 16) << 10) | ((uint32_t)sz << 20)); }
            }
        }
    }
    free(pos);
    return NULL;
}

int go_edt_mt(const int8_t *type, int X, int Y, int Z, int nthreads, int32_t *dist_sq, int32_t *coc_packed)
{
    if (!type || !dist_sq || X < 1 || Y < 1 || Z < 1 || X > 1024 || Y > 1024 || Z > 1024) return 1;
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 1024) nthreads = 1024;
    const size_t N = (size_t)X * Y * Z;
    int16_t *cx = (int16_t *)malloc(N * sizeof(int16_t));
    uint32_t *cxy = (uint32_t *)malloc(N * sizeof(uint32_t));
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * nthreads);
    if (!cx || !cxy || !th || !jobs) { free(cx); free(cxy); free(th); free(jobs); return 2; }
    for (int pass = 0; pass < 3; pass++) {
        for (int i = 0; i < nthreads; i++) {
            job_t jb = { type, X, Y, Z, cx, cxy, dist_sq, coc_packed, pass, i, nthreads };
            jobs[i] = jb;
            if (pthread_create(&th[i], NULL, worker, &jobs[i]) != 0) { worker(&jobs[i]); th[i] = 0; }
        }
        for (int i = 0; i < nthreads; i++) if (th[i]) pthread_join(th[i], NULL);
    }
    free(cx); free(cxy); free(th); free(jobs);
    return 0;
}
