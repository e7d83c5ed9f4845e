/*
 * oracle/sot_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-threaded, op-for-op restatement of the reference's 1-D
 * spectral optimal-transport loss (reference: losses.py:129-313, utils.py:135-142).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared library; the product (HIP) path never does.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors, so
 * this restatement is pinned against outputs of the reference itself, imported
 * in the build container by oracle/make_golden.py (fixtures under
 * tests/golden/, checked by tests/test_oracle_golden.py): every intermediate
 * (row mass, normalised weights, CDFs, merged levels, ranks, quantiles, row
 * loss) is compared bit-for-bit.
 *
 * Each function cites the reference lines it restates.  The arithmetic of the
 * reference lives in PyTorch ATen CPU kernels (third-party; reference pins
 * pytorch=1.13.1 in environment.yml:222, the container has torch 2.10.0), so
 * three ATen behaviours are restated from their published algorithm and
 * verified bit-exact against torch in the build container:
 *   - sum(dim) of fp32: SumKernel.cpp cascade_sum/vectorized_inner_sum,
 *     8-lane vectors x 4 ILP accumulators (32 interleaved columns), 4 cascade
 *     levels of 16 steps (the AVX2 kernel, which ATen also dispatches on
 *     AVX-512 hosts);
 *   - cumsum of fp32 accumulates in double, each prefix rounded to fp32;
 *   - searchsorted default side='left'.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SOT_FLAG_SQUARE 1u         /* Wasserstein1D(square_dist=True)      losses.py:172-174 */
#define SOT_FLAG_DONT_NORMALIZE 2u /* dont_normalize (ctor or call kwarg)  losses.py:180-184 */
#define SOT_FLAG_LIMIT_Q 4u        /* limit_quantile_range                 losses.py:306-307 */
#define SOT_FLAG_REQUIRE_SORT 8u   /* require_sort                         losses.py:286-290 */

/* ------------------------------------------------------------------------- */
/* ATen fp32 sum over a contiguous row (what torch.sum(x, dim=1) computes for
 * losses.py:177,184 and torch.sum(delta*diff, 1) for losses.py:312-313).      */

enum { NLEV = 4, ILP = 4, VLEN = 8 };

/* multi_row_sum: `ncol` interleaved columns, `size` steps, element (i,c) at
 * in[i*ncol + c]; result in out[ncol]. */
static void multi_row_sum(const float *in, int64_t size, int ncol, float *out)
{
    float acc[NLEV][ILP * VLEN];
    int64_t ceil_log2 = 0;
    while (((int64_t)1 << ceil_log2) < size) ceil_log2++;
    int64_t level_power = ceil_log2 / NLEV;
    if (level_power < 4) level_power = 4;
    const int64_t level_step = (int64_t)1 << level_power;
    const int64_t level_mask = level_step - 1;
    for (int j = 0; j < NLEV; j++)
        for (int c = 0; c < ncol; c++) acc[j][c] = 0.0f;

    int64_t i = 0;
    while (i + level_step <= size) {
        for (int64_t s = 0; s < level_step; s++, i++)
            for (int c = 0; c < ncol; c++) acc[0][c] += in[i * ncol + c];
        for (int j = 1; j < NLEV; j++) {
            for (int c = 0; c < ncol; c++) {
                acc[j][c] += acc[j - 1][c];
                acc[j - 1][c] = 0.0f;
            }
            const int64_t mask = level_mask << (j * level_power);
            if ((i & mask) != 0) break;
        }
    }
    for (; i < size; i++)
        for (int c = 0; c < ncol; c++) acc[0][c] += in[i * ncol + c];
    for (int j = 1; j < NLEV; j++)
        for (int c = 0; c < ncol; c++) acc[0][c] += acc[j][c];
    for (int c = 0; c < ncol; c++) out[c] = acc[0][c];
}

float sot_oracle_aten_sum_f32(const float *x, int64_t n)
{
    float part[ILP * VLEN];
    if (n >= VLEN) { /* vectorized_inner_sum */
        const int64_t vec_size = n / VLEN;
        const int64_t size_ilp = vec_size / ILP;
        multi_row_sum(x, size_ilp, ILP * VLEN, part);
        for (int64_t v = size_ilp * ILP; v < vec_size; v++)
            for (int l = 0; l < VLEN; l++) part[l] += x[v * VLEN + l];
        for (int k = 1; k < ILP; k++)
            for (int l = 0; l < VLEN; l++) part[l] += part[k * VLEN + l];
        float fin = 0.0f;
        for (int64_t k = vec_size * VLEN; k < n; k++) fin += x[k];
        for (int l = 0; l < VLEN; l++) fin += part[l];
        return fin;
    }
    /* scalar_inner_sum */
    const int64_t size_ilp = n / ILP;
    multi_row_sum(x, size_ilp, ILP, part);
    for (int64_t i = size_ilp * ILP; i < n; i++) part[0] += x[i];
    for (int k = 1; k < ILP; k++) part[0] += part[k];
    return part[0];
}

/* utils.py:135-142  safe_divide: den <= 1e-7 -> 1e-7 (the eps is a float32 tensor) */
static float safe_den(float s) { return (s <= 1e-7f) ? 1e-7f : s; }

/* ------------------------------------------------------------------------- */
typedef struct { float key; int64_t idx; } kv_t;

static int cmp_kv(const void *a, const void *b)
{
    const kv_t *p = (const kv_t *)a, *q = (const kv_t *)b;
    if (p->key < q->key) return -1;
    if (p->key > q->key) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx); /* lowest index first: stable order */
}
static int cmp_f(const void *a, const void *b)
{
    const float p = *(const float *)a, q = *(const float *)b;
    return (p > q) - (p < q);
}

/* torch.sort(keys, 1) -> (values, indices); ties resolved lowest-index-first
 * (torch's default CPU sort is not stable; on distinct keys the permutation is
 * unique, which is every position grid the reference passes: SURVEY B.3). */
void sot_oracle_sort_rows(const float *keys, int64_t B, int n, int64_t row_stride,
                          float *values, int64_t *indices)
{
    kv_t *tmp = (kv_t *)malloc(sizeof(kv_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < B; r++) {
        const float *k = keys + r * row_stride;
        for (int i = 0; i < n; i++) { tmp[i].key = k[i]; tmp[i].idx = i; }
        qsort(tmp, (size_t)n, sizeof(kv_t), cmp_kv);
        for (int i = 0; i < n; i++) {
            if (values) values[r * (int64_t)n + i] = tmp[i].key;
            if (indices) indices[r * (int64_t)n + i] = tmp[i].idx;
        }
    }
    free(tmp);
}

/* torch.searchsorted(cws, q) with side='left': first i with cws[i] >= q. losses.py:219 */
static int64_t searchsorted_left(const float *cws, int64_t n, float q)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = lo + (hi - lo) / 2;
        if (cws[mid] < q) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* Debug/inspection outputs; every pointer may be NULL. Layout is row-major. */
typedef struct {
    float *mass;        /* [B,2]  S_x, S_y (S_y = S_x in dont_normalize mode) */
    float *a, *b;       /* [B,n], [B,m] normalised (and position-sorted) weights */
    float *U, *V;       /* [B,n], [B,m] CDFs                                   */
    float *Q;           /* [B,n+m] merged quantile levels                       */
    int64_t *iu, *iv;   /* [B,n+m] clamped left ranks                           */
    float *uq, *vq;     /* [B,n+m] quantile-function values                     */
    int64_t *xsorter, *ysorter; /* [B,n], [B,m] position sort permutations       */
} sot_oracle_debug_t;

/*
 * Forward, restating Wasserstein1D.forward steps (3)-(5) (losses.py:172-196)
 * and wasserstein_1d (losses.py:271-313), one row at a time.
 *   x [B,n], y [B,m] row-major contiguous; xpos/ypos with row stride
 *   xpos_stride/ypos_stride in elements (0 = one shared row: the stride-0
 *   expand of losses.py:167-170).
 * Writes row_loss[B] = W_p^p per row (no p-th root, losses.py:309-313).
 */
int sot_oracle_forward(const float *x, const float *y, const float *xpos, const float *ypos,
                       int64_t B, int n, int m, int64_t xpos_stride, int64_t ypos_stride,
                       float p, uint32_t flags, float *row_loss, const sot_oracle_debug_t *dbg)
{
    if (!(p >= 1.0f)) return 1; /* losses.py:271 assert p >= 1 */
    if (n <= 0 || m <= 0) return 2;
    const int K = n + m;
    /* Rows are independent (losses.py:273-313 has no cross-row operation): with -fopenmp (oracle/Makefile) they are spread over the
     * host's cores, every thread with its own scratch; without it the pragmas are ignored.  Results do not depend on the thread count. */
#pragma omp parallel
    {
    float *a = (float *)malloc(sizeof(float) * (size_t)n), *b = (float *)malloc(sizeof(float) * (size_t)m);
    float *xs = (float *)malloc(sizeof(float) * (size_t)n), *ys = (float *)malloc(sizeof(float) * (size_t)m);
    float *U = (float *)malloc(sizeof(float) * (size_t)n), *V = (float *)malloc(sizeof(float) * (size_t)m);
    float *Q = (float *)malloc(sizeof(float) * (size_t)K), *term = (float *)malloc(sizeof(float) * (size_t)K);
    float *tmpw = (float *)malloc(sizeof(float) * (size_t)(n > m ? n : m));
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)(n > m ? n : m));

#pragma omp for schedule(static)
    for (int64_t r = 0; r < B; r++) {
        const float *xr = x + r * (int64_t)n, *yr = y + r * (int64_t)m;
        const float *xp = xpos + r * xpos_stride, *yp = ypos + r * ypos_stride;
        /* losses.py:172-174 */
        for (int i = 0; i < n; i++) a[i] = (flags & SOT_FLAG_SQUARE) ? xr[i] * xr[i] : xr[i];
        for (int j = 0; j < m; j++) b[j] = (flags & SOT_FLAG_SQUARE) ? yr[j] * yr[j] : yr[j];
        /* losses.py:177-184 */
        const float Sx = sot_oracle_aten_sum_f32(a, n);
        const float Sy = (flags & SOT_FLAG_DONT_NORMALIZE) ? Sx : sot_oracle_aten_sum_f32(b, m);
        const float dx = safe_den(Sx), dy = safe_den(Sy);
        for (int i = 0; i < n; i++) a[i] = a[i] / dx;
        for (int j = 0; j < m; j++) b[j] = b[j] / dy;
        if (dbg && dbg->mass) { dbg->mass[2 * r] = Sx; dbg->mass[2 * r + 1] = Sy; }

        /* losses.py:286-290: sort positions, gather weights */
        if (flags & SOT_FLAG_REQUIRE_SORT) {
            for (int i = 0; i < n; i++) { kv[i].key = xp[i]; kv[i].idx = i; }
            qsort(kv, (size_t)n, sizeof(kv_t), cmp_kv);
            for (int i = 0; i < n; i++) { xs[i] = kv[i].key; tmpw[i] = a[kv[i].idx];
                if (dbg && dbg->xsorter) dbg->xsorter[r * (int64_t)n + i] = kv[i].idx; }
            memcpy(a, tmpw, sizeof(float) * (size_t)n);
            for (int j = 0; j < m; j++) { kv[j].key = yp[j]; kv[j].idx = j; }
            qsort(kv, (size_t)m, sizeof(kv_t), cmp_kv);
            for (int j = 0; j < m; j++) { ys[j] = kv[j].key; tmpw[j] = b[kv[j].idx];
                if (dbg && dbg->ysorter) dbg->ysorter[r * (int64_t)m + j] = kv[j].idx; }
            memcpy(b, tmpw, sizeof(float) * (size_t)m);
        } else {
            memcpy(xs, xp, sizeof(float) * (size_t)n);
            memcpy(ys, yp, sizeof(float) * (size_t)m);
        }
        /* losses.py:292-293: cumsum (double accumulator, fp32 outputs) */
        double acc = 0.0;
        for (int i = 0; i < n; i++) { acc += (double)a[i]; U[i] = (float)acc; }
        acc = 0.0;
        for (int j = 0; j < m; j++) { acc += (double)b[j]; V[j] = (float)acc; }
        /* losses.py:295: qs = sort(cat(U, V)) (values only) */
        memcpy(Q, U, sizeof(float) * (size_t)n);
        memcpy(Q + n, V, sizeof(float) * (size_t)m);
        qsort(Q, (size_t)K, sizeof(float), cmp_f);
        /* losses.py:297-313 */
        for (int k = 0; k < K; k++) {
            int64_t iu = searchsorted_left(U, n, Q[k]);   /* losses.py:219 */
            int64_t iv = searchsorted_left(V, m, Q[k]);
            if (iu > n - 1) iu = n - 1;                   /* clamp, losses.py:220 */
            if (iv > m - 1) iv = m - 1;
            const float uq = xs[iu], vq = ys[iv];
            float delta = Q[k] - (k ? Q[k - 1] : 0.0f);   /* losses.py:301-304 */
            if ((flags & SOT_FLAG_LIMIT_Q) && Q[k] > 1.0f) delta = 0.0f; /* :306-307 */
            float d = fabsf(uq - vq);                     /* :309 */
            if (p == 2.0f) d = d * d;                     /* torch pow(2) is x*x */
            else if (p != 1.0f) d = powf(d, p);           /* :313 */
            term[k] = delta * d;
            if (dbg) {
                const int64_t o = r * (int64_t)K + k;
                if (dbg->iu) dbg->iu[o] = iu;
                if (dbg->iv) dbg->iv[o] = iv;
                if (dbg->uq) dbg->uq[o] = uq;
                if (dbg->vq) dbg->vq[o] = vq;
            }
        }
        row_loss[r] = sot_oracle_aten_sum_f32(term, K);   /* torch.sum(.., 1) */
        if (dbg) {
            if (dbg->a) memcpy(dbg->a + r * (int64_t)n, a, sizeof(float) * (size_t)n);
            if (dbg->b) memcpy(dbg->b + r * (int64_t)m, b, sizeof(float) * (size_t)m);
            if (dbg->U) memcpy(dbg->U + r * (int64_t)n, U, sizeof(float) * (size_t)n);
            if (dbg->V) memcpy(dbg->V + r * (int64_t)m, V, sizeof(float) * (size_t)m);
            if (dbg->Q) memcpy(dbg->Q + r * (int64_t)K, Q, sizeof(float) * (size_t)K);
        }
    }
    free(a); free(b); free(xs); free(ys); free(U); free(V); free(Q); free(term); free(tmpw); free(kv);
    }   /* omp parallel */
    return 0;
}

/* torch.mean(loss) over all rows (losses.py:211, dims=None): ATen sum / numel.
 * (Exact for B below ATen's 32768-element parallel grain; above it the
 * reference's own result depends on its thread count.) */
float sot_oracle_mean(const float *row_loss, int64_t B)
{
    return sot_oracle_aten_sum_f32(row_loss, B) / (float)B;
}

/*
 * Backward: the closed form of the autograd graph of losses.py:172-313
 * (SURVEY Appendix A.4), evaluated in double so that it can serve as a
 * tight checker for fp32 kernels.  grad_row[B] is dL/d(row_loss).
 * gx [B,n] and/or gy [B,m] may be NULL.  Requires sorted positions or
 * REQUIRE_SORT (the permutation is undone on output).
 * Tie convention: among equal merged levels, U-elements precede V-elements and
 * lower indices come first (a stable sort of cat(U,V)); the reference's default
 * sort is unstable, so on rows with tied levels it may route the run's gradient
 * to a different member of the run (a different valid subgradient at a kink).
 */
int sot_oracle_backward(const float *x, const float *y, const float *xpos, const float *ypos,
                        int64_t B, int n, int m, int64_t xpos_stride, int64_t ypos_stride,
                        float p, uint32_t flags, const float *grad_row, float *gx, float *gy)
{
    if (!(p >= 1.0f)) return 1;
    const int K = n + m;
    const int nm = n > m ? n : m;
#pragma omp parallel
    {
    float *a = (float *)malloc(sizeof(float) * (size_t)n), *b = (float *)malloc(sizeof(float) * (size_t)m);
    float *w = (float *)malloc(sizeof(float) * (size_t)nm);
    float *xs = (float *)malloc(sizeof(float) * (size_t)n), *ys = (float *)malloc(sizeof(float) * (size_t)m);
    float *U = (float *)malloc(sizeof(float) * (size_t)n), *V = (float *)malloc(sizeof(float) * (size_t)m);
    int64_t *px = (int64_t *)malloc(sizeof(int64_t) * (size_t)n), *py = (int64_t *)malloc(sizeof(int64_t) * (size_t)m);
    double *gU = (double *)malloc(sizeof(double) * (size_t)n), *gV = (double *)malloc(sizeof(double) * (size_t)m);
    double *dk = (double *)malloc(sizeof(double) * (size_t)(K + 1));
    int *src = (int *)malloc(sizeof(int) * (size_t)K);
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)nm);

#pragma omp for schedule(static)
    for (int64_t r = 0; r < B; r++) {
        const float *xr = x + r * (int64_t)n, *yr = y + r * (int64_t)m;
        const float *xp = xpos + r * xpos_stride, *yp = ypos + r * ypos_stride;
        const int sq = (flags & SOT_FLAG_SQUARE) != 0, dn = (flags & SOT_FLAG_DONT_NORMALIZE) != 0;
        for (int i = 0; i < n; i++) a[i] = sq ? xr[i] * xr[i] : xr[i];
        for (int j = 0; j < m; j++) b[j] = sq ? yr[j] * yr[j] : yr[j];
        const float Sx = sot_oracle_aten_sum_f32(a, n);
        const float Sy = dn ? Sx : sot_oracle_aten_sum_f32(b, m);
        const float dx = safe_den(Sx), dy = safe_den(Sy);
        for (int i = 0; i < n; i++) { px[i] = i; xs[i] = xp[i]; }
        for (int j = 0; j < m; j++) { py[j] = j; ys[j] = yp[j]; }
        if (flags & SOT_FLAG_REQUIRE_SORT) {
            for (int i = 0; i < n; i++) { kv[i].key = xp[i]; kv[i].idx = i; }
            qsort(kv, (size_t)n, sizeof(kv_t), cmp_kv);
            for (int i = 0; i < n; i++) { xs[i] = kv[i].key; px[i] = kv[i].idx; }
            for (int j = 0; j < m; j++) { kv[j].key = yp[j]; kv[j].idx = j; }
            qsort(kv, (size_t)m, sizeof(kv_t), cmp_kv);
            for (int j = 0; j < m; j++) { ys[j] = kv[j].key; py[j] = kv[j].idx; }
        }
        double acc = 0.0;
        for (int i = 0; i < n; i++) { acc += (double)(a[px[i]] / dx); U[i] = (float)acc; }
        acc = 0.0;
        for (int j = 0; j < m; j++) { acc += (double)(b[py[j]] / dy); V[j] = (float)acc; }
        /* merge (U first on ties, lower index first = the order of a STABLE sort of cat(U,V)),
         * recording the cost d_k of each level.  searchsorted ranks (losses.py:219) are the same for
         * every member of a run of equal levels, so d_k is constant along a run and is evaluated with
         * the counts at the run's first element (= #{U < q}, #{V < q}). */
        int i = 0, j = 0;
        float qprev = 0.0f;
        for (int k = 0; k < K; k++) {
            const int take_u = (j >= m) || (i < n && U[i] <= V[j]);
            const float q = take_u ? U[i] : V[j];
            if (k == 0 || q != qprev) {
                const float uq = xs[i < n ? i : n - 1], vq = ys[j < m ? j : m - 1];
                double d = fabs((double)uq - (double)vq);
                d = (p == 1.0f) ? d : (p == 2.0f ? d * d : pow(d, (double)p));
                if ((flags & SOT_FLAG_LIMIT_Q) && q > 1.0f) d = 0.0; /* m_k = 0 */
                dk[k] = d;
            } else {
                dk[k] = dk[k - 1];
            }
            qprev = q;
            src[k] = take_u ? i : -(j + 1);
            if (take_u) i++; else j++;
        }
        dk[K] = 0.0;
        for (int k = 0; k < K; k++) { /* g_k = m_k d_k - m_{k+1} d_{k+1}: non-zero only at a run's last member */
            const double g = dk[k] - dk[k + 1];
            if (src[k] >= 0) gU[src[k]] = g; else gV[-src[k] - 1] = g;
        }
        /* reverse cumsum, normalisation, square, upstream scale; unsort on store */
        const double gr = (double)grad_row[r];
        double gSx = 0.0, gSy = 0.0;
        acc = 0.0;
        for (int ii = n - 1; ii >= 0; ii--) { acc += gU[ii]; gU[ii] = acc; gSx -= acc * (double)a[px[ii]]; }
        acc = 0.0;
        for (int jj = m - 1; jj >= 0; jj--) { acc += gV[jj]; gV[jj] = acc; gSy -= acc * (double)b[py[jj]]; }
        if (dn) { gSx += gSy; gSy = 0.0; }
        gSx = (Sx > 1e-7f) ? gSx / ((double)dx * (double)dx) : 0.0;
        gSy = (Sy > 1e-7f) ? gSy / ((double)dy * (double)dy) : 0.0;
        if (gx) for (int ii = 0; ii < n; ii++) {
            double g = gU[ii] / (double)dx + gSx;
            if (sq) g *= 2.0 * (double)xr[px[ii]];
            gx[r * (int64_t)n + px[ii]] = (float)(g * gr);
        }
        if (gy) for (int jj = 0; jj < m; jj++) {
            double g = gV[jj] / (double)dy + (dn ? 0.0 : gSy);
            if (sq) g *= 2.0 * (double)yr[py[jj]];
            gy[r * (int64_t)m + py[jj]] = (float)(g * gr);
        }
    }
    (void)w;   /* inside the parallel region: every thread's own */
    free(a); free(b); free(w); free(xs); free(ys); free(U); free(V); free(px); free(py);
    free(gU); free(gV); free(dk); free(src); free(kv);
    }   /* omp parallel */
    return 0;
}
