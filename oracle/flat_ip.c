/* oracle/flat_ip.c - CPU restatement of exact inner-product top-k in plain C +
 * OpenMP.  TEST INFRASTRUCTURE: used by tests as a second checker and by
 * bench.py as the "port" CPU baseline.  Never linked into the product.
 *
 * PARITY UNPINNED (see oracle/__init__.py): follows the documented semantics of
 * faiss.IndexFlatIP.search at the reference call site
 * /root/reference/core/index_manager.py:371 (scores desc, int64 labels,
 * -1 / -FLT_MAX padding, /root/reference/core/index_manager.py:376);
 * ties broken by ascending id.
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float s; int64_t id; } cand_t;

static inline int better(float s, int64_t id, const cand_t* c) {
    return s > c->s || (s == c->s && id < c->id);
}

/* sorted insertion into a best-first list of length k */
static inline void push(cand_t* top, int k, float s, int64_t id) {
    if (!better(s, id, &top[k - 1])) return;
    int j = k - 1;
    while (j > 0 && better(s, id, &top[j - 1])) { top[j] = top[j - 1]; --j; }
    top[j].s = s; top[j].id = id;
}

static inline float dot(const float* a, const float* b, int d) {
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = 0;
    for (; i + 8 <= d; i += 8)
        for (int j = 0; j < 8; ++j) acc[j] += a[i + j] * b[i + j];
    float r = 0;
    for (; i < d; ++i) r += a[i] * b[i];
    return ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])) + r;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* X: (n,d) fp32 row-major, Q: (nq,d); D: (nq,k), I: (nq,k) */
int oracle_flat_ip_search(const float* X, int64_t n, int d, const float* Q, int64_t nq, int k, float* D, int64_t* I) {
    if (k < 1 || d < 1 || n < 0 || nq < 0) return 1;
    int nt = oracle_num_threads();
    cand_t* tops = (cand_t*)malloc(sizeof(cand_t) * (size_t)nt * (size_t)k);
    if (!tops) return 2;
    for (int64_t q = 0; q < nq; ++q) {
        const float* qv = Q + q * d;
        for (size_t i = 0; i < (size_t)nt * k; ++i) { tops[i].s = -FLT_MAX; tops[i].id = INT64_MAX; }
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            cand_t* top = tops + (size_t)omp_get_thread_num() * k;
#else
            cand_t* top = tops;
#endif
#pragma omp for schedule(static)
            for (int64_t r = 0; r < n; ++r) push(top, k, dot(X + r * d, qv, d), r);
        }
        cand_t* out = tops; /* merge every thread's list into thread 0's */
        for (int t = 1; t < nt; ++t)
            for (int j = 0; j < k; ++j) {
                const cand_t* c = &tops[(size_t)t * k + j];
                if (c->id != INT64_MAX) push(out, k, c->s, c->id);
            }
        for (int j = 0; j < k; ++j) {
            if (out[j].id == INT64_MAX) { D[q * k + j] = -FLT_MAX; I[q * k + j] = -1; }
            else { D[q * k + j] = out[j].s; I[q * k + j] = out[j].id; }
        }
    }
    free(tops);
    return 0;
}
