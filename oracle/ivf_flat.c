/* oracle/ivf_flat.c - CPU restatement of an IVF-Flat (inner product) search in plain C + OpenMP.
 * TEST INFRASTRUCTURE: a second checker for oracle/ivf_flat.py and bench.py's IVF "port" CPU baseline.
 * Never linked into the product.
 *
 * PARITY UNPINNED (see oracle/__init__.py): follows the documented semantics of
 * faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT).search as the reference builds and
 * drives it (/root/reference/core/index_factory.py:142-146, /root/reference/core/index_manager.py:371):
 * the nprobe lists whose centroids have the largest inner product with the query are visited and their
 * members ranked by exact inner product; scores descending, ties by ascending id, -1 / -FLT_MAX padding.
 *
 * Layout: the caller passes the stored rows grouped by list (Xs row r belongs to the list l with
 * off[l] <= r < off[l+1]) and the id of every grouped row.
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { float s; int64_t id; } cand_t;

static inline int better(float s, int64_t id, const cand_t* c) { return s > c->s || (s == c->s && id < c->id); }
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
static int nthreads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
static void clear(cand_t* t, size_t n) { for (size_t i = 0; i < n; ++i) { t[i].s = -FLT_MAX; t[i].id = INT64_MAX; } }

/* C: (nlist,d) centroids; Xs: rows grouped by list; ids: id of each grouped row; off: nlist+1 list bounds.
 * probes_out (optional): (nq,nprobe) probed lists, best first, -1 padded. */
int oracle_ivf_flat_search(const float* C, int nlist, int d, const float* Xs, const int64_t* ids, const int64_t* off,
                           const float* Q, int64_t nq, int nprobe, int k, float* D, int64_t* I, int64_t* probes_out) {
    if (k < 1 || d < 1 || nlist < 1 || nq < 0 || nprobe < 1) return 1;
    if (nprobe > nlist) nprobe = nlist;
    const int nt = nthreads();
    const int kk = k > nprobe ? k : nprobe;
    cand_t* tops = (cand_t*)malloc(sizeof(cand_t) * (size_t)nt * (size_t)kk);
    if (!tops) return 2;
    for (int64_t q = 0; q < nq; ++q) {
        const float* qv = Q + q * d;
        /* coarse quantizer: exact top-nprobe centroids (an IndexFlatIP over the centroids) */
        clear(tops, (size_t)nt * nprobe);
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            cand_t* top = tops + (size_t)omp_get_thread_num() * nprobe;
#else
            cand_t* top = tops;
#endif
#pragma omp for schedule(static)
            for (int l = 0; l < nlist; ++l) push(top, nprobe, dot(C + (size_t)l * d, qv, d), l);
        }
        for (int t = 1; t < nt; ++t)
            for (int j = 0; j < nprobe; ++j) {
                const cand_t* c = &tops[(size_t)t * nprobe + j];
                if (c->id != INT64_MAX) push(tops, nprobe, c->s, c->id);
            }
        int64_t* probes = (int64_t*)malloc(sizeof(int64_t) * (size_t)nprobe);
        if (!probes) { free(tops); return 2; }
        for (int j = 0; j < nprobe; ++j) {
            probes[j] = tops[j].id == INT64_MAX ? -1 : tops[j].id;
            if (probes_out) probes_out[q * nprobe + j] = probes[j];
        }
        /* list scan: members of the probed lists ranked by exact inner product */
        clear(tops, (size_t)nt * k);
#pragma omp parallel num_threads(nt)
        {
#ifdef _OPENMP
            cand_t* top = tops + (size_t)omp_get_thread_num() * k;
#else
            cand_t* top = tops;
#endif
#pragma omp for schedule(dynamic, 1)
            for (int j = 0; j < nprobe; ++j) {
                if (probes[j] < 0) continue;
                for (int64_t r = off[probes[j]]; r < off[probes[j] + 1]; ++r)
                    push(top, k, dot(Xs + (size_t)r * d, qv, d), ids[r]);
            }
        }
        free(probes);
        for (int t = 1; t < nt; ++t)
            for (int j = 0; j < k; ++j) {
                const cand_t* c = &tops[(size_t)t * k + j];
                if (c->id != INT64_MAX) push(tops, k, c->s, c->id);
            }
        for (int j = 0; j < k; ++j) {
            if (tops[j].id == INT64_MAX) { D[q * k + j] = -FLT_MAX; I[q * k + j] = -1; }
            else { D[q * k + j] = tops[j].s; I[q * k + j] = tops[j].id; }
        }
    }
    free(tops);
    return 0;
}
