/*
 * ssmi.h - C-ABI of libssmi.so: MI355X (gfx950) Stage-1 vector retrieval engine.
 *
 * This is the drop-in boundary for the one hot path this repository replaces:
 * the calls that EricRollei/Semantic-Search makes into CPU FAISS from
 * core/index_factory.py and core/index_manager.py.  The reference has no FFI
 * layer of its own (it talks to the SWIG `faiss` module); each entry point
 * below names the reference call site (file:line under the reference tree)
 * whose FAISS call it stands in for.  The Python facade that re-creates the
 * `faiss.Index` duck type on top of these functions lives in
 * semantic-search_amd/faiss_compat.py; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on failure; the message is
 *     available from ssmi_last_error() (thread-local).  The Python facade maps
 *     nonzero -> RuntimeError, which is what the reference catches
 *     (core/index_factory.py:275,376; core/index_manager.py:610).
 *   - caller owns every input/output buffer; the library copies on add.
 *   - one handle lives on ONE GPU (one process per GPU; sharding across GPUs is
 *     done above this ABI with torch.distributed/RCCL, see sharded.py).
 *   - ids are implicit sequential row numbers, as with FAISS add() without ids
 *     (core/index_manager.py:319-325); int64 on the ABI, < 2^32-1 per handle.
 *   - short results are padded with id -1 / score -FLT_MAX, which is what the
 *     reference filters on (core/index_manager.py:376).
 *   - result order: score descending, ties broken by ascending id.
 *   - calls on one handle must be serialised by the caller (the reference
 *     holds a per-index RLock, core/index_manager.py:175); distinct handles may
 *     be used from distinct threads.  No global mutable state besides the
 *     thread-local error string.
 */
#ifndef SSMI_H
#define SSMI_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssmi_index ssmi_index;

/* element types */
#define SSMI_F32 0
#define SSMI_F16 1
/* bfloat16 rows.  The reference's producer emits bf16 model outputs cast to fp32 (core/model_wrapper.py:201-203), so
 * for its embeddings bf16 storage is LOSSLESS at half the bytes of fp32: reconstruct returns the input bit for bit
 * and every search path returns the fp32 result up to summation order.  Other fp32 values are rounded to nearest
 * even at add (8 significant bits: coarser than SSMI_F16's 11). */
#define SSMI_BF16 2
/* where a caller buffer lives */
#define SSMI_HOST 0
#define SSMI_DEVICE 1
/* index kinds (ssmi_index_get_kind) */
#define SSMI_KIND_FLAT 0
#define SSMI_KIND_IVFFLAT 1

/* thread-local message of the last failing call on this thread */
const char* ssmi_last_error(void);
/* ABI version, bumped on incompatible change */
int ssmi_abi_version(void);
/* number of visible HIP devices (does not initialise a context on any) */
int ssmi_device_count(int* out);

/* ---- construction ------------------------------------------------------ */
/* replaces faiss.IndexFlatIP(dim)                 core/index_factory.py:140 */
int ssmi_index_create_flat(int d, int storage_dtype, int device, ssmi_index** out);
/* replaces faiss.IndexIVFFlat(quantizer, dim, nlist, METRIC_INNER_PRODUCT)
 *                                                  core/index_factory.py:144-146 */
int ssmi_index_create_ivfflat(int d, int nlist, int storage_dtype, int device, ssmi_index** out);
int ssmi_index_free(ssmi_index* h);

/* ---- attributes (faiss.Index .d .ntotal .is_trained .nlist .nprobe;
 *      core/index_factory.py:169-170,223-233,317-318) --------------------- */
int ssmi_index_get_kind(const ssmi_index* h, int* out);
int ssmi_index_get_d(const ssmi_index* h, int* out);
int ssmi_index_get_ntotal(const ssmi_index* h, int64_t* out);
int ssmi_index_get_is_trained(const ssmi_index* h, int* out);
int ssmi_index_get_nlist(const ssmi_index* h, int* out);
int ssmi_index_get_nprobe(const ssmi_index* h, int* out);
int ssmi_index_set_nprobe(ssmi_index* h, int nprobe);
int ssmi_index_get_storage_dtype(const ssmi_index* h, int* out);
int ssmi_index_get_device(const ssmi_index* h, int* out);

/* ---- train / add -------------------------------------------------------- */
/* replaces index.train(x)                          core/index_factory.py:191
 * x: (n,d) fp32 row-major.  Flat: no-op.  IVF: k-means of nlist centroids. */
int ssmi_index_train(ssmi_index* h, int64_t n, const float* x, int mem_kind);
/* IVF only: install given centroids (nlist,d) fp32 instead of running k-means
 * (no reference counterpart: used by parity tests and by ssmi_index_read). */
int ssmi_index_set_centroids(ssmi_index* h, const float* centroids, int mem_kind);
int ssmi_index_get_centroids(const ssmi_index* h, float* out_host);
/* replaces index.add(x)     core/index_manager.py:299,308,319,1188;
 *                           core/index_factory.py:307,411
 * x: (n,d) row-major of src_dtype, host or device. */
int ssmi_index_add(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind);
/* list-sharded IVF (no reference counterpart; north-star: "IVF lists by list across
 * the GPUs"): every rank passes the same vectors; this handle keeps the ones whose
 * list l has l % nshards == shard, stored under the global ids id0 + i. */
int ssmi_index_add_ivf_shard(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind,
                             int64_t id0, int shard, int nshards);
/* The same sharding without redundant work (no reference counterpart; multi-GPU add of sharded.py): a rank
 * assigns only ITS slice of the incoming vectors (ssmi_index_assign: list of the max-inner-product centroid per
 * vector, what IndexIVFFlat.add computes internally), the ranks exchange (vector, list, global id) triples so
 * that each list's members reach the rank that owns the list, and the owner appends what it received with
 * ssmi_index_add_preassigned.  labels/ids are host arrays; x is (n,d) of src_dtype, host or device. */
int ssmi_index_assign(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, int32_t* labels_out_host);
int ssmi_index_add_preassigned(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind,
                               const int32_t* labels_host, const int64_t* ids_host);
/* bench/test helper (no reference counterpart): append n rows of i.i.d.
 * N(0,1) components, L2-normalised per row, generated on the device. */
int ssmi_index_add_random_unit(ssmi_index* h, int64_t n, uint64_t seed);

/* ---- search ------------------------------------------------------------- */
/* replaces index.search(x, k)                      core/index_manager.py:371
 * q: (nq,d) fp32 row-major (host or device per q_mem_kind).
 * D: (nq,k) fp32, I: (nq,k) int64, row-major, host or device per out_mem_kind.
 * id_offset is added to every valid id (row-sharded databases).
 * stream: the hipStream_t the work is enqueued on.  When q and D/I are all host
 * buffers NULL selects the handle's own stream; when either lives on the device
 * the given stream is used as is (NULL = the HIP null stream).  With host
 * outputs the call returns after the results are in D/I; with device outputs
 * it returns after enqueueing on `stream` (caller synchronises). */
int ssmi_index_search(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind,
                      int k, float* D, int64_t* I, int out_mem_kind,
                      int64_t id_offset, void* stream);

/* ssmi_index_search with a score floor: only results with score >= min_score are reported (the rest of a row is
 * padded with -1 / -FLT_MAX like any short result).  This is the search nodes' min_score filter
 * (nodes/search_nodes.py:152-159: `[r for r in results if r.score >= min_score]`) applied inside the kernels: the
 * scans start their top-k thresholds at the floor instead of at "anything goes" and the merge cuts what is
 * left below it, so no host pass over the results is needed.  min_score <= -FLT_MAX
 * means no floor; ssmi_index_search is this call with no floor. */
int ssmi_index_search_min(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind, int k, float min_score, float* D,
                          int64_t* I, int out_mem_kind, int64_t id_offset, void* stream);

/* Allocation-free variant of ssmi_index_search.  NO COUNTERPART IN THE REFERENCE: it exists for callers that
 * cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  Flat indexes whose dimension is a multiple of 8; q, D, I are DEVICE buffers; work is enqueued on
 * `stream` (NULL = the HIP null stream) and the call returns after enqueueing.  Always the exact streaming scan
 * (K1 + merge tail) whatever the batch size.  The kernels' scratch (per-block partial k-lists, ticket counters)
 * lives in one file-scope device array of SSMI_STATIC_SCRATCH_BYTES per GPU, held until the process ends; the
 * call fails with a clear message when nq * row_slices * k * 8 bytes exceed it (row_slices <= 2 * CUs; k=10 on
 * 256 slices: ~400 queries, k=1024: 3) and never calls hipMalloc/hipFree.  All calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two handles, two threads) race; serialise them. */
#define SSMI_STATIC_SCRATCH_BYTES (8u << 20)
int ssmi_index_search_static(ssmi_index* h, int64_t nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev,
                             int64_t id_offset, void* stream);
int ssmi_static_scratch_bytes(size_t* out);

/* merge step of a sharded search (the RCCL all-gather's consumer; no reference
 * counterpart - the reference is single-index).  Inputs are `nlists` result
 * sets on the device, each (nq,k) row-major; set l starts at D_lists +
 * l*d_list_stride and I_lists + l*i_list_stride (strides in elements, so one
 * all-gathered buffer holding both scores and ids can be merged in place);
 * output is the (nq,k) top-k of their union, same ordering rules.  Ids (after every shard's id_offset) must stay
 * below 2^32-1: the merge orders 64-bit (score, id) keys with 32 bits of id (sharded.py checks the global row count
 * when the shards agree on their offsets). */
int ssmi_topk_merge(int device, int nlists, int64_t nq, int k,
                    const float* D_lists, int64_t d_list_stride,
                    const int64_t* I_lists, int64_t i_list_stride,
                    float* D_out, int64_t* I_out, void* stream);

/* ---- read back ---------------------------------------------------------- */
/* replaces index.reconstruct(i)   core/index_factory.py:274,372;
 *                                 core/index_manager.py:609    (bulk form)
 * out: (n,d) fp32 host.  IVF handles need ssmi_index_make_direct_map first,
 * otherwise this fails like FAISS does without a direct map. */
int ssmi_index_reconstruct_n(ssmi_index* h, int64_t n, const int64_t* ids, float* out_host);
/* the same with the rows delivered into DEVICE memory ((n,d) fp32 on the index's GPU; the call returns when
 * they are there): rebuild_index / compact_index (core/index_factory.py:273-274,371-372) move an index into a
 * new one chunk by chunk without a host copy of the vectors. */
int ssmi_index_reconstruct_n_dev(ssmi_index* h, int64_t n, const int64_t* ids, float* out_dev);
/* replaces the per-candidate reconstruct + np.dot of the exclusion search (core/index_manager.py:604-621):
 * out_host[i*n_neg + j] = <stored vector ids[i], neg[j]>, fp32 products of the rows as stored, computed on the
 * device (one launch for all candidates and negatives).  ids, neg (n_neg,d) fp32 and out are host buffers. */
int ssmi_index_score_ids(ssmi_index* h, int64_t n, const int64_t* ids, int n_neg, const float* neg_host, float* out_host);
int ssmi_index_make_direct_map(ssmi_index* h);

/* ---- persistence -------------------------------------------------------- */
/* replaces faiss.write_index(index, path)          core/index_manager.py:241 */
int ssmi_index_write(ssmi_index* h, const char* path);
/* same file, written incrementally: if `path` holds an earlier state of this Flat index (a prefix of its rows)
 * only the rows added since are appended and the header counts patched - byte-identical to ssmi_index_write,
 * at O(new rows) instead of O(N) per checkpoint (the reference rewrites the whole index every 10 batches,
 * core/index_manager.py:859-860 -> :241).  Anything else falls back to the full write.  SURVEY.md 8f rank 4. */
int ssmi_index_write_append(ssmi_index* h, const char* path);
/* replaces faiss.read_index(path)                  core/index_manager.py:185 */
int ssmi_index_read(const char* path, int storage_dtype, int device, ssmi_index** out);

/* ---- introspection ------------------------------------------------------ */
typedef struct ssmi_timing {
    double last_search_ms;   /* hipEvent time of the last search's kernels */
    double last_scan_ms;     /* ... of its dominant (scan / contraction) kernel */
    double last_merge_ms;
    int64_t last_scan_bytes; /* algorithmic bytes the dominant kernel streamed */
    int64_t last_scan_flops;
    int last_path;           /* 0 = streaming scan, 1 = MFMA contraction, 2 = IVF */
    int last_grid;
} ssmi_timing;
int ssmi_index_get_timing(const ssmi_index* h, ssmi_timing* out);
/* hipEvent durations (ms) of the dominant kernel (scan / contraction) of the last
 * min(max_n, 256) searches on this handle, oldest first (bench.py's roofline leg). */
int ssmi_index_get_scan_times(const ssmi_index* h, int max_n, float* ms_out, int* n_out);
/* The 16-bit shadow (no reference counterpart).  Query batches of >= 9 over an fp32 index (Flat; IVF when the
 * batch revisits lists) run on the matrix cores, which need 16-bit operands: the first such search builds a 16-bit
 * copy of the rows (+50 % HBM; fp16, or bf16 when that is lossless: ssmi_index_get_shadow_dtype), later adds keep it
 * in step, and those batches then carry that rounding of rows and queries (fp16: scores within ~1e-4 relative, ids
 * may swap at near-ties; batches of <= 8 stay exact).
 * policy 1 (default): build it on demand; if it does not fit in HBM the search falls back to the exact fp32
 * scan instead of failing.  policy 0: never - every batch stays exact; an existing shadow is freed.
 * get_shadow_state: the policy, how many rows the shadow currently holds (0 = none), and whether the last
 * attempt to build it ran out of memory. */
int ssmi_index_set_shadow_policy(ssmi_index* h, int policy);
int ssmi_index_get_shadow_state(const ssmi_index* h, int* policy, int64_t* shadow_rows, int* failed);
/* Element type of the shadow: SSMI_F16 or SSMI_BF16, -1 while there is none.  By default it is bf16 as long as every
 * fp32 row is bf16-valued (what the reference's embedder produces: bf16 model outputs cast to fp32,
 * core/model_wrapper.py:201-203) - the shadow is then an exact copy and batched searches return the exact scan's ids
 * for bf16-valued queries - and fp16 otherwise; env SSMI_SHADOW_DTYPE=f16|bf16|auto fixes it.  No reference
 * counterpart. */
int ssmi_index_get_shadow_dtype(const ssmi_index* h, int* out);
/* name of the dominant kernel (scan / contraction) the last search on this handle launched: a static string
 * owned by the library, "" before the first search (bench.py labels its roofline with it; rocprofv3 traces
 * show the same name). */
int ssmi_index_get_last_kernel(const ssmi_index* h, const char** out);
/* The shader clock the chip held under the last search's contraction kernel, when that was the 256-query MFMA kernel
 * (batches above 128; an error otherwise): workgroup 0 stamps the shader-clock counter and the 100 MHz counter at its
 * start and its end; cycles / elapsed = MHz.  The kernel is power-limited, so the same binary runs at 1.6-1.9 GHz from
 * box to box and from operand traffic to operand traffic; bench.py prints the figure next to every MFMA fraction so
 * that a slow box and a slow kernel can be told apart.  Waits for the handle's stream.  No counterpart in the reference
 * (measurement plumbing of SURVEY.md 8(d)). */
int ssmi_index_get_kernel_clock(const ssmi_index* h, double* mhz_out, int64_t* cycles_out, double* ms_out);
/* Switches the per-search hipEvent triple on (default; env SSMI_TIMING=0 starts with it off) or off.
 * An event record between two kernels costs the stream about 6 us of idle time on this device, three
 * of them 12 % of a batch-1 IVF search, so a serving loop turns it off; get_timing then reports -1 ms
 * and get_scan_times only the searches that ran with it on.  No counterpart in the reference (FAISS has
 * no per-search device timing); measurement plumbing of SURVEY.md 8(d). */
int ssmi_index_set_timing(ssmi_index* h, int enabled);
/* IVF introspection for tests: list sizes (nlist int64) */
int ssmi_index_get_list_sizes(const ssmi_index* h, int64_t* out_host);
/* ... and the list each stored vector was assigned to (ntotal int32, indexed by id) */
int ssmi_index_get_assignments(const ssmi_index* h, int32_t* out_host);
/* The members of inverted list `list` in stored order: their ids (int64) and, when vecs_out_host is not NULL, their
 * vectors ((n, d) fp32).  cap = entries the buffers hold; *n_out = the list's size (nothing is written when it
 * exceeds cap).  What faiss.write_index walks per list (invlists->get_ids / get_codes [FAISS-lib]) behind
 * core/index_manager.py:241: the single-process multi-GPU index (multi_device.py) writes ONE FAISS file from the
 * lists its shards own. */
int ssmi_index_get_list_members(ssmi_index* h, int list, int64_t cap, int64_t* ids_out_host, float* vecs_out_host, int64_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* SSMI_H */
