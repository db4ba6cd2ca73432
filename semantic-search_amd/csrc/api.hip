// api.hip - the C-ABI of libssmi.so (include/ssmi.h) and the host-side index
// objects behind it.  Host logic only; kernels are in scan_*/merge/util/...
#include "../../include/ssmi.h"
#include "kernels.h"
#include "index.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>

using namespace ssmi;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local std::string g_err;

namespace ssmi {
int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}
}  // namespace ssmi

// ---------------------------------------------------------------------------
// RowStore
// ---------------------------------------------------------------------------
namespace ssmi {

int RowStore::init(int d_, int dtype_, hipStream_t stream) {
    d = d_;
    dtype = dtype_;
    elsize = (size_t)dtype_elsize(dtype);
    ld = (d + 7) / 8 * 8;
    row_bytes = (size_t)ld * elsize;
    // ~1 GiB segments, between 2^10 and 2^24 rows
    seg_shift = 10;
    while (seg_shift < 24 && (row_bytes << (seg_shift + 1)) <= ((size_t)1 << 30)) ++seg_shift;
    // tests use small segments to exercise the table.  Never below 2^8 rows: the tile kernels (mfma_r128 / mfma_n128:
    // 256-row tiles, ivf_lists_mfma: 32-slot pages) address a whole tile from its first row's segment, and slices start
    // at multiples of 256 rows, so a segment must hold whole tiles (found by tests/test_fuzz_gpu.py with 64-row
    // segments: reads past the end of a segment)
    const char* env = getenv("SSMI_SEG_SHIFT");
    if (env) { int v = atoi(env); if (v >= 8 && v <= 24) seg_shift = v; }
    nrows = 0;
    seg0_cap = 0;
    HIPCHECK(hipMalloc((void**)&d_segs, sizeof(void*) * kMaxSegs));
    HIPCHECK(zero_device_sync(d_segs, sizeof(void*) * kMaxSegs, stream));
    return 0;
}

void RowStore::destroy() {
    for (void* p : segs) if (p) (void)hipFree(p);
    segs.clear();
    if (d_segs) (void)hipFree(d_segs);
    d_segs = nullptr;
    nrows = 0;
    seg0_cap = 0;
}

int RowStore::reserve(long long total, hipStream_t stream) {
    const long long seg_rows = 1ll << seg_shift;
    if (total <= 0) return 0;
    const long long nseg_needed = (total + seg_rows - 1) / seg_rows;
    if (nseg_needed > kMaxSegs) return fail("index too large: %lld rows exceed %d segments", total, kMaxSegs);
    bool table_dirty = false;
    // segment 0 grows by doubling until it reaches the full segment size
    long long want0 = std::min(total, seg_rows);
    if (want0 > seg0_cap) {
        long long cap = seg0_cap > 0 ? seg0_cap : 256;
        while (cap < want0) cap <<= 1;
        if (cap > seg_rows) cap = seg_rows;
        void* p = nullptr;
        HIPCHECK(hipMalloc(&p, (size_t)cap * row_bytes));
        if (!segs.empty() && segs[0]) {
            const long long live = std::min(nrows, seg0_cap);
            hipError_t e = live > 0 ? hipMemcpyAsync(p, segs[0], (size_t)live * row_bytes, hipMemcpyDeviceToDevice, stream) : hipSuccess;
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) {  // the old segment stays the index's; the new block must not be lost
                (void)hipFree(p);
                return fail("growing segment 0 failed: %s (%s:%d)", hipGetErrorString(e), __FILE__, __LINE__);
            }
            HIPCHECK(hipFree(segs[0]));
            segs[0] = p;
        } else if (segs.empty()) {
            segs.push_back(p);
        } else {
            segs[0] = p;
        }
        seg0_cap = cap;
        table_dirty = true;
    }
    while ((long long)segs.size() < nseg_needed) {
        void* p = nullptr;
        HIPCHECK(hipMalloc(&p, (size_t)seg_rows * row_bytes));
        segs.push_back(p);
        table_dirty = true;
    }
    if (table_dirty) {
        HIPCHECK(hipMemcpyAsync(d_segs, segs.data(), sizeof(void*) * segs.size(), hipMemcpyHostToDevice, stream));
        HIPCHECK(hipStreamSynchronize(stream));
    }
    return 0;
}

RowView RowStore::view() const {
    RowView v;
    v.segs = (const void* const*)d_segs;
    v.seg_shift = seg_shift;
    v.ld = ld;
    v.nrows = nrows;
    return v;
}

}  // namespace ssmi

// ---------------------------------------------------------------------------
// index object helpers
// ---------------------------------------------------------------------------
namespace ssmi {
int ensure_ws(ssmi_index* h, int slot, size_t bytes) {
    if (h->ws_bytes[slot] >= bytes && h->ws[slot]) return 0;
    // a caller-provided stream may still be using the old buffer: drain the device (rare path)
    if (h->ws[slot]) { HIPCHECK(hipDeviceSynchronize()); HIPCHECK(hipFree(h->ws[slot])); h->ws[slot] = nullptr; h->ws_bytes[slot] = 0; }
    if (slot == 4) h->last_clk = nullptr;  // the clock stamps of the last MFMA launch lived in the old buffer
    size_t cap = std::max(bytes, (size_t)1 << 20);
    cap = (cap + 255) / 256 * 256;
    HIPCHECK(hipMalloc(&h->ws[slot], cap));
    h->ws_bytes[slot] = cap;
    return 0;
}
}  // namespace ssmi

namespace ssmi {
int order_search_on(ssmi_index* h, hipStream_t s) {
    if (h->have_last_search && h->last_search_stream != s) {
        // the previous search may still be running on its stream and owns the handle's workspaces until it ends
        if (hipEventRecord(h->ev_order, h->last_search_stream) == hipSuccess) HIPCHECK(hipStreamWaitEvent(s, h->ev_order, 0));
        else (void)hipGetLastError();  // that stream no longer exists: its work has drained
    }
    h->last_search_stream = s;
    h->have_last_search = true;
    return 0;
}
void quiesce(ssmi_index* h) {
    if (h->have_last_search && h->last_search_stream != h->stream)
        if (hipStreamSynchronize(h->last_search_stream) != hipSuccess) (void)hipGetLastError();
    h->have_last_search = false;
}
}  // namespace ssmi

static int create_common(int d, int storage_dtype, int device, int kind, ssmi_index** out) {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (d < 1 || d > 65536) return fail("dimension %d out of range [1, 65536]", d);
    if (!dtype_known(storage_dtype)) return fail("unknown storage dtype %d", storage_dtype);
    int ndev = 0;
    HIPCHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d not available (%d visible)", device, ndev);
    DeviceGuard g(device);
    if (!g.ok) return fail("hipSetDevice(%d) failed", device);
    // everything below may fail half-way: the guard hands a partly built handle to ssmi_index_free (which
    // takes zeroed members) unless the construction completes
    struct Guard {
        ssmi_index* h;
        ~Guard() { if (h) ssmi_index_free(h); }
    } guard{new ssmi_index()};
    ssmi_index* h = guard.h;
    h->kind = kind;
    h->device = device;
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, device));
    h->num_cu = prop.multiProcessorCount;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("libssmi is built for gfx950 (MI355X) only; device %d is %s", device, prop.gcnArchName);
    CHECK(static_scratch_setup(device));  // ssmi_index_search_static's per-device scratch address
    HIPCHECK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    HIPCHECK(hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming));
    for (int i = 0; i < ssmi_index::kEvRing; ++i)
        for (int j = 0; j < 3; ++j) HIPCHECK(hipEventCreate(&h->ev[i][j]));
    if (!getenv("SSMI_NO_TAIL")) {
        HIPCHECK(hipMalloc((void**)&h->tail_counters, ssmi_index::kTailCounters * sizeof(unsigned)));
        HIPCHECK(zero_device_sync(h->tail_counters, ssmi_index::kTailCounters * sizeof(unsigned), h->stream));
    }
    if (const char* e = getenv("SSMI_TIMING")) h->timing_on = e[0] != '0';
    if (const char* e = getenv("SSMI_SHADOW_DTYPE")) h->shadow_pref = !strcmp(e, "f16") ? SSMI_F16 : (!strcmp(e, "bf16") ? SSMI_BF16 : -1);
    HIPCHECK(hipMalloc((void**)&h->d_flag, 256));
    CHECK(h->rows.init(d, storage_dtype, h->stream));
    guard.h = nullptr;
    *out = h;
    return 0;
}

// Pads/stages queries on the device; returns the (nq, ld) fp32 device pointer.
namespace ssmi {
int stage_queries(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind, hipStream_t s, const float** out) {
    const RowStore& R = h->rows;
    if (q_mem_kind == SSMI_DEVICE && R.ld == R.d) { *out = q; return 0; }
    CHECK(ensure_ws(h, 1, (size_t)nq * R.ld * 4));
    float* qp = (float*)h->ws[1];
    if (q_mem_kind == SSMI_HOST) {
        if (R.ld == R.d) {
            HIPCHECK(hipMemcpyAsync(qp, q, (size_t)nq * R.d * 4, hipMemcpyHostToDevice, s));
        } else {
            HIPCHECK(hipMemsetAsync(qp, 0, (size_t)nq * R.ld * 4, s));
            HIPCHECK(hipMemcpy2DAsync(qp, (size_t)R.ld * 4, q, (size_t)R.d * 4, (size_t)R.d * 4, (size_t)nq,
                                      hipMemcpyHostToDevice, s));
        }
    } else {
        HIPCHECK(launch_pad_queries(q, (int)nq, R.d, R.ld, qp, s));
    }
    *out = qp;
    return 0;
}
}  // namespace ssmi

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

const char* ssmi_last_error(void) { return g_err.c_str(); }
int ssmi_abi_version(void) { return 1; }
int ssmi_device_count(int* out) {
    return guarded([&]() -> int {
    if (!out) return fail("out is NULL");
    HIPCHECK(hipGetDeviceCount(out));
    return 0;
    });
}

int ssmi_index_create_flat(int d, int storage_dtype, int device, ssmi_index** out) {
    return guarded([&]() -> int {
    return create_common(d, storage_dtype, device, SSMI_KIND_FLAT, out);
    });
}

int ssmi_index_free(ssmi_index* h) {
    return guarded([&]() -> int {
    if (!h) return 0;
    DeviceGuard g(h->device);
    quiesce(h);
    (void)hipStreamSynchronize(h->stream);
    if (h->ev_order) (void)hipEventDestroy(h->ev_order);
    ivf_destroy(h);
    h->rows.destroy();
    if (h->has_shadow) h->shadow16.destroy();
    for (int i = 0; i < kWsSlots; ++i) if (h->ws[i]) (void)hipFree(h->ws[i]);
    if (h->tail_counters) (void)hipFree(h->tail_counters);
    if (h->d_flag) (void)hipFree(h->d_flag);
    if (h->pin) (void)hipHostFree(h->pin);
    for (int i = 0; i < ssmi_index::kEvRing; ++i)
        for (int j = 0; j < 3; ++j) if (h->ev[i][j]) (void)hipEventDestroy(h->ev[i][j]);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return 0;
    });
}

#define GETTER(name, type, expr)                              \
    int ssmi_index_get_##name(const ssmi_index* h, type* out) { \
        if (!h || !out) return fail("NULL argument");         \
        *out = (expr);                                        \
        return 0;                                             \
    }
GETTER(kind, int, h->kind)
GETTER(d, int, h->rows.d)
GETTER(ntotal, int64_t, h->kind == SSMI_KIND_FLAT ? h->rows.nrows : h->ivf_ntotal)
GETTER(is_trained, int, h->kind == SSMI_KIND_FLAT ? 1 : (h->trained ? 1 : 0))
GETTER(storage_dtype, int, h->rows.dtype)
GETTER(device, int, h->device)

int ssmi_index_get_nlist(const ssmi_index* h, int* out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("nlist: not an IVF index");
    *out = h->nlist;
    return 0;
    });
}
int ssmi_index_get_nprobe(const ssmi_index* h, int* out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("nprobe: not an IVF index");
    *out = h->nprobe;
    return 0;
    });
}
int ssmi_index_set_nprobe(ssmi_index* h, int nprobe) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("nprobe: not an IVF index");
    if (nprobe < 1) return fail("nprobe must be >= 1");
    h->nprobe = nprobe;
    return 0;
    });
}

// --------------------------------------------------------------------- add
}  // extern "C" (reopened below)
namespace ssmi {
// Append n rows (host or device, fp32 or fp16) to a row store, converting/padding as needed.
int store_add(ssmi_index* h, RowStore& R, int64_t n, const void* x, int src_dtype, int mem_kind) {
    const long long row0 = R.nrows;
    if ((unsigned long long)(row0 + n) >= 0xFFFFFFFFull) return fail("index would exceed 2^32-2 vectors");
    CHECK(R.reserve(row0 + n, h->stream));
    const size_t src_el = (size_t)dtype_elsize(src_dtype);
    const long long seg_rows = 1ll << R.seg_shift;
    if (src_dtype == R.dtype && R.ld == R.d) {
        // rows are bit-identical to the source: plain copies, one per segment touched
        long long done = 0;
        while (done < n) {
            const long long row = row0 + done;
            const long long seg = row >> R.seg_shift, off = row & (seg_rows - 1);
            const long long cnt = std::min<long long>(n - done, seg_rows - off);
            HIPCHECK(hipMemcpyAsync((char*)R.segs[seg] + (size_t)off * R.row_bytes,
                                    (const char*)x + (size_t)done * R.row_bytes, (size_t)cnt * R.row_bytes,
                                    mem_kind == SSMI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, h->stream));
            done += cnt;
        }
    } else {
        RowView v = R.view();
        const long long chunk = std::max<long long>(1, ((size_t)64 << 20) / ((size_t)R.d * src_el));
        long long done = 0;
        while (done < n) {
            const long long cnt = std::min<long long>(chunk, n - done);
            const void* src = (const char*)x + (size_t)done * R.d * src_el;
            if (mem_kind == SSMI_HOST) {
                CHECK(ensure_ws(h, 0, (size_t)cnt * R.d * src_el));
                HIPCHECK(hipMemcpyAsync(h->ws[0], src, (size_t)cnt * R.d * src_el, hipMemcpyHostToDevice, h->stream));
                src = h->ws[0];
            }
            HIPCHECK(launch_cvt_store(src, src_dtype, cnt, R.d, v, R.dtype, (void* const*)R.d_segs, row0 + done, h->stream));
            if (mem_kind == SSMI_HOST) HIPCHECK(hipStreamSynchronize(h->stream));  // staging buffer reused
            done += cnt;
        }
    }
    HIPCHECK(hipStreamSynchronize(h->stream));
    R.nrows = row0 + n;
    return 0;
}
}  // namespace ssmi
extern "C" {

int ssmi_index_add(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!x) return fail("NULL vectors");
    if (!dtype_known(src_dtype)) return fail("unknown source dtype %d", src_dtype);
    if (mem_kind != SSMI_HOST && mem_kind != SSMI_DEVICE) return fail("unknown mem_kind %d", mem_kind);
    DeviceGuard g(h->device);
    quiesce(h);
    h->shadow_failed = false;
    if (h->kind == SSMI_KIND_FLAT) return store_add(h, h->rows, n, x, src_dtype, mem_kind);
    return ivf_add(h, n, x, src_dtype, mem_kind);
    });
}

int ssmi_index_add_random_unit(ssmi_index* h, int64_t n, uint64_t seed) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    DeviceGuard g(h->device);
    quiesce(h);
    if (h->kind != SSMI_KIND_FLAT) return ivf_add_random_unit(h, n, seed);
    RowStore& R = h->rows;
    const long long row0 = R.nrows;
    if ((unsigned long long)(row0 + n) >= 0xFFFFFFFFull) return fail("index would exceed 2^32-2 vectors");
    CHECK(R.reserve(row0 + n, h->stream));
    HIPCHECK(launch_fill_random_unit(R.view(), R.dtype, (void* const*)R.d_segs, row0, n, R.d, seed, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    R.nrows = row0 + n;
    return 0;
    });
}

int ssmi_index_train(ssmi_index* h, int64_t n, const float* x, int mem_kind) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (h->kind == SSMI_KIND_FLAT) return 0;  // IndexFlatIP.train is a no-op
    DeviceGuard g(h->device);
    quiesce(h);
    return ivf_train(h, n, x, mem_kind);
    });
}

// ------------------------------------------------------------------ search
// Smallest query batch that takes the MFMA path.  fp16 / bf16 rows: 5 (the register-FMA scan is VALU-bound beyond 4
// queries, the 128-query MFMA stream is HBM-bound); fp32 rows: 9 (below that the exact fp32 scan, above it the
// fp16 shadow copy).
static int mfma_min_batch(int dtype) {
    const char* e = getenv("SSMI_MFMA_MIN_B");
    int v = e ? atoi(e) : (dtype_is16(dtype) ? 5 : 9);
    return v < 1 ? 1 : v;
}

static int store_search_mfma(ssmi_index* h, const RowStore& R, int64_t nq, const float* qdev, int k, float* Ddev,
                             int64_t* Idev, int64_t id_offset, hipStream_t s, bool record, unsigned min_ord) {
    // 16 query tiles -> 16 database slices on 256 CUs.  A small store (an IVF coarse quantizer: add and k-means
    // assign 65536 vectors at a time) is walked whole by every workgroup instead: one pass with 256 query
    // tiles, not 16 passes of one row tile per workgroup
    const int64_t kMaxBatch = R.nrows <= 65536 ? 65536 : 4096;
    hipEvent_t* ev = h->ev[h->n_searches % ssmi_index::kEvRing];
    MfmaPlan p = {};
    for (int64_t done = 0; done < nq; done += kMaxBatch) {
        const int nb = (int)std::min<int64_t>(kMaxBatch, nq - done);
        if (!plan_flat_mfma(R.dtype, R.ld, R.nrows, nb, k, h->num_cu, &p)) return fail("internal: MFMA plan failed");
        CHECK(ensure_ws(h, 4, p.qh_bytes));
        CHECK(ensure_ws(h, 5, p.cand_bytes));
        CHECK(ensure_ws(h, 2, (size_t)nb * p.nslices * k * sizeof(skey_t)));
        skey_t* partial = (skey_t*)h->ws[2];
        if (record && h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[0], s));
        HIPCHECK(launch_flat_mfma(R.view(), qdev + (size_t)done * R.ld, nb, k, p, h->ws[4], h->ws[5], partial, s, min_ord));
        if (record && h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[1], s));
        HIPCHECK(launch_merge_keys(partial, nb, p.nslices, k, Ddev + (size_t)done * k, Idev + (size_t)done * k,
                                   id_offset, s, min_ord));
        if (record && h->timing_on && done == 0) { HIPCHECK(hipEventRecord(ev[2], s)); h->merge_event = true; }
    }
    if (!record) return 0;
    if (h->timing_on) h->n_searches++;
    h->events_valid = h->timing_on;
    h->timing_valid = true;
    h->timing.last_path = 1;
    h->last_kernel = p.kernel;
    h->last_clk = p.clk_off ? reinterpret_cast<const unsigned long long*>(static_cast<const unsigned char*>(h->ws[4]) + p.clk_off) : nullptr;
    h->timing.last_grid = p.grid;
    h->timing.last_scan_bytes = (int64_t)R.nrows * R.d * R.elsize;
    h->timing.last_scan_flops = 2ll * std::min<int64_t>(nq, kMaxBatch) * R.nrows * R.d;
    return 0;
}

}  // extern "C" (reopened below)
namespace ssmi {
// Exact top-k of (nq, ld) fp32 device queries over one row store: K1 (streaming scan)
// for small batches, K2 (MFMA) for large batches over fp16 rows; K3 merges.
// record = false: no events/timing (used for the IVF coarse quantizer).
int store_search(ssmi_index* h, const RowStore& R, int64_t nq, const float* qdev /*(nq,ld)*/, int k, float* Ddev,
                 int64_t* Idev, int64_t id_offset, hipStream_t s, bool record, ProbePlanRequest* plan, unsigned min_ord) {
    if (dtype_is16(R.dtype) && (R.ld % 64) == 0 && nq >= mfma_min_batch(R.dtype))
        return store_search_mfma(h, R, nq, qdev, k, Ddev, Idev, id_offset, s, record, min_ord);
    if (&R == &h->rows && h->kind == SSMI_KIND_FLAT && R.dtype == SSMI_F32 && (R.ld % 64) == 0 &&
        nq >= mfma_min_batch(SSMI_F32) && h->shadow_policy != 0 && !h->shadow_failed && !getenv("SSMI_NO_SHADOW")) {
        // bring the 16-bit shadow up to date, then run the MFMA contraction on it.  A shadow that does not fit in
        // HBM is not an error of the search: it is dropped and this and later batches stay on the exact scan.
        // Its element type (shadow_pref, env SSMI_SHADOW_DTYPE=f16|bf16|auto): auto = bf16 as long as every row
        // survives the narrowing bit for bit - the reference's embeddings are bf16 model outputs cast to fp32
        // (core/model_wrapper.py:201-203), for which a bf16 shadow is LOSSLESS and the batched search returns the
        // exact scan's ids - and fp16 (11 significant bits against 8) as soon as one row does not: the shadow is
        // then converted afresh from the fp32 rows, once.
        auto bring_up = [&]() -> int {
            if (!h->has_shadow) {
                CHECK(h->shadow16.init(R.d, h->shadow_pref == SSMI_F16 ? SSMI_F16 : SSMI_BF16, s));
                h->has_shadow = true;
            }
            RowStore& S = h->shadow16;
            if (S.nrows < R.nrows) {
                CHECK(S.reserve(R.nrows, s));
                const bool probe = h->shadow_pref < 0 && S.dtype == SSMI_BF16;
                if (probe) HIPCHECK(hipMemsetAsync(h->d_flag, 0, 4, s));
                HIPCHECK(launch_shadow_rows(R.view(), (void* const*)S.d_segs, S.seg_shift, S.dtype, S.nrows, R.nrows - S.nrows,
                                            probe ? h->d_flag : nullptr, s));
                int lossy = 0;
                if (probe) {
                    HIPCHECK(hipMemcpyAsync(&lossy, h->d_flag, 4, hipMemcpyDeviceToHost, s));
                    HIPCHECK(hipStreamSynchronize(s));
                }
                if (lossy) {  // not bf16-valued after all: the shadow becomes fp16, from row 0
                    S.destroy();
                    CHECK(S.init(R.d, SSMI_F16, s));
                    CHECK(S.reserve(R.nrows, s));
                    HIPCHECK(launch_shadow_rows(R.view(), (void* const*)S.d_segs, S.seg_shift, SSMI_F16, 0, R.nrows, nullptr, s));
                }
                S.nrows = R.nrows;
            }
            return 0;
        };
        if (bring_up() == 0) return store_search_mfma(h, h->shadow16, nq, qdev, k, Ddev, Idev, id_offset, s, record, min_ord);
        (void)hipGetLastError();
        (void)hipStreamSynchronize(s);
        if (h->has_shadow) { h->shadow16.destroy(); h->has_shadow = false; }
        h->shadow_failed = true;
    }
    ScanPlan p = {};
    std::string err;
    // bound the partial-list buffer: process queries in batches
    int64_t done = 0;
    bool first = true;
    hipEvent_t* ev = h->ev[h->n_searches % ssmi_index::kEvRing];
    while (done < nq) {
        int64_t nb = nq - done;
        if (!plan_flat_scan(R.dtype, R.ld, R.nrows, (int)std::min<int64_t>(nb, 1 << 20), k, h->num_cu, &p, &err))
            return fail("%s", err.c_str());
        const size_t per_q = (size_t)p.grid_x * k * sizeof(skey_t);
        const int64_t cap = std::max<int64_t>(p.bt, (int64_t)(((size_t)256 << 20) / per_q) / p.bt * p.bt);
        if (nb > cap) nb = cap;
        p.grid_y = (int)((nb + p.bt - 1) / p.bt);
#ifdef SSMI_SCAN_STAMP
        CHECK(ensure_ws(h, 2, (size_t)nb * per_q + (size_t)p.grid_x * 16 * 8));
#else
        CHECK(ensure_ws(h, 2, (size_t)nb * per_q));
#endif
        skey_t* partial = (skey_t*)h->ws[2];
        const float* qb = qdev + (size_t)done * R.ld;
        // the scan merges its own per-block lists when a ticket counter per query tile is at hand
        const bool tail = h->tail_counters && p.grid_y <= ssmi_index::kTailCounters;
        p.tail = TailMerge{tail ? h->tail_counters : nullptr, Ddev + (size_t)done * k, Idev + (size_t)done * k, id_offset,
                           nullptr, nullptr, nullptr, nullptr, min_ord};
        if (tail && plan && p.r == 1 && done == 0 && nb == nq) {  // IVF coarse quantizer: K7 rides along
            p.tail.plan_sizes = plan->sizes;
            p.tail.plan_prefix = plan->prefix;
            p.tail.plan_page_off = plan->page_off;
            p.tail.plan_po = plan->po;
            plan->done = true;
        }
        const bool rec = record && first && h->timing_on;
#ifdef SSMI_SCAN_STAMP
        if (getenv("SSMI_SCAN_STAMP_SWAP")) p.tail.id_offset |= 1ll << 62;
#endif
        if (rec) HIPCHECK(hipEventRecord(ev[0], s));
        HIPCHECK(launch_flat_scan(R.dtype, R.view(), qb, (int)nb, k, partial, p, s));
        if (rec) HIPCHECK(hipEventRecord(ev[1], s));
#ifdef SSMI_SCAN_STAMP
        if (getenv("SSMI_SCAN_STAMP_PRINT") && record) {  // finish time of the rows per XCD (block b runs on XCD b % 8)
            HIPCHECK(hipStreamSynchronize(s));
            const int nwv = p.threads / 64;
            std::vector<unsigned long long> st((size_t)p.grid_x * nwv);
            HIPCHECK(hipMemcpy(st.data(), (char*)h->ws[2] + (size_t)nb * per_q, st.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (auto v : st) { t0 = std::min(t0, v); t1 = std::max(t1, v); }
            double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < p.grid_x; ++b)
                for (int w = 0; w < nwv; ++w) xcd[b % 8] += (double)(st[(size_t)b * nwv + w] - t0) * 0.01 / (nwv * (p.grid_x / 8.0));
            fprintf(stderr, "[scan stamp] grid %d x %d waves; rows-done after the first wave to finish, mean per XCD: %.0f %.0f %.0f %.0f %.0f %.0f %.0f %.0f us; last wave %.0f us\n",
                    p.grid_x, nwv, xcd[0], xcd[1], xcd[2], xcd[3], xcd[4], xcd[5], xcd[6], xcd[7], (double)(t1 - t0) * 0.01);
        }
#endif
        if (!tail)
            HIPCHECK(launch_merge_keys(partial, (int)nb, p.grid_x, k, Ddev + (size_t)done * k, Idev + (size_t)done * k,
                                       id_offset, s, min_ord));
        if (rec && !tail) HIPCHECK(hipEventRecord(ev[2], s));  // with a merge tail there is nothing between ev[1] and ev[2]
        if (rec) h->merge_event = !tail;
        first = false;
        done += nb;
    }
    if (!record) return 0;
    if (h->timing_on) h->n_searches++;
    h->events_valid = h->timing_on;
    h->timing_valid = true;
    h->timing.last_path = 0;
    h->last_kernel = p.mm ? "flat_scan_mm_kernel" : "flat_scan_topk_kernel";
    h->last_clk = nullptr;
    h->timing.last_grid = p.grid_x * p.grid_y;
    h->timing.last_scan_bytes = (int64_t)R.nrows * R.d * R.elsize * ((nq + p.bt - 1) / p.bt);
    h->timing.last_scan_flops = 2ll * nq * R.nrows * R.d;
    return 0;
}
}  // namespace ssmi
extern "C" {

int ssmi_index_search_min(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind, int k, float min_score, float* D,
                          int64_t* I, int out_mem_kind, int64_t id_offset, void* stream) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (nq < 0) return fail("negative nq");
    if (nq == 0) return 0;
    if (!q || !D || !I) return fail("NULL buffer");
    if (k < 1) return fail("k must be >= 1");
    if (k > kMaxK) return fail("k=%d exceeds the supported maximum %d", k, kMaxK);
    if (nq > (1ll << 24)) return fail("nq too large");
    if (min_score != min_score) return fail("min_score is NaN");
    // f32_to_ord(min_score); -FLT_MAX or below (incl. -inf) = no floor, encoded as 0
    const unsigned min_ord = min_score <= SSMI_NEG_MAX ? 0u : f32_to_ord(min_score);
    if (q_mem_kind != SSMI_HOST && q_mem_kind != SSMI_DEVICE) return fail("unknown q_mem_kind %d", q_mem_kind);
    if (out_mem_kind != SSMI_HOST && out_mem_kind != SSMI_DEVICE) return fail("unknown out_mem_kind %d", out_mem_kind);
    DeviceGuard g(h->device);
    // all-host call with no stream: the handle's own stream; otherwise exactly the caller's
    // stream (NULL then means the HIP null stream, e.g. torch's default stream)
    const bool any_dev = q_mem_kind == SSMI_DEVICE || out_mem_kind == SSMI_DEVICE;
    hipStream_t s = (stream || any_dev) ? (hipStream_t)stream : h->stream;
    // An untrained IVF index holds nothing (add refuses until train): like FAISS, whose IndexIVF::search does not test
    // is_trained and finds no list to probe, the search returns the "no result" padding.  The reference searches
    // without a guard while an IVF index still buffers its training vectors (core/index_manager.py:371).
    CHECK(order_search_on(h, s));

    float* Ddev = D;
    int64_t* Idev = I;
    const size_t d_bytes = ((size_t)nq * k * 4 + 255) / 256 * 256;
    const size_t out_bytes = d_bytes + (size_t)nq * k * 8;
    if (out_mem_kind == SSMI_HOST) {
        CHECK(ensure_ws(h, 3, out_bytes));
        Ddev = (float*)h->ws[3];
        Idev = (int64_t*)((char*)h->ws[3] + d_bytes);
    }
    // all-host call: queries and results go through one pinned buffer [queries (nq, ld)][D][I].  (With device
    // outputs the call returns before the copies have run, so the pinned block could not be reused safely.)
    const size_t q_bytes = ((size_t)nq * h->rows.ld * 4 + 255) / 256 * 256;
    unsigned char* pin = nullptr;
    if (q_mem_kind == SSMI_HOST && out_mem_kind == SSMI_HOST && q_bytes + out_bytes <= ssmi_index::kPinMax) {
        if (h->pin_bytes < q_bytes + out_bytes) {
            size_t want = std::max<size_t>(64 << 10, h->pin_bytes);
            while (want < q_bytes + out_bytes) want *= 2;
            void* p = nullptr;
            if (hipHostMalloc(&p, want, hipHostMallocDefault) == hipSuccess) {
                if (h->pin) (void)hipHostFree(h->pin);
                h->pin = p;
                h->pin_bytes = want;
            } else {
                (void)hipGetLastError();  // no pinned memory: the pageable copies below still work
            }
        }
        if (h->pin_bytes >= q_bytes + out_bytes) pin = (unsigned char*)h->pin;
    }
    const int64_t ntotal = h->kind == SSMI_KIND_FLAT ? h->rows.nrows : h->ivf_ntotal;
    if (ntotal == 0 || (h->kind == SSMI_KIND_IVFFLAT && !h->trained)) {
        HIPCHECK(launch_fill_empty(Ddev, Idev, nq * k, s));
    } else {
        const float* qdev = nullptr;
        if (pin) {
            const RowStore& R = h->rows;
            float* qpin = (float*)pin;
            if (R.ld == R.d) {
                memcpy(qpin, q, (size_t)nq * R.d * 4);
            } else {
                for (int64_t i = 0; i < nq; ++i) {
                    memcpy(qpin + (size_t)i * R.ld, q + (size_t)i * R.d, (size_t)R.d * 4);
                    memset(qpin + (size_t)i * R.ld + R.d, 0, (size_t)(R.ld - R.d) * 4);
                }
            }
            CHECK(ensure_ws(h, 1, (size_t)nq * R.ld * 4));
            HIPCHECK(hipMemcpyAsync(h->ws[1], qpin, (size_t)nq * R.ld * 4, hipMemcpyHostToDevice, s));
            qdev = (const float*)h->ws[1];
        } else {
            CHECK(stage_queries(h, nq, q, q_mem_kind, s, &qdev));
        }
        if (h->kind == SSMI_KIND_FLAT) CHECK(store_search(h, h->rows, nq, qdev, k, Ddev, Idev, id_offset, s, true, nullptr, min_ord));
        else CHECK(ivf_search(h, nq, qdev, k, Ddev, Idev, id_offset, s, min_ord));
    }
    if (pin) {  // (D, I) sit back to back in workspace 3: one copy, then out of the pinned block
        unsigned char* opin = pin + q_bytes;
        HIPCHECK(hipMemcpyAsync(opin, Ddev, out_bytes, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
        memcpy(D, opin, (size_t)nq * k * 4);
        memcpy(I, opin + d_bytes, (size_t)nq * k * 8);
    } else if (out_mem_kind == SSMI_HOST) {
        HIPCHECK(hipMemcpyAsync(D, Ddev, (size_t)nq * k * 4, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipMemcpyAsync(I, Idev, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        HIPCHECK(hipStreamSynchronize(s));
    }
    return 0;
    });
}

int ssmi_index_search(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind, int k, float* D, int64_t* I,
                      int out_mem_kind, int64_t id_offset, void* stream) {
    return ssmi_index_search_min(h, nq, q, q_mem_kind, k, SSMI_NEG_MAX, D, I, out_mem_kind, id_offset, stream);
}

int ssmi_topk_merge(int device, int nlists, int64_t nq, int k, const float* D_lists, int64_t d_list_stride,
                    const int64_t* I_lists, int64_t i_list_stride, float* D_out, int64_t* I_out, void* stream) {
    return guarded([&]() -> int {
    if (nlists < 1 || nq < 0 || k < 1) return fail("bad merge shape");
    if (k > kMaxK) return fail("k=%d exceeds the supported maximum %d", k, kMaxK);
    if (nq == 0) return 0;
    if (!D_lists || !I_lists || !D_out || !I_out) return fail("NULL buffer");
    DeviceGuard g(device);
    if (!g.ok) return fail("hipSetDevice(%d) failed", device);
    if (d_list_stride < nq * k || i_list_stride < nq * k) return fail("list stride smaller than nq*k");
    HIPCHECK(launch_merge_lists(D_lists, d_list_stride, I_lists, i_list_stride, nlists, (int)nq, k, D_out, I_out,
                                (hipStream_t)stream));
    return 0;
    });
}

// ------------------------------------------------------------- reconstruct
}  // extern "C" (reopened below)
namespace {
// ids (host) -> row numbers of h->rows (identity for Flat, the direct map for IVF), validated
int resolve_rows(ssmi_index* h, int64_t n, const int64_t* ids, std::vector<int64_t>* slots, const int64_t** rows) {
    const int64_t ntotal = h->kind == SSMI_KIND_FLAT ? h->rows.nrows : h->ivf_ntotal;
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] < 0 || ids[i] >= ntotal) return fail("reconstruct: id %lld out of range [0, %lld)", (long long)ids[i], (long long)ntotal);
    *rows = ids;
    if (h->kind == SSMI_KIND_IVFFLAT) {
        CHECK(ivf_ids_to_slots(h, n, ids, slots));
        *rows = slots->data();
    }
    return 0;
}
// rows by id -> (n, d) fp32, to a host buffer (staged through the workspace) or straight into a device buffer
int gather_by_ids(ssmi_index* h, int64_t n, const int64_t* ids, float* out, bool out_on_device) {
    const RowStore& R = h->rows;
    std::vector<int64_t> slots;
    const int64_t* src_ids = nullptr;
    CHECK(resolve_rows(h, n, ids, &slots, &src_ids));
    const int64_t chunk = std::max<int64_t>(1, ((size_t)64 << 20) / ((size_t)R.d * 4));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const size_t ids_bytes = ((size_t)cnt * 8 + 255) / 256 * 256;
        CHECK(ensure_ws(h, 0, ids_bytes + 256 + (out_on_device ? 0 : (size_t)cnt * R.d * 4)));
        int64_t* dids = (int64_t*)h->ws[0];
        int* bad = (int*)((char*)h->ws[0] + ids_bytes);
        float* dout = out_on_device ? out + (size_t)done * R.d : (float*)((char*)h->ws[0] + ids_bytes + 256);
        HIPCHECK(hipMemcpyAsync(dids, src_ids + done, (size_t)cnt * 8, hipMemcpyHostToDevice, h->stream));
        HIPCHECK(hipMemsetAsync(bad, 0, 4, h->stream));
        RowView v = R.view();
        if (h->kind == SSMI_KIND_IVFFLAT) v.nrows = h->ivf_slots_used;
        HIPCHECK(launch_gather_rows(v, R.dtype, dids, cnt, R.d, dout, bad, h->stream));
        if (!out_on_device)
            HIPCHECK(hipMemcpyAsync(out + (size_t)done * R.d, dout, (size_t)cnt * R.d * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    return 0;
}
}  // namespace
extern "C" {

int ssmi_index_reconstruct_n(ssmi_index* h, int64_t n, const int64_t* ids, float* out_host) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!ids || !out_host) return fail("NULL buffer");
    DeviceGuard g(h->device);
    quiesce(h);
    return gather_by_ids(h, n, ids, out_host, false);
    });
}

int ssmi_index_reconstruct_n_dev(ssmi_index* h, int64_t n, const int64_t* ids, float* out_dev) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!ids || !out_dev) return fail("NULL buffer");
    DeviceGuard g(h->device);
    quiesce(h);
    return gather_by_ids(h, n, ids, out_dev, true);
    });
}

int ssmi_index_score_ids(ssmi_index* h, int64_t n, const int64_t* ids, int n_neg, const float* neg_host, float* out_host) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (n < 0 || n_neg < 0) return fail("negative count");
    if (n == 0 || n_neg == 0) return 0;
    if (!ids || !neg_host || !out_host) return fail("NULL buffer");
    if (n > (1ll << 24) || n_neg > 4096) return fail("score_ids: too many candidates or negatives");
    DeviceGuard g(h->device);
    quiesce(h);
    const RowStore& R = h->rows;
    std::vector<int64_t> slots;
    const int64_t* src_ids = nullptr;
    CHECK(resolve_rows(h, n, ids, &slots, &src_ids));
    const size_t ids_bytes = ((size_t)n * 8 + 255) / 256 * 256;
    const size_t neg_bytes = ((size_t)n_neg * R.d * 4 + 255) / 256 * 256;
    CHECK(ensure_ws(h, 0, ids_bytes + 256 + neg_bytes + (size_t)n * n_neg * 4));
    int64_t* dids = (int64_t*)h->ws[0];
    int* bad = (int*)((char*)h->ws[0] + ids_bytes);
    float* dneg = (float*)((char*)h->ws[0] + ids_bytes + 256);
    float* dout = (float*)((char*)h->ws[0] + ids_bytes + 256 + neg_bytes);
    HIPCHECK(hipMemcpyAsync(dids, src_ids, (size_t)n * 8, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(dneg, neg_host, (size_t)n_neg * R.d * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemsetAsync(bad, 0, 4, h->stream));
    RowView v = R.view();
    if (h->kind == SSMI_KIND_IVFFLAT) v.nrows = h->ivf_slots_used;
    HIPCHECK(launch_score_rows(v, R.dtype, dids, n, R.d, dneg, n_neg, dout, bad, h->stream));
    HIPCHECK(hipMemcpyAsync(out_host, dout, (size_t)n * n_neg * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return 0;
    });
}

// ----------------------------------------------------------------- timing
int ssmi_index_get_timing(const ssmi_index* h, ssmi_timing* out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail("NULL argument");
    if (!h->timing_valid) return fail("no search has run on this index");
    DeviceGuard g(h->device);
    ssmi_timing t = h->timing;
    float a = -1.f, b = -1.f;  // -1: the last search ran with timing switched off
    if (h->events_valid) {
        hipEvent_t const* ev = h->ev[(h->n_searches - 1) % ssmi_index::kEvRing];
        HIPCHECK(hipEventSynchronize(ev[h->merge_event ? 2 : 1]));
        HIPCHECK(hipEventElapsedTime(&a, ev[0], ev[1]));
        b = 0.f;  // the scan merged its own lists (tail_merge.h): no separate merge launch
        if (h->merge_event) HIPCHECK(hipEventElapsedTime(&b, ev[1], ev[2]));
    }
    t.last_scan_ms = a;
    t.last_merge_ms = b;
    t.last_search_ms = h->events_valid ? a + b : -1.f;
    if (t.last_path == 2 && t.last_grid > 0 && h->ws[10]) {
        // per-query IVF scan: rows probed by the first query of the last search (its prefix-sum total)
        long long rows = 0;
        const int nprobe = std::min(h->nprobe, h->nlist);
        // on the stream of the search that wrote it (a copy on the NULL stream is not ordered against other streams)
        hipStream_t ls = h->have_last_search ? h->last_search_stream : h->stream;
        HIPCHECK(hipMemcpyAsync(&rows, (const long long*)h->ws[10] + nprobe, sizeof rows, hipMemcpyDeviceToHost, ls));
        HIPCHECK(hipStreamSynchronize(ls));
        t.last_scan_bytes = rows * (int64_t)h->rows.d * (int64_t)h->rows.elsize;
        t.last_scan_flops = 2 * rows * (int64_t)h->rows.d;
    }
    *out = t;
    return 0;
    });
}

int ssmi_index_set_shadow_policy(ssmi_index* h, int policy) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (policy != 0 && policy != 1) return fail("shadow policy must be 0 (never) or 1 (on demand)");
    DeviceGuard g(h->device);
    quiesce(h);
    h->shadow_policy = policy;
    h->shadow_failed = false;
    if (policy == 0 && h->has_shadow) {  // give the memory back
        HIPCHECK(hipDeviceSynchronize());
        h->shadow16.destroy();
        h->has_shadow = false;
    }
    return 0;
    });
}

int ssmi_index_get_shadow_state(const ssmi_index* h, int* policy, int64_t* shadow_rows, int* failed) {
    if (!h || !policy || !shadow_rows || !failed) return fail("NULL argument");
    *policy = h->shadow_policy;
    *shadow_rows = h->has_shadow ? h->shadow16.nrows : 0;
    *failed = h->shadow_failed ? 1 : 0;
    return 0;
}

int ssmi_index_get_shadow_dtype(const ssmi_index* h, int* out) {
    if (!h || !out) return fail("NULL argument");
    *out = h->has_shadow ? h->shadow16.dtype : -1;
    return 0;
}

int ssmi_index_get_last_kernel(const ssmi_index* h, const char** out) {
    return guarded([&]() -> int {
    if (!h || !out) return fail("NULL argument");
    *out = h->last_kernel;
    return 0;
    });
}

int ssmi_index_get_kernel_clock(const ssmi_index* h, double* mhz_out, int64_t* cycles_out, double* ms_out) {
    return guarded([&]() -> int {
    if (!h || !mhz_out || !cycles_out || !ms_out) return fail("NULL argument");
    if (!h->last_clk) return fail("the last search did not run the 256-query MFMA kernel");
    DeviceGuard g(h->device);
    unsigned long long st[4];
    // the search may have run on a caller's stream: wait for that one, then copy
    if (h->have_last_search && hipStreamSynchronize(h->last_search_stream) != hipSuccess) {
        (void)hipGetLastError();            // that stream no longer exists
        HIPCHECK(hipDeviceSynchronize());
    }
    HIPCHECK(hipMemcpy(st, h->last_clk, sizeof(st), hipMemcpyDeviceToHost));
    if (st[1] <= st[3] || st[0] <= st[2]) return fail("no clock stamps yet");
    const double us = (double)(st[1] - st[3]) / 100.0;   // s_memrealtime ticks at 100 MHz
    *cycles_out = (int64_t)(st[0] - st[2]);
    *ms_out = us / 1000.0;
    *mhz_out = (double)(st[0] - st[2]) / us;
    return 0;
    });
}

int ssmi_index_set_timing(ssmi_index* h, int enabled) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    h->timing_on = enabled != 0;
    return 0;
    });
}

// durations (ms) of the dominant kernel of the last `max_n` searches, oldest first
int ssmi_index_get_scan_times(const ssmi_index* h, int max_n, float* ms_out, int* n_out) {
    return guarded([&]() -> int {
    if (!h || !ms_out || !n_out) return fail("NULL argument");
    DeviceGuard g(h->device);
    long long n = std::min<long long>(std::min<long long>(max_n, h->n_searches), ssmi_index::kEvRing);
    for (long long i = 0; i < n; ++i) {
        hipEvent_t const* ev = h->ev[(h->n_searches - n + i) % ssmi_index::kEvRing];
        HIPCHECK(hipEventSynchronize(ev[1]));
        HIPCHECK(hipEventElapsedTime(&ms_out[i], ev[0], ev[1]));
    }
    *n_out = (int)n;
    return 0;
    });
}

}  // extern "C"
