// index.h - host-side index objects behind the opaque ssmi_index handle.
#pragma once
#include "../../include/ssmi.h"
#include "common.h"
#include <exception>
#include <new>
#include <vector>
#include <string>

namespace ssmi {

static const int kMaxSegs = 4096;
static const int kWsSlots = 16;

int fail(const char* fmt, ...);

// Device row store: equally sized segments (~1 GiB) addressed through a device
// pointer table, so an index grows without ever copying more than segment 0.
// Rows are padded to a multiple of 8 elements with zeros (16-byte row alignment).
struct RowStore {
    int d = 0, ld = 0, dtype = 0, seg_shift = 0;
    size_t elsize = 0, row_bytes = 0;
    long long nrows = 0;
    long long seg0_cap = 0;  // rows allocated in segment 0 (grows by doubling)
    std::vector<void*> segs;
    void** d_segs = nullptr;

    int init(int d, int dtype, hipStream_t stream);  // `stream`: where the segment table is zeroed (see zero_device_sync)
    void destroy();
    int reserve(long long total_rows, hipStream_t stream);
    RowView view() const;
};

}  // namespace ssmi

struct ssmi_index {
    int kind = 0;
    int device = 0;
    int num_cu = 256;
    ssmi::RowStore rows;  // Flat: the database.  IVF: the slot store (pages of list members)
    // fp16 copy of an fp32 Flat database, built lazily by the first large-batch search so that
    // it can run on the MFMA kernel (costs +50 % HBM; batch-1..8 searches keep using the fp32 rows)
    ssmi::RowStore shadow16;
    bool has_shadow = false;
    // 1 (default): large batches over fp32 rows may build and use the shadow; 0: never (every batch stays on the
    // exact fp32 scan).  ssmi_index_set_shadow_policy; env SSMI_NO_SHADOW=1 overrides to 0.
    int shadow_policy = 1;
    bool shadow_failed = false;  // the shadow did not fit in HBM: searches fell back to the exact scan (reset by add)
    // element type of the shadow: SSMI_F16, SSMI_BF16, or -1 = auto (bf16 while every row is bf16-valued, i.e. the
    // shadow is lossless; fp16 otherwise).  env SSMI_SHADOW_DTYPE=f16|bf16|auto
    int shadow_pref = -1;
    int* d_flag = nullptr;       // 256 bytes of device scratch for one-word answers (the shadow's "lossy" flag)
    hipStream_t stream = nullptr;
    // Searches with device buffers return after ENQUEUEING on the caller's stream, while the workspaces, the IVF
    // page tables and the ticket counters belong to the handle.  The handle therefore remembers the stream of
    // its last search: a search on another stream is ordered behind it (one event, recorded only at the
    // switch), and every call that changes shared state first waits for it (quiesce).
    hipStream_t last_search_stream = nullptr;
    bool have_last_search = false;
    hipEvent_t ev_order = nullptr;
    // ring of (scan start, scan stop, merge stop) event triples, one per search call
    static const int kEvRing = 256;
    hipEvent_t ev[kEvRing][3] = {};
    long long n_searches = 0;  // searches that recorded an event triple
    void* ws[ssmi::kWsSlots] = {};
    size_t ws_bytes[ssmi::kWsSlots] = {};
    ssmi_timing timing = {};
    const unsigned long long* last_clk = nullptr;  // clock stamps of the last 256-query MFMA launch (device), or nullptr
    const char* last_kernel = "";  // name of the dominant kernel of the last search (ssmi_index_get_last_kernel)
    bool timing_valid = false;
    bool timing_on = true;  // record the event triple of every search (ssmi_index_set_timing)
    bool events_valid = false;  // the last search recorded its events
    bool merge_event = false;   // ... including the third one (a separate merge launch followed the scan)
    // ticket counters of the merge tails (tail_merge.h): zero between launches, one per query (tile)
    static const int kTailCounters = 4096;
    unsigned* tail_counters = nullptr;
    // pinned host staging of an all-host search (numpy queries in, numpy results out: the reference's call shape):
    // copies from and to pageable memory cost ~20 us each, pinned ones ~5 us; grow-only, up to kPinMax bytes
    void* pin = nullptr;
    size_t pin_bytes = 0;
    static const size_t kPinMax = (size_t)4 << 20;

    // ---- IVF state (kind == SSMI_KIND_IVFFLAT) ----
    int nlist = 0, nprobe = 1;
    bool trained = false;
    long long ivf_ntotal = 0;
    long long ivf_slots_used = 0;  // slots handed out in `rows` (pages * page_rows)
    void* ivf = nullptr;           // ssmi::IvfState*
};

// Zero device memory ON `stream` and wait for it.  Not hipMemset: that runs on the NULL stream, and the handles' own
// streams are non-blocking, i.e. not ordered against it - with a caller's search still in flight on the NULL stream
// (torch's default stream) the zeros would land AFTER what the handle's stream wrote into the same buffer in the
// meantime (found by tests/test_fuzz_api_gpu.py: a new index's segment table zeroed behind its first add).
inline hipError_t zero_device_sync(void* p, size_t bytes, hipStream_t stream) {
    const hipError_t e = hipMemsetAsync(p, 0, bytes, stream);
    return e != hipSuccess ? e : hipStreamSynchronize(stream);
}

#define HIPCHECK(expr)                                                                             \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess) return fail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
    } while (0)
#define CHECK(expr)        \
    do {                   \
        int rc__ = (expr); \
        if (rc__) return rc__; \
    } while (0)

// Every extern "C" body runs inside guarded(): a C++ exception (std::bad_alloc from a corrupt size field, a
// std::length_error, ...) must not cross the C boundary - ctypes would see std::terminate and the process abort,
// where faiss.read_index & co. raise a Python exception.  It becomes an error code + message instead.
template <typename F>
inline int guarded(F&& f) noexcept {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return ssmi::fail("out of host memory");
    } catch (const std::exception& e) {
        return ssmi::fail("%s", e.what());
    } catch (...) {
        return ssmi::fail("unknown C++ exception");
    }
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};


namespace ssmi {
int store_add(ssmi_index* h, RowStore& R, int64_t n, const void* x, int src_dtype, int mem_kind);
// grow-only device workspace slots (0: staging, 1: queries, 2: partial lists, 3: host-output staging, 4+: IVF)
int ensure_ws(ssmi_index* h, int slot, size_t bytes);
// query staging shared by the flat and IVF paths: returns (nq, ld) fp32 on device
int stage_queries(ssmi_index* h, int64_t nq, const float* q, int q_mem_kind, hipStream_t s, const float** out);

// IVF coarse quantizer: a scan that merges its own lists (tail_merge.h) can also write the probe plan of every
// query (exclusive prefix sums of the probed lists' sizes and their page-table offsets); `done` reports whether
// it did, otherwise the caller runs the planning kernel
struct ProbePlanRequest {
    const int* sizes;
    long long* prefix;
    const int* page_off;
    int* po;
    bool done;
};
int store_search(ssmi_index* h, const RowStore& R, int64_t nq, const float* qdev, int k, float* Ddev, int64_t* Idev,
                 int64_t id_offset, hipStream_t s, bool record, ProbePlanRequest* plan = nullptr, unsigned min_ord = 0);

// static_search.hip: resolves this device's copy of the static scratch array (once per device, device current)
int static_scratch_setup(int device);

// stream ordering of the calls on one handle (see ssmi_index::last_search_stream)
int order_search_on(ssmi_index* h, hipStream_t s);
void quiesce(ssmi_index* h);

// IVF entry points (ivf.hip)
int ivf_create(ssmi_index* h, int nlist);
void ivf_destroy(ssmi_index* h);
int ivf_train(ssmi_index* h, int64_t n, const float* x, int mem_kind);
int ivf_add(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind);
int ivf_add_shard(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, int64_t id0, int shard, int nshards);
int ivf_add_random_unit(ssmi_index* h, int64_t n, uint64_t seed);
int ivf_search(ssmi_index* h, int64_t nq, const float* qdev, int k, float* Ddev, int64_t* Idev, int64_t id_offset,
               hipStream_t s, unsigned min_ord = 0);
int ivf_ids_to_slots(ssmi_index* h, int64_t n, const int64_t* ids, std::vector<int64_t>* slots);
}  // namespace ssmi
