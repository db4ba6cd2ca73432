// static_search.hip - ssmi_index_search_static: the Flat search for callers that cannot allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
//
// Same kernels as ssmi_index_search's exact streaming path (K1 + merge tail, scan_impl.h /
// tail_merge.h); the only difference is where the kernels' scratch lives.  ssmi_index_search keeps
// the per-block partial k-lists and the merge tail's ticket counters in grow-only per-handle
// workspaces (hipMalloc on first use / growth); this entry point keeps them in ONE file-scope
// device array of fixed size.  Every GPU the library runs on holds its own copy of that array at
// its own address until the process ends; the address is looked up once per device
// (static_scratch_setup, called with that device current from the handle constructors) and only
// that pointer - never the array's name, which on the host is a stand-in and not a device address -
// is handed to kernels and memsets.
//
// No counterpart in the reference (FAISS allocates freely inside search).
#include "../../include/ssmi.h"
#include "kernels.h"
#include "index.h"

#include <mutex>

using namespace ssmi;

namespace {

constexpr size_t kCounterBytes = (size_t)ssmi_index::kTailCounters * sizeof(unsigned);  // 16 KiB
constexpr size_t kPartialBytes = (size_t)SSMI_STATIC_SCRATCH_BYTES - kCounterBytes;
constexpr int kMaxDevices = 64;

// [ticket counters: kTailCounters x u32][partial k-lists: nq x grid_x x k keys]
static __device__ __align__(256) unsigned char scratch[SSMI_STATIC_SCRATCH_BYTES];

std::mutex g_mu;
void* g_scratch_ptr[kMaxDevices] = {};

}  // namespace

namespace ssmi {
// Once per device, with that device current: resolve the device address of this GPU's copy of the array.
int static_scratch_setup(int device) {
    if (device < 0 || device >= kMaxDevices) return fail("device %d out of range for the static scratch table", device);
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_scratch_ptr[device]) return 0;
    void* p = nullptr;
    HIPCHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
    if (!p) return fail("hipGetSymbolAddress returned NULL for the static scratch on device %d", device);
    g_scratch_ptr[device] = p;
    return 0;
}
}  // namespace ssmi

extern "C" {

int ssmi_static_scratch_bytes(size_t* out) {
    return guarded([&]() -> int {
    if (!out) return fail("out is NULL");
    *out = (size_t)SSMI_STATIC_SCRATCH_BYTES;
    return 0;
    });
}

int ssmi_index_search_static(ssmi_index* h, int64_t nq, const float* q_dev, int k, float* D_dev, int64_t* I_dev,
                             int64_t id_offset, void* stream) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (h->kind != SSMI_KIND_FLAT) return fail("search_static: Flat indexes only");
    if (nq < 0) return fail("negative nq");
    if (nq == 0) return 0;
    if (!q_dev || !D_dev || !I_dev) return fail("NULL buffer");
    if (k < 1) return fail("k must be >= 1");
    if (k > kMaxK) return fail("k=%d exceeds the supported maximum %d", k, kMaxK);
    const RowStore& R = h->rows;
    if (R.ld != R.d)
        return fail("search_static: dimension %d is not a multiple of 8 (the queries would need a padded staging copy)", R.d);
    DeviceGuard g(h->device);
    if (!g.ok) return fail("hipSetDevice(%d) failed", h->device);
    void* base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        base = g_scratch_ptr[h->device];
    }
    if (!base) return fail("search_static: static scratch of device %d was never set up", h->device);
    hipStream_t s = (hipStream_t)stream;
    // host bookkeeping + at most one event wait (allocates nothing): add / write / free on this handle wait for this
    // launch like for any other search (quiesce), and a search still running on another stream stays ordered
    CHECK(order_search_on(h, s));
    if (R.nrows == 0) {
        HIPCHECK(launch_fill_empty(D_dev, I_dev, nq * k, s));
        return 0;
    }
    if (nq > (1ll << 20)) return fail("search_static: nq=%lld exceeds the maximum 1048576", (long long)nq);
    ScanPlan p = {};
    std::string err;
    if (!plan_flat_scan(R.dtype, R.ld, R.nrows, (int)nq, k, h->num_cu, &p, &err)) return fail("%s", err.c_str());
    const size_t need = (size_t)nq * p.grid_x * k * sizeof(skey_t);
    if (need > kPartialBytes)
        return fail("search_static: nq=%lld, k=%d over %d row slices needs %zu bytes of partial lists; the static scratch "
                    "holds %zu (use ssmi_index_search or fewer queries per call)",
                    (long long)nq, k, p.grid_x, need, kPartialBytes);
    if (p.grid_y > ssmi_index::kTailCounters)
        return fail("search_static: %d query tiles exceed the %d ticket counters of the static scratch", p.grid_y,
                    ssmi_index::kTailCounters);
    unsigned* counters = reinterpret_cast<unsigned*>(base);
    skey_t* partial = reinterpret_cast<skey_t*>(static_cast<unsigned char*>(base) + kCounterBytes);
    // the array keeps the previous call's contents: tickets must start at zero (a finished launch re-arms its own,
    // but this entry point does not rely on the previous user of the array having finished cleanly)
    HIPCHECK(hipMemsetAsync(counters, 0, (size_t)p.grid_y * sizeof(unsigned), s));
    p.tail = TailMerge{counters, D_dev, I_dev, id_offset, nullptr, nullptr, nullptr, nullptr, 0};
    HIPCHECK(launch_flat_scan(R.dtype, R.view(), q_dev, (int)nq, k, partial, p, s));
    return 0;
    });
}

}  // extern "C"
