// ivf_state.h - host-side state of an IVF-Flat index (internal to libssmi; shared by ivf.hip and persist.hip).
#pragma once
#include "index.h"
#include <vector>

namespace ssmi {

static const int kPageRows = 32;
static const int kPageShift = 5;

struct IvfState {
    RowStore cent32, cent16;
    std::vector<std::vector<int>> pages;  // per list: page numbers in order
    std::vector<long long> size;          // per list: vectors stored
    long long npages = 0;
    uint32_t* d_slot_ids = nullptr;
    long long slot_cap = 0;
    int* d_page_off = nullptr;  // nlist + 1
    int* d_size = nullptr;      // nlist
    int* d_pages = nullptr;     // npages (CSR payload)
    long long d_pages_cap = 0;
    bool csr_dirty = true;
    bool has_direct_map = false;
    std::vector<int64_t> direct_map;  // id -> slot
    // incremental persistence (persist.hip): the file this index was last written to / read from in full, and
    // how much of every list that file plus its journal already hold
    std::string persist_path;              // empty: no full write or read yet
    long long persist_base_ntotal = -1;    // ntotal in the base file's header
    unsigned long long persist_base_bytes = 0;  // size of the base file
    long long persist_journal_rows = 0;    // vectors the journal holds
    // bytes of "<path>.journal" known to be a valid header + complete records for THIS base file (0: no journal, a
    // foreign one, or one not looked at yet).  Appends start exactly there: a torn tail or a stale journal left by a
    // crash is cut off / rewritten before anything is written behind it
    unsigned long long persist_journal_bytes = 0;
    std::vector<long long> persist_size;   // per list: members already on disk (base + journal)
    int niter = 10;                   // [FAISS-lib] ClusteringParameters default for the IVF quantizer
    int max_points_per_centroid = 256;
    uint64_t seed = 1234;
    // [FAISS-lib] IndexIVF's constructor makes the level-1 clustering spherical when the metric is inner product
    // (centroids L2-normalised after every update); this index is inner-product only.  env SSMI_IVF_SPHERICAL=0
    // keeps plain means (the round-1/2 behaviour)
    bool spherical = true;
};

inline IvfState* st(ssmi_index* h) { return reinterpret_cast<IvfState*>(h->ivf); }
inline const IvfState* st(const ssmi_index* h) { return reinterpret_cast<const IvfState*>(h->ivf); }

// ivf.hip, used by the file reader (persist.hip)
int rows_to_device_f32(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, const float** out);
int install_centroids(ssmi_index* h, const float* c, int mem_kind);
// append vectors (on the device as (n, ld) fp32) to given lists; ids == nullptr: sequential ids from ntotal
int ivf_append_preassigned(ssmi_index* h, int64_t n, const float* xdev, const int* labels, const uint32_t* ids_host);

}  // namespace ssmi
