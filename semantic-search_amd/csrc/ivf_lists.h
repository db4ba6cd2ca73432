// ivf_lists.h - launcher of the list-major IVF scan (ivf_lists.hip), called from ivf.hip.
#pragma once
#include "common.h"

namespace ssmi {

struct IvfListScanArgs {
    RowView X;                  // slot store
    const float* Q;             // (nq, ld) fp32
    const int64_t* Ic;          // (nq, nprobe) probed lists
    int nq, nprobe, nlist, k;
    const int* page_off;        // nlist+1
    const int* pages;           // CSR payload
    const int* list_size;       // nlist
    const uint32_t* slot_ids;
    // workspaces
    int* cnt;                   // nlist probes per list, followed by ...
    int* cursor;                // ... nlist fill cursors (cursor == cnt + nlist) and ONE work counter (cnt[2*nlist]),
                                // all zeroed together
    int* eoff;                  // nlist+1
    int* uoff;                  // nlist+1
    int* entries;               // nq*nprobe (query*nprobe + probe), grouped by list
    skey_t* partial;            // [nq*nprobe][k]
    unsigned* qthr;             // MFMA form: [nq] highest k-th score any probed list of the query has reached (f32_to_ord
                                // bits, 0 = none), zeroed before the launch; nullptr: every (query, list) pair on its own
    int bt, threads, max_units;
    size_t lds;
};

// queries per work item, block size and LDS bytes for this row length and k; false when a query does not fit LDS
bool plan_ivf_list_scan(int dtype, int ld, int k, int* bt, int* threads, size_t* lds);
hipError_t launch_ivf_probe_table(const IvfListScanArgs& A, hipStream_t s);
hipError_t launch_ivf_list_scan(const IvfListScanArgs& A, int dtype, hipStream_t s);

// MFMA form (ivf_lists_mfma.hip): fp16 or bf16 lists, rows of a multiple of 512 elements, k <= 256
bool ivf_list_mfma_supported(int dtype, int ld, int k);
int ivf_list_mfma_bt(long long nq, int nprobe, int nlist);  // queries per work item: 16, 64 or 128
size_t ivf_list_mfma_cand_bytes(int num_cu);
// A.bt from ivf_list_mfma_bt; qh: (nq, ld) queries narrowed to the lists' 16-bit format (bf16 says which); cand:
// ivf_list_mfma_cand_bytes workspace
hipError_t launch_ivf_list_scan_mfma(const IvfListScanArgs& A, const void* qh, void* cand, int num_cu, bool bf16, hipStream_t s);

}  // namespace ssmi
