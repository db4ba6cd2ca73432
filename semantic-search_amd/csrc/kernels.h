// kernels.h - host-callable launchers of the HIP kernels (internal to libssmi).
#pragma once
#include "common.h"
#include "tail_merge.h"

namespace ssmi {

struct ScanPlan {
    int bt;        // queries per block pass
    int r;         // per-wave list length / 64
    int nch;       // static chunks per row (0 = runtime loop)
    bool qreg;     // query in registers
    bool nt;       // non-temporal row loads
    bool mm;       // K1m (scan_mm.h): four rows per wave pass, multiply-adds on the matrix pipe
    int mm_g;      // K1m: 16-byte loads per lane and half stage (8 or 4)
    int threads;
    int grid_x;    // row slices == partial lists per query
    int grid_y;    // query tiles
    size_t lds;    // dynamic LDS bytes
    TailMerge tail = {nullptr, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0};  // counter set: the scan writes (D, I) itself, no K3 launch
};

// K1 planning + launch
bool plan_flat_scan(int dtype, int ld, long long nrows, int nq, int k, int num_cu, ScanPlan* p, std::string* err);
hipError_t launch_flat_scan_f32(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s);
hipError_t launch_flat_scan_f16(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s);
hipError_t launch_flat_scan_bf16(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s);
inline hipError_t launch_flat_scan(int dtype, const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s) {
    return dtype == 0 ? launch_flat_scan_f32(X, Q, nq, k, partial, p, s)
         : dtype == 1 ? launch_flat_scan_f16(X, Q, nq, k, partial, p, s) : launch_flat_scan_bf16(X, Q, nq, k, partial, p, s);
}

// K2 planning + launch (fp16 or bf16 storage, ld % 64 == 0)
struct MfmaPlan {
    int nqt, nslices, grid;
    int gtop_m;   // scores every slice publishes per query for the cross-slice threshold bound (0: none)
    bool narrow;  // <= 128 queries: the HBM-bound 128-query-tile stream (mfma_n128.hip)
    long long rows_per_slice;
    int q_pitch;  // halves per row of the 16-bit query workspace (ld, or ld + 64 for the 256-query kernel)
    int dtype;    // SSMI_F16 or SSMI_BF16: the rows' type; queries are narrowed to the same type
    size_t qh_bytes, cand_bytes, lds;
    size_t clk_off;  // 256-query kernel: byte offset of its four clock stamps inside the query workspace (0: none)
    const char* kernel;  // which contraction kernel launch_flat_mfma will run
};
bool plan_flat_mfma(int dtype, int ld, long long nrows, int nq, int k, int num_cu, MfmaPlan* p);
// Q: (nq, ld) fp32 on device; qh/cand: workspaces of p.qh_bytes / p.cand_bytes; partial: [nq][p.nslices][k]
hipError_t launch_flat_mfma(const RowView& X, const float* Q, int nq, int k, const MfmaPlan& p, void* qh, void* cand,
                            skey_t* partial, hipStream_t s, unsigned min_ord = 0);

// (nq, ld) fp32 queries -> (rows_out >= nq, ld) fp16 (bf16 = false) or bf16, zero rows beyond nq
hipError_t launch_cvt_queries_16(const float* q, int nq, int rows_out, int ld, void* out, bool bf16, hipStream_t s);

// K3: merge P sorted k-lists of keys per query -> (D, I)
hipError_t launch_merge_keys(const skey_t* partial, int nq, int P, int k, float* D, int64_t* I,
                             int64_t id_offset, hipStream_t s, unsigned min_ord = 0);
// K3': merge `nlists` (D,I) result sets, each (nq,k) row-major, list l at Dl + l*d_stride / Il + l*i_stride
hipError_t launch_merge_lists(const float* Dl, long long d_stride, const int64_t* Il, long long i_stride, int nlists,
                              int nq, int k, float* D, int64_t* I, hipStream_t s);

// K4: rows (n,d) of src dtype -> storage rows [row0, row0+n) with zero padding to ld
hipError_t launch_cvt_store(const void* src, int src_dtype, long long n, int d, const RowView& X, int dst_dtype,
                            void* const* segs_rw, long long row0, hipStream_t s);
// K9: gather rows by id -> fp32 (n,d); ids out of range produce NaN-free zeros and set *bad
hipError_t launch_gather_rows(const RowView& X, int dtype, const int64_t* ids, long long n, int d, float* out,
                              int* bad, hipStream_t s);
// candidates x negatives inner products from the row store: out (n, n_neg) fp32
hipError_t launch_score_rows(const RowView& X, int dtype, const int64_t* ids, long long n, int d, const float* neg,
                             int n_neg, float* out, int* bad, hipStream_t s);
// K10: append unit-norm gaussian rows [row0,row0+n)
hipError_t launch_fill_random_unit(const RowView& X, int dtype, void* const* segs_rw, long long row0, long long n,
                                   int d, uint64_t seed, hipStream_t s);
// fp32 store rows -> 16-bit shadow store rows (dst_dtype 1 = fp16, 2 = bf16); *inexact (device int, may be null) is
// set when some value does not survive the narrowing exactly
hipError_t launch_shadow_rows(const RowView& src, void* const* dsegs, int dshift, int dst_dtype, long long row0, long long n,
                              int* inexact, hipStream_t s);
// fill (nq,k) outputs with the "no result" padding
hipError_t launch_fill_empty(float* D, int64_t* I, long long n, hipStream_t s);
// pad queries (nq,d) -> (nq,ld) fp32 (zero tail)
hipError_t launch_pad_queries(const float* q, int nq, int d, int ld, float* out, hipStream_t s);

// K7b (ivf_bucket.hip): vectors bucketed by inverted list - histogram, exclusive prefix sum (off), stable scatter
// (perm: members of every list in input order; rank: position of every vector among its list's vectors of this batch)
static constexpr int64_t kBucketSkip = -2;  // label of a vector that is left out (list owned by another rank)
static constexpr int kBucketSmall = 1024;    // batches up to this size are ranked by one workgroup (launch_ivf_bucket_small)
size_t ivf_bucket_ws_bytes(long long n, int nlist, int* chunk_out = nullptr);
// rank / last: n ints; touched: n+1 int2 ((list, count) at the last vector of every list, (-1, 0) elsewhere; [n].x = bad flag)
hipError_t launch_ivf_bucket_small(const int64_t* labels, int n, int nlist, int* rank, int* last, int2* touched, hipStream_t s);
hipError_t launch_ivf_bucket(const int64_t* labels, long long n, int nlist, void* ws, int* off, int* perm, int* rank,
                             const int** total_out, hipStream_t s);

}  // namespace ssmi
