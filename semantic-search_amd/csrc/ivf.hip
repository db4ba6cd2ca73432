// ivf.hip - IVF-Flat on MI355X: coarse quantizer, paged inverted lists, probe
// planning (prefix sum) and the bucketed list scan with fused top-k (K5-K8).
//
// Stands behind faiss.IndexIVFFlat(IndexFlatIP(d), d, nlist, METRIC_INNER_PRODUCT)
// as the reference builds it (core/index_factory.py:142-146) and drives it
// (train: core/index_factory.py:191; add: core/index_manager.py:299,319;
// search: core/index_manager.py:371).  Semantics restated from FAISS docs
// [FAISS-lib]: k-means centroids (plain means, inner-product assignment), a
// vector goes to the list of its max-inner-product centroid, a search visits the
// nprobe best lists and ranks their members by exact inner product.
//
// Layout in HBM
//   centroids   two small row stores (fp32 for exact small-batch assignment via
//               K1, fp16 for large batches via the MFMA kernel K2)
//   list rows   one slot store (`h->rows`) cut into pages of 32 slots; a list is
//               a host-side vector of page numbers (mirrored to the device as a
//               CSR), so appending never moves existing vectors
//   slot_ids    uint32 id of the vector in each slot
// Search (ivf_search)
//   per-query form (batches whose probes do not revisit lists), two launches for a batch-1 search:
//     coarse top-nprobe over the centroids (K1; its merge tail also writes the probe plan: exclusive prefix
//     sums of the probed lists' sizes and their page-table offsets)  ->  ivf_scan_qreg_kernel: the probed rows
//     of a query form one virtual range [0, T) split evenly over all waves of the grid (row granular, so short
//     and long lists balance); query and probe plan in registers, K1's coalesced-row / register-top-k loop,
//     merge tail (tail_merge.h).  ivf_scan_topk_kernel is the generic form for other row lengths.
//   list-major form (batches that revisit lists): ivf_lists.hip (register FMA) / ivf_lists_mfma.hip (matrix
//     cores; fp32 lists through an fp16 shadow of the slot store)  ->  K3 merges the nprobe lists of a query.
// Persistence lives in persist.hip, the host-side state in ivf_state.h.
#include "index.h"
#include "kernels.h"
#include "scan_impl.h"

#include "ivf_lists.h"
#include "ivf_state.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <random>

using namespace ssmi;

namespace ssmi {


// ---------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------

// (n,d) rows of S -> (n,ld) fp32 with zero tail
template <typename S>
__global__ void rows_to_f32_kernel(const S* __restrict__ src, int d, int ld, float* __restrict__ out) {
    const size_t i = blockIdx.x;
    for (int e = threadIdx.x; e < ld; e += blockDim.x) out[i * ld + e] = (e < d) ? (float)src[i * d + e] : 0.f;
}

// rows (n,ld) fp32 -> slot store rows dst_slot[i] (negative: the row is not stored); records the id of each slot
template <typename T>
__global__ void scatter_rows_kernel(const float* __restrict__ src, int ld, const int* __restrict__ dst_slot,
                                    void* const* segs, int seg_shift, uint32_t* __restrict__ slot_ids, uint32_t id0,
                                    const uint32_t* __restrict__ ids) {
    const size_t i = blockIdx.x;
    const long long slot = dst_slot[i];
    if (slot < 0) return;
    const long long mask = (1ll << seg_shift) - 1;
    T* dst = reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(segs[slot >> seg_shift]) +
                                  (size_t)(slot & mask) * (size_t)ld * sizeof(T));
    for (int e = threadIdx.x; e < ld; e += blockDim.x) dst[e] = (T)src[i * ld + e];
    if (threadIdx.x == 0) slot_ids[slot] = ids ? ids[i] : id0 + (uint32_t)i;
}

// list-sharded add: vectors whose list belongs to another rank are left out of the bucketing
__global__ void mask_shard_labels_kernel(int64_t* __restrict__ labels, long long n, int nlist, int shard, int nshards) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = labels[i];
    if (l >= 0 && l < nlist && (int)(l % nshards) != shard) labels[i] = kBucketSkip;
}

// int32 labels from the host -> the int64 labels the bucketing reads
__global__ void widen_labels_kernel(const int* __restrict__ in, long long n, int64_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

// slot of every new vector from its rank among the batch's vectors of its list.  fill[l] = {first free slot in the
// list's last page, free slots there, first of the consecutive pages this add gave the list}
// (the fill entry of vector i is that of its list, or of entry[i] when the table is per batch: small batches)
__global__ void slots_from_rank_kernel(const int64_t* __restrict__ labels, const int* __restrict__ entry,
                                       const int* __restrict__ rank, long long n, const int* __restrict__ fill,
                                       int* __restrict__ dst_slot) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = rank[i];
    if (r < 0) { dst_slot[i] = -1; return; }
    const int* f = fill + 3 * (size_t)(entry ? entry[i] : labels[i]);
    const int room = f[1];
    const int over = r - room;
    dst_slot[i] = over < 0 ? f[0] + r : (f[2] + (over >> kPageShift)) * kPageRows + (over & (kPageRows - 1));
}

// K6: centroid l = mean of its members (fixed summation order -> deterministic)
__global__ void centroid_update_kernel(const float* __restrict__ X, int ld, const int* __restrict__ perm,
                                       const int* __restrict__ off, void* const* segs32, int shift32,
                                       void* const* segs16, int shift16) {
    const int l = blockIdx.x;
    const int b = off[l], e = off[l + 1];
    if (e == b) return;  // empty cluster: handled on the host by splitting a big one
    float* c32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(segs32[l >> shift32]) +
                                          (size_t)(l & ((1 << shift32) - 1)) * (size_t)ld * 4);
    __half* c16 = reinterpret_cast<__half*>(reinterpret_cast<unsigned char*>(segs16[l >> shift16]) +
                                            (size_t)(l & ((1 << shift16) - 1)) * (size_t)ld * 2);
    const float inv = 1.0f / (float)(e - b);
    for (int x = threadIdx.x; x < ld; x += blockDim.x) {
        float acc = 0.f;
        for (int i = b; i < e; ++i) acc += X[(size_t)perm[i] * ld + x];
        acc *= inv;
        c32[x] = acc;
        c16[x] = __float2half(acc);
    }
}

// [FAISS-lib] empty-cluster fix: copy a populated centroid and nudge the pair apart
__global__ void split_centroid_kernel(int ld, void* const* segs32, int shift32, void* const* segs16, int shift16,
                                      int dst, int src, float eps) {
    float* s32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(segs32[src >> shift32]) +
                                          (size_t)(src & ((1 << shift32) - 1)) * (size_t)ld * 4);
    float* d32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(segs32[dst >> shift32]) +
                                          (size_t)(dst & ((1 << shift32) - 1)) * (size_t)ld * 4);
    __half* s16 = reinterpret_cast<__half*>(reinterpret_cast<unsigned char*>(segs16[src >> shift16]) +
                                            (size_t)(src & ((1 << shift16) - 1)) * (size_t)ld * 2);
    __half* d16 = reinterpret_cast<__half*>(reinterpret_cast<unsigned char*>(segs16[dst >> shift16]) +
                                            (size_t)(dst & ((1 << shift16) - 1)) * (size_t)ld * 2);
    for (int x = threadIdx.x; x < ld; x += blockDim.x) {
        const float v = s32[x];
        const float a = (x & 1) ? v * (1.f + eps) : v * (1.f - eps);
        const float b = (x & 1) ? v * (1.f - eps) : v * (1.f + eps);
        d32[x] = a; d16[x] = __float2half(a);
        s32[x] = b; s16[x] = __float2half(b);
    }
}

// [FAISS-lib] spherical k-means: IndexIVF sets ClusteringParameters::spherical for METRIC_INNER_PRODUCT, and
// Clustering::post_process_centroids then L2-normalises every centroid (after the update and the empty-cluster
// split of each iteration, and once after the initialisation); a zero centroid is left alone.  One block per centroid.
__global__ void renorm_centroids_kernel(int ld, void* const* segs32, int shift32, void* const* segs16, int shift16) {
    __shared__ float wsum[4];
    const int l = blockIdx.x;
    float* c32 = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(segs32[l >> shift32]) +
                                          (size_t)(l & ((1 << shift32) - 1)) * (size_t)ld * 4);
    __half* c16 = reinterpret_cast<__half*>(reinterpret_cast<unsigned char*>(segs16[l >> shift16]) +
                                            (size_t)(l & ((1 << shift16) - 1)) * (size_t)ld * 2);
    float ss = 0.f;
    for (int x = threadIdx.x; x < ld; x += blockDim.x) ss = fmaf(c32[x], c32[x], ss);
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float tot = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
    if (!(tot > 0.f)) return;
    const float inv = 1.0f / sqrtf(tot);
    for (int x = threadIdx.x; x < ld; x += blockDim.x) {
        const float v = c32[x] * inv;
        c32[x] = v;
        c16[x] = __float2half(v);
    }
}

// K7 (search side): per query, exclusive prefix sum of the probed lists' sizes.
// prefix: (nq, nprobe+1) int64; one 256-thread block per query.
__global__ void ivf_plan_kernel(const int64_t* __restrict__ Ic, int nprobe, const int* __restrict__ list_size,
                                long long* __restrict__ prefix, const int* __restrict__ page_off, int* __restrict__ po) {
    __shared__ long long wsum[4];
    __shared__ long long carry;
    const int q = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t* ic = Ic + (size_t)q * nprobe;
    long long* pf = prefix + (size_t)q * (nprobe + 1);
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nprobe; base += 256) {
        const int j = base + threadIdx.x;
        long long v = 0;
        if (j < nprobe) {
            const int64_t l = ic[j];
            v = (l >= 0) ? (long long)list_size[l] : 0;
            po[(size_t)q * nprobe + j] = (l >= 0) ? page_off[l] : 0;
        }
        long long incl = v;  // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long up = (long long)shfl_up64((skey_t)incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        long long woff = carry;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (j < nprobe) pf[j] = woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) pf[nprobe] = carry;
}

struct IvfScanParams {
    RowView X;  // slot store
    const float* Q;             // (nq, ld) fp32
    const int64_t* Ic;          // (nq, nprobe) probed lists, best first
    const long long* prefix;    // (nq, nprobe+1)
    const int* po;              // (nq, nprobe) page-table offsets of the probed lists (page_off[Ic])
    int nsegs;                  // entries of X.segs
#ifdef SSMI_IVF_STAMP
    unsigned long long* dbg;    // diagnostic build: [wave][4] wall-clock stamps (100 MHz)
#endif
    const int* page_off;        // nlist+1
    const int* pages;           // CSR payload
    const uint32_t* slot_ids;
    int nprobe, k;
    skey_t* partial;            // [nq][gridDim.x][k]
    TailMerge tail;             // counter set: the last block of a query merges its lists into (D, I)
};

// K8: bucketed list scan with fused per-wavefront top-k.  grid = (gx, nq), 4 waves.
template <typename T, int R>
__global__ void __launch_bounds__(256) ivf_scan_topk_kernel(IvfScanParams P) {
    constexpr int V = ElemTraits<T>::V;
    extern __shared__ __align__(16) unsigned char smem[];
    float* qs = reinterpret_cast<float*>(smem);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int q = blockIdx.y;
    const int ld = P.X.ld;
    const int nch = (ld + 64 * V - 1) / (64 * V);
    const int ldq = nch * 64 * V;
    const int row_bytes = ld * (int)sizeof(T);
    for (int e = threadIdx.x; e < ldq; e += blockDim.x)
        qs[q_lds_index<V>(e)] = (e < ld) ? P.Q[(size_t)q * ld + e] : 0.f;
    __syncthreads();

    const long long* pf = P.prefix + (size_t)q * (P.nprobe + 1);
    const int64_t* ic = P.Ic + (size_t)q * P.nprobe;
    const long long total = pf[P.nprobe];
    const long long nw = (long long)gridDim.x * 4, wv = (long long)blockIdx.x * 4 + wave;
    long long r = total * wv / nw;
    const long long r_hi = total * (wv + 1) / nw;
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    WaveTopK<R> top;
    top.init();
    const skey_t thr_floor = floor_key(P.tail.min_ord);  // the search's score floor (0: none)
    const float thr_floor_s = P.tail.min_ord ? ord_to_f32(P.tail.min_ord) : SSMI_NEG_MAX;
    skey_t thr = thr_floor;
    float thr_s = thr_floor_s;

    if (r < r_hi) {
        int lo = 0, hi = P.nprobe - 1;  // last j with pf[j] <= r
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pf[mid] <= r) lo = mid; else hi = mid - 1;
        }
        int j = lo;
        while (r < r_hi) {
            while (pf[j + 1] <= r) ++j;
            const long long base = pf[j];
            const long long lend = pf[j + 1] < r_hi ? pf[j + 1] : r_hi;
            const int* lp = P.pages + P.page_off[ic[j]];
            for (; r < lend; ++r) {
                const long long o = r - base;
                const long long slot = (long long)lp[o >> kPageShift] * kPageRows + (o & (kPageRows - 1));
                const unsigned char* rowp = reinterpret_cast<const unsigned char*>(P.X.segs[slot >> P.X.seg_shift]) +
                                            (size_t)(slot & seg_mask) * (size_t)row_bytes;
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                constexpr int G = 8;
#pragma unroll 1
                for (int c0 = 0; c0 < nch; c0 += G) {
                    uint4 x[G];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const int off = ((c0 + g) * 64 + lane) * 16;
                        x[g] = make_uint4(0, 0, 0, 0);
                        if (off < row_bytes) x[g] = load16_nt(rowp + off);
                    }
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        if (c0 + g < nch) {
                            float xf[V];
                            Unpack<T>::run(x[g], xf);
#pragma unroll
                            for (int p = 0; p < V / 4; ++p) {
                                const float4 t4 = *reinterpret_cast<const float4*>(qs + (c0 + g) * 64 * V + p * 256 + lane * 4);
                                a0 = fmaf(xf[4 * p], t4.x, a0);
                                a1 = fmaf(xf[4 * p + 1], t4.y, a1);
                                a2 = fmaf(xf[4 * p + 2], t4.z, a2);
                                a3 = fmaf(xf[4 * p + 3], t4.w, a3);
                            }
                        }
                    }
                }
                float s = wave_sum((a0 + a1) + (a2 + a3));
                s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));
                if (s >= thr_s) {  // cheap float gate; the exact (score, id) order needs the id
                    const skey_t key = make_key(s, P.slot_ids[slot]);
                    if (key > thr) {
                        top.insert(key);
                        thr = kmax(top.kth(P.k), thr_floor);
                        thr_s = (thr == 0) ? SSMI_NEG_MAX : fmaxf(key_score(thr), thr_floor_s);
                    }
                }
            }
        }
    }
    __syncthreads();
    skey_t* lds_keys = reinterpret_cast<skey_t*>(smem);
    block_merge<R>(top, lds_keys, wave, 4);
    if (wave == 0) publish_list<R>(top, P.partial + ((size_t)q * gridDim.x + blockIdx.x) * P.k, P.k);
    if (P.tail.counter == nullptr) return;
    if (!tail_last_block(P.tail.counter + q, gridDim.x, reinterpret_cast<unsigned*>(lds_keys))) return;
    merge_partials<R, true>(P.partial + (size_t)q * gridDim.x * P.k, (int)gridDim.x, P.k, lds_keys, wave, 4,
                      P.tail.D + (size_t)q * P.k, P.tail.I + (size_t)q * P.k, P.tail.id_offset,
                      PlanOut{nullptr, nullptr, nullptr, nullptr}, P.tail.min_ord);
}

// K8, register-query form for rows of NCH KiB (d = 4096: NCH 8 for fp16, 16 for fp32; also 2048, 1024):
// the query lives in registers as in K1's batch-1 kernel, every chunk of a row is in flight before
// the first FMA, and the page
// number of a list position is looked up once per page, not once per row.  grid = (gx, nq),
// blockDim = 64 * nwaves; one k-list per block.
// ROWS = rows in flight per wave: 1 with 12 waves per CU (K1's shape), 2 with 8 waves per CU (a 768-thread block
// with two rows in flight needs more than its 170 registers per lane: it spilled the query registers and
// reloaded them between the row loads, which serialised the loads: 5.3 TB/s instead of 6.7).
template <typename T, int NCH, int R, int ROWS>
__global__ void __launch_bounds__((R <= 4 && ROWS == 1 && NCH * ElemTraits<T>::V <= 64) ? 768 : 512) ivf_scan_qreg_kernel(IvfScanParams P) {
    constexpr int V = ElemTraits<T>::V;
    extern __shared__ __align__(16) unsigned char smem[];
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int q = blockIdx.y;
    const int ld = P.X.ld;
    constexpr int row_bytes = NCH * 1024;

#ifdef SSMI_IVF_STAMP
    unsigned long long st0 = wall_clock64(), st1 = 0, st2 = 0;
#endif
    float qreg[NCH][V];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const float4 a = *reinterpret_cast<const float4*>(P.Q + (size_t)q * ld + (c * 64 + lane) * V);
        qreg[c][0] = a.x; qreg[c][1] = a.y; qreg[c][2] = a.z; qreg[c][3] = a.w;
        if (V == 8) {
            const float4 b = *reinterpret_cast<const float4*>(P.Q + (size_t)q * ld + (c * 64 + lane) * V + 4);
            qreg[c][V - 4] = b.x; qreg[c][V - 3] = b.y; qreg[c][V - 2] = b.z; qreg[c][V - 1] = b.w;
        }
    }

    // segment base pointers -> LDS: a wave-uniform lookup then costs an LDS read instead of a memory round trip
    // in front of every page's row loads
    const unsigned char** seg_l = reinterpret_cast<const unsigned char**>(smem + (size_t)nwaves * 64 * R * sizeof(skey_t));
    for (int i = threadIdx.x; i < P.nsegs; i += blockDim.x) seg_l[i] = reinterpret_cast<const unsigned char*>(P.X.segs[i]);
    const long long* pf = P.prefix + (size_t)q * (P.nprobe + 1);
    const int* po = P.po + (size_t)q * P.nprobe;
    const long long total = pf[P.nprobe];
    __syncthreads();
    // every wave owns a contiguous range of the probed rows (the waves of a block interleaving over a
    // block range instead measured 30 % slower: fewer loads in flight per page chunk)
    const long long nw = (long long)gridDim.x * nwaves, wv = (long long)blockIdx.x * nwaves + wave;
    long long r = total * wv / nw;
    const long long r_hi = total * (wv + 1) / nw;
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    WaveTopK<R> top;
    top.init();
    const skey_t thr_floor = floor_key(P.tail.min_ord);  // the search's score floor (0: none)
    const float thr_floor_s = P.tail.min_ord ? ord_to_f32(P.tail.min_ord) : SSMI_NEG_MAX;
    skey_t thr = thr_floor;
    float thr_s = thr_floor_s;

    auto consider = [&](float s, uint32_t id) {
        if (s >= thr_s) {  // cheap float gate
            const skey_t key = make_key(s, id);
            if (key > thr) {
                top.insert(key);
                thr = kmax(top.kth(P.k), thr_floor);
                thr_s = (thr == 0) ? SSMI_NEG_MAX : fmaxf(key_score(thr), thr_floor_s);
            }
        }
    };
    auto dot = [&](const uint4 (&x)[NCH]) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float xf[V];
            Unpack<T>::run(x[c], xf);
#pragma unroll
            for (int j = 0; j < V; j += 4) {
                a0 = fmaf(xf[j], qreg[c][j], a0);
                a1 = fmaf(xf[j + 1], qreg[c][j + 1], a1);
                a2 = fmaf(xf[j + 2], qreg[c][j + 2], a2);
                a3 = fmaf(xf[j + 3], qreg[c][j + 3], a3);
            }
        }
        float s = wave_sum((a0 + a1) + (a2 + a3));
        return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));
    };

    if (r < r_hi) {
        // The probe table of the query lives in registers, 64 probes at a time: lane i holds the
        // prefix pf[c0+i], pf[c0+i+1] and the CSR offset of list ic[c0+i].  Finding the list of a
        // position is then one ballot instead of a chain of dependent global loads.
        int c0 = -64;
        long long pf_lo = 0, pf_hi = 0;
        int po_l = 0;
        int j = 0;
        while (r < r_hi) {
            if (c0 < 0 || r >= __shfl(pf_hi, 63)) {
                // (re)load the 64-probe window that contains position r
                do {
                    c0 += 64;
                    const int jj = c0 + lane;
                    const bool in = jj < P.nprobe;
                    pf_lo = in ? pf[jj] : total;
                    pf_hi = in ? pf[jj + 1] : total;
                    po_l = in ? po[jj] : 0;
                } while (c0 + 64 < P.nprobe && r >= __shfl(pf_hi, 63));
            }
            // list of position r: the lane whose [pf_lo, pf_hi) contains it (empty lists never match)
            const unsigned long long hit = __builtin_amdgcn_ballot_w64(pf_lo <= r && r < pf_hi);
            if (hit == 0) break;  // cannot happen: r < total lies in exactly one non-empty list
            j = __builtin_ctzll(hit);
            const long long base = (long long)readlane64((skey_t)pf_lo, j);
            const long long lnext = (long long)readlane64((skey_t)pf_hi, j);
            const long long lend = lnext < r_hi ? lnext : r_hi;
            const int* lp = P.pages + __builtin_amdgcn_readlane(po_l, j);
            const int np_list = (int)((lnext - base + kPageRows - 1) >> kPageShift);
            int pi = (int)((r - base) >> kPageShift);
            int pg = lp[pi];
#ifdef SSMI_IVF_STAMP
            if (st1 == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); st1 = wall_clock64(); }
#endif
            while (r < lend) {
                // the next page's number travels under this page's row loads: no dependent load at the page switch
                const int pg_next = (pi + 1 < np_list) ? lp[pi + 1] : 0;
                const long long o = r - base;
                const int in_page = (int)(o & (kPageRows - 1));
                long long cnt = kPageRows - in_page;
                if (cnt > lend - r) cnt = lend - r;
                const long long slot0 = (long long)pg * kPageRows + in_page;
                // a page never straddles a segment (segment sizes are multiples of the page size)
                const unsigned char* rowp = seg_l[slot0 >> P.X.seg_shift] +
                                            (size_t)(slot0 & seg_mask) * (size_t)row_bytes + (size_t)lane * 16;
                // ids of the page's rows: one coalesced load up front instead of a dependent load per
                // candidate (a wave sees few rows, so most of them enter its still-short list)
                const int ids_l = (lane < (int)cnt) ? (int)P.slot_ids[slot0 + lane] : 0;
                int i = 0;
                if (ROWS == 2) {
                    for (; i + 2 <= (int)cnt; i += 2) {
                        uint4 x0[NCH], x1[NCH];
#pragma unroll
                        for (int c = 0; c < NCH; ++c) x0[c] = load16_nt_g(rowp + (size_t)i * row_bytes + c * 1024);
#pragma unroll
                        for (int c = 0; c < NCH; ++c) x1[c] = load16_nt_g(rowp + (size_t)(i + 1) * row_bytes + c * 1024);
                        const float s0 = dot(x0);
                        const float s1 = dot(x1);
                        consider(s0, (uint32_t)__builtin_amdgcn_readlane(ids_l, i));
                        consider(s1, (uint32_t)__builtin_amdgcn_readlane(ids_l, i + 1));
                    }
                }
                for (; i < (int)cnt; ++i) {
                    uint4 x0[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) x0[c] = load16_nt_g(rowp + (size_t)i * row_bytes + c * 1024);
                    consider(dot(x0), (uint32_t)__builtin_amdgcn_readlane(ids_l, i));
                }
                r += cnt;
                ++pi;
                pg = pg_next;
            }
        }
    }
#ifdef SSMI_IVF_STAMP
    st2 = wall_clock64();
#endif
    skey_t* lds_keys = reinterpret_cast<skey_t*>(smem);
    block_merge<R>(top, lds_keys, wave, nwaves);
    if (wave == 0) publish_list<R>(top, P.partial + ((size_t)q * gridDim.x + blockIdx.x) * P.k, P.k);
#ifdef SSMI_IVF_STAMP
    if (lane == 0) {
        unsigned long long* d = P.dbg + ((size_t)blockIdx.x * nwaves + wave) * 4;
        d[0] = st0; d[1] = st1 ? st1 : st2; d[2] = st2; d[3] = wall_clock64();
    }
#endif
    if (P.tail.counter == nullptr) return;
    if (!tail_last_block(P.tail.counter + q, gridDim.x, reinterpret_cast<unsigned*>(lds_keys))) return;
    merge_partials<R, true>(P.partial + (size_t)q * gridDim.x * P.k, (int)gridDim.x, P.k, lds_keys, wave, nwaves,
                      P.tail.D + (size_t)q * P.k, P.tail.I + (size_t)q * P.k, P.tail.id_offset,
                      PlanOut{nullptr, nullptr, nullptr, nullptr}, P.tail.min_ord);
#ifdef SSMI_IVF_STAMP
    if (threadIdx.x == 0) P.dbg[(size_t)gridDim.x * nwaves * 4] = wall_clock64();
#endif
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

int ivf_create(ssmi_index* h, int nlist) {
    if (nlist < 1 || nlist > (1 << 24)) return fail("nlist %d out of range [1, 2^24]", nlist);
    IvfState* S = new IvfState();
    h->ivf = S;
    h->nlist = nlist;
    h->nprobe = 1;  // FAISS default
    h->trained = false;
    if (const char* e = getenv("SSMI_IVF_SPHERICAL")) S->spherical = e[0] != '0';
    if (S->cent32.init(h->rows.d, SSMI_F32, h->stream)) return 1;
    if (S->cent16.init(h->rows.d, SSMI_F16, h->stream)) return 1;
    S->pages.assign(nlist, {});
    S->size.assign(nlist, 0);
    HIPCHECK(hipMalloc((void**)&S->d_page_off, sizeof(int) * (nlist + 1)));
    HIPCHECK(hipMalloc((void**)&S->d_size, sizeof(int) * nlist));
    HIPCHECK(zero_device_sync(S->d_page_off, sizeof(int) * (nlist + 1), h->stream));
    HIPCHECK(zero_device_sync(S->d_size, sizeof(int) * nlist, h->stream));
    return 0;
}

void ivf_destroy(ssmi_index* h) {
    IvfState* S = st(h);
    if (!S) return;
    S->cent32.destroy();
    S->cent16.destroy();
    if (S->d_slot_ids) (void)hipFree(S->d_slot_ids);
    if (S->d_page_off) (void)hipFree(S->d_page_off);
    if (S->d_size) (void)hipFree(S->d_size);
    if (S->d_pages) (void)hipFree(S->d_pages);
    delete S;
    h->ivf = nullptr;
}

// (n,d) rows of src_dtype, host or device -> (n, ld) fp32 on the device (workspace slot 8 when a copy is needed)
int rows_to_device_f32(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, const float** out) {
    const RowStore& R = h->rows;
    if (mem_kind == SSMI_DEVICE && src_dtype == SSMI_F32 && R.ld == R.d) { *out = (const float*)x; return 0; }
    CHECK(ensure_ws(h, 8, (size_t)n * R.ld * 4));
    float* dst = (float*)h->ws[8];
    if (src_dtype == SSMI_F32 && R.ld == R.d) {
        HIPCHECK(hipMemcpyAsync(dst, x, (size_t)n * R.d * 4, hipMemcpyHostToDevice, h->stream));
        *out = dst;
        return 0;
    }
    const size_t el = (size_t)dtype_elsize(src_dtype);
    const void* src = x;
    if (mem_kind == SSMI_HOST) {
        CHECK(ensure_ws(h, 0, (size_t)n * R.d * el));
        HIPCHECK(hipMemcpyAsync(h->ws[0], x, (size_t)n * R.d * el, hipMemcpyHostToDevice, h->stream));
        src = h->ws[0];
    }
    if (src_dtype == SSMI_F32)
        hipLaunchKernelGGL((rows_to_f32_kernel<float>), dim3((unsigned)n), dim3(256), 0, h->stream, (const float*)src, R.d, R.ld, dst);
    else if (src_dtype == SSMI_F16)
        hipLaunchKernelGGL((rows_to_f32_kernel<__half>), dim3((unsigned)n), dim3(256), 0, h->stream, (const __half*)src, R.d, R.ld, dst);
    else
        hipLaunchKernelGGL((rows_to_f32_kernel<bf16_t>), dim3((unsigned)n), dim3(256), 0, h->stream, (const bf16_t*)src, R.d, R.ld, dst);
    HIPCHECK(hipGetLastError());
    *out = dst;
    return 0;
}

// smallest batch whose coarse quantizer runs on the fp16 centroids (MFMA kernels); below it the exact fp32 scan
static int coarse_mfma_min() {
    const char* e = getenv("SSMI_IVF_COARSE_MFMA_MIN");
    const int v = e ? atoi(e) : 9;
    return v < 1 ? 1 : v;
}

// K5: nearest centroid (max inner product) of n device rows -> int64 labels on the device (default: workspace 7)
static int assign_to_centroids(ssmi_index* h, int64_t n, const float* xdev, int64_t** labels_dev, int64_t* into = nullptr) {
    IvfState* S = st(h);
    const RowStore& C = (n >= coarse_mfma_min() && (S->cent16.ld % 64) == 0) ? S->cent16 : S->cent32;
    CHECK(ensure_ws(h, 6, (size_t)n * 4));
    if (!into) {
        CHECK(ensure_ws(h, 7, (size_t)n * 8));
        into = (int64_t*)h->ws[7];
    }
    CHECK(store_search(h, C, n, xdev, 1, (float*)h->ws[6], into, 0, h->stream, false));
    if (labels_dev) *labels_dev = into;
    return 0;
}

// ... and read back (ssmi_index_assign)
static int assign_to_centroids_host(ssmi_index* h, int64_t n, const float* xdev, std::vector<int>* labels) {
    int64_t* dev = nullptr;
    CHECK(assign_to_centroids(h, n, xdev, &dev));
    std::vector<int64_t> tmp((size_t)n);
    HIPCHECK(hipMemcpyAsync(tmp.data(), dev, (size_t)n * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    labels->resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (tmp[i] < 0 || tmp[i] >= h->nlist) return fail("internal: centroid assignment produced list %lld", (long long)tmp[i]);
        (*labels)[i] = (int)tmp[i];
    }
    return 0;
}

// The quantizer changed (trained, centroids replaced): a base file written before describes another index, so the next
// incremental checkpoint must be a full write, not a journal record on top of it (found by
// tests/test_fuzz_manager_gpu.py: saved while buffering, trained, saved again, reloaded untrained).
static void forget_persisted_state(IvfState* S) {
    S->persist_path.clear();
    S->persist_base_ntotal = -1;
    S->persist_journal_rows = 0;
}

int install_centroids(ssmi_index* h, const float* c, int mem_kind) {
    IvfState* S = st(h);
    S->cent32.nrows = 0;
    S->cent16.nrows = 0;
    CHECK(store_add(h, S->cent32, h->nlist, c, SSMI_F32, mem_kind));
    CHECK(store_add(h, S->cent16, h->nlist, c, SSMI_F32, mem_kind));
    h->trained = true;
    forget_persisted_state(S);
    return 0;
}

int ivf_train(ssmi_index* h, int64_t n, const float* x, int mem_kind) {
    IvfState* S = st(h);
    if (h->trained) return 0;  // like FAISS: training a trained index is a no-op
    if (n < 1 || !x) return fail("train: no training vectors");
    const int nlist = h->nlist;
    if (n < nlist) return fail("train: number of training points (%lld) should be at least as large as number of clusters (%d)", (long long)n, nlist);
    const RowStore& R = h->rows;
    const int d = R.d, ld = R.ld;
    std::mt19937_64 rng(S->seed);

    // [FAISS-lib] subsample to at most max_points_per_centroid * nlist training points
    int64_t nt = n;
    std::vector<int64_t> pick;
    const int64_t cap = (int64_t)S->max_points_per_centroid * nlist;
    if (n > cap) {
        pick.resize((size_t)n);
        std::iota(pick.begin(), pick.end(), (int64_t)0);
        for (int64_t i = 0; i < cap; ++i) std::swap(pick[i], pick[i + (int64_t)(rng() % (uint64_t)(n - i))]);
        pick.resize((size_t)cap);
        std::sort(pick.begin(), pick.end());
        nt = cap;
    }
    // training set as (nt, ld) fp32 on the device (own allocation: it outlives workspace reuse)
    float* X = nullptr;
    HIPCHECK(hipMalloc((void**)&X, (size_t)nt * ld * 4));
    struct Free { float* p; ~Free() { if (p) (void)hipFree(p); } } freeX{X};
    if (ld != d) HIPCHECK(hipMemsetAsync(X, 0, (size_t)nt * ld * 4, h->stream));
    const hipMemcpyKind kind = mem_kind == SSMI_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (pick.empty()) {
        HIPCHECK(hipMemcpy2DAsync(X, (size_t)ld * 4, x, (size_t)d * 4, (size_t)d * 4, (size_t)nt, kind, h->stream));
    } else {
        for (int64_t i = 0; i < nt;) {  // runs of consecutive picks copy together
            int64_t j = i + 1;
            while (j < nt && pick[j] == pick[j - 1] + 1) ++j;
            HIPCHECK(hipMemcpy2DAsync(X + (size_t)i * ld, (size_t)ld * 4, x + (size_t)pick[i] * d, (size_t)d * 4,
                                      (size_t)d * 4, (size_t)(j - i), kind, h->stream));
            i = j;
        }
    }
    HIPCHECK(hipStreamSynchronize(h->stream));

    // initial centroids: nlist distinct random training points
    std::vector<int64_t> perm((size_t)nt);
    std::iota(perm.begin(), perm.end(), (int64_t)0);
    for (int i = 0; i < nlist; ++i) std::swap(perm[i], perm[i + (int64_t)(rng() % (uint64_t)(nt - i))]);
    float* C0 = nullptr;
    HIPCHECK(hipMalloc((void**)&C0, (size_t)nlist * d * 4));
    struct Free2 { float* p; ~Free2() { if (p) (void)hipFree(p); } } freeC{C0};
    for (int i = 0; i < nlist; ++i)
        HIPCHECK(hipMemcpyAsync(C0 + (size_t)i * d, X + (size_t)perm[i] * ld, (size_t)d * 4, hipMemcpyDeviceToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    CHECK(install_centroids(h, C0, SSMI_DEVICE));
    h->trained = false;
    auto renorm = [&]() -> int {
        if (!S->spherical) return 0;
        hipLaunchKernelGGL(renorm_centroids_kernel, dim3(nlist), dim3(256), 0, h->stream, ld, (void* const*)S->cent32.d_segs,
                           S->cent32.seg_shift, (void* const*)S->cent16.d_segs, S->cent16.seg_shift);
        HIPCHECK(hipGetLastError());
        return 0;
    };
    CHECK(renorm());

    // per iteration: K5 labels -> K7b buckets (perm, off) -> K6 means, all on the device; the host only sees the
    // nlist list sizes (for the empty-cluster fix)
    int *d_perm = nullptr, *d_off = nullptr;
    int64_t* d_labels = nullptr;
    HIPCHECK(hipMalloc((void**)&d_perm, (size_t)nt * 4));
    struct Free3 { int* p; ~Free3() { if (p) (void)hipFree(p); } } freeP{d_perm};
    HIPCHECK(hipMalloc((void**)&d_off, (size_t)(nlist + 1) * 4));
    struct Free4 { int* p; ~Free4() { if (p) (void)hipFree(p); } } freeO{d_off};
    HIPCHECK(hipMalloc((void**)&d_labels, (size_t)nt * 8));
    struct Free5 { int64_t* p; ~Free5() { if (p) (void)hipFree(p); } } freeL{d_labels};
    CHECK(ensure_ws(h, 15, ivf_bucket_ws_bytes(nt, nlist)));

    std::vector<int> count((size_t)nlist + 1);
    for (int it = 0; it < S->niter; ++it) {
        const int64_t chunk = 1 << 16;  // bounds the query workspace of the assignment
        for (int64_t b = 0; b < nt; b += chunk)
            CHECK(assign_to_centroids(h, std::min(chunk, nt - b), X + (size_t)b * ld, nullptr, d_labels + b));
        const int* d_total = nullptr;
        HIPCHECK(launch_ivf_bucket(d_labels, nt, nlist, h->ws[15], d_off, d_perm, nullptr, &d_total, h->stream));
        hipLaunchKernelGGL(centroid_update_kernel, dim3(nlist), dim3(256), 0, h->stream, X, ld, d_perm, d_off,
                           (void* const*)S->cent32.d_segs, S->cent32.seg_shift, (void* const*)S->cent16.d_segs,
                           S->cent16.seg_shift);
        HIPCHECK(hipGetLastError());
        HIPCHECK(hipMemcpyAsync(count.data(), d_total, (size_t)(nlist + 1) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (count[nlist]) return fail("internal: centroid assignment produced a list outside [0, %d)", nlist);
        // empty clusters: split the currently largest cluster (deterministic variant of FAISS' fix)
        for (int l = 0; l < nlist; ++l) {
            if (count[l] != 0) continue;
            const int big = (int)(std::max_element(count.begin(), count.end() - 1) - count.begin());
            if (count[big] < 2) break;
            hipLaunchKernelGGL(split_centroid_kernel, dim3(1), dim3(256), 0, h->stream, ld, (void* const*)S->cent32.d_segs,
                               S->cent32.seg_shift, (void* const*)S->cent16.d_segs, S->cent16.seg_shift, l, big,
                               1.0f / 1024.0f);
            HIPCHECK(hipGetLastError());
            count[l] = count[big] / 2;
            count[big] -= count[l];
        }
        CHECK(renorm());
        HIPCHECK(hipStreamSynchronize(h->stream));
    }
    h->trained = true;
    forget_persisted_state(S);
    return 0;
}

static int sync_csr(ssmi_index* h) {
    IvfState* S = st(h);
    if (!S->csr_dirty) return 0;
    const int nlist = h->nlist;
    std::vector<int> off((size_t)nlist + 1, 0), sz((size_t)nlist), flat;
    flat.reserve((size_t)S->npages);
    for (int l = 0; l < nlist; ++l) {
        off[l] = (int)flat.size();
        flat.insert(flat.end(), S->pages[l].begin(), S->pages[l].end());
        sz[l] = (int)S->size[l];
    }
    off[nlist] = (int)flat.size();
    if ((long long)flat.size() > S->d_pages_cap) {
        HIPCHECK(hipDeviceSynchronize());
        if (S->d_pages) HIPCHECK(hipFree(S->d_pages));
        S->d_pages_cap = std::max<long long>(1024, (long long)flat.size() * 2);
        HIPCHECK(hipMalloc((void**)&S->d_pages, (size_t)S->d_pages_cap * 4));
    }
    HIPCHECK(hipMemcpyAsync(S->d_page_off, off.data(), (size_t)(nlist + 1) * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipMemcpyAsync(S->d_size, sz.data(), (size_t)nlist * 4, hipMemcpyHostToDevice, h->stream));
    if (!flat.empty())
        HIPCHECK(hipMemcpyAsync(S->d_pages, flat.data(), flat.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    S->csr_dirty = false;
    return 0;
}

static int ensure_slots(ssmi_index* h, long long npages) {
    IvfState* S = st(h);
    const long long slots = npages * kPageRows;
    CHECK(h->rows.reserve(slots, h->stream));
    if (slots > S->slot_cap) {
        long long cap = std::max<long long>(4096, S->slot_cap);
        while (cap < slots) cap *= 2;
        uint32_t* p = nullptr;
        HIPCHECK(hipMalloc((void**)&p, (size_t)cap * 4));
        HIPCHECK(hipMemsetAsync(p, 0xff, (size_t)cap * 4, h->stream));
        if (S->d_slot_ids) {
            HIPCHECK(hipMemcpyAsync(p, S->d_slot_ids, (size_t)S->slot_cap * 4, hipMemcpyDeviceToDevice, h->stream));
            HIPCHECK(hipDeviceSynchronize());
            HIPCHECK(hipFree(S->d_slot_ids));
        }
        S->d_slot_ids = p;
        S->slot_cap = cap;
    }
    h->rows.nrows = slots;
    h->ivf_slots_used = slots;
    return 0;
}

// append vectors (on the device as (n, ld) fp32) to the lists their labels name (int64 on the device; kBucketSkip
// leaves a vector out).  Vector i takes the id ids_dev[i], or id0 + i.  K7b on the device ranks every vector inside
// its list; the host sees the nlist per-list counts, gives every list its pages (consecutive numbers per list and
// add) and hands back three ints per list from which a kernel derives every vector's slot.
// The host-side list state (sizes, page tables, direct map) is committed only when every step that can fail
// (slot store growth, workspace, launches) has succeeded; a failed add leaves the index as it was.
static int ivf_append_labelled(ssmi_index* h, int64_t n, const float* xdev, const int64_t* labels_dev, uint32_t id0,
                               const uint32_t* ids_dev, bool sequential_ids, int64_t* n_added) {
    if (n_added) *n_added = 0;
    if (n == 0) return 0;
    IvfState* S = st(h);
    const int nlist = h->nlist;
    struct Undo { int list; int new_pages; long long old_size; };
    std::vector<Undo> undo;
    const size_t dm_size0 = S->direct_map.size();
    const long long npages0 = S->npages;
    long long added = 0;
    // list l takes c more vectors: pages for them (consecutive numbers), and the three ints from which a kernel
    // derives the slot of the vector of rank r in this batch
    auto take = [&](int l, int c, int* f) {
        const long long pos = S->size[l];
        const int in_page = (int)(pos & (kPageRows - 1));
        const int room = in_page ? kPageRows - in_page : 0;
        const int new_pages = c > room ? (c - room + kPageRows - 1) >> kPageShift : 0;
        f[0] = room ? S->pages[l].back() * kPageRows + in_page : 0;
        f[1] = room;
        f[2] = (int)S->npages;
        undo.push_back({l, new_pages, pos});
        for (int p = 0; p < new_pages; ++p) S->pages[l].push_back((int)(S->npages + p));
        S->npages += new_pages;
        S->size[l] = pos + c;
        added += c;
    };
    const bool small = n <= kBucketSmall;
    int *d_rank, *d_slot, *d_fill, *d_entry = nullptr;
    std::vector<int> fill;
    if (small) {
        // workspace 9: [rank n][slot n][last n][fill 3n][touched n+1 int2]
        CHECK(ensure_ws(h, 9, (size_t)n * 4 * 8 + 8));
        d_rank = (int*)h->ws[9];
        d_slot = d_rank + n;
        d_entry = d_slot + n;
        d_fill = d_entry + n;
        int2* d_touched = (int2*)(d_fill + 3 * n);
        HIPCHECK(launch_ivf_bucket_small(labels_dev, (int)n, nlist, d_rank, d_entry, d_touched, h->stream));
        std::vector<int2> touched((size_t)n + 1);
        HIPCHECK(hipMemcpyAsync(touched.data(), d_touched, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (touched[(size_t)n].x) return fail("add: a vector names a list outside [0, %d)", nlist);
        fill.assign((size_t)n * 3, 0);
        for (int64_t i = 0; i < n; ++i)
            if (touched[(size_t)i].x >= 0) take(touched[(size_t)i].x, touched[(size_t)i].y, &fill[(size_t)i * 3]);
    } else {
        // workspace 15: [K7b tables][off nlist+1][fill 3*nlist]; workspace 9: [rank n][slot n]
        const size_t bucket_bytes = (ivf_bucket_ws_bytes(n, nlist) + 255) & ~(size_t)255;
        CHECK(ensure_ws(h, 15, bucket_bytes + ((size_t)nlist * 4 + 1) * 4));
        CHECK(ensure_ws(h, 9, (size_t)n * 8));
        int* d_off = (int*)((char*)h->ws[15] + bucket_bytes);
        d_fill = d_off + nlist + 1;
        d_rank = (int*)h->ws[9];
        d_slot = d_rank + n;
        const int* d_total = nullptr;
        HIPCHECK(launch_ivf_bucket(labels_dev, n, nlist, h->ws[15], d_off, nullptr, d_rank, &d_total, h->stream));
        std::vector<int> count((size_t)nlist + 1);
        HIPCHECK(hipMemcpyAsync(count.data(), d_total, (size_t)(nlist + 1) * 4, hipMemcpyDeviceToHost, h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        if (count[nlist]) return fail("add: a vector names a list outside [0, %d)", nlist);
        fill.assign((size_t)nlist * 3, 0);
        for (int l = 0; l < nlist; ++l)
            if (count[l]) take(l, count[l], &fill[(size_t)l * 3]);
    }
    auto rollback = [&]() {
        for (const Undo& u : undo) {
            S->size[u.list] = u.old_size;
            S->pages[u.list].resize(S->pages[u.list].size() - (size_t)u.new_pages);
        }
        S->npages = npages0;
        S->direct_map.resize(dm_size0);
    };
    auto body = [&]() -> int {
        if (S->npages * kPageRows >= (1ll << 31)) return fail("IVF slot store would exceed 2^31 slots");
        CHECK(ensure_slots(h, S->npages));
        HIPCHECK(hipMemcpyAsync(d_fill, fill.data(), fill.size() * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(slots_from_rank_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, labels_dev,
                           (const int*)d_entry, (const int*)d_rank, (long long)n, (const int*)d_fill, d_slot);
        HIPCHECK(hipGetLastError());
        const RowStore& R = h->rows;
        if (R.dtype == SSMI_F32)
            hipLaunchKernelGGL((scatter_rows_kernel<float>), dim3((unsigned)n), dim3(256), 0, h->stream, xdev, R.ld, (const int*)d_slot,
                               (void* const*)R.d_segs, R.seg_shift, S->d_slot_ids, id0, ids_dev);
        else if (R.dtype == SSMI_F16)
            hipLaunchKernelGGL((scatter_rows_kernel<__half>), dim3((unsigned)n), dim3(256), 0, h->stream, xdev, R.ld, (const int*)d_slot,
                               (void* const*)R.d_segs, R.seg_shift, S->d_slot_ids, id0, ids_dev);
        else
            hipLaunchKernelGGL((scatter_rows_kernel<bf16_t>), dim3((unsigned)n), dim3(256), 0, h->stream, xdev, R.ld, (const int*)d_slot,
                               (void* const*)R.d_segs, R.seg_shift, S->d_slot_ids, id0, ids_dev);
        HIPCHECK(hipGetLastError());
        if (h->has_shadow && R.dtype == SSMI_F32) {
            // the fp16 shadow of the slot store (built by the first batched search, see ivf_search) follows every add
            RowStore& H = h->shadow16;
            CHECK(H.reserve(R.nrows, h->stream));
            hipLaunchKernelGGL((scatter_rows_kernel<__half>), dim3((unsigned)n), dim3(256), 0, h->stream, xdev, R.ld, (const int*)d_slot,
                               (void* const*)H.d_segs, H.seg_shift, S->d_slot_ids, id0, ids_dev);
            HIPCHECK(hipGetLastError());
            H.nrows = R.nrows;
        }
        if (S->has_direct_map && sequential_ids) {  // id -> slot on the host: the one per-vector read-back
            std::vector<int> slots((size_t)n);
            HIPCHECK(hipMemcpyAsync(slots.data(), d_slot, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
            HIPCHECK(hipStreamSynchronize(h->stream));
            for (int64_t i = 0; i < n; ++i) S->direct_map.push_back(slots[(size_t)i]);
        }
        HIPCHECK(hipStreamSynchronize(h->stream));
        return 0;
    };
    if (body()) { rollback(); return 1; }
    h->ivf_ntotal += added;
    S->csr_dirty = true;
    if (n_added) *n_added = added;
    return 0;
}

// the same for labels and ids held by the host (file reader, journal replay, routed add)
int ivf_append_preassigned(ssmi_index* h, int64_t n, const float* xdev, const int* labels, const uint32_t* ids_host) {
    if (n == 0) return 0;
    // workspace 7: [labels int64 n][labels int32 n][ids n]
    CHECK(ensure_ws(h, 7, (size_t)n * 16));
    int64_t* lab64 = (int64_t*)h->ws[7];
    int* lab32 = (int*)(lab64 + n);
    uint32_t* ids_dev = ids_host ? (uint32_t*)(lab32 + n) : nullptr;
    HIPCHECK(hipMemcpyAsync(lab32, labels, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    if (ids_host) HIPCHECK(hipMemcpyAsync(ids_dev, ids_host, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(widen_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, (const int*)lab32,
                       (long long)n, lab64);
    HIPCHECK(hipGetLastError());
    return ivf_append_labelled(h, n, xdev, lab64, (uint32_t)h->ivf_ntotal, ids_dev, ids_host == nullptr, nullptr);
}

// place vectors (already on the device as (n, ld) fp32) into the lists of their nearest centroids
static int ivf_add_device_f32(ssmi_index* h, int64_t n, const float* xdev) {
    int64_t* labels = nullptr;
    CHECK(assign_to_centroids(h, n, xdev, &labels));
    return ivf_append_labelled(h, n, xdev, labels, (uint32_t)h->ivf_ntotal, nullptr, true, nullptr);
}

// list-sharded add: every rank sees the same vectors, assigns them all, and keeps those whose list l
// satisfies l % nshards == shard, under their GLOBAL ids id0 + i
int ivf_add_shard(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, int64_t id0, int shard, int nshards) {
    if (!h->trained) return fail("add on an untrained IVF index (call train first)");
    if (nshards < 1 || shard < 0 || shard >= nshards) return fail("bad shard %d of %d", shard, nshards);
    if (id0 < 0 || (unsigned long long)(id0 + n) >= 0xFFFFFFFFull) return fail("ids would exceed 2^32-2");
    if (st(h)->has_direct_map) return fail("a list-sharded index cannot maintain a direct map");
    const RowStore& R = h->rows;
    const size_t el = (size_t)dtype_elsize(src_dtype);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, ((size_t)1 << 30) / ((size_t)R.ld * 4)));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const float* xdev = nullptr;
        int64_t* labels = nullptr;
        CHECK(rows_to_device_f32(h, cnt, (const char*)x + (size_t)done * R.d * el, src_dtype, mem_kind, &xdev));
        CHECK(assign_to_centroids(h, cnt, xdev, &labels));
        hipLaunchKernelGGL(mask_shard_labels_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, labels,
                           (long long)cnt, h->nlist, shard, nshards);
        HIPCHECK(hipGetLastError());
        CHECK(ivf_append_labelled(h, cnt, xdev, labels, (uint32_t)(id0 + done), nullptr, false, nullptr));
    }
    return 0;
}

int ivf_add(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind) {
    if (!h->trained) return fail("add on an untrained IVF index (call train first)");
    if ((unsigned long long)(h->ivf_ntotal + n) >= 0xFFFFFFFFull) return fail("index would exceed 2^32-2 vectors");
    const RowStore& R = h->rows;
    const size_t el = (size_t)dtype_elsize(src_dtype);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, ((size_t)1 << 30) / ((size_t)R.ld * 4)));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const float* xdev = nullptr;
        CHECK(rows_to_device_f32(h, cnt, (const char*)x + (size_t)done * R.d * el, src_dtype, mem_kind, &xdev));
        CHECK(ivf_add_device_f32(h, cnt, xdev));
    }
    return 0;
}

int ivf_add_random_unit(ssmi_index* h, int64_t n, uint64_t seed) {
    if (!h->trained) return fail("add on an untrained IVF index (call train first)");
    // generate into a scratch flat store chunk by chunk, then route through the normal add path
    const RowStore& R = h->rows;
    RowStore tmp;
    if (tmp.init(R.d, SSMI_F32, h->stream)) return 1;
    struct D { RowStore* s; ~D() { s->destroy(); } } guard{&tmp};
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, ((size_t)1 << 30) / ((size_t)R.ld * 4)));
    if (chunk > (1ll << tmp.seg_shift)) return fail("internal: chunk exceeds scratch segment");
    CHECK(tmp.reserve(std::min<int64_t>(chunk, n), h->stream));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        tmp.nrows = 0;
        // rows are numbered by their final id so the data does not depend on the chunking
        HIPCHECK(launch_fill_random_unit(tmp.view(), SSMI_F32, (void* const*)tmp.d_segs, 0, cnt, R.d, seed + 0x9E3779B97F4A7C15ull * (uint64_t)(done / chunk + 1), h->stream));
        HIPCHECK(hipStreamSynchronize(h->stream));
        CHECK(ivf_add_device_f32(h, cnt, (const float*)tmp.segs[0]));
    }
    return 0;
}

template <typename T, int NCH, int R, int ROWS>
static hipError_t launch_qreg_rows(int gx, int nb, int threads, size_t lds, hipStream_t s, const IvfScanParams& P) {
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_qreg_kernel<T, NCH, R, ROWS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((ivf_scan_qreg_kernel<T, NCH, R, ROWS>), dim3(gx, nb), dim3(threads), lds, s, P);
    return hipGetLastError();
}
template <typename T, int NCH, int R>
static hipError_t launch_qreg(int rows, int gx, int nb, int threads, size_t lds, hipStream_t s, const IvfScanParams& P) {
    if constexpr (NCH <= 8 && R <= 8) {  // (16 KiB rows and k > 512 do not leave room for a second row)
        if (rows == 2) return launch_qreg_rows<T, NCH, R, 2>(gx, nb, threads, lds, s, P);
    }
    return launch_qreg_rows<T, NCH, R, 1>(gx, nb, threads, lds, s, P);
}
// threads per block of ivf_scan_qreg_kernel<T, NCH, R, ROWS> (its launch bound)
static int qreg_threads(int dtype, int row_kib, int r, int rows) {
    const int query_regs = dtype == SSMI_F32 ? row_kib * 4 : row_kib * 8;  // NCH * V
    return (r <= 4 && rows == 1 && query_regs <= 64) ? 768 : 512;
}

int ivf_search(ssmi_index* h, int64_t nq, const float* qdev, int k, float* Ddev, int64_t* Idev, int64_t id_offset,
               hipStream_t s, unsigned min_ord) {
    IvfState* S = st(h);
    const RowStore& R = h->rows;
    CHECK(sync_csr(h));
    const int nprobe = std::min(h->nprobe, h->nlist);
    if (nprobe > kMaxK) return fail("nprobe=%d exceeds the supported maximum %d", nprobe, kMaxK);
    int r = 1;
    while (64 * r < k) r <<= 1;
    // register-query kernel for rows of 1, 2, 4, 8 or 16 KiB (d = 4096 and its Matryoshka truncations)
    const int row_kib = (int)(R.row_bytes / 1024);
    const bool qreg = (R.row_bytes % 1024) == 0 && (row_kib == 1 || row_kib == 2 || row_kib == 4 || row_kib == 8 || row_kib == 16) &&
                      !getenv("SSMI_IVF_NO_QREG");
    // rows in flight per wave of the register-query kernel: 2 (8 waves per CU) or 1 (12 waves per CU)
    int qrows = 1;
    if (const char* e = getenv("SSMI_IVF_ROWS")) qrows = atoi(e) == 2 ? 2 : 1;
    if (row_kib > 8 || r > 8) qrows = 1;  // 16 KiB rows, k > 512: one row already fills the registers
    const int qthreads = qreg_threads(R.dtype, row_kib, r, qrows);
    int qbpc = 1;  // blocks per CU for one query
    if (const char* e = getenv("SSMI_IVF_BPC")) qbpc = atoi(e);
    // several queries in one launch share two rounds of blocks (2 x CUs); rounded DOWN per query, so that the launch never
    // spills into a third, nearly empty round (17 queries x 31 blocks = 527 ran 87 us per query against 77 us at 16 x 32)
    const int gx = qreg ? (int)std::max<int64_t>(1, std::min<int64_t>((int64_t)h->num_cu * qbpc, ((int64_t)h->num_cu * 2 * qbpc) / nq))
                        : (int)std::max<int64_t>(1, std::min<int64_t>(1024, ((int64_t)h->num_cu * 6 + nq - 1) / nq));
    const int V = R.dtype == SSMI_F32 ? 4 : 8;
    const int ldq = (R.ld + 64 * V - 1) / (64 * V) * 64 * V;
    const size_t lds = qreg ? (size_t)(qthreads / 64) * 64 * r * sizeof(skey_t) + R.segs.size() * sizeof(void*)
                            : std::max((size_t)ldq * 4, (size_t)4 * 64 * r * sizeof(skey_t));
    hipEvent_t* ev = h->ev[h->n_searches % ssmi_index::kEvRing];
    // batches whose probes revisit the lists (on average >= 2 probes per list) take the list-major scan; large indexes
    // already from one probe per three lists: the per-query scan costs the probed bytes once per query, the list-major one
    // the union of the probed lists plus a fixed part (probe-table inversion, work items; 0.8 ms at nlist 4096). Measured on
    // configs[3] (tools/time_ivf_batch_sweep.py, SSMI_SWEEP_AB=1): 32 queries 2.38 vs 2.52 ms, 48: 3.88 vs 3.35, 96: 7.26 vs
    // 5.15, 192: 14.3 vs 7.4 ms
    int lbt = 0, lthreads = 0;
    size_t llds = 0;
    const char* env_lm = getenv("SSMI_IVF_LISTMAJOR");  // tests: 0 = never, 1 = always (when it fits)
    const double per_query_bytes = (double)nq * nprobe / h->nlist * (double)R.nrows * R.ld * dtype_elsize(R.dtype);
    bool list_major = nq >= 16 && (nq * (int64_t)nprobe >= 2 * (int64_t)h->nlist ||
                                   (3 * nq * (int64_t)nprobe >= (int64_t)h->nlist && per_query_bytes >= 8e9));
    if (env_lm) list_major = env_lm[0] == '1';
    list_major = list_major && plan_ivf_list_scan(R.dtype, R.ld, k, &lbt, &lthreads, &llds);
    const char* env_lmm = getenv("SSMI_IVF_LIST_MFMA");  // 0: keep the register-FMA form
    // fp32 lists run the matrix-core scan on an fp16 shadow of the slot store (same policy as Flat: batches on
    // fp32 storage trade exact fp32 products for the MFMA path; +50 % HBM; SSMI_NO_SHADOW=1 keeps them exact)
    bool via_shadow = list_major && R.dtype == SSMI_F32 && h->shadow_policy != 0 && !h->shadow_failed && !getenv("SSMI_NO_SHADOW") &&
                      ivf_list_mfma_supported(SSMI_F16, R.ld, k) && !(env_lmm && env_lmm[0] == '0');
    if (via_shadow && !h->has_shadow) {
        auto bring_up = [&]() -> int {
            CHECK(h->shadow16.init(R.d, SSMI_F16, s));
            h->has_shadow = true;
            CHECK(h->shadow16.reserve(R.nrows, s));
            RowView src = R.view();
            HIPCHECK(launch_shadow_rows(src, (void* const*)h->shadow16.d_segs, h->shadow16.seg_shift, SSMI_F16, 0, R.nrows, nullptr, s));
            h->shadow16.nrows = R.nrows;
            return 0;
        };
        if (bring_up()) {  // no room for the shadow: not an error of the search, the exact register-FMA scan takes over
            (void)hipGetLastError();
            (void)hipStreamSynchronize(s);
            if (h->has_shadow) { h->shadow16.destroy(); h->has_shadow = false; }
            h->shadow_failed = true;
            via_shadow = false;
        }
    }
    const bool list_mfma = via_shadow || (list_major && ivf_list_mfma_supported(R.dtype, R.ld, k) && !(env_lmm && env_lmm[0] == '0'));
    const size_t per_q = list_major ? (size_t)nprobe * k * sizeof(skey_t) : (size_t)gx * k * sizeof(skey_t);
    const int64_t qbatch = std::max<int64_t>(1, std::min<int64_t>(nq, ((size_t)256 << 20) / per_q));
    for (int64_t done = 0; done < nq; done += qbatch) {
        const int nb = (int)std::min(qbatch, nq - done);
        const float* qb = qdev + (size_t)done * R.ld;
        // coarse quantizer: top-nprobe centroids per query
        CHECK(ensure_ws(h, 6, (size_t)nb * nprobe * 4));
        CHECK(ensure_ws(h, 7, (size_t)nb * nprobe * 8));
        CHECK(ensure_ws(h, 10, (size_t)nb * (nprobe + 1) * 8));
        CHECK(ensure_ws(h, 13, (size_t)nb * nprobe * 4));
        const RowStore& C = (nb >= coarse_mfma_min() && (S->cent16.ld % 64) == 0) ? S->cent16 : S->cent32;
        ProbePlanRequest plan = {S->d_size, (long long*)h->ws[10], S->d_page_off, (int*)h->ws[13], false};
        CHECK(store_search(h, C, nb, qb, nprobe, (float*)h->ws[6], (int64_t*)h->ws[7], 0, s, false, list_major ? nullptr : &plan));
        if (list_major) {
            CHECK(ensure_ws(h, 2, (size_t)nb * nprobe * k * sizeof(skey_t)));
            CHECK(ensure_ws(h, 11, (size_t)(4 * h->nlist + 3) * 4 + (size_t)nb * 4));
            CHECK(ensure_ws(h, 12, (size_t)nb * nprobe * 4));
            IvfListScanArgs A = {};
            A.X = R.view(); A.Q = qb; A.Ic = (const int64_t*)h->ws[7]; A.nq = nb; A.nprobe = nprobe; A.nlist = h->nlist;
            A.k = k; A.page_off = S->d_page_off; A.pages = S->d_pages; A.list_size = S->d_size; A.slot_ids = S->d_slot_ids;
            A.cnt = (int*)h->ws[11]; A.cursor = A.cnt + h->nlist; A.eoff = A.cursor + h->nlist + 1; A.uoff = A.eoff + h->nlist + 1;
            A.entries = (int*)h->ws[12]; A.partial = (skey_t*)h->ws[2];
            A.qthr = nullptr;
            if (list_mfma) {
                // matrix-core form: up to 128 queries per list pass, queries rounded to fp16
                A.bt = ivf_list_mfma_bt(nb, nprobe, h->nlist); A.threads = 512; A.lds = 0; A.max_units = 0;
                if (via_shadow) A.X = h->shadow16.view();
                CHECK(ensure_ws(h, 4, (size_t)nb * R.ld * 2));
                CHECK(ensure_ws(h, 5, ivf_list_mfma_cand_bytes(h->num_cu)));
                const bool lists_bf16 = !via_shadow && R.dtype == SSMI_BF16;  // (the shadow of fp32 lists is always fp16)
                HIPCHECK(launch_cvt_queries_16(qb, nb, nb, R.ld, h->ws[4], lists_bf16, s));
                if (!getenv("SSMI_IVF_LIST_SHARE_THR") || getenv("SSMI_IVF_LIST_SHARE_THR")[0] != '0') {
                    A.qthr = (unsigned*)(A.uoff + h->nlist + 1);  // behind the four tables of workspace 11
                    HIPCHECK(hipMemsetAsync(A.qthr, 0, (size_t)nb * 4, s));
                }
                if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[0], s));
                HIPCHECK(launch_ivf_list_scan_mfma(A, h->ws[4], h->ws[5], h->num_cu, lists_bf16, s));
                if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[1], s));
            } else {
                A.bt = lbt; A.threads = lthreads; A.lds = llds;
                A.max_units = (int)std::min<int64_t>((int64_t)nb * nprobe, (int64_t)nb * nprobe / lbt + h->nlist);
                if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[0], s));
                HIPCHECK(launch_ivf_list_scan(A, R.dtype, s));
                if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[1], s));
            }
            HIPCHECK(launch_merge_keys((const skey_t*)h->ws[2], nb, nprobe, k, Ddev + (size_t)done * k,
                                       Idev + (size_t)done * k, id_offset, s, min_ord));
            if (h->timing_on && done == 0) { HIPCHECK(hipEventRecord(ev[2], s)); h->merge_event = true; }
            continue;
        }
        if (!plan.done) {
            hipLaunchKernelGGL(ivf_plan_kernel, dim3(nb), dim3(256), 0, s, (const int64_t*)h->ws[7], nprobe, S->d_size,
                               (long long*)h->ws[10], S->d_page_off, (int*)h->ws[13]);
            HIPCHECK(hipGetLastError());
        }
        CHECK(ensure_ws(h, 2, (size_t)nb * gx * k * sizeof(skey_t)));
        IvfScanParams P = {};
        P.X = R.view(); P.Q = qb; P.Ic = (const int64_t*)h->ws[7]; P.prefix = (const long long*)h->ws[10];
        P.po = (const int*)h->ws[13]; P.nsegs = (int)R.segs.size();
#ifdef SSMI_IVF_STAMP
        CHECK(ensure_ws(h, 14, ((size_t)gx * (qthreads / 64) * 4 + 1) * 8));
        P.dbg = (unsigned long long*)h->ws[14];
#endif
        P.page_off = S->d_page_off; P.pages = S->d_pages; P.slot_ids = S->d_slot_ids; P.nprobe = nprobe; P.k = k;
        P.partial = (skey_t*)h->ws[2];
        const bool tail = h->tail_counters && nb <= ssmi_index::kTailCounters;
        P.tail = TailMerge{tail ? h->tail_counters : nullptr, Ddev + (size_t)done * k, Idev + (size_t)done * k, id_offset,
                           nullptr, nullptr, nullptr, nullptr, min_ord};
        if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[0], s));
#define IVF_LAUNCH(TT, RR)                                                                                       \
    case RR: {                                                                                                   \
        if (lds > 48 * 1024)                                                                                     \
            HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_topk_kernel<TT, RR>),           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL((ivf_scan_topk_kernel<TT, RR>), dim3(gx, nb), dim3(256), lds, s, P);                 \
        break;                                                                                                   \
    }
#define IVF_QLAUNCH(TT, NN, RR)                                                                                  \
    case RR: {                                                                                                   \
        HIPCHECK((launch_qreg<TT, NN, RR>(qrows, gx, nb, qthreads, lds, s, P)));                                 \
        break;                                                                                                   \
    }
#define IVF_QLAUNCH_R(TT, NN)                                                                                    \
    switch (r) { IVF_QLAUNCH(TT, NN, 1) IVF_QLAUNCH(TT, NN, 2) IVF_QLAUNCH(TT, NN, 4) IVF_QLAUNCH(TT, NN, 8)     \
                 IVF_QLAUNCH(TT, NN, 16) default: return fail("bad k"); }
        if (qreg && R.dtype == SSMI_F32) {
            switch (row_kib) {
                case 16: IVF_QLAUNCH_R(float, 16) break;
                case 8: IVF_QLAUNCH_R(float, 8) break;
                case 4: IVF_QLAUNCH_R(float, 4) break;
                case 2: IVF_QLAUNCH_R(float, 2) break;
                default: IVF_QLAUNCH_R(float, 1) break;
            }
        } else if (qreg && R.dtype == SSMI_F16) {
            switch (row_kib) {
                case 16: IVF_QLAUNCH_R(__half, 16) break;
                case 8: IVF_QLAUNCH_R(__half, 8) break;
                case 4: IVF_QLAUNCH_R(__half, 4) break;
                case 2: IVF_QLAUNCH_R(__half, 2) break;
                default: IVF_QLAUNCH_R(__half, 1) break;
            }
        } else if (qreg) {
            switch (row_kib) {
                case 16: IVF_QLAUNCH_R(bf16_t, 16) break;
                case 8: IVF_QLAUNCH_R(bf16_t, 8) break;
                case 4: IVF_QLAUNCH_R(bf16_t, 4) break;
                case 2: IVF_QLAUNCH_R(bf16_t, 2) break;
                default: IVF_QLAUNCH_R(bf16_t, 1) break;
            }
        } else if (R.dtype == SSMI_F32) {
            switch (r) { IVF_LAUNCH(float, 1) IVF_LAUNCH(float, 2) IVF_LAUNCH(float, 4) IVF_LAUNCH(float, 8) IVF_LAUNCH(float, 16) default: return fail("bad k"); }
        } else if (R.dtype == SSMI_F16) {
            switch (r) { IVF_LAUNCH(__half, 1) IVF_LAUNCH(__half, 2) IVF_LAUNCH(__half, 4) IVF_LAUNCH(__half, 8) IVF_LAUNCH(__half, 16) default: return fail("bad k"); }
        } else {
            switch (r) { IVF_LAUNCH(bf16_t, 1) IVF_LAUNCH(bf16_t, 2) IVF_LAUNCH(bf16_t, 4) IVF_LAUNCH(bf16_t, 8) IVF_LAUNCH(bf16_t, 16) default: return fail("bad k"); }
        }
#undef IVF_LAUNCH
#undef IVF_QLAUNCH
#undef IVF_QLAUNCH_R
        HIPCHECK(hipGetLastError());
        if (h->timing_on && done == 0) HIPCHECK(hipEventRecord(ev[1], s));
        if (!tail)
            HIPCHECK(launch_merge_keys((const skey_t*)h->ws[2], nb, gx, k, Ddev + (size_t)done * k, Idev + (size_t)done * k,
                                       id_offset, s, min_ord));
#ifdef SSMI_IVF_STAMP
        if (qreg && getenv("SSMI_IVF_STAMP_PRINT")) {
            HIPCHECK(hipStreamSynchronize(s));
            const size_t nwv = (size_t)gx * (qthreads / 64);
            std::vector<unsigned long long> d(nwv * 4 + 1);
            HIPCHECK(hipMemcpy(d.data(), h->ws[14], d.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull;
            for (size_t w = 0; w < nwv; ++w) t0 = std::min(t0, d[w * 4]);
            double sum[4] = {0, 0, 0, 0}, mx[4] = {0, 0, 0, 0}, mn[4] = {1e30, 1e30, 1e30, 1e30};
            for (size_t w = 0; w < nwv; ++w)
                for (int i = 0; i < 4; ++i) {
                    const double us = (double)(d[w * 4 + i] - t0) * 0.01;
                    sum[i] += us; mx[i] = std::max(mx[i], us); mn[i] = std::min(mn[i], us);
                }
            {   // is the spread of the finish times inside the blocks or between them?
                const int nwb = qthreads / 64;
                double within = 0, bmin = 1e30, bmax = 0;
                double xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // mean finish time of the blocks of every XCD (block b runs on XCD b % 8)
                for (int b = 0; b < gx; ++b)
                    for (int w = 0; w < nwb; ++w) xcd[b % 8] += (double)(d[((size_t)b * nwb + w) * 4 + 2] - t0) * 0.01 / (nwb * (gx / 8.0));
                fprintf(stderr, "[ivf stamp] rows-done mean per XCD: %.1f %.1f %.1f %.1f %.1f %.1f %.1f %.1f us\n", xcd[0], xcd[1], xcd[2],
                        xcd[3], xcd[4], xcd[5], xcd[6], xcd[7]);
                if (getenv("SSMI_IVF_STAMP_BLOCKS")) {  // mean finish time of every block, 32 a line
                    for (int b = 0; b < gx; ++b) {
                        double m = 0;
                        for (int w = 0; w < nwb; ++w) m += (double)(d[((size_t)b * nwb + w) * 4 + 2] - t0) * 0.01 / nwb;
                        fprintf(stderr, "%.0f%c", m, (b % 32 == 31) ? '\n' : ' ');
                    }
                }
                for (int b = 0; b < gx; ++b) {
                    double lo = 1e30, hi = 0, sum2 = 0;
                    for (int w = 0; w < nwb; ++w) {
                        const double us = (double)(d[((size_t)b * nwb + w) * 4 + 2] - t0) * 0.01;
                        lo = std::min(lo, us); hi = std::max(hi, us); sum2 += us;
                    }
                    within += hi - lo;
                    bmin = std::min(bmin, sum2 / nwb); bmax = std::max(bmax, sum2 / nwb);
                }
                fprintf(stderr, "[ivf stamp] rows-done spread: inside a block %.1f us on average; block means from %.1f to %.1f us\n",
                        within / gx, bmin, bmax);
            }
            fprintf(stderr, "[ivf stamp] us since first wave start (min/mean/max over %zu waves): start %.1f/%.1f/%.1f  first row load %.1f/%.1f/%.1f  "
                            "rows done %.1f/%.1f/%.1f  list published %.1f/%.1f/%.1f  merged %.1f\n", nwv,
                    mn[0], sum[0] / nwv, mx[0], mn[1], sum[1] / nwv, mx[1], mn[2], sum[2] / nwv, mx[2], mn[3], sum[3] / nwv, mx[3],
                    (double)(d[nwv * 4] - t0) * 0.01);
        }
#endif
        if (h->timing_on && done == 0 && !tail) HIPCHECK(hipEventRecord(ev[2], s));
        if (h->timing_on && done == 0) h->merge_event = !tail;
    }
    if (h->timing_on) h->n_searches++;
    h->events_valid = h->timing_on;
    h->timing_valid = true;
    h->timing.last_path = 2;
    h->last_kernel = list_major ? (list_mfma ? "ivf_scan_lists_mfma_kernel" : "ivf_scan_lists_kernel")
                                : (qreg ? "ivf_scan_qreg_kernel" : "ivf_scan_topk_kernel");
    h->timing.last_grid = list_major ? -(list_mfma ? ivf_list_mfma_bt(std::min<int64_t>(nq, qbatch), nprobe, h->nlist) : lbt) : gx;  // negative: list-major scan with that many queries per work item
    h->timing.last_scan_bytes = -1;  // depends on the probed lists; see ssmi_index_get_list_sizes
    h->timing.last_scan_flops = -1;
    return 0;
}

int ivf_ids_to_slots(ssmi_index* h, int64_t n, const int64_t* ids, std::vector<int64_t>* slots) {
    IvfState* S = st(h);
    if (!S->has_direct_map) return fail("direct map not initialized (call make_direct_map first)");
    slots->resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) (*slots)[i] = S->direct_map[(size_t)ids[i]];
    return 0;
}

}  // namespace ssmi

// ---------------------------------------------------------------------------
// C ABI (IVF part)
// ---------------------------------------------------------------------------
extern "C" {

int ssmi_index_create_ivfflat(int d, int nlist, int storage_dtype, int device, ssmi_index** out) {
    return guarded([&]() -> int {
    if (!out) return fail("out is NULL");
    *out = nullptr;
    if (nlist < 1) return fail("nlist must be >= 1");
    ssmi_index* h = nullptr;
    CHECK(ssmi_index_create_flat(d, storage_dtype, device, &h));
    h->kind = SSMI_KIND_IVFFLAT;
    DeviceGuard g(device);
    if (ivf_create(h, nlist)) { ssmi_index_free(h); return 1; }
    *out = h;
    return 0;
    });
}

int ssmi_index_set_centroids(ssmi_index* h, const float* centroids, int mem_kind) {
    return guarded([&]() -> int {
    if (!h || !centroids) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("set_centroids: not an IVF index");
    if (h->ivf_ntotal != 0) return fail("set_centroids: index already holds vectors");
    DeviceGuard g(h->device);
    quiesce(h);
    return install_centroids(h, centroids, mem_kind);
    });
}

int ssmi_index_get_centroids(const ssmi_index* hc, float* out_host) {
    return guarded([&]() -> int {
    ssmi_index* h = const_cast<ssmi_index*>(hc);
    if (!h || !out_host) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("get_centroids: not an IVF index");
    if (!h->trained) return fail("get_centroids: index is not trained");
    DeviceGuard g(h->device);
    const RowStore& C = st(h)->cent32;
    const long long seg_rows = 1ll << C.seg_shift;
    for (long long r = 0; r < h->nlist;) {
        const long long seg = r >> C.seg_shift, off = r & (seg_rows - 1);
        const long long cnt = std::min<long long>(h->nlist - r, seg_rows - off);
        HIPCHECK(hipMemcpy2D(out_host + (size_t)r * C.d, (size_t)C.d * 4, (const char*)C.segs[seg] + (size_t)off * C.row_bytes,
                             C.row_bytes, (size_t)C.d * 4, (size_t)cnt, hipMemcpyDeviceToHost));
        r += cnt;
    }
    return 0;
    });
}

int ssmi_index_make_direct_map(ssmi_index* h) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return 0;  // flat rows are their own direct map
    DeviceGuard g(h->device);
    IvfState* S = st(h);
    std::vector<uint32_t> ids((size_t)h->ivf_slots_used);
    if (!ids.empty())
        HIPCHECK(hipMemcpy(ids.data(), S->d_slot_ids, ids.size() * 4, hipMemcpyDeviceToHost));
    S->direct_map.assign((size_t)h->ivf_ntotal, -1);
    for (int l = 0; l < h->nlist; ++l)
        for (long long pos = 0; pos < S->size[l]; ++pos) {
            const long long slot = (long long)S->pages[l][pos >> kPageShift] * kPageRows + (pos & (kPageRows - 1));
            const uint32_t id = ids[(size_t)slot];
            if (id >= (uint32_t)h->ivf_ntotal) return fail("internal: slot %lld holds id %u", slot, id);
            S->direct_map[id] = slot;
        }
    S->has_direct_map = true;
    return 0;
    });
}

int ssmi_index_add_ivf_shard(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, int64_t id0,
                             int shard, int nshards) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("add_ivf_shard: not an IVF index");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!x) return fail("NULL vectors");
    if (!dtype_known(src_dtype)) return fail("unknown source dtype %d", src_dtype);
    if (mem_kind != SSMI_HOST && mem_kind != SSMI_DEVICE) return fail("unknown mem_kind %d", mem_kind);
    DeviceGuard g(h->device);
    quiesce(h);
    return ivf_add_shard(h, n, x, src_dtype, mem_kind, id0, shard, nshards);
    });
}

int ssmi_index_assign(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind, int32_t* labels_out_host) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("assign: not an IVF index");
    if (!h->trained) return fail("assign on an untrained IVF index (call train first)");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!x || !labels_out_host) return fail("NULL buffer");
    if (!dtype_known(src_dtype)) return fail("unknown source dtype %d", src_dtype);
    if (mem_kind != SSMI_HOST && mem_kind != SSMI_DEVICE) return fail("unknown mem_kind %d", mem_kind);
    DeviceGuard g(h->device);
    quiesce(h);
    const RowStore& R = h->rows;
    const size_t el = (size_t)dtype_elsize(src_dtype);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, ((size_t)1 << 30) / ((size_t)R.ld * 4)));
    std::vector<int> labels;
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const float* xdev = nullptr;
        CHECK(rows_to_device_f32(h, cnt, (const char*)x + (size_t)done * R.d * el, src_dtype, mem_kind, &xdev));
        CHECK(assign_to_centroids_host(h, cnt, xdev, &labels));
        std::copy(labels.begin(), labels.end(), labels_out_host + done);
    }
    return 0;
    });
}

int ssmi_index_add_preassigned(ssmi_index* h, int64_t n, const void* x, int src_dtype, int mem_kind,
                               const int32_t* labels_host, const int64_t* ids_host) {
    return guarded([&]() -> int {
    if (!h) return fail("NULL index");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("add_preassigned: not an IVF index");
    if (!h->trained) return fail("add on an untrained IVF index (call train first)");
    if (n < 0) return fail("negative n");
    if (n == 0) return 0;
    if (!x || !labels_host || !ids_host) return fail("NULL buffer");
    if (!dtype_known(src_dtype)) return fail("unknown source dtype %d", src_dtype);
    if (mem_kind != SSMI_HOST && mem_kind != SSMI_DEVICE) return fail("unknown mem_kind %d", mem_kind);
    if (st(h)->has_direct_map) return fail("an index filled with explicit ids cannot maintain a direct map");
    std::vector<uint32_t> ids32((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        if (labels_host[i] < 0 || labels_host[i] >= h->nlist)
            return fail("add_preassigned: list %d of vector %lld out of range [0, %d)", labels_host[i], (long long)i, h->nlist);
        if (ids_host[i] < 0 || (unsigned long long)ids_host[i] >= 0xFFFFFFFFull)
            return fail("add_preassigned: id %lld out of range [0, 2^32-1)", (long long)ids_host[i]);
        ids32[(size_t)i] = (uint32_t)ids_host[i];
    }
    DeviceGuard g(h->device);
    quiesce(h);
    const RowStore& R = h->rows;
    const size_t el = (size_t)dtype_elsize(src_dtype);
    const int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(1 << 16, ((size_t)1 << 30) / ((size_t)R.ld * 4)));
    for (int64_t done = 0; done < n; done += chunk) {
        const int64_t cnt = std::min(chunk, n - done);
        const float* xdev = nullptr;
        CHECK(rows_to_device_f32(h, cnt, (const char*)x + (size_t)done * R.d * el, src_dtype, mem_kind, &xdev));
        CHECK(ivf_append_preassigned(h, cnt, xdev, labels_host + done, ids32.data() + done));
    }
    return 0;
    });
}

int ssmi_index_get_list_sizes(const ssmi_index* h, int64_t* out_host) {
    return guarded([&]() -> int {
    if (!h || !out_host) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("get_list_sizes: not an IVF index");
    for (int l = 0; l < h->nlist; ++l) out_host[l] = st(h)->size[l];
    return 0;
    });
}

int ssmi_index_get_assignments(const ssmi_index* hc, int32_t* out_host) {
    return guarded([&]() -> int {
    ssmi_index* h = const_cast<ssmi_index*>(hc);
    if (!h || !out_host) return fail("NULL argument");
    if (h->kind != SSMI_KIND_IVFFLAT) return fail("get_assignments: not an IVF index");
    DeviceGuard g(h->device);
    IvfState* S = st(h);
    std::vector<uint32_t> ids((size_t)h->ivf_slots_used);
    if (!ids.empty()) HIPCHECK(hipMemcpy(ids.data(), S->d_slot_ids, ids.size() * 4, hipMemcpyDeviceToHost));
    for (int l = 0; l < h->nlist; ++l)
        for (long long pos = 0; pos < S->size[l]; ++pos) {
            const long long slot = (long long)S->pages[l][pos >> kPageShift] * kPageRows + (pos & (kPageRows - 1));
            const uint32_t id = ids[(size_t)slot];
            if (id >= (uint32_t)h->ivf_ntotal) return fail("internal: slot %lld holds id %u", slot, id);
            out_host[id] = l;
        }
    return 0;
    });
}

}  // extern "C"
