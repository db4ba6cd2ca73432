// ivf_lists.hip - K8b: list-major ("bucketed") IVF scan for query batches.
//
// The per-query scan of ivf.hip reads every probed list once PER QUERY; with B
// queries and nprobe probes each, a list is probed by B*nprobe/nlist queries on
// average, so for batches the same rows are streamed many times.  Here the
// probe table is inverted on the device (histogram -> exclusive prefix sum ->
// scatter: the list-offset prefix-sum step of the north star) and the work item
// becomes (list, group of up to BT queries that probe it): the list's rows are
// streamed ONCE per group, every row is scored against the group's queries held
// in LDS (the K1 inner loop), each wave keeps BT register top-k lists, the block
// merges them and writes one k-list per (query, probe) pair; K3 merges the
// nprobe lists of a query.
//
// Stands behind faiss.IndexIVFFlat.search for nq > 1 (the reference itself only
// issues batch-1 searches, core/index_manager.py:371; batched search is the
// additive surface of SURVEY.md 8b/8f).
#include "ivf_lists.h"
#include "scan_impl.h"

namespace ssmi {

static const int kPageRows = 32;
static const int kPageShift = 5;

// probes per list
__global__ void ivf_probe_count_kernel(const int64_t* __restrict__ Ic, long long n_entries, int* __restrict__ cnt) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const int64_t l = Ic[e];
    if (l >= 0) atomicAdd(&cnt[l], 1);
}

// exclusive prefix sums over the lists: entry offsets (probes) and unit offsets (groups of BT probes).
// One 1024-thread block; thread t owns a contiguous run of lists.
__global__ void __launch_bounds__(1024) ivf_probe_offsets_kernel(const int* __restrict__ cnt, int nlist, int bt,
                                                                  int* __restrict__ eoff, int* __restrict__ uoff) {
    __shared__ int se[1024], su[1024];
    const int t = threadIdx.x;
    const int per = (nlist + 1023) / 1024;
    const int b = t * per, e = min(nlist, b + per);
    int sum_e = 0, sum_u = 0;
    for (int l = b; l < e; ++l) { sum_e += cnt[l]; sum_u += (cnt[l] + bt - 1) / bt; }
    se[t] = sum_e; su[t] = sum_u;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
        const int ve = (t >= d) ? se[t - d] : 0, vu = (t >= d) ? su[t - d] : 0;
        __syncthreads();
        se[t] += ve; su[t] += vu;
        __syncthreads();
    }
    int oe = se[t] - sum_e, ou = su[t] - sum_u;
    for (int l = b; l < e; ++l) {
        eoff[l] = oe; uoff[l] = ou;
        oe += cnt[l]; ou += (cnt[l] + bt - 1) / bt;
    }
    if (t == 1023) { eoff[nlist] = se[1023]; uoff[nlist] = su[1023]; }
}

// scatter the (query, probe) pairs into their list's run; a pair with no list gets an empty result
__global__ void ivf_probe_fill_kernel(const int64_t* __restrict__ Ic, long long n_entries, const int* __restrict__ eoff,
                                      int* __restrict__ cursor, int* __restrict__ entries, skey_t* __restrict__ partial,
                                      int k) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_entries) return;
    const int64_t l = Ic[e];
    if (l >= 0) {
        const int pos = atomicAdd(&cursor[l], 1);
        entries[eoff[l] + pos] = (int)e;
    } else {
        for (int i = 0; i < k; ++i) partial[(size_t)e * k + i] = 0;
    }
}

template <typename T, int BT, int R>
__global__ void __launch_bounds__(1024) ivf_scan_lists_kernel(IvfListScanArgs A) {
    constexpr int V = ElemTraits<T>::V;
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int s_list;
    float* qs = reinterpret_cast<float*>(smem);
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int unit = blockIdx.x;
    if (unit >= A.uoff[A.nlist]) return;  // the grid is an upper bound on the number of units

    if (threadIdx.x == 0) {  // last list whose first unit is <= unit (lists without probes have no units)
        int lo = 0, hi = A.nlist - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (A.uoff[mid] <= unit) lo = mid; else hi = mid - 1;
        }
        s_list = lo;
    }
    __syncthreads();
    const int l = s_list;
    const int g = unit - A.uoff[l];
    const int e0 = A.eoff[l] + g * BT;
    const int ne = min(BT, A.eoff[l + 1] - e0);

    const int ld = A.X.ld;
    const int nch = (ld + 64 * V - 1) / (64 * V);
    const int ldq = nch * 64 * V;
    const int row_bytes = ld * (int)sizeof(T);
    for (int b = 0; b < BT; ++b) {
        const int q = A.entries[e0 + (b < ne ? b : 0)] / A.nprobe;  // short groups repeat their first query
        for (int e = threadIdx.x; e < ldq; e += blockDim.x)
            qs[b * ldq + q_lds_index<V>(e)] = (e < ld) ? A.Q[(size_t)q * ld + e] : 0.f;
    }
    __syncthreads();

    WaveTopK<R> top[BT];
    skey_t thr[BT];
    float thr_s[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) { top[b].init(); thr[b] = 0; thr_s[b] = SSMI_NEG_MAX; }

    const int nrows = A.list_size[l];
    const int* lp = A.pages + A.page_off[l];
    const long long seg_mask = (1ll << A.X.seg_shift) - 1;
    for (int r = wave; r < nrows; r += nwaves) {
        const long long slot = (long long)lp[r >> kPageShift] * kPageRows + (r & (kPageRows - 1));
        const unsigned char* rowp = reinterpret_cast<const unsigned char*>(A.X.segs[slot >> A.X.seg_shift]) +
                                    (size_t)(slot & seg_mask) * (size_t)row_bytes;
        float acc[BT][4];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.f;
        constexpr int G = BT >= 8 ? (V == 8 ? 1 : 4) : 8;  // 8 queries x 8 elements per 16 B: one chunk keeps fp16 rows inside the register budget
#pragma unroll 1
        for (int c0 = 0; c0 < nch; c0 += G) {
            uint4 x[G];
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                const int off = ((c0 + gg) * 64 + lane) * 16;
                x[gg] = make_uint4(0, 0, 0, 0);
                if (off < row_bytes) x[gg] = load16_nt(rowp + off);
            }
#pragma unroll
            for (int gg = 0; gg < G; ++gg) {
                if (c0 + gg < nch) {
                    float xf[V];
                    Unpack<T>::run(x[gg], xf);
#pragma unroll
                    for (int b = 0; b < BT; ++b) {
#pragma unroll
                        for (int p = 0; p < V / 4; ++p) {
                            const float4 t4 = *reinterpret_cast<const float4*>(qs + b * ldq + (c0 + gg) * 64 * V + p * 256 + lane * 4);
                            acc[b][0] = fmaf(xf[4 * p], t4.x, acc[b][0]);
                            acc[b][1] = fmaf(xf[4 * p + 1], t4.y, acc[b][1]);
                            acc[b][2] = fmaf(xf[4 * p + 2], t4.z, acc[b][2]);
                            acc[b][3] = fmaf(xf[4 * p + 3], t4.w, acc[b][3]);
                        }
                    }
                }
            }
        }
        uint32_t id = 0;
        bool have_id = false;
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            float s = wave_sum((acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]));
            s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));
            if (s >= thr_s[b]) {  // cheap float gate; the exact (score, id) order needs the id
                if (!have_id) { id = A.slot_ids[slot]; have_id = true; }
                const skey_t key = make_key(s, id);
                if (key > thr[b]) {
                    top[b].insert(key);
                    thr[b] = top[b].kth(A.k);
                    thr_s[b] = (thr[b] == 0) ? SSMI_NEG_MAX : key_score(thr[b]);
                }
            }
        }
    }

    __syncthreads();
    skey_t* lds_keys = reinterpret_cast<skey_t*>(smem);
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        block_merge<R>(top[b], lds_keys, wave, nwaves);
        if (wave == 0 && b < ne) top[b].store(A.partial + (size_t)A.entries[e0 + b] * A.k, A.k);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
bool plan_ivf_list_scan(int dtype, int ld, int k, int* bt, int* threads, size_t* lds) {
    const int V = dtype == 0 ? 4 : 8;
    const int ldq = (ld + 64 * V - 1) / (64 * V) * 64 * V;
    int r = 1;
    while (64 * r < k) r <<= 1;
    int b = r == 1 ? 8 : (r == 2 ? 4 : (r == 4 ? 2 : 1));
    while (b > 1 && (size_t)b * ldq * 4 > 144 * 1024) b >>= 1;
    if ((size_t)b * ldq * 4 > 144 * 1024) return false;
    const size_t qbytes = (size_t)b * ldq * 4;
    *threads = qbytes > 80 * 1024 ? 1024 : (qbytes > 40 * 1024 ? 512 : 256);
    const size_t mbytes = (size_t)(*threads / 64) * 64 * r * sizeof(skey_t);
    *lds = qbytes > mbytes ? qbytes : mbytes;
    *bt = b;
    return true;
}

template <typename T, int BT, int R>
static hipError_t launch_one(const IvfListScanArgs& A, hipStream_t s) {
    auto kern = ivf_scan_lists_kernel<T, BT, R>;
    if (A.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)A.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(A.max_units), dim3(A.threads), A.lds, s, A);
    return hipGetLastError();
}

template <typename T>
static hipError_t launch_t(const IvfListScanArgs& A, hipStream_t s) {
    int r = 1;
    while (64 * r < A.k) r <<= 1;
    switch (A.bt * 100 + r) {
        case 801: return launch_one<T, 8, 1>(A, s);
        case 401: return launch_one<T, 4, 1>(A, s);
        case 402: return launch_one<T, 4, 2>(A, s);
        case 201: return launch_one<T, 2, 1>(A, s);
        case 202: return launch_one<T, 2, 2>(A, s);
        case 204: return launch_one<T, 2, 4>(A, s);
        case 101: return launch_one<T, 1, 1>(A, s);
        case 102: return launch_one<T, 1, 2>(A, s);
        case 104: return launch_one<T, 1, 4>(A, s);
        case 108: return launch_one<T, 1, 8>(A, s);
        case 116: return launch_one<T, 1, 16>(A, s);
    }
    return hipErrorInvalidValue;
}

// inverts the probe table: cnt/eoff/uoff/entries (A.cnt, A.cursor and the work counter behind them are zeroed first)
hipError_t launch_ivf_probe_table(const IvfListScanArgs& A, hipStream_t s) {
    const long long n_entries = (long long)A.nq * A.nprobe;
    hipError_t e = hipMemsetAsync(A.cnt, 0, sizeof(int) * (2 * (size_t)A.nlist + 1), s);  // cnt, cursor, work counter
    if (e != hipSuccess) return e;
    const unsigned eb = (unsigned)((n_entries + 255) / 256);
    hipLaunchKernelGGL(ivf_probe_count_kernel, dim3(eb), dim3(256), 0, s, A.Ic, n_entries, A.cnt);
    hipLaunchKernelGGL(ivf_probe_offsets_kernel, dim3(1), dim3(1024), 0, s, (const int*)A.cnt, A.nlist, A.bt, A.eoff, A.uoff);
    hipLaunchKernelGGL(ivf_probe_fill_kernel, dim3(eb), dim3(256), 0, s, A.Ic, n_entries, (const int*)A.eoff, A.cursor,
                       A.entries, A.partial, A.k);
    return hipGetLastError();
}

hipError_t launch_ivf_list_scan(const IvfListScanArgs& A, int dtype, hipStream_t s) {
    hipError_t e = launch_ivf_probe_table(A, s);
    if (e != hipSuccess) return e;
    return dtype == 0 ? launch_t<float>(A, s) : (dtype == 1 ? launch_t<__half>(A, s) : launch_t<bf16_t>(A, s));
}

}  // namespace ssmi
