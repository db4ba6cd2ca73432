// ivf_lists_mfma.hip - K8m: list-major IVF scan of a query batch on the matrix cores
// (fp16 list storage, rows of a multiple of 512 elements, k <= 256).
//
// Same work decomposition as ivf_lists.hip (probe table inverted on the device; work item =
// (list, group of queries probing it)), but a group holds up to 16 * NT queries (16, 64 or 128)
// and is scored with v_mfma_f32_16x16x32_f16 in the register-streaming form of mfma_r128.hip:
// a list page (32 slots) is exactly one wave's 32-row block, so wave w of the 8-wave workgroup
// streams pages w, w+8, ... of the list straight into registers (4 global_load_dwordx4 per
// 64-deep K step, ring of PD steps, hand-counted vmcnt), while the group's query rows are gathered by
// LDS-DMA into a ring of query slabs.  With B*nprobe/nlist probes per list on average, a list is
// streamed ONCE for up to 128 of them instead of once per 8.
//
// Workgroups are persistent (one per CU) and pull work items from a device counter, because items
// differ in length (lists of 1.3k .. 4.7k rows at BASELINE configs[3]).  The threshold filter and the
// list compaction are those of the Flat kernels (mfma_common.h); survivors carry the stored vector
// id (slot_ids) so that ties order by id exactly as on the per-query path.
//
// Queries are rounded to fp16 for the contraction, like the Flat MFMA kernels.
#include "ivf_lists.h"
#include "mfma_common.h"

namespace ssmi {

static constexpr int LQ_SLOT = 16 * 1024;  // one query slab: 2 k-halves x (128 queries x 64 B)

// Wave-uniform table look-ups as scalar loads: an ordinary (vector) load would make the compiler wait with
// vmcnt(0) for its result, which drains the register ring of row loads once per tile.
__device__ __forceinline__ int sload32(const int* p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}
__device__ __forceinline__ const unsigned char* sload_ptr(const void* const* p) {
    const unsigned char* v;
    asm volatile("s_load_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <int R, int PD, int NT, bool BF = false>
__global__ void __launch_bounds__(512) ivf_scan_lists_mfma_kernel(IvfListScanArgs A, const _Float16* __restrict__ Qh,
                                                                  skey_t* __restrict__ cand_all, int* __restrict__ work_counter) {
    constexpr int QD = PD + 1;
    constexpr int BT = 16 * NT;  // queries per work item
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + QD * LQ_SLOT);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + N128_TN);
    unsigned* kept_l = cnt_l + N128_TN;
    unsigned* flag_l = kept_l + N128_TN;
    int* s_unit = reinterpret_cast<int*>(flag_l + 1);  // [0] unit, [1] list

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int fch = (fq ^ ((-(frow >> 2)) & 3)) * 16;
    const int b_off = frow * 64 + fch;
    const int ld = A.X.ld;
    const int ns = ld / 64;
    const size_t row_bytes = (size_t)ld * 2;
    const long long seg_mask = (1ll << A.X.seg_shift) - 1;
    skey_t* cand = cand_all + (size_t)blockIdx.x * N128_TN * CAND_CAP;
    const int nunits = A.uoff[A.nlist];

    for (;;) {
        // ---- next work item ----
        __syncthreads();
        if (tid == 0) {
            const int u = atomicAdd(work_counter, 1);
            int l = 0;
            if (u < nunits) {  // last list whose first unit is <= u
                int lo = 0, hi = A.nlist - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (A.uoff[mid] <= u) lo = mid; else hi = mid - 1;
                }
                l = lo;
            }
            s_unit[0] = u;
            s_unit[1] = l;
        }
        __syncthreads();
        const int unit = s_unit[0];
        if (unit >= nunits) break;
        const int l = s_unit[1];
        const int gidx = unit - A.uoff[l];
        const int e0 = A.eoff[l] + gidx * BT;
        const int ne = min(BT, A.eoff[l + 1] - e0);
        const int nrows = A.list_size[l];
        const int npages = (nrows + 31) >> 5;
        const int* lp = A.pages + A.page_off[l];
        const int ntiles = (npages + 7) >> 3;  // a "tile" = 8 pages, one per wave
        const int total = ntiles * ns;

        // A (query, list) pair starts from the best k-th score any other list of the query has reached so far
        // (qthr): k rows of the query score at least that, so nothing below it can be in the final top-k.  Without
        // it every pair starts empty: all 256 rows of its first tile survive, are stored and sorted.
        if (tid < N128_TN) {
            float t0 = (tid < ne) ? SSMI_NEG_MAX : __builtin_inff();
            if (tid < ne && A.qthr) {
                const unsigned seen = __hip_atomic_load(&A.qthr[A.entries[e0 + tid] / A.nprobe], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen) t0 = ord_to_f32(seen);
            }
            thr_l[tid] = t0;
            cnt_l[tid] = 0;
            kept_l[tid] = 0;
        }
        // after a fold: hand this pair's k-th score to the query's other lists and take theirs
        auto share_thr = [&]() {
            if (A.qthr && lane < BT / 8) {
                const int ql = wave + 8 * lane;
                if (ql < ne) {
                    const float t = thr_l[ql];
                    const unsigned mine = (t > SSMI_NEG_MAX) ? f32_to_ord(t) : 0u;
                    const unsigned seen = atomicMax(&A.qthr[A.entries[e0 + ql] / A.nprobe], mine);
                    if (seen > mine) thr_l[ql] = ord_to_f32(seen);
                }
            }
        };
        if (tid == 0) *flag_l = 0;
        __syncthreads();

        if (total > 0) {
            // query rows of the group, gathered by all 8 waves in equal shares (QSlabDma; short groups repeat
            // their last query): every wave issues 4 row loads + QOPS DMAs per step, so one counted wait serves
            // all of them.  Steps past the end of the list repeat the last one (into ring positions nobody
            // reads any more).
            using QD_t = QSlabDma<NT>;
            constexpr int VOPS = 4 + QD_t::OPS;
            const int qrow = QD_t::qrow(wave, lane);
            const int qe = A.entries[e0 + (qrow < ne ? qrow : ne - 1)] / A.nprobe;
            const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(Qh) + (size_t)qe * row_bytes + QD_t::src_off(wave, lane);
            auto issue_q = [&](int g) {
                const int s = (g < total ? g : total - 1) % ns;
                QD_t::issue(qsrc + (size_t)s * 128, smem + (g % QD) * LQ_SLOT + QD_t::dst_off(wave));
            };
            // this wave's page of tile t: row base of the lane's two 16-row blocks, first slot, valid rows
            auto tile_page = [&](int t, const unsigned char* (&base)[2], long long& slot0, int& rv) {
                const int pidx = t * 8 + wave;
                const int pg = sload32(lp + (pidx < npages ? pidx : npages - 1));  // waves without a page re-read the last one
                slot0 = (long long)pg * 32;
                rv = pidx < npages ? min(32, nrows - pidx * 32) : 0;
                const unsigned char* segbase = sload_ptr(A.X.segs + (slot0 >> A.X.seg_shift));
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    base[m] = segbase + (size_t)((slot0 & seg_mask) + m * 16 + frow) * row_bytes + fq * 16;
            };
            half8_t areg[PD][2][2];
            // The ring registers are written by the memory system long after the load instruction: between a
            // load and its counted wait the compiler must neither copy nor reuse them.  Each ring slot is
            // therefore defined by exactly ONE load statement and ONE wait statement per step (no alternative
            // definitions that would meet in a phi and be copied), and the ring is drained with the registers
            // still named before anything else may take them.
            auto load_a = [&](half8_t (&dst)[2][2], const unsigned char* const (&base)[2], int s) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int kh = 0; kh < 2; ++kh) {
                        const unsigned char* ptr = base[m] + (size_t)s * 128 + kh * 64;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[m][kh]) : "v"(ptr) : "memory");
                    }
            };

            const unsigned char* abase[2];
            long long slot0 = 0;
            int rv = 0;
            tile_page(0, abase, slot0, rv);
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                issue_q(u);
                load_a(areg[u], abase, u);
            }
            f32x4_t acc[2][NT];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

            unsigned tiles_done = 0;
            int g = 0;
            for (int t = 0; t < ntiles; ++t) {
                const unsigned char* nbase[2] = {abase[0], abase[1]};
                long long nslot0 = slot0;
                int nrv = rv;
                if (t + 1 < ntiles) tile_page(t + 1, nbase, nslot0, nrv);
                for (int s0 = 0; s0 < ns; s0 += PD) {
                    const bool refill_next_tile = (s0 + PD >= ns);
#pragma unroll
                    for (int u = 0; u < PD; ++u, ++g) {
#define RING_WAIT(N)                                                                                             \
    asm volatile("s_waitcnt vmcnt(%4)"                                                                           \
                 : "+v"(areg[u][0][0]), "+v"(areg[u][0][1]), "+v"(areg[u][1][0]), "+v"(areg[u][1][1])           \
                 : "n"(N)                                                                                        \
                 : "memory")
                        RING_WAIT((PD - 1) * VOPS);  // step g landed; the PD-1 younger steps may fly
                        __builtin_amdgcn_s_barrier();
                        issue_q(g + PD);
                        const unsigned char* slot = smem + (g % QD) * LQ_SLOT + b_off;
#pragma unroll
                        for (int kh = 0; kh < 2; ++kh) {
                            half8_t b[NT];
#pragma unroll
                            for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const half8_t*>(slot + kh * 8192 + n * 1024);
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                            for (int m = 0; m < 2; ++m)
#pragma unroll
                                for (int n = 0; n < NT; ++n)
                                    acc[m][n] = mfma16<BF>(areg[u][m][kh], b[n], acc[m][n]);
                        }
                        {   // refill ring slot u with step g + PD (next tile's page when this one is used up; past the
                            // end of the list: this step again, never consumed)
                            const bool past = g + PD >= total;
                            const unsigned char* rb[2];
                            rb[0] = (refill_next_tile && !past) ? nbase[0] : abase[0];
                            rb[1] = (refill_next_tile && !past) ? nbase[1] : abase[1];
                            const int rs = past ? s0 + u : (refill_next_tile ? s0 + PD - ns + u : s0 + PD + u);
                            load_a(areg[u], rb, rs);
                        }
                    }
                }
                // ---- the wave's page is scored: filter (row index = slot within the page), fold when due ----
                filter_tile<2, NT, false>(acc, 0, 0, frow, fq, 0, rv, cand, thr_l, cnt_l, flag_l, A.slot_ids + slot0);
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                abase[0] = nbase[0];
                abase[1] = nbase[1];
                slot0 = nslot0;
                rv = nrv;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                ++tiles_done;
                if (((tiles_done & (tiles_done - 1)) == 0 && (tiles_done & 0x55555555u)) || *flag_l) {
                    __syncthreads();
                    for (int ql = wave; ql < ne; ql += 8)
                        compact_list<R>(ql, A.partial + (size_t)A.entries[e0 + ql] * A.k, A.k, cand + (size_t)ql * CAND_CAP,
                                        thr_l, cnt_l, kept_l);
                    share_thr();
                    __syncthreads();
                    if (tid == 0) *flag_l = 0;
                }
            }
            // drain the ring (the repeated steps of the tail) while its registers are still reserved
#pragma unroll
            for (int u = 0; u < PD; ++u) RING_WAIT(0);
#undef RING_WAIT
        }
        // ---- end of the work item: every (query, probe) pair of the group gets its k-list ----
        __syncthreads();
        for (int ql = wave; ql < ne; ql += 8)
            compact_list<R>(ql, A.partial + (size_t)A.entries[e0 + ql] * A.k, A.k, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l,
                            kept_l);
        share_thr();
    }
}

bool ivf_list_mfma_supported(int dtype, int ld, int k) { return dtype_is16(dtype) && (ld % 512) == 0 && k <= 256; }

// queries per work item for this batch: the smallest of 16 / 64 / 128 that covers twice the average probes per list
int ivf_list_mfma_bt(long long nq, int nprobe, int nlist) {
    if (const char* e = getenv("SSMI_IVF_LIST_BT")) {  // experiments: force the group size
        const int v = atoi(e);
        if (v == 16 || v == 64 || v == 128) return v;
    }
    const double avg = (double)nq * nprobe / nlist;
    return avg <= 8 ? 16 : (avg <= 32 ? 64 : 128);
}

size_t ivf_list_mfma_cand_bytes(int num_cu) { return (size_t)num_cu * N128_TN * CAND_CAP * sizeof(skey_t); }

template <int PD, int NT>
static hipError_t launch_nt(const IvfListScanArgs& A, const _Float16* Qh, skey_t* cand, int* counter, int r, int grid,
                            bool bf16, hipStream_t s) {
    const size_t lds = (size_t)(PD + 1) * LQ_SLOT + 3 * N128_TN * 4 + 32;
    hipError_t e;
#define LAUNCH_K(...)                                                                                        \
    {                                                                                                        \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_scan_lists_mfma_kernel<__VA_ARGS__>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                       \
        if (e != hipSuccess) return e;                                                                       \
        hipLaunchKernelGGL((ivf_scan_lists_mfma_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, s, A, Qh, cand, counter); \
    }
#define LAUNCH_R(RR) case RR: if (bf16) LAUNCH_K(RR, PD, NT, true) else LAUNCH_K(RR, PD, NT, false) break;
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
#undef LAUNCH_K
    return hipGetLastError();
}

// A.bt must be 16, 64 or 128; A.cnt has 2*nlist+1 ints (the last one is the work counter); grid = one block per CU.
hipError_t launch_ivf_list_scan_mfma(const IvfListScanArgs& A, const void* qh, void* cand, int num_cu, bool bf16, hipStream_t s) {
    const long long n_entries = (long long)A.nq * A.nprobe;
    hipError_t e = launch_ivf_probe_table(A, s);
    if (e != hipSuccess) return e;
    (void)n_entries;
    int r = 1;
    while (64 * r < A.k) r <<= 1;
    int* counter = A.cnt + 2 * A.nlist;
    const _Float16* Qh = (const _Float16*)qh;
    if (A.bt == 16) return launch_nt<8, 1>(A, Qh, (skey_t*)cand, counter, r, num_cu, bf16, s);
    if (A.bt == 64) return launch_nt<8, 4>(A, Qh, (skey_t*)cand, counter, r, num_cu, bf16, s);
    return launch_nt<4, 8>(A, Qh, (skey_t*)cand, counter, r, num_cu, bf16, s);
}

}  // namespace ssmi
