// ivf_bucket.hip - K7b: bucketing of vectors by inverted list on the device.
//
//   labels (list of every vector)  ->  histogram  ->  exclusive prefix sum (list offsets)  ->  stable scatter
//
// Used by both writers of an IVF index: k-means training (the members of every centroid as one permutation + CSR
// offsets for the fixed-order mean, K6) and add() (the position of every new vector inside its list; the host
// then only allocates pages per LIST, never touches a vector).  Stands behind the list bookkeeping inside
// faiss.IndexIVFFlat.train / .add (reference call sites core/index_factory.py:191, core/index_manager.py:299,319).
//
// The sort is STABLE (vectors of one list keep their input order), so k-means sums in a fixed order and a list's
// storage order is the add order: results, files and assignments() are reproducible run to run.  Stability without
// a global sort: the input is cut into blocks of `chunk` consecutive vectors, one wavefront each;
//   count    per (block, list) histogram                                 cnt[b][l]
//   scan     per list: exclusive scan over the blocks (cnt[b][l] becomes the block's base inside the list) and the
//            list total; then ONE exclusive scan over the lists -> off[0..nlist]
//   scatter  every wave walks its block in order, 64 vectors a step; the lanes that share a list find each other with
//            ballots (one per distinct list of the step) and take consecutive positions behind the block's cursor of
//            that list, which the group's first lane advances with one atomic (LDS, or the global table when nlist is
//            too large for LDS).
#include "kernels.h"
#include "topk.h"

namespace ssmi {

static constexpr int kBucketLdsLists = 32768;  // cursors of up to this many lists live in LDS (128 KiB)

__global__ void __launch_bounds__(256) bucket_count_kernel(const int64_t* __restrict__ labels, long long n, int nlist,
                                                           int chunk, int* __restrict__ cnt, int* __restrict__ bad) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t l = labels[i];
    if (l == kBucketSkip) return;
    if (l < 0 || l >= nlist) { atomicExch(bad, 1); return; }
    atomicAdd(&cnt[(size_t)(i / chunk) * nlist + (size_t)l], 1);
}

// one thread per list: exclusive scan over the blocks, in place; total[l] = members of list l
__global__ void __launch_bounds__(256) bucket_scan_blocks_kernel(int* __restrict__ cnt, int nb, int nlist, int* __restrict__ total) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= nlist) return;
    int run = 0;
    for (int b = 0; b < nb; ++b) {
        const int c = cnt[(size_t)b * nlist + l];
        cnt[(size_t)b * nlist + l] = run;
        run += c;
    }
    total[l] = run;
}

// exclusive prefix sum of total[0..nlist) -> off[0..nlist] (one 1024-thread block, carried chunk by chunk)
__global__ void __launch_bounds__(1024) bucket_scan_lists_kernel(const int* __restrict__ total, int nlist, int* __restrict__ off) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nlist; base += 1024) {
        const int j = base + threadIdx.x;
        const int v = j < nlist ? total[j] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int woff = carry;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        if (j < nlist) off[j] = woff + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) off[nlist] = carry;
}

// one wave per block of `chunk` vectors.  perm (optional): perm[off[l] + position in list] = vector index;
// rank (optional): rank[i] = position of vector i among the batch's vectors of its list (-1 for skipped vectors).
template <bool LDS_CURSOR>
__global__ void __launch_bounds__(64) bucket_scatter_kernel(const int64_t* __restrict__ labels, long long n, int nlist, int chunk,
                                                            int* __restrict__ cnt /*[nb][nlist] block bases*/,
                                                            const int* __restrict__ off, int* __restrict__ perm,
                                                            int* __restrict__ rank) {
    extern __shared__ int cur_l[];
    const int lane = lane_id();
    const long long b = blockIdx.x;
    int* cur = LDS_CURSOR ? cur_l : cnt + (size_t)b * nlist;
    if (LDS_CURSOR) {
        for (int l = lane; l < nlist; l += 64) cur_l[l] = cnt[(size_t)b * nlist + l];
        __syncthreads();  // one wave: a wait on the LDS writes
    }
    const long long i0 = b * chunk;
    const long long i1 = i0 + chunk < n ? i0 + chunk : n;
    const unsigned long long below = (1ull << lane) - 1ull;
    for (long long base = i0; base < i1; base += 64) {
        const long long i = base + lane;
        const bool in = i < i1;
        int lab = -1;
        if (in) {
            const int64_t l64 = labels[i];
            lab = (l64 >= 0 && l64 < nlist) ? (int)l64 : -1;  // skipped (-2) or reported by the count kernel
        }
        // the lanes of one list find each other: one ballot per distinct list of this step, no memory traffic
        unsigned long long todo = __builtin_amdgcn_ballot_w64(lab >= 0);
        unsigned long long mine = 0;
        while (todo) {
            const int l = __builtin_amdgcn_readlane(lab, __builtin_ctzll(todo));
            const unsigned long long same = __builtin_amdgcn_ballot_w64(lab == l);
            if (lab == l) mine = same;
            todo &= ~same;
        }
        // the first lane of every list advances the block's cursor of that list by the group's size (distinct
        // addresses within a step; steps of one wave are ordered by the returned value), the others read its result
        const int first = mine ? __builtin_ctzll(mine) : lane;
        int start = 0;
        if (mine && lane == first) start = atomicAdd(&cur[lab], (int)__builtin_popcountll(mine));
        start = __shfl(start, first);
        const int pos = start + (int)__builtin_popcountll(mine & below);
        if (in) {
            if (rank) rank[i] = lab >= 0 ? pos : -1;
            if (perm && lab >= 0) perm[(size_t)off[lab] + pos] = (int)i;
        }
    }
}

// Small batches (n <= kBucketSmall: the reference adds 16 vectors at a time, core/config.py:59): one workgroup ranks
// the vectors against each other through LDS, no histogram over all nlist lists.  last[i] = index of the last vector
// of i's list in the batch; that vector reports (list, count) in touched[i], every other one (-1, 0); touched[n].x is
// the out-of-range flag.  The host then visits at most n lists instead of nlist.
__global__ void __launch_bounds__(kBucketSmall) bucket_small_kernel(const int64_t* __restrict__ labels, int n, int nlist,
                                                                    int* __restrict__ rank, int* __restrict__ last,
                                                                    int2* __restrict__ touched) {
    __shared__ int lab[kBucketSmall];
    __shared__ int bad;
    const int i = threadIdx.x;
    if (i == 0) bad = 0;
    __syncthreads();
    const int64_t l64 = i < n ? labels[i] : kBucketSkip;
    const bool valid = l64 >= 0 && l64 < nlist;
    if (!valid && l64 != kBucketSkip) bad = 1;
    lab[i] = valid ? (int)l64 : -1 - i;  // a value no other entry holds
    __syncthreads();
    if (i < n) {
        int r = -1, la = i;
        if (valid) {
            const int l = (int)l64;
            r = 0;
            for (int j = 0; j < n; ++j) {  // broadcast LDS reads: every thread reads the same word
                const bool same = lab[j] == l;
                r += (same && j < i) ? 1 : 0;
                la = same ? j : la;
            }
        }
        rank[i] = r;
        last[i] = la;
        touched[i] = (valid && la == i) ? make_int2((int)l64, r + 1) : make_int2(-1, 0);
    }
    if (i == 0) touched[n] = make_int2(bad, 0);
}

hipError_t launch_ivf_bucket_small(const int64_t* labels, int n, int nlist, int* rank, int* last, int2* touched, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (n > kBucketSmall) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bucket_small_kernel, dim3(1), dim3(kBucketSmall), 0, s, labels, n, nlist, rank, last, touched);
    return hipGetLastError();
}

size_t ivf_bucket_ws_bytes(long long n, int nlist, int* chunk_out) {
    // at most 256 blocks and 2^26 (block, list) counters (256 MiB), at least 1024 vectors per block
    long long nb = (1ll << 26) / nlist;
    nb = nb < 1 ? 1 : (nb > 256 ? 256 : nb);
    long long chunk = (n + nb - 1) / nb;
    if (chunk < 1024) chunk = 1024;
    chunk = (chunk + 63) / 64 * 64;
    if (chunk_out) *chunk_out = (int)chunk;
    nb = (n + chunk - 1) / chunk;
    return ((size_t)nb * nlist + (size_t)nlist + 64) * sizeof(int);  // cnt[nb][nlist], total[nlist], bad flag
}

// labels: n int64 on the device (kBucketSkip = leave the vector out); ws: ivf_bucket_ws_bytes; off: nlist+1 ints
// (device); perm / rank: n ints or nullptr.  total_out (device): the nlist list totals FOLLOWED by the flag that an
// out-of-range label was seen (nlist+1 ints, one read-back).
hipError_t launch_ivf_bucket(const int64_t* labels, long long n, int nlist, void* ws, int* off, int* perm, int* rank,
                             const int** total_out, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    int chunk = 0;
    const size_t bytes = ivf_bucket_ws_bytes(n, nlist, &chunk);
    const int nb = (int)((n + chunk - 1) / chunk);
    int* cnt = static_cast<int*>(ws);
    int* total = cnt + (size_t)nb * nlist;
    int* bad = total + nlist;
    hipError_t e = hipMemsetAsync(ws, 0, bytes, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(bucket_count_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, labels, n, nlist, chunk, cnt, bad);
    hipLaunchKernelGGL(bucket_scan_blocks_kernel, dim3((unsigned)((nlist + 255) / 256)), dim3(256), 0, s, cnt, nb, nlist, total);
    hipLaunchKernelGGL(bucket_scan_lists_kernel, dim3(1), dim3(1024), 0, s, total, nlist, off);
    if (nlist <= kBucketLdsLists) {
        const size_t lds = (size_t)nlist * sizeof(int);
        if (lds > 48 * 1024) {
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_scatter_kernel<true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((bucket_scatter_kernel<true>), dim3(nb), dim3(64), lds, s, labels, n, nlist, chunk, cnt, off, perm, rank);
    } else {
        hipLaunchKernelGGL((bucket_scatter_kernel<false>), dim3(nb), dim3(64), 0, s, labels, n, nlist, chunk, cnt, off, perm, rank);
    }
    if (total_out) *total_out = total;
    return hipGetLastError();
}

}  // namespace ssmi
