// merge.hip - K3: merge sorted per-block (or per-shard) top-k lists into the
// final (D, I) of a search.  One workgroup per query; each wave folds a strided
// subset of the lists into its register list with bitonic merges (topk.h), the
// waves combine through LDS, wave 0 decodes keys to (score, id).
//
// Stands in for the heap bookkeeping inside faiss.IndexFlatIP.search
// (reference call site core/index_manager.py:371) and, in the sharded form, for
// the result merge the north-star asks for after the RCCL all-gather.
#include "kernels.h"
#include "topk.h"
#include "tail_merge.h"
#include <type_traits>

namespace ssmi {

template <int R>
__global__ void merge_keys_kernel(const skey_t* __restrict__ partial, int P, int k, float* __restrict__ D,
                                  int64_t* __restrict__ I, int64_t id_offset, unsigned min_ord) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int q = blockIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    merge_partials<R, false>(partial + (size_t)q * P * k, P, k, reinterpret_cast<skey_t*>(smem), wave, (int)(blockDim.x >> 6),
                      D + (size_t)q * k, I + (size_t)q * k, id_offset, PlanOut{nullptr, nullptr, nullptr, nullptr}, min_ord);
}

// k == 1 (centroid assignment of an add or a k-means pass: 65536 queries a launch): the merge is a maximum over
// the P keys of a query - one thread per query, no workgroup per query, no LDS
__global__ void __launch_bounds__(256) merge_top1_kernel(const skey_t* __restrict__ partial, int nq, int P, float* __restrict__ D,
                                                         int64_t* __restrict__ I, int64_t id_offset, unsigned min_ord) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const skey_t* p = partial + (size_t)q * P;
    skey_t best = 0;
    int e = 0;
    if ((P & 1) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0) {
        const ulonglong2* p2 = reinterpret_cast<const ulonglong2*>(p);
        for (; e < P; e += 2) {
            const ulonglong2 v = p2[e >> 1];
            best = kmax(best, kmax((skey_t)v.x, (skey_t)v.y));
        }
    }
    for (; e < P; ++e) best = kmax(best, p[e]);
    if (best == 0 || (unsigned)(best >> 32) < min_ord) { D[q] = SSMI_NEG_MAX; I[q] = -1; }
    else { D[q] = key_score(best); I[q] = (int64_t)key_id(best) + id_offset; }
}

// lists given as (D, I) pairs laid out [nlists][nq][k]
template <int R>
__global__ void merge_lists_kernel(const float* __restrict__ Dl, long long d_stride, const int64_t* __restrict__ Il,
                                   long long i_stride, int nlists, int nq, int k, float* __restrict__ D,
                                   int64_t* __restrict__ I) {
    extern __shared__ __align__(16) unsigned char smem[];
    skey_t* lds = reinterpret_cast<skey_t*>(smem);
    const int KP = WaveTopK<R>::KP;
    const int q = blockIdx.x;
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    // stage this wave's current list as keys in its LDS slot, then merge from there
    skey_t* slot = lds + (size_t)wave * KP;
    WaveTopK<R> t;
    t.init();
    for (int p = wave; p < nlists; p += nwaves) {
        const size_t dbase = (size_t)p * d_stride + (size_t)q * k, ibase = (size_t)p * i_stride + (size_t)q * k;
        for (int e = lane; e < k; e += 64) {
            const int64_t id = Il[ibase + e];
            slot[e] = (id < 0) ? 0 : make_key(Dl[dbase + e], (uint32_t)id);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave visible to itself
        __builtin_amdgcn_wave_barrier();
        t.merge_from(slot, k);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    block_merge<R>(t, lds, wave, nwaves);
    if (wave == 0) write_result<R>(t, k, D + (size_t)q * k, I + (size_t)q * k, 0);
}

static int pow2_ceil(int x) { int p = 1; while (p < x) p <<= 1; return p; }

template <typename F>
static hipError_t dispatch_r(int r, F f) {
    switch (r) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 4: return f(std::integral_constant<int, 4>());
        case 8: return f(std::integral_constant<int, 8>());
        case 16: return f(std::integral_constant<int, 16>());
    }
    return hipErrorInvalidValue;
}

static int merge_waves(int P, int kp) {
    int w = P < 16 ? (P < 1 ? 1 : P) : 16;
    if (kp == 64 && P > 4) w = (P * 64 / 256 + 1) < 16 ? (P * 64 / 256 + 1) : 16;
    while (w > 1 && (size_t)w * kp * sizeof(skey_t) > 64 * 1024) --w;
    return w;
}

hipError_t launch_merge_keys(const skey_t* partial, int nq, int P, int k, float* D, int64_t* I, int64_t id_offset,
                             hipStream_t s, unsigned min_ord) {
    if (nq <= 0) return hipSuccess;
    if (k == 1 && nq >= 1024) {
        hipLaunchKernelGGL(merge_top1_kernel, dim3((nq + 255) / 256), dim3(256), 0, s, partial, nq, P, D, I, id_offset, min_ord);
        return hipGetLastError();
    }
    const int r = pow2_ceil((k + 63) / 64);
    const int w = merge_waves(P, 64 * r);
    const size_t lds = (size_t)w * 64 * r * sizeof(skey_t);
    return dispatch_r(r, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        hipLaunchKernelGGL((merge_keys_kernel<R>), dim3(nq), dim3(64 * w), lds, s, partial, P, k, D, I, id_offset, min_ord);
        return hipGetLastError();
    });
}

hipError_t launch_merge_lists(const float* Dl, long long d_stride, const int64_t* Il, long long i_stride, int nlists,
                              int nq, int k, float* D, int64_t* I, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    const int r = pow2_ceil((k + 63) / 64);
    const int w = merge_waves(nlists, 64 * r);
    const size_t lds = (size_t)w * 64 * r * sizeof(skey_t);
    return dispatch_r(r, [&](auto rc) {
        constexpr int R = decltype(rc)::value;
        hipLaunchKernelGGL((merge_lists_kernel<R>), dim3(nq), dim3(64 * w), lds, s, Dl, d_stride, Il, i_stride, nlists, nq, k, D, I);
        return hipGetLastError();
    });
}

}  // namespace ssmi
