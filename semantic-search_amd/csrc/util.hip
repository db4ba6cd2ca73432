// util.hip - small data-movement kernels around the row store:
//   K4  cvt_store         add():  (n,d) fp32/fp16/bf16 rows -> padded storage rows
//   K9  gather_rows       reconstruct_n(): storage rows by id -> (n,d) fp32
//   K10 fill_random_unit  bench/test data: N(0,1) components, unit L2 norm
// Reference call sites these stand behind: index.add (core/index_manager.py:319),
// index.reconstruct (core/index_factory.py:274, core/index_manager.py:609).
#include "kernels.h"
#include "topk.h"

namespace ssmi {

__device__ __forceinline__ unsigned char* row_ptr_rw(void* const* segs, int seg_shift, long long row, size_t row_bytes) {
    const long long mask = (1ll << seg_shift) - 1;
    return reinterpret_cast<unsigned char*>(segs[row >> seg_shift]) + (size_t)(row & mask) * row_bytes;
}

template <typename S, typename T>
__global__ void cvt_store_kernel(const S* __restrict__ src, long long n, int d, int ld, void* const* segs,
                                 int seg_shift, long long row0) {
    const long long i = blockIdx.x;
    if (i >= n) return;
    T* dst = reinterpret_cast<T*>(row_ptr_rw(segs, seg_shift, row0 + i, (size_t)ld * sizeof(T)));
    const S* s = src + (size_t)i * d;
    for (int e = threadIdx.x; e < ld; e += blockDim.x) {
        const float v = (e < d) ? (float)s[e] : 0.f;
        dst[e] = (T)v;
    }
}

hipError_t launch_cvt_store(const void* src, int src_dtype, long long n, int d, const RowView& X, int dst_dtype,
                            void* const* segs_rw, long long row0, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)n), block(256);
    if (!dtype_known(src_dtype) || !dtype_known(dst_dtype)) return hipErrorInvalidValue;
#define CVT(S, T) hipLaunchKernelGGL((cvt_store_kernel<S, T>), grid, block, 0, s, (const S*)src, n, d, X.ld, segs_rw, X.seg_shift, row0)
#define CVT_TO(S) { if (dst_dtype == 0) CVT(S, float); else if (dst_dtype == 1) CVT(S, __half); else CVT(S, bf16_t); }
    if (src_dtype == 0) CVT_TO(float)
    else if (src_dtype == 1) CVT_TO(__half)
    else CVT_TO(bf16_t)
#undef CVT_TO
#undef CVT
    return hipGetLastError();
}

template <typename T>
__global__ void gather_rows_kernel(RowView X, const int64_t* __restrict__ ids, long long n, int d,
                                   float* __restrict__ out, int* __restrict__ bad) {
    const long long i = blockIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    float* o = out + (size_t)i * d;
    if (id < 0 || id >= X.nrows) {
        for (int e = threadIdx.x; e < d; e += blockDim.x) o[e] = 0.f;
        if (threadIdx.x == 0) atomicExch(bad, 1);
        return;
    }
    const long long mask = (1ll << X.seg_shift) - 1;
    const T* src = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(X.segs[id >> X.seg_shift]) +
                                              (size_t)(id & mask) * (size_t)X.ld * sizeof(T));
    for (int e = threadIdx.x; e < d; e += blockDim.x) o[e] = (float)src[e];
}

hipError_t launch_gather_rows(const RowView& X, int dtype, const int64_t* ids, long long n, int d, float* out,
                              int* bad, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)n), block(256);
    if (dtype == 0) hipLaunchKernelGGL((gather_rows_kernel<float>), grid, block, 0, s, X, ids, n, d, out, bad);
    else if (dtype == 1) hipLaunchKernelGGL((gather_rows_kernel<__half>), grid, block, 0, s, X, ids, n, d, out, bad);
    else hipLaunchKernelGGL((gather_rows_kernel<bf16_t>), grid, block, 0, s, X, ids, n, d, out, bad);
    return hipGetLastError();
}

// candidates x negatives inner products straight from the row store (the exclusion filter of
// core/index_manager.py:604-621: v = index.reconstruct(id); np.dot(v, neg)): one block per candidate id,
// fp32 products and sums whatever the storage type; out is (n, n_neg) row-major.
template <typename T>
__global__ void __launch_bounds__(256) score_rows_kernel(RowView X, const int64_t* __restrict__ ids, long long n, int d,
                                                         const float* __restrict__ neg, int n_neg, float* __restrict__ out,
                                                         int* __restrict__ bad) {
    __shared__ float part[4];
    const long long i = blockIdx.x;
    if (i >= n) return;
    const int64_t id = ids[i];
    if (id < 0 || id >= X.nrows) {
        for (int j = threadIdx.x; j < n_neg; j += blockDim.x) out[(size_t)i * n_neg + j] = 0.f;
        if (threadIdx.x == 0) atomicExch(bad, 1);
        return;
    }
    const long long mask = (1ll << X.seg_shift) - 1;
    const T* src = reinterpret_cast<const T*>(reinterpret_cast<const unsigned char*>(X.segs[id >> X.seg_shift]) +
                                              (size_t)(id & mask) * (size_t)X.ld * sizeof(T));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int j = 0; j < n_neg; ++j) {
        const float* q = neg + (size_t)j * d;
        float acc = 0.f;
        for (int e = threadIdx.x; e < d; e += blockDim.x) acc = fmaf((float)src[e], q[e], acc);
        acc = wave_sum(acc);
        if (lane == 0) part[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) out[(size_t)i * n_neg + j] = (part[0] + part[1]) + (part[2] + part[3]);
        __syncthreads();
    }
}

hipError_t launch_score_rows(const RowView& X, int dtype, const int64_t* ids, long long n, int d, const float* neg,
                             int n_neg, float* out, int* bad, hipStream_t s) {
    if (n <= 0 || n_neg <= 0) return hipSuccess;
    const dim3 grid((unsigned)n), block(256);
    if (dtype == 0) hipLaunchKernelGGL((score_rows_kernel<float>), grid, block, 0, s, X, ids, n, d, neg, n_neg, out, bad);
    else if (dtype == 1) hipLaunchKernelGGL((score_rows_kernel<__half>), grid, block, 0, s, X, ids, n, d, neg, n_neg, out, bad);
    else hipLaunchKernelGGL((score_rows_kernel<bf16_t>), grid, block, 0, s, X, ids, n, d, neg, n_neg, out, bad);
    return hipGetLastError();
}

// ---- Philox4x32-10 ---------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t* out) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void normal4(long long row, int quad, uint64_t seed, float* z) {
    uint32_t r[4];
    philox4x32_10((uint32_t)quad, (uint32_t)(row & 0xFFFFFFFFll), (uint32_t)((unsigned long long)row >> 32), 0x5eed5eedu,
                  (uint32_t)(seed & 0xFFFFFFFFull), (uint32_t)(seed >> 32), r);
    const float u1 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    const float u2 = (float)r[1] * 2.3283064365386963e-10f;
    const float u3 = ((float)r[2] + 1.0f) * 2.3283064365386963e-10f;
    const float u4 = (float)r[3] * 2.3283064365386963e-10f;
    const float ra = sqrtf(-2.f * __logf(u1 > 1e-30f ? u1 : 1e-30f));
    const float rb = sqrtf(-2.f * __logf(u3 > 1e-30f ? u3 : 1e-30f));
    float sa, ca, sb, cb;
    __sincosf(6.283185307179586f * u2, &sa, &ca);
    __sincosf(6.283185307179586f * u4, &sb, &cb);
    z[0] = ra * ca; z[1] = ra * sa; z[2] = rb * cb; z[3] = rb * sb;
}

// one wave per row; two passes (norm, then regenerate + scale + store)
template <typename T>
__global__ void fill_random_unit_kernel(void* const* segs, int seg_shift, int ld, long long row0, long long n, int d,
                                        uint64_t seed) {
    const int lane = lane_id();
    const long long i = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= n) return;
    const long long row = row0 + i;
    const int nquad = (ld + 3) / 4;
    float ss = 0.f;
    for (int qd = lane; qd < nquad; qd += 64) {
        float z[4];
        normal4(row, qd, seed, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) if (qd * 4 + j < d) ss += z[j] * z[j];
    }
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss > 1e-30f ? ss : 1e-30f);
    T* dst = reinterpret_cast<T*>(row_ptr_rw(segs, seg_shift, row, (size_t)ld * sizeof(T)));
    for (int qd = lane; qd < nquad; qd += 64) {
        float z[4];
        normal4(row, qd, seed, z);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = qd * 4 + j;
            if (e < ld) dst[e] = (T)((e < d) ? z[j] * inv : 0.f);
        }
    }
}

hipError_t launch_fill_random_unit(const RowView& X, int dtype, void* const* segs_rw, long long row0, long long n,
                                   int d, uint64_t seed, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    const int wpb = 4;
    const dim3 grid((unsigned)((n + wpb - 1) / wpb)), block(64 * wpb);
    if (dtype == 0) hipLaunchKernelGGL((fill_random_unit_kernel<float>), grid, block, 0, s, segs_rw, X.seg_shift, X.ld, row0, n, d, seed);
    else if (dtype == 1) hipLaunchKernelGGL((fill_random_unit_kernel<__half>), grid, block, 0, s, segs_rw, X.seg_shift, X.ld, row0, n, d, seed);
    else hipLaunchKernelGGL((fill_random_unit_kernel<bf16_t>), grid, block, 0, s, segs_rw, X.seg_shift, X.ld, row0, n, d, seed);
    return hipGetLastError();
}

// fp32 store rows [row0,row0+n) -> 16-bit store rows (same row numbers): the shadow that lets large query batches
// over an fp32 index run on the MFMA kernel.  BF: bf16 shadow (else fp16); *inexact (optional) is set when a value
// does not survive the narrowing bit for bit (bf16: any of the low 16 bits set; fp16: the round trip differs).
template <bool BF>
__global__ void shadow_rows_kernel(RowView src, void* const* dsegs, int dshift, long long row0, long long n, int* inexact) {
    const long long i = blockIdx.x;
    if (i >= n) return;
    const long long r = row0 + i;
    const float* s = reinterpret_cast<const float*>(reinterpret_cast<const unsigned char*>(src.segs[r >> src.seg_shift]) +
                                                    (size_t)(r & ((1ll << src.seg_shift) - 1)) * (size_t)src.ld * 4);
    unsigned char* d = row_ptr_rw(dsegs, dshift, r, (size_t)src.ld * 2);
    bool lossy = false;
    for (int e = threadIdx.x * 4; e < src.ld; e += blockDim.x * 4) {
        const float4 v = *reinterpret_cast<const float4*>(s + e);
        uint2 o;
        if (BF) {
            const bf16_t a(v.x), b(v.y), c(v.z), dd(v.w);
            o.x = (unsigned)a.x | ((unsigned)b.x << 16);
            o.y = (unsigned)c.x | ((unsigned)dd.x << 16);
            lossy |= (float)a != v.x || (float)b != v.y || (float)c != v.z || (float)dd != v.w;
        } else {
            __half2 a = __floats2half2_rn(v.x, v.y), b = __floats2half2_rn(v.z, v.w);
            o.x = *reinterpret_cast<unsigned*>(&a);
            o.y = *reinterpret_cast<unsigned*>(&b);
            const float2 ra = __half22float2(a), rb = __half22float2(b);
            lossy |= ra.x != v.x || ra.y != v.y || rb.x != v.z || rb.y != v.w;
        }
        *reinterpret_cast<uint2*>(d + (size_t)e * 2) = o;
    }
    if (inexact && __builtin_amdgcn_ballot_w64(lossy) != 0 && lane_id() == 0) atomicExch(inexact, 1);
}
hipError_t launch_shadow_rows(const RowView& src, void* const* dsegs, int dshift, int dst_dtype, long long row0, long long n,
                              int* inexact, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    if (dst_dtype == 2) hipLaunchKernelGGL(shadow_rows_kernel<true>, dim3((unsigned)n), dim3(256), 0, s, src, dsegs, dshift, row0, n, inexact);
    else hipLaunchKernelGGL(shadow_rows_kernel<false>, dim3((unsigned)n), dim3(256), 0, s, src, dsegs, dshift, row0, n, inexact);
    return hipGetLastError();
}

__global__ void fill_empty_kernel(float* D, int64_t* I, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { D[i] = SSMI_NEG_MAX; I[i] = -1; }
}
hipError_t launch_fill_empty(float* D, int64_t* I, long long n, hipStream_t s) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, D, I, n);
    return hipGetLastError();
}

__global__ void pad_queries_kernel(const float* __restrict__ q, int d, int ld, float* __restrict__ out) {
    const int i = blockIdx.x;
    for (int e = threadIdx.x; e < ld; e += blockDim.x) out[(size_t)i * ld + e] = (e < d) ? q[(size_t)i * d + e] : 0.f;
}
hipError_t launch_pad_queries(const float* q, int nq, int d, int ld, float* out, hipStream_t s) {
    if (nq <= 0) return hipSuccess;
    hipLaunchKernelGGL(pad_queries_kernel, dim3(nq), dim3(256), 0, s, q, d, ld, out);
    return hipGetLastError();
}

}  // namespace ssmi
