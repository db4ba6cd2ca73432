// mfma_n128.hip - K2n: Flat search of a MEDIUM query batch (5..128 queries on fp16 rows) as an
// MFMA stream over the database rows.
//
// With at most 128 queries the 256-query tile of mfma_pp.hip is mostly padding and its
// contraction time (2.2 ms per 1M x 4096 fp16 rows) exceeds the time to stream the rows
// (1.2 ms at 7 TB/s); the register-FMA scan (K1) is VALU-bound beyond 4 queries.  This kernel
// keeps a 128-query tile per workgroup and is organised as a deep LDS-DMA stream instead of a
// matrix-pipe schedule:
//
//   workgroup = 8 waves (4 x 2), tile = 256 rows x 128 queries, wave tile 64 x 64
//   (16 accumulators of 16x16, v_mfma_f32_16x16x32_f16), K walked in phases of 32;
//   a ring of SIX 24 KiB LDS slots (X slab 256 rows x 64 B + Q slab 128 queries x 64 B), filled by
//   LDS-DMA five phases ahead and ACROSS row-tile boundaries, so that 64-80 KiB of database rows are
//   in flight per CU at all times; ONE barrier per phase (it publishes slab p; slot p-1 is recycled
//   right behind it).  Measured 4.4 TB/s (1.82 ms per 1M x 4096 fp16 rows for <= 16 queries, 2.3 ms
//   at 128) against 2.2 ms of the 256-query kernel: bytes in flight, not the matrix pipe, bound it
//   (reading the next slab's fragments under the MFMAs changed nothing).
//
// Slab image, swizzle, fragment layout, threshold filter and list compaction are those of
// mfma_pp.hip (mfma_common.h).  Every workgroup owns one database slice and all queries.
// Stands behind faiss.IndexFlatIP.search for small query batches (batched search is the
// additive surface of SURVEY.md 8b; reference call site core/index_manager.py:371).
#include "mfma_common.h"

namespace ssmi {

static constexpr int NSLAB_X = 256 * 64;  // 16 KiB
static constexpr int NSLAB_Q = 128 * 64;  // 8 KiB
static_assert(NSLAB_X + NSLAB_Q == N128_SLOT, "slot size");

template <int R, bool BF = false>
__global__ void __launch_bounds__(512) flat_mfma_n128_kernel(MfmaParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + N128_RING * N128_SLOT);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + N128_TN);
    unsigned* kept_l = cnt_l + N128_TN;
    unsigned* flag_l = kept_l + N128_TN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int slice = blockIdx.x;
    const long long r_begin = (long long)slice * P.rows_per_slice;
    const long long r_end = (r_begin + P.rows_per_slice < P.X.nrows) ? (r_begin + P.rows_per_slice) : P.X.nrows;
    skey_t* cand = P.cand + (size_t)blockIdx.x * N128_TN * CAND_CAP;

    if (tid < N128_TN) {
        thr_l[tid] = (tid < P.nq) ? P.thr_floor : __builtin_inff();
        cnt_l[tid] = 0;
        kept_l[tid] = 0;
    }
    if (tid == 0) *flag_l = 0;
    __syncthreads();
    if (r_begin >= r_end) return;

    const int ld = P.X.ld;
    const int np = ld / 32;  // phases per row tile
    const int ntiles = (int)((r_end - r_begin + TM - 1) / TM);
    const int total = ntiles * np;
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    // DMA pieces (1 KiB = 16 rows x 64 B): wave w brings X pieces 2w, 2w+1 and Q piece w of every slab.
    // Lane i fills row i>>2, physical chunk i&3 = logical chunk (i&3) ^ f(row).
    int xrow[2], xcoff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        xrow[j] = (wave * 2 + j) * 16 + (lane >> 2);
        xcoff[j] = ((lane & 3) ^ ((-(xrow[j] >> 2)) & 3)) * 16;
    }
    const int qrow = wave * 16 + (lane >> 2);
    const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(P.Qh) + (size_t)qrow * ld * 2 +
                                ((lane & 3) ^ ((-(qrow >> 2)) & 3)) * 16;

    // fragment read offsets inside a slot
    const int frow = lane & 15, fq = lane >> 4;
    const int fch = (fq ^ ((-(frow >> 2)) & 3)) * 16;
    const int a_off = (wm * 64 + frow) * 64 + fch;
    const int b_off = NSLAB_X + (wn * 64 + frow) * 64 + fch;

    // issue cursor: phase `ig` = (row tile it, phase ip); its X source rows are recomputed per row tile.
    // All rows of a tile lie in one segment (slices and segments are multiples of the tile height).
    int ig = 0, it = 0, ip = 0;
    const unsigned char* asrc[2] = {nullptr, nullptr};
    auto issue_next = [&]() {
        if (ip == 0) {
            const long long t0 = r_begin + (long long)it * TM;
            const unsigned char* segbase = reinterpret_cast<const unsigned char*>(P.X.segs[t0 >> P.X.seg_shift]);
            const long long left = r_end - t0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const long long rr = (xrow[j] < left) ? xrow[j] : left - 1;  // padding rows repeat the last valid row
                asrc[j] = segbase + (size_t)((t0 & seg_mask) + rr) * (size_t)ld * 2 + xcoff[j];
            }
        }
        unsigned char* dst = smem + (ig % N128_RING) * N128_SLOT;
        const size_t koff = (size_t)ip * 64;
        GLDS16(asrc[0] + koff, dst + (wave * 2) * 1024);
        GLDS16(asrc[1] + koff, dst + (wave * 2 + 1) * 1024);
        GLDS16(qsrc + koff, dst + NSLAB_X + wave * 1024);
        ++ig;
        if (++ip == np) { ip = 0; ++it; }
    };
    for (int i = 0; i < N128_RING - 1 && ig < total; ++i) issue_next();

    f32x4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    unsigned tiles_done = 0;
    int p = 0;
    long long row0 = r_begin;
    for (int g = 0; g < total; ++g) {
        // own DMA of slab g landed (slabs g+1 .. g+4 may stay in flight: 3 instructions each), then publish it
        const int younger = (total - 1 - g) < (N128_RING - 2) ? (total - 1 - g) : (N128_RING - 2);
        switch (younger) {
            case 4: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        }
        __builtin_amdgcn_s_barrier();
        // every wave has finished reading slab g-1 (it passed the barrier): its slot takes slab g+5
        if (ig < total) issue_next();
        const unsigned char* slab = smem + (g % N128_RING) * N128_SLOT;
        half8_t a[4], b[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = *reinterpret_cast<const half8_t*>(slab + a_off + m * 16 * 64);
#pragma unroll
        for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const half8_t*>(slab + b_off + n * 16 * 64);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[m][n] = mfma16<BF>(a[m], b[n], acc[m][n]);
        __builtin_amdgcn_sched_barrier(0);

        if (++p == np) {  // row tile complete: filter its scores, maybe fold the candidate buffers
            p = 0;
            const long long left = r_end - row0;
            if (left >= TM) filter_tile<4, 4, true>(acc, wm * 64, wn * 64, frow, fq, row0, TM, cand, thr_l, cnt_l, flag_l);
            else filter_tile<4, 4, false>(acc, wm * 64, wn * 64, frow, fq, row0, (int)left, cand, thr_l, cnt_l, flag_l);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            row0 += TM;
            __syncthreads();
            ++tiles_done;
            if (((tiles_done & (tiles_done - 1)) == 0 && (tiles_done & 0x55555555u)) || *flag_l) {
                if (R == 1 && P.fold16) fold_queries16(wave, 0, P, slice, cand, thr_l, cnt_l, kept_l);
                else
                    for (int ql = wave; ql < N128_TN; ql += 8)
                        if (ql < P.nq) compact_query<R>(ql, ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
                __syncthreads();
                if (tid == 0) *flag_l = 0;
            }
        }
    }

    // ---- end of slice: fold what is left; the k-list is the partial result ----
    __syncthreads();
    if (R == 1 && P.fold16) fold_queries16(wave, 0, P, slice, cand, thr_l, cnt_l, kept_l);
    else
        for (int ql = wave; ql < N128_TN; ql += 8)
            if (ql < P.nq) compact_query<R>(ql, ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
}

hipError_t launch_flat_mfma_n128(const MfmaParams& P, int r, int grid, size_t lds, hipStream_t s) {
    hipError_t e;
#define LAUNCH_K(...)                                                                                   \
    {                                                                                                   \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_n128_kernel<__VA_ARGS__>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                  \
        hipLaunchKernelGGL((flat_mfma_n128_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, s, P);    \
    }
#define LAUNCH_R(RR) case RR: if (P.bf16) LAUNCH_K(RR, true) else LAUNCH_K(RR, false) break;
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8) LAUNCH_R(16)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
#undef LAUNCH_K
    return hipGetLastError();
}

}  // namespace ssmi
