// mfma.hip - K2: large-batch Flat search as an MFMA fp16 contraction with a fused
// top-k epilogue (the score matrix is never written).
//
// Stands behind faiss.IndexFlatIP.search for nq >> 1 (the reference only ever
// issues batch-1 searches, core/index_manager.py:371; batched search is the
// additive surface of SURVEY.md 8b).
//
//   S[row][query] = sum_k X[row][k] * Q[query][k]      fp16 x fp16 -> fp32
//
// Workgroup = 8 waves (2 x 4), tile = 256 database rows x 256 queries x BK 64.
// Both operands are K-contiguous in memory, so a tile of either is 256 rows of
// 128 bytes; it is brought in by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB
// per wave-instruction = 8 rows) into a double-buffered LDS image whose 16-byte
// chunks are XOR-swizzled by (row & 7) on the SOURCE address, so that the
// ds_read_b128 fragment reads of v_mfma_f32_16x16x32_f16 are bank-conflict free.
// Each wave owns a 128 x 64 sub-tile = 8 x 4 accumulators of 16 x 16.
//
// A workgroup keeps its query tile and walks a slice of the database (persistent
// along N).  C/D layout puts the query on the lane (col = lane & 15) and four
// rows in the registers, so the epilogue is one compare per accumulator element
// against that query's running threshold (LDS); survivors are appended as 64-bit
// keys to a per-(workgroup, query) candidate buffer.  When a buffer reaches
// TRIG entries a wave folds it into the query's sorted k-list (topk.h) and
// raises the threshold.  At the end of the slice the k-list IS the per-slice
// partial result that K3 (merge.hip) merges across slices.
#include "mfma_common.h"

namespace ssmi {

template <int R>
__global__ void __launch_bounds__(512) flat_mfma_topk_kernel(MfmaParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + 4 * TILE_BYTES);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + 256);
    unsigned* kept_l = cnt_l + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // XCD-aware block order: blocks b and b+8 share an XCD, so give each XCD a
    // contiguous range of logical ids; consecutive logical ids are the query
    // tiles of ONE database slice and share its X tiles through that XCD's L2.
    const int nblocks = gridDim.x;
    int logical = blockIdx.x;
    if ((nblocks & 7) == 0) logical = (blockIdx.x & 7) * (nblocks >> 3) + (blockIdx.x >> 3);
    const int slice = logical / P.nqt, qt = logical % P.nqt;
    const int q0 = qt * TN;
    const long long r_begin = (long long)slice * P.rows_per_slice;
    const long long r_end = (r_begin + P.rows_per_slice < P.X.nrows) ? (r_begin + P.rows_per_slice) : P.X.nrows;
    skey_t* cand = P.cand + (size_t)blockIdx.x * TN * CAND_CAP;

    if (tid < 256) {
        thr_l[tid] = (q0 + tid < P.nq) ? SSMI_NEG_MAX : __builtin_inff();
        cnt_l[tid] = 0;
        kept_l[tid] = 0;
    }
    __syncthreads();

    const int ld = P.X.ld;
    const int nk = ld / BK;
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    // LDS-DMA slots of this thread: 4 per operand tile.  Slot j of wave w fills
    // physical 16-byte chunks (w*4+j)*64 + lane of the tile image; physical chunk
    // p = row*8 + pc holds logical chunk pc ^ (row & 7) of that row.
    int slot_row[4], slot_coff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = (wave * 4 + j) * 64 + lane;
        slot_row[j] = p >> 3;
        slot_coff[j] = ((p & 7) ^ ((p >> 3) & 7)) * 16;
    }
    const unsigned char* qsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        qsrc[j] = reinterpret_cast<const unsigned char*>(P.Qh) + (size_t)(q0 + slot_row[j]) * ld * 2 + slot_coff[j];

    // fragment read offsets (bytes inside an operand tile image), without the k-substep part
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[2], b_off[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c = (kk * 4 + fq) ^ (frow & 7);
        a_off[kk] = ((wm * 128 + frow) * 8 + c) * 16;
        b_off[kk] = ((wn * 64 + frow) * 8 + c) * 16;
    }

    for (long long row0 = r_begin; row0 < r_end; row0 += TM) {
        const unsigned char* asrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long long r = row0 + slot_row[j];
            if (r >= r_end) r = r_end - 1;  // tail tile: re-read the last row, masked in the epilogue
            asrc[j] = reinterpret_cast<const unsigned char*>(P.X.segs[r >> P.X.seg_shift]) +
                      (size_t)(r & seg_mask) * (size_t)ld * 2 + slot_coff[j];
        }

        f32x4_t acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // the pointer-table loads above must have landed before the counted waits below
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

        // prologue: tile 0 -> buffer 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            GLDS16(asrc[j], smem + (wave * 4 + j) * 1024);
            GLDS16(qsrc[j], smem + TILE_BYTES + (wave * 4 + j) * 1024);
        }

        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) {
                unsigned char* dst = smem + (buf ^ 1) * 2 * TILE_BYTES;
                const size_t koff = (size_t)(kt + 1) * BK * 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    GLDS16(asrc[j] + koff, dst + (wave * 4 + j) * 1024);
                    GLDS16(qsrc[j] + koff, dst + TILE_BYTES + (wave * 4 + j) * 1024);
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // tile kt landed, tile kt+1 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();

            const unsigned char* As = smem + buf * 2 * TILE_BYTES;
            const unsigned char* Bs = As + TILE_BYTES;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                half8_t a[8], b[4];
#pragma unroll
                for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const half8_t*>(As + a_off[kk] + m * 16 * 128);
#pragma unroll
                for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const half8_t*>(Bs + b_off[kk] + n * 16 * 128);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();  // everyone is done reading buf before it is refilled
        }

        // ---- epilogue: threshold filter, survivors -> candidate buffers ----
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int ql = wn * 64 + n * 16 + frow;
            const float thr = thr_l[ql];
            skey_t* cq = cand + (size_t)ql * CAND_CAP;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float s = acc[m][n][i];
                    const long long r = row0 + wm * 128 + m * 16 + fq * 4 + i;
                    if (s >= thr && r < r_end) {
                        const unsigned pos = atomicAdd(&cnt_l[ql], 1u);
                        if (pos < (unsigned)CAND_CAP) cq[pos] = make_key(s, (uint32_t)r);
                    }
                }
            }
        }
        __syncthreads();
        const int need = (tid < 256) ? (cnt_l[tid] >= (unsigned)CAND_TRIG) : 0;
        if (__syncthreads_or(need)) {
            for (int ql = wave; ql < 256; ql += 8)
                if (cnt_l[ql] >= (unsigned)CAND_TRIG && q0 + ql < P.nq)
                    compact_query<R>(ql, q0 + ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
            __syncthreads();
        }
    }

    // ---- end of slice: fold what is left; the k-list is the partial result ----
    __syncthreads();
    for (int ql = wave; ql < 256; ql += 8)
        if (q0 + ql < P.nq) compact_query<R>(ql, q0 + ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
}

__global__ void cvt_queries_f16_kernel(const float* __restrict__ q, int nq, int ld, _Float16* __restrict__ out) {
    const int i = blockIdx.x;  // padded row
    for (int e = threadIdx.x; e < ld; e += blockDim.x)
        out[(size_t)i * ld + e] = (i < nq) ? (_Float16)q[(size_t)i * ld + e] : (_Float16)0.f;
}

bool plan_flat_mfma(int dtype, int ld, long long nrows, int nq, int k, int num_cu, MfmaPlan* p) {
    if (dtype != 1 || (ld % BK) != 0 || nrows < 1 || nq < 1 || k < 1 || k > kMaxK) return false;
    p->nqt = (nq + TN - 1) / TN;
    const long long row_tiles = (nrows + TM - 1) / TM;
    long long ns = num_cu / p->nqt;
    if (ns < 1) ns = 1;
    if (ns > row_tiles) ns = row_tiles;
    p->rows_per_slice = (row_tiles + ns - 1) / ns * TM;
    p->nslices = (int)((nrows + p->rows_per_slice - 1) / p->rows_per_slice);
    p->grid = p->nqt * p->nslices;
    p->qh_bytes = (size_t)p->nqt * TN * ld * 2;
    p->cand_bytes = (size_t)p->grid * TN * CAND_CAP * sizeof(skey_t);
    p->lds = 4 * TILE_BYTES + 3 * 256 * 4;
    return true;
}

hipError_t launch_flat_mfma(const RowView& X, const float* Q, int nq, int k, const MfmaPlan& p, void* qh, void* cand,
                            skey_t* partial, hipStream_t s) {
    hipLaunchKernelGGL(cvt_queries_f16_kernel, dim3(p.nqt * TN), dim3(256), 0, s, Q, nq, X.ld, (_Float16*)qh);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    MfmaParams P;
    P.X = X; P.Qh = (const _Float16*)qh; P.nq = nq; P.k = k; P.nqt = p.nqt; P.nslices = p.nslices;
    P.rows_per_slice = p.rows_per_slice; P.partial = partial; P.cand = (skey_t*)cand;
    int r = 1;
    while (64 * r < k) r <<= 1;
    const char* env = getenv("SSMI_MFMA_VARIANT");
    if (!(env && env[0] == '1')) return launch_flat_mfma_pp(P, r, p.grid, p.lds, s);
#define LAUNCH_R(RR)                                                                                              \
    case RR: {                                                                                                    \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_topk_kernel<RR>),                        \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);                          \
        if (e != hipSuccess) return e;                                                                            \
        hipLaunchKernelGGL((flat_mfma_topk_kernel<RR>), dim3(p.grid), dim3(512), p.lds, s, P);                   \
        break;                                                                                                    \
    }
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8) LAUNCH_R(16)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
    return hipGetLastError();
}

}  // namespace ssmi
