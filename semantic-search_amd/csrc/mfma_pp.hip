// mfma_pp.hip - K2, ping-pong variant: the same contraction and fused top-k
// epilogue as mfma.hip, re-scheduled so that the matrix pipe of every SIMD is
// fed continuously.
//
// Workgroup = 8 waves = two groups of four (group g = waves 4g..4g+3, one wave
// of each group per SIMD).  The K dimension is walked in PHASES of 32: phase p
// needs one 16 KiB slab of X (256 rows x 64 B) and one of Q.  Slabs live in a
// ring of four (128 KiB LDS) and are filled by LDS-DMA three phases ahead.
//
//   group body, phase p:   ds_read fragments of slab p            (LOAD segment)
//                          s_waitcnt vmcnt(4)   -> own DMA of slab p+1 landed
//                          s_barrier
//                          s_waitcnt lgkmcnt(0)
//                          issue DMA of slab p+3, 32 MFMAs          (MFMA segment)
//                          s_barrier
//
// Group 1 runs one barrier behind group 0, so between any two barrier events one
// group is in its MFMA segment while the other reads LDS: the matrix pipe only
// idles when a LOAD segment outlasts 32 MFMAs.  Hazards (all by construction):
//   RAW  slab p+1 is read one phase after every wave's vmcnt wait + a barrier
//   WAR  slab p+3 overwrites slab p-1, whose last ds_reads were retired by an
//        lgkmcnt(0) at least one barrier event earlier
// Slab image: 64-byte rows, 16-byte chunk c of row r stored at chunk
// c ^ ((-(r >> 2)) & 3) (applied to the DMA SOURCE address and to the reads),
// which makes every ds_read_b128 fragment read bank-conflict free.
#include "mfma_common.h"

namespace ssmi {

static constexpr int SLAB = 256 * 64;          // one operand slab: 16 KiB
static constexpr int PHASE_LDS = 2 * SLAB;     // X slab + Q slab
static constexpr int RING = 4;

template <int R>
__global__ void __launch_bounds__(512) flat_mfma_pp_kernel(MfmaParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + RING * PHASE_LDS);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + 256);
    unsigned* kept_l = cnt_l + 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;  // wm == ping-pong group

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
    const int np = ld / 32;  // phases per row tile
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    // DMA slots: slab = 16 wave-instructions of 1 KiB (16 rows x 64 B); wave w owns
    // instructions 2w, 2w+1 of the X slab and of the Q slab.  Lane i fills row i>>2,
    // physical chunk i&3, i.e. logical chunk (i&3) ^ f(row).
    int slot_row[2], slot_coff[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 16 + (lane >> 2);
        slot_row[j] = row;
        slot_coff[j] = (((lane & 3) ^ ((-(row >> 2)) & 3))) * 16;
    }
    const unsigned char* qsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
        qsrc[j] = reinterpret_cast<const unsigned char*>(P.Qh) + (size_t)(q0 + slot_row[j]) * ld * 2 + slot_coff[j];

    // fragment read offsets inside a slab
    const int frow = lane & 15, fq = lane >> 4;
    const int fch = (fq ^ ((-(frow >> 2)) & 3)) * 16;
    const int a_off = (wm * 128 + frow) * 64 + fch;
    const int b_off = SLAB + (wn * 64 + frow) * 64 + fch;

    const unsigned char* asrc[2];
    auto set_rows = [&](long long row0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            long long r = row0 + slot_row[j];
            if (r >= r_end) r = r_end - 1;
            asrc[j] = reinterpret_cast<const unsigned char*>(P.X.segs[r >> P.X.seg_shift]) +
                      (size_t)(r & seg_mask) * (size_t)ld * 2 + slot_coff[j];
        }
    };
    auto issue = [&](int p) {  // DMA of phase p's slabs into ring slot p & 3
        unsigned char* dst = smem + (p & (RING - 1)) * PHASE_LDS + (wave * 2) * 1024;
        const size_t koff = (size_t)p * 64;
        GLDS16(asrc[0] + koff, dst);
        GLDS16(asrc[1] + koff, dst + 1024);
        GLDS16(qsrc[0] + koff, dst + SLAB);
        GLDS16(qsrc[1] + koff, dst + SLAB + 1024);
    };

    if (r_begin < r_end) {
        set_rows(r_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        issue(0);
        if (1 < np) issue(1);
        if (2 < np) issue(2);
    }

    for (long long row0 = r_begin; row0 < r_end; row0 += TM) {
        f32x4_t acc[8][4];
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        // slab 0 of this row tile: wait for every wave's DMA (the epilogue's stores of the
        // previous tile are younger in the queue, hence a full drain), then make it visible
        if (np >= 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier event behind

        for (int p = 0; p < np; ++p) {
            const unsigned char* slab = smem + (p & (RING - 1)) * PHASE_LDS;
            half8_t a[8], b[4];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const half8_t*>(slab + a_off + m * 16 * 64);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const half8_t*>(slab + b_off + n * 16 * 64);
            // own DMA of slab p+1 must have landed before this phase's first barrier
            if (p + 2 < np) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            // the four DMA instructions of slab p+3 are spread between the MFMAs so that their
            // issue cost hides under matrix-pipe time
            const bool pre = (p + 3 < np);
            unsigned char* dst = smem + ((p + 3) & (RING - 1)) * PHASE_LDS + (wave * 2) * 1024;
            const size_t koff = (size_t)(p + 3) * 64;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
                if (pre) {
                    if (m == 0) GLDS16(asrc[0] + koff, dst);
                    if (m == 2) GLDS16(asrc[1] + koff, dst + 1024);
                    if (m == 4) GLDS16(qsrc[0] + koff, dst + SLAB);
                    if (m == 6) GLDS16(qsrc[1] + koff, dst + SLAB + 1024);
                }
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();  // realign the two groups

        // start the next row tile's pipeline before the epilogue: every ring slot is free now
        const long long next0 = row0 + TM;
        if (next0 < r_end) {
            set_rows(next0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            issue(0);
            if (1 < np) issue(1);
            if (2 < np) issue(2);
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

    __syncthreads();
    for (int ql = wave; ql < 256; ql += 8)
        if (q0 + ql < P.nq) compact_query<R>(ql, q0 + ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
}

hipError_t launch_flat_mfma_pp(const MfmaParams& P, int r, int grid, size_t lds, hipStream_t s) {
    hipError_t e;
#define LAUNCH_R(RR)                                                                                    \
    case RR: {                                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_pp_kernel<RR>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                  \
        hipLaunchKernelGGL((flat_mfma_pp_kernel<RR>), dim3(grid), dim3(512), lds, s, P);               \
        break;                                                                                          \
    }
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8) LAUNCH_R(16)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
    return hipGetLastError();
}

}  // namespace ssmi
