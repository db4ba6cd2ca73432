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
// The ring runs on across row tiles: when the next row tile is whole and lies in the
// same storage segment, the last three phases of a tile fetch slabs 0..2 of the next
// one (its rows are one tile stride further on), so the epilogue is not followed by a
// pipeline fill.  Otherwise (segment boundary, last tile, phases not a multiple of the
// ring) the fill is issued after the K loop, before the epilogue.
// Slab image: 64-byte rows, 16-byte chunk c of row r stored at chunk
// c ^ ((-(r >> 2)) & 3) (applied to the DMA SOURCE address and to the reads),
// which makes every ds_read_b128 fragment read bank-conflict free.
#include "mfma_common.h"
#ifdef SSMI_STAMP
#include <cstdio>
#include <vector>
#include <algorithm>
#endif

#ifndef SSMI_PP_PRIO
#define SSMI_PP_PRIO 0  // 0: setprio(1) around every MFMA segment; 1: static priority for waves 4-7; 2: none
#endif

namespace ssmi {

static constexpr int SLAB = 256 * 64;          // one operand slab: 16 KiB
static constexpr int PHASE_LDS = 2 * SLAB;     // X slab + Q slab
static constexpr int RING = 4;

// ARGMAX (k == 1: IVF centroid assignment, K5): no threshold, candidate buffers or folds at all - every lane keeps
// the best (score, row) of each of its four query columns across the row tiles of the slice, the four lanes and two
// waves that share a column combine once at the end.  Assignment batches have no score structure to prune with
// (every centroid scores alike on unclustered rows), so the filter path spent its time on survivors there.
template <int R, bool ARGMAX = false>
__global__ void __launch_bounds__(512) flat_mfma_pp_kernel(MfmaParams P) {
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + RING * PHASE_LDS);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + 256);
    unsigned* kept_l = cnt_l + 256;
    unsigned* flag_l = kept_l + 256;

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
#ifdef SSMI_STAMP_NOPASS
        thr_l[tid] = __builtin_inff();
#else
        thr_l[tid] = (q0 + tid < P.nq) ? P.thr_floor : __builtin_inff();
#endif
        cnt_l[tid] = 0;
        kept_l[tid] = 0;
    }
    if (tid == 0) *flag_l = 0;
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
    // DMA of phase PH's slabs (PH = 0, 1, 2: the pipeline fill of a row tile) into ring slot PH; the K
    // offset rides in the instruction's immediate field, so the fill needs no extra address registers
#define ISSUE_FILL(PH)                                                                       \
    {                                                                                        \
        unsigned char* dst_ = smem + (PH) * PHASE_LDS + (wave * 2) * 1024;                   \
        GLDS16O(asrc[0], dst_, (PH) * 64);                                                   \
        GLDS16O(asrc[1], dst_ + 1024, (PH) * 64);                                            \
        GLDS16O(qsrc[0], dst_ + SLAB, (PH) * 64);                                            \
        GLDS16O(qsrc[1], dst_ + SLAB + 1024, (PH) * 64);                                     \
    }

    if (r_begin < r_end) {
        set_rows(r_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ISSUE_FILL(0)
        if (1 < np) ISSUE_FILL(1)
        if (2 < np) ISSUE_FILL(2)
    }

#ifdef SSMI_STAMP
    unsigned long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stv[i] += now_ - stl_; stl_ = now_; }
    unsigned long long stl_ = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#endif
    unsigned tiles_done = 0;
    // Cross-slice bound in two cheap steps after a fold (tops_bound is a memory round trip and a 64-key sort per query,
    // too much for every slice to do for all 256 queries): one row tile after the fold, when every slice of the
    // query tile has published its tops, this slice computes the bound of ITS share of the queries (ql = slice mod
    // nslices) and pushes it into gthr; another row tile later every wave pulls gthr for its queries.
    // scheduled folds after row tiles 1, 4, 16, 64, ...; with the cross-slice bound (which matures much faster than a
    // slice's own k-th score) after 1, 16, 256, ... only
    const unsigned fold_mask = P.fold_mask;
    int xstate = 0;  // 1: push bounds at the next tile end, 2: pull thresholds at the next tile end
    auto push_bounds = [&]() {
        for (int ql = slice + wave * P.nslices; ql < 256 && q0 + ql < P.nq; ql += 8 * P.nslices) {
            const unsigned b = tops_bound(P, q0 + ql);
            if (b && lane == 0) atomicMax(&P.gthr[q0 + ql], b);
        }
    };
    auto pull_thr = [&]() {
        if (lane < 32) {
            const int ql = wave + 8 * lane;
            if (q0 + ql < P.nq) {
                const unsigned seen = __hip_atomic_load(&P.gthr[q0 + ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen && ord_to_f32(seen) > thr_l[ql]) lds_store(reinterpret_cast<unsigned*>(&thr_l[ql]), __float_as_uint(ord_to_f32(seen)));
            }
        }
    };
    float best_s[4];
    uint32_t best_r[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) { best_s[n] = -__builtin_inff(); best_r[n] = 0xFFFFFFFFu; }
    const size_t tile_stride = (size_t)TM * (size_t)ld * 2;
    for (long long row0 = r_begin; row0 < r_end; row0 += TM) {
        // The ring runs on ACROSS row tiles when the next tile is whole and lies in the same storage segment
        // (its rows are then tile_stride bytes further on, no segment lookup) and the tile's phases fill whole
        // turns of the ring: the last three phases fetch the next tile's slabs 0..2 instead of nothing, so the
        // fill's HBM round trip hides behind MFMAs and the epilogue instead of following it.
        const long long next0 = row0 + TM;
        const bool cont = (np & (RING - 1)) == 0 && np >= RING && next0 + TM <= r_end &&
                          (row0 >> P.X.seg_shift) == ((next0 + TM - 1) >> P.X.seg_shift);
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
        STAMP(0)

#if SSMI_PP_PRIO == 1
        if (wm == 1) __builtin_amdgcn_s_setprio(1);  // static priority for the younger half, no per-segment flips
#endif
        for (int p = 0; p < np; ++p) {
            const unsigned char* slab = smem + (p & (RING - 1)) * PHASE_LDS;
            // ---- LOAD segment (the partner group owns the matrix pipe meanwhile) ----
            // DMA of slab p+3 into the ring slot of slab p-1: every ds_read of slab p-1 was retired by
            // an lgkmcnt(0) that its wave executed BEFORE a barrier event this wave has already passed.
            if (p + 3 < np || cont) {
                unsigned char* dst = smem + ((p + 3) & (RING - 1)) * PHASE_LDS + (wave * 2) * 1024;
                const bool here = p + 3 < np;
                const size_t qoff = (size_t)(here ? p + 3 : p + 3 - np) * 64;
                const size_t koff = here ? qoff : tile_stride + qoff;
                GLDS16(asrc[0] + koff, dst);
                GLDS16(asrc[1] + koff, dst + 1024);
                GLDS16(qsrc[0] + qoff, dst + SLAB);
                GLDS16(qsrc[1] + qoff, dst + SLAB + 1024);
            }
            half8_t a[8], b[4];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = *reinterpret_cast<const half8_t*>(slab + a_off + m * 16 * 64);
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const half8_t*>(slab + b_off + n * 16 * 64);
            // own DMA of slab p+1 must have landed before this phase's first barrier; the four
            // instructions just issued (slab p+3) and the four of slab p+2 may stay in flight
            if (p + 3 < np || cont) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (p + 2 < np) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers: slab p may be recycled
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            // ---- MFMA segment ----
#if SSMI_PP_PRIO == 0
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], b[n], acc[m][n], 0, 0, 0);
#if SSMI_PP_PRIO == 0
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
        }
#if SSMI_PP_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        if (wm == 0) __builtin_amdgcn_s_barrier();  // realign the two groups
        STAMP(1)

        // start the next row tile's pipeline before the epilogue: every ring slot is free now
        if (cont) {
            asrc[0] += tile_stride;
            asrc[1] += tile_stride;
        } else if (next0 < r_end) {
            set_rows(next0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ISSUE_FILL(0)
            if (1 < np) ISSUE_FILL(1)
            if (2 < np) ISSUE_FILL(2)
        }

        STAMP(2)
        const long long left = r_end - row0;
        if constexpr (ARGMAX) {
            // ---- epilogue, k == 1: running maximum per query column, in registers ----
            const int rows_valid = left >= TM ? TM : (int)left;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int rr = wm * 128 + m * 16 + fq * 4 + i;
                        const float sc = acc[m][n][i];
                        // a lane meets its rows in ascending order: strict > keeps the lowest id among equal scores
                        const bool take = sc > best_s[n] && rr < rows_valid;
                        best_s[n] = take ? sc : best_s[n];
                        best_r[n] = take ? (uint32_t)(row0 + rr) : best_r[n];
                    }
            continue;
        }
        // ---- epilogue: threshold filter, survivors -> candidate buffers ----
        // (first tile of the slice, k <= 16: the tile seeds its own threshold - see filter_tile)
        if (left >= TM) filter_tile<8, 4, true>(acc, wm * 128, wn * 64, frow, fq, row0, TM, cand, thr_l, cnt_l, flag_l, nullptr,
                                                tiles_done == 0 && P.k <= 16 && P.seed_first);
        else filter_tile<8, 4, false>(acc, wm * 128, wn * 64, frow, fq, row0, (int)left, cand, thr_l, cnt_l, flag_l);
        STAMP(3)
        __syncthreads();
        STAMP(4)
        // fold the buffers after row tiles 1, 4, 16, 64, ... of the slice and whenever one is half full
        ++tiles_done;
#ifdef SSMI_STAMP_NOPASS
        if (false) {
#else
        const bool fold_due = ((tiles_done & (tiles_done - 1)) == 0 && (tiles_done & fold_mask)) || *flag_l;
        if (xstate && !fold_due) {  // (the K loop's barriers publish the new thresholds before the next filter)
            if (xstate == 1) push_bounds(); else pull_thr();
            xstate = xstate == 1 ? 2 : 0;
        }
        if (fold_due) {
#endif
            fold_queries<R>(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
            STAMP(5)
            // Share the thresholds between the slices of this query tile: the k-th score of ANY slice is a lower
            // bound of the final k-th score, so every slice may filter with the highest one reached so far (its
            // own list then holds fewer than k keys; the merge takes that).  One atomic max per query and fold,
            // the wave's 32 queries in one round trip.
            // The tighter bound from the slices' published tops follows one and two row tiles later (push_bounds,
            // pull_thr above).
            xstate = P.gtop ? 1 : 0;
            if (P.gthr && lane < 32) {
                const int ql = wave + 8 * lane;
                if (q0 + ql < P.nq) {
                    const float t = thr_l[ql];
                    const unsigned mine = (t > SSMI_NEG_MAX) ? f32_to_ord(t) : 0u;
                    const unsigned seen = atomicMax(&P.gthr[q0 + ql], mine);
                    if (seen > mine) thr_l[ql] = ord_to_f32(seen);
                }
            }
            STAMP(6)
            __syncthreads();
            if (tid == 0) *flag_l = 0;
        }
        STAMP(7)
    }
#ifdef SSMI_STAMP
    if (lane == 0 && P.dbg)
        for (int i = 0; i < 8; ++i) P.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stv[i];
#endif

    // ---- end of slice: fold what is left; the k-list is the partial result ----
    __syncthreads();
    if constexpr (ARGMAX) {
        skey_t* red = reinterpret_cast<skey_t*>(smem);  // the ring is idle: every wave is past its last fragment read
        skey_t key[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            key[n] = best_r[n] == 0xFFFFFFFFu ? 0 : make_key(best_s[n], best_r[n]);
            key[n] = kmax(key[n], xor_lane64<16>(key[n]));   // the four lanes (fq) that hold rows of this column
            key[n] = kmax(key[n], xor_lane64<32>(key[n]));
            if (wm == 1 && fq == 0) red[wn * 64 + n * 16 + frow] = key[n];
        }
        __syncthreads();
        if (wm == 0 && fq == 0) {
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int ql = wn * 64 + n * 16 + frow;
                if (q0 + ql < P.nq) P.partial[((size_t)(q0 + ql) * P.nslices + slice) * P.k] = kmax(key[n], red[ql]);
            }
        }
        return;
    }
    fold_queries<R>(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
}

hipError_t launch_flat_mfma_pp(const MfmaParams& P_in, int r, int grid, size_t lds, hipStream_t s) {
    hipError_t e;
    MfmaParams P = P_in;
#ifdef SSMI_STAMP
    static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc((void**)&dbg, 4096 * 8 * 8 * 8);
    P.dbg = dbg;
#endif
#define LAUNCH_R(RR)                                                                                    \
    case RR: {                                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_pp_kernel<RR>),                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                  \
        hipLaunchKernelGGL((flat_mfma_pp_kernel<RR>), dim3(grid), dim3(512), lds, s, P);               \
        break;                                                                                          \
    }
    if (P.k == 1) {  // nearest-centroid assignment and any other top-1 batch
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_pp_kernel<1, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((flat_mfma_pp_kernel<1, true>), dim3(grid), dim3(512), lds, s, P);
    } else
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8) LAUNCH_R(16)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
#ifdef SSMI_STAMP
    {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> h((size_t)grid * 64);
        (void)hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost);
        const char* names[8] = {"wait", "loop", "prefetch", "filter", "sync", "fold", "exchange", "fold_sync"};
        {   // how evenly do the workgroups finish?  (sum of all stamped segments of wave 0 per workgroup)
            double mn = 1e30, mx = 0, sum = 0;
            for (int b = 0; b < grid; ++b) {
                double t = 0;
                for (int i = 0; i < 8; ++i) t += (double)h[((size_t)b * 8) * 8 + i];
                mn = std::min(mn, t); mx = std::max(mx, t); sum += t;
            }
            fprintf(stderr, "[stamp] workgroup totals (cycles): min %.0f mean %.0f max %.0f\n", mn, sum / grid, mx);
        }
        for (int w : {0, 4}) {
            fprintf(stderr, "[stamp] wave %d:", w);
            for (int i = 0; i < 8; ++i) {
                double sum = 0;
                for (int b = 0; b < grid; ++b) sum += (double)h[((size_t)b * 8 + w) * 8 + i];
                fprintf(stderr, " %s %.0f", names[i], sum / grid);
            }
            fprintf(stderr, "\n");
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace ssmi
