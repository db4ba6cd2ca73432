// mfma_pp.hip - K2, ping-pong variant: the same contraction and fused top-k
// epilogue as mfma.hip, re-scheduled so that the matrix pipe of every SIMD is
// fed continuously.
//
// Workgroup = 8 waves = two groups of four (group g = waves 4g..4g+3, one wave
// of each group per SIMD).  The K dimension is walked in PHASES of 32 (one LOAD and
// one MFMA segment each); the operands arrive in DOUBLE slabs of 64 (256 rows x 128 B
// = 32 KiB), so that every LDS-DMA instruction reads 8 rows x 128 B = 8 whole cache
// lines instead of 16 half lines that the next phase fetched again.  The row slabs
// (X) cover phases (2D, 2D+1) and are issued in even phases, the query slabs (Q) cover
// phases (2E+1, 2E+2) and are issued in odd phases: every LOAD segment issues four
// pieces per wave, as before.  (All eight of a pair in the even phase measured 1 %
// SLOWER than half-line pieces: a piece costs 100+ cycles of issue in a LOAD segment,
// and a segment of eight outlasts the partner's MFMAs.)  Query rows are stored with 64
// bytes in front and the first 64 bytes repeated behind (cvt_queries_f16_kernel), so
// the odd-phase pieces are whole lines too and the piece behind the last phase holds
// phase 0 of the next row tile.  Ring: two X slabs + two Q slabs = 128 KiB LDS.
//
//   group body, phase p:   issue the 4 pieces of X slab p/2+1 (p even) or Q slab (p+1)/2 (p odd)
//                          ds_read fragments of phase p                       (LOAD segment)
//                          group 1: s_waitcnt vmcnt(4) -> own pieces of the PREVIOUS phase landed
//                          s_barrier
//                          s_waitcnt lgkmcnt(0), 32 MFMAs                     (MFMA segment)
//                          group 0: s_waitcnt vmcnt(4)
//                          s_barrier
//
// Group 1 runs one barrier behind group 0, so between any two barrier events one
// group is in its MFMA segment while the other reads LDS: the matrix pipe only
// idles when a LOAD segment outlasts 32 MFMAs.  Barrier events of group 0: LOAD(p)
// ends with event 2p, MFMA(p) with 2p+1; group 1 is one event later.  Hazards (all
// by construction): a slab issued in LOAD(p) is first read in LOAD(p+2) and
// overwrites one last read in LOAD(p-1).
//   RAW  group 0 enters LOAD(p+2) after event 2p+3; every wave has waited for its own
//        pieces before that event (group 0 behind the MFMAs of phase p+1, group 1 at
//        the end of its LOAD(p+1))
//   WAR  the issue follows event 2p-1 (group 0) or 2p (group 1); the ds_reads of
//        LOAD(p-1) were retired by an lgkmcnt(0) before event 2p-2 (group 0) / 2p-1
//        (group 1)
// The ring runs on across row tiles: when the next row tile is whole and lies in the
// same storage segment, the last even phase fetches X slab 0 of the next tile (its
// rows are one tile stride further on) and the last odd phase its Q slab 0; its phase
// 0 reads the query piece behind this tile's last phase.  Otherwise (segment boundary,
// last tile, phases not a multiple of four) the fill is issued after the K loop,
// before the epilogue.
// Slab image: 128-byte rows, 16-byte chunk c of row r stored at chunk c ^ ((r >> 1) & 7)
// (applied to the DMA SOURCE address and to the reads): the 16 lanes of one pass of a
// ds_read_b128 then cover all 64 banks once.
#include "mfma_common.h"
#ifdef SSMI_STAMP
#include <cstdio>
#include <vector>
#include <algorithm>
#endif

// timing-only builds (wrong results): SSMI_PP_HOT bit 0: every X piece reads the same cache-hot addresses, bit 1: every Q piece,
// bit 2 / 3: the pieces may land one / two phases late, bit 4: half of the row fragments are read (a third less LDS read traffic),
// bit 5: no operand DMA
#ifndef SSMI_PP_HOT
#define SSMI_PP_HOT 0
#endif
#if SSMI_PP_HOT & 4   // the pieces may land a phase later than they are read (how much of the time is waiting for them?)
#define PP_T_WAIT4 "s_waitcnt vmcnt(8)"
#elif SSMI_PP_HOT & 8  // ... two phases later
#define PP_T_WAIT4 "s_waitcnt vmcnt(12)"
#else
#define PP_T_WAIT4 "s_waitcnt vmcnt(4)"
#endif
#if SSMI_PP_HOT & 32   // no operand DMA at all: fragment reads, MFMAs and the epilogue alone
#define PP_T_DMA false
#else
#define PP_T_DMA true
#endif
#if SSMI_PP_HOT & 16
#define PP_T_AREADS 4
#else
#define PP_T_AREADS 8
#endif
#if SSMI_PP_HOT & 1
#define PP_T_OFFX(x) ((x) & 0u)
#else
#define PP_T_OFFX(x) (x)
#endif
#if SSMI_PP_HOT & 2
#define PP_T_OFFQ(x) ((x) & 0u)
#else
#define PP_T_OFFQ(x) (x)
#endif
#ifndef SSMI_PP_PRIO
#define SSMI_PP_PRIO 0  // 0: setprio(1) around every MFMA segment; 1: static priority for waves 4-7; 2: none
#endif

namespace ssmi {

static constexpr int SLAB = 256 * 128;         // one operand's double slab (two phases): 32 KiB
static constexpr int PHASE_LDS = 2 * SLAB;     // one X slab + one Q slab
static constexpr int RING = 2;                 // double slabs per operand in the ring: X at [0, 64 KiB), Q at [64, 128 KiB)
static constexpr int QRING = RING * SLAB;      // offset of the Q slabs

// ARGMAX (k == 1: IVF centroid assignment, K5): no threshold, candidate buffers or folds at all - every lane keeps
// the best (score, row) of each of its four query columns across the row tiles of the slice, the four lanes and two
// waves that share a column combine once at the end.  Assignment batches have no score structure to prune with
// (every centroid scores alike on unclustered rows), so the filter path spent its time on survivors there.
// BF: the operands are bf16 (v_mfma_f32_16x16x32_bf16), else fp16; nothing else differs.
template <int R, bool ARGMAX = false, bool BF = false>
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
    // An XCD's workgroups (consecutive logical numbers) form a patch of patch_a query tiles x slices: per row-tile step
    // the XCD then pulls patch_a query tiles and (its workgroups / patch_a) row tiles through its L2 instead of all query
    // tiles and two row tiles (the fabric-side traffic is what costs this kernel its clock)
    int slice = logical / P.nqt, qt = logical % P.nqt;
    if (P.patch_a > 0) {
        const int per = P.nslices * P.patch_a, rem = logical % per;
        slice = rem / P.patch_a;
        qt = (logical / per) * P.patch_a + rem % P.patch_a;
    }
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
    if (tid == 0 && blockIdx.x == 0 && P.clk) { P.clk[2] = __builtin_amdgcn_s_memtime(); P.clk[3] = __builtin_amdgcn_s_memrealtime(); }
    if (tid == 0) { flag_l[0] = 0; flag_l[1] = 0; }   // [0]: a candidate buffer is half full, [1]: meetings in a row that the others did not come to (2: this workgroup has stopped meeting them)
    __syncthreads();

    const int ld = P.X.ld;
    const int np = ld / 32;  // phases per row tile
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    // DMA pieces: a double slab = 32 wave-instructions of 1 KiB (8 rows x 128 B); wave w owns pieces 4w..4w+3 of an X slab
    // and of a Q slab.  In piece j lane i fills row (4w+j)*8 + (i>>3), physical chunk i&7, i.e. logical chunk
    // (i&7) ^ ((row>>1)&7) = (i&7) ^ (i>>4) ^ 4(j&1): one lane offset, flipped by 64 bytes in the odd pieces.
    const int piece_row0 = wave * 32 + (lane >> 3);
    const int piece_coff = ((lane & 7) ^ (lane >> 4)) * 16;
    const unsigned char* qbase = reinterpret_cast<const unsigned char*>(P.Qh) + (size_t)(q0 + piece_row0) * P.q_pitch * 2;
    const unsigned q_step = 16u * (unsigned)P.q_pitch;   // from one piece of the wave to the next

    // fragment read offsets inside a double slab, first and second phase of the pair
    const int frow = lane & 15, fq = lane >> 4;
    const int fsw = (frow >> 1) & 7;
    const int fch0 = (fq ^ fsw) * 16, fch1 = ((4 + fq) ^ fsw) * 16;
    const int a_row = (wm * 128 + frow) * 128;
    const int b_row = QRING + (wn * 64 + frow) * 128;

    const unsigned char* asrc[4];
    auto set_rows = [&](long long row0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            long long r = row0 + piece_row0 + 8 * j;
            if (r >= r_end) r = r_end - 1;
            asrc[j] = reinterpret_cast<const unsigned char*>(P.X.segs[r >> P.X.seg_shift]) +
                      (size_t)(r & seg_mask) * (size_t)ld * 2 + (piece_coff ^ ((j & 1) << 6));
        }
    };
    // the wave's four pieces of an X slab (K offset KOFF bytes into the rows, TOFF more: the next tile) / of a Q slab
    // (OFF bytes into the stored query rows) into ring slot SL
#define ISSUE_X(SL, KOFF, TOFF)                                                              \
    if (PP_T_DMA) {                                                                          \
        unsigned char* dst_ = smem + (SL) * SLAB + (wave * 4) * 1024;                        \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) GLDS16(asrc[j_] + PP_T_OFFX((TOFF) + (KOFF)), dst_ + j_ * 1024);          \
    }
#define ISSUE_Q(SL, OFF)                                                                     \
    if (PP_T_DMA) {                                                                          \
        unsigned char* dst_ = smem + QRING + (SL) * SLAB + (wave * 4) * 1024;                \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                \
            GLDS16(qbase + j_ * q_step + (piece_coff ^ ((j_ & 1) << 6)) + PP_T_OFFQ(OFF), dst_ + j_ * 1024);                       \
    }
    // pipeline fill of a row tile: X slab 0, the query piece in front of phase 0 (Q slot 0) and Q slab 0 (Q slot 1)
#define ISSUE_FILL()                                                                         \
    {                                                                                        \
        ISSUE_X(0, 0u, 0u)                                                                   \
        ISSUE_Q(0, 0u)                                                                       \
        ISSUE_Q(1, 128u)                                                                     \
    }

    if (r_begin < r_end) {
        set_rows(r_begin);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ISSUE_FILL()
    }

#ifdef SSMI_STAMP
    unsigned long long stv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stv[i] += now_ - stl_; stl_ = now_; }
    unsigned long long stl_ = __builtin_amdgcn_s_memtime();
#else
#define STAMP(i)
#endif
    unsigned tiles_done = 0;
    // The workgroups of an XCD (blockIdx % 8: consecutive workgroups go round the XCDs) meet behind every row tile.  They
    // share their operand slabs through the XCD's L2 - a row slab is read by all workgroups of its slice, a query slab by
    // all workgroups of its query tile - but only while they are within a few microseconds of each other (the L2 turns over
    // in about eight), and running free they drift apart by whole folds.  One arrival counter per XCD: a workgroup adds 1
    // behind each of its tiles and waits until the count reaches (tiles done) x (workgroups of the XCD); a workgroup whose
    // slice is shorter pays the missing arrivals when it leaves.  The wait is BOUNDED: if the others do not arrive twice in a
    // row (they are not all resident: another kernel holds CUs), the workgroup stops meeting for the rest of the launch, and
    // so will the others in their turn - no workgroup ever depends on another for progress.
    const int XSYNC_POLLS = P.xsync_polls;   // bound of one wait
    bool meet = P.xsync != nullptr;
    unsigned* const xctr = P.xsync ? P.xsync + (blockIdx.x & 7) * XSYNC_STRIDE : nullptr;
    const unsigned xmembers = (unsigned)(nblocks >> 3);
    const unsigned tiles_max = (unsigned)(P.rows_per_slice / TM);
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
    const unsigned tile_stride = (unsigned)TM * (unsigned)ld * 2u;   // 512 ld bytes: fits 32 bits for any row length in reach
    for (long long row0 = r_begin; row0 < r_end; row0 += TM) {
        // The ring runs on ACROSS row tiles when the next tile is whole and lies in the same storage segment
        // (its rows are then tile_stride bytes further on, no segment lookup) and the tile's phases fill whole
        // turns of the ring: the last three phases fetch the next tile's slabs 0..2 instead of nothing, so the
        // fill's HBM round trip hides behind MFMAs and the epilogue instead of following it.
        const long long next0 = row0 + TM;
        const bool cont = (np & 3) == 0 && np >= 4 && next0 + TM <= r_end &&
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
            const int h = p & 1;
            const unsigned char* slab_x = smem + ((p >> 1) & 1) * SLAB;           // X slab p/2, half h
            const unsigned char* slab_q = smem + (((p + 1) >> 1) & 1) * SLAB;     // Q slab (p-1)/2 (slot (p+1)/2 & 1), half 1-h
            // ---- LOAD segment (the partner group owns the matrix pipe meanwhile) ----
            // four pieces into the ring slot whose last ds_reads (previous phase) every wave retired by an lgkmcnt(0)
            // BEFORE a barrier event this wave has already passed
            const bool more = p + 2 < np;
            const bool issue = more || cont;
            if (issue) {
                if (h == 0) {
                    const int d1 = (p >> 1) + 1;
                    ISSUE_X(more ? (d1 & 1) : 0, more ? (unsigned)d1 * 128u : 0u, more ? 0u : tile_stride)
                } else {
                    const int e1 = (p + 1) >> 1;   // Q slab e1 = phases p+2, p+3; behind the last phase: slab 0 of the next tile
                    ISSUE_Q(more ? ((e1 + 1) & 1) : 1, more ? (unsigned)(e1 + 1) * 128u : 128u)
                }
            }
            half8_t a[8], b[4];
            const int ao = a_row + (h ? fch1 : fch0), bo = b_row + (h ? fch0 : fch1);
#pragma unroll
            for (int m = 0; m < PP_T_AREADS; ++m) a[m] = *reinterpret_cast<const half8_t*>(slab_x + ao + m * 16 * 128);
#if PP_T_AREADS < 8   // timing-only: a third less fragment traffic (what a 128 x 128 wave tile would read per flop)
#pragma unroll
            for (int m = PP_T_AREADS; m < 8; ++m) a[m] = a[m - PP_T_AREADS];
#endif
#pragma unroll
            for (int n = 0; n < 4; ++n) b[n] = *reinterpret_cast<const half8_t*>(slab_q + bo + n * 16 * 128);
            // the pieces of the PREVIOUS phase are read from the next phase on.  Every wave's own must have landed before
            // the barrier event that lets group 0 into that phase: group 1 waits here, group 0 behind its MFMAs (one
            // event later in its own sequence, the same event of the workgroup); this phase's four may stay in flight
            if (wm == 1) { if (issue) asm volatile(PP_T_WAIT4 ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments in registers
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
                    acc[m][n] = mfma16<BF>(a[m], b[n], acc[m][n]);
#if SSMI_PP_PRIO == 0
            __builtin_amdgcn_s_setprio(0);
#endif
            __builtin_amdgcn_sched_barrier(0);
            if (wm == 0) { if (issue) asm volatile(PP_T_WAIT4 ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
            __builtin_amdgcn_s_barrier();
        }
#if SSMI_PP_PRIO == 1
        __builtin_amdgcn_s_setprio(0);
#endif
        if (wm == 0) __builtin_amdgcn_s_barrier();  // realign the two groups
        STAMP(1)

        // start the next row tile's pipeline before the epilogue: every ring slot is free now
        if (cont) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asrc[j] += tile_stride;
        } else if (next0 < r_end) {
            set_rows(next0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            ISSUE_FILL()
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
            if (R == 1 && P.fold16) fold_queries16(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
            else fold_queries<R>(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
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
        if (xctr) {
            if (tid == 0) {
                __hip_atomic_fetch_add(xctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (meet && tiles_done >= (unsigned)P.xsync_from) {
                    const unsigned target = tiles_done * xmembers;
                    int polls = 0;
                    while (__hip_atomic_load(xctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++polls < XSYNC_POLLS) __builtin_amdgcn_s_sleep(8);
                    // two waits in a row that ran out: the others are not coming
                    lds_store(flag_l + 1, polls >= XSYNC_POLLS ? flag_l[1] + 1u : 0u);
                }
            }
            if (meet && tiles_done >= (unsigned)P.xsync_from) {
                __syncthreads();
                meet = flag_l[1] < 2u;
            }
        }
    }
    // a shorter slice pays the arrivals the others still wait for
    if (xctr && tid == 0 && tiles_done < tiles_max) __hip_atomic_fetch_add(xctr, tiles_max - tiles_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef SSMI_STAMP
    if (lane == 0 && P.dbg)
        for (int i = 0; i < 8; ++i) P.dbg[((size_t)blockIdx.x * 8 + wave) * 8 + i] = stv[i];
#endif

    // ---- end of slice: fold what is left; the k-list is the partial result ----
    __syncthreads();
    if (tid == 0 && blockIdx.x == 0 && P.clk) { P.clk[0] = __builtin_amdgcn_s_memtime(); P.clk[1] = __builtin_amdgcn_s_memrealtime(); }
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
    if (R == 1 && P.fold16) fold_queries16(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
    else fold_queries<R>(wave, q0, P, slice, cand, thr_l, cnt_l, kept_l);
}

hipError_t launch_flat_mfma_pp(const MfmaParams& P_in, int r, int grid, size_t lds, hipStream_t s) {
    hipError_t e;
    MfmaParams P = P_in;
#ifdef SSMI_STAMP
    static unsigned long long* dbg = nullptr;
    if (!dbg) (void)hipMalloc((void**)&dbg, 4096 * 8 * 8 * 8);
    P.dbg = dbg;
#endif
#define LAUNCH_K(...)                                                                                   \
    {                                                                                                   \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_pp_kernel<__VA_ARGS__>),       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                  \
        hipLaunchKernelGGL((flat_mfma_pp_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, s, P);      \
    }
#define LAUNCH_R(RR) case RR: if (P.bf16) LAUNCH_K(RR, false, true) else LAUNCH_K(RR, false, false) break;
    if (P.k == 1) {  // nearest-centroid assignment and any other top-1 batch
        if (P.bf16) LAUNCH_K(1, true, true) else LAUNCH_K(1, true, false)
    } else
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4) LAUNCH_R(8) LAUNCH_R(16)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
#undef LAUNCH_K
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
