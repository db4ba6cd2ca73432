// mfma_r128.hip - K2r: Flat search of a medium query batch (5..128 queries, fp16 rows whose
// length is a multiple of 512) as an HBM-bound MFMA stream with the DATABASE ROWS STREAMED STRAIGHT
// INTO REGISTERS.
//
// mfma_n128.hip stages the rows through a ring of LDS slots, which caps the bytes in flight per
// CU at 64-80 KiB and the bandwidth at 4.4 TB/s.  Here only the query slabs live in LDS:
//
//   workgroup = 8 waves, row tile = 256 rows, wave w owns rows 32w .. 32w+31 (two 16-row MFMA
//   blocks) and ALL 128 queries: 2 x 8 accumulators of 16x16 (v_mfma_f32_16x16x32_f16);
//   K is walked in steps of 64.  The A fragments of a step (lane (r, q): 16 B of row r at k + 8q) are
//   global_load_dwordx4 in exactly the MFMA operand layout, issued PD steps ahead into a register
//   ring (PD x 16 VGPRs): with PD = 8 that is 32 KiB in flight per wave, 256 KiB per CU, and
//   no barrier or LDS traffic for the rows at all; the stream runs across row-tile boundaries.
//   The B fragments come from a ring of PD+1 LDS slots of 16 KiB (128 queries x 64 k, stored as two
//   swizzled 64-byte-row slabs like mfma_pp.hip), filled by LDS-DMA PD steps ahead as well (the
//   vmcnt counter is in order, so both streams must run equally far ahead); one barrier per step
//   publishes a query slab.
//
// Threshold filter and list compaction are those of mfma_pp.hip (mfma_common.h); the candidate stores
// are drained (vmcnt(0)) only in front of a compaction, so the row stream keeps flowing through the
// end of a row tile.
#include "mfma_common.h"

namespace ssmi {

static constexpr int RQ_SLOT = 16 * 1024;  // one query slab: 2 k-halves x (128 queries x 64 B)

// NT: 16-query blocks actually scored (1, 4 or 8): a batch of <= 16 queries costs 4 MFMAs and 2 fragment reads
// per step and wave instead of 32 and 16, and leaves the registers for a ring of 8 steps.
template <int R, int PD, int NT, bool BF = false>
__global__ void __launch_bounds__(512) flat_mfma_r128_kernel(MfmaParams P) {
    constexpr int QD = PD + 1;          // query ring slots
    extern __shared__ __align__(16) unsigned char smem[];
    float* thr_l = reinterpret_cast<float*>(smem + QD * RQ_SLOT);
    unsigned* cnt_l = reinterpret_cast<unsigned*>(thr_l + N128_TN);
    unsigned* kept_l = cnt_l + N128_TN;
    unsigned* flag_l = kept_l + N128_TN;
    unsigned* seed_l = flag_l + 4;  // [N128_TN] first-tile threshold seeds (f32_to_ord bits, minimum over the waves)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int slice = blockIdx.x;
    const long long r_begin = (long long)slice * P.rows_per_slice;
    const long long r_end = (r_begin + P.rows_per_slice < P.X.nrows) ? (r_begin + P.rows_per_slice) : P.X.nrows;
    skey_t* cand = P.cand + (size_t)blockIdx.x * N128_TN * CAND_CAP;

    if (tid < N128_TN) {
        thr_l[tid] = (tid < P.nq) ? P.thr_floor : __builtin_inff();
        cnt_l[tid] = 0;
        kept_l[tid] = 0;
        seed_l[tid] = 0xFFFFFFFFu;
    }
    if (tid == 0) *flag_l = 0;
    __syncthreads();
    if (r_begin >= r_end) return;

    const int ld = P.X.ld;
    const int ns = ld / 64;  // steps per row tile (a multiple of PD: the host checks ld % 512 == 0)
    const int ntiles = (int)((r_end - r_begin + TM - 1) / TM);
    const int total = ntiles * ns;
    const long long seg_mask = (1ll << P.X.seg_shift) - 1;

    const int frow = lane & 15, fq = lane >> 4;
    const int fch = (fq ^ ((-(frow >> 2)) & 3)) * 16;
    const int b_off = frow * 64 + fch;  // + kh * 8192 + n * 1024

    // query slab DMA, shared evenly by the 8 waves (QSlabDma): every wave issues 4 row loads + QOPS DMAs per
    // step, so one counted wait serves all of them.  Steps past the end of the slice repeat the last one (into
    // ring positions nobody reads any more).
    using QD_t = QSlabDma<NT>;
    constexpr int VOPS = 4 + QD_t::OPS;  // vector-memory instructions per wave and step
    const unsigned char* qsrc = reinterpret_cast<const unsigned char*>(P.Qh) + (size_t)QD_t::qrow(wave, lane) * ld * 2 +
                                QD_t::src_off(wave, lane);
    auto issue_q = [&](int g /*global step*/) {
        const int s = (g < total ? g : total - 1) % ns;
        QD_t::issue(qsrc + (size_t)s * 128, smem + (g % QD) * RQ_SLOT + QD_t::dst_off(wave));
    };
    // row base of this lane's two 16-row blocks in row tile `t` (padding rows repeat the last valid row)
    auto tile_rows = [&](int t, const unsigned char* (&base)[2]) {
        const long long t0 = r_begin + (long long)t * TM;
        const unsigned char* segbase = reinterpret_cast<const unsigned char*>(P.X.segs[t0 >> P.X.seg_shift]);
        const long long left = r_end - t0;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            long long rr = wave * 32 + m * 16 + frow;
            if (rr >= left) rr = left - 1;
            base[m] = segbase + (size_t)((t0 & seg_mask) + rr) * (size_t)ld * 2 + fq * 16;
        }
    };

    half8_t areg[PD][2][2];  // [ring slot][16-row block][k-half]
    const unsigned char* abase[2];
    // The row loads are issued as opaque instructions: the compiler's wait-count pass cannot count loads
    // across this loop nest and would put s_waitcnt vmcnt(0) in front of every use, which turns the register
    // ring into a blocking load.  Completion is enforced by the counted waits below, which take the ring
    // slot's registers as in/out operands so that no use can be scheduled above them.
    // The ring registers are written by the memory system long after the load instruction: between a load
    // and its counted wait the compiler must neither copy nor reuse them.  Each ring slot is therefore defined
    // by exactly ONE load statement and ONE wait statement per step (alternative definitions would meet in a
    // phi and be copied while in flight), and the ring is drained with the registers still named before the
    // kernel's tail may take them.  tools/check_ring_regs.py verifies this on the generated assembly.
    auto load_a = [&](half8_t (&dst)[2][2], const unsigned char* const (&base)[2], int s) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const unsigned char* ptr = base[m] + (size_t)s * 128 + kh * 64;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[m][kh]) : "v"(ptr) : "memory");
            }
    };

    // pipeline fill: steps 0 .. PD-1 (all in the first row tile: ns >= PD)
    tile_rows(0, abase);
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
    // cross-slice bound from the slices' published tops, as in mfma_pp.hip: one row tile after a fold this slice
    // computes the bound of its share of the queries and pushes it into gthr, another tile later every wave pulls
    // scheduled folds after row tiles 1, 4, 16, 64, ...; with the cross-slice bound (which matures much faster than a
    // slice's own k-th score) after 1, 16, 256, ... only
    const unsigned fold_mask = P.gtop ? 0x11111111u : 0x55555555u;
    int xstate = 0;
    auto push_bounds = [&]() {
        for (int ql = slice + wave * P.nslices; ql < P.nq; ql += 8 * P.nslices) {
            const unsigned b = tops_bound(P, ql);
            if (b && lane == 0) atomicMax(&P.gthr[ql], b);
        }
    };
    auto pull_thr = [&]() {
        if (lane < 16) {
            const int ql = wave + 8 * lane;
            if (ql < P.nq) {
                const unsigned seen = __hip_atomic_load(&P.gthr[ql], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (seen && ord_to_f32(seen) > thr_l[ql]) thr_l[ql] = ord_to_f32(seen);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    int g = 0;
    for (int t = 0; t < ntiles; ++t) {
        const long long row0 = r_begin + (long long)t * TM;
        const unsigned char* nbase[2] = {abase[0], abase[1]};
        if (t + 1 < ntiles) tile_rows(t + 1, nbase);
        for (int s0 = 0; s0 < ns; s0 += PD) {
            const bool refill_next_tile = (s0 + PD >= ns);
#pragma unroll
            for (int u = 0; u < PD; ++u, ++g) {
                // this wave's loads and DMA of step g have landed: everything younger (PD-1 steps of VOPS) may fly
#define RING_WAIT(N)                                                                                             \
    asm volatile("s_waitcnt vmcnt(%4)"                                                                           \
                 : "+v"(areg[u][0][0]), "+v"(areg[u][0][1]), "+v"(areg[u][1][0]), "+v"(areg[u][1][1])           \
                 : "n"(N)                                                                                        \
                 : "memory")
                RING_WAIT((PD - 1) * VOPS);
                __builtin_amdgcn_s_barrier();  // query slab g visible; every wave is done with slab g-1
                issue_q(g + PD);               // into the slot of slab g-1
                const unsigned char* slot = smem + (g % QD) * RQ_SLOT + b_off;
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
                {   // refill the ring slot with step g + PD (the next row tile's rows when this one is used up;
                    // past the end of the slice: this step again, never consumed)
                    const bool past = g + PD >= total;
                    const unsigned char* rb[2];
                    rb[0] = (refill_next_tile && !past) ? nbase[0] : abase[0];
                    rb[1] = (refill_next_tile && !past) ? nbase[1] : abase[1];
                    const int rs = past ? s0 + u : (refill_next_tile ? s0 + PD - ns + u : s0 + PD + u);
                    load_a(areg[u], rb, rs);
                }
            }
        }
        abase[0] = nbase[0];
        abase[1] = nbase[1];

        // ---- row tile complete: threshold filter; fold the candidate buffers when due ----
        const long long left = r_end - row0;
        if (t == 0 && left >= TM && P.k <= 16 && P.seed_first) {
            // The first row tile seeds the thresholds (k <= 16, whole tile): a slice starts without a list, so all 256
            // rows would pass the filter and the first fold would sort 256 candidates per query.  Every wave takes the
            // SECOND largest score of its 32 rows per query; the minimum of those over the 8 waves has at least 16 rows
            // of the tile at or above it, i.e. it is a lower bound of the final k-th score and a legitimate threshold.
#pragma unroll
            for (int n = 0; n < NT; ++n) {
                float a1 = -__builtin_inff(), a2 = a1;
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float v = acc[m][n][i];
                        const float lo = fminf(v, a1);
                        a1 = fmaxf(v, a1);
                        a2 = fmaxf(a2, lo);
                    }
                {   // the four lanes (fq) that hold the other rows of this query column
                    float b1 = __int_as_float(xor_lane<16>(__float_as_int(a1))), b2 = __int_as_float(xor_lane<16>(__float_as_int(a2)));
                    a2 = fmaxf(fminf(a1, b1), fmaxf(a2, b2)); a1 = fmaxf(a1, b1);
                    b1 = __int_as_float(xor_lane<32>(__float_as_int(a1))); b2 = __int_as_float(xor_lane<32>(__float_as_int(a2)));
                    a2 = fmaxf(fminf(a1, b1), fmaxf(a2, b2)); a1 = fmaxf(a1, b1);
                }
                if (fq == 0) lds_min_u32(&seed_l[n * 16 + frow], f32_to_ord(a2));
            }
            __builtin_amdgcn_s_barrier();
            if (tid < N128_TN && tid < P.nq) {
                const float sd = ord_to_f32(seed_l[tid]);
                if (sd > thr_l[tid]) lds_store(reinterpret_cast<unsigned*>(&thr_l[tid]), __float_as_uint(sd));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (left >= TM) filter_tile<2, NT, true>(acc, wave * 32, 0, frow, fq, row0, TM, cand, thr_l, cnt_l, flag_l);
        else filter_tile<2, NT, false>(acc, wave * 32, 0, frow, fq, row0, (int)left, cand, thr_l, cnt_l, flag_l);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc[m][n] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // every wave's counters and flag are in LDS
        ++tiles_done;
        const bool fold_due = ((tiles_done & (tiles_done - 1)) == 0 && (tiles_done & fold_mask)) || *flag_l;
        if (xstate && !fold_due) {  // (the per-step barriers publish the new thresholds before the next filter)
            if (xstate == 1) push_bounds(); else pull_thr();
            xstate = xstate == 1 ? 2 : 0;
        }
        if (fold_due) {
            __syncthreads();  // drains the candidate stores (and the row stream, once per compaction)
            if (R == 1 && P.fold16) fold_queries16(wave, 0, P, slice, cand, thr_l, cnt_l, kept_l);   // k <= 16: four queries per wave pass
            else
                for (int ql = wave; ql < N128_TN; ql += 8)
                    if (ql < P.nq) compact_query<R>(ql, ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
            // Share the thresholds between the slices (one per CU, a few thousand rows each: a slice's own k-th score
            // stays weak to the end): the k-th score of ANY slice bounds the final k-th score from below, so every
            // slice filters with the highest one reached anywhere.  Same exchange as the 256-query kernel: one
            // atomic max per query and fold, the wave's 16 queries in one round trip.
            xstate = P.gtop ? 1 : 0;
            if (P.gthr && lane < 16) {
                const int ql = wave + 8 * lane;
                if (ql < P.nq) {
                    const float t = thr_l[ql];
                    const unsigned mine = (t > SSMI_NEG_MAX) ? f32_to_ord(t) : 0u;
                    const unsigned seen = atomicMax(&P.gthr[ql], mine);
                    if (seen > mine) thr_l[ql] = ord_to_f32(seen);
                }
            }
            __syncthreads();
            if (tid == 0) *flag_l = 0;
            // the loads of the register ring were drained with everything else, which the counted waits
            // of the next steps tolerate (they only ever wait for MORE than is outstanding)
        }
    }

    // drain the ring (the repeated steps of the tail) while its registers are still reserved
#pragma unroll
    for (int u = 0; u < PD; ++u) RING_WAIT(0);
#undef RING_WAIT

    // ---- end of slice: fold what is left; the k-list is the partial result ----
    __syncthreads();
    if (R == 1 && P.fold16) fold_queries16(wave, 0, P, slice, cand, thr_l, cnt_l, kept_l);
    else
        for (int ql = wave; ql < N128_TN; ql += 8)
            if (ql < P.nq) compact_query<R>(ql, ql, P, slice, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l);
}

template <int PD, int NT>
static hipError_t launch_pd(const MfmaParams& P, int r, int grid, hipStream_t s) {
    const size_t lds = (size_t)(PD + 1) * RQ_SLOT + 4 * N128_TN * 4 + 16;
    hipError_t e;
#define LAUNCH_K(...)                                                                                   \
    {                                                                                                   \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(flat_mfma_r128_kernel<__VA_ARGS__>),     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
        if (e != hipSuccess) return e;                                                                  \
        hipLaunchKernelGGL((flat_mfma_r128_kernel<__VA_ARGS__>), dim3(grid), dim3(512), lds, s, P);    \
    }
#define LAUNCH_R(RR) case RR: if (P.bf16) LAUNCH_K(RR, PD, NT, true) else LAUNCH_K(RR, PD, NT, false) break;
    switch (r) {
        LAUNCH_R(1) LAUNCH_R(2) LAUNCH_R(4)
        default: return hipErrorInvalidValue;
    }
#undef LAUNCH_R
#undef LAUNCH_K
    return hipGetLastError();
}

// Ring depth by query blocks: 1 or 4 blocks leave room for a ring of 8 steps (256 KiB of rows in flight per CU);
// 8 blocks (128 queries: 64 accumulator + 32 fragment registers) for 4.  Every instantiation must compile to
// `ScratchSize: 0`: a spill of a ring register whose load is still in flight would store garbage.
bool mfma_r128_supported(int ld, int k) { return (ld % 512) == 0 && k <= 256; }
hipError_t launch_flat_mfma_r128(const MfmaParams& P, int r, int grid, hipStream_t s) {
    if (P.nq <= 16) return launch_pd<8, 1>(P, r, grid, s);
    if (P.nq <= 64) return launch_pd<8, 4>(P, r, grid, s);
    return launch_pd<4, 8>(P, r, grid, s);
}

}  // namespace ssmi
