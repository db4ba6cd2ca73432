// scan_mm.h - K1m: the streaming scan of scan_impl.h with its multiply-adds on the matrix pipe
// (v_mfma_f32_4x4x1_16b_f32: fp32 in, fp32 accumulate, bitwise an fmaf chain), for the batch sizes at which
// the register-FMA form is bound by its LDS reads (5-8 queries on fp32 rows, 3-4 on 16-bit rows: one
// ds_read_b128 per query and 16 B of row there, a quarter of that here).
//
// Same place on the path as K1: faiss.IndexFlatIP.search as called from core/index_manager.py:371 of the
// reference, with the handful of queries its multi-index and exclusion flows issue.
//
// Mapping: one wavefront owns FOUR consecutive rows at a time.  The instruction multiplies, in each of its 16
// blocks, a 4x1 column (lane 4b+i: row i) with a 1x4 row (lane 4b+j: query j); block b of lane group b takes
// the row elements at 16-byte slot b of a 256-byte chunk, so one load instruction reads 256 contiguous bytes
// of each of the four rows, and a chunk of V elements per lane feeds V instructions per query quad.  The 16
// blocks hold partial sums over disjoint dimensions: they are added across lanes once per row group.  Queries
// sit in LDS in the order the lanes read them (1 KiB contiguous per wave read).  Everything from the score
// onwards (threshold compare, register list, block merge, tail merge) is K1's.
#pragma once
#include "scan_impl.h"

namespace ssmi {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// a[lane] + a[lane ^ 32] in the lower half of the wave, the same of b in the upper half: v_permlane32_swap hands the upper
// half of a and the lower half of b to each other, so two registers are folded over one lane bit by one swap and one add
__device__ __forceinline__ float fold32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(a), __float_as_int(b), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}
// the same over lane bit 4: a in the even rows of 16 lanes, b in the odd ones
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(a), __float_as_int(b), false, false);
    return __int_as_float(r[0]) + __int_as_float(r[1]);
}

// LDS float index of element `elem` of local query `ql` (BT queries per pass, V elements per 16 bytes of row)
template <int V, int BT>
__device__ __forceinline__ int qmm_lds_index(int ql, int elem) {
    const int c = elem / (16 * V), in = elem % (16 * V);
    const int b = in / V, w = in % V;
    return ((((c * (BT / 4) + ql / 4) * (V / 4) + w / 4) * 64) + 4 * b + (ql & 3)) * 4 + (w & 3);
}

// requires X.ld % (16 * V * 2 * G) == 0 (plan_flat_scan), BT in {4, 8}.  G: 16-byte loads per lane and half stage;
// two half stages are in flight (2*G KiB per wave), the second one crossing into the next row group.
template <typename T, int BT, int R, int G>
__global__ void __launch_bounds__(G <= 8 ? 1024 : 512) flat_scan_mm_kernel(RowView X, const float* __restrict__ Q, int nq, int k,
                                                            skey_t* partial, TailMerge tail) {
    constexpr int V = ElemTraits<T>::V;
    constexpr int QQ = BT / 4, PL = V / 4;
    static_assert(BT == 4 || BT == 8, "query quads");
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int ld = X.ld;
    const int q0 = blockIdx.y * BT;
    const int nch = ld / (16 * V);

    float* qs = reinterpret_cast<float*>(smem);
    for (int b = 0; b < BT; ++b) {
        const int qi = (q0 + b) < nq ? (q0 + b) : nq - 1;
        for (int e = threadIdx.x; e < ld; e += blockDim.x) qs[qmm_lds_index<V, BT>(b, e)] = Q[(size_t)qi * ld + e];
    }
    __syncthreads();

    WaveTopK<R> top[BT];
    skey_t thr[BT];
    const skey_t thr_floor = floor_key(tail.min_ord);
#pragma unroll
    for (int b = 0; b < BT; ++b) { top[b].init(); thr[b] = thr_floor; }
    // per-lane float gate: the score below which a candidate of the lane's query (see the fold below) cannot pass thr - the
    // score field of the threshold key; key 0 decodes to a NaN, against which nothing is "below"
    float gate = ord_to_f32((uint32_t)(thr_floor >> 32));

    // row groups -> waves: the front sweep of K1 (scan_impl.h), in units of four rows
    const unsigned vb = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const long long grp_step = (long long)gridDim.x * nwaves;
    const long long ngroups = (X.nrows + 3) >> 2;
    const long long seg_mask = (1ll << X.seg_shift) - 1;
    const size_t row_bytes = (size_t)ld * sizeof(T);
    const float* qlane = qs + lane * 4;

    // lane 4b+i reads 16-byte slot b of every 256-byte chunk of row i of the group (a segment holds a multiple of
    // 256 rows, so the four rows share one; a row beyond the last is read as the last and never offered to a list)
    auto group_base = [&](long long grp) -> const unsigned char* {
        const long long row0 = grp << 2;
        long long myrow = row0 + (lane & 3);
        const int slot = lane >> 2;
        if (myrow >= X.nrows) myrow = X.nrows - 1;
        return reinterpret_cast<const unsigned char*>(X.segs[row0 >> X.seg_shift]) + (size_t)(myrow & seg_mask) * row_bytes +
               (size_t)slot * 16;
    };
    auto load_stage = [&](uint4 (&x)[G], const unsigned char* p) {
#pragma unroll
        for (int g = 0; g < G; ++g) x[g] = load16_nt_g(p + (size_t)g * 256);
    };

    long long grp = (long long)vb * nwaves + wave;
    if (grp >= ngroups) grp = -1;
    const unsigned char* base = nullptr;
    uint4 xa[G], xb[G];
    if (grp >= 0) {
        base = group_base(grp);
        load_stage(xa, base);
    }

    while (grp >= 0) {
        const long long row0 = grp << 2;
        const long long grp_next = grp + grp_step < ngroups ? grp + grp_step : -1;
        const unsigned char* base_next = grp_next < 0 ? base : group_base(grp_next);  // the last stage issued is never used
        f32x4_t acc[QQ];
#pragma unroll
        for (int qq = 0; qq < QQ; ++qq) acc[qq] = f32x4_t{0.f, 0.f, 0.f, 0.f};

        auto consume = [&](const uint4 (&x)[G], int c0) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float xf[V];
                Unpack<T>::run(x[g], xf);
#pragma unroll
                for (int p = 0; p < PL; ++p) {
#pragma unroll
                    for (int qq = 0; qq < QQ; ++qq) {
                        const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(qlane + (((c0 + g) * QQ + qq) * PL + p) * 256);
                        acc[qq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[4 * p], t4[0], acc[qq], 0, 0, 0);
                        acc[qq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[4 * p + 1], t4[1], acc[qq], 0, 0, 0);
                        acc[qq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[4 * p + 2], t4[2], acc[qq], 0, 0, 0);
                        acc[qq] = __builtin_amdgcn_mfma_f32_4x4x1f32(xf[4 * p + 3], t4[3], acc[qq], 0, 0, 0);
                    }
                }
            }
        };
        // the scheduler would otherwise sink every load to its first use (two or three in flight instead of 2*G)
#pragma unroll 1
        for (int c0 = 0; c0 < nch; c0 += 2 * G) {
            load_stage(xb, base + (size_t)(c0 + G) * 256);
            __builtin_amdgcn_sched_barrier(0);
            consume(xa, c0);
            __builtin_amdgcn_sched_barrier(0);
            const bool last = c0 + 2 * G >= nch;
            load_stage(xa, last ? base_next : base + (size_t)(c0 + 2 * G) * 256);
            __builtin_amdgcn_sched_barrier(0);
            consume(xb, c0 + G);
            __builtin_amdgcn_sched_barrier(0);
        }
        // Register i of lane 4b+j holds block b's share of <row i, query quad's j>: the 16 blocks (lane bits 2-5) are summed.
        // Bits 5 and 4 are folded together with a halving of the registers (fold32 / fold16), bits 3 and 2 with DPP adds:
        // 8 queries: w[t] of lane (q5, i4, *, *, j) = score of row t + 2*i4, query 4*q5 + j;
        // 4 queries: w[0] of lane (i5, i4, *, *, j) = score of row 2*i5 + i4, query j
        float w[QQ];
        if constexpr (QQ == 2) {
            const float u0 = fold32(acc[0][0], acc[1][0]), u1 = fold32(acc[0][1], acc[1][1]);
            const float u2 = fold32(acc[0][2], acc[1][2]), u3 = fold32(acc[0][3], acc[1][3]);
            w[0] = fold16(u0, u2);
            w[1] = fold16(u1, u3);
        } else {
            const float u0 = fold32(acc[0][0], acc[0][2]), u1 = fold32(acc[0][1], acc[0][3]);
            w[0] = fold16(u0, u1);
        }
#pragma unroll
        for (int t = 0; t < QQ; ++t) {
            w[t] += __int_as_float(xor_lane<8>(__float_as_int(w[t])));
            w[t] += __int_as_float(xor_lane<4>(__float_as_int(w[t])));
        }
        // one vector compare per score against the lanes' gates: a group without a candidate (nearly all of them once the
        // lists are warm) costs QQ compares and one branch instead of 4*BT scalar key compares
        bool cand = !(w[0] < gate);
        if constexpr (QQ == 2) cand |= !(w[1] < gate);
        if (__builtin_amdgcn_ballot_w64(cand) != 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int src = QQ == 2 ? ((b >> 2) << 5) | ((i >> 1) << 4) | (b & 3) : ((i >> 1) << 5) | ((i & 1) << 4) | (b & 3);
                    const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w[QQ == 2 ? (i & 1) : 0]), src));
                    const long long row = row0 + i;
                    const skey_t key = make_key(s, (uint32_t)row);
                    if (row < X.nrows && key > thr[b]) {
                        top[b].insert(key);
                        thr[b] = kmax(top[b].kth(k), thr_floor);
                        const float gnew = ord_to_f32((uint32_t)(thr[b] >> 32));
                        const bool mine = (lane & 3) == (b & 3) && (QQ == 1 || (lane >> 5) == (b >> 2));
                        gate = mine ? gnew : gate;
                    }
                }
            }
        }
        grp = grp_next;
        base = base_next;
    }

    // block-level merge through LDS (reuses the query area), publication and tail merge: as in K1
    __syncthreads();
    skey_t* lds_keys = reinterpret_cast<skey_t*>(smem);
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        block_merge<R>(top[b], lds_keys, wave, nwaves);
        if (wave == 0 && (q0 + b) < nq)
            publish_list<R>(top[b], partial + ((size_t)(q0 + b) * gridDim.x + blockIdx.x) * k, k);
        __syncthreads();
    }
    if (tail.counter == nullptr) return;
    if (!tail_last_block(tail.counter + blockIdx.y, gridDim.x, reinterpret_cast<unsigned*>(smem))) return;
#pragma unroll 1
    for (int b = 0; b < BT; ++b) {
        const int q = q0 + b;
        if (q >= nq) break;
        merge_partials<R, true>(partial + (size_t)q * gridDim.x * k, (int)gridDim.x, k, lds_keys, wave, nwaves,
                          tail.D + (size_t)q * k, tail.I + (size_t)q * k, tail.id_offset,
                          PlanOut{tail.plan_sizes, tail.plan_sizes ? tail.plan_prefix + (size_t)q * (k + 1) : nullptr,
                                  tail.plan_page_off, tail.plan_sizes ? tail.plan_po + (size_t)q * k : nullptr}, tail.min_ord);
        __syncthreads();
    }
}

}  // namespace ssmi
