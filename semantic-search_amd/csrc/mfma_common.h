// mfma_common.h - shared pieces of the two large-batch MFMA kernels (mfma.hip, mfma_pp.hip).
#pragma once
#include "kernels.h"
#include "topk.h"

namespace ssmi {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

static constexpr int TM = 256, TN = 256, BK = 64;
static constexpr int TILE_BYTES = TM * BK * 2;  // one operand tile: 32 KiB
static constexpr int CAND_CAP = 512, CAND_TRIG = 256;

struct MfmaParams {
    RowView X;                 // fp16 rows, ld % 64 == 0
    const _Float16* Qh;        // [nqt*256][ld] fp16, zero rows beyond nq
    int nq, k, nqt, nslices;
    long long rows_per_slice;  // multiple of 256
    skey_t* partial;           // [nq][nslices][k]
    skey_t* cand;              // [gridDim.x][256][CAND_CAP]
};

#define GLDS16(gptr, ldsptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

template <int R>
__device__ __forceinline__ void compact_query(int ql, int q, const MfmaParams& P, int slice, skey_t* cand_q,
                                              float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    const int lane = lane_id();
    skey_t* kept = P.partial + ((size_t)q * P.nslices + slice) * P.k;
    WaveTopK<R> t;
    t.init();
    if (kept_l[ql]) t.merge_from(kept, P.k);
    const int n = (int)min(cnt_l[ql], (unsigned)CAND_CAP);
    // candidates arrive unsorted: sort them 256 at a time in registers (bitonic network), then one
    // bitonic merge with the kept list - far cheaper than one sorted insertion per candidate
    for (int base = 0; base < n; base += 256) {
        WaveTopK<4> c;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = base + r * 64 + lane;
            c.v[r] = (e < n) ? cand_q[e] : 0;
        }
        c.sort_desc();
        t.merge_regs(c);
    }
    const skey_t thrk = t.kth(P.k);
    t.store(kept, P.k);
    if (lane == 0) {
        kept_l[ql] = 1;
        cnt_l[ql] = 0;
        thr_l[ql] = (thrk == 0) ? SSMI_NEG_MAX : key_score(thrk);
    }
}

// ping-pong variant (mfma_pp.hip)
hipError_t launch_flat_mfma_pp(const MfmaParams& P, int r, int grid, size_t lds, hipStream_t s);

}  // namespace ssmi
