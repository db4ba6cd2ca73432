// topk.h - per-wavefront top-k kept in registers (gfx950, wave64).
//
// A wave keeps its best 64*R candidates as 64-bit keys (common.h), sorted
// best-first and spread over the lanes: element e lives in register e/64 of
// lane e%64.  Operations:
//   insert()      one candidate, guarded by the caller's threshold compare;
//                 rare once the list is warm, so it may cost tens of cycles.
//   merge_from()  a whole sorted list (LDS or global): element-wise max against
//                 the reversed list gives a bitonic sequence holding the best
//                 64*R of the union, which a log2(64*R)-stage bitonic merge
//                 re-sorts.  Used for the block-level and cross-block merges,
//                 so the score matrix is never written anywhere.
//   sort_desc()   full bitonic sort of unsorted register contents.
//   merge_regs()  merge_from for a list that already sits in registers.
// Lane exchanges lane ^ M use DPP (quad_perm / row_half_mirror / row_ror) and the
// gfx950 v_permlane16_swap / v_permlane32_swap instead of ds_bpermute: a bitonic
// stage is a dependent chain, and a DPP move costs a few cycles where a
// ds_bpermute round trip costs ~100.
#pragma once
#include "common.h"

namespace ssmi {

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// value of lane (lane ^ M), M a power of two < 64
template <int M>
__device__ __forceinline__ int xor_lane(int v) {
    static_assert(M == 1 || M == 2 || M == 4 || M == 8 || M == 16 || M == 32, "xor_lane: M must be a power of two < 64");
    if constexpr (M == 1) {
        return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);  // quad_perm [1,0,3,2]
    } else if constexpr (M == 2) {
        return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    } else if constexpr (M == 4) {  // row_half_mirror (i -> 7-i) then quad_perm [3,2,1,0] (i -> i^3): i ^ 7 ^ 3
        return __builtin_amdgcn_mov_dpp(__builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true), 0x1B, 0xF, 0xF, true);
    } else if constexpr (M == 8) {
        return __builtin_amdgcn_mov_dpp(v, 0x128, 0xF, 0xF, true);  // row_ror:8
    } else if constexpr (M == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (lane_id() & 16) ? r[0] : r[1];
    } else {
        const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (lane_id() & 32) ? r[0] : r[1];
    }
}

template <int M>
__device__ __forceinline__ skey_t xor_lane64(skey_t v) {
    const int lo = xor_lane<M>((int)(unsigned)(v & 0xFFFFFFFFull));
    const int hi = xor_lane<M>((int)(unsigned)(v >> 32));
    return ((skey_t)(unsigned)hi << 32) | (skey_t)(unsigned)lo;
}

__device__ __forceinline__ skey_t shfl_up64(skey_t v, int d) {
    int lo = (int)(unsigned)(v & 0xFFFFFFFFull), hi = (int)(unsigned)(v >> 32);
    lo = __shfl_up(lo, d); hi = __shfl_up(hi, d);
    return ((skey_t)(unsigned)hi << 32) | (skey_t)(unsigned)lo;
}
__device__ __forceinline__ skey_t readlane64(skey_t v, int lane /*wave-uniform*/) {
    int lo = (int)(unsigned)(v & 0xFFFFFFFFull), hi = (int)(unsigned)(v >> 32);
    lo = __builtin_amdgcn_readlane(lo, lane); hi = __builtin_amdgcn_readlane(hi, lane);
    return ((skey_t)(unsigned)hi << 32) | (skey_t)(unsigned)lo;
}
__device__ __forceinline__ skey_t kmax(skey_t a, skey_t b) { return a > b ? a : b; }
__device__ __forceinline__ skey_t kmin(skey_t a, skey_t b) { return a < b ? a : b; }

// butterfly sum: every lane ends with the same bits (fp32 add is commutative)
__device__ __forceinline__ float wave_sum(float v) {
    v += __int_as_float(xor_lane<32>(__float_as_int(v)));
    v += __int_as_float(xor_lane<16>(__float_as_int(v)));
    v += __int_as_float(xor_lane<8>(__float_as_int(v)));
    v += __int_as_float(xor_lane<4>(__float_as_int(v)));
    v += __int_as_float(xor_lane<2>(__float_as_int(v)));
    v += __int_as_float(xor_lane<1>(__float_as_int(v)));
    return v;
}

template <int R>
struct WaveTopK {
    static_assert(R == 1 || R == 2 || R == 4 || R == 8 || R == 16, "R must be a power of two <= 16");
    static const int KP = 64 * R;
    skey_t v[R];

    __device__ __forceinline__ void init() {
#pragma unroll
        for (int r = 0; r < R; ++r) v[r] = 0;
    }

    // k-th best (1-based k), wave-uniform.  0 while fewer than k candidates.
    __device__ __forceinline__ skey_t kth(int k) const {
        const int e = k - 1;
        skey_t sel = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) sel = ((e >> 6) == r) ? v[r] : sel;
        return readlane64(sel, e & 63);
    }

    // Insert one wave-uniform key.  All 64 lanes must call.
    __device__ __forceinline__ void insert(skey_t key) {
        const int lane = lane_id();
        skey_t carry = ~0ull;  // virtual element -1: better than anything
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const skey_t old = v[r];
            const skey_t up = shfl_up64(old, 1);
            const skey_t prev = (lane == 0) ? carry : up;
            carry = readlane64(old, 63);
            v[r] = (old >= key) ? old : ((prev >= key) ? key : prev);
        }
    }

    // one compare-exchange stage between elements e and e ^ STRIDE (STRIDE < 64); a block of
    // 2*SIZE_BIT... elements with (e & size_bit) == 0 sorts best-first, the others worst-first
    template <int STRIDE>
    __device__ __forceinline__ void lane_stage(int size_bit /*0: always best-first*/) {
        const int lane = lane_id();
        const bool lower = (lane & STRIDE) == 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool desc = ((r * 64 + lane) & size_bit) == 0;
            const skey_t p = xor_lane64<STRIDE>(v[r]);
            v[r] = (desc == lower) ? kmax(v[r], p) : kmin(v[r], p);
        }
    }
    __device__ __forceinline__ void reg_stage(int rs, int size_bit) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if ((r & rs) == 0 && r + rs < R) {
                const bool desc = ((r * 64) & size_bit) == 0;
                const skey_t a = v[r], b = v[r + rs];
                const skey_t hi = kmax(a, b), lo = kmin(a, b);
                v[r] = desc ? hi : lo;
                v[r + rs] = desc ? lo : hi;
            }
        }
    }
    // all stages with stride <= START (a power of two) of the merge step for blocks of size 2*... `size_bit`
    template <int START>
    __device__ __forceinline__ void lane_stages(int size_bit) {
        if constexpr (START >= 32) lane_stage<32>(size_bit);
        if constexpr (START >= 16) lane_stage<16>(size_bit);
        if constexpr (START >= 8) lane_stage<8>(size_bit);
        if constexpr (START >= 4) lane_stage<4>(size_bit);
        if constexpr (START >= 2) lane_stage<2>(size_bit);
        lane_stage<1>(size_bit);
    }

    // re-sort a bitonic sequence, best first
    __device__ __forceinline__ void bitonic_merge() {
#pragma unroll
        for (int rs = R / 2; rs >= 1; rs >>= 1) reg_stage(rs, 0);
        lane_stages<32>(0);
    }

    // Full bitonic sort of arbitrary contents, best first.
    __device__ __forceinline__ void sort_desc() {
        // sizes 2..64 live inside a register (lane strides only); a final size uses size_bit 0
        lane_stages<1>(KP == 2 ? 0 : 2);
        lane_stages<2>(KP == 4 ? 0 : 4);
        lane_stages<4>(KP == 8 ? 0 : 8);
        lane_stages<8>(16);
        lane_stages<16>(32);
        lane_stages<32>(R == 1 ? 0 : 64);
#pragma unroll
        for (int size = 128; size <= KP; size <<= 1) {
            const int sb = (size == KP) ? 0 : size;
#pragma unroll
            for (int rs = size >> 7; rs >= 1; rs >>= 1) reg_stage(rs, sb);
            lane_stages<32>(sb);
        }
    }

    // Merge a best-first sorted list of `len` (<= KP) keys into this list.
    __device__ __forceinline__ void merge_from(const skey_t* src, int len) {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = KP - 1 - (r * 64 + lane);
            const skey_t b = (j < len) ? src[j] : 0;
            v[r] = kmax(v[r], b);
        }
        bitonic_merge();
    }

    // Merge another register-resident sorted list (any length) into this one.
    template <int RC>
    __device__ __forceinline__ void merge_regs(const WaveTopK<RC>& c) {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            // reversed element KP-1-(r*64+lane) of c = register R-1-r, lane 63-lane
            skey_t b = 0;
            if (R - 1 - r < RC) {
                const skey_t src = c.v[(R - 1 - r) < RC ? (R - 1 - r) : 0];
                int lo = (int)(unsigned)(src & 0xFFFFFFFFull), hi = (int)(unsigned)(src >> 32);
                lo = __shfl(lo, 63 - lane); hi = __shfl(hi, 63 - lane);
                b = ((skey_t)(unsigned)hi << 32) | (skey_t)(unsigned)lo;
            }
            v[r] = kmax(v[r], b);
        }
        bitonic_merge();
    }

    // store the first n (<= KP) elements
    __device__ __forceinline__ void store(skey_t* dst, int n) const {
        const int lane = lane_id();
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = r * 64 + lane;
            if (e < n) dst[e] = v[r];
        }
    }
};

// Block-level merge: every wave contributes its list; wave 0 ends up with the
// best KP of the block.  `lds` needs nwaves*KP keys.  All threads must call.
template <int R>
__device__ __forceinline__ void block_merge(WaveTopK<R>& t, skey_t* lds, int wave, int nwaves) {
    const int KP = WaveTopK<R>::KP;
    t.store(lds + wave * KP, KP);
    for (int s = 1; s < nwaves; s <<= 1) {
        __syncthreads();
        const bool active = ((wave & (2 * s - 1)) == 0) && (wave + s < nwaves);
        if (active) t.merge_from(lds + (wave + s) * KP, KP);
        __syncthreads();
        if (active) t.store(lds + wave * KP, KP);
    }
}

}  // namespace ssmi
