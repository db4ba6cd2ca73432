// topk.h - per-wavefront top-k kept in registers (gfx950, wave64).
//
// A wave keeps its best 64*R candidates as 64-bit keys (common.h), sorted
// best-first and spread over the lanes: element e lives in register e/64 of
// lane e%64.  Two operations:
//   insert()      one candidate, guarded by the caller's threshold compare;
//                 rare once the list is warm, so it may cost tens of cycles.
//   merge_from()  a whole sorted list (LDS or global): element-wise max against
//                 the reversed list gives a bitonic sequence holding the best
//                 64*R of the union, which a log2(64*R)-stage bitonic merge
//                 re-sorts.  Used for the block-level and cross-block merges,
//                 so the score matrix is never written anywhere.
#pragma once
#include "common.h"

namespace ssmi {

__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ skey_t shfl_xor64(skey_t v, int m) {
    int lo = (int)(unsigned)(v & 0xFFFFFFFFull), hi = (int)(unsigned)(v >> 32);
    lo = __shfl_xor(lo, m); hi = __shfl_xor(hi, m);
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

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
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

    // Full bitonic sort of arbitrary contents, best first.
    __device__ __forceinline__ void sort_desc() {
        const int lane = lane_id();
#pragma unroll
        for (int size = 2; size <= KP; size <<= 1) {
#pragma unroll
            for (int stride = size >> 1; stride >= 1; stride >>= 1) {
                if (stride >= 64) {
                    const int rs = stride >> 6;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        if ((r & rs) == 0) {
                            const bool desc = ((r * 64) & size) == 0;  // size >= 128: the bit lives in r
                            const skey_t a = v[r], b = v[r + rs];
                            const skey_t hi = kmax(a, b), lo = kmin(a, b);
                            v[r] = desc ? hi : lo;
                            v[r + rs] = desc ? lo : hi;
                        }
                    }
                } else {
                    const bool lower = (lane & stride) == 0;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool desc = (((r * 64 + lane) & size) == 0);
                        const skey_t p = shfl_xor64(v[r], stride);
                        v[r] = (desc == lower) ? kmax(v[r], p) : kmin(v[r], p);
                    }
                }
            }
        }
    }

    // re-sort a bitonic sequence, best first
    __device__ __forceinline__ void bitonic_merge() {
        const int lane = lane_id();
#pragma unroll
        for (int s = R / 2; s >= 1; s >>= 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if ((r & s) == 0) {
                    const skey_t a = v[r], b = v[r + s];
                    v[r] = kmax(a, b);
                    v[r + s] = kmin(a, b);
                }
            }
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const bool first = (lane & s) == 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const skey_t p = shfl_xor64(v[r], s);
                v[r] = first ? kmax(v[r], p) : kmin(v[r], p);
            }
        }
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
