// scan_impl.h - K1: bandwidth-bound streaming scan with fused per-wavefront
// top-k (small query batches).  Included by scan_f32.hip / scan_f16.hip / scan_bf16.hip.
//
// Replaces the hot loop behind faiss.IndexFlatIP.search as called from
// core/index_manager.py:371 of the reference (batch-1 in the reference).
//
// Mapping: one wavefront owns one database row at a time.  A row of ld
// elements is ld*sizeof(T) contiguous bytes; lane l loads 16 B at byte offset
// (c*64 + l)*16 for chunk c, so every wave-instruction is one fully coalesced
// 1 KiB read and all chunks of a row are issued back to back before the first
// FMA consumes them (memory-level parallelism is what reaches the HBM roof; the
// arithmetic is < 5 % of the VALU budget).  The query lives in registers
// (QREG, one query) or in LDS (up to BT queries share each row read).  Scores
// never leave the wave: after a 6-step butterfly the wave-uniform score is
// compared with the wave's current k-th best and, rarely, inserted into the
// register-resident sorted list (topk.h).  At the end the waves of a block
// merge their lists through LDS and the block publishes ONE k-entry list per
// query; the block that publishes the last list of a query tile merges them all
// into (D, I) (tail_merge.h), so a batch-1 search is one launch.  Without a ticket
// counter the lists are left to a K3 launch (merge.hip).
#pragma once
#include "common.h"
#include "topk.h"
#include "tail_merge.h"

namespace ssmi {

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> { static const int V = 4; };
template <> struct ElemTraits<__half> { static const int V = 8; };
template <> struct ElemTraits<bf16_t> { static const int V = 8; };

__device__ __forceinline__ uint4 load16(const void* p) {
    return *reinterpret_cast<const uint4*>(p);
}
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load16_nt(const void* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// the same through an explicitly global pointer (a row pointer that was read from LDS or memory is a flat
// pointer to the compiler, and flat loads also tick the LDS counter)
__device__ __forceinline__ uint4 load16_nt_g(const void* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const __attribute__((address_space(1))) u32x4_t*>(
        reinterpret_cast<uintptr_t>(p)));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <typename T> struct Unpack;
template <> struct Unpack<float> {
    __device__ __forceinline__ static void run(const uint4& x, float* f) {
        f[0] = __uint_as_float(x.x); f[1] = __uint_as_float(x.y);
        f[2] = __uint_as_float(x.z); f[3] = __uint_as_float(x.w);
    }
};
template <> struct Unpack<__half> {
    __device__ __forceinline__ static void run(const uint4& x, float* f) {
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const __half2 h = *reinterpret_cast<const __half2*>(&w[i]);
            const float2 p = __half22float2(h);
            f[2 * i] = p.x; f[2 * i + 1] = p.y;
        }
    }
};

template <> struct Unpack<bf16_t> {   // bf16 -> fp32 is a shift / a mask: element 2i sits in the low half of word i
    __device__ __forceinline__ static void run(const uint4& x, float* f) {
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};

// LDS image of one query: per 64*V-element chunk, V/4 planes of [64 lanes][4]
// floats, so that every ds_read_b128 of a wave is 1 KiB contiguous.
template <int V>
__device__ __forceinline__ int q_lds_index(int elem) {
    const int chunk = elem / (64 * V), in = elem % (64 * V);
    const int lane = in / V, j = in % V;
    return chunk * 64 * V + (j / 4) * 256 + lane * 4 + (j % 4);
}

// T: storage type; NCH: chunks per row when known at compile time (0 = runtime)
// BT: queries per block pass; R: 64*R = per-wave list length; QREG: query in
// registers (BT == 1, NCH > 0 only).
template <typename T, int NCH, int BT, int R, bool QREG, bool NT>
__global__ void __launch_bounds__(QREG ? (R <= 4 ? 768 : 512) : 1024) flat_scan_topk_kernel(RowView X, const float* __restrict__ Q, int nq, int k,
                                      skey_t* partial, TailMerge tail) {
    constexpr int V = ElemTraits<T>::V;
    static_assert(!QREG || (BT == 1 && NCH > 0), "QREG needs one query and a static row length");
    extern __shared__ __align__(16) unsigned char smem[];

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int ld = X.ld;
    const int q0 = blockIdx.y * BT;
    const int nch = NCH > 0 ? NCH : (ld + 64 * V - 1) / (64 * V);

    // Rows -> blocks: the grid sweeps the database as ONE front.  A stripe is gridDim.x*nwaves consecutive rows, block
    // vb takes rows vb*nwaves .. vb*nwaves+nwaves-1 of every stripe, and the blocks of one XCD (blockIdx.x % 8) sit on
    // neighbouring stripes parts.  At any moment the whole chip reads a window of a few 2 MiB pages (an XCD: one or two)
    // instead of one page per block spread over the database: at 100 GB per GPU (12.5M fp16 rows) one contiguous slice
    // per block measured 0.83 / 0.859 of 8 TB/s on two boxes, this form 0.855 / 0.864; no difference at 1M rows.
#ifdef SSMI_SCAN_STAMP
    // diagnostic: tail.id_offset bit 62 swaps the stripe parts of neighbouring blocks (is a slow block slow because
    // of its XCD or because of its rows?)
    const unsigned pb = (tail.id_offset >> 62) & 1 ? (blockIdx.x ^ 1u) : blockIdx.x;
#else
    const unsigned pb = blockIdx.x;
#endif
    const unsigned vb = (gridDim.x % 8 == 0) ? (pb % 8) * (gridDim.x / 8) + pb / 8 : pb;
    const long long row_step = (long long)gridDim.x * nwaves;
    const long long seg_mask = (1ll << X.seg_shift) - 1;

    float qreg[QREG ? NCH : 1][V];
    float* qs = reinterpret_cast<float*>(smem);
    const int ldq = nch * 64 * V;  // LDS row length (ld rounded up to whole chunks)
    if (QREG) {
        const int qi = q0 < nq ? q0 : nq - 1;
#pragma unroll
        for (int c = 0; c < (QREG ? NCH : 1); ++c) {
            const float4 a = *reinterpret_cast<const float4*>(Q + (size_t)qi * ld + (c * 64 + lane) * V);
            qreg[c][0] = a.x; qreg[c][1] = a.y; qreg[c][2] = a.z; qreg[c][3] = a.w;
            if (V == 8) {
                const float4 b = *reinterpret_cast<const float4*>(Q + (size_t)qi * ld + (c * 64 + lane) * V + 4);
                qreg[c][V - 4] = b.x; qreg[c][V - 3] = b.y; qreg[c][V - 2] = b.z; qreg[c][V - 1] = b.w;
            }
        }
    } else {
        for (int b = 0; b < BT; ++b) {
            const int qi = (q0 + b) < nq ? (q0 + b) : nq - 1;
            for (int e = threadIdx.x; e < ldq; e += blockDim.x)
                qs[b * ldq + q_lds_index<V>(e)] = (e < ld) ? Q[(size_t)qi * ld + e] : 0.f;
        }
        __syncthreads();
    }

    WaveTopK<R> top[BT];
    skey_t thr[BT];
    const skey_t thr_floor = floor_key(tail.min_ord);  // scores under the search's floor never enter a list
#pragma unroll
    for (int b = 0; b < BT; ++b) { top[b].init(); thr[b] = thr_floor; }

    for (long long row = (long long)vb * nwaves + wave; row < X.nrows; row += row_step) {
        const unsigned char* base = reinterpret_cast<const unsigned char*>(X.segs[row >> X.seg_shift]) +
                                    (size_t)(row & seg_mask) * (size_t)ld * sizeof(T);
        float acc[BT][4];
#pragma unroll
        for (int b = 0; b < BT; ++b) acc[b][0] = acc[b][1] = acc[b][2] = acc[b][3] = 0.f;

        if (QREG) {
            // all chunks of the row (NCH KiB per wave) are issued back to back, then consumed
            uint4 x[NCH > 0 ? NCH : 1];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const size_t off = (size_t)(c * 64 + lane) * 16;
                x[c] = NT ? load16_nt(base + off) : load16(base + off);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                float xf[V];
                Unpack<T>::run(x[c], xf);
#pragma unroll
                for (int j = 0; j < V; ++j) acc[0][j & 3] = fmaf(xf[j], qreg[c][j], acc[0][j & 3]);
            }
        } else {
            // G chunks in flight per wave; queries come from LDS (one conflict-free
            // ds_read_b128 per query per 16 B of row data)
            constexpr int G = BT >= 8 ? (V == 8 ? 1 : 4) : 8;  // 8 queries x 8 elements per 16 B: one chunk keeps fp16 rows inside the register budget
            const int row_bytes = ld * (int)sizeof(T);
#pragma unroll 1
            for (int c0 = 0; c0 < nch; c0 += G) {
                uint4 x[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const int off = ((c0 + g) * 64 + lane) * 16;
                    x[g] = make_uint4(0, 0, 0, 0);
                    if (off < row_bytes) x[g] = NT ? load16_nt(base + off) : load16(base + off);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (c0 + g < nch) {
                        float xf[V];
                        Unpack<T>::run(x[g], xf);
#pragma unroll
                        for (int b = 0; b < BT; ++b) {
#pragma unroll
                            for (int p = 0; p < V / 4; ++p) {
                                const float4 t4 = *reinterpret_cast<const float4*>(qs + b * ldq + (c0 + g) * 64 * V + p * 256 + lane * 4);
                                acc[b][0] = fmaf(xf[4 * p], t4.x, acc[b][0]);
                                acc[b][1] = fmaf(xf[4 * p + 1], t4.y, acc[b][1]);
                                acc[b][2] = fmaf(xf[4 * p + 2], t4.z, acc[b][2]);
                                acc[b][3] = fmaf(xf[4 * p + 3], t4.w, acc[b][3]);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            float s = (acc[b][0] + acc[b][1]) + (acc[b][2] + acc[b][3]);
            s = wave_sum(s);
            s = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s)));
            const skey_t key = make_key(s, (uint32_t)row);
            if (key > thr[b]) {
                top[b].insert(key);
                thr[b] = kmax(top[b].kth(k), thr_floor);
            }
        }
    }

#ifdef SSMI_SCAN_STAMP
    // diagnostic build: wall clock (100 MHz) at which every wave finished its rows, behind the partial lists
    if (lane == 0 && blockIdx.y == 0)
        reinterpret_cast<unsigned long long*>(partial + (size_t)nq * gridDim.x * k)[(size_t)blockIdx.x * nwaves + wave] = wall_clock64();
#endif
    // block-level merge through LDS (reuses the query area)
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
    // the block that publishes the last list of this query tile merges them all (tail_merge.h)
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
