// mfma_common.h - shared pieces of the two large-batch MFMA kernels (mfma.hip, mfma_pp.hip).
#pragma once
#include "kernels.h"
#include "topk.h"

namespace ssmi {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

// One 16x16x32 MFMA on 16-bit operands, fp32 accumulate.  The kernels move the operands as opaque 16-byte fragments
// (half8_t is only the container); BF says which format the bits are in.  Same rate, same layout for both.
template <bool BF>
__device__ __forceinline__ f32x4_t mfma16(const half8_t& a, const half8_t& b, const f32x4_t& c) {
    if constexpr (BF) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

static constexpr int TM = 256, TN = 256, BK = 64;
static constexpr int TILE_BYTES = TM * BK * 2;  // one operand tile: 32 KiB
static constexpr int CAND_CAP = 512, CAND_TRIG = 256;
static constexpr int XSYNC_STRIDE = 32;   // unsigned words between the arrival counters of two XCDs (one cache line each)

struct MfmaParams {
    RowView X;                 // fp16 or bf16 rows (bf16 says which), ld % 64 == 0
    bool bf16;
    const _Float16* Qh;        // [nqt*256][q_pitch] 16-bit queries in the rows' format, zero rows beyond nq
    int q_pitch;               // ld (128-query kernels) or ld + 64 (256-query kernel: see cvt_queries_f16_kernel)
    int nq, k, nqt, nslices;
    long long rows_per_slice;  // multiple of 256
    skey_t* partial;           // [nq][nslices][k]
    skey_t* cand;              // [gridDim.x][256][CAND_CAP]
    float thr_floor;           // score floor of the search (SSMI_NEG_MAX = none): thresholds start and stay at or above it
    unsigned* gthr;            // [nqt*256] per query: highest k-th score any slice has reached (f32_to_ord bits,
                               // 0 = none), zeroed by the query conversion; 256-query kernel only
    // [nqt*256][nslices][gtop_m] the best gtop_m scores of every slice's list (f32_to_ord bits, 0 = none), zeroed
    // by the query conversion: every entry is the score of a distinct database row, so the k-th largest entry of a
    // query is a lower bound of its final k-th score - far tighter than any single slice's own k-th score
    unsigned* gtop;
    int gtop_m;
    int seed_first;            // 256-query kernel, k <= 16: the first row tile of a slice seeds its own threshold
    unsigned fold_mask;        // 256-query kernel: a scheduled fold follows row tile 2^b of the slice when bit b is set
    int patch_a;               // 256-query kernel: query tiles per XCD patch (0: an XCD takes whole slices, all query tiles)
    unsigned* xsync;           // 256-query kernel: [8][32] arrival counters, one per XCD (workgroups i % 8), zeroed by the query
                               // conversion; nullptr: the workgroups of an XCD run free
    // 256-query kernel: {shader clock, 100 MHz clock} at the end and at the start of workgroup 0 (s_memtime / s_memrealtime):
    // cycles / real time = the clock the chip held under this launch (ssmi_index_get_kernel_clock)
    unsigned long long* clk;
    int xsync_polls;           // ... bound of one wait, in polls of about a microsecond
    int fold16;                // k <= 16: the four-queries-per-wave fold (fold_queries16)
    int xsync_from;            // ... they meet from this row tile of their slices on (the first tiles carry the heavy folds)
#ifdef SSMI_STAMP
    unsigned long long* dbg;   // diagnostic build only: per-wave cycle sums
#endif
};

// same with a compile-time byte offset on the global address.  The instruction adds its immediate
// field to BOTH the global address and the LDS address (M0 + imm), so the LDS base is pre-biased.
#define GLDS16O(gptr, ldsptr, imm)                                                                \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                     (__attribute__((address_space(3))) void*)((ldsptr) - (imm)), 16, imm, 0)
#define GLDS16(gptr, ldsptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 16, 0, 0)

#define GLDS4(gptr, ldsptr)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),      \
                                     (__attribute__((address_space(3))) void*)(ldsptr), 4, 0, 0)

// Query-slab DMA of the register-ring kernels (mfma_r128.hip, ivf_lists_mfma.hip).  The slab of one
// 64-deep K step is [k-half][16*NT queries][64 B] (k-half stride 8 KiB, 16-B chunks XOR-swizzled by
// row); its 2*NT KiB are split EVENLY over the 8 waves, so that every wave issues the same number of
// vector-memory instructions per step and one counted s_waitcnt serves all of them:
//   NT = 8: wave w brings queries 16w..16w+15 of both k-halves      (2 x 16 B per lane)
//   NT = 4: wave w brings queries 16(w&3).. of k-half w>>2          (1 x 16 B per lane)
//   NT = 1: wave w brings queries 4(w&3)..4(w&3)+3 of k-half w>>2   (1 x  4 B per lane)
template <int NT>
struct QSlabDma {
    static constexpr int OPS = NT == 8 ? 2 : 1;  // DMA instructions per wave and step
    // query row (within the tile) whose bytes this lane moves
    __device__ __forceinline__ static int qrow(int wave, int lane) {
        return NT == 8 ? wave * 16 + (lane >> 2) : (NT == 4 ? (wave & 3) * 16 + (lane >> 2) : (wave & 3) * 4 + (lane >> 4));
    }
    // byte offset of the lane's source inside its query row, for K step 0 (a step advances 128 B)
    __device__ __forceinline__ static int src_off(int wave, int lane) {
        const int f = (-(qrow(wave, lane) >> 2)) & 3;
        if (NT == 8) return ((lane & 3) ^ f) * 16;
        if (NT == 4) return (wave >> 2) * 64 + ((lane & 3) ^ f) * 16;
        return (wave >> 2) * 64 + ((((lane & 15) >> 2) ^ f) * 16) + (lane & 3) * 4;
    }
    // byte offset of the wave's piece inside a slab slot
    __device__ __forceinline__ static int dst_off(int wave) {
        return NT == 8 ? wave * 1024 : (NT == 4 ? (wave >> 2) * 8192 + (wave & 3) * 1024 : (wave >> 2) * 8192 + (wave & 3) * 256);
    }
    __device__ __forceinline__ static void issue(const unsigned char* src /*row + src_off + step*128*/, unsigned char* dst) {
        if (NT == 8) { GLDS16(src, dst); GLDS16(src + 64, dst + 8192); }
        else if (NT == 4) GLDS16(src, dst);
        else GLDS4(src, dst);
    }
};

// LDS words behind the four operand slabs: thr[256] (float), cnt[256], kept[256], flag
static constexpr int EPI_LDS_BYTES = 3 * 256 * 4 + 16;

// ---- published tops (MfmaParams::gtop) ----
// The wave that has just folded (q, slice) publishes the list's best gtop_m scores (lanes 0.. of its first register)
// with device-scope stores; any wave may later read all of a query's entries with device-scope loads.  A stale or
// still empty entry only weakens the bound, it never invalidates it.
__device__ __forceinline__ void publish_tops(unsigned* gtop_qs /*entry of (q, slice)*/, int m, skey_t v0) {
    const int lane = lane_id();
    if (gtop_qs && lane < m) __hip_atomic_store(gtop_qs + lane, (unsigned)(v0 >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// 64*RV unsigned values spread over the lanes (element e in register e/64 of lane e%64), sorted largest first by
// the bitonic network of topk.h on 32-bit values (half the instructions of the 64-bit key sort)
template <int RV>
struct WaveSort32 {
    unsigned v[RV];
    template <int STRIDE>
    __device__ __forceinline__ void lane_stage(int size_bit) {
        const int lane = lane_id();
        const bool lower = (lane & STRIDE) == 0;
#pragma unroll
        for (int r = 0; r < RV; ++r) {
            const bool desc = ((r * 64 + lane) & size_bit) == 0;
            const unsigned p = (unsigned)xor_lane<STRIDE>((int)v[r]);
            v[r] = (desc == lower) ? max(v[r], p) : min(v[r], p);
        }
    }
    __device__ __forceinline__ void reg_stage(int rs, int size_bit) {
#pragma unroll
        for (int r = 0; r < RV; ++r)
            if ((r & rs) == 0 && r + rs < RV) {
                const bool desc = ((r * 64) & size_bit) == 0;
                const unsigned hi = max(v[r], v[r + rs]), lo = min(v[r], v[r + rs]);
                v[r] = desc ? hi : lo;
                v[r + rs] = desc ? lo : hi;
            }
    }
    template <int START>
    __device__ __forceinline__ void lane_stages(int size_bit) {
        if constexpr (START >= 32) lane_stage<32>(size_bit);
        if constexpr (START >= 16) lane_stage<16>(size_bit);
        if constexpr (START >= 8) lane_stage<8>(size_bit);
        if constexpr (START >= 4) lane_stage<4>(size_bit);
        if constexpr (START >= 2) lane_stage<2>(size_bit);
        lane_stage<1>(size_bit);
    }
    __device__ __forceinline__ void sort_desc() {
        lane_stages<1>(2);
        lane_stages<2>(4);
        lane_stages<4>(8);
        lane_stages<8>(16);
        lane_stages<16>(32);
        lane_stages<32>(RV == 1 ? 0 : 64);
#pragma unroll
        for (int size = 128; size <= 64 * RV; size <<= 1) {
            const int sb = (size == 64 * RV) ? 0 : size;
#pragma unroll
            for (int rs = size >> 7; rs >= 1; rs >>= 1) reg_stage(rs, sb);
            lane_stages<32>(sb);
        }
    }
    __device__ __forceinline__ unsigned kth(int k) const {  // 1-based, wave-uniform
        const int e = k - 1;
        unsigned sel = 0;
#pragma unroll
        for (int r = 0; r < RV; ++r) sel = ((e >> 6) == r) ? v[r] : sel;
        return (unsigned)__builtin_amdgcn_readlane((int)sel, e & 63);
    }
};
template <int RV>
__device__ __forceinline__ unsigned tops_bound_r(const unsigned* g, int V, int k) {
    const int lane = lane_id();
    WaveSort32<RV> t;
#pragma unroll
    for (int r = 0; r < RV; ++r) {
        const int e = r * 64 + lane;
        t.v[r] = e < V ? __hip_atomic_load(g + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    }
    t.sort_desc();
    return t.kth(k);
}
// k-th largest of the first V published scores of query q (f32_to_ord bits; 0: fewer than k published).  Any
// subset of the entries bounds as well, so V is only as large as k needs: 64 entries up to k = 48, 128 up to
// k = 100, 256 beyond.  Wave-uniform, all lanes call.
__device__ __forceinline__ unsigned tops_bound(const MfmaParams& P, int q) {
    const int cap = P.k <= 48 ? 64 : (P.k <= 100 ? 128 : 256);
    int V = P.nslices * P.gtop_m;
    if (V > cap) V = cap;
    if (P.k > V) return 0u;
    const unsigned* g = P.gtop + (size_t)q * P.nslices * P.gtop_m;
    if (V <= 64) return tops_bound_r<1>(g, V, P.k);
    if (V <= 128) return tops_bound_r<2>(g, V, P.k);
    return tops_bound_r<4>(g, V, P.k);
}

// Fold the candidate buffer of one query into its sorted k-list and raise its threshold.
// One wave per query.  Candidates arrive unsorted: they are sorted in registers (bitonic
// network, 64 or 256 at a time) and merged with the kept list by one bitonic merge - far cheaper
// than one sorted insertion per candidate.
// `kept`: the query's k-list in global memory (k keys), valid once kept_l[ql] is set.
template <int R>
__device__ __forceinline__ void compact_list(int ql, skey_t* kept, int k, skey_t* cand_q, float* thr_l, unsigned* cnt_l,
                                             unsigned* kept_l, float thr_floor = SSMI_NEG_MAX, unsigned* gtop_qs = nullptr,
                                             int gtop_m = 0) {
    const int lane = lane_id();
    const int n = (int)min(cnt_l[ql], (unsigned)CAND_CAP);
    if (n == 0 && kept_l[ql]) return;  // nothing new: list and threshold stand
    WaveTopK<R> t;
    t.init();
    if (kept_l[ql]) t.merge_from(kept, k);
    if (n <= 64) {
        WaveTopK<1> c;
        c.v[0] = (lane < n) ? cand_q[lane] : 0;
        c.sort_desc();
        t.merge_regs(c);
    } else {
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
    }
    const skey_t thrk = t.kth(k);
    t.store(kept, k);
    publish_tops(gtop_qs, gtop_m, t.v[0]);
    if (lane == 0) {
        kept_l[ql] = 1;
        cnt_l[ql] = 0;
        thr_l[ql] = (thrk == 0) ? thr_floor : fmaxf(key_score(thrk), thr_floor);
    }
}
// Flat kernels: query q of slice `slice` keeps its list at partial[q][slice][k]
template <int R>
__device__ __forceinline__ void compact_query(int ql, int q, const MfmaParams& P, int slice, skey_t* cand_q,
                                              float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    compact_list<R>(ql, P.partial + ((size_t)q * P.nslices + slice) * P.k, P.k, cand_q, thr_l, cnt_l, kept_l, P.thr_floor,
                    P.gtop ? P.gtop + ((size_t)q * P.nslices + slice) * P.gtop_m : nullptr, P.gtop_m);
}

// LDS atomic add / store issued as opaque instructions: after an LDS-DMA (global_load_lds) the
// compiler's wait-count pass puts `s_waitcnt vmcnt(0)` in front of every LDS atomic or store it
// can see, because it cannot tell that they do not touch the DMA's destination slabs; in the
// epilogue that would stall on the next tile's pipeline fill (an HBM round trip per row tile).
__device__ __forceinline__ unsigned lds_add_rtn(unsigned* p, unsigned v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)p;
    unsigned old;
    asm volatile("ds_add_rtn_u32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(old) : "v"(addr), "v"(v) : "memory");
    return old;
}
__device__ __forceinline__ void lds_min_u32(unsigned* p, unsigned v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)p;
    asm volatile("ds_min_u32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_store(unsigned* p, unsigned v) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned*)p;
    asm volatile("ds_write_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(addr), "v"(v) : "memory");
}

// The fold of a whole 256-query tile, eight waves striding over the queries.  One fold is a chain of memory
// round trips (candidate keys and kept list in, sorted list out) around a short sort, so each wave loads the
// NEXT query's candidates and list before it sorts the current one, and writes its LDS words through the
// opaque store above (no compiler-added wait for the list store in front of them).
template <int R>
struct FoldOperands {
    int n;            // candidates in the buffer (clamped to its capacity)
    unsigned has;     // the query already owns a k-list
    skey_t c;         // candidate `lane` (the first 64)
    skey_t kept[R];   // list element r*64 + lane, best first
};
template <int R>
__device__ __forceinline__ FoldOperands<R> fold_load(int ql, const skey_t* kept, int k, const skey_t* cand_q,
                                                     const unsigned* cnt_l, const unsigned* kept_l) {
    const int lane = lane_id();
    FoldOperands<R> o;
    o.n = (int)min(cnt_l[ql], (unsigned)CAND_CAP);
    o.has = kept_l[ql];
    o.c = (lane < o.n) ? cand_q[lane] : 0;
#pragma unroll
    for (int r = 0; r < R; ++r) o.kept[r] = (o.has && r * 64 + lane < k) ? kept[r * 64 + lane] : 0;
    return o;
}
template <int R>
__device__ __forceinline__ void fold_one(int ql, const FoldOperands<R>& o, skey_t* kept, int k, const skey_t* cand_q,
                                         float* thr_l, unsigned* cnt_l, unsigned* kept_l, float thr_floor,
                                         unsigned* gtop_qs, int gtop_m) {
    const int lane = lane_id();
    if (o.n == 0 && o.has) return;  // nothing new: list and threshold stand
    WaveTopK<R> t;
#pragma unroll
    for (int r = 0; r < R; ++r) t.v[r] = o.kept[r];  // sorted best first, zero padded
    if (o.n <= 64) {
        WaveTopK<1> c;
        c.v[0] = o.c;
        c.sort_desc();
        t.merge_regs(c);
    } else {
        for (int base = 0; base < o.n; base += 256) {
            WaveTopK<4> c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = base + r * 64 + lane;
                c.v[r] = (e < o.n) ? cand_q[e] : 0;
            }
            c.sort_desc();
            t.merge_regs(c);
        }
    }
    const skey_t thrk = t.kth(k);
    t.store(kept, k);
    publish_tops(gtop_qs, gtop_m, t.v[0]);
    if (lane == 0) {
        lds_store(&kept_l[ql], 1u);
        lds_store(&cnt_l[ql], 0u);
        lds_store(reinterpret_cast<unsigned*>(&thr_l[ql]), __float_as_uint((thrk == 0) ? thr_floor : fmaxf(key_score(thrk), thr_floor)));
    }
}
template <int R>
__device__ __forceinline__ void fold_queries(int wave, int q0, const MfmaParams& P, int slice, skey_t* cand,
                                             float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    int nql = P.nq - q0;
    if (nql > 256) nql = 256;
    auto list_of = [&](int ql) { return P.partial + ((size_t)(q0 + ql) * P.nslices + slice) * P.k; };
    int ql = wave;
    if (ql >= nql) return;
    FoldOperands<R> cur = fold_load<R>(ql, list_of(ql), P.k, cand + (size_t)ql * CAND_CAP, cnt_l, kept_l);
    for (; ql < nql; ql += 8) {
        const int qn = ql + 8;
        FoldOperands<R> nxt = cur;
        if (qn < nql) nxt = fold_load<R>(qn, list_of(qn), P.k, cand + (size_t)qn * CAND_CAP, cnt_l, kept_l);
        fold_one<R>(ql, cur, list_of(ql), P.k, cand + (size_t)ql * CAND_CAP, thr_l, cnt_l, kept_l, P.thr_floor,
                    P.gtop ? P.gtop + ((size_t)(q0 + ql) * P.nslices + slice) * P.gtop_m : nullptr, P.gtop_m);
        cur = nxt;
    }
}

// The same fold for k <= 16, FOUR queries per wave: a query's list and a chunk of 16 of its candidates live in one
// row of 16 lanes, so every compare-exchange partner is a DPP move away (no permlane swaps, no ds_bpermute) and one
// pass of the network serves four queries.  Per chunk: bitonic sort of the 16 candidates (10 stages), element-wise
// maximum with the mirrored list (the best 16 of the union, as a bitonic sequence), bitonic merge (4 stages).  Wave w
// takes the queries w, w+8, w+16, ... as fold_queries does (the threshold exchange behind the fold relies on that),
// four at a time: row g of pass `it` holds query w + 8*(4*it + g).  The next pass's operands are loaded ahead.
__device__ __forceinline__ skey_t row_mirror64(skey_t v) {
    const int lo = __builtin_amdgcn_mov_dpp((int)(unsigned)(v & 0xFFFFFFFFull), 0x140, 0xF, 0xF, true);  // row_mirror
    const int hi = __builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), 0x140, 0xF, 0xF, true);
    return ((skey_t)(unsigned)hi << 32) | (skey_t)(unsigned)lo;
}
struct Fold16Operands {
    int ql;        // the row's query (within the tile), -1: none
    int n;         // its candidates (clamped to the buffer's capacity)
    unsigned has;  // it already owns a k-list
    skey_t list;   // list element lane % 16, best first (0: none)
    skey_t c;      // candidate lane % 16 of the first chunk
};
// list_of(ql): the query's k-list in global memory; tops_of(ql): where it publishes its best gtop_m scores (or nullptr)
template <class ListOf>
__device__ __forceinline__ Fold16Operands fold16_load(int it, int wave, int nql, int k, ListOf list_of, const skey_t* cand,
                                                      const unsigned* cnt_l, const unsigned* kept_l) {
    const int lane = lane_id(), j = lane & 15;
    Fold16Operands o;
    o.ql = wave + 8 * (4 * it + (lane >> 4));
    if (o.ql >= nql) o.ql = -1;
    const int qs = o.ql < 0 ? 0 : o.ql;
    o.n = o.ql < 0 ? 0 : (int)min(cnt_l[qs], (unsigned)CAND_CAP);
    o.has = o.ql < 0 ? 1u : kept_l[qs];
    o.list = (o.ql >= 0 && o.has && j < k) ? list_of(qs)[j] : 0;
    o.c = (j < o.n) ? cand[(size_t)qs * CAND_CAP + j] : 0;
    return o;
}
template <class ListOf, class TopsOf>
__device__ __forceinline__ void fold16_pass(const Fold16Operands& o, int k, float thr_floor, ListOf list_of, TopsOf tops_of, int gtop_m,
                                            const skey_t* cand, float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    const int lane = lane_id(), j = lane & 15;
    const bool work = o.ql >= 0 && !(o.n == 0 && o.has);   // nothing new: list and threshold stand
    if (__builtin_amdgcn_ballot_w64(work) == 0) return;
    int nmax = o.n;                          // chunks: as many as the longest of the four buffers needs (n is uniform
    nmax = max(nmax, xor_lane<16>(nmax));    // within a row, so any lane of the other rows will do)
    nmax = max(nmax, xor_lane<32>(nmax));
    nmax = __builtin_amdgcn_readfirstlane(nmax);
    WaveTopK<1> t, c;
    t.v[0] = o.list;
    c.v[0] = o.c;
    const size_t cbase = (size_t)(o.ql < 0 ? 0 : o.ql) * CAND_CAP;
    for (int base = 0; base < nmax; base += 16) {
        if (base) c.v[0] = (base + j < o.n) ? cand[cbase + base + j] : 0;
        c.lane_stages<1>(2);
        c.lane_stages<2>(4);
        c.lane_stages<4>(8);
        c.lane_stages<8>(0);                 // every row best first
        t.v[0] = kmax(t.v[0], row_mirror64(c.v[0]));
        t.lane_stages<8>(0);
    }
    if (!work) return;
    if (j < k) list_of(o.ql)[j] = t.v[0];
    unsigned* tops = tops_of(o.ql);
    if (tops && j < gtop_m) __hip_atomic_store(tops + j, (unsigned)(t.v[0] >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (j == 0) {
        lds_store(&kept_l[o.ql], 1u);
        lds_store(&cnt_l[o.ql], 0u);
    }
    if (j == k - 1) {                        // the lane that holds the k-th key raises the threshold
        const skey_t thrk = t.v[0];
        lds_store(reinterpret_cast<unsigned*>(&thr_l[o.ql]), __float_as_uint((thrk == 0) ? thr_floor : fmaxf(key_score(thrk), thr_floor)));
    }
}
template <class ListOf, class TopsOf>
__device__ __forceinline__ void fold_lists16(int wave, int nql, int k, float thr_floor, ListOf list_of, TopsOf tops_of, int gtop_m,
                                             const skey_t* cand, float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    if (wave >= nql) return;
    const int passes = (nql - wave + 31) / 32;   // queries w, w+8, ...: four per pass
    Fold16Operands cur = fold16_load(0, wave, nql, k, list_of, cand, cnt_l, kept_l);
    for (int it = 0; it < passes; ++it) {
        Fold16Operands nxt = cur;
        if (it + 1 < passes) nxt = fold16_load(it + 1, wave, nql, k, list_of, cand, cnt_l, kept_l);
        fold16_pass(cur, k, thr_floor, list_of, tops_of, gtop_m, cand, thr_l, cnt_l, kept_l);
        cur = nxt;
    }
}
// Flat kernels: query q0 + ql of slice `slice` keeps its list at partial[q][slice][k]
__device__ __forceinline__ void fold_queries16(int wave, int q0, const MfmaParams& P, int slice, const skey_t* cand,
                                               float* thr_l, unsigned* cnt_l, unsigned* kept_l) {
    int nql = P.nq - q0;
    if (nql > 256) nql = 256;
    fold_lists16(wave, nql, P.k, P.thr_floor,
                 [&](int ql) { return P.partial + ((size_t)(q0 + ql) * P.nslices + slice) * P.k; },
                 [&](int ql) -> unsigned* { return P.gtop ? P.gtop + ((size_t)(q0 + ql) * P.nslices + slice) * P.gtop_m : nullptr; },
                 P.gtop_m, cand, thr_l, cnt_l, kept_l);
}

__device__ __forceinline__ float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Threshold filter over one wave's 128 x 64 accumulator tile (acc[m][n]: rows m*16 + fq*4 + i of
// the wave's row half, query n*16 + frow of its query quarter).  Per query column the 32 values
// of a lane are first reduced to per-accumulator maxima and one column maximum: in steady state
// almost no column holds a survivor and the whole column costs 20 VALU instructions and one
// wave-uniform branch; in a column that does, the lanes that hold one walk their accumulators.
// FULL = false adds the row-range test for the last, partial row tile of a slice (its padding
// rows repeat the last valid row).
template <int MT, int NT, bool FULL>
__device__ __forceinline__ void filter_tile(const f32x4_t (&acc)[MT][NT], int row_base, int q_base, int frow, int fq,
                                            long long row0, int rows_valid, skey_t* cand, const float* thr_l,
                                            unsigned* cnt_l, unsigned* flag_l, const uint32_t* ids = nullptr,
                                            bool seed16 = false) {
    // ids == nullptr: the id of row rr of the tile is row0 + rr (Flat).  Otherwise ids[rr - row_base] (IVF: the
    // wave's rows are the slots of one list page, whose vector ids are looked up only for survivors).
    // acc[m][n]: rows row_base + m*16 + fq*4 + i of the row tile, query q_base + n*16 + frow of the query tile.
    // keep the per-column LDS/global addresses out of the main loop's register budget: they are
    // recomputed here per row tile instead of being hoisted (and spilled) by the compiler
    asm volatile("" : "+v"(frow));
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int ql = q_base + n * 16 + frow;
        float thr = thr_l[ql];
        if (FULL && seed16) {
            // First row tile of a slice, k <= 16 (wave-uniform flag): no list exists yet, so every row would pass the
            // filter and the first fold would sort MT*16*4*(waves per column) candidates per query.  The 4th largest of
            // a lane's values, minimised over the four lanes that share the column, has at least 16 of THIS wave's
            // rows at or above it: every row of the tile's best 16 that lies in this wave's rows is among them, so it
            // may serve as this wave's threshold for this tile (whole tiles only: padding rows repeat a row).
            float a0 = -__builtin_inff(), a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float v = acc[m][n][i];
                    const float t = fminf(v, a0); a0 = fmaxf(v, a0);
                    const float u = fminf(t, a1); a1 = fmaxf(t, a1);
                    const float w = fminf(u, a2); a2 = fmaxf(u, a2);
                    a3 = fmaxf(w, a3);
                }
            a3 = fminf(a3, __int_as_float(xor_lane<16>(__float_as_int(a3))));
            a3 = fminf(a3, __int_as_float(xor_lane<32>(__float_as_int(a3))));
            thr = fmaxf(thr, a3);
        }
        float mm[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) mm[m] = vmax2(vmax3(acc[m][n][0], acc[m][n][1], acc[m][n][2]), acc[m][n][3]);
        float mx = mm[0];
#pragma unroll
        for (int m = 1; m + 1 < MT; m += 2) mx = vmax3(mx, mm[m], mm[m + 1]);
        if ((MT & 1) == 0) mx = vmax2(mx, mm[MT - 1]);
        if (__builtin_amdgcn_ballot_w64(mx >= thr) == 0) continue;
        // Survivors in this column: typically one to three values in one or two lanes.  Only those lanes go on
        // (the others are masked off and the compiler skips empty blocks with execz branches): accumulator by
        // accumulator, a slot from the query's LDS counter per survivor, one store.  The order of a buffer's
        // candidates is immaterial (the fold sorts them).
        if (mx >= thr) {
            skey_t* cq = cand + (size_t)ql * CAND_CAP;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                // wave-uniform and marked unlikely: the accumulator's code is laid out of line, so the usual
                // accumulator without a survivor costs a compare and a branch that is not taken
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(mm[m] >= thr) != 0, 0)) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float s = acc[m][n][i];
                        const int rr = row_base + m * 16 + fq * 4 + i;
                        if (s >= thr && (FULL || rr < rows_valid)) {
                            const unsigned pos = lds_add_rtn(&cnt_l[ql], 1u);
                            if (pos + 1 >= (unsigned)CAND_TRIG) lds_store(flag_l, 1u);
                            if (pos < (unsigned)CAND_CAP) cq[pos] = make_key(s, ids ? ids[rr - row_base] : (uint32_t)(row0 + rr));
                        }
                    }
                }
            }
        }
    }
}

// ping-pong kernel, 256-query tile (mfma_pp.hip)
hipError_t launch_flat_mfma_pp(const MfmaParams& P, int r, int grid, size_t lds, hipStream_t s);
// streaming kernel, 128-query tile, for batches up to 128 (mfma_n128.hip)
hipError_t launch_flat_mfma_n128(const MfmaParams& P, int r, int grid, size_t lds, hipStream_t s);
// register-streaming kernel, 128-query tile, rows of a multiple of 512 elements (mfma_r128.hip)
hipError_t launch_flat_mfma_r128(const MfmaParams& P, int r, int grid, hipStream_t s);
bool mfma_r128_supported(int ld, int k);
static constexpr int N128_RING = 6, N128_SLOT = 24 * 1024, N128_TN = 128;

}  // namespace ssmi
