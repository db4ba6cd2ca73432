// tail_merge.h - the merge of the per-block k-lists of a query (K3) as the TAIL of the kernel
// that produced them: every block publishes its list, takes a ticket from a device counter,
// and the block that draws the last ticket folds all lists of the query into (D, I).
//
// A batch-1 search is a short chain of launches (Flat: scan, merge; IVF: coarse scan, merge,
// plan, list scan, merge) and a dependent launch costs 5-10 us on this device while the merge
// itself needs 3-4 us, so the scans carry their merge with them.  No block ever waits for
// another one (the last ticket holder finds every list published), so there is no co-residency
// requirement.  Lists cross XCDs (one L2 each): they are published with device-scope stores that
// complete before the ticket is taken and read back with device-scope loads; a release/acquire
// fence pair instead (an L2 write-back and invalidate per block) measured +18 us per launch.
//
// Why this is sound without a fence pair, and what guards it: (1) the list is published with agent-scope relaxed
// atomic stores, which gfx950 emits as `global_store_dwordx2 ... sc1`: write-through, the bytes leave the CU and
// the XCD's L2 (MI355X_MICROARCH.md, "stores of each flavour"); (2) the publishing wave drains them with an
// inline-asm `s_waitcnt vmcnt(0)` (invisible to the compiler's wait-count pass, so it cannot be dropped or moved)
// and only then - behind the workgroup barrier of tail_last_block - one lane takes the ticket with a returning
// device-scope atomic add; (3) the block that drew the last ticket reads the lists with agent-scope relaxed atomic
// loads = `global_load_dwordx2 ... sc1`, which bypass its L1 and are served by the memory side, after a workgroup
// barrier behind the ticket.  This is the guide's "sc1 stores + drained counter add + sc1 loads" hand-off (one
// lane of each storing workgroup adds to ONE counter, the workgroup whose add came last reads).  All three are
// properties of the generated instructions, so tools/check_tail_merge.py asserts them on the assembly of every
// kernel with a tail (tests/test_tail_merge_isa_cpu.py), and tools/soak_tail_merge.py soaks the behaviour.
//
// Stands in for the heap merge inside faiss.IndexFlatIP.search / IndexIVFFlat.search
// (reference call site core/index_manager.py:371).
#pragma once
#include "common.h"
#include "topk.h"

namespace ssmi {

struct TailMerge {
    unsigned* counter;  // one ticket counter per query tile (blockIdx.y), zero between launches; nullptr: no tail, K3 merges
    float* D;           // (nq, k)
    int64_t* I;         // (nq, k)
    int64_t id_offset;
    // IVF coarse quantizer only (k = nprobe <= 64): the merged ids are the probed lists, and the block that
    // merges them also writes the exclusive prefix sum of their sizes, (nq, k+1) int64 (K7's output)
    const int* plan_sizes;
    long long* plan_prefix;
    const int* plan_page_off;  // CSR offsets of the lists' page tables ...
    int* plan_po;              // ... gathered for the probed lists, (nq, k) int32
    // score floor of the search (ssmi_index_search_min): f32_to_ord(min_score), 0 = none.  Results below it are
    // never reported; the scans also start their thresholds there instead of at "anything goes"
    unsigned min_ord;
};
// the key just below every key whose score is >= the floor (0: no floor)
__host__ __device__ inline skey_t floor_key(unsigned min_ord) { return min_ord ? (((skey_t)min_ord << 32) - 1) : 0; }

// K7 folded into a merge: ids of the sorted list -> exclusive prefix sum of list sizes (one wave, k <= 64)
struct PlanOut {
    const int* sizes;
    long long* prefix;
    const int* page_off;
    int* po;
};
__device__ __forceinline__ void write_plan(const WaveTopK<1>& t, int k, const PlanOut& pl) {
    const int lane = lane_id();
    const skey_t key = t.v[0];
    const int* sizes = pl.sizes;
    long long* prefix = pl.prefix;
    const bool have = lane < k && key != 0;
    const long long v = have ? (long long)sizes[key_id(key)] : 0;
    if (lane < k) pl.po[lane] = have ? pl.page_off[key_id(key)] : 0;
    long long incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const long long up = (long long)shfl_up64((skey_t)incl, d);
        if (lane >= d) incl += up;
    }
    if (lane < k) prefix[lane] = incl - v;
    if (lane == k - 1) prefix[k] = incl;
}

template <int R>
__device__ __forceinline__ void write_result(const WaveTopK<R>& t, int k, float* D, int64_t* I, int64_t id_offset,
                                             unsigned min_ord = 0) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        if (e < k) {
            const skey_t key = t.v[r];
            // empty slot, or a score under the search's floor (the list is sorted: the cut is a suffix)
            if (key == 0 || (unsigned)(key >> 32) < min_ord) { D[e] = SSMI_NEG_MAX; I[e] = -1; }
            else { D[e] = key_score(key); I[e] = (int64_t)key_id(key) + id_offset; }
        }
    }
}

// A key of a list that another block of the SAME launch published: read at device scope (the
// load bypasses this XCD's L2 copy of the line), no fence needed.
template <bool COHERENT>
__device__ __forceinline__ skey_t load_key(const skey_t* p) {
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

static constexpr int kMergeSurvivorCap = 256;

// All threads of a block: P sorted k-lists at `lists` ([P][k] keys) -> D[0..k), I[0..k).
// `lds` holds nwaves * 64 * R keys.
//
// k <= 64 (R == 1), the common case: the heads of the lists are P distinct candidates, so the k-th
// largest of 64 of them (the lane maxima of a wave that spreads the heads over its lanes) is a lower
// bound of the k-th best overall.  Wave 0 sorts those 64 values while every thread already has its
// share of the P*k keys in flight; the keys at or above the bound (a few dozen) are appended to a
// short LDS list and sorted by one wave.  A full bitonic sort of all P*k keys on one CU costs
// ~10 us, this form 2-3 us.  Inputs that defeat the bound (more than 256 survivors) take the full sort.
template <int R, bool COHERENT>
__device__ __forceinline__ void merge_partials(const skey_t* lists, int P, int k, skey_t* lds, int wave, int nwaves,
                                               float* D, int64_t* I, int64_t id_offset,
                                               PlanOut plan = PlanOut{nullptr, nullptr, nullptr, nullptr}, unsigned min_ord = 0) {
    const int lane = lane_id();
    const int total = P * k;
    if (R == 1 && nwaves >= 5 && total > kMergeSurvivorCap) {
        skey_t* bound_l = lds;                                        // [0]
        unsigned* count_l = reinterpret_cast<unsigned*>(lds + 1);    // [1]
        skey_t* surv = lds + 2;                                       // [2, 2 + cap)
        const int nthreads = nwaves * 64;
        const int tid = wave * 64 + lane;
        // own share of the keys: four loads in flight per thread and round; the first round is issued BEFORE the
        // bound is known (wave 0 sorts the heads meanwhile), so the merge costs one memory round trip, not two
        if (tid == 0) *count_l = 0;
        skey_t key[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + j * nthreads;
            key[j] = (e < total) ? load_key<COHERENT>(lists + e) : 0;
        }
        skey_t bound = 0;
        if (wave == 0) {
            WaveTopK<1> hd;
            hd.v[0] = 0;
            for (int p = lane; p < P; p += 64) hd.v[0] = kmax(hd.v[0], load_key<COHERENT>(lists + (size_t)p * k));
            hd.sort_desc();
            bound = hd.kth(k < 64 ? k : 64);
            if (lane == 0) *bound_l = bound;
        }
        __syncthreads();
        bound = *bound_l;
        for (int base = tid;;) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (key[j] != 0 && key[j] >= bound) {
                    const unsigned pos = atomicAdd(count_l, 1u);
                    if (pos < (unsigned)kMergeSurvivorCap) surv[pos] = key[j];
                }
            }
            base += 4 * nthreads;
            if (base >= total) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = base + j * nthreads;
                key[j] = (e < total) ? load_key<COHERENT>(lists + e) : 0;
            }
        }
        __syncthreads();
        const unsigned n = *count_l;
        if (n <= (unsigned)kMergeSurvivorCap) {
            if (wave == 0) {
                if (n <= 64) {
                    WaveTopK<1> t;
                    t.v[0] = (unsigned)lane < n ? surv[lane] : 0;
                    t.sort_desc();
                    write_result<1>(t, k, D, I, id_offset, min_ord);
                    if (plan.sizes) write_plan(t, k, plan);
                } else {
                    WaveTopK<4> t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) t.v[r] = (unsigned)(r * 64 + lane) < n ? surv[r * 64 + lane] : 0;
                    t.sort_desc();
                    WaveTopK<1> o;
                    o.v[0] = t.v[0];
                    write_result<1>(o, k, D, I, id_offset, min_ord);
                    if (plan.sizes) write_plan(o, k, plan);
                }
            }
            return;
        }
        __syncthreads();  // the bound did not thin the pool: fall through to the full sort
    }
    WaveTopK<R> t;
    t.init();
    if (R == 1) {
        // short lists: treat the P*k keys as one unsorted pool, 256 per wave-step, all loads of a
        // step in flight together (a chain of P dependent little merges is latency-bound)
        for (int base = wave * 256; base < total; base += nwaves * 256) {
            WaveTopK<4> c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = base + r * 64 + lane;
                c.v[r] = (e < total) ? load_key<COHERENT>(lists + e) : 0;
            }
            c.sort_desc();
            t.merge_regs(c);
        }
    } else if (COHERENT) {
        // stage each list through this wave's LDS slot with device-scope loads, then merge from there
        const int KP = WaveTopK<R>::KP;
        skey_t* slot = lds + (size_t)wave * KP;
        for (int p = wave; p < P; p += nwaves) {
            for (int e = lane; e < k; e += 64) slot[e] = load_key<true>(lists + (size_t)p * k + e);
            __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            t.merge_from(slot, k);
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    } else {
        for (int p = wave; p < P; p += nwaves) t.merge_from(lists + (size_t)p * k, k);
    }
    block_merge<R>(t, lds, wave, nwaves);
    if (wave == 0) {
        write_result<R>(t, k, D, I, id_offset, min_ord);
        if constexpr (R == 1) {
            if (plan.sizes) write_plan(t, k, plan);
        }
    }
}

// Publishes a block's k-list to the other blocks of the launch: device-scope (write-through) stores,
// complete before the caller takes its ticket.  Executed by the wave that holds the list.
template <int R>
__device__ __forceinline__ void publish_list(const WaveTopK<R>& t, skey_t* dst, int n) {
    const int lane = lane_id();
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = r * 64 + lane;
        if (e < n) __hip_atomic_store(dst + e, t.v[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// All threads of a block, after wave 0 (which holds thread 0) published the block's lists with
// publish_list().  Returns true in every thread of the block that arrived last at `counter` (and re-arms
// the counter for the next launch); that block may then read every list with device-scope loads.
__device__ __forceinline__ bool tail_last_block(unsigned* counter, unsigned nblocks, unsigned* lds_flag) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = atomicAdd(counter, 1u);
        const bool last = ticket == nblocks - 1;
        if (last) atomicExch(counter, 0u);
        *lds_flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *reinterpret_cast<volatile unsigned*>(lds_flag) != 0;
    __syncthreads();  // the flag word is free again
    return last;
}

}  // namespace ssmi
