// mfma.hip - K2: large-batch Flat search as an MFMA fp16 contraction with a fused
// top-k epilogue (the score matrix is never written).
//
// Stands behind faiss.IndexFlatIP.search for nq >> 1 (the reference only ever
// issues batch-1 searches, core/index_manager.py:371; batched search is the
// additive surface of SURVEY.md 8b).
//
//   S[row][query] = sum_k X[row][k] * Q[query][k]      fp16 x fp16 -> fp32
//
// Workgroup = 8 waves (2 x 4), tile = 256 database rows x 256 queries; the kernel
// itself (K walked in phases of 32 through a four-slab LDS ring, two wave groups
// alternating LDS reads and MFMAs) is in mfma_pp.hip; this file holds the plan,
// the query conversion and the launch.  Each wave owns a 128 x 64 sub-tile = 8 x 4
// accumulators of 16 x 16 (v_mfma_f32_16x16x32_f16).
//
// A workgroup keeps its query tile and walks a slice of the database (persistent
// along N).  C/D layout puts the query on the lane (col = lane & 15) and four
// rows in the registers, so the epilogue is one compare per accumulator element
// against that query's running threshold (LDS); survivors are appended as 64-bit
// keys to a per-(workgroup, query) candidate buffer.  The buffers are folded
// into the queries' sorted k-lists (topk.h) - which raises the thresholds -
// after row tiles 1, 4, 16, 64, ... of the slice (so the threshold is never staler
// than a factor four in rows seen and a period adds about 3k candidates per
// query) and whenever a buffer is half full.  At the end of the slice the k-list
// IS the per-slice partial result that K3 (merge.hip) merges across slices.
// After every fold the slices of a query tile exchange thresholds through `gthr`
// (one atomic max per query): the k-th score of any slice bounds the final k-th
// score from below, so a slice may filter with the highest one reached anywhere;
// its own list then holds fewer than k keys, which K3 takes (key 0 = empty).
#include "mfma_common.h"
#include <cstdlib>
#include <algorithm>

namespace ssmi {

// pitch == ld: plain rows.  pitch == ld + 64 (the 256-query kernel, mfma_pp.hip): a row is [32 halves of zeros][ld
// values][the first 32 values again], so that the 128-byte pieces of the query stream that start at an ODD 32-deep
// phase are whole cache lines and the piece behind the last phase holds the first phase of the next row tile.
// H: _Float16 or bf16_t (round to nearest even either way) - the operand type of the rows the batch will meet.
template <typename H>
__global__ void cvt_queries_16_kernel(const float* __restrict__ q, int nq, int ld, H* __restrict__ out,
                                      unsigned* __restrict__ gthr, unsigned* __restrict__ gtop, int gtop_words, int pitch,
                                      unsigned* __restrict__ xsync = nullptr) {
    const int i = blockIdx.x;  // padded row
    if (gthr && threadIdx.x == 0) gthr[i] = 0;
    if (xsync && i == 0 && threadIdx.x < 8 * XSYNC_STRIDE) xsync[threadIdx.x] = 0;
    if (gtop)
        for (int e = threadIdx.x; e < gtop_words; e += blockDim.x) gtop[(size_t)i * gtop_words + e] = 0;
    const int off = (pitch - ld) / 2;
    for (int e = threadIdx.x; e < ld; e += blockDim.x)
        out[(size_t)i * pitch + off + e] = (i < nq) ? (H)q[(size_t)i * ld + e] : (H)0.f;
    if (off) {
        for (int e = threadIdx.x; e < off; e += blockDim.x) {
            out[(size_t)i * pitch + e] = (H)0.f;
            out[(size_t)i * pitch + off + ld + e] = (i < nq) ? (H)q[(size_t)i * ld + e] : (H)0.f;
        }
    }
}

// (nq, ld) fp32 -> (rows_out, ld) fp16 / bf16, zero rows beyond nq
hipError_t launch_cvt_queries_16(const float* q, int nq, int rows_out, int ld, void* out, bool bf16, hipStream_t s) {
    if (bf16) hipLaunchKernelGGL(cvt_queries_16_kernel<bf16_t>, dim3(rows_out), dim3(256), 0, s, q, nq, ld, (bf16_t*)out, (unsigned*)nullptr,
                                 (unsigned*)nullptr, 0, ld);
    else hipLaunchKernelGGL(cvt_queries_16_kernel<_Float16>, dim3(rows_out), dim3(256), 0, s, q, nq, ld, (_Float16*)out, (unsigned*)nullptr,
                            (unsigned*)nullptr, 0, ld);
    return hipGetLastError();
}

bool plan_flat_mfma(int dtype, int ld, long long nrows, int nq, int k, int num_cu, MfmaPlan* p) {
    if (!dtype_is16(dtype) || (ld % BK) != 0 || nrows < 1 || nq < 1 || k < 1 || k > kMaxK) return false;
    p->dtype = dtype;
    p->nqt = (nq + TN - 1) / TN;
    const long long row_tiles = (nrows + TM - 1) / TM;
    long long ns = num_cu / p->nqt;
    if (ns < 1) ns = 1;
    if (ns > row_tiles) ns = row_tiles;
    p->rows_per_slice = (row_tiles + ns - 1) / ns * TM;
    p->nslices = (int)((nrows + p->rows_per_slice - 1) / p->rows_per_slice);
    p->grid = p->nqt * p->nslices;
    // scores every slice publishes per query (gtop): enough of them that the k-th largest exists and is tight -
    // about max(64, 1.25 k) entries per query over the slices, at most 8 per slice
    int m = (int)((std::max<long long>(64, (long long)k * 5 / 4) + p->nslices - 1) / p->nslices);
    m = std::max(1, std::min(std::min(m, 8), k));
    p->gtop_m = (p->nslices > 1 && (long long)p->nslices * m >= k) ? m : 0;
    const char* env_n = getenv("SSMI_MFMA_NARROW");
    p->narrow = nq <= N128_TN && !(env_n && env_n[0] == '0');
    // fp16 queries (the 256-query kernel takes rows of ld + 64: see cvt_queries_f16_kernel) + shared thresholds + published tops
    p->q_pitch = p->narrow ? ld : ld + 64;
    p->qh_bytes = (size_t)p->nqt * TN * p->q_pitch * 2 + (size_t)p->nqt * TN * sizeof(unsigned) +
                  (size_t)p->nqt * TN * p->nslices * p->gtop_m * sizeof(unsigned) + 8 * XSYNC_STRIDE * sizeof(unsigned);
    p->clk_off = 0;
    if (!p->narrow) { p->qh_bytes = (p->qh_bytes + 7) / 8 * 8; p->clk_off = p->qh_bytes; p->qh_bytes += 4 * sizeof(unsigned long long); }
    p->cand_bytes = (size_t)p->grid * TN * CAND_CAP * sizeof(skey_t);
    const char* env_r = getenv("SSMI_MFMA_R128");
    p->kernel = !p->narrow ? "flat_mfma_pp_kernel"
                : (mfma_r128_supported(ld, k) && !(env_r && env_r[0] == '0')) ? "flat_mfma_r128_kernel" : "flat_mfma_n128_kernel";
    p->lds = p->narrow ? (size_t)N128_RING * N128_SLOT + 3 * N128_TN * 4 + 16 : (size_t)4 * TILE_BYTES + EPI_LDS_BYTES;
    return true;
}

hipError_t launch_flat_mfma(const RowView& X, const float* Q, int nq, int k, const MfmaPlan& p, void* qh, void* cand,
                            skey_t* partial, hipStream_t s, unsigned min_ord) {
    unsigned* gthr = reinterpret_cast<unsigned*>(static_cast<unsigned char*>(qh) + (size_t)p.nqt * TN * p.q_pitch * 2);
    const char* env_t = getenv("SSMI_MFMA_SHARE_THR");
    const bool share_thr = !(env_t && env_t[0] == '0');
    const char* env_g = getenv("SSMI_MFMA_SHARE_TOPS");
    // (a batch of a few queries on the register-ring kernel gains nothing and pays for the ring drains of the exchange)
    const bool share_tops = share_thr && p.gtop_m > 0 && k > 1 && !(p.narrow && nq < 32 && k <= 16) && !(env_g && env_g[0] == '0');
    unsigned* gtop = gthr + (size_t)p.nqt * TN;
    unsigned* xsync = gtop + (size_t)p.nqt * TN * p.nslices * p.gtop_m;
    if (p.dtype == 2)
        hipLaunchKernelGGL(cvt_queries_16_kernel<bf16_t>, dim3(p.nqt * TN), dim3(256), 0, s, Q, nq, X.ld, (bf16_t*)qh, gthr,
                           share_tops ? gtop : (unsigned*)nullptr, p.nslices * p.gtop_m, p.q_pitch, xsync);
    else
        hipLaunchKernelGGL(cvt_queries_16_kernel<_Float16>, dim3(p.nqt * TN), dim3(256), 0, s, Q, nq, X.ld, (_Float16*)qh, gthr,
                           share_tops ? gtop : (unsigned*)nullptr, p.nslices * p.gtop_m, p.q_pitch, xsync);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    MfmaParams P = {};
    P.bf16 = p.dtype == 2;
    P.X = X; P.Qh = (const _Float16*)qh; P.q_pitch = p.q_pitch; P.nq = nq; P.k = k; P.nqt = p.nqt; P.nslices = p.nslices;
    P.rows_per_slice = p.rows_per_slice; P.partial = partial; P.cand = (skey_t*)cand;
    P.gthr = (share_thr && p.nslices > 1) ? gthr : nullptr;
    P.gtop = share_tops ? gtop : nullptr;
    P.gtop_m = p.gtop_m;
    P.thr_floor = min_ord ? ord_to_f32(min_ord) : SSMI_NEG_MAX;
    P.clk = p.clk_off ? reinterpret_cast<unsigned long long*>(static_cast<unsigned char*>(qh) + p.clk_off) : nullptr;
    { const char* e = getenv("SSMI_MFMA_SEED_FIRST"); P.seed_first = !(e && e[0] == '0'); }
    // scheduled folds after row tiles 1, 4, 16, 64, ... of a slice; with the cross-slice bound (which matures much faster
    // than a slice's own k-th score) after 1, 16, 256, ... only (SSMI_MFMA_FOLD_MASK: bit b = a fold after tile 2^b)
    P.fold_mask = P.gtop ? 0x11111111u : 0x55555555u;
    if (const char* e = getenv("SSMI_MFMA_FOLD_MASK")) P.fold_mask = (unsigned)strtoul(e, nullptr, 16) | 1u;
    // The workgroups of an XCD meet behind every row tile (mfma_pp.hip), so that the operand slabs they share are L2 hits and
    // not fabric reads: the fabric-side traffic costs this power-limited kernel its clock.  Worth it on long slices only (a
    // meeting costs the spread of the workgroups' tile times, the heavy folds come early): from 128 row tiles per slice on,
    // from row tile 17 of a slice on.  With meetings an XCD takes a patch of 8 query tiles x 4 slices instead of all 16 query
    // tiles of 2 slices (12 instead of 18 operand streams from beyond its L2).  SSMI_MFMA_SYNC=0/1 and SSMI_MFMA_PATCH=<query
    // tiles per patch, 0 = none> override; SSMI_MFMA_SYNC_FROM / SSMI_MFMA_SYNC_POLLS tune the meeting.
    const bool can_meet = !p.narrow && k > 1 && (p.grid % 8) == 0 && p.nslices > 1;
    bool meet_on = can_meet && p.rows_per_slice / TM >= 128;
    if (const char* e = getenv("SSMI_MFMA_SYNC")) meet_on = can_meet && e[0] == '1';
    P.xsync = meet_on ? xsync : nullptr;
    {
        const char* e = getenv("SSMI_MFMA_FOLD16");
        P.fold16 = (k <= 16 && !(e && e[0] == '0')) ? 1 : 0;
    }
    P.xsync_from = 17;
    if (const char* e = getenv("SSMI_MFMA_SYNC_FROM")) P.xsync_from = atoi(e);
    P.xsync_polls = 2048;   // about 2 ms: longer than any fold round
    if (const char* e = getenv("SSMI_MFMA_SYNC_POLLS")) P.xsync_polls = atoi(e);
    int patch = (meet_on && p.nqt >= 16 && p.nqt % 8 == 0) ? 8 : 0;
    if (const char* e = getenv("SSMI_MFMA_PATCH")) patch = atoi(e);
    P.patch_a = (patch > 0 && patch < p.nqt && p.nqt % patch == 0 && !p.narrow) ? patch : 0;
    int r = 1;
    while (64 * r < k) r <<= 1;
    if (p.narrow) {
        const char* env_r = getenv("SSMI_MFMA_R128");
        if (mfma_r128_supported(X.ld, k) && !(env_r && env_r[0] == '0')) return launch_flat_mfma_r128(P, r, p.grid, s);
        return launch_flat_mfma_n128(P, r, p.grid, p.lds, s);
    }
    return launch_flat_mfma_pp(P, r, p.grid, p.lds, s);
}

}  // namespace ssmi
