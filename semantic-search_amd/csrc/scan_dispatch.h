// scan_dispatch.h - template dispatch for K1 (included once per storage type).
#pragma once
#include "scan_impl.h"
#include "scan_mm.h"
#include "kernels.h"

namespace ssmi {

template <typename T, int NCH, int BT, int R, bool QREG, bool NT>
static hipError_t launch_one(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p,
                             hipStream_t s) {
    auto kern = flat_scan_topk_kernel<T, NCH, BT, R, QREG, NT>;
    if (p.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid_x, p.grid_y), dim3(p.threads), p.lds, s, X, Q, nq, k, partial, p.tail);
    return hipGetLastError();
}

template <typename T, int NCH, int BT, bool QREG>
static hipError_t launch_r(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p,
                           hipStream_t s) {
    switch (p.r) {
        case 1: return launch_one<T, NCH, BT, 1, QREG, true>(X, Q, nq, k, partial, p, s);
        case 2: if (BT <= 4) return launch_one<T, NCH, (BT <= 4 ? BT : 1), 2, QREG, true>(X, Q, nq, k, partial, p, s); break;
        case 4: if (BT <= 2) return launch_one<T, NCH, (BT <= 2 ? BT : 1), 4, QREG, true>(X, Q, nq, k, partial, p, s); break;
        case 8: if (BT <= 1) return launch_one<T, NCH, (BT <= 1 ? BT : 1), 8, QREG, true>(X, Q, nq, k, partial, p, s); break;
        case 16: if (BT <= 1) return launch_one<T, NCH, (BT <= 1 ? BT : 1), 16, QREG, true>(X, Q, nq, k, partial, p, s); break;
    }
    return hipErrorInvalidValue;
}

template <typename T, int NCH>
static hipError_t launch_bt(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p,
                            hipStream_t s) {
    switch (p.bt) {
        case 1: return launch_r<T, NCH, 1, false>(X, Q, nq, k, partial, p, s);
        case 2: return launch_r<T, NCH, 2, false>(X, Q, nq, k, partial, p, s);
        case 4: return launch_r<T, NCH, 4, false>(X, Q, nq, k, partial, p, s);
        case 8: return launch_r<T, NCH, 8, false>(X, Q, nq, k, partial, p, s);
    }
    return hipErrorInvalidValue;
}

template <typename T, int BT, int R, int G>
static hipError_t launch_mm_g(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s) {
    auto kern = flat_scan_mm_kernel<T, BT, R, G>;
    if (p.threads != (G <= 8 ? 1024 : 512) || (X.ld % (16 * ElemTraits<T>::V * 2 * G)) != 0 || X.seg_shift < 2) return hipErrorInvalidValue;
    if (p.lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(p.grid_x, p.grid_y), dim3(p.threads), p.lds, s, X, Q, nq, k, partial, p.tail);
    return hipGetLastError();
}

template <typename T, int BT, int R>
static hipError_t launch_mm(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p, hipStream_t s) {
    switch (p.mm_g) {
        case 1: return launch_mm_g<T, BT, R, 1>(X, Q, nq, k, partial, p, s);
        case 2: return launch_mm_g<T, BT, R, 2>(X, Q, nq, k, partial, p, s);
        case 4: return launch_mm_g<T, BT, R, 4>(X, Q, nq, k, partial, p, s);
        case 8: return launch_mm_g<T, BT, R, 8>(X, Q, nq, k, partial, p, s);
        case 16: return launch_mm_g<T, BT, R, 16>(X, Q, nq, k, partial, p, s);
    }
    return hipErrorInvalidValue;
}

template <typename T>
static hipError_t launch_flat_scan_t(const RowView& X, const float* Q, int nq, int k, skey_t* partial,
                                     const ScanPlan& p, hipStream_t s) {
    constexpr int V = ElemTraits<T>::V;
    constexpr int NF = 4096 / (64 * V);  // chunks of a 4096-d row
    if (p.mm) {
        if (p.bt == 8 && p.r == 1) return launch_mm<T, 8, 1>(X, Q, nq, k, partial, p, s);
        if (p.bt == 4 && p.r == 1) return launch_mm<T, 4, 1>(X, Q, nq, k, partial, p, s);
        if (p.bt == 4 && p.r == 2) return launch_mm<T, 4, 2>(X, Q, nq, k, partial, p, s);
        return hipErrorInvalidValue;
    }
    if (p.qreg) {
        if (p.bt != 1) return hipErrorInvalidValue;
        if (p.nch == NF) {
            if (p.r == 1 && !p.nt) return launch_one<T, NF, 1, 1, true, false>(X, Q, nq, k, partial, p, s);
            return launch_r<T, NF, 1, true>(X, Q, nq, k, partial, p, s);
        }
        if (p.nch == NF / 2) return launch_r<T, NF / 2, 1, true>(X, Q, nq, k, partial, p, s);
        if (p.nch == NF / 4) return launch_r<T, NF / 4, 1, true>(X, Q, nq, k, partial, p, s);
        return hipErrorInvalidValue;
    }
    return launch_bt<T, 0>(X, Q, nq, k, partial, p, s);
}

}  // namespace ssmi
