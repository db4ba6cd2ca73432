// scan_bf16.hip - K1 instantiations for bf16 row storage.
#include "scan_dispatch.h"
namespace ssmi {
hipError_t launch_flat_scan_bf16(const RowView& X, const float* Q, int nq, int k, skey_t* partial, const ScanPlan& p,
                                 hipStream_t s) {
    return launch_flat_scan_t<bf16_t>(X, Q, nq, k, partial, p, s);
}
}  // namespace ssmi
