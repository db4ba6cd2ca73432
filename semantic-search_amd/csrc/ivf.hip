// ivf.hip - IVF-Flat (placeholder until the list kernels land in this round).
#include "index.h"
#include "kernels.h"
namespace ssmi {
int ivf_create(ssmi_index*, int) { return fail("IVF-Flat is not built yet"); }
void ivf_destroy(ssmi_index*) {}
int ivf_train(ssmi_index*, int64_t, const float*, int) { return fail("IVF-Flat is not built yet"); }
int ivf_add(ssmi_index*, int64_t, const void*, int, int) { return fail("IVF-Flat is not built yet"); }
int ivf_add_random_unit(ssmi_index*, int64_t, uint64_t) { return fail("IVF-Flat is not built yet"); }
int ivf_search(ssmi_index*, int64_t, const float*, int, float*, int64_t*, int64_t, hipStream_t) { return fail("IVF-Flat is not built yet"); }
int ivf_ids_to_slots(ssmi_index*, int64_t, const int64_t*, std::vector<int64_t>*) { return fail("IVF-Flat is not built yet"); }
}  // namespace ssmi
extern "C" {
int ssmi_index_create_ivfflat(int, int, int, int, ssmi_index** out) { if (out) *out = nullptr; return ssmi::fail("IVF-Flat is not built yet"); }
int ssmi_index_set_centroids(ssmi_index*, const float*, int) { return ssmi::fail("IVF-Flat is not built yet"); }
int ssmi_index_get_centroids(const ssmi_index*, float*) { return ssmi::fail("IVF-Flat is not built yet"); }
int ssmi_index_make_direct_map(ssmi_index*) { return ssmi::fail("IVF-Flat is not built yet"); }
int ssmi_index_get_list_sizes(const ssmi_index*, int64_t*) { return ssmi::fail("IVF-Flat is not built yet"); }
int ssmi_index_write(ssmi_index*, const char*) { return ssmi::fail("persistence is not built yet"); }
int ssmi_index_read(const char*, int, int, ssmi_index** out) { if (out) *out = nullptr; return ssmi::fail("persistence is not built yet"); }
}
