// Calibration of the 8 TB/s roof (like tools/gemm_ceiling.py for the matrix roof): what a kernel that does nothing but
// read 16 GiB of HBM and add it up reaches on this box, in the forms K1 could take.  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o tools/micro/hbm_read tools/micro/hbm_read.hip && tools/micro/hbm_read
// Variants: threads per block, blocks per CU, 16-byte loads in flight per thread, the walk (one moving front over the
// buffer / one contiguous slice per block), plain or nontemporal loads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool SLICE>
__global__ void read_kernel(const f4* __restrict__ p, size_t n16, float* out) {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    f4 acc[U];
    for (int u = 0; u < U; ++u) acc[u] = f4{0, 0, 0, 0};
    if (SLICE) {
        // one contiguous slice per block, the block's threads side by side inside it
        const size_t per_block = n16 / gridDim.x;
        const f4* b = p + (size_t)blockIdx.x * per_block;
        for (size_t i = threadIdx.x; i + (size_t)(U - 1) * blockDim.x < per_block; i += (size_t)U * blockDim.x) {
            f4 v[U];
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(b + i + (size_t)u * blockDim.x) : b[i + (size_t)u * blockDim.x];
            for (int u = 0; u < U; ++u) acc[u] += v[u];
        }
    } else {
        // one front: the whole grid side by side, U grid-widths per step
        for (size_t i = tid; i + (size_t)(U - 1) * nthreads < n16; i += (size_t)U * nthreads) {
            f4 v[U];
            for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + (size_t)u * nthreads) : p[i + (size_t)u * nthreads];
            for (int u = 0; u < U; ++u) acc[u] += v[u];
        }
    }
    f4 s = acc[0];
    for (int u = 1; u < U; ++u) s += acc[u];
    const float r = s.x + s.y + s.z + s.w;
    if (r == 123.456f) out[tid % 64] = r;   // never true for the data below: keeps the loads alive
}

template <int U, bool NT, bool SLICE>
static double run(const f4* p, size_t n16, float* out, int blocks, int threads, int iters) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<float> ms;
    for (int it = 0; it < iters + 2; ++it) {
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL((read_kernel<U, NT, SLICE>), dim3(blocks), dim3(threads), 0, 0, p, n16, out);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float t; CK(hipEventElapsedTime(&t, a, b));
        if (it >= 2) ms.push_back(t);
    }
    CK(hipGetLastError());
    std::sort(ms.begin(), ms.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return ms[ms.size() / 2];
}

__global__ void fill_kernel(f4* p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const float v = (float)(i & 1023) * 1e-3f;
        p[i] = f4{v, -v, v * 0.5f, 1.0f};
    }
}

int main(int argc, char** argv) {
    const size_t bytes = (argc > 1 ? atoll(argv[1]) : 16384000000ll) / 16 * 16;   // the bytes of 1M x 4096 fp32 rows
    const size_t n16 = bytes / 16;
    f4* p; float* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 256));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, p, n16);
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s, %d CUs, %.3f GB per launch, median of 9 launches\n", prop.name, cus, bytes / 1e9);
    printf("%-10s %-8s %-8s %-4s %-6s %-4s %10s %10s\n", "walk", "threads", "blk/CU", "U", "loads", "", "ms", "TB/s");
#define ROW(U, NT, SLICE, T, BPC) do { \
        const double ms = run<U, NT, SLICE>(p, n16, out, cus * (BPC), (T), 9); \
        printf("%-10s %-8d %-8d %-4d %-6s %-4s %10.3f %10.3f\n", SLICE ? "slice" : "front", (T), (BPC), U, NT ? "nt" : "plain", "", ms, bytes / ms / 1e9); \
        fflush(stdout); } while (0)
    ROW(4, false, false, 256, 4); ROW(4, false, false, 256, 8); ROW(4, false, false, 512, 2); ROW(4, false, false, 512, 4);
    ROW(4, false, false, 1024, 1); ROW(4, false, false, 1024, 2);
    ROW(8, false, false, 256, 4); ROW(8, false, false, 256, 8); ROW(8, false, false, 512, 2); ROW(8, false, false, 512, 4); ROW(8, false, false, 1024, 1); ROW(8, false, false, 1024, 2);
    ROW(16, false, false, 256, 2); ROW(16, false, false, 256, 4); ROW(16, false, false, 512, 1); ROW(16, false, false, 512, 2); ROW(16, false, false, 768, 1);
    ROW(8, true, false, 256, 4); ROW(8, true, false, 512, 2); ROW(8, true, false, 1024, 1); ROW(16, true, false, 512, 1); ROW(16, true, false, 768, 1);
    ROW(8, false, true, 256, 4); ROW(8, false, true, 512, 2); ROW(8, false, true, 1024, 1); ROW(16, false, true, 512, 1); ROW(16, false, true, 768, 1);
    ROW(8, true, true, 512, 2); ROW(16, true, true, 768, 1);
    CK(hipFree(p)); CK(hipFree(out));
    return 0;
}
