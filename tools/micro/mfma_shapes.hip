// Micro-benchmark: sustained rate of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 from registers only
// (no memory traffic), 2 waves per SIMD on every CU, to see what the chip sustains under pure matrix load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

template <int SHAPE>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    half8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)(threadIdx.x * 0.001f + i); b[i][j] = (_Float16)(j * 0.01f + i); }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4_t acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0];
    } else {
        f32x16_t acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 200000;
    for (int rep = 0; rep < 3; ++rep)
        for (int shape : {16, 32}) {
            hipEventRecord(e0);
            if (shape == 16) hipLaunchKernelGGL(k<16>, dim3(256), dim3(512), 0, 0, out, iters);
            else hipLaunchKernelGGL(k<32>, dim3(256), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per iteration and wave: 16 x 16384 flop (16x16x32) or 8 x 32768 flop (32x32x16) = 262144 flop
            const double flops = 256.0 * 8 * iters * 262144.0;
            printf("shape %dx: %.2f ms  %.1f TFLOP/s\n", shape, ms, flops / ms / 1e9);
        }
    return 0;
}
