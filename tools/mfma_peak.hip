// Attainable dense bf16 MFMA rate on this device: a bare loop of v_mfma_f32_16x16x32_bf16 on register-resident
// operands (no memory traffic), 2 waves per SIMD on every CU, with random and with all-zero operands.  The
// chip lowers its clock under MFMA load on random data, so this -- not the nominal 2.5 PFLOP/s -- is what a
// perfect GEMM inner loop could reach.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak && ./tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_loop(const uint4 *__restrict__ in, float *__restrict__ out, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, in[(gid * 8 + i) & 0xffff]);
        b[i] = __builtin_bit_cast(bf16x8, in[(gid * 8 + 4 + i) & 0xffff]);
    }
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[gid] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

// the same with v_mfma_f32_32x32x16_bf16: twice the MACs per operand register read
__global__ __launch_bounds__(512) void mfma_loop32(const uint4 *__restrict__ in, float *__restrict__ out, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, in[(gid * 8 + i) & 0xffff]);
        b[i] = __builtin_bit_cast(bf16x8, in[(gid * 8 + 4 + i) & 0xffff]);
    }
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i * 2 + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) s += acc[i][k];
    out[gid] = s;
}

int main() {
    const int blocks = 256, threads = 512, iters = 20000;
    uint4 *in;
    float *out;
    hipMalloc(&in, 65536 * sizeof(uint4));
    hipMalloc(&out, (size_t)blocks * threads * sizeof(float));
    std::vector<uint16_t> h(65536 * 8);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        for (auto &v : h) {
            // random bf16 in (-2, 2): random sign/mantissa, exponent 2^-3 .. 2^0
            const uint16_t r = (uint16_t)rand();
            v = mode == 0 ? (uint16_t)((r & 0x807f) | ((124 + (rand() & 3)) << 7)) : (uint16_t)0;
        }
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        // ~2 s of back-to-back launches first, so that the clock has settled under load
        for (int w = 0; w < 60; ++w) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)reps * blocks * (threads / 64) * iters * 16.0 * 16384.0;
        printf("16x16x32, %s operands: %.1f TFLOP/s (%.2f ms per launch)\n", mode == 0 ? "random" : "zero  ", flops / (ms * 1e-3) / 1e12,
               ms / reps);
        for (int w = 0; w < 30; ++w) hipLaunchKernelGGL(mfma_loop32, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop32, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        const double flops32 = (double)reps * blocks * (threads / 64) * iters * 4.0 * 32768.0;
        printf("32x32x16, %s operands: %.1f TFLOP/s (%.2f ms per launch)\n", mode == 0 ? "random" : "zero  ", flops32 / (ms * 1e-3) / 1e12,
               ms / reps);
    }
    return 0;
}
