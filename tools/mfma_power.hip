// Diagnostic harness (not part of the library): does the ORDER in which a wave issues the MFMAs of its register tile
// change the clock the chip holds?  The chip is power-limited under MFMA load on random data (tools/mfma_peak.hip), and
// an MFMA whose A (or B) operand is the one the previous MFMA used toggles fewer operand lines.
// A wave owns an 8 x 4 tile of 16x16 fragments (the tiled kernel's 512 x 128 shape): 8 A registers, 4 B registers,
// 32 accumulators, register-resident, no memory traffic, 2 waves per SIMD on every CU.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_power.hip -o tools/bin/mfma_power
// Orders: 0 = pairs of A, B outer, A inner (the tiled kernel's group order: A changes every MFMA, B every 2nd)
//         1 = A outer, B inner (A changes every 4th, B every MFMA)
//         2 = serpentine (exactly one operand changes per MFMA)
//         3 = B outer, A inner (B changes every 8th, A every MFMA)
//         4 = pairs of A, serpentine inside the pair's 8 MFMAs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ORDER>
__global__ __launch_bounds__(512) void mfma_tile(const uint4 *__restrict__ in, float *__restrict__ out, int iters) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_bit_cast(bf16x8, in[(gid * 12 + i) & 0xffff]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = __builtin_bit_cast(bf16x8, in[(gid * 12 + 8 + i) & 0xffff]);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#define MF(i, j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    for (int it = 0; it < iters; ++it) {
        if constexpr (ORDER == 0) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int j = 0; j < 4; ++j) { MF(2 * pp, j) MF(2 * pp + 1, j) }
        } else if constexpr (ORDER == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { MF(i, j) }
        } else if constexpr (ORDER == 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) { MF(i, (i & 1) ? 3 - j : j) }
        } else if constexpr (ORDER == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) { MF(i, j) }
        } else {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                MF(2 * pp, 0) MF(2 * pp + 1, 0) MF(2 * pp + 1, 1) MF(2 * pp, 1)
                MF(2 * pp, 2) MF(2 * pp + 1, 2) MF(2 * pp + 1, 3) MF(2 * pp, 3)
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#undef MF
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    out[gid] = s;
}

template <int ORDER>
static void run(const uint4 *in, float *out, const char *what) {
    const int blocks = 256, threads = 512, iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(mfma_tile<ORDER>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_tile<ORDER>, dim3(blocks), dim3(threads), 0, 0, in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)reps * blocks * (threads / 64) * iters * 32.0 * 16384.0;
    printf("order %d, %s operands: %.1f TFLOP/s\n", ORDER, what, flops / (ms * 1e-3) / 1e12);
}

int main() {
    uint4 *in;
    float *out;
    hipMalloc(&in, 65536 * sizeof(uint4));
    hipMalloc(&out, (size_t)256 * 512 * sizeof(float));
    std::vector<uint16_t> h(65536 * 8);
    for (int mode = 0; mode < 3; ++mode) {
        srand(1);
        for (auto &v : h) {
            const uint16_t r = (uint16_t)rand();
            // 0: random sign/mantissa, exponent 2^-3 .. 2^0; 1: every bit random but the exponent's top (finite, |v| < 2); 2: zero
            v = mode == 0 ? (uint16_t)((r & 0x807f) | ((124 + (rand() & 3)) << 7))
              : mode == 1 ? (uint16_t)((r & 0x807f) | ((96 + (rand() & 31)) << 7)) : (uint16_t)0;
        }
        hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const char *what = mode == 0 ? "narrow-exponent" : mode == 1 ? "wide-exponent" : "zero";
        run<0>(in, out, what);
        run<1>(in, out, what);
        run<2>(in, out, what);
        run<3>(in, out, what);
        run<4>(in, out, what);
    }
    return 0;
}
