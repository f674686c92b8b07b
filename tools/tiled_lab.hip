// Diagnostic harness (not part of the library): times ablated builds of the tiled kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vllm_pq_amd/csrc tools/tiled_lab.hip -o tools/tiled_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "pq_gemm_tiled_impl.cuh"

using namespace pq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int WAVES_T, int WN, int WT, int KT, int ABL>
float run(const char *label, int T, int N, int G, int nparts) {
    const int K = 256, D = 8, M = 1;
    constexpr int BN = (TL_WAVES / WAVES_T) * WN * 16, BT = WAVES_T * WT * 16;
    Parts parts;
    parts.num = nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) parts.size[p] = p < nparts ? N / nparts : 0;
    const int nchunks = (G + 63) / 64;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    void *pk, *x, *cb, *sc, *y;
    CK(hipMalloc(&pk, pk_bytes)); CK(hipMemset(pk, 0x5a, pk_bytes));
    CK(hipMalloc(&x, (size_t)T * G * D * 2)); CK(hipMemset(x, 0x3c, (size_t)T * G * D * 2));
    CK(hipMalloc(&cb, (size_t)nparts * K * D * 2)); CK(hipMemset(cb, 0x3c, (size_t)nparts * K * D * 2));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    // random-ish data so that DVFS sees realistic operands
    {
        unsigned *h = (unsigned *)malloc((size_t)T * G * D * 2);
        unsigned s = 12345;
        for (size_t i = 0; i < (size_t)T * G * D / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s & 0x807f807fu) | 0x3f003f00u; }
        CK(hipMemcpy(x, h, (size_t)T * G * D * 2, hipMemcpyHostToDevice));
        for (size_t i = 0; i < (size_t)nparts * K * D / 2; ++i) { s = s * 1664525u + 1013904223u; h[i] = (s & 0x807f807fu) | 0x3f003f00u; }
        CK(hipMemcpy(cb, h, (size_t)nparts * K * D * 2, hipMemcpyHostToDevice));
        unsigned *hc = (unsigned *)malloc(pk_bytes);
        for (size_t i = 0; i < pk_bytes / 4; ++i) { s = s * 1664525u + 1013904223u; hc[i] = s; }
        CK(hipMemcpy(pk, hc, pk_bytes, hipMemcpyHostToDevice));
        free(h); free(hc);
    }
    const int R = stream_replication<D>(K, M);
    const int n_tiles = tiles_in(parts, BN), t_tiles = (T + BT - 1) / BT;
    const int per_xcd = (n_tiles * t_tiles + 7) / 8;
    const size_t lds = (size_t)ST_MAX_LDS_CB + 3 * (size_t)KT * BT * TL_BK * 2;
    auto kern = gemm_tiled_kernel<bf16_t, int8_t, D, M, WAVES_T, WN, WT, KT, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(per_xcd * 8), dim3(TL_THREADS), lds, 0, (const uint16_t *)x, T, (int64_t)G * D,
                           (const u32x4 *)pk, (const uint16_t *)cb, (const uint16_t *)sc, (const uint16_t *)nullptr,
                           make_tile_map(parts, BN), G, K, 8, R, (uint16_t *)y, (int64_t)N, n_tiles, t_tiles);
    };
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    const int nrep = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < nrep; ++i) launch();
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / nrep;
    printf("%-36s tile %dx%d T=%5d N=%5d Kin=%5d  %9.2f us  %8.1f TFLOP/s\n", label, BN, BT, T, N, G * D, us,
           2.0 * T * N * G * D / us * 1e-6);
    CK(hipFree(pk)); CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y));
    return us;
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 4096;
    printf("== gate_up 28672 x 4096, 512x128 tile ==\n");
    run<2, 8, 4, 2, 0>("full", T, 28672, 512, 2);
    run<2, 8, 4, 2, 1>("no wait / barrier", T, 28672, 512, 2);
    run<2, 8, 4, 2, 2>("no x DMA", T, 28672, 512, 2);
    run<2, 8, 4, 2, 3>("no x DMA, wait, barrier", T, 28672, 512, 2);
    run<2, 8, 4, 2, 4>("no epilogue stores", T, 28672, 512, 2);
    run<2, 8, 4, 2, 8>("no gathers", T, 28672, 512, 2);
    run<2, 8, 4, 2, 16>("no x fragment reads", T, 28672, 512, 2);
    run<2, 8, 4, 2, 24>("no gathers, no x fragment reads", T, 28672, 512, 2);
    run<2, 8, 4, 2, 15>("mfma + x frag reads only", T, 28672, 512, 2);
    run<2, 8, 4, 2, 31>("mfma only", T, 28672, 512, 2);
    return 0;
}
