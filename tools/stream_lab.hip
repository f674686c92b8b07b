// Diagnostic harness (not part of the library): times ablated builds of the streaming kernel on one
// shape with hipEvents, rotating over weight copies so the codes come from HBM.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vllm_pq_amd/csrc tools/stream_lab.hip -o /tmp/stream_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "pq_gemm_stream_impl.cuh"

using namespace pq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NF, int ABL>
float run(const char *label, int T, int N, int G, int nparts, int copies, int nwarm, int nrep) {
    const int K = 256, D = 8, M = 1;
    Parts parts;
    parts.num = nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) parts.size[p] = p < nparts ? N / nparts : 0;
    const int nchunks = (G + 63) / 64;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    std::vector<void *> pk(copies);
    for (int i = 0; i < copies; ++i) { CK(hipMalloc(&pk[i], pk_bytes)); CK(hipMemset(pk[i], i + 1, pk_bytes)); }
    void *x, *cb, *sc, *y;
    CK(hipMalloc(&x, (size_t)T * G * D * 2)); CK(hipMemset(x, 0x3c, (size_t)T * G * D * 2));
    CK(hipMalloc(&cb, (size_t)nparts * K * D * 2)); CK(hipMemset(cb, 0x3c, (size_t)nparts * K * D * 2));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    const int R = stream_replication<D>(K, M);
    size_t lds = (size_t)M * K * 2 * D * R;
    const size_t red = (size_t)ST_WAVES * NF * 64 * 16;
    if (red > lds) lds = red;
    dim3 grid(tiles_in(parts, NF * 16), (T + 15) / 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int i) {
        hipLaunchKernelGGL((gemm_stream_kernel<bf16_t, int8_t, D, M, NF, 1, true, ABL>), grid, dim3(ST_THREADS), lds, 0,
                           (const uint16_t *)x, T, (int64_t)G * D, (const void *)pk[i % copies], (const uint16_t *)cb,
                           (const uint16_t *)sc, (const uint16_t *)nullptr, make_tile_map(parts, NF * 16), G, K, 8, R, (uint16_t *)y, (int64_t)N, 1, nullptr, nullptr);
    };
    for (int i = 0; i < nwarm; ++i) launch(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < nrep; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / nrep;
    printf("%-34s NF=%d T=%2d N=%5d G=%4d blocks=%4d  %7.2f us/launch  %7.1f GB/s(codes)\n", label, NF, T, N, G, grid.x,
           us, (double)N * G / us * 1e-3);
    for (auto p : pk) CK(hipFree(p));
    CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y));
    return us;
}

template <int NF>
void sweep(int T, int N, int G, int nparts, int copies) {
    run<NF, 0>("full", T, N, G, nparts, copies, 10, 200);
    run<NF, 1>("no gather", T, N, G, nparts, copies, 10, 200);
    run<NF, 2>("no mfma", T, N, G, nparts, copies, 10, 200);
    run<NF, 3>("no gather, no mfma", T, N, G, nparts, copies, 10, 200);
    run<NF, 4>("no code loads", T, N, G, nparts, copies, 10, 200);
    run<NF, 8>("no x loads", T, N, G, nparts, copies, 10, 200);
    run<NF, 16 + 1>("no image, no gather", T, N, G, nparts, copies, 10, 200);
    run<NF, 32>("no epilogue", T, N, G, nparts, copies, 10, 200);
    run<NF, 4 + 8>("no global loads", T, N, G, nparts, copies, 10, 200);
    run<NF, 1 + 2 + 16 + 32>("loads only", T, N, G, nparts, copies, 10, 200);
    run<NF, 1 + 2 + 4 + 8 + 16 + 32>("empty", T, N, G, nparts, copies, 10, 200);
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 1;
    printf("== gate_up 28672 x 4096 ==\n");
    sweep<8>(T, 28672, 512, 2, 40);
    sweep<4>(T, 28672, 512, 2, 40);
    printf("== o_proj 4096 x 4096 ==\n");
    sweep<1>(T, 4096, 512, 1, 280);
    return 0;
}
