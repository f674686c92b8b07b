// Diagnostic harness: per-block phase stamps of the streaming kernel (wall_clock64, 100 MHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "pq_gemm_stream_impl.cuh"
using namespace pq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NF, int TF, bool ACT = false>
void run(const char *label, int T, int N, int G, int nparts) {
    const int K = 256, D = 8, M = 1;
    Parts parts; parts.num = nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) parts.size[p] = p < nparts ? N / nparts : 0;
    const int nchunks = (G + 63) / 64;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    const int copies = 6;
    std::vector<void *> pk(copies);
    for (int i = 0; i < copies; ++i) { CK(hipMalloc(&pk[i], pk_bytes)); CK(hipMemset(pk[i], i + 1, pk_bytes)); }
    void *x, *cb, *sc, *y, *st;
    CK(hipMalloc(&x, (size_t)T * G * D * 2)); CK(hipMemset(x, 0x3c, (size_t)T * G * D * 2));
    CK(hipMalloc(&cb, (size_t)nparts * K * D * 2)); CK(hipMemset(cb, 0x3c, (size_t)nparts * K * D * 2));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    const int R = stream_replication<D>(K, M);
    constexpr int RF = NF * TF < 8 ? NF * TF : 8;
    size_t lds = (size_t)M * K * 2 * D * R;
    if ((size_t)ST_WAVES * RF * 1024 > lds) lds = (size_t)ST_WAVES * RF * 1024;
    Parts gate = parts;
    if (ACT) { gate.num = 1; gate.size[1] = 0; lds = 2 * 65536; }
    dim3 grid(ACT ? tiles_in(gate, NF * 8) : tiles_in(parts, NF * 16), (T + 16 * TF - 1) / (16 * TF));
    auto kern = gemm_stream_kernel<bf16_t, int8_t, D, M, NF, TF, true, 64, ACT>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536));
    const int nblk = grid.x * grid.y;
    CK(hipMalloc(&st, (size_t)nblk * 64)); CK(hipMemset(st, 0, (size_t)nblk * 64));
    for (int i = 0; i < copies; ++i) {
        hipLaunchKernelGGL(kern, grid, dim3(ST_THREADS), lds, 0,
                           (const uint16_t *)x, T, (int64_t)G * D, (const void *)pk[i], (const uint16_t *)cb,
                           (const uint16_t *)sc, (const uint16_t *)st, ACT ? make_tile_map(gate, NF * 8) : make_tile_map(parts, NF * 16), G, K, 8, R, (uint16_t *)y, (int64_t)N, 1, nullptr, nullptr, ACT ? N / 2 : 0);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> h((size_t)nblk * 8);
    CK(hipMemcpy(h.data(), st, (size_t)nblk * 64, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 5]); }
    printf("%s NF=%d TF=%d T=%d N=%d Kin=%d blocks=%d: span first-start..last-end = %.2f us\n", label, NF, TF, T, N, G * D,
           nblk, (t1 - t0) / 100.0);
    const char *names[] = {"start(since first)", "loads issued", "image built", "wave0 mfma done", "all waves done", "stores retired",
                           "tile found", "codes+x requested"};
    for (int k : {0, 6, 7, 1, 2, 3, 4, 5}) {
        std::vector<double> v;
        for (int b = 0; b < nblk; ++b) v.push_back((h[b * 8 + k] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("   %-20s min %6.2f  med %6.2f  max %6.2f us\n", names[k], v[0], v[v.size() / 2], v.back());
    }
    for (auto p : pk) CK(hipFree(p));
    CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y)); CK(hipFree(st));
}
int main() {
    run<8, 1>("gate_up", 16, 28672, 512, 2);
    run<8, 1, true>("gate_up fused act", 16, 28672, 512, 2);
    run<8, 1>("gate_up", 1, 28672, 512, 2);
    run<8, 1, true>("gate_up fused act", 1, 28672, 512, 2);
    return 0;
}
