// Diagnostic harness (not part of the library): times ablated builds of the tiled kernel.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vllm_pq_amd/csrc tools/tiled_lab.hip -o tools/tiled_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "pq_gemm_tiled_impl.cuh"

using namespace pq;
static int g_slots = 32;   // blocks per XCD (32 CUs each); a large value gives one tile per block
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int WAVES_T, int WN, int WT, int KT, int ABL, int NWV = 8>
float run(const char *label, int T, int N, int G, int nparts) {
    const int K = 256, D = 8, M = 1;
    constexpr int BN = (NWV / WAVES_T) * WN * 16, BT = WAVES_T * WT * 16;
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
    const int per_xcd = (n_tiles * t_tiles + 7) / 8, nslots = per_xcd < g_slots ? per_xcd : g_slots;
    const size_t lds = (size_t)ST_MAX_LDS_CB + 3 * (size_t)KT * BT * TL_BK * 2;
    auto kern = gemm_tiled_kernel<bf16_t, int8_t, D, M, WAVES_T, WN, WT, KT, ABL, NWV>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&]() {
        hipLaunchKernelGGL(kern, dim3(nslots * 8), dim3(NWV * 64), lds, 0, (const uint16_t *)x, T, (int64_t)G * D,
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

// per-block phase stamps (wall_clock64, 100 MHz) of one launch
template <int WAVES_T, int WN, int WT, int KT, int ABLX = 0>
void stamps(int T, int N, int G, int nparts) {
    const int K = 256, D = 8, M = 1;
    constexpr int BN = (8 / WAVES_T) * WN * 16, BT = WAVES_T * WT * 16;
    Parts parts;
    parts.num = nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) parts.size[p] = p < nparts ? N / nparts : 0;
    const int nchunks = (G + 63) / 64;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    void *pk, *x, *cb, *sc, *y, *st;
    CK(hipMalloc(&pk, pk_bytes)); CK(hipMemset(pk, 0x5a, pk_bytes));
    CK(hipMalloc(&x, (size_t)T * G * D * 2)); CK(hipMemset(x, 0x3c, (size_t)T * G * D * 2));
    CK(hipMalloc(&cb, (size_t)nparts * K * D * 2)); CK(hipMemset(cb, 0x3c, (size_t)nparts * K * D * 2));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    const int R = stream_replication<D>(K, M);
    const int n_tiles = tiles_in(parts, BN), t_tiles = (T + BT - 1) / BT;
    const int per_xcd = (n_tiles * t_tiles + 7) / 8, nslots = per_xcd < g_slots ? per_xcd : g_slots, nblk = nslots * 8;
    CK(hipMalloc(&st, (size_t)nblk * 64)); CK(hipMemset(st, 0, (size_t)nblk * 64));
    const size_t lds = (size_t)ST_MAX_LDS_CB + 3 * (size_t)KT * BT * TL_BK * 2;
    auto kern = gemm_tiled_kernel<bf16_t, int8_t, D, M, WAVES_T, WN, WT, KT, 32 | ABLX>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (ABLX) printf("(ablated build %d) ", ABLX);
    // random operands: the clock the chip holds depends on the data
    {
        std::vector<unsigned> hx((size_t)T * G * D / 2), hc(pk_bytes / 4), hb((size_t)nparts * K * D / 2);
        unsigned s = 777;
        for (auto &v : hx) { s = s * 1664525u + 1013904223u; v = (s & 0x807f807fu) | 0x3f003f00u; }
        for (auto &v : hb) { s = s * 1664525u + 1013904223u; v = (s & 0x807f807fu) | 0x3f003f00u; }
        for (auto &v : hc) { s = s * 1664525u + 1013904223u; v = s; }
        CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(cb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(pk, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 200; ++i) {   // back to back: the last launch's stamps are read, taken on a loaded chip
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(TL_THREADS), lds, 0, (const uint16_t *)x, T, (int64_t)G * D,
                           (const u32x4 *)pk, (const uint16_t *)cb, (const uint16_t *)sc, (const uint16_t *)st,
                           make_tile_map(parts, BN), G, K, 8, R, (uint16_t *)y, (int64_t)N, n_tiles, t_tiles);
    }
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nblk * 8);
    CK(hipMemcpy(h.data(), st, (size_t)nblk * 64, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < nblk; ++b) { t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 4]); }
    printf("tile %dx%d T=%d N=%d Kin=%d blocks=%d: first start .. last end %.2f us\n", BN, BT, T, N, G * D, nblk, (t1 - t0) / 100.0);
    const char *names[] = {"prologue", "first main loop", "first epilogue", "rest of block", "whole block"};
    for (int k = 0; k < 5; ++k) {
        std::vector<double> v;
        for (int b = 0; b < nblk; ++b) v.push_back(k < 4 ? (h[b * 8 + k + 1] - h[b * 8 + k]) / 100.0 : (h[b * 8 + 4] - h[b * 8]) / 100.0);
        std::sort(v.begin(), v.end());
        double sum = 0; for (double d : v) sum += d;
        printf("   %-16s min %6.2f  med %6.2f  mean %6.2f  max %6.2f us\n", names[k], v[0], v[v.size() / 2], sum / v.size(), v.back());
    }
    {
        std::vector<double> v;
        for (int b = 0; b < nblk; ++b) v.push_back((h[b * 8 + 7] - h[b * 8 + 2]) / 100.0);
        std::sort(v.begin(), v.end());
        printf("   of the first epilogue, until its operands (scales, the next tile's first codes) have landed: med %.2f us\n", v[v.size() / 2]);
    }
    {
        std::vector<double> cyc, ghz;
        for (int b = 0; b < nblk; ++b) {
            cyc.push_back((double)(h[b * 8 + 6] - h[b * 8 + 5]));
            ghz.push_back((double)(h[b * 8 + 6] - h[b * 8 + 5]) / ((h[b * 8 + 2] - h[b * 8 + 1]) * 10.0));
        }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double mfma_cyc = 2.0 * (double)G * D / 32 * WN * WT * 16;   // two waves per SIMD
        printf("   main loop: median %.0f shader cycles (MFMA pipe needs %.0f = %.1f %%), clock median %.2f GHz (min %.2f max %.2f)\n",
               cyc[cyc.size() / 2], mfma_cyc, 100.0 * mfma_cyc / cyc[cyc.size() / 2], ghz[ghz.size() / 2], ghz[0], ghz.back());
    }
    // gaps between consecutive blocks on one CU cannot be seen directly; rounds: sort starts
    std::vector<double> starts;
    for (int b = 0; b < nblk; ++b) starts.push_back((h[b * 8] - t0) / 100.0);
    std::sort(starts.begin(), starts.end());
    printf("   block starts at (us, every 256th):");
    for (size_t i = 0; i < starts.size(); i += 256) printf(" %.1f", starts[i]);
    printf("  last %.1f\n", starts.back());
    CK(hipFree(pk)); CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y)); CK(hipFree(st));
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 4096;
    if (argc > 3) g_slots = atoi(argv[3]);
    if (argc > 2 && argv[2][0] == 's') {
        stamps<2, 8, 4, 2>(T, 28672, 512, 2);
        stamps<2, 8, 4, 2, 4>(T, 28672, 512, 2);  // the same without the epilogue's stores
        stamps<2, 8, 4, 2>(T, 4096, 512, 1);      // o_proj: one tile per block
        stamps<2, 8, 4, 2>(T, 4096, 1792, 1);     // down_proj
        return 0;
    }
    if (argc > 2) {   // per-round constant against per-k slope: the same grid at growing Kin
        for (int G : {256, 512, 1024, 2048}) {
            run<2, 8, 4, 2, 0>("full", T, 28672, G, 2);
            run<2, 8, 4, 2, 31>("mfma only", T, 28672, G, 2);
            run<2, 8, 4, 2, 4>("no epilogue stores", T, 28672, G, 2);
        }
        return 0;
    }
    printf("== gate_up 28672 x 4096, 4 waves ==\n");
    run<1, 8, 8, 2, 0, 4>("4w 512x128 full", T, 28672, 512, 2);
    run<2, 8, 8, 1, 0, 4>("4w 256x256 full", T, 28672, 512, 2);
    run<1, 8, 8, 2, 1, 4>("4w 512x128 no wait / barrier", T, 28672, 512, 2);
    run<1, 8, 8, 2, 8, 4>("4w 512x128 no gathers", T, 28672, 512, 2);
    run<1, 8, 8, 2, 31, 4>("4w 512x128 mfma only", T, 28672, 512, 2);
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
