// Diagnostic harness (not part of the library): the mid-batch kernel (pq_gemm_wide_impl.cuh) on the Llama-3-8B layer
// shapes -- launch time of the full kernel and of ablated builds (no MFMAs, no gathers, neither), and per-block phase
// stamps (wall clock, 100 MHz) of one launch under back-to-back load.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I vllm_pq_amd/csrc tools/wide_lab.hip -o tools/bin/wide_lab
//   tools/bin/wide_lab T [nf rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "pq_gemm_wide_impl.cuh"

using namespace pq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Shape { const char *name; int nparts; int parts[4]; int Kin; };

static WideParams make_params(const Shape &sh, int T, int nf, int rounds, void *x, void *pk, void *cb, void *sc, void *bias, void *y,
                              dim3 *grid, size_t *lds) {
    Parts parts;
    parts.num = sh.nparts;
    int N = 0;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) { parts.size[p] = p < sh.nparts ? sh.parts[p] : 0; N += parts.size[p]; }
    WideParams p;
    const int nchunks = sh.Kin / 512;
    p.x = (const unsigned char *)x; p.codes = (const unsigned char *)pk; p.codebooks = (const unsigned char *)cb;
    p.scales = (const uint16_t *)sc; p.bias = (const uint16_t *)bias; p.y = (uint16_t *)y; p.y_stride = N;
    p.x_stride_bytes = sh.Kin * 2; p.T = T; p.nchunks = nchunks; p.K = 256; p.nf = nf; p.rounds = rounds;
    p.wk_log2 = nchunks >= 5 ? 3 : nchunks >= 3 ? 2 : nchunks >= 2 ? 1 : 0;
    p.img_bytes = 65536; p.spt = 8; p.wave_bytes = (uint32_t)wd_wave_bytes(nf); p.ring = wd_ring(nf);
    const int frags_blk = (WD_WAVES >> p.wk_log2) * rounds * nf;
    p.sc_off = p.img_bytes + WD_WAVES * p.wave_bytes;
    p.tmap = make_tile_map(parts, frags_blk * 16);
    p.n_tiles = tiles_in(parts, frags_blk * 16); p.tgroups = (T + WD_TOKENS - 1) / WD_TOKENS;
    *grid = dim3((p.n_tiles + 7) / 8 * 8 * p.tgroups);
    *lds = wd_lds_bytes(p.img_bytes, nf, frags_blk);
    return p;
}

template <bool STAMPS, int ABL, int NF>
static float run_nf(const Shape &sh, int T, int rounds, const char *label, bool stamps_out) {
    const int nf = NF;
    int N = 0;
    for (int p = 0; p < sh.nparts; ++p) N += sh.parts[p];
    const int nchunks = sh.Kin / 512;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    const int copies = std::max(2, (int)(600e6 / pk_bytes));
    std::vector<void *> pk(copies);
    void *x, *cb, *sc, *y, *st;
    std::vector<unsigned> h(std::max(pk_bytes, (size_t)T * sh.Kin * 2) / 4);
    unsigned s = 777;
    for (int i = 0; i < copies; ++i) {
        CK(hipMalloc(&pk[i], pk_bytes));
        for (size_t k = 0; k < pk_bytes / 4; ++k) { s = s * 1664525u + 1013904223u; h[k] = s; }
        CK(hipMemcpy(pk[i], h.data(), pk_bytes, hipMemcpyHostToDevice));
        if (i >= 3) { CK(hipMemcpy(pk[i], pk[i % 3], pk_bytes, hipMemcpyDeviceToDevice)); }
    }
    CK(hipMalloc(&x, (size_t)T * sh.Kin * 2));
    for (size_t k = 0; k < (size_t)T * sh.Kin / 2; ++k) { s = s * 1664525u + 1013904223u; h[k] = (s & 0x807f807fu) | 0x3f003f00u; }
    CK(hipMemcpy(x, h.data(), (size_t)T * sh.Kin * 2, hipMemcpyHostToDevice));
    CK(hipMalloc(&cb, (size_t)sh.nparts * 4096));
    for (size_t k = 0; k < (size_t)sh.nparts * 1024; ++k) { s = s * 1664525u + 1013904223u; h[k] = (s & 0x807f807fu) | 0x3f003f00u; }
    CK(hipMemcpy(cb, h.data(), (size_t)sh.nparts * 4096, hipMemcpyHostToDevice));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    dim3 grid; size_t lds;
    WideParams p0 = make_params(sh, T, nf, rounds, x, pk[0], cb, sc, nullptr, y, &grid, &lds);
    const int nblk = grid.x;
    CK(hipMalloc(&st, (size_t)nblk * 56 * 8)); CK(hipMemset(st, 0, (size_t)nblk * 56 * 8));
    auto kern = gemm_wide_kernel<bf16_t, NF, STAMPS, ABL>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    auto launch = [&](int i) {
        WideParams p = p0;
        p.codes = (const unsigned char *)pk[i % copies];
        if (STAMPS) p.bias = (const uint16_t *)st;
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, 0, p);
    };
    for (int i = 0; i < 20; ++i) launch(i);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nrep = 200;
    CK(hipEventRecord(e0));
    for (int i = 0; i < nrep; ++i) launch(i);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const float us = ms * 1e3f / nrep;
    const double flops = 2.0 * T * N * (double)sh.Kin;
    printf("%-10s T=%4d %-22s nf=%d rounds=%2d blocks=%4d  %7.2f us/launch  %6.1f TFLOP/s  frac %.3f\n", sh.name, T, label, nf, rounds,
           nblk, us, flops / us * 1e-6, flops / us * 1e-6 / 2500.0);
    if (STAMPS && stamps_out) {
        std::vector<unsigned long long> hs((size_t)nblk * 56);
        CK(hipMemcpy(hs.data(), st, (size_t)nblk * 56 * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nblk; ++b) if (hs[(size_t)b * 56]) t0 = std::min(t0, hs[(size_t)b * 56]);
        const char *names[12] = {"block start", "tile found", "entries + x requested", "first x fragment in regs", "x in registers",
                                 "barrier", "image stored + barrier", "first gathers issued", "first round's units done", "all rounds done", "", "stores retired"};
        for (int k : {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 11}) {
            std::vector<double> v;
            for (int b = 0; b < nblk; ++b) if (hs[(size_t)b * 56]) v.push_back((hs[(size_t)b * 56 + k] - t0) / 100.0);
            std::sort(v.begin(), v.end());
            printf("     %-26s min %6.2f  med %6.2f  max %6.2f us (wave 0)\n", names[k], v[0], v[v.size() / 2], v.back());
        }
        const char *rn[7] = {"round starts", "first unit done", "units done", "barrier passed", "tiles summed", "second barrier passed", "stores issued"};
        for (int w = 0; w < 2; ++w)
            for (int k = 1; k < 7; ++k) {
                std::vector<double> v;
                for (int b = 0; b < nblk; ++b)
                    if (hs[(size_t)b * 56] && hs[(size_t)b * 56 + w * 28 + 12 + k] && hs[(size_t)b * 56 + w * 28 + 12])
                        v.push_back((double)(hs[(size_t)b * 56 + w * 28 + 12 + k] - hs[(size_t)b * 56 + w * 28 + 12]) / 100.0);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                printf("       second round, wave %d: %-22s +%5.2f (min %5.2f max %5.2f) us after the round's start\n", w ? 7 : 0, rn[k], v[v.size() / 2], v[0], v.back());
            }
        const char *un[8] = {"unit starts", "next piece requested", "second-half gathers issued", "piece waited for, codes read issued", "first-half MFMAs issued",
                             "second-half gathers landed", "next first-half gathers issued", "second-half MFMAs issued, tiles stored"};
        for (int w = 0; w < 2; ++w)
            for (int k = 1; k < 8; ++k) {
                std::vector<double> v;
                for (int b = 0; b < nblk; ++b)
                    if (hs[(size_t)b * 56] && hs[(size_t)b * 56 + w * 28 + 20 + k] && hs[(size_t)b * 56 + w * 28 + 20])
                        v.push_back((double)(hs[(size_t)b * 56 + w * 28 + 20 + k] - hs[(size_t)b * 56 + w * 28 + 20]));
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                printf("         second unit, wave %d: %-40s +%6.0f shader cycles (min %6.0f max %6.0f)\n", w ? 7 : 0, un[k], v[v.size() / 2], v[0], v.back());
            }
    }
    for (auto q : pk) CK(hipFree(q));
    CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y)); CK(hipFree(st));
    return us;
}

template <bool STAMPS, int ABL>
static float run(const Shape &sh, int T, int nf, int rounds, const char *label, bool stamps_out) {
    switch (nf) {
    case 1: return run_nf<STAMPS, ABL, 1>(sh, T, rounds, label, stamps_out);
    case 2: return run_nf<STAMPS, ABL, 2>(sh, T, rounds, label, stamps_out);
    case 3: return run_nf<STAMPS, ABL, 3>(sh, T, rounds, label, stamps_out);
    default: return run_nf<STAMPS, ABL, 4>(sh, T, rounds, label, stamps_out);
    }
}

static void choose(const Shape &sh, int T, int *nf, int *rounds) {
    Parts parts;
    parts.num = sh.nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) parts.size[p] = p < sh.nparts ? sh.parts[p] : 0;
    const int nchunks = sh.Kin / 512;
    const int wk_log2 = nchunks >= 5 ? 3 : nchunks >= 3 ? 2 : nchunks >= 2 ? 1 : 0;
    const int WR = WD_WAVES >> wk_log2;
    double best = -1;
    for (int f = 1; f <= WD_MAX_NF; ++f)
        for (int r = 1; WR * r * f <= WD_MAX_FRAGS; ++r) {
            const long blocks = (long)tiles_in(parts, WR * r * f * 16) * ((T + 31) / 32);
            const double cost = (double)((blocks + 255) / 256) * (7.0 + r * (0.65 * f + 0.7));
            if (best < 0 || cost < best - 1e-9) { best = cost; *nf = f; *rounds = r; }
        }
}

int main(int argc, char **argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 128;
    const Shape shapes[] = {{"8B.qkv", 3, {4096, 1024, 1024, 0}, 4096}, {"8B.o", 1, {4096, 0, 0, 0}, 4096},
                            {"8B.gate_up", 2, {14336, 14336, 0, 0}, 4096}, {"tp4.o", 1, {4096, 0, 0, 0}, 1024},
                            {"tp4.qkv", 3, {1024, 256, 256, 0}, 4096}, {"tp4.gate_up", 2, {3584, 3584, 0, 0}, 4096}};
    for (const Shape &sh : shapes) {
        int nf = 4, rounds = 1;
        if (argc > 3) { nf = atoi(argv[2]); rounds = atoi(argv[3]); } else choose(sh, T, &nf, &rounds);
        run<false, 0>(sh, T, nf, rounds, "full", false);
        run<true, 0>(sh, T, nf, rounds, "full, stamped", true);
    }
    return 0;
}
