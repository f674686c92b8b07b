// Diagnostic harness (not part of the library): phase stamps and back-to-back launch times of the decode kernel on
// the per-rank model shapes, codes rotating over copies so that they come from HBM.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I vllm_pq_amd/csrc tools/decode_lab.hip \
//         vllm_pq_amd/csrc/pq_gemm_decode.hip -o tools/decode_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "pq_gemm_decode_impl.cuh"
using namespace pq;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#include "pq_knobs.h"
namespace pq {
DecodeGeom choose_decode_geom(const GemmArgs &a);
const Knobs &knobs() { static Knobs k{-1, 0, 0, 0, -1, 0, 0, 0}; return k; }
}

struct Shape { const char *name; int nparts; int parts[3]; int Kin; };

template <int ABL = 0>
static void run(const Shape &sh, int T, int waves_force, int nf_force, bool stamps) {
    const int K = 256;
    int N = 0;
    GemmArgs a;
    a.parts.num = sh.nparts;
    for (int p = 0; p < PQ_MAX_PARTS; ++p) { a.parts.size[p] = p < sh.nparts ? sh.parts[p] : 0; N += a.parts.size[p]; }
    const int G = sh.Kin / 8, nchunks = G / 64;
    const size_t pk_bytes = (size_t)(N / 16) * nchunks * 1024;
    int copies = (int)(600000000ull / pk_bytes);
    copies = copies < 2 ? 2 : copies > 64 ? 64 : copies;
    std::vector<void *> pk(copies);
    for (int i = 0; i < copies; ++i) { CK(hipMalloc(&pk[i], pk_bytes)); CK(hipMemset(pk[i], i + 1, pk_bytes)); }
    void *x, *cb, *sc, *y, *st;
    CK(hipMalloc(&x, (size_t)T * sh.Kin * 2)); CK(hipMemset(x, 0x3c, (size_t)T * sh.Kin * 2));
    CK(hipMalloc(&cb, (size_t)sh.nparts * K * 16)); CK(hipMemset(cb, 0x3c, (size_t)sh.nparts * K * 16));
    CK(hipMalloc(&sc, (size_t)N * 2)); CK(hipMemset(sc, 0x3c, (size_t)N * 2));
    CK(hipMalloc(&y, (size_t)T * N * 2));
    a.x = x; a.T = T; a.x_stride = sh.Kin; a.codes = pk[0]; a.codebooks = cb; a.scales = sc; a.bias = nullptr;
    a.N = N; a.G = G; a.M = 1; a.K = K; a.d = 8; a.code_bytes = 1; a.dtype = PQ_DTYPE_BF16; a.y = y; a.y_stride = N;
    static void *ws = nullptr;   // K-split scratch (counters start at zero and every launch leaves them at zero)
    if (!ws) { CK(hipMalloc(&ws, WS_COUNTER_BYTES + WS_PARTIAL_BYTES)); CK(hipMemset(ws, 0, WS_COUNTER_BYTES + WS_PARTIAL_BYTES)); }
    a.workspace = ws; a.workspace_bytes = WS_COUNTER_BYTES + WS_PARTIAL_BYTES; a.packed = true;
    DecodeGeom g = choose_decode_geom(a);
    if (g.nf == 0) { printf("%-16s Kin=%d: not taken by the decode kernel\n", sh.name, sh.Kin); return; }
    if (waves_force) { g.waves = waves_force; g.nf = nf_force; }
    DecodeParams p;
    p.x = (const unsigned char *)x; p.codebooks = (const unsigned char *)cb; p.scales = (const uint16_t *)sc; p.bias = nullptr;
    p.y = (uint16_t *)y; p.y_stride = N; p.x_stride_bytes = sh.Kin * 2; p.T = T; p.nchunks = nchunks; p.K = K; p.nf = g.nf;
    p.wk_log2 = g.wk_log2; p.img_bytes = K * 256; p.spt = K * 16 >= g.waves * 64 ? K * 16 / (g.waves * 64) : 1;
    p.aux_off = (uint32_t)dc_lds_main_bytes(g.waves, p.img_bytes);
    p.ksplit = g.ksplit; p.cpb = g.cpb; p.n_total = N; p.ws_cnt = (unsigned *)a.workspace;
    p.ws_part = (float *)((unsigned char *)a.workspace + WS_COUNTER_BYTES);
    p.act_rows = 0; p.phases = 2;
    const int rows = (g.waves >> g.wk_log2) * g.nf * 16;
    p.tmap = make_tile_map(a.parts, rows);
    const int ntile = tiles_in(a.parts, rows), nblk = ntile * g.ksplit;
    const size_t lds = dc_lds_bytes(g.waves, p.img_bytes);
    CK(hipMalloc(&st, (size_t)nblk * 192)); CK(hipMemset(st, 0, (size_t)nblk * 192));
    auto launch = [&](int i, bool stamped) {
        p.codes = (const unsigned char *)pk[i % copies];
        p.bias = stamped ? (const uint16_t *)st : nullptr;
        if (stamped) hipLaunchKernelGGL((gemm_decode_kernel<bf16_t, true>), dim3(ntile, g.ksplit), dim3(g.waves * 64), lds, 0, p);
        else hipLaunchKernelGGL((gemm_decode_kernel<bf16_t, false, ABL>), dim3(ntile, g.ksplit), dim3(g.waves * 64), lds, 0, p);
    };
    static bool attr = false;
    if (!attr) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_decode_kernel<bf16_t, false, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dc_lds_bytes(DC_MAX_WAVES, 65536)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_decode_kernel<bf16_t, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dc_lds_bytes(DC_MAX_WAVES, 65536)));
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_decode_kernel<bf16_t, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dc_lds_bytes(DC_MAX_WAVES, 65536)));
        attr = true;
    }
    // timed, back to back
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) launch(i, false);
    CK(hipDeviceSynchronize());
    const int nrep = 200;
    CK(hipEventRecord(e0));
    for (int i = 0; i < nrep; ++i) launch(i, false);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / nrep;
    const double bytes = (double)N * G + (double)sh.nparts * K * 16 + N * 2.0 + (double)T * sh.Kin * 2 + (double)T * N * 2;
    printf("%-16s%s T=%2d N=%5d Kin=%5d waves=%d wk=%d nf=%d c=%d blocks=%4d: %6.2f us/launch  %6.0f GB/s  (%.3f of 8 TB/s)\n", sh.name, ABL == 0 ? "" : ABL == 1 ? " [no MFMA]" : ABL == 2 ? " [no gathers]" : " [neither]", T, N,
           sh.Kin, g.waves, 1 << g.wk_log2, g.nf, g.c, nblk, us, bytes / us * 1e-3, bytes / us * 1e-3 / 8000.0);
    if (stamps) {
        for (int i = 0; i < 4; ++i) { launch(i, true); CK(hipDeviceSynchronize()); }
        std::vector<unsigned long long> h((size_t)nblk * 24);
        CK(hipMemcpy(h.data(), st, (size_t)nblk * 192, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nblk; ++b) t0 = std::min(t0, std::min(h[(size_t)b * 24], h[(size_t)b * 24 + 12]));
        const char *names[] = {"start", "tile found", "x requested", "x landed", "operands read", "barrier (x out)",
                               "codes requested", "image, barrier", "-", "own units done", "barrier (units)", "stores retired"};
        printf("      %-22s %s\n", "", "first wave: min / med / max        last wave: min / med / max   (us after the first start)");
        for (int k = 0; k < 12; ++k) {
            std::vector<double> v, w;
            if (h[k] == 0) continue;
            for (int b = 0; b < nblk; ++b) { v.push_back((h[(size_t)b * 24 + k] - t0) / 100.0); w.push_back((h[(size_t)b * 24 + 12 + k] - t0) / 100.0); }
            std::sort(v.begin(), v.end()); std::sort(w.begin(), w.end());
            printf("      %-22s %6.2f %6.2f %6.2f            %6.2f %6.2f %6.2f\n", names[k], v[0], v[v.size() / 2], v.back(), w[0], w[w.size() / 2], w.back());
        }
    }
    for (auto q : pk) CK(hipFree(q));
    CK(hipFree(x)); CK(hipFree(cb)); CK(hipFree(sc)); CK(hipFree(y)); CK(hipFree(st));
}

int main(int argc, char **argv) {
    const bool stamps = argc > 1 && strchr(argv[1], 's');
    static const Shape shapes[] = {
        {"8B.qkv", 3, {4096, 1024, 1024}, 4096}, {"8B.o", 1, {4096, 0, 0}, 4096}, {"8B.gate_up", 2, {14336, 14336, 0}, 4096},
        {"8B.down", 1, {4096, 0, 0}, 14336}, {"8B/tp4.qkv", 3, {1024, 256, 256}, 4096}, {"8B/tp4.o", 1, {4096, 0, 0}, 1024},
        {"8B/tp4.gate_up", 2, {3584, 3584, 0}, 4096}, {"8B/tp4.down", 1, {4096, 0, 0}, 3584},
        {"70B/tp8.qkv", 3, {1024, 128, 128}, 8192}, {"70B/tp8.o", 1, {8192, 0, 0}, 1024},
        {"70B/tp8.gate_up", 2, {3584, 3584, 0}, 8192}, {"70B/tp8.down", 1, {8192, 0, 0}, 3584}};
    for (const Shape &sh : shapes) run(sh, 16, 0, 0, stamps);
    run(shapes[2], 1, 0, 0, stamps);
    for (const Shape &sh : shapes) run(sh, 8, 0, 0, false);
    run(shapes[1], 1, 0, 0, stamps);
    printf("---- ablations (wrong results): gate_up and down_proj at 16 tokens without the MFMAs / the gathers / both ----\n");
    run<1>(shapes[2], 16, 0, 0, false); run<2>(shapes[2], 16, 0, 0, false); run<3>(shapes[2], 16, 0, 0, false);
    run<1>(shapes[3], 16, 0, 0, false); run<2>(shapes[3], 16, 0, 0, false); run<3>(shapes[3], 16, 0, 0, false);
    // alternatives for the big ones
    run(shapes[2], 16, 8, 8, false);
    run(shapes[0], 16, 8, 1, false);
    run(shapes[0], 16, 8, 3, false);
    return 0;
}
