// Diagnostic harness (not part of the library): how fast can the 8 waves of a block take in 128 KiB that every block
// of the launch reads (the activations of a decode launch: 16 tokens x 4096 features, resident in L2)?  Variants of
// the load instruction and its cache policy, same bytes per wave (16 wave-loads of 1 KiB each).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ingest_lab.hip -o tools/bin/ingest_lab
// Every variant prints, over the 256 blocks of one launch, the time from the first block's start until the block's
// last wave has its data (min / median / max), and the duration of back-to-back launches inside a HIP graph.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz

// pattern 0: 8 rows x 128 bytes per wave-load (row pitch 8 KiB); pattern 1: 1 KiB contiguous per wave-load
template <int VAR>
__global__ __launch_bounds__(512) void ingest(const unsigned char *__restrict__ x, unsigned long long *stamps, unsigned *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned long long t0 = rt();
    u32x4 v[16];
    uint32_t off;
    constexpr bool CONTIG = VAR == 5 || VAR == 7;   // (variants 6, 8, 9 set their own offsets)
    if (CONTIG) off = (uint32_t)wave * 16384u + (uint32_t)lane * 16u;                           // + m * 1024
    else off = (uint32_t)(lane & 7) * 8192u + (uint32_t)((lane >> 4) + 4 * ((lane >> 3) & 1)) * 16u + (uint32_t)wave * 1024u;   // + m * 128 (+ 8 rows)
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + (uint32_t)wave * 16384u;
#define LD(i, OFFS, POL) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFFS " " POL : "=v"(v[i]) : "v"(off), "s"(x) : "memory");
#define LD16(POL) LD(0, 0, POL) LD(2, 128, POL) LD(4, 256, POL) LD(6, 384, POL) LD(8, 512, POL) LD(10, 640, POL) LD(12, 768, POL) LD(14, 896, POL) \
    { const unsigned char *x2 = x + 65536; \
      asm volatile("global_load_dwordx4 %0, %1, %2" POL : "=v"(v[1]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:128" POL : "=v"(v[3]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:256" POL : "=v"(v[5]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:384" POL : "=v"(v[7]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:512" POL : "=v"(v[9]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:640" POL : "=v"(v[11]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:768" POL : "=v"(v[13]) : "v"(off), "s"(x2) : "memory"); \
      asm volatile("global_load_dwordx4 %0, %1, %2 offset:896" POL : "=v"(v[15]) : "v"(off), "s"(x2) : "memory"); }
    if constexpr (VAR == 0) { LD16("") }
    else if constexpr (VAR == 1) { LD16(" nt") }
    else if constexpr (VAR == 2) { LD16(" sc0") }
    else if constexpr (VAR == 3) { LD16(" sc1") }
    else if constexpr (VAR == 4) { LD16(" sc0 sc1") }
    else if constexpr (VAR == 5) {
        // LDS-DMA, 1 KiB contiguous per instruction, no VGPR destination
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds + m * 1024), "v"(off + m * 1024), "s"(x) : "memory", "m0");
        }
#pragma unroll
        for (int m = 0; m < 16; ++m) v[m] = u32x4{0u, 0u, 0u, 0u};
    } else if constexpr (VAR == 6) {
        // 4 rows x 256 bytes per wave-load
        off = (uint32_t)(lane & 3) * 8192u + (uint32_t)((lane >> 4) + 4 * ((lane >> 2) & 3)) * 16u + (uint32_t)wave * 1024u;
#define LQ(i, R4, OFFS) { const unsigned char *xq = x + (R4) * 32768; asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFFS : "=v"(v[i]) : "v"(off), "s"(xq) : "memory"); }
        LQ(0, 0, 0) LQ(1, 1, 0) LQ(2, 2, 0) LQ(3, 3, 0) LQ(4, 0, 256) LQ(5, 1, 256) LQ(6, 2, 256) LQ(7, 3, 256)
        LQ(8, 0, 512) LQ(9, 1, 512) LQ(10, 2, 512) LQ(11, 3, 512) LQ(12, 0, 768) LQ(13, 1, 768) LQ(14, 2, 768) LQ(15, 3, 768)
#undef LQ
    } else if constexpr (VAR == 8) {
        // 8 rows x 128 bytes per wave-load, consecutive lanes on consecutive pieces of one row
        off = (uint32_t)(lane >> 3) * 8192u + (uint32_t)(lane & 7) * 16u + (uint32_t)wave * 1024u;
        LD16("")
    } else if constexpr (VAR == 9 || VAR == 10) {
        // 16 rows x 64 bytes per wave-load (one k-step of all 16 tokens), consecutive lanes on consecutive pieces
        off = (uint32_t)(lane >> 2) * 8192u + (uint32_t)(lane & 3) * 16u + (uint32_t)wave * 1024u;
#define LK(i, OFFS) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFFS : "=v"(v[i]) : "v"(off), "s"(x) : "memory");
        LK(0, 0) LK(1, 64) LK(2, 128) LK(3, 192) LK(4, 256) LK(5, 320) LK(6, 384) LK(7, 448)
        LK(8, 512) LK(9, 576) LK(10, 640) LK(11, 704) LK(12, 768) LK(13, 832) LK(14, 896) LK(15, 960)
#undef LK
    } else {
        // 1 KiB contiguous per wave-load into registers
#pragma unroll
        for (int m = 0; m < 16; ++m) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v[m]) : "v"(off + m * 1024), "s"(x) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (VAR == 10) {
        // lane (q, c) takes what lane 4 c + q loaded: the MFMA B operand of the k-step (token c, k-octet q)
        const int src = (((lane & 15) << 2) | (lane >> 4)) << 2;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("" : "+v"(v[m]));
            v[m].x = __builtin_amdgcn_ds_bpermute(src, v[m].x); v[m].y = __builtin_amdgcn_ds_bpermute(src, v[m].y);
            v[m].z = __builtin_amdgcn_ds_bpermute(src, v[m].z); v[m].w = __builtin_amdgcn_ds_bpermute(src, v[m].w);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = rt();
    uint32_t acc = 0;
#pragma unroll
    for (int m = 0; m < 16; ++m) { asm volatile("" : "+v"(v[m])); acc ^= v[m].x ^ v[m].y ^ v[m].z ^ v[m].w; }
    __syncthreads();
    const unsigned long long t2 = rt();
    if (acc == 0x12345678u) sink[tid] = acc + smem[lane];
    if (tid == 0) { stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; }
}

template <int VAR>
static void run(const char *what, const unsigned char *x, unsigned long long *st, unsigned *sink) {
    auto k = ingest<VAR>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const size_t lds = 131072;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, s, x, st, sink);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> h(256 * 4);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < 256; ++b) t0 = std::min(t0, h[b * 4]);
    std::vector<double> land, bar;
    for (int b = 0; b < 256; ++b) { land.push_back((h[b * 4 + 1] - t0) / 100.0); bar.push_back((h[b * 4 + 2] - t0) / 100.0); }
    std::sort(land.begin(), land.end()); std::sort(bar.begin(), bar.end());
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 128; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, s, x, st, sink);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s wave 0 landed %5.2f %5.2f %5.2f   block barrier %5.2f %5.2f %5.2f us   graph: %5.2f us/launch\n", what, land[0],
           land[128], land[255], bar[0], bar[128], bar[255], ms * 1e3 / 1280);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
}

int main() {
    unsigned char *x; unsigned long long *st; unsigned *sink;
    CK(hipMalloc(&x, 1 << 20)); CK(hipMemset(x, 0x3c, 1 << 20));
    CK(hipMalloc(&st, 256 * 4 * 8)); CK(hipMalloc(&sink, 4096));
    run<0>("8 rows x 128 B, default policy", x, st, sink);
    run<1>("8 rows x 128 B, nt", x, st, sink);
    run<2>("8 rows x 128 B, sc0", x, st, sink);
    run<3>("8 rows x 128 B, sc1", x, st, sink);
    run<4>("8 rows x 128 B, sc0 sc1", x, st, sink);
    run<6>("4 rows x 256 B, default policy", x, st, sink);
    run<8>("8 rows x 128 B, lanes along the row", x, st, sink);
    run<9>("16 rows x 64 B, lanes along the row", x, st, sink);
    run<10>("16 rows x 64 B along the row + 64 bpermute", x, st, sink);
    run<7>("1 KiB contiguous, registers", x, st, sink);
    run<5>("1 KiB contiguous, LDS-DMA", x, st, sink);
    return 0;
}
