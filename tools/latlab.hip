// Diagnostic harness (not part of the library): where the first microseconds of a short decode launch go.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/latlab.hip -o tools/latlab
// Tests (each prints medians over blocks / waves):
//   empty     back-to-back launch cost of an empty 256 x 512 kernel (with and without 64 KiB of LDS)
//   icache    a straight-line body of 8-byte instructions run twice by every wave: first pass (cold
//             instruction cache) against second pass (warm)
//   first     dependent first loads of a block: a small shared table, a per-block 64 KiB piece of a large
//             buffer never touched before (and touched by the previous launch), a shared 128 KiB buffer
//   burst     every wave issues NC loads of its block's piece and NX loads of the shared buffer without
//             waiting; stamps: issued, landed
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned long long rt() { return __builtin_amdgcn_s_memrealtime(); }   // 100 MHz
__device__ inline unsigned long long ct() { return __builtin_amdgcn_s_memtime(); }       // shader clock

__global__ __launch_bounds__(512) void empty_k(int *p) {
    extern __shared__ unsigned char smem[];
    if (p && threadIdx.x == 9999) p[0] = smem[0];
}

template <int N>
__global__ __launch_bounds__(512) void icache_k(unsigned long long *out, int passes) {
    unsigned v = threadIdx.x, w = 3;
    unsigned long long t[4];
    for (int it = 0; it < passes; ++it) {
        t[it] = ct();
        asm volatile(".rept %c2\n\tv_add3_u32 %0, %0, %1, %1\n\t.endr" : "+v"(v) : "v"(w), "n"(N));
    }
    t[passes] = ct();
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *o = out + ((size_t)blockIdx.x * 8 + (threadIdx.x >> 6)) * 4;
        for (int i = 0; i < 3; ++i) o[i] = i < passes ? t[i + 1] - t[i] : 0;
        o[3] = v;
    }
}

// stamps per block (thread 0): 0 entry, 1 table landed, 2 piece landed, 3 shared landed
__global__ __launch_bounds__(512) void first_k(const u32x4 *table, const u32x4 *big, size_t piece_slots, const u32x4 *shared_x,
                                               unsigned long long *stamps, u32x4 *sink) {
    const int tid = threadIdx.x;
    unsigned long long s0 = rt();
    u32x4 a = table[tid & 255];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(a));
    unsigned long long s1 = rt();
    u32x4 b = __builtin_nontemporal_load(&big[(size_t)blockIdx.x * piece_slots + tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(b));
    unsigned long long s2 = rt();
    u32x4 c = shared_x[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(c));
    unsigned long long s3 = rt();
    if (tid == 0) {
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    if (a.x + b.x + c.x == 0x12345u) sink[tid] = a;
}

// every wave: NC 1-KiB loads of its block's piece, then NX 1-KiB loads of the shared buffer; ORDER 1: shared first
template <int NC, int NX, int ORDER>
__global__ __launch_bounds__(512) void burst_k(const u32x4 *big, size_t piece_slots, const u32x4 *shared_x,
                                               unsigned long long *stamps, u32x4 *sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long s0 = rt();
    u32x4 c[NC > 0 ? NC : 1], x[NX > 0 ? NX : 1];
    auto ld_c = [&]() {
#pragma unroll
        for (int i = 0; i < NC; ++i) c[i] = __builtin_nontemporal_load(&big[(size_t)blockIdx.x * piece_slots + (size_t)(i * 8 + wave) * 64 + lane]);
    };
    auto ld_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) x[i] = shared_x[(size_t)(wave * NX + i) * 64 + lane];
    };
    if (ORDER == 0) { ld_c(); ld_x(); } else { ld_x(); ld_c(); }
    unsigned long long s1 = rt();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NC; ++i) acc += c[i].x;
#pragma unroll
    for (int i = 0; i < NX; ++i) acc += x[i].x;
    asm volatile("" ::"v"(acc));
    unsigned long long s2 = rt();
    __syncthreads();
    unsigned long long s3 = rt();
    if (tid == 0) {
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    if (acc == 0x12345u) sink[tid] = c[0];
}

// the decode kernel's activation loads: wave w, load s, lane (t = l & 15, q = l >> 4) reads 16 bytes at
// t * pitch + w * 1024 + s * 64 + q * 16: 16 rows x 64 bytes per wave-instruction
__global__ __launch_bounds__(512) void xpat_k(const unsigned char *x, unsigned pitch, int nloads, unsigned long long *stamps, u32x4 *sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long s0 = rt();
    const unsigned char *base = x + (size_t)(lane & 15) * pitch + wave * 1024 + (lane >> 4) * 16;
    u32x4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = i < nloads ? *reinterpret_cast<const u32x4 *>(base + i * 64) : u32x4{0, 0, 0, 0};
    unsigned long long s1 = rt();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x;
    asm volatile("" ::"v"(acc));
    unsigned long long s2 = rt();
    __syncthreads();
    unsigned long long s3 = rt();
    if (tid == 0) {
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    if (acc == 0x12345u) sink[tid] = v[0];
}

// the same bytes with whole 128-byte lines per lane group: load i, lane (q = l >> 4, c = l & 15) reads 16 bytes of
// token (c & 7) + 8 (i & 1) at line (i >> 1), piece q + 4 (c >> 3): 8 rows x 128 bytes per wave-instruction
__global__ __launch_bounds__(512) void xpat2_k(const unsigned char *x, unsigned pitch, int mode, unsigned long long *stamps, u32x4 *sink) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    unsigned long long s0 = rt();
    const int q = lane >> 4, c = lane & 15;
    u32x4 v[16];
    if (mode == 0 || mode >= 2) {
        // mode 2: the wave's chunk rotated by the block index; mode 3: also the line order within the chunk
        const int wv = mode >= 2 ? (wave + blockIdx.x) & 7 : wave;
        const int rot = mode == 3 ? (blockIdx.x >> 3) & 7 : 0;
        const unsigned char *base = x + (size_t)(c & 7) * pitch + wv * 1024 + (q + 4 * (c >> 3)) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const u32x4 *>(base + (size_t)(i & 1) * 8 * pitch + (((i >> 1) + rot) & 7) * 128);
    } else {   // 4 rows x 256 bytes: token (c & 3) + 4 (i & 3), piece q + 4 (c >> 2)
        const unsigned char *base = x + (size_t)(c & 3) * pitch + wave * 1024 + (q + 4 * (c >> 2)) * 16;
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = *reinterpret_cast<const u32x4 *>(base + (size_t)(i & 3) * 4 * pitch + (i >> 2) * 256);
    }
    unsigned long long s1 = rt();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += v[i].x;
    asm volatile("" ::"v"(acc));
    unsigned long long s2 = rt();
    __syncthreads();
    unsigned long long s3 = rt();
    if (tid == 0) {
        unsigned long long *o = stamps + (size_t)blockIdx.x * 8;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
    if (acc == 0x12345u) sink[tid] = v[0];
}

static void report(const char *label, const std::vector<unsigned long long> &h, int nblk, int nst, const char *const *names) {
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nblk; ++b) t0 = std::min(t0, h[(size_t)b * 8]);
    printf("%s\n", label);
    for (int k = 0; k < nst; ++k) {
        std::vector<double> v;
        for (int b = 0; b < nblk; ++b) v.push_back((h[(size_t)b * 8 + k] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        printf("   %-18s min %6.2f  med %6.2f  max %6.2f us\n", names[k], v[0], v[v.size() / 2], v.back());
    }
}

int main() {
    const int NB = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // ---- empty
    {
        auto k = empty_k;
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        for (size_t lds : {(size_t)0, (size_t)65536}) {
            for (int nb : {256, 1}) {
                for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(512), lds, 0, (int *)nullptr);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0));
                for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(512), lds, 0, (int *)nullptr);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("empty: %3d blocks x 512, lds %6zu: %.2f us/launch (eager, back to back)\n", nb, lds, ms * 1e3 / 400);
            }
        }
        // the same in a graph
        hipStream_t s;
        CK(hipStreamCreate(&s));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 128; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(512), 65536, s, (int *)nullptr);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty: 256 blocks x 512, lds 65536, graph of 128: %.2f us/launch\n", ms * 1e3 / 1280);
    }
    // ---- icache
    {
        unsigned long long *out;
        CK(hipMalloc(&out, (size_t)NB * 8 * 4 * 8));
        std::vector<unsigned long long> h((size_t)NB * 8 * 4);
        auto run = [&](auto kern, int n, const char *label) {
            for (int rep = 0; rep < 3; ++rep) {
                hipLaunchKernelGGL(kern, dim3(NB), dim3(512), 0, 0, out, 2);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> p1, p2;
            for (size_t w = 0; w < (size_t)NB * 8; ++w) { p1.push_back((double)h[w * 4]); p2.push_back((double)h[w * 4 + 1]); }
            std::sort(p1.begin(), p1.end()); std::sort(p2.begin(), p2.end());
            printf("icache %s: %d x 8-byte VALU per wave, 8 waves/CU: pass 1 med %.0f max %.0f cycles (%.2f/instr), pass 2 med %.0f max %.0f (%.2f/instr)\n",
                   label, n, p1[p1.size() / 2], p1.back(), p1[p1.size() / 2] / n, p2[p2.size() / 2], p2.back(), p2[p2.size() / 2] / n);
        };
        run(icache_k<512>, 512, "4 KiB");
        run(icache_k<2048>, 2048, "16 KiB");
        run(icache_k<6144>, 6144, "48 KiB");
        CK(hipFree(out));
    }
    // ---- first loads / bursts
    {
        const size_t piece = 65536, piece_slots = piece / 16;
        const int copies = 24;
        const size_t big_bytes = (size_t)NB * piece;   // 16 MiB per copy
        std::vector<u32x4 *> big(copies);
        for (int i = 0; i < copies; ++i) { CK(hipMalloc(&big[i], big_bytes)); CK(hipMemset(big[i], i + 1, big_bytes)); }
        u32x4 *table, *shared_x, *sink;
        unsigned long long *st;
        CK(hipMalloc(&table, 4096)); CK(hipMemset(table, 1, 4096));
        CK(hipMalloc(&shared_x, 131072)); CK(hipMemset(shared_x, 2, 131072));
        CK(hipMalloc(&sink, 8192));
        CK(hipMalloc(&st, (size_t)NB * 64));
        std::vector<unsigned long long> h((size_t)NB * 8);
        const char *n1[] = {"entry", "table landed", "piece landed", "shared landed"};
        for (int i = 0; i < copies; ++i) {   // fresh copy every launch
            hipLaunchKernelGGL(first_k, dim3(NB), dim3(512), 0, 0, table, big[i], piece_slots, shared_x, st, sink);
            CK(hipDeviceSynchronize());
            if (i == 0 || i == copies - 1) {
                CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
                report(i == 0 ? "first: very first launch" : "first: fresh piece every launch (steady state)", h, NB, 4, n1);
            }
        }
        hipLaunchKernelGGL(first_k, dim3(NB), dim3(512), 0, 0, table, big[copies - 1], piece_slots, shared_x, st, sink);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
        report("first: same piece as the previous launch", h, NB, 4, n1);

        const char *n2[] = {"entry", "issued", "landed", "block barrier"};
        auto burst = [&](auto kern, const char *label) {
            for (int i = 0; i < copies; ++i) {
                hipLaunchKernelGGL(kern, dim3(NB), dim3(512), 0, 0, big[i], piece_slots, shared_x, st, sink);
                CK(hipDeviceSynchronize());
            }
            CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
            report(label, h, NB, 4, n2);
            // and timed back to back over the copies
            CK(hipEventRecord(e0));
            for (int i = 0; i < 10 * copies; ++i)
                hipLaunchKernelGGL(kern, dim3(NB), dim3(512), 0, 0, big[i % copies], piece_slots, shared_x, st, sink);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("   back to back: %.2f us/launch\n", ms * 1e3 / (10 * copies));
        };
        burst(burst_k<8, 0, 0>, "burst: 8 piece loads per wave (64 KiB/CU from HBM)");
        burst(burst_k<0, 16, 0>, "burst: 16 shared loads per wave (128 KiB/CU from L2)");
        burst(burst_k<8, 16, 0>, "burst: 8 piece + 16 shared, piece first");
        burst(burst_k<8, 16, 1>, "burst: 8 piece + 16 shared, shared first");
        burst(burst_k<1, 16, 0>, "burst: 1 piece + 16 shared (o_proj-like)");
        burst(burst_k<4, 8, 0>, "burst: 4 piece + 8 shared");
        // activation pattern against the row pitch
        {
            unsigned char *xb;
            CK(hipMalloc(&xb, 16 * 16384 + 65536)); CK(hipMemset(xb, 3, 16 * 16384 + 65536));
            for (unsigned pitch : {8192u, 16384u}) {
                for (int i = 0; i < 6; ++i) {
                    hipLaunchKernelGGL(xpat_k, dim3(NB), dim3(512), 0, 0, xb, pitch, 16, st, sink);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
                char label[128];
                snprintf(label, sizeof label, "xpat: 16 rows x 64 B per load, 16 loads per wave, row pitch %u B", pitch);
                report(label, h, NB, 4, n2);
                CK(hipEventRecord(e0));
                for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(xpat_k, dim3(NB), dim3(512), 0, 0, xb, pitch, 16, st, sink);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("   back to back: %.2f us/launch\n", ms * 1e3 / 200);
            }
            for (int mode : {0, 1, 2, 3}) {
                for (int i = 0; i < 6; ++i) {
                    hipLaunchKernelGGL(xpat2_k, dim3(NB), dim3(512), 0, 0, xb, 8192u, mode, st, sink);
                    CK(hipDeviceSynchronize());
                }
                CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
                report(mode == 0 ? "xpat2: 8 rows x 128 B per load, 16 loads per wave, row pitch 8192 B"
                       : mode == 1 ? "xpat2: 4 rows x 256 B per load, 16 loads per wave, row pitch 8192 B"
                       : mode == 2 ? "xpat2: 8 rows x 128 B, chunk rotated by block" : "xpat2: 8 rows x 128 B, chunk and line rotated by block", h, NB, 4, n2);
                CK(hipEventRecord(e0));
                for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(xpat2_k, dim3(NB), dim3(512), 0, 0, xb, 8192u, mode, st, sink);
                CK(hipEventRecord(e1));
                CK(hipDeviceSynchronize());
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("   back to back: %.2f us/launch\n", ms * 1e3 / 200);
            }
            CK(hipFree(xb));
        }
        for (auto p : big) CK(hipFree(p));
    }
    return 0;
}
