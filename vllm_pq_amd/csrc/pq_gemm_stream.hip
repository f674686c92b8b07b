// Host side of the streaming gather+MFMA kernel (format checks, dtype dispatch) and the load-time
// re-layout of the codes into the kernel-native PACKED order.
//
// PACKED layout: for every 16-row fragment f and every chunk c of CG input groups, one 1-KiB piece at
// byte offset ((f * nchunks + c) * 64 + lane) * 16; lane = (r, q) with r = row within the fragment
// (lane & 15) and q = k quad (lane >> 4).  Code position (s*CPS + u)*M + m of the lane's 16 bytes is
// codes[16 f + r][c*CG + 4*CPS*s + CPS*q + u][m] (zero past G), so that MFMA k-step s of the chunk
// covers the 32 consecutive input features starting at (c*CG)*d + 32 s.
#include "pq_common.cuh"
#include "pq_kernels.h"
#include "pq_stream_geom.cuh"

namespace pq {

static int replication_for(int d, int K, int M) {
    return d == 8 ? stream_replication<8>(K, M) : d == 4 ? stream_replication<4>(K, M) : stream_replication<2>(K, M);
}

bool stream_format_supported(int d, int K, int M, int code_bytes) {
    if (!(d == 8 || d == 4 || d == 2)) return false;
    if (!(code_bytes == 1 || code_bytes == 2)) return false;
    if (!(M == 1 || (M == 2 && d == 8 && code_bytes == 1))) return false;
    if (K < 2 || (K & (K - 1))) return false;
    if (replication_for(d, K, M) == 0) return false;
    if (((size_t)M * K * d * 2) % 16) return false;   // every partition's codebooks stay 16-byte aligned
    return true;
}

static bool x_ok(const GemmArgs &a) {
    return !(reinterpret_cast<uintptr_t>(a.x) & 15) && (a.x_stride % 8 == 0) &&
           !(reinterpret_cast<uintptr_t>(a.codebooks) & 15);
}

bool gemm_stream_supported(const GemmArgs &a) {
    if (!stream_format_supported(a.d, a.K, a.M, a.code_bytes) || !x_ok(a)) return false;
    if (a.packed) return !(reinterpret_cast<uintptr_t>(a.codes) & 15);
    const int gpl = (16 / a.code_bytes) / a.M;
    return a.G % gpl == 0 && !(reinterpret_cast<uintptr_t>(a.codes) & 15);
}


hipError_t launch_gemm_stream_bf16(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gemm_stream_fp16(const GemmArgs &a, hipStream_t stream);

hipError_t launch_gemm_stream(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? launch_gemm_stream_bf16(a, stream) : launch_gemm_stream_fp16(a, stream);
}

// ---- re-layout ------------------------------------------------------------------------------------
bool packable(const Parts &parts, int d, int K, int M, int code_bytes) {
    if (!stream_format_supported(d, K, M, code_bytes)) return false;
    for (int p = 0; p < parts.num; ++p)
        if (parts.size[p] % 16) return false;   // fragments must not straddle two codebooks
    return true;
}

size_t packed_bytes(int N, int G, int M, int code_bytes) {
    const int cg = chunk_groups(code_bytes, M);
    const size_t nchunks = (size_t)(G + cg - 1) / cg;
    return (size_t)(N / 16) * nchunks * 1024;
}

template <typename CodeT>
__global__ __launch_bounds__(256) void repack_kernel(const CodeT *__restrict__ codes, int N, int G, int M, int d,
                                                     int nchunks, CodeT *__restrict__ packed) {
    constexpr int CPL = 16 / sizeof(CodeT);
    const int gpl = CPL / M;          // groups per lane
    const int cg = 4 * gpl;           // groups per chunk
    const int cps = 8 / d;            // groups per k-step per lane
    const long piece = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long npieces = (long)(N / 16) * nchunks;
    if (piece >= npieces) return;
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int f = (int)(piece / nchunks), c = (int)(piece % nchunks);
    const long n = (long)f * 16 + r;
    CodeT out[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const int m = i % M;
        const int gi = i / M;                   // group slot within the lane: s*cps + u
        const int s = gi / cps, u = gi % cps;
        const int g = c * cg + 4 * cps * s + cps * q + u;
        out[i] = g < G ? codes[(n * G + g) * M + m] : (CodeT)0;
    }
    CodeT *dst = packed + (piece * 64 + lane) * CPL;
#pragma unroll
    for (int i = 0; i < CPL; ++i) dst[i] = out[i];
}

hipError_t launch_repack(const void *codes, int N, int G, int M, int d, int code_bytes, void *packed,
                         hipStream_t stream) {
    const int cg = chunk_groups(code_bytes, M);
    const int nchunks = (G + cg - 1) / cg;
    const long npieces = (long)(N / 16) * nchunks;
    if (npieces == 0) return hipSuccess;
    const dim3 grid((unsigned)((npieces + 3) / 4));
    if (code_bytes == 1)
        hipLaunchKernelGGL(repack_kernel<int8_t>, grid, dim3(256), 0, stream, static_cast<const int8_t *>(codes), N, G,
                           M, d, nchunks, static_cast<int8_t *>(packed));
    else
        hipLaunchKernelGGL(repack_kernel<int16_t>, grid, dim3(256), 0, stream, static_cast<const int16_t *>(codes), N,
                           G, M, d, nchunks, static_cast<int16_t *>(packed));
    return hipGetLastError();
}

}  // namespace pq
