// Format-agnostic fused dequant + GEMM for gfx950: any sub-vector length d, any power-of-two K,
// any number of additive codebooks M, int8 or int16 codes, ragged shapes.  One wave per output row;
// lanes stride over the input groups, gather the centroid(s) from L2, and accumulate
// TB tokens at a time in fp32; a wave reduction finishes the dot products.  This is the fallback
// for shapes the MFMA kernels do not take; it is correct everywhere and fast nowhere.
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

constexpr int GEN_TB = 8;       // tokens per pass
constexpr int GEN_WAVES = 4;    // rows per block

template <typename AT, typename CodeT>
__global__ __launch_bounds__(GEN_WAVES * 64) void gemm_generic_kernel(
    const uint16_t *__restrict__ x, int T, int64_t x_stride, const CodeT *__restrict__ codes,
    const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales, const uint16_t *__restrict__ bias,
    Parts parts, int N, int G, int M, int K, int d, uint16_t *__restrict__ y, int64_t y_stride) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * GEN_WAVES + (threadIdx.x >> 6);
    if (n >= N) return;  // whole wave leaves together
    int p = 0;
    {
        int end = 0;
#pragma unroll
        for (int i = 0; i < PQ_MAX_PARTS; ++i) {
            if (i < parts.num) {
                end += parts.size[i];
                if (n >= end) p = i + 1;
            }
        }
    }
    const uint32_t kmask = (uint32_t)K - 1u;
    const uint16_t *cb = codebooks + (size_t)p * M * K * d;
    const float s = Act<AT>::to_f32(scales[n]);
    const float b = bias ? Act<AT>::to_f32(bias[n]) : 0.f;
    const CodeT *crow = codes + (size_t)n * G * M;

    for (int t0 = blockIdx.y * GEN_TB; t0 < T; t0 += gridDim.y * GEN_TB) {
        float acc[GEN_TB];
#pragma unroll
        for (int i = 0; i < GEN_TB; ++i) acc[i] = 0.f;
        for (int g = lane; g < G; g += 64) {
            for (int j = 0; j < d; ++j) {
                float w = 0.f;
                for (int m = 0; m < M; ++m)
                    w += Act<AT>::to_f32(cb[((size_t)m * K + code_index(crow[(size_t)g * M + m], kmask)) * d + j]);
                // the reference materialises the sum over codebooks in the act dtype (embedding_bag
                // output, aqlm.py:60-64; __hadd2 in its CUDA kernel): round it the same way
                if (M > 1) w = Act<AT>::to_f32(Act<AT>::from_f32(w));
                const int k = g * d + j;
#pragma unroll
                for (int i = 0; i < GEN_TB; ++i)
                    if (t0 + i < T) acc[i] += w * Act<AT>::to_f32(x[(size_t)(t0 + i) * x_stride + k]);
            }
        }
#pragma unroll
        for (int i = 0; i < GEN_TB; ++i) {
            float v = acc[i];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && t0 + i < T) y[(size_t)(t0 + i) * y_stride + n] = Act<AT>::from_f32(v * s + b);
        }
    }
}

template <typename AT, typename CodeT>
static hipError_t launch_t(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    dim3 grid((a.N + GEN_WAVES - 1) / GEN_WAVES, 1);
    int ty = (a.T + GEN_TB - 1) / GEN_TB;
    grid.y = ty > 64 ? 64 : ty;
    hipLaunchKernelGGL((gemm_generic_kernel<AT, CodeT>), grid, dim3(GEN_WAVES * 64), 0, stream,
                       static_cast<const uint16_t *>(a.x), a.T, a.x_stride, static_cast<const CodeT *>(a.codes),
                       static_cast<const uint16_t *>(a.codebooks), static_cast<const uint16_t *>(a.scales),
                       static_cast<const uint16_t *>(a.bias), a.parts, a.N, a.G, a.M, a.K, a.d,
                       static_cast<uint16_t *>(a.y), a.y_stride);
    return hipGetLastError();
}

hipError_t launch_gemm_generic(const GemmArgs &a, hipStream_t stream) {
    if (a.dtype == PQ_DTYPE_BF16)
        return a.code_bytes == 1 ? launch_t<bf16_t, int8_t>(a, stream) : launch_t<bf16_t, int16_t>(a, stream);
    return a.code_bytes == 1 ? launch_t<fp16_t, int8_t>(a, stream) : launch_t<fp16_t, int16_t>(a, stream);
}

}  // namespace pq
