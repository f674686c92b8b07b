// The element-wise steps either side of the PQ linears in a decoder layer, as HBM-streaming kernels:
//   silu_and_mul         out[t][j] = silu(x[t][j]) * x[t][I + j]              (after gate_up_proj)
//   rms_norm             out[t][:] = x[t][:] / rms(x[t]) * w                   (before qkv_proj / gate_up_proj)
//   fused_add_rms_norm   z = x + residual; residual <- z; x <- z / rms(z) * w
// Numerics follow the reference's PyTorch-native definitions (SiluAndMul.forward_native,
// vllm/model_executor/layers/activation.py:73-76; RMSNorm.forward_native, layers/layernorm.py:38-74):
// silu in fp32 rounded to the act dtype, then the product rounded; the sum and the mean of squares in fp32
// (the residual written back is the rounded sum, the variance uses the unrounded one), z * rsqrt(var + eps)
// rounded to the act dtype, then multiplied by the weight and rounded again.
//
// All three move every byte once with 16-byte accesses: one thread owns 8 consecutive elements; a token row
// of the norms stays in registers between the sum of squares and the scaling (rows of up to 16384
// elements), so x is read once.  One block per token row for the norms, a 2-D grid for the gate.
#include <hip/hip_runtime.h>

#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

constexpr int EW_THREADS = 256;

template <typename AT>
__device__ inline float silu_round(float g) {
    // x / (1 + exp(-x)) in fp32 (csrc/activation_kernels.cu:35-38), rounded to the act dtype as F.silu returns it
    return Act<AT>::to_f32(Act<AT>::from_f32(g / (1.0f + expf(-g))));
}

template <typename AT>
__global__ __launch_bounds__(EW_THREADS) void silu_and_mul_vec_kernel(const uint16_t *__restrict__ x, int64_t x_stride,
                                                                       int T, int I, uint16_t *__restrict__ out,
                                                                       int64_t out_stride) {
    const int v = blockIdx.x * EW_THREADS + threadIdx.x;   // 8-element piece of the row
    if (v * 8 >= I) return;
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        const uint16_t *row = x + (size_t)t * x_stride;
        const u32x4 g = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(row + v * 8));
        const u32x4 u = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(row + I + v * 8));
        u32x4 o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t packed = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float gf = Act<AT>::to_f32((uint16_t)(g[w] >> (16 * h)));
                const float uf = Act<AT>::to_f32((uint16_t)(u[w] >> (16 * h)));
                packed |= (uint32_t)Act<AT>::from_f32(silu_round<AT>(gf) * uf) << (16 * h);
            }
            o[w] = packed;
        }
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out + (size_t)t * out_stride + v * 8));
    }
}

// any I, any alignment: one element per thread
template <typename AT>
__global__ __launch_bounds__(EW_THREADS) void silu_and_mul_scalar_kernel(const uint16_t *__restrict__ x,
                                                                          int64_t x_stride, int T, int I,
                                                                          uint16_t *__restrict__ out,
                                                                          int64_t out_stride) {
    const int j = blockIdx.x * EW_THREADS + threadIdx.x;
    if (j >= I) return;
    for (int t = blockIdx.y; t < T; t += gridDim.y) {
        const uint16_t *row = x + (size_t)t * x_stride;
        const float gf = Act<AT>::to_f32(row[j]), uf = Act<AT>::to_f32(row[I + j]);
        out[(size_t)t * out_stride + j] = Act<AT>::from_f32(silu_round<AT>(gf) * uf);
    }
}

// sum over the block, the same value in every thread; fixed order (lane tree, then the waves in order)
template <int THREADS>
__device__ inline float block_sum(float v, float *s_part) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_part[wave] = v;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) tot += s_part[w];
    return tot;
}

// One block of THREADS per token row, NV 8-element pieces per thread kept in registers (rows up to
// NV * THREADS * 8 elements): as many threads as the row has pieces, so that the whole row is requested at once.
// ADD: x += residual first.
template <typename AT, int THREADS, int NV, bool ADD>
__global__ __launch_bounds__(THREADS) void rms_norm_kernel(const uint16_t *x, int64_t x_stride,
                                                               uint16_t *residual, int64_t res_stride,
                                                               const uint16_t *__restrict__ weight, float eps, int H,
                                                               uint16_t *out, int64_t out_stride) {   // out may be x
    __shared__ float s_part[THREADS / 64];
    const int t = blockIdx.x;
    const uint16_t *xr = x + (size_t)t * x_stride;
    uint16_t *rr = ADD ? residual + (size_t)t * res_stride : nullptr;
    const int nvec = H / 8;
    float z[NV][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = i * THREADS + threadIdx.x;
        if (v < nvec) {
            const u32x4 a = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(xr + v * 8));
            u32x4 r{0u, 0u, 0u, 0u}, zr;
            if constexpr (ADD) r = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rr + v * 8));
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t packed = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float f = Act<AT>::to_f32((uint16_t)(a[w] >> (16 * h)));
                    if constexpr (ADD) {
                        f += Act<AT>::to_f32((uint16_t)(r[w] >> (16 * h)));
                        packed |= (uint32_t)Act<AT>::from_f32(f) << (16 * h);
                    }
                    z[i][2 * w + h] = f;
                    ss += f * f;
                }
                zr[w] = packed;
            }
            if constexpr (ADD) __builtin_nontemporal_store(zr, reinterpret_cast<u32x4 *>(rr + v * 8));
        }
    }
    const float rs = rsqrtf(block_sum<THREADS>(ss, s_part) / (float)H + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int v = i * THREADS + threadIdx.x;
        if (v < nvec) {
            const u32x4 wv = *reinterpret_cast<const u32x4 *>(weight + v * 8);
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t packed = 0;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float n = Act<AT>::to_f32(Act<AT>::from_f32(rounded_f32(z[i][2 * w + h] * rs)));
                    packed |= (uint32_t)Act<AT>::from_f32(n * Act<AT>::to_f32((uint16_t)(wv[w] >> (16 * h)))) << (16 * h);
                }
                o[w] = packed;
            }
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out + (size_t)t * out_stride + v * 8));
        }
    }
}

// any H, any alignment: two passes over the row, element by element (the second pass re-reads what the
// first one read or wrote: L2 hits)
template <typename AT, bool ADD>
__global__ __launch_bounds__(EW_THREADS) void rms_norm_scalar_kernel(const uint16_t *x, int64_t x_stride,
                                                                      uint16_t *residual, int64_t res_stride,
                                                                      const uint16_t *__restrict__ weight, float eps,
                                                                      int H, uint16_t *out, int64_t out_stride) {
    __shared__ float s_part[EW_THREADS / 64];
    extern __shared__ float s_row[];   // the fp32 row (H floats)
    const int t = blockIdx.x;
    const uint16_t *xr = x + (size_t)t * x_stride;
    uint16_t *rr = ADD ? residual + (size_t)t * res_stride : nullptr;
    float ss = 0.f;
    for (int j = threadIdx.x; j < H; j += EW_THREADS) {
        float f = Act<AT>::to_f32(xr[j]);
        if constexpr (ADD) {
            f += Act<AT>::to_f32(rr[j]);
            rr[j] = Act<AT>::from_f32(f);
        }
        s_row[j] = f;
        ss += f * f;
    }
    const float rs = rsqrtf(block_sum<EW_THREADS>(ss, s_part) / (float)H + eps);
    for (int j = threadIdx.x; j < H; j += EW_THREADS) {
        const float n = Act<AT>::to_f32(Act<AT>::from_f32(rounded_f32(s_row[j] * rs)));
        out[(size_t)t * out_stride + j] = Act<AT>::from_f32(n * Act<AT>::to_f32(weight[j]));
    }
}

static bool aligned16(const void *p) { return !(reinterpret_cast<uintptr_t>(p) & 15); }

bool rms_norm_vectorised(const ElementwiseArgs &a) {
    return a.H % 8 == 0 && a.H <= 8 * 2048 && a.x_stride % 8 == 0 && a.out_stride % 8 == 0 && aligned16(a.x) &&
           aligned16(a.out) && aligned16(a.weight) && (!a.residual || (a.res_stride % 8 == 0 && aligned16(a.residual)));
}

template <typename AT>
static hipError_t silu_and_mul_at(const ElementwiseArgs &a, hipStream_t stream) {
    const uint16_t *x = static_cast<const uint16_t *>(a.x);
    uint16_t *out = static_cast<uint16_t *>(a.out);
    const unsigned gy = (unsigned)(a.T < 65535 ? a.T : 65535);
    const bool vec = a.H % 8 == 0 && a.x_stride % 8 == 0 && a.out_stride % 8 == 0 && aligned16(a.x) && aligned16(a.out);
    if (vec) {
        const dim3 grid((unsigned)((a.H / 8 + EW_THREADS - 1) / EW_THREADS), gy);
        hipLaunchKernelGGL(silu_and_mul_vec_kernel<AT>, grid, dim3(EW_THREADS), 0, stream, x, a.x_stride, a.T, a.H, out,
                           a.out_stride);
    } else {
        const dim3 grid((unsigned)((a.H + EW_THREADS - 1) / EW_THREADS), gy);
        hipLaunchKernelGGL(silu_and_mul_scalar_kernel<AT>, grid, dim3(EW_THREADS), 0, stream, x, a.x_stride, a.T, a.H,
                           out, a.out_stride);
    }
    return hipGetLastError();
}

hipError_t launch_silu_and_mul(const ElementwiseArgs &a, hipStream_t stream) {
    if (a.T == 0 || a.H == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? silu_and_mul_at<bf16_t>(a, stream) : silu_and_mul_at<fp16_t>(a, stream);
}

template <typename AT, bool ADD>
static hipError_t rms_norm_at(const ElementwiseArgs &a, hipStream_t stream) {
    const uint16_t *x = static_cast<const uint16_t *>(a.x);
    uint16_t *res = static_cast<uint16_t *>(a.residual);
    const uint16_t *w = static_cast<const uint16_t *>(a.weight);
    uint16_t *out = static_cast<uint16_t *>(a.out);
    const dim3 grid((unsigned)a.T), block(EW_THREADS);
    if (!rms_norm_vectorised(a)) {
        hipLaunchKernelGGL((rms_norm_scalar_kernel<AT, ADD>), grid, block, (size_t)a.H * 4, stream, x, a.x_stride, res,
                           a.res_stride, w, a.eps, a.H, out, a.out_stride);
        return hipGetLastError();
    }
    const int nvec = a.H / 8;
#define PQ_RMS(THREADS, NV)                                                                                            \
    hipLaunchKernelGGL((rms_norm_kernel<AT, THREADS, NV, ADD>), grid, dim3(THREADS), 0, stream, x, a.x_stride, res,    \
                       a.res_stride, w, a.eps, a.H, out, a.out_stride)
    if (nvec <= 128) PQ_RMS(128, 1);
    else if (nvec <= 256) PQ_RMS(256, 1);
    else if (nvec <= 512) PQ_RMS(512, 1);
    else if (nvec <= 1024) PQ_RMS(1024, 1);
    else PQ_RMS(1024, 2);
#undef PQ_RMS
    return hipGetLastError();
}

hipError_t launch_rms_norm(const ElementwiseArgs &a, hipStream_t stream) {
    if (a.T == 0 || a.H == 0) return hipSuccess;
    const bool add = a.residual != nullptr;
    if (a.dtype == PQ_DTYPE_BF16) return add ? rms_norm_at<bf16_t, true>(a, stream) : rms_norm_at<bf16_t, false>(a, stream);
    return add ? rms_norm_at<fp16_t, true>(a, stream) : rms_norm_at<fp16_t, false>(a, stream);
}

}  // namespace pq
