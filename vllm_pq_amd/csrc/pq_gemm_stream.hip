// Fused codebook gather + MFMA GEMM for small token counts (decode) on gfx950.
//
// y[t][n] = scale[n] * sum_k x[t][k] * W[n][k] + bias[n],   W[n][8g..8g+7] = centroid(code[n][g])
//
// Shape of the computation.  The reconstructed weight is the A operand of
// v_mfma_f32_16x16x32_{bf16,f16}: lane l supplies A[row l&15][k = 8(l>>4) .. +7], i.e. eight
// consecutive k of ONE weight row -- exactly one d=8 centroid (two d=4, four d=2).  So a lane's
// A fragment is read straight out of the LDS codebook with one ds_read_b128 at the address its code
// byte selects: no shuffle, no staging of W.  The activations are the B operand (lane l supplies
// x[t = l&15][the same 8 k]) and stay in registers: they are loaded once per k-chunk and reused by
// every weight fragment of the tile.  A 16x16 accumulator holds y^T for 16 rows x 16 tokens.
//
// Work split.  A block owns NF fragments (16*NF weight rows of one output partition) and all of K;
// its waves split K in chunks (one chunk = what a lane's 16-byte code load covers, times 4 lane
// quads), each wave streaming its codes from HBM with 16-byte loads.  Partial accumulators of the
// waves are summed in a fixed order through LDS (deterministic), then scaled, biased and stored.
//
// LDS codebook image.  Entry e is stored R times back to back; lane l reads copy l mod R.  With
// R = 16 (d=8), 32 (d=4, d=2) every lane of an LDS lane group reads a different bank slot whatever
// the codes are, so the gather runs at the conflict-free ds_read rate.
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

constexpr int ST_MAX_LDS_CB = 64 * 1024;

template <int D>
struct StreamGeom {
    static constexpr int EB = 2 * D;              // bytes per codebook entry
    static constexpr int RMAX = D == 8 ? 16 : 32; // copies for a conflict-free gather
    static constexpr int RMIN = 16 / EB;          // image is written in 16-byte slots
};

// Replication factor: largest power of two <= RMAX with M*K*EB*R <= 64 KiB (0: does not fit).
template <int D>
__host__ __device__ inline int stream_replication(int K, int M) {
    int r = StreamGeom<D>::RMAX;
    while (r >= StreamGeom<D>::RMIN && (size_t)M * K * StreamGeom<D>::EB * r > (size_t)ST_MAX_LDS_CB) r >>= 1;
    return r >= StreamGeom<D>::RMIN ? r : 0;
}

template <typename AT, int D>
__device__ inline typename Act<AT>::vec8 gather_frag(const unsigned char *cb_lds, const uint32_t (&idx)[8 / D],
                                                      uint32_t shift, uint32_t lane_off) {
    typedef typename Act<AT>::vec8 vec8;
    if constexpr (D == 8) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(cb_lds + ((idx[0] << shift) + lane_off));
        return __builtin_bit_cast(vec8, v);
    } else if constexpr (D == 4) {
        const u32x2 lo = *reinterpret_cast<const u32x2 *>(cb_lds + ((idx[0] << shift) + lane_off));
        const u32x2 hi = *reinterpret_cast<const u32x2 *>(cb_lds + ((idx[1] << shift) + lane_off));
        u32x4 v;
        v.x = lo.x; v.y = lo.y; v.z = hi.x; v.w = hi.y;
        return __builtin_bit_cast(vec8, v);
    } else {
        u32x4 v;
        v.x = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[0] << shift) + lane_off));
        v.y = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[1] << shift) + lane_off));
        v.z = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[2] << shift) + lane_off));
        v.w = *reinterpret_cast<const uint32_t *>(cb_lds + ((idx[3] << shift) + lane_off));
        return __builtin_bit_cast(vec8, v);
    }
}

// Code `i` of the 16 bytes a lane loaded (CB = bytes per code), reduced mod K = low nbits.
template <int CB>
__device__ inline uint32_t code_at(const u32x4 &w, int i, uint32_t nbits) {
    if constexpr (CB == 1) {
        return __builtin_amdgcn_ubfe(w[i >> 2], (uint32_t)((i & 3) * 8), nbits);
    } else {
        return __builtin_amdgcn_ubfe(w[i >> 1], (uint32_t)((i & 1) * 16), nbits);
    }
}

template <typename AT, typename CodeT, int D, int M, int NF>
__global__ __launch_bounds__(512) void gemm_stream_kernel(
    const uint16_t *__restrict__ x, int T, int64_t x_stride, const CodeT *__restrict__ codes,
    const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales, const uint16_t *__restrict__ bias,
    Parts parts, int G, int K, int nbits, int R, uint16_t *__restrict__ y, int64_t y_stride) {
    typedef typename Act<AT>::vec8 vec8;
    constexpr int CB = sizeof(CodeT);
    constexpr int EB = StreamGeom<D>::EB;
    constexpr int CPL = 16 / CB;            // codes per lane per 16-byte load
    constexpr int GPL = CPL / M;            // input groups per lane per load
    constexpr int CG = 4 * GPL;             // input groups per chunk (4 lane quads)
    constexpr int KS = GPL * D / 8;         // MFMA k-steps (32 k each) per chunk
    constexpr int CPS = 8 / D;              // groups per k-step per lane
    static_assert(CPL % M == 0 && KS >= 1, "format does not fill an MFMA k-step");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    const int r16 = lane & 15;   // weight row within a fragment (A) / token within the chunk (B)
    const int q = lane >> 4;     // k quad

    const TileLoc loc = locate_tile(parts, blockIdx.x, NF * 16);
    const int t0 = blockIdx.y * 16;
    const int row_end = loc.row0 + loc.nrows;

    f32x4 acc[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = (G + CG - 1) / CG;
    const bool tok_ok = (t0 + r16) < T;
    const uint16_t *xrow = x + (size_t)(t0 + (tok_ok ? r16 : 0)) * x_stride;

    // ---- first chunk's global loads go out before the codebook image is built ----------------
    u32x4 cw[NF];
    vec8 xf[KS];
    auto load_chunk = [&](int c) {
        const int g0 = c * CG + GPL * q;           // first input group of this lane
        const bool g_ok = g0 < G;                  // G % GPL == 0: all-or-nothing per lane
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            int n = loc.row0 + 16 * i + r16;
            n = n < row_end ? n : row_end - 1;     // clamp: rows past the tile are never stored
            cw[i] = g_ok ? *reinterpret_cast<const u32x4 *>(codes + ((size_t)n * G + g0) * M) : u32x4{0, 0, 0, 0};
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            u32x4 v{0, 0, 0, 0};
            if (g_ok && tok_ok) v = *reinterpret_cast<const u32x4 *>(xrow + (size_t)g0 * D + 8 * s);
            xf[s] = __builtin_bit_cast(vec8, v);
        }
    };
    if (wave < nchunks) load_chunk(wave);

    // ---- replicated codebook image: 16-byte slot j holds entry (16 j) / (EB R), repeated -------
    {
        const unsigned char *cb_g = reinterpret_cast<const unsigned char *>(codebooks) + (size_t)loc.part * M * K * EB;
        const int slots = (M * K * EB * R) / 16;
        const int per_entry = (EB * R) / 16;       // slots per entry (>= 1)
        for (int j = tid; j < slots; j += blockDim.x) {
            const int e = j / per_entry;           // entry index over the M codebooks
            u32x4 v;
            if constexpr (D == 8) {
                v = *reinterpret_cast<const u32x4 *>(cb_g + (size_t)e * 16);
            } else if constexpr (D == 4) {
                const u32x2 s2 = *reinterpret_cast<const u32x2 *>(cb_g + (size_t)e * 8);
                v.x = s2.x; v.y = s2.y; v.z = s2.x; v.w = s2.y;
            } else {
                const uint32_t s1 = *reinterpret_cast<const uint32_t *>(cb_g + (size_t)e * 4);
                v.x = s1; v.y = s1; v.z = s1; v.w = s1;
            }
            *reinterpret_cast<u32x4 *>(smem + (size_t)j * 16) = v;
        }
    }
    __syncthreads();

    const uint32_t shift = (uint32_t)__builtin_ctz((unsigned)(EB * R));   // log2 of the entry stride
    const uint32_t lane_off = (uint32_t)(lane & (R - 1)) * EB;
    const uint32_t cb_stride = (uint32_t)K * EB * R;                       // bytes per codebook image

    for (int c = wave; c < nchunks; c += nwaves) {
        if (c != wave) load_chunk(c);
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                vec8 a;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    uint32_t idx[CPS];
#pragma unroll
                    for (int u = 0; u < CPS; ++u) idx[u] = code_at<CB>(cw[i], (s * CPS + u) * M + m, (uint32_t)nbits);
                    const vec8 am = gather_frag<AT, D>(smem + m * cb_stride, idx, shift, lane_off);
                    // additive codebooks: the weight is the act-dtype-rounded sum of the entries, as the
                    // reference materialises it (embedding_bag output aqlm.py:60-64; __hadd2 in
                    // gemm_kernels.cu:169)
                    a = m == 0 ? am : Act<AT>::add8(a, am);
                }
                acc[i] = Act<AT>::mfma16(a, xf[s], acc[i]);
            }
        }
    }

    // ---- deterministic cross-wave sum through LDS (the codebook image is dead now) -------------
    __syncthreads();
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);   // [wave][NF][64]
#pragma unroll
    for (int i = 0; i < NF; ++i) red[(wave * NF + i) * 64 + lane] = acc[i];
    __syncthreads();
    for (int o = tid; o < NF * 64; o += blockDim.x) {
        const int i = o >> 6;
        const int l = o & 63;
        f32x4 sum = red[i * 64 + l];
        for (int w = 1; w < nwaves; ++w) {
            const f32x4 v = red[(w * NF + i) * 64 + l];
            sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        // accumulator map: column (token) = l & 15, row = (l >> 4) * 4 + reg
        const int t = t0 + (l & 15);
        const int nb = loc.row0 + 16 * i + (l >> 4) * 4;
        if (t < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = nb + r;
                if (n < row_end) {
                    const float s = Act<AT>::to_f32(scales[n]);
                    const float b = bias ? Act<AT>::to_f32(bias[n]) : 0.f;
                    y[(size_t)t * y_stride + n] = Act<AT>::from_f32(sum[r] * s + b);
                }
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------
static bool stream_format_ok(const GemmArgs &a) {
    if (!(a.d == 8 || a.d == 4 || a.d == 2)) return false;
    if (!(a.code_bytes == 1 || a.code_bytes == 2)) return false;
    if (!(a.M == 1 || (a.M == 2 && a.d == 8 && a.code_bytes == 1))) return false;
    if (a.K < 2 || (a.K & (a.K - 1))) return false;
    const int cpl = 16 / a.code_bytes;
    const int gpl = cpl / a.M;
    if (a.G % gpl) return false;
    int r = a.d == 8 ? stream_replication<8>(a.K, a.M) : a.d == 4 ? stream_replication<4>(a.K, a.M)
                                                                  : stream_replication<2>(a.K, a.M);
    if (r == 0) return false;
    if (reinterpret_cast<uintptr_t>(a.codes) & 15) return false;
    if (reinterpret_cast<uintptr_t>(a.codebooks) & 15) return false;
    if ((reinterpret_cast<uintptr_t>(a.x) & 15) || (a.x_stride % 8)) return false;
    if (((size_t)a.M * a.K * a.d * 2) % 16) return false;   // partition codebooks stay 16-byte aligned
    return true;
}

bool gemm_stream_supported(const GemmArgs &a) { return stream_format_ok(a); }

template <typename AT, typename CodeT, int D, int M, int NF>
static hipError_t launch_nf(const GemmArgs &a, hipStream_t stream) {
    const int R = stream_replication<D>(a.K, a.M);
    const int nbits = __builtin_ctz((unsigned)a.K);
    const int nwaves = 8;
    size_t lds = (size_t)a.M * a.K * 2 * D * R;
    const size_t red = (size_t)nwaves * NF * 64 * 16;
    if (red > lds) lds = red;
    const dim3 grid(tiles_in(a.parts, NF * 16), (a.T + 15) / 16);
    hipLaunchKernelGGL((gemm_stream_kernel<AT, CodeT, D, M, NF>), grid, dim3(nwaves * 64), lds, stream,
                       static_cast<const uint16_t *>(a.x), a.T, a.x_stride, static_cast<const CodeT *>(a.codes),
                       static_cast<const uint16_t *>(a.codebooks), static_cast<const uint16_t *>(a.scales),
                       static_cast<const uint16_t *>(a.bias), a.parts, a.G, a.K, nbits, R,
                       static_cast<uint16_t *>(a.y), a.y_stride);
    return hipGetLastError();
}

template <typename AT, typename CodeT, int D, int M>
static hipError_t launch_fmt(const GemmArgs &a, hipStream_t stream) {
    // Largest tile that still gives every CU a block; small T keeps NF low because the activations
    // are cheap to re-read and more blocks hide more latency.
    int nf = 1;
    if (tiles_in(a.parts, 64) >= 240) nf = 4;
    else if (tiles_in(a.parts, 32) >= 240) nf = 2;
    switch (nf) {
    case 4: return launch_nf<AT, CodeT, D, M, 4>(a, stream);
    case 2: return launch_nf<AT, CodeT, D, M, 2>(a, stream);
    default: return launch_nf<AT, CodeT, D, M, 1>(a, stream);
    }
}

template <typename AT>
static hipError_t launch_at(const GemmArgs &a, hipStream_t stream) {
    if (a.M == 2) return launch_fmt<AT, int8_t, 8, 2>(a, stream);
    if (a.code_bytes == 1) {
        switch (a.d) {
        case 8: return launch_fmt<AT, int8_t, 8, 1>(a, stream);
        case 4: return launch_fmt<AT, int8_t, 4, 1>(a, stream);
        default: return launch_fmt<AT, int8_t, 2, 1>(a, stream);
        }
    }
    switch (a.d) {
    case 8: return launch_fmt<AT, int16_t, 8, 1>(a, stream);
    case 4: return launch_fmt<AT, int16_t, 4, 1>(a, stream);
    default: return launch_fmt<AT, int16_t, 2, 1>(a, stream);
    }
}

hipError_t launch_gemm_stream(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? launch_at<bf16_t>(a, stream) : launch_at<fp16_t>(a, stream);
}

}  // namespace pq
