// Dense-weight reconstruction (codebook gather) for gfx950.
//
// W[n][g*d + j] = sum_m codebooks[p(n)*M + m][ codes[n][g][m] mod K ][j]   (* scales[n])
//
// HBM-bound on the output stream (N*Kin*2 B written against N*G*M*code_bytes read): each thread
// produces one 16-byte piece of an output row per step, so a wave writes 1 KiB contiguous; the
// codebooks of the tile's partition sit in LDS when they fit (M*K*d*2 B <= 64 KiB), else they are
// gathered from L2.
#include "pq_common.cuh"
#include "pq_kernels.h"

namespace pq {

constexpr int DQ_THREADS = 256;
constexpr int DQ_ROWS = 8;                 // rows per tile
constexpr int DQ_LDS_BYTES = 64 * 1024;    // codebook budget

// D = sub-vector length in elements, restricted to divisors of 8 so that one thread's 8 output
// elements are 8/D whole groups.  USE_LDS: codebooks staged in LDS.
template <typename AT, typename CodeT, int D, bool USE_LDS>
__global__ __launch_bounds__(DQ_THREADS) void dequant_vec8_kernel(
    const CodeT *__restrict__ codes, const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales,
    Parts parts, int G, int M, int K, uint16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *cb_lds = reinterpret_cast<uint16_t *>(smem);

    const TileLoc loc = locate_tile(parts, blockIdx.x, DQ_ROWS);
    const int Kin = G * D;
    const uint32_t kmask = (uint32_t)K - 1u;
    const uint16_t *cb_g = codebooks + (size_t)loc.part * M * K * D;

    if (USE_LDS) {
        const int n16 = (M * K * D * 2) / 16;  // M*K*D*2 is a multiple of 16 for K >= 8/D ... checked on the host
        const u32x4 *src = reinterpret_cast<const u32x4 *>(cb_g);
        u32x4 *dst = reinterpret_cast<u32x4 *>(cb_lds);
        for (int i = threadIdx.x; i < n16; i += DQ_THREADS) dst[i] = src[i];
        __syncthreads();
    }
    const uint16_t *cb = USE_LDS ? cb_lds : cb_g;

    constexpr int GPT = 8 / D;            // groups per thread-step
    const int pieces_per_row = Kin / 8;   // 16-byte pieces
    const int total = loc.nrows * pieces_per_row;
    for (int i = threadIdx.x; i < total; i += DQ_THREADS) {
        const int r = i / pieces_per_row;
        const int c = i - r * pieces_per_row;
        const int n = loc.row0 + r;
        const CodeT *crow = codes + ((size_t)n * G + (size_t)c * GPT) * M;
        uint16_t h[8];
        if (M == 1) {
#pragma unroll
            for (int gg = 0; gg < GPT; ++gg) {
                const uint32_t idx = code_index(crow[gg], kmask);
                const uint16_t *e = cb + (size_t)idx * D;
#pragma unroll
                for (int j = 0; j < D; ++j) h[gg * D + j] = e[j];
            }
        } else {
#pragma unroll
            for (int gg = 0; gg < GPT; ++gg) {
                float acc[D];
#pragma unroll
                for (int j = 0; j < D; ++j) acc[j] = 0.f;
                for (int m = 0; m < M; ++m) {
                    const uint32_t idx = code_index(crow[gg * M + m], kmask);
                    const uint16_t *e = cb + ((size_t)m * K + idx) * D;
#pragma unroll
                    for (int j = 0; j < D; ++j) acc[j] += Act<AT>::to_f32(e[j]);
                }
#pragma unroll
                for (int j = 0; j < D; ++j) h[gg * D + j] = Act<AT>::from_f32(acc[j]);
            }
        }
        if (scales) {
            const float s = Act<AT>::to_f32(scales[n]);
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = Act<AT>::from_f32(Act<AT>::to_f32(h[j]) * s);
        }
        u32x4 v;
        v.x = h[0] | ((uint32_t)h[1] << 16);
        v.y = h[2] | ((uint32_t)h[3] << 16);
        v.z = h[4] | ((uint32_t)h[5] << 16);
        v.w = h[6] | ((uint32_t)h[7] << 16);
        *reinterpret_cast<u32x4 *>(out + (size_t)n * Kin + (size_t)c * 8) = v;
    }
}

// d = 8, one codebook: the bulk of the traffic of this format family.  A block owns DQ8_ROWS rows of one partition;
// the codebook sits in LDS R times (lane l reads copy l mod R: no bank conflict whatever the codes are); a wave
// step covers 256 consecutive groups of a row: four code loads of 64 consecutive codes, four gathers, four 16-byte
// stores of one contiguous KiB each, non-temporal (the output is a stream nobody in this kernel reads back).
constexpr int DQ8_ROWS = 16;
constexpr int DQ8_MAX_R = 4;     // copies of the codebook in LDS: 16 KiB at K = 256, eight blocks per CU
template <typename AT, typename CodeT>
__global__ __launch_bounds__(DQ_THREADS) void dequant_d8_kernel(
    const CodeT *__restrict__ codes, const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales,
    TileMap tmap, int G, int K, int R, uint16_t *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const TileLoc loc = locate_tile(tmap, blockIdx.x, DQ8_ROWS);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t kmask = (uint32_t)K - 1u;
    {   // image: entry e, copy r at (e * R + r) * 16
        const u32x4 *src = reinterpret_cast<const u32x4 *>(codebooks + (size_t)loc.part * K * 8);
        u32x4 *dst = reinterpret_cast<u32x4 *>(smem);
        const int slots = K * R;
        const int rshift = __builtin_ctz((unsigned)R);
        for (int i = tid; i < slots; i += DQ_THREADS) dst[i] = src[i >> rshift];
        __syncthreads();
    }
    const uint32_t lane_off = (uint32_t)(lane & (R - 1)) * 16u;
    const uint32_t eshift = 4 + __builtin_ctz((unsigned)R);
    const int Kin = G * 8;
    const int spans = (G + 255) / 256;                  // 256-group spans per row
    const int total = loc.nrows * spans;
    // the codes of the wave's next step are requested before the current one is gathered and stored
    auto load_step = [&](int w, uint32_t (&c4)[4], float &s) {
        const int wc = w < total ? w : total - 1;           // past the end: re-read the last step, never stored
        const int r = wc / spans, sp = wc - r * spans;
        const int n = loc.row0 + r;
        const int g0 = sp * 256;
        const CodeT *crow = codes + (size_t)n * G + g0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int g = 64 * j + lane;                    // store j writes pieces 64 j .. 64 j + 63: one contiguous KiB
            c4[j] = code_index(crow[g0 + g < G ? g : G - 1 - g0], kmask);   // clamped at the row's end, not stored
        }
        s = scales ? Act<AT>::to_f32(scales[n]) : 1.f;
    };
    uint32_t cur[4], nxt[4];
    float s_cur, s_nxt;
    load_step(wave, cur, s_cur);
    for (int w = wave; w < total; w += DQ_THREADS / 64) {
        load_step(w + DQ_THREADS / 64, nxt, s_nxt);
        const int r = w / spans, sp = w - r * spans;
        const int n = loc.row0 + r;
        const int g0 = sp * 256;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            u32x4 e = *reinterpret_cast<const u32x4 *>(smem + ((size_t)cur[j] << eshift) + lane_off);
            if (scales) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint16_t lo = Act<AT>::from_f32(Act<AT>::to_f32((uint16_t)e[k]) * s_cur);
                    const uint16_t hi = Act<AT>::from_f32(Act<AT>::to_f32((uint16_t)(e[k] >> 16)) * s_cur);
                    e[k] = lo | ((uint32_t)hi << 16);
                }
            }
            const int g = g0 + 64 * j + lane;
            if (g < G) __builtin_nontemporal_store(e, reinterpret_cast<u32x4 *>(out + (size_t)n * Kin + (size_t)g * 8));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nxt[j];
        s_cur = s_nxt;
    }
}

// Any d, any Kin: one output element per thread-step.
template <typename AT, typename CodeT>
__global__ __launch_bounds__(DQ_THREADS) void dequant_elem_kernel(
    const CodeT *__restrict__ codes, const uint16_t *__restrict__ codebooks, const uint16_t *__restrict__ scales,
    Parts parts, int G, int M, int K, int d, uint16_t *__restrict__ out) {
    const TileLoc loc = locate_tile(parts, blockIdx.x, DQ_ROWS);
    const int Kin = G * d;
    const uint32_t kmask = (uint32_t)K - 1u;
    const uint16_t *cb = codebooks + (size_t)loc.part * M * K * d;
    const int total = loc.nrows * Kin;
    for (int i = threadIdx.x; i < total; i += DQ_THREADS) {
        const int r = i / Kin;
        const int k = i - r * Kin;
        const int g = k / d;
        const int j = k - g * d;
        const int n = loc.row0 + r;
        const CodeT *crow = codes + ((size_t)n * G + g) * M;
        uint16_t h;
        if (M == 1) {
            h = cb[(size_t)code_index(crow[0], kmask) * d + j];
        } else {
            float acc = 0.f;
            for (int m = 0; m < M; ++m)
                acc += Act<AT>::to_f32(cb[((size_t)m * K + code_index(crow[m], kmask)) * d + j]);
            h = Act<AT>::from_f32(acc);
        }
        if (scales) h = Act<AT>::from_f32(Act<AT>::to_f32(h) * Act<AT>::to_f32(scales[n]));
        out[(size_t)n * Kin + k] = h;
    }
}

template <typename AT, typename CodeT>
static hipError_t launch_dequant_t(const DequantArgs &a, hipStream_t stream) {
    const int tiles = tiles_in(a.parts, DQ_ROWS);
    if (tiles == 0) return hipSuccess;
    const CodeT *codes = static_cast<const CodeT *>(a.codes);
    const uint16_t *cb = static_cast<const uint16_t *>(a.codebooks);
    const uint16_t *sc = static_cast<const uint16_t *>(a.scales);
    uint16_t *out = static_cast<uint16_t *>(a.out);
    const size_t cb_bytes = (size_t)a.M * a.K * a.d * 2;
    const bool vec_ok = (a.d == 1 || a.d == 2 || a.d == 4 || a.d == 8) && ((a.G * a.d) % 8 == 0) &&
                        ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (!vec_ok) {
        hipLaunchKernelGGL((dequant_elem_kernel<AT, CodeT>), dim3(tiles), dim3(DQ_THREADS), 0, stream, codes, cb, sc,
                           a.parts, a.G, a.M, a.K, a.d, out);
        return hipGetLastError();
    }
    const bool use_lds = cb_bytes <= (size_t)DQ_LDS_BYTES && (cb_bytes % 16 == 0) &&
                         ((reinterpret_cast<uintptr_t>(cb) & 15) == 0);
    if (a.d == 8 && a.M == 1 && use_lds) {
        int R = DQ8_MAX_R;
        while (R > 1 && cb_bytes * R > (size_t)DQ_LDS_BYTES / 4) R >>= 1;
        const int tiles8 = tiles_in(a.parts, DQ8_ROWS);
        hipLaunchKernelGGL((dequant_d8_kernel<AT, CodeT>), dim3(tiles8), dim3(DQ_THREADS), cb_bytes * R, stream, codes,
                           cb, sc, make_tile_map(a.parts, DQ8_ROWS), a.G, a.K, R, out);
        return hipGetLastError();
    }
    const size_t lds = use_lds ? cb_bytes : 0;
#define PQ_DQ_LAUNCH(DD, LL)                                                                                    \
    hipLaunchKernelGGL((dequant_vec8_kernel<AT, CodeT, DD, LL>), dim3(tiles), dim3(DQ_THREADS), lds, stream, codes, \
                       cb, sc, a.parts, a.G, a.M, a.K, out)
    if (use_lds) {
        switch (a.d) {
        case 1: PQ_DQ_LAUNCH(1, true); break;
        case 2: PQ_DQ_LAUNCH(2, true); break;
        case 4: PQ_DQ_LAUNCH(4, true); break;
        default: PQ_DQ_LAUNCH(8, true); break;
        }
    } else {
        switch (a.d) {
        case 1: PQ_DQ_LAUNCH(1, false); break;
        case 2: PQ_DQ_LAUNCH(2, false); break;
        case 4: PQ_DQ_LAUNCH(4, false); break;
        default: PQ_DQ_LAUNCH(8, false); break;
        }
    }
#undef PQ_DQ_LAUNCH
    return hipGetLastError();
}

// Rows of the reconstructed weight selected by an index vector: out[t][:] = W[ids[t]][:], the embedding lookup of
// a PQ-compressed table (F.embedding over dequantize_weight's result; each element takes dequantize_weight's own
// operations in its order: sum over the codebooks in fp32, one rounding, times the row's scale, one rounding).
// One block per index; an index outside [0, N) gives a row of zeros (callers mask their indices into range).
template <typename AT, typename CodeT>
__global__ __launch_bounds__(DQ_THREADS) void embed_rows_kernel(
    const int64_t *__restrict__ ids, const CodeT *__restrict__ codes, const uint16_t *__restrict__ codebooks,
    const uint16_t *__restrict__ scales, Parts parts, int N, int G, int M, int K, int d, bool vec,
    uint16_t *__restrict__ out, int64_t out_stride) {
    const int t = blockIdx.x;
    const int64_t n = ids[t];
    const int Kin = G * d;
    uint16_t *orow = out + (size_t)t * out_stride;
    if (n < 0 || n >= N) {
        for (int k = threadIdx.x; k < Kin; k += DQ_THREADS) orow[k] = 0;
        return;
    }
    int part = 0, start = 0;
#pragma unroll
    for (int p = 0; p < PQ_MAX_PARTS; ++p) {
        if (p + 1 < parts.num && n >= start + parts.size[p]) { start += parts.size[p]; part = p + 1; }
    }
    const uint32_t kmask = (uint32_t)K - 1u;
    const uint16_t *cb = codebooks + (size_t)part * M * K * d;
    const CodeT *crow = codes + (size_t)n * G * M;
    const bool scaled = scales != nullptr;
    const float s = scaled ? Act<AT>::to_f32(scales[n]) : 1.f;
    auto element = [&](int k) -> uint16_t {
        const int g = k / d, j = k - g * d;
        uint16_t h;
        if (M == 1) {
            h = cb[(size_t)code_index(crow[g], kmask) * d + j];
        } else {
            float acc = 0.f;
            for (int m = 0; m < M; ++m) acc += Act<AT>::to_f32(cb[((size_t)m * K + code_index(crow[g * M + m], kmask)) * d + j]);
            h = Act<AT>::from_f32(acc);
        }
        if (scaled) h = Act<AT>::from_f32(Act<AT>::to_f32(h) * s);
        return h;
    };
    if (vec && d == 8 && M == 1) {   // one code, one 16-byte codebook entry (L1/L2-resident), one 16-byte store
        const bool cb_al = (reinterpret_cast<uintptr_t>(cb) & 15) == 0;
        for (int c = threadIdx.x; c < G; c += DQ_THREADS) {
            const uint16_t *e = cb + (size_t)code_index(crow[c], kmask) * 8;
            u32x4 v;
            if (cb_al) {
                v = *reinterpret_cast<const u32x4 *>(e);
            } else {
                v.x = e[0] | ((uint32_t)e[1] << 16); v.y = e[2] | ((uint32_t)e[3] << 16);
                v.z = e[4] | ((uint32_t)e[5] << 16); v.w = e[6] | ((uint32_t)e[7] << 16);
            }
            if (scaled) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint16_t lo = Act<AT>::from_f32(Act<AT>::to_f32((uint16_t)v[k]) * s);
                    const uint16_t hi = Act<AT>::from_f32(Act<AT>::to_f32((uint16_t)(v[k] >> 16)) * s);
                    v[k] = lo | ((uint32_t)hi << 16);
                }
            }
            __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(orow + (size_t)c * 8));
        }
    } else if (vec) {   // 16 bytes per thread-step: Kin % 8 == 0 and 16-byte aligned rows, checked on the host
        for (int c = threadIdx.x; c < Kin / 8; c += DQ_THREADS) {
            uint16_t h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = element(c * 8 + j);
            u32x4 v;
            v.x = h[0] | ((uint32_t)h[1] << 16);
            v.y = h[2] | ((uint32_t)h[3] << 16);
            v.z = h[4] | ((uint32_t)h[5] << 16);
            v.w = h[6] | ((uint32_t)h[7] << 16);
            *reinterpret_cast<u32x4 *>(orow + (size_t)c * 8) = v;
        }
    } else {
        for (int k = threadIdx.x; k < Kin; k += DQ_THREADS) orow[k] = element(k);
    }
}

hipError_t launch_embed_rows(const DequantArgs &a, const int64_t *ids, int T, int64_t out_stride, hipStream_t stream) {
    if (T == 0 || a.G == 0) return hipSuccess;
    const uint16_t *cb = static_cast<const uint16_t *>(a.codebooks);
    const uint16_t *sc = static_cast<const uint16_t *>(a.scales);
    uint16_t *out = static_cast<uint16_t *>(a.out);
    const bool vec = ((a.G * a.d) % 8 == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) && (out_stride % 8 == 0);
#define PQ_EMBED_LAUNCH(AT, CT)                                                                                     \
    hipLaunchKernelGGL((embed_rows_kernel<AT, CT>), dim3(T), dim3(DQ_THREADS), 0, stream, ids,                      \
                       static_cast<const CT *>(a.codes), cb, sc, a.parts, a.N, a.G, a.M, a.K, a.d, vec, out, out_stride)
    if (a.dtype == PQ_DTYPE_BF16) {
        if (a.code_bytes == 1) PQ_EMBED_LAUNCH(bf16_t, int8_t);
        else PQ_EMBED_LAUNCH(bf16_t, int16_t);
    } else {
        if (a.code_bytes == 1) PQ_EMBED_LAUNCH(fp16_t, int8_t);
        else PQ_EMBED_LAUNCH(fp16_t, int16_t);
    }
#undef PQ_EMBED_LAUNCH
    return hipGetLastError();
}

hipError_t launch_dequant(const DequantArgs &a, hipStream_t stream) {
    if (a.dtype == PQ_DTYPE_BF16) {
        return a.code_bytes == 1 ? launch_dequant_t<bf16_t, int8_t>(a, stream)
                                 : launch_dequant_t<bf16_t, int16_t>(a, stream);
    }
    return a.code_bytes == 1 ? launch_dequant_t<fp16_t, int8_t>(a, stream)
                             : launch_dequant_t<fp16_t, int16_t>(a, stream);
}

}  // namespace pq
