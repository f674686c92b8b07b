// Host side of the streaming gather+MFMA kernel (format checks, dtype dispatch) and the load-time
// re-layout of the codes into the kernel-native PACKED order.
//
// PACKED layout: for every 16-row fragment f and every chunk c of CG input groups, one 1-KiB piece at
// byte offset ((f * nchunks + c) * 64 + lane) * 16; lane = (r, q) with r = row within the fragment
// (lane & 15) and q = k quad (lane >> 4).  Code position (s*CPS + u)*M + m of the lane's 16 bytes is
// codes[16 f + r][c*CG + 4*CPS*s + CPS*q + u][m] (zero past G), so that MFMA k-step s of the chunk
// covers the 32 consecutive input features starting at (c*CG)*d + 32 s.
#include <stdio.h>
#include <stdlib.h>

#include "pq_common.cuh"
#include "pq_kernels.h"
#include "pq_knobs.h"
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


// ---- tile shape --------------------------------------------------------------------------------------
// {NF weight fragments, TF token fragments} per block.  Every block re-reads its k-range of the
// activations from L2, so more rows per block means less L2 traffic for x, and more token fragments
// mean more MFMAs per gathered weight fragment; fewer rows / tokens per block mean more blocks in
// flight.  `light` formats (d = 8, one codebook, one-byte codes) need the fewest registers per fragment;
// the others (several gathers or a rounding add per operand) keep tiles small enough not to spill.
static bool stream_light_format(const GemmArgs &a) { return a.d == 8 && a.M == 1 && a.code_bytes == 1; }

// Duration model of one launch in microseconds, fitted on the MI355X for 17 <= T <= 512: the blocks run
// in rounds of one per CU; a block costs a fixed 4 us (launch, codebook image, first loads, cross-wave
// sum) plus 1.9 us per activation sub-load of a wave: with more than 16 tokens the round trips of the
// activation loads to L2, only one sub-chunk ahead, set the pace, not the code stream or the MFMAs.  A
// sub-load is 32 registers, i.e. 8 / TF k-steps of the chunk's KS.
// What a K split over blocks adds to a launch: 2 us for the count and the last block's second pass, and the fp32
// partial sums written and read again past the L2 (about 2.5 TB/s): negligible at 16 tokens, 1.7 us for 64 tokens
// of a 4096-row layer split in two.
static double ksplit_us(const GemmArgs &a, int ksplit) {
    return ksplit > 1 ? 2.0 + 2.0 * ksplit * (double)a.T * a.N * 4.0 / 2.5e6 : 0.0;
}

static double stream_model_us(const GemmArgs &a, int nf, int tf, int ksplit) {
    const int cg = chunk_groups(a.code_bytes, a.M);
    const int ks = (cg / 4) * a.d / 8;                               // k-steps per chunk
    const double nchunks = (double)((a.G + cg - 1) / cg) / ksplit;   // per block
    const long blocks = (long)tiles_in(a.parts, nf * 16) * ((a.T + 16 * tf - 1) / (16 * tf)) * ksplit;
    const double rounds = (double)((blocks + 255) / 256);
    const double per_wave = nchunks / 8.0 < 1.0 ? 1.0 : nchunks / 8.0;
    const int subk = 8 / tf < ks ? 8 / tf : ks;
    const double tall = nf > 4 ? 1.3 * (nf - 4) : 0.0;   // gathers of the fragments beyond four, per chunk
    const double fixed = stream_light_format(a) ? 4.0 : 7.0;   // measured: 3 us more for d < 8, two-byte codes, M = 2
    // (from 16 tokens on the packed layout's activation loads run with the lanes along the rows: 1.55 us per sub-load)
    // (64-token blocks: 1.3 us, measured on down_proj and o_proj at 256 tokens -- a sub-load of theirs is two k-steps)
    const double sub_us = a.packed && a.T >= 16 ? (tf == 4 ? 1.3 : 1.55) : 1.9;
    return rounds * (fixed + ksplit_us(a, ksplit) + per_wave * ((ks / subk) * sub_us + tall));
}

// The same for blocks of one token fragment (TF = 1) and NF weight fragments: 5.3 us fixed (3.8 when a short K
// leaves three quarters of the waves without a chunk, in proportion between); a wave's first
// chunk costs 1 us for its activations plus 0.5 us per weight fragment gathered (a chunk of 16 k-steps;
// shorter chunks in proportion), every further chunk of the wave 1.55 us per activation sub-load (round trips
// that nothing overlaps) + 0.2 us per fragment.  Measured at 4 <= T <= 128 on 4096 x 4096 ... 8192 x 8192 layers.
static double stream_tall_model_us(const GemmArgs &a, int nf, int ksplit) {
    const int cg = chunk_groups(a.code_bytes, a.M);
    const int ks = (cg / 4) * a.d / 8;
    const double nchunks = (double)((a.G + cg - 1) / cg) / ksplit;
    const long blocks = (long)tiles_in(a.parts, nf * 16) * ((a.T + 15) / 16) * ksplit;
    // (d = 2 with two-byte codes at a few tokens: two of its blocks share a CU -- measured on 8192 x 8192, a K split that
    // doubles the grid to 512 blocks is 20 % shorter than the unsplit launch)
    const bool pairs_share_cu = a.d == 2 && a.code_bytes == 2 && a.T <= 8;
    const double rounds = (double)((blocks + (pairs_share_cu ? 511 : 255)) / (pairs_share_cu ? 512 : 256));
    const double per_wave = nchunks / 8.0 < 1.0 ? 1.0 : nchunks / 8.0;
    const int nsub = ks > 8 ? ks / 8 : 1;   // sub-loads per chunk at TF = 1
    const double scale = ks / 16.0;
    const double ld = a.packed && a.T >= 16 ? 0.85 : 1.0;   // activation loads with the lanes along the rows
    // (a chunk of fewer than 8 k-steps has a shorter sub-load: measured on the d = 2 formats, where pricing it at the
    // full 1.55 us made a K split look worth its hand-over at 16 tokens)
    const double short_chunk = ks < 8 ? 0.5 + ks / 16.0 : 1.0;
    const double chunks_us = scale * (1.0 * ld + 0.5 * nf) + (per_wave - 1.0) * (1.55 * ld * nsub * short_chunk + scale * 0.2 * nf);
    const double busy = nchunks / 8.0 < 1.0 ? nchunks / 8.0 : 1.0;   // waves of a block that have a chunk at all
    // (a second pass over the chip costs more than the first: measured 46.6 against a modelled 41.5 us, down_proj at 256 tokens)
    return rounds * (rounds > 1.0 ? 1.1 : 1.0) * (3.3 + 2.0 * busy + ksplit_us(a, ksplit) + chunks_us);
}

// ---- K split over blocks ---------------------------------------------------------------------------
// Decode batches on a layer with few rows and a long K (down_proj: 4096 x 14336) leave two bad choices
// without it: 16-row tiles (one block per CU, but every block then reads all of x -- 16 x the bytes of its
// codes at 16 tokens) or taller tiles (less x traffic, a quarter of the CUs idle).  With the K range of a
// tile spread over `ksplit` blocks the tile can be tall and the grid full.
// `forced`: PQ_AMD_STREAM_KSPLIT (tests and tuning), 0 when unset.  Otherwise up to 4 blocks per tile when K has
// at least 24 chunks, 2 from 16, and only up to 128 tokens (larger batches fill the chip without it).
static int ksplit_limit(int T, int G, int M, int code_bytes, int *forced) {
    const int env_split = knobs().stream_ksplit;
    const int cg = chunk_groups(code_bytes, M);
    const int nchunks = (G + cg - 1) / cg;
    int s = nchunks >= 24 ? 4 : nchunks >= 16 ? 2 : 1;
    *forced = 0;
    if (env_split) s = *forced = env_split;
    if (T > 128) s = 1;
    if (s > nchunks) s = nchunks;
    s = s < 1 ? 1 : s > 8 ? 8 : s;
    if (*forced) *forced = s;
    return s;
}

bool stream_ksplit_ok(const GemmArgs &a, int nf, int tf, int ksplit) {
    if (ksplit <= 1) return ksplit == 1;
    if (!a.packed || !a.workspace || a.workspace_bytes <= WS_COUNTER_BYTES) return false;
    int forced;
    if (ksplit > ksplit_limit(a.T, a.G, a.M, a.code_bytes, &forced)) return false;
    const long tiles = (long)tiles_in(a.parts, nf * 16) * ((a.T + 16 * tf - 1) / (16 * tf));
    if (tiles * 4 > (long)WS_COUNTER_BYTES) return false;
    if ((size_t)ksplit * a.T * a.N * 4 > a.workspace_bytes - WS_COUNTER_BYTES) return false;
    return !(reinterpret_cast<uintptr_t>(a.workspace) & 15);
}

size_t stream_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes) {
    if (!stream_format_supported(d, K, M, code_bytes)) return 0;
    int forced;
    const int s = ksplit_limit(T, G, M, code_bytes, &forced);
    if (s == 1) return 0;
    // beyond the cap a launch takes the largest split whose partial sums still fit
    const size_t part = (size_t)s * T * N * 4;
    return WS_COUNTER_BYTES + (part > WS_PARTIAL_BYTES ? WS_PARTIAL_BYTES : part);
}

size_t stream_preferred_workspace_bytes(int T, int N, int G, int M, int K, int d, int code_bytes) {
    if (!stream_format_supported(d, K, M, code_bytes)) return 0;
    int forced;
    const int s = ksplit_limit(T, G, M, code_bytes, &forced);
    return s == 1 ? 0 : WS_COUNTER_BYTES + (size_t)s * T * N * 4;
}

StreamCfg choose_stream_cfg(const GemmArgs &a, double *model_us) {
    const bool light = stream_light_format(a), packed = a.packed;
    if (model_us) *model_us = 0.0;
    int forced;
    const int smax = ksplit_limit(a.T, a.G, a.M, a.code_bytes, &forced);
    auto split_for = [&](int nf, int tf, int want) { return stream_ksplit_ok(a, nf, tf, want) ? want : 1; };
    if (knobs().stream_nf) {   // PQ_AMD_STREAM_CFG = "NFxTF", for tuning
        const int nf = knobs().stream_nf, tf = knobs().stream_tf;
        {
            const bool ok1 = tf == 1 && (nf == 1 || nf == 2 || (light && (nf == 4 || (packed && (nf == 7 || nf == 8)))));
            const bool okw = packed && ((nf == 4 && tf == 4 && light) || (nf == 4 && tf == 2) || (nf == 2 && tf == 4) ||
                                        (nf == 7 && tf == 2 && light));
            if (ok1 || okw) return StreamCfg{nf, tf, split_for(nf, tf, forced ? forced : 1)};
        }
    }
    if (!packed || a.T <= 16) {
        // One token fragment: the shape and K split with the shortest modelled launch.  In short: a grid that
        // just fits the chip in one round (one block per CU), and from 4 tokens on a long K split over blocks
        // when that keeps the grid within a round (each block then reads only its slice of x).
        static const int cand[] = {8, 7, 4, 2, 1};
        StreamCfg best{1, 1, 1};
        double best_cost = -1.0;
        for (int nf : cand) {
            if (nf > 2 && !light) continue;
            if (nf > 4 && !packed) continue;
            for (int want = 1; want <= smax; want *= 2) {
                // (below 4 tokens a split pays only where a chunk is short: formats with 64 features per chunk)
                if (forced ? want != 1 : (want > 1 && a.T < 4 && chunk_groups(a.code_bytes, a.M) * a.d > 64)) continue;
                const int ks = split_for(nf, 1, forced ? forced : want);
                const double cost = stream_tall_model_us(a, nf, ks);
                if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = StreamCfg{nf, 1, ks}; }
            }
        }
        return best;
    }
    // 17 tokens and more: the shape and split with the shortest modelled launch
    static const StreamCfg wide[] = {{4, 4, 1}, {2, 4, 1}, {4, 2, 1}, {7, 2, 1}};
    StreamCfg best = wide[2];
    double best_us = -1.0;
    const bool quad_gather = a.d == 2 && a.code_bytes == 2;   // four gathers per operand from an unreplicated image
    for (const StreamCfg &c : wide) {
        if (c.nf * c.tf > 8 && !light) continue;
        if (quad_gather && a.T > 32 && !(c.nf == 4 && c.tf == 2)) continue;   // measured: 64 x 32 tiles, by 20 - 45 %
        if (c.tf == 4 && a.T <= 32) continue;   // half the token tile would be padding
        for (int want = 1; want <= smax; want *= 2) {
            if (forced && want != 1) continue;
            const int ks = split_for(c.nf, c.tf, forced ? forced : want);
            const double us = stream_model_us(a, c.nf, c.tf, ks);
            if (best_us < 0.0 || us < best_us) { best_us = us; best = StreamCfg{c.nf, c.tf, ks}; }
        }
    }
    // ... or one token fragment per block and the batch spread over grid.y: every block then loads only 16
    // tokens of x (a wide block's waves spend most of their time loading theirs), at the price of gathering
    // each weight fragment once per 16 tokens.  Pays while the whole grid still fits the chip in one round.
    // (the formats with several gathers per operand gather too much per token that way beyond 64 tokens -- d = 2
    // with two-byte codes beyond 32: measured, tools/autotune.py --square)
    const bool no_tall = !light && (a.T > 64 || (a.T > 32 && a.d == 2 && a.code_bytes == 2));
    static const int tall[] = {8, 7, 4, 2, 1};
    for (int nf : tall) {
        if (no_tall) break;
        if (nf > 2 && !light) continue;
        for (int want = 1; want <= smax; want *= 2) {
            if (forced && want != 1) continue;
            const int ks = split_for(nf, 1, forced ? forced : want);
            const double us = stream_tall_model_us(a, nf, ks);
            if (us < best_us) { best_us = us; best = StreamCfg{nf, 1, ks}; }
        }
    }
    if (model_us) *model_us = best_us;
    return best;
}

// ---- fused gate/up + SiLU-and-mul ---------------------------------------------------------------------
constexpr int STREAM_ACT_MAX_T = 16;   // beyond: 7x2 / tiled GEMM + stand-alone silu_and_mul is faster (measured)

bool gemm_stream_act_supported(const GemmArgs &a) {
    if (!a.packed || !stream_light_format(a) || !gemm_stream_supported(a)) return false;
    if (a.parts.num != 2 || a.parts.size[0] != a.parts.size[1] || a.parts.size[0] % 16) return false;
    if ((size_t)a.K * 16 * stream_replication<8>(a.K, 1) > (size_t)ST_MAX_LDS_CB) return false;
    return a.T <= STREAM_ACT_MAX_T;
}

// NF (gate + up fragments per block) with the shortest modelled launch: rounds x (6.3 us + 0.5 us per fragment)
int choose_stream_act_nf(const GemmArgs &a) {
    static const int cand[] = {8, 4, 2};
    int best = 8;
    double best_cost = -1.0;
    for (int nf : cand) {
        const long blocks = (long)((a.parts.size[0] / 16 + nf / 2 - 1) / (nf / 2)) * ((a.T + 15) / 16);
        const double cost = (double)((blocks + 255) / 256) * (6.3 + 0.5 * nf);
        if (best_cost < 0.0 || cost < best_cost) { best_cost = cost; best = nf; }
    }
    return best;
}

hipError_t launch_gemm_stream_act_bf16(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gemm_stream_act_fp16(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gemm_stream_act(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? launch_gemm_stream_act_bf16(a, stream) : launch_gemm_stream_act_fp16(a, stream);
}

hipError_t launch_gemm_stream_bf16(const GemmArgs &a, hipStream_t stream);
hipError_t launch_gemm_stream_fp16(const GemmArgs &a, hipStream_t stream);

hipError_t launch_gemm_stream(const GemmArgs &a, hipStream_t stream) {
    if (a.N == 0 || a.T == 0) return hipSuccess;
    return a.dtype == PQ_DTYPE_BF16 ? launch_gemm_stream_bf16(a, stream) : launch_gemm_stream_fp16(a, stream);
}

// ---- re-layout ------------------------------------------------------------------------------------
bool packable(const Parts &parts, int G, int d, int K, int M, int code_bytes) {
    if (!stream_format_supported(d, K, M, code_bytes)) return false;
    if (((long)G * d) % 8) return false;        // rows of x (at least G * d long) must be able to start 16-byte aligned
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
                                                     int nchunks, uint32_t kmask, CodeT *__restrict__ packed) {
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
        out[i] = g < G ? (CodeT)code_index(codes[(n * G + g) * M + m], kmask) : (CodeT)0;   // stored as the index, mod K
    }
    CodeT *dst = packed + (piece * 64 + lane) * CPL;
#pragma unroll
    for (int i = 0; i < CPL; ++i) dst[i] = out[i];
}

hipError_t launch_repack(const void *codes, int N, int G, int M, int K, int d, int code_bytes, void *packed,
                         hipStream_t stream) {
    const int cg = chunk_groups(code_bytes, M);
    const int nchunks = (G + cg - 1) / cg;
    const long npieces = (long)(N / 16) * nchunks;
    if (npieces == 0) return hipSuccess;
    const dim3 grid((unsigned)((npieces + 3) / 4));
    if (code_bytes == 1)
        hipLaunchKernelGGL(repack_kernel<int8_t>, grid, dim3(256), 0, stream, static_cast<const int8_t *>(codes), N, G,
                           M, d, nchunks, (uint32_t)K - 1u, static_cast<int8_t *>(packed));
    else
        hipLaunchKernelGGL(repack_kernel<int16_t>, grid, dim3(256), 0, stream, static_cast<const int16_t *>(codes), N,
                           G, M, d, nchunks, (uint32_t)K - 1u, static_cast<int16_t *>(packed));
    return hipGetLastError();
}

}  // namespace pq
